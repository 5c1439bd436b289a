// capi.hip -- error reporting and version of libprcnn_hip.so (see include/prcnn_hip.h).
#include "common.hpp"
#include <stdarg.h>
#include <stdio.h>
#include <mutex>

namespace prcnn {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return PRCNN_ELAUNCH;
    }
    return PRCNN_OK;
}

// ---- per-stream scratch ---------------------------------------------------------------------
struct Scratch {
    hipStream_t stream;
    int slot;
    char *ptr;
    size_t bytes;
};
static Scratch g_scratch[16];
static int g_scratch_n = 0;
static std::mutex g_scratch_mu;

char *scratch_for(hipStream_t st, size_t bytes, int slot)
{
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    Scratch *s = nullptr;
    for (int i = 0; i < g_scratch_n; ++i)
        if (g_scratch[i].stream == st && g_scratch[i].slot == slot) s = &g_scratch[i];
    if (!s) {
        if (g_scratch_n == 16) {   // recycle the first slot (its stream must be idle by contract)
            s = &g_scratch[0];
            (void)hipStreamSynchronize(s->stream);
            s->stream = st; s->slot = slot;
        } else {
            s = &g_scratch[g_scratch_n++];
            s->stream = st; s->slot = slot; s->ptr = nullptr; s->bytes = 0;
        }
    }
    if (s->bytes < bytes) {
        if (s->ptr) { (void)hipStreamSynchronize(st); (void)hipFree(s->ptr); }
        s->ptr = nullptr; s->bytes = 0;
        if (hipMalloc((void **)&s->ptr, bytes) != hipSuccess) return nullptr;
        s->bytes = bytes;
    }
    return s->ptr;
}


}  // namespace prcnn

extern "C" int prcnn_version(void) { return 100; }
extern "C" const char *prcnn_last_error(void) { return prcnn::g_err; }
