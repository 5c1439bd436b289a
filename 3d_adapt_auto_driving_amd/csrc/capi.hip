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
constexpr int SCRATCH_ENTRIES = 256;     // (stream, slot) pairs; a pipelined host uses ~5 slots on 2-8 streams
static Scratch g_scratch[SCRATCH_ENTRIES];
static int g_scratch_n = 0;
static unsigned g_scratch_rr = 0;
static std::mutex g_scratch_mu;

char *scratch_for(hipStream_t st, size_t bytes, int slot)
{
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    Scratch *s = nullptr;
    for (int i = 0; i < g_scratch_n; ++i)
        if (g_scratch[i].stream == st && g_scratch[i].slot == slot) s = &g_scratch[i];
    if (!s) {
        if (g_scratch_n == SCRATCH_ENTRIES) {
            // table full: take over the entry of the least recently created pair after draining the device
            // (its buffer may still be read by work queued on its old stream)
            (void)hipDeviceSynchronize();
            s = &g_scratch[g_scratch_rr++ % SCRATCH_ENTRIES];
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
