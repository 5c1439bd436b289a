// capi.hip -- error reporting and version of libprcnn_hip.so (see include/prcnn_hip.h).
#include "common.hpp"
#include <stdarg.h>
#include <stdio.h>
#include <map>
#include <mutex>
#include <utility>

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

int mfma_grid_cap()
{
    static const int cap = [] { const char *e = getenv("PRCNN_MFMA_GRID"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 512; }();
    return cap;
}

int current_device()
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    return dev;
}

// ---- dynamic-LDS limit of a kernel, raised once per (device, kernel): function attributes are per device
static std::map<std::pair<int, const void *>, size_t> g_lds_cfg;
static std::mutex g_lds_mu;

int ensure_dynamic_lds(const void *kernel, size_t bytes, const char *what)
{
    const int dev = current_device();
    std::lock_guard<std::mutex> lock(g_lds_mu);
    size_t &have = g_lds_cfg[std::make_pair(dev, kernel)];
    if (bytes <= have) return PRCNN_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        (void)hipGetLastError();
        set_error("%s: cannot reserve %zu bytes of LDS", what, bytes);
        return PRCNN_ELAUNCH;
    }
    have = bytes;
    return PRCNN_OK;
}

// ---- per-stream scratch ---------------------------------------------------------------------
struct Scratch {
    int device;
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
    // keyed by DEVICE as well: the default stream handle (0) is the same on every GPU of a process
    const int dev = current_device();
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    Scratch *s = nullptr;
    for (int i = 0; i < g_scratch_n; ++i)
        if (g_scratch[i].device == dev && g_scratch[i].stream == st && g_scratch[i].slot == slot) s = &g_scratch[i];
    if (!s) {
        if (g_scratch_n == SCRATCH_ENTRIES) {
            // table full: take over the entry of the least recently created pair after draining the device
            // (its buffer may still be read by work queued on its old stream)
            (void)hipDeviceSynchronize();
            s = &g_scratch[g_scratch_rr++ % SCRATCH_ENTRIES];
            if (s->device != dev && s->ptr) {      // the old buffer lives on another device: free it there
                (void)hipSetDevice(s->device);
                (void)hipDeviceSynchronize();
                (void)hipFree(s->ptr);
                (void)hipSetDevice(dev);
                s->ptr = nullptr; s->bytes = 0;
            }
            s->device = dev; s->stream = st; s->slot = slot;
        } else {
            s = &g_scratch[g_scratch_n++];
            s->device = dev; s->stream = st; s->slot = slot; s->ptr = nullptr; s->bytes = 0;
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
