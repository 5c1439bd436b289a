// capi.hip -- error reporting and version of libprcnn_hip.so (see include/prcnn_hip.h).
#include <cstring>
#include "common.hpp"
#include <stdarg.h>
#include <stdio.h>
#include <map>
#include <mutex>
#include <utility>

namespace prcnn {

static thread_local char g_err[768] = "";
static thread_local char g_note[256] = "";     // why scratch_for() just refused (its callers report "cannot allocate" in their own words)

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    const int n = vsnprintf(g_err, sizeof(g_err) - sizeof(g_note) - 4, fmt, ap);
    va_end(ap);
    if (g_note[0] && n >= 0) {
        const size_t at = strlen(g_err);
        snprintf(g_err + at, sizeof(g_err) - at, " [%s]", g_note);
        g_note[0] = 0;
    }
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
    bool captured;                       // a hipGraph captured on this stream holds `ptr`: the buffer may never move again
};
constexpr int SCRATCH_ENTRIES = 256;     // (stream, slot) pairs; a pipelined host uses ~5 slots on 2-8 streams
static Scratch g_scratch[SCRATCH_ENTRIES];
static int g_scratch_n = 0;
static unsigned g_scratch_rr = 0;
static std::mutex g_scratch_mu;

char *scratch_for(hipStream_t st, size_t bytes, int slot, bool *fresh)
{
    if (fresh) *fresh = false;
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
            // -- never one that a captured hipGraph points into: its buffer would be shared with another stream and freed on growth
            // under the graph's next replay
            (void)hipDeviceSynchronize();
            for (int tries = 0; tries < SCRATCH_ENTRIES; ++tries) {
                Scratch *c = &g_scratch[g_scratch_rr++ % SCRATCH_ENTRIES];
                if (!c->captured) { s = c; break; }
            }
            if (!s) {
                snprintf(g_note, sizeof(g_note), "scratch: all %d (stream, slot) entries are held by captured graphs", SCRATCH_ENTRIES);
                return nullptr;
            }
            if (fresh) *fresh = true;              // the new owner must not trust what the previous one left in the buffer
            if (s->device != dev && s->ptr) {      // the old buffer lives on another device: free it there
                (void)hipSetDevice(s->device);
                (void)hipDeviceSynchronize();
                (void)hipFree(s->ptr);
                (void)hipSetDevice(dev);
                s->ptr = nullptr; s->bytes = 0;
            }
            s->device = dev; s->stream = st; s->slot = slot; s->captured = false;
        } else {
            s = &g_scratch[g_scratch_n++];
            s->device = dev; s->stream = st; s->slot = slot; s->ptr = nullptr; s->bytes = 0; s->captured = false;
        }
    }
    // hipGraph capture (eval_rcnn.GraphedRunner): the captured launches keep this pointer for the life of the graph.  The stream must
    // have been warmed up with the same shapes BEFORE the capture (no hipMalloc inside a capture).  A later, larger request on the
    // same stream gets a NEW buffer and the old one is left to the graphs that point to it (never freed: a few MB per growth, and a
    // process captures for a handful of shapes) -- launches on one stream are ordered, so the two never serve the same call.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    if (s->bytes < bytes && capturing) {
        snprintf(g_note, sizeof(g_note), "scratch: %zu bytes wanted, %zu held (slot %d): the stream is being captured -- run the same "
                 "calls once on it before the capture", bytes, s->bytes, slot);
        return nullptr;
    }
    if (capturing) s->captured = true;
    if (s->bytes < bytes) {
        if (s->ptr && !s->captured) { (void)hipStreamSynchronize(st); (void)hipFree(s->ptr); }
        s->ptr = nullptr; s->bytes = 0; s->captured = false;
        if (hipMalloc((void **)&s->ptr, bytes) != hipSuccess) return nullptr;
        s->bytes = bytes;
        if (fresh) *fresh = true;
    }
    return s->ptr;
}


}  // namespace prcnn

extern "C" int prcnn_version(void) { return 100; }
extern "C" const char *prcnn_last_error(void) { return prcnn::g_err; }
