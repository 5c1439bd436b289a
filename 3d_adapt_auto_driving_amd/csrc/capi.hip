// capi.hip -- error reporting and version of libprcnn_hip.so (see include/prcnn_hip.h).
#include "common.hpp"
#include <stdarg.h>
#include <stdio.h>

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

}  // namespace prcnn

extern "C" int prcnn_version(void) { return 100; }
extern "C" const char *prcnn_last_error(void) { return prcnn::g_err; }
