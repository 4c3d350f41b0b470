// Error reporting and device queries shared by every entry point of libpcrl_hip.so.
#include "common.h"
#include <cstdarg>
#include <cstdio>

namespace pcrl {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int num_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
        else
            cus = 256;   // MI355X
    }
    return cus;
}

}  // namespace pcrl

extern "C" const char* pcrl_last_error(void) { return pcrl::g_err; }
extern "C" int pcrl_version(void) { return 100; }   // 0.1.0
