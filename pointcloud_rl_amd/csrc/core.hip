// Error reporting and device queries shared by every entry point of libpcrl_hip.so.
#include "common.h"
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <vector>

namespace pcrl {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// Both caches are per DEVICE: a process that drives several GPUs (one context per device) must query / configure each of them.
constexpr int kMaxDevices = 64;

int num_cus() {
    static std::atomic<int> cus[kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 256;
    int v = cus[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        hipDeviceProp_t prop;
        v = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;   // MI355X
        cus[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

int ensure_dynamic_lds(const void* kernel, size_t bytes) {
    struct Entry { const void* kernel; size_t bytes; };
    static std::mutex mu;
    static std::vector<Entry> done[kMaxDevices];
    int dev = 0;
    PCRL_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(PCRL_E_LAUNCH, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(mu);
    for (const Entry& e : done[dev])
        if (e.kernel == kernel && e.bytes >= bytes) return PCRL_OK;
    PCRL_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    done[dev].push_back(Entry{kernel, bytes});
    return PCRL_OK;
}

}  // namespace pcrl

extern "C" const char* pcrl_last_error(void) { return pcrl::g_err; }
extern "C" int pcrl_version(void) { return 100; }   // 0.1.0
