// Library-level entry points of libnerf_amd.so (see include/nerf_amd.h).
#include "common.h"

namespace nerf {
char *error_buffer() {
    static thread_local char buf[256] = {0};
    return buf;
}

namespace {
constexpr int MAX_DEVICES = 64;
std::atomic<int> cu_count[MAX_DEVICES];
}  // namespace

int device_cus() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return 256;
    int cus = cu_count[dev].load(std::memory_order_relaxed);
    if (cus == 0) {
        hipDeviceProp_t prop;
        cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
        cu_count[dev].store(cus, std::memory_order_relaxed);
    }
    return cus;
}

int ensure_dynamic_lds(const void *kernel, int bytes, DeviceMask &done, const char *what) {
    int dev = 0;
    const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < MAX_DEVICES;
    if (known && (done.load(std::memory_order_relaxed) >> dev & 1ull)) return NERF_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
        return check_launch(what);
    if (known) done.fetch_or(1ull << dev, std::memory_order_relaxed);
    return NERF_OK;
}
}  // namespace nerf

NERF_API int nerf_amd_abi_version(void) { return NERF_AMD_ABI_VERSION; }
NERF_API const char *nerf_amd_last_error(void) { return nerf::error_buffer(); }
