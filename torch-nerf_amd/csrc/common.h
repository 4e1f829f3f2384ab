// Shared helpers for the gfx950 kernels behind include/nerf_amd.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

#include "../../include/nerf_amd.h"

#define NERF_API extern "C" __attribute__((visibility("default")))

constexpr int WAVE = 64;  // CDNA wavefront width

namespace nerf {

char *error_buffer();  // thread-local, 256 bytes

inline int fail(int code, const char *what) {
    snprintf(error_buffer(), 256, "%s", what);
    return code;
}

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(error_buffer(), 256, "%s: %s", what, hipGetErrorString(e));
        return NERF_ERR_LAUNCH;
    }
    return NERF_OK;
}

inline hipStream_t as_stream(nerf_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Per-DEVICE lazily created state (the only mutable state of the library): the CU count of the current device
// and, per kernel, the set of device ordinals on which its dynamic-LDS limit has been raised
// (hipFuncSetAttribute applies to the current device only).  Both are idempotent, so races are benign.
int device_cus();
typedef std::atomic<unsigned long long> DeviceMask;
int ensure_dynamic_lds(const void *kernel, int bytes, DeviceMask &done, const char *what);

}  // namespace nerf

#define NERF_REQUIRE(cond, msg) \
    do {                        \
        if (!(cond)) return nerf::fail(NERF_ERR_ARG, msg); \
    } while (0)
