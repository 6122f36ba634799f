// Shared helpers for the libddrl_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/ddrl.h"

namespace ddrl {

void set_error(const char *fmt, ...);

#define DDRL_HIP_CHECK(expr)                                                                  \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ::ddrl::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return DDRL_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

#define DDRL_REQUIRE(cond, msg)                                               \
    do {                                                                      \
        if (!(cond)) {                                                        \
            ::ddrl::set_error("%s:%d: bad argument: %s", __FILE__, __LINE__, msg); \
            return DDRL_ERR_BAD_ARG;                                          \
        }                                                                     \
    } while (0)

#define DDRL_LAUNCH_CHECK() DDRL_HIP_CHECK(hipGetLastError())

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// RAII device guard so that handles work with any current device.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
        active = (prev != dev);
    }
    ~DeviceGuard() { if (active && prev >= 0) (void)hipSetDevice(prev); }
    bool active = false;
};

// ---- counter-based noise: identical integer arithmetic in oracle/noise_oracle.py ----------
__host__ __device__ static inline uint32_t mix32(uint32_t x) {  // "lowbias32" finaliser
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__host__ __device__ static inline uint32_t hash3(uint32_t seed, uint32_t a, uint32_t b) {
    uint32_t h = mix32(seed ^ 0x9E3779B9u);
    h = mix32(h + a * 0x85EBCA6Bu + 0x27D4EB2Fu);
    h = mix32(h ^ (b * 0xC2B2AE35u + 0x165667B1u));
    return h;
}
// U[0,1) with 24 bits: exact in fp32
__host__ __device__ static inline float u01(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

}  // namespace ddrl
