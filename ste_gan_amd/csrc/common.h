// Shared host/device helpers for libstegan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/stegan_hip.h"

#define STEGAN_CHECK_ARG(cond) \
    do {                       \
        if (!(cond)) return STEGAN_E_ARG; \
    } while (0)

#define STEGAN_LAUNCH_CHECK()                 \
    do {                                      \
        hipError_t e__ = hipGetLastError();   \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

// Division by a runtime-invariant divisor d (1 <= d < 2^31) for n < 2^31:
//   q = (umulhi(n, mul) + n) >> shift      (Granlund & Montgomery round-up method)
struct FastDiv {
    uint32_t mul, shift, d, _pad;
};

static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    f._pad = 0;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    f.shift = l;
    f.mul = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
    return f;
}

__host__ __device__ static inline uint32_t fd_div(uint32_t n, const FastDiv& f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (__umulhi(n, f.mul) + n) >> f.shift;
#else
    return (uint32_t)((((uint64_t)n * f.mul) >> 32) + n) >> f.shift;
#endif
}

__host__ __device__ static inline void fd_divmod(uint32_t n, const FastDiv& f, uint32_t& q, uint32_t& r) {
    q = fd_div(n, f);
    r = n - q * f.d;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// wave64 sum
__device__ static inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// conv_wgrad.hip: direct weight-gradient kernel; returns 1 when the caller should use the generic implicit GEMM
int stegan_wgrad_direct(const stegan_conv_desc* d, const float* x, const float* dy, float* dw, float* db,
                        float* workspace, int64_t workspace_floats, hipStream_t stream);
