// Shared host/device helpers for libstegan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/stegan_hip.h"

#define STEGAN_CHECK_ARG(cond) \
    do {                       \
        if (!(cond)) return STEGAN_E_ARG; \
    } while (0)

// Kernel-timestamp timing of the conv launches (stegan_kernel_timing_*, conv_bf16.hip): while it is on, the bf16 conv and
// weight-gradient launches go through hipExtLaunchKernelGGL with a start / stop event pair, which carry the dispatch's OWN
// begin / end timestamps (what the rocprofv3 kernel trace reports) -- no marker packets or dispatch gaps inside the bracket.
#include <hip/hip_ext.h>
bool stegan_timing_on();
void stegan_timing_pair(hipEvent_t* start, hipEvent_t* stop);      // a fresh pair, appended to the list of this run
#define STEGAN_TIMED_LAUNCH(kernel, grid, block, lds, stream, ...)                                             \
    do {                                                                                                        \
        if (stegan_timing_on()) {                                                                               \
            hipEvent_t e0__, e1__;                                                                              \
            stegan_timing_pair(&e0__, &e1__);                                                                   \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, e0__, e1__, 0, __VA_ARGS__);                \
        } else {                                                                                                \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                  \
        }                                                                                                       \
    } while (0)

// Experiment knobs of the host-side planners (tile shapes, slice counts, ...): read from the environment ONCE per process, at the
// first launch that consults them -- never again on the launch path.  None of them changes results; the switches that do
// (ablations, what-if skips, stamp buffers) exist in the diagnostic builds only (scripts/build_stamps.sh).
#define STEGAN_ENV(name_) ([]() -> const char* { static const char* const v__ = getenv(name_); return v__; }())

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

// Block barrier used by every kernel of this library instead of a bare __syncthreads().  hipcc lowers __syncthreads() on
// gfx950 to "s_waitcnt vmcnt(0); s_barrier" -- it does NOT wait for the issuing wave's outstanding LDS stores (lgkmcnt),
// on the memory-model assumption that the LDS of a CU performs its operations in order.  That does not hold across SIMDs
// with another kernel's waves co-resident on the CU: a consumer wave's first ds_read after the barrier was observed to
// overtake the producer wave's last ds_write_b128 before it (scripts/concurrency_stress.py: 2-10 % of bf16 conv launches
// wrong in one lane class of one wave tile whenever a second stream was busy; never when the kernel ran alone -- which is
// why every single-stream parity test passed).  The explicit lgkmcnt(0) closes it.
__device__ __forceinline__ void block_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
}

// THE HALF-PRECISION OPERAND TYPE of the matrix-core kernels: a build parameter of the library.  libstegan_hip.so is built with
// bf16 operands (BASELINE configs 3 / 4: v_mfma_f32_32x32x16_bf16), libstegan_hip_f16.so from the same sources with
// -DSTEGAN_OPERAND_F16 (BASELINE config 5, the reference's fp16 autocast: v_mfma_f32_32x32x16_f16, same rate).  Everything
// that is not the rounding of an operand -- layouts, staging, the packed-halves ReLU (the sign bit is the int16 sign in both
// formats), fp32 accumulation -- is the same code.  ("bf16" in names below means "the 2-byte operand type".)
#ifdef STEGAN_OPERAND_F16
typedef _Float16 hop_t;
#define STEGAN_MFMA_H(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_f16(a_, b_, c_, 0, 0, 0)
#define STEGAN_OPERAND_NAME "f16"
#else
typedef __bf16 hop_t;
#define STEGAN_MFMA_H(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b_, c_, 0, 0, 0)
#define STEGAN_OPERAND_NAME "bf16"
#endif

// The value that gets rounded to the operand type is the FP32 value the caller computed.  With fp16 operands hipcc would fuse
// `(half)(a * b + c)` into v_fma_mixlo_f16, which rounds the EXACT product once -- one ulp of fp16 away from
// half(float(a * b + c)) in ~1e-4 of the cases, and then the weight copies and images would no longer be "the fp32 tensor,
// rounded" (what the tests, the oracle's emulation and the fp32 masters all mean).  The empty asm pins the fp32 value.
__device__ __forceinline__ float hop_src(float f) {
#ifdef STEGAN_OPERAND_F16
    asm volatile("" : "+v"(f));
#endif
    return f;
}

// one element of a bf16 activation image [B][CB = ceil(C / 8)][T][8 channels] (include/stegan_hip.h, "ACTIVATION IMAGES"): what
// the elementwise producers write beside their fp32 output so that the conv kernel behind them stages it by LDS-DMA
__device__ __forceinline__ void img_store(unsigned short* __restrict__ img, int CB, int T, int b, int c, int t, float v) {
    img[(((size_t)b * CB + (c >> 3)) * T + t) * 8 + (c & 7)] = __builtin_bit_cast(unsigned short, (hop_t)hop_src(v));
}

// wave64 sum
__device__ static inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// db[c] += sum_{b,t} dy[b,c,t] for channel c over batch slice `slice` (per rows each).  The block's threads run over the
// flattened (b, t) pairs of its slice -- 16 bytes per lane when rows are float4 aligned -- so short rows (T = 12) keep
// every lane busy (the first version put the threads along t only: 12 of 256 lanes active, up to 119 us per call in the
// round-2 profile).  256 threads.  ONE block per channel owns db[c] (bias_grad_slices() == 1 since round 4): the sum has a
// fixed order and the add is a plain read-modify-write -- no float atomics anywhere in the gradient path, so two launches
// of one train step give bit-identical parameters (launches that add into the same db are stream ordered).
__device__ static inline void bias_grad_body(const float* __restrict__ dy, float* __restrict__ db, int B, int C, int T,
                                             int per, FastDiv fdT, int vec, int c, int slice) {
    const int b0 = slice * per, b1 = min(B, b0 + per);
    float s = 0.f;
    if (vec) {                                      // T % 4 == 0, aligned base: fdT divides by T / 4
        const int T4 = T >> 2, n = (b1 - b0) * T4;
        for (int r = threadIdx.x; r < n; r += 256) {
            const int bi = (int)fd_div((uint32_t)r, fdT), t4 = r - bi * T4;
            const float4 v = *reinterpret_cast<const float4*>(dy + ((size_t)(b0 + bi) * C + c) * T + 4 * t4);
            s += (v.x + v.y) + (v.z + v.w);
        }
    } else {
        const int n = (b1 - b0) * T;
        for (int r = threadIdx.x; r < n; r += 256) {
            const int bi = (int)fd_div((uint32_t)r, fdT), t = r - bi * T;
            s += dy[((size_t)(b0 + bi) * C + c) * T + t];
        }
    }
    __shared__ float bias_part[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) bias_part[threadIdx.x >> 6] = s;
    block_sync();
    if (threadIdx.x == 0) db[c] += (bias_part[0] + bias_part[1]) + (bias_part[2] + bias_part[3]);
}

// batch slices per channel: one.  (Rounds 1-3 cut a channel's rows into slices that met in an fp32 atomic: the order of
// those adds changed from launch to launch, and in bf16 mode a last-bit difference in a bias flips operand roundings two
// layers on -- the step was not reproducible.  A channel's B x T values are at most a few 10^4: one block sums them.)
static inline int bias_grad_slices(int B, int C, int T) {
    (void)B; (void)C; (void)T;
    return 1;
}

// conv_wgrad.hip: direct weight-gradient kernel; returns 1 when the caller should use the generic implicit GEMM
// conv_wgrad_small.hip: the 8-input-channel first layers of the discriminators (exact fp32, register-blocked); returns 1 when the
// shape is not one of them (or there is no workspace for its slabs)
int stegan_wgrad_small(const stegan_conv_desc* d, const float* x, const float* dy, float* dw, float* db, float* workspace,
                       int64_t workspace_floats, hipStream_t stream);
int stegan_wgrad_direct(const stegan_conv_desc* d, const float* x, const float* dy, float* dw, float* db,
                        float* workspace, int64_t workspace_floats, hipStream_t stream);
