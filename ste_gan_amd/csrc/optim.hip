// Fused AdamW over a flat fp32 parameter arena (one launch per network, HBM-bound: 4 reads + 3 writes
// per element = 28 B/param).  Step-dependent scalars live in device memory so the launch can be
// replayed from a hipGraph.
#include "common.h"

namespace {

// hyper = {lr, 1 - beta1^t, 1 - beta2^t, t (int32 bits)}.  The step counter lives on the device and is advanced by a
// kernel, not by a host write: a stream may run many steps (or graph replays) behind the host, and an async copy from
// a host staging word would read whatever the host has written by the time the copy executes.
// `scaler` (nullable; stegan_grad_scaler_*): [3] != 0 <=> this optimiser step is skipped (non-finite gradients)
__global__ void adamw_advance_kernel(float* hyper, double beta1, double beta2, const float* scaler) {
    if (scaler && scaler[3] != 0.f) return;              // a skipped step does not count (torch: optimizer.step() not called)
    int* ti = reinterpret_cast<int*>(hyper + 3);
    const int t = *ti + 1;
    *ti = t;
    hyper[1] = (float)(1.0 - pow(beta1, (double)t));
    hyper[2] = (float)(1.0 - pow(beta2, (double)t));
}

// zero_g: the gradient arena is zeroed as it is consumed (the next step's zero_grad() then costs no launch: two fills of
// 47 + 94 MB sat at the very start of every step, on its critical path)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long n,
                                                    const float* __restrict__ hyper, float beta1, float beta2,
                                                    float eps, float wd, float grad_scale, int zero_g,
                                                    const float* __restrict__ scaler) {
    if (scaler) {
        if (scaler[3] != 0.f) {                          // skipped: parameters and moments stay; the gradients are consumed
            if (zero_g) {
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i + 3 < n; i += (long)gridDim.x * 1024)
                    *reinterpret_cast<float4*>(g + i) = z;
                if (blockIdx.x == 0 && threadIdx.x < (n & 3)) g[(n & ~3L) + threadIdx.x] = 0.f;
            }
            return;
        }
        grad_scale *= scaler[1];                         // unscale: 1 / loss scale
    }
    const float lr = hyper[0], bc1 = hyper[1], bc2 = hyper[2];
    const float decay = 1.f - lr * wd;
    const float step = lr / bc1;
    const float inv_sqrt_bc2 = 1.f / sqrtf(bc2);
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const long stride = (long)gridDim.x * 1024;
    for (; i + 3 < n; i += stride) {
        float4 pp = *reinterpret_cast<float4*>(p + i);
        float4 gg = *reinterpret_cast<const float4*>(g + i);
        float4 mm = *reinterpret_cast<float4*>(m + i);
        float4 vv = *reinterpret_cast<float4*>(v + i);
        float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = G[k] * grad_scale;
            P[k] *= decay;
            M[k] = M[k] + (gk - M[k]) * (1.f - beta1);           // torch: exp_avg.lerp_(grad, 1-beta1)
            V[k] = V[k] * beta2 + gk * gk * (1.f - beta2);
            const float denom = sqrtf(V[k]) * inv_sqrt_bc2 + eps;
            P[k] -= step * (M[k] / denom);
        }
        *reinterpret_cast<float4*>(p + i) = pp;
        *reinterpret_cast<float4*>(m + i) = mm;
        *reinterpret_cast<float4*>(v + i) = vv;
        if (zero_g) *reinterpret_cast<float4*>(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // tail (n % 4) handled by the first threads of block 0
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long j = (n & ~3L) + threadIdx.x;
        const float gk = g[j] * grad_scale;
        float pj = p[j] * decay;
        const float mj = m[j] + (gk - m[j]) * (1.f - beta1);
        const float vj = v[j] * beta2 + gk * gk * (1.f - beta2);
        pj -= step * (mj / (sqrtf(vj) * inv_sqrt_bc2 + eps));
        p[j] = pj; m[j] = mj; v[j] = vj;
        if (zero_g) g[j] = 0.f;
    }
}

}  // namespace

extern "C" int stegan_adamw_advance(float* hyper_dev, double beta1, double beta2, void* stream) {
    STEGAN_CHECK_ARG(hyper_dev && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0);
    hipLaunchKernelGGL(adamw_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, hyper_dev, beta1, beta2,
                       (const float*)nullptr);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

namespace {

// ---- loss scaling on the device: torch.cuda.amp.GradScaler semantics (ste_gan/train.py:151,198-199,266-268) ----------------
// state[8] floats: [0] scale S, [1] 1 / S, [2] growth tracker (int bits), [3] non-finite gradients seen by the check of the
// CURRENT optimiser step, [4] ... by any check since the last update, [5] skipped optimiser steps so far (int bits).
__global__ __launch_bounds__(256) void scaler_check_kernel(const float* __restrict__ g, long n, float* __restrict__ state) {
    bool bad = false;
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const long stride = (long)gridDim.x * 1024;
    for (; i + 3 < n; i += stride) {
        const float4 v = *reinterpret_cast<const float4*>(g + i);
        // finite <=> |v| <= FLT_MAX (NaN compares false)
        bad = bad || !(fabsf(v.x) <= 3.402823466e38f) || !(fabsf(v.y) <= 3.402823466e38f) || !(fabsf(v.z) <= 3.402823466e38f) ||
              !(fabsf(v.w) <= 3.402823466e38f);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) bad = bad || !(fabsf(g[(n & ~3L) + threadIdx.x]) <= 3.402823466e38f);
    if (__any(bad) && (threadIdx.x & 63) == 0) { state[3] = 1.f; state[4] = 1.f; }     // (every writer writes the same words)
}

__global__ void scaler_update_kernel(float* state, float growth, float backoff, int interval) {
    int* tracker = reinterpret_cast<int*>(state + 2);
    int* skipped = reinterpret_cast<int*>(state + 5);
    float S = state[0];
    if (state[4] != 0.f) { S *= backoff; *tracker = 0; }
    else if (++*tracker == interval) { S *= growth; *tracker = 0; }
    state[0] = S; state[1] = 1.f / S; state[3] = 0.f; state[4] = 0.f;
    (void)skipped;
}

__global__ void scaler_begin_kernel(float* state) { state[3] = 0.f; }
__global__ void scaler_count_skip_kernel(float* state) {
    if (state[3] != 0.f) ++*reinterpret_cast<int*>(state + 5);
}

}  // namespace

extern "C" int stegan_grad_scaler_check(const float* grads, int64_t n, float* state, void* stream) {
    STEGAN_CHECK_ARG(grads && state && n > 0 && (uintptr_t)grads % 16 == 0);
    hipStream_t s = (hipStream_t)stream;
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(scaler_begin_kernel, dim3(1), dim3(1), 0, s, state);
    hipLaunchKernelGGL(scaler_check_kernel, dim3((int)blocks), dim3(256), 0, s, grads, (long)n, state);
    hipLaunchKernelGGL(scaler_count_skip_kernel, dim3(1), dim3(1), 0, s, state);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_grad_scaler_update(float* state, float growth_factor, float backoff_factor, int32_t growth_interval,
                                         void* stream) {
    STEGAN_CHECK_ARG(state && growth_factor >= 1.f && backoff_factor > 0.f && backoff_factor <= 1.f && growth_interval >= 1);
    hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, growth_factor, backoff_factor,
                       growth_interval);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_adamw_advance_s(float* hyper_dev, double beta1, double beta2, const float* scaler_state, void* stream) {
    STEGAN_CHECK_ARG(hyper_dev && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0);
    hipLaunchKernelGGL(adamw_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, hyper_dev, beta1, beta2, scaler_state);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_adamw_step_zs(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                    const float* hyper_dev, float beta1, float beta2, float eps, float weight_decay,
                                    float grad_scale, int32_t zero_grads, const float* scaler_state, void* stream);

extern "C" int stegan_adamw_step_z(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                   const float* hyper_dev, float beta1, float beta2, float eps, float weight_decay,
                                   float grad_scale, int32_t zero_grads, void* stream) {
    return stegan_adamw_step_zs(params, grads, exp_avg, exp_avg_sq, n, hyper_dev, beta1, beta2, eps, weight_decay, grad_scale,
                                zero_grads, nullptr, stream);
}

extern "C" int stegan_adamw_step_zs(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                    const float* hyper_dev, float beta1, float beta2, float eps, float weight_decay,
                                    float grad_scale, int32_t zero_grads, const float* scaler_state, void* stream) {
    STEGAN_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && hyper_dev && n > 0);
    STEGAN_CHECK_ARG(((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0);
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adamw_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                       exp_avg_sq, (long)n, hyper_dev, beta1, beta2, eps, weight_decay, grad_scale, zero_grads, scaler_state);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                 const float* hyper_dev, float beta1, float beta2, float eps, float weight_decay,
                                 float grad_scale, void* stream) {
    return stegan_adamw_step_z(params, const_cast<float*>(grads), exp_avg, exp_avg_sq, n, hyper_dev, beta1, beta2, eps,
                               weight_decay, grad_scale, 0, stream);
}
