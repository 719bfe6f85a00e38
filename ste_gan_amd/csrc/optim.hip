// Fused AdamW over a flat fp32 parameter arena (one launch per network, HBM-bound: 4 reads + 3 writes
// per element = 28 B/param).  Step-dependent scalars live in device memory so the launch can be
// replayed from a hipGraph.
#include "common.h"

namespace {

// hyper = {lr, 1 - beta1^t, 1 - beta2^t, t (int32 bits)}.  The step counter lives on the device and is advanced by a
// kernel, not by a host write: a stream may run many steps (or graph replays) behind the host, and an async copy from
// a host staging word would read whatever the host has written by the time the copy executes.
__global__ void adamw_advance_kernel(float* hyper, double beta1, double beta2) {
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
                                                    float eps, float wd, float grad_scale, int zero_g) {
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
    hipLaunchKernelGGL(adamw_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, hyper_dev, beta1, beta2);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_adamw_step_z(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                   const float* hyper_dev, float beta1, float beta2, float eps, float weight_decay,
                                   float grad_scale, int32_t zero_grads, void* stream) {
    STEGAN_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && hyper_dev && n > 0);
    STEGAN_CHECK_ARG(((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0);
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adamw_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                       exp_avg_sq, (long)n, hyper_dev, beta1, beta2, eps, weight_decay, grad_scale, zero_grads);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                 const float* hyper_dev, float beta1, float beta2, float eps, float weight_decay,
                                 float grad_scale, void* stream) {
    return stegan_adamw_step_z(params, const_cast<float*>(grads), exp_avg, exp_avg_sq, n, hyper_dev, beta1, beta2, eps,
                               weight_decay, grad_scale, 0, stream);
}
