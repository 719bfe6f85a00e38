// Small HBM-bound elementwise kernels of the backward pass.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                      float* __restrict__ out, long n, int act) {
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const long stride = (long)gridDim.x * 1024;
    for (; i < n; i += stride) {
        if (i + 3 < n) {
            float4 d = *reinterpret_cast<const float4*>(dy + i);
            const float4 v = *reinterpret_cast<const float4*>(y + i);
            if (act == STEGAN_ACT_LRELU || act == STEGAN_ACT_RELU) {
                const float ns = act == STEGAN_ACT_LRELU ? 0.1f : 0.f;
                d.x *= v.x > 0.f ? 1.f : ns; d.y *= v.y > 0.f ? 1.f : ns;
                d.z *= v.z > 0.f ? 1.f : ns; d.w *= v.w > 0.f ? 1.f : ns;
            } else {
                d.x *= 1.f - v.x * v.x; d.y *= 1.f - v.y * v.y;
                d.z *= 1.f - v.z * v.z; d.w *= 1.f - v.w * v.w;
            }
            *reinterpret_cast<float4*>(out + i) = d;
        } else {
            for (long j = i; j < n; ++j) {
                const float v = y[j];
                const float ns = act == STEGAN_ACT_LRELU ? 0.1f : 0.f;
                out[j] = dy[j] * (act == STEGAN_ACT_TANH ? (1.f - v * v) : (v > 0.f ? 1.f : ns));
            }
        }
    }
}

}  // namespace

extern "C" int stegan_act_bwd(const float* dy, const float* y, float* out, int64_t n, int32_t act, void* stream) {
    STEGAN_CHECK_ARG(dy && y && out && n > 0);
    STEGAN_CHECK_ARG(act == STEGAN_ACT_LRELU || act == STEGAN_ACT_TANH || act == STEGAN_ACT_RELU);
    STEGAN_CHECK_ARG(((uintptr_t)dy | (uintptr_t)y | (uintptr_t)out) % 16 == 0);
    long blocks = (n + 1023) / 1024;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(act_bwd_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, dy, y, out, (long)n, act);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}
