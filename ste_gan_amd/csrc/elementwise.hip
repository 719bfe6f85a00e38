// Small HBM-bound elementwise kernels of the backward pass.
#include "common.h"

namespace {

// add != nullptr: a second gradient of the same tensor (the feature-matching term of a discriminator feature map, known
// since the forward pass) is summed in here instead of by a separate elementwise kernel in front of this one
// out_h != nullptr: the result is also written as a bf16 activation image [B][C / 8][T][8] (C % 8 == 0; any T since round 4)
// for the conv kernels that consume this gradient next
__device__ __forceinline__ void act_bwd_body(const float* __restrict__ dy, const float* __restrict__ add,
                                             const float* __restrict__ y, float* __restrict__ out, long n, int act,
                                             unsigned short* __restrict__ out_h = nullptr, int Cc = 0, int T = 0) {
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const long stride = (long)gridDim.x * 1024;
    for (; i < n; i += stride) {
        if (i + 3 < n) {
            float4 d = *reinterpret_cast<const float4*>(dy + i);
            if (add) {
                const float4 e = *reinterpret_cast<const float4*>(add + i);
                d.x += e.x; d.y += e.y; d.z += e.z; d.w += e.w;
            }
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
            if (out_h) {
                const long r = i / T;
                const int t = (int)(i - r * T), b = (int)(r / Cc), c = (int)(r - (long)b * Cc);
                if (t + 3 < T) {                           // the four values lie in one row (always when T % 4 == 0)
                    img_store(out_h, Cc >> 3, T, b, c, t, d.x); img_store(out_h, Cc >> 3, T, b, c, t + 1, d.y);
                    img_store(out_h, Cc >> 3, T, b, c, t + 2, d.z); img_store(out_h, Cc >> 3, T, b, c, t + 3, d.w);
                } else {                                   // ragged rows (the period discriminators' lengths): they straddle
                    const float dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const long re = (i + e) / T;
                        const int te = (int)(i + e - re * T), be = (int)(re / Cc), ce = (int)(re - (long)be * Cc);
                        img_store(out_h, Cc >> 3, T, be, ce, te, dv[e]);
                    }
                }
            }
        } else {
            for (long j = i; j < n; ++j) {
                const float v = y[j];
                const float ns = act == STEGAN_ACT_LRELU ? 0.1f : 0.f;
                const float o = (dy[j] + (add ? add[j] : 0.f)) * (act == STEGAN_ACT_TANH ? (1.f - v * v) : (v > 0.f ? 1.f : ns));
                out[j] = o;
                if (out_h) {
                    const long rj = j / T;
                    img_store(out_h, Cc >> 3, T, (int)(rj / Cc), (int)(rj % Cc), (int)(j - rj * T), o);
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ add,
                                                      const float* __restrict__ y, float* __restrict__ out, long n, int act,
                                                      unsigned short* __restrict__ out_h, int Cc, int T) {
    act_bwd_body(dy, add, y, out, n, act, out_h, Cc, T);
}

// the same for up to ACT_MULTI tensors in one launch (layer L of the five period discriminators): blockIdx.y = tensor
constexpr int ACT_MULTI = 8;
struct ActMulti {
    const float* dy[ACT_MULTI]; const float* add[ACT_MULTI]; const float* y[ACT_MULTI]; float* out[ACT_MULTI]; long n[ACT_MULTI];
    unsigned short* out_h[ACT_MULTI]; int Cc[ACT_MULTI], T[ACT_MULTI];
};

__global__ __launch_bounds__(256) void act_bwd_multi_kernel(const ActMulti a, int act) {
    const int k = blockIdx.y;
    act_bwd_body(a.dy[k], a.add[k], a.y[k], a.out[k], a.n[k], act, a.out_h[k], a.Cc[k], a.T[k]);
}

}  // namespace

extern "C" int stegan_act_bwd_add_img(const float* dy, const float* add, const float* y, float* out, uint16_t* out_h,
                                      int64_t n, int32_t C, int32_t T, int32_t act, void* stream) {
    STEGAN_CHECK_ARG(dy && y && out && n > 0);
    if (out_h) STEGAN_CHECK_ARG(C > 0 && T > 0 && C % 8 == 0 && n % ((int64_t)C * T) == 0);
    STEGAN_CHECK_ARG(act == STEGAN_ACT_LRELU || act == STEGAN_ACT_TANH || act == STEGAN_ACT_RELU);
    STEGAN_CHECK_ARG(((uintptr_t)dy | (uintptr_t)add | (uintptr_t)y | (uintptr_t)out) % 16 == 0);
    long blocks = (n + 1023) / 1024;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(act_bwd_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, dy, add, y, out, (long)n, act,
                       out_h, C, T);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_act_bwd_add(const float* dy, const float* add, const float* y, float* out, int64_t n, int32_t act,
                                  void* stream) {
    return stegan_act_bwd_add_img(dy, add, y, out, nullptr, n, 0, 0, act, stream);
}

extern "C" int stegan_act_bwd_add_multi_img(int32_t m, const float* const* dy, const float* const* add, const float* const* y,
                                            float* const* out, uint16_t* const* out_h, const int64_t* n, const int32_t* Cs,
                                            const int32_t* Ts, int32_t act, void* stream) {
    STEGAN_CHECK_ARG(m >= 1 && m <= ACT_MULTI && dy && y && out && n);
    STEGAN_CHECK_ARG(!out_h || (Cs && Ts));
    STEGAN_CHECK_ARG(act == STEGAN_ACT_RELU || act == STEGAN_ACT_LRELU || act == STEGAN_ACT_TANH);
    ActMulti a;
    long blocks = 1;
    for (int k = 0; k < m; ++k) {
        STEGAN_CHECK_ARG(dy[k] && y[k] && out[k] && n[k] > 0);
        a.dy[k] = dy[k]; a.add[k] = add ? add[k] : nullptr; a.y[k] = y[k]; a.out[k] = out[k]; a.n[k] = (long)n[k];
        a.out_h[k] = out_h ? out_h[k] : nullptr; a.Cc[k] = out_h ? Cs[k] : 0; a.T[k] = out_h ? Ts[k] : 0;
        if (a.out_h[k]) STEGAN_CHECK_ARG(a.Cc[k] > 0 && a.T[k] > 0 && a.Cc[k] % 8 == 0 && n[k] % ((int64_t)a.Cc[k] * a.T[k]) == 0);
        STEGAN_CHECK_ARG(((uintptr_t)a.dy[k] | (uintptr_t)a.add[k] | (uintptr_t)a.y[k] | (uintptr_t)a.out[k]) % 16 == 0);
        const long b = ((long)n[k] + 1023) / 1024;
        if (b > blocks) blocks = b;
    }
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(act_bwd_multi_kernel, dim3((unsigned)blocks, (unsigned)m), dim3(256), 0, (hipStream_t)stream, a, act);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_act_bwd_add_multi(int32_t m, const float* const* dy, const float* const* add, const float* const* y,
                                        float* const* out, const int64_t* n, int32_t act, void* stream) {
    return stegan_act_bwd_add_multi_img(m, dy, add, y, out, nullptr, n, nullptr, nullptr, act, stream);
}

extern "C" int stegan_act_bwd(const float* dy, const float* y, float* out, int64_t n, int32_t act, void* stream) {
    return stegan_act_bwd_add(dy, nullptr, y, out, n, act, stream);
}

// ---- generator input (round 4): session-embedding gather + broadcast over time + concat + 'b t c -> b c t' in ONE launch
// (generator.py:143-154; rounds 1-3 left these to ATen: gather, expand, cat, transpose-copy, and five launches backward).
//   out[b, c, t] = c < F ? x[b, t, c] : (c < F + E ? w1[id1[b], c - F] : w2[id2[b], c - F - E])        (B, F + E (+ E), T)
// and the bf16 image of out when the channel count is a multiple of 8.  Tiles of 32 time steps x 32 channels go through
// LDS so that both the (b, t, c) reads and the (b, c, t) writes are coalesced.
namespace {
__global__ __launch_bounds__(256) void embed_concat_fwd_kernel(const float* __restrict__ x, const long long* __restrict__ id1,
                                                               const float* __restrict__ w1, const long long* __restrict__ id2,
                                                               const float* __restrict__ w2, float* __restrict__ out,
                                                               unsigned short* __restrict__ out_h, int B, int T, int F, int E,
                                                               int S1, int S2) {
    __shared__ float tile[32][33];
    const int Ct = F + E + (w2 ? E : 0);
    const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;               // 32 x 8 threads
    for (int r = ly; r < 32; r += 8) {                                     // r: time step of the tile, lx: channel
        const int t = t0 + r, c = c0 + lx;
        float v = 0.f;
        if (t < T && c < Ct) {
            if (c < F) v = x[((size_t)b * T + t) * F + c];
            else if (c < F + E) { const long long s = id1[b]; v = (s >= 0 && s < S1) ? w1[s * E + (c - F)] : 0.f; }
            else { const long long s = id2[b]; v = (s >= 0 && s < S2) ? w2[s * E + (c - F - E)] : 0.f; }
        }
        tile[r][lx] = v;
    }
    block_sync();
    for (int r = ly; r < 32; r += 8) {                                     // r: channel of the tile, lx: time step
        const int c = c0 + r, t = t0 + lx;
        if (c < Ct && t < T) {
            const float v = tile[lx][r];
            out[((size_t)b * Ct + c) * T + t] = v;
            if (out_h) img_store(out_h, Ct >> 3, T, b, c, t, v);
        }
    }
}

// dW[s, j] += sum over the batch rows with id[b] == s, over t, of dout[b, c_off + j, t]: one block per table row s, one
// thread per embedding column j, the batch walked in order -- a fixed summation order (deterministic), no atomics.
__global__ __launch_bounds__(64) void embed_concat_bwd_kernel(const float* __restrict__ dout, const long long* __restrict__ id,
                                                              float* __restrict__ dw, int B, int T, int Ct, int c_off, int E) {
    const int s = blockIdx.x;
    for (int j = threadIdx.x; j < E; j += 64) {
        float acc = 0.f;
        bool any = false;
        for (int b = 0; b < B; ++b) {
            if (id[b] != s) continue;
            any = true;
            const float* row = dout + ((size_t)b * Ct + c_off + j) * T;
            float a = 0.f;
            for (int t = 0; t < T; ++t) a += row[t];
            acc += a;
        }
        if (any) dw[(size_t)s * E + j] += acc;
    }
}
}  // namespace

extern "C" int stegan_embed_concat_fwd(const float* x, const int64_t* id1, const float* w1, const int64_t* id2, const float* w2,
                                       float* out, uint16_t* out_h, int32_t B, int32_t T, int32_t F, int32_t E, int32_t S1,
                                       int32_t S2, void* stream) {
    STEGAN_CHECK_ARG(x && id1 && w1 && out && B > 0 && T > 0 && F > 0 && E > 0 && S1 > 0);
    STEGAN_CHECK_ARG((id2 == nullptr) == (w2 == nullptr) && (!w2 || S2 > 0));
    const int Ct = F + E + (w2 ? E : 0);
    STEGAN_CHECK_ARG(B <= 65535 && (!out_h || Ct % 8 == 0));
    hipLaunchKernelGGL(embed_concat_fwd_kernel, dim3((unsigned)ceil_div(T, 32), (unsigned)ceil_div(Ct, 32), (unsigned)B), dim3(256),
                       0, (hipStream_t)stream, x, (const long long*)id1, w1, (const long long*)id2, w2, out, out_h, B, T, F, E, S1, S2);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_embed_concat_bwd(const float* dout, const int64_t* ids, float* dw, int32_t B, int32_t T, int32_t Ct,
                                       int32_t c_off, int32_t E, int32_t S, void* stream) {
    STEGAN_CHECK_ARG(dout && ids && dw && B > 0 && T > 0 && E > 0 && S > 0 && c_off >= 0 && c_off + E <= Ct);
    hipLaunchKernelGGL(embed_concat_bwd_kernel, dim3((unsigned)S), dim3(64), 0, (hipStream_t)stream, dout, (const long long*)ids,
                       dw, B, T, Ct, c_off, E);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}
