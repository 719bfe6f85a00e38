// Loss kernels: every loss produces its scalar (accumulated into a device "loss slot") AND the
// gradient w.r.t. its input in the same pass -- the reductions are wavefront shuffles + one atomic
// per block, nothing is re-read.  All fp32.
#include "common.h"
#include <vector>

namespace {

// Loss scaling (stegan_loss_grad_scale): while set, every GRADIENT a loss entry point writes is multiplied by *g_gscale, read
// on the device when the kernel runs (so a captured graph follows the scaler's updates); the loss VALUES are not scaled.
const float* g_gscale = nullptr;

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    block_sync();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    block_sync();
    return red[0] + red[1] + red[2] + red[3];
}

// mode 0: L1 |a-b|, mode 1: MSE (a-b)^2;  b == nullptr -> constant target
// b == nullptr: constant target -- `target` for elements [0, split), `target2` from `split` on (the discriminator step's
// batched fake | real logits: zeros for the first half of the batch, ones for the second)
__device__ __forceinline__ void pair_loss_body(const float* __restrict__ a, const float* __restrict__ b, float target, long n,
                                               int mode, float w_over_n, float* __restrict__ loss_slot,
                                               float* __restrict__ da, int vec, long bid, long nblocks,
                                               const float* __restrict__ gscale,
                                               long split = (1l << 62), float target2 = 0.f,
                                               float* __restrict__ part = nullptr) {
    __shared__ float red[4];
    float s = 0.f;
    const float gw = gscale ? w_over_n * *gscale : w_over_n;      // the gradient's factor (loss scaling); the value keeps w_over_n
    // HBM-bound (read a, b; write da): 16 bytes per lane when the three arrays are 16-byte aligned
    const long n4 = vec ? (n >> 2) : 0;
    for (long i = bid * 256 + threadIdx.x; i < n4; i += nblocks * 256) {
        const float4 av = reinterpret_cast<const float4*>(a)[i];
        const long e0 = i << 2;
        float4 bv = make_float4(e0 < split ? target : target2, e0 + 1 < split ? target : target2,
                                e0 + 2 < split ? target : target2, e0 + 3 < split ? target : target2);
        if (b) bv = reinterpret_cast<const float4*>(b)[i];
        const float d[4] = {av.x - bv.x, av.y - bv.y, av.z - bv.z, av.w - bv.w};
        float g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (mode == 0) { s += fabsf(d[k]); g[k] = d[k] > 0.f ? gw : (d[k] < 0.f ? -gw : 0.f); }
            else { s += d[k] * d[k]; g[k] = 2.f * d[k] * gw; }
        }
        if (da) reinterpret_cast<float4*>(da)[i] = make_float4(g[0], g[1], g[2], g[3]);
    }
    for (long i = (n4 << 2) + bid * 256 + threadIdx.x; i < n; i += nblocks * 256) {
        const float d = a[i] - (b ? b[i] : (i < split ? target : target2));
        if (mode == 0) {
            s += fabsf(d);
            if (da) da[i] = d > 0.f ? gw : (d < 0.f ? -gw : 0.f);
        } else {
            s += d * d;
            if (da) da[i] = 2.f * d * gw;
        }
    }
    s = block_sum_256(s, red);
    // the block's share of the value: a slot of its own when the caller gave room for them (summed in a fixed order by
    // pair_loss_sum_kernel: two launches of a step report the same bits), else an atomic add into the loss word
    if (threadIdx.x == 0) {
        if (part) *part = s * w_over_n;
        else if (loss_slot) atomicAdd(loss_slot, s * w_over_n);
    }
}

// *loss_slot += sum of the blocks' shares, in a fixed order (thread t takes shares t, t + 256, ...; then the block's tree)
__global__ __launch_bounds__(256) void pair_loss_sum_kernel(const float* __restrict__ part, int n, float* __restrict__ loss_slot) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) *loss_slot += s;
}

__global__ __launch_bounds__(256) void pair_loss_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        float target, long n, int mode, float w_over_n,
                                                        float* __restrict__ loss_slot, float* __restrict__ da, int vec,
                                                        const float* __restrict__ gscale) {
    pair_loss_body(a, b, target, n, mode, w_over_n, loss_slot, da, vec, blockIdx.x, gridDim.x, gscale);
}

// Up to PL_MULTI pairs in ONE launch (the ~35 terms of the feature-matching sum, train.py:257-262: one after the other
// their launches were 0.35 ms of a 12 ms step, each a few microseconds of work): pair i owns blocks first[i]..first[i+1].
constexpr int PL_MULTI = 64;
struct PairMulti {
    int n, mode;
    float* loss_slot;
    float* part;                  // nullable: a share per block (deterministic value), block b of the launch writes part[b]
    const float* gscale;
    int first[PL_MULTI + 1];
    struct E { const float* a; const float* b; float* da; long n, split; float w_over_n, target, target2; int vec; } e[PL_MULTI];
};
static_assert(sizeof(PairMulti) <= 4096, "kernel argument segment limit");

__global__ __launch_bounds__(256) void pair_loss_multi_kernel(const PairMulti m) {
    int lo = 0, hi = m.n - 1;                       // last i with first[i] <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (m.first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const PairMulti::E& e = m.e[lo];
    pair_loss_body(e.a, e.b, e.target, e.n, m.mode, e.w_over_n, m.loss_slot, e.da, e.vec, (long)blockIdx.x - m.first[lo],
                   (long)m.first[lo + 1] - m.first[lo], m.gscale, e.split, e.target2, m.part ? m.part + blockIdx.x : nullptr);
}

__global__ __launch_bounds__(256) void scale_by_scalar_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                              float* __restrict__ out, long n) {
    const float k = *s;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = x[i] * k;
}

// ---------------------------------------------------------------------------------------------
// Multi-resolution time-domain feature loss.  One block per (batch, channel) row of length T.
// ---------------------------------------------------------------------------------------------
struct TDScale { int W, S, P, F; float inv_n; };   // window, shift, pad = W/2, frames, 1/(B*F*C*4)

__device__ __forceinline__ int refl(int t, int T) {        // reflect without edge repeat
    if (t < 0) t = -t;
    if (t >= T) t = 2 * T - 2 - t;
    return t;
}

// barrier of the time-domain kernels: their row buffers may live in global memory (long rows), so the wave's outstanding
// stores AND loads are drained, not only its LDS operations
__device__ __forceinline__ void td_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// dst[t] = (1/AW) sum_{j=-H..H} src[refl(t+j)],  H = AW / 2  (AverageFilter: reflect pad + AvgPool1d(AW, stride 1))
__device__ __forceinline__ void boxN(const float* src, float* dst, int T, int AW) {
    const int H = AW >> 1;
    const float inv = 1.f / (float)AW;
    for (int t = threadIdx.x; t < T; t += 256) {
        float s = 0.f;
        for (int j = -H; j <= H; ++j) s += src[refl(t + j, T)];
        dst[t] = s * inv;
    }
}

// adjoint of boxN: dsrc[s] = sum over padded taps that read src[s]
__device__ __forceinline__ float boxN_pad_adj(const float* dy, int tau, int T, int AW) {   // d/dxpad[tau], tau in [0,T+AW-1)
    if (tau < 0 || tau >= T + AW - 1) return 0.f;
    float s = 0.f;
    const int lo = max(0, tau - (AW - 1)), hi = min(T - 1, tau);
    for (int t = lo; t <= hi; ++t) s += dy[t];
    return s / (float)AW;
}
__device__ __forceinline__ void boxN_adjoint(const float* dy, float* dsrc, int T, int AW) {
    const int H = AW >> 1;
    for (int s = threadIdx.x; s < T; s += 256) {
        float g = boxN_pad_adj(dy, s + H, T, AW);
        if (s >= 1 && s <= H) g += boxN_pad_adj(dy, H - s, T, AW);
        if (s >= T - 1 - H && s <= T - 2) g += boxN_pad_adj(dy, H + 2 * (T - 1) - s, T, AW);
        dsrc[s] = g;
    }
}

__global__ __launch_bounds__(256) void multi_td_kernel(const float* __restrict__ x_real, const float* __restrict__ x_gen,
                                                       int T, int Cc, TDScale s0, TDScale s1, TDScale s2, float weight,
                                                       float* __restrict__ partial, float* __restrict__ dx_gen,
                                                       int maxF, int AW, float* __restrict__ scratch,
                                                       const float* __restrict__ gscale) {
    if (gscale) weight *= *gscale;       // enters the gradient coefficients only (td_finish_kernel weighs the value)
    extern __shared__ float sm_dyn[];
    // layout: xr, xg, low_r, low_g, rect_r, rect_g, tmp, dlow, drect  (9*T) + frame coefs 4*maxF -- in LDS, or, for rows too
    // long for it (whole-utterance validation), in a global scratch row of the same layout (block-private: the barriers
    // below order its stores and loads within the block)
    float* sm = scratch ? scratch + (size_t)blockIdx.x * ((size_t)9 * T + 4 * maxF) : sm_dyn;
    float* xr = sm;            float* xg = xr + T;
    float* low_r = xg + T;     float* low_g = low_r + T;
    float* rect_r = low_g + T; float* rect_g = rect_r + T;
    float* tmp = rect_g + T;   float* dlow = tmp + T;   float* drect = dlow + T;
    float* cf = drect + T;     // [4][maxF]: g_mlow/W, 2*g_plow, 2*g_prect, g_mrect/W   (already * inv_n * weight)
    __shared__ float red[4];
    const int row = blockIdx.x;            // row = b*C + c
    const int b = row / Cc, c = row % Cc;
    for (int t = threadIdx.x; t < T; t += 256) {
        xr[t] = x_real[((size_t)b * T + t) * Cc + c];
        xg[t] = x_gen[((size_t)b * T + t) * Cc + c];
        dlow[t] = 0.f;
        drect[t] = 0.f;
    }
    td_sync();
    boxN(xr, tmp, T, AW);   td_sync();
    boxN(tmp, low_r, T, AW); td_sync();
    boxN(xg, tmp, T, AW);   td_sync();
    boxN(tmp, low_g, T, AW); td_sync();
    for (int t = threadIdx.x; t < T; t += 256) {
        rect_r[t] = fabsf(xr[t] - low_r[t]);
        rect_g[t] = fabsf(xg[t] - low_g[t]);
    }
    td_sync();

    const TDScale sc[3] = {s0, s1, s2};
    float loss_acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int si = 0; si < 3; ++si) {
        const TDScale S = sc[si];
        float l = 0.f;
        for (int f = threadIdx.x; f < S.F; f += 256) {
            float ml_r = 0.f, pl_r = 0.f, pr_r = 0.f, mr_r = 0.f, ml_g = 0.f, pl_g = 0.f, pr_g = 0.f, mr_g = 0.f;
            for (int j = 0; j < S.W; ++j) {
                const int t = refl(f * S.S + j - S.P, T);
                const float a = low_r[t], r = rect_r[t], ag = low_g[t], rg = rect_g[t];
                ml_r += a; pl_r += a * a; pr_r += r * r; mr_r += r;
                ml_g += ag; pl_g += ag * ag; pr_g += rg * rg; mr_g += rg;
            }
            const float invW = 1.f / (float)S.W;
            const float d0 = (ml_g - ml_r) * invW, d1 = pl_g - pl_r, d2 = pr_g - pr_r, d3 = (mr_g - mr_r) * invW;
            l += fabsf(d0) + fabsf(d1) + fabsf(d2) + fabsf(d3);
            const float k = S.inv_n * weight;
            auto sgn = [](float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); };
            cf[0 * maxF + f] = sgn(d0) * k * invW;
            cf[1 * maxF + f] = sgn(d1) * k * 2.f;
            cf[2 * maxF + f] = sgn(d2) * k * 2.f;
            cf[3 * maxF + f] = sgn(d3) * k * invW;
        }
        loss_acc[si] = l * S.inv_n;
        td_sync();
        if (dx_gen) {
            // gather form of the frame/reflect adjoint: position t receives from padded taus t+P and its mirrors
            for (int t = threadIdx.x; t < T; t += 256) {
                float gl = 0.f, gr = 0.f;
                const float lv = low_g[t], rv = rect_g[t];
                int taus[3] = {t + S.P, (t >= 1 && t <= S.P) ? S.P - t : -1,
                               (t >= T - 1 - S.P && t <= T - 2) ? S.P + 2 * (T - 1) - t : -1};
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int tau = taus[q];
                    if (tau < 0) continue;
                    // frames f with f*S <= tau < f*S + W
                    int f_hi = tau / S.S;
                    if (f_hi > S.F - 1) f_hi = S.F - 1;
                    int f_lo = tau - S.W + 1;
                    f_lo = f_lo <= 0 ? 0 : (f_lo + S.S - 1) / S.S;
                    for (int f = f_lo; f <= f_hi; ++f) {
                        gl += cf[0 * maxF + f] + cf[1 * maxF + f] * lv;
                        gr += cf[3 * maxF + f] + cf[2 * maxF + f] * rv;
                    }
                }
                dlow[t] += gl;
                drect[t] += gr;
            }
        }
        td_sync();
    }
    // loss partials: [row][3]
    for (int si = 0; si < 3; ++si) {
        const float tot = block_sum_256(loss_acc[si], red);
        if (threadIdx.x == 0) partial[(size_t)row * 3 + si] = tot;
    }
    if (!dx_gen) return;
    // rect = |x - low|:  dh = sign(x-low)*drect ; dx = dh + A^T A^T (dlow - dh)
    td_sync();
    for (int t = threadIdx.x; t < T; t += 256) {
        const float h = xg[t] - low_g[t];
        const float dh = (h > 0.f ? 1.f : (h < 0.f ? -1.f : 0.f)) * drect[t];
        rect_g[t] = dh;                  // reuse as dh
        dlow[t] -= dh;
    }
    td_sync();
    boxN_adjoint(dlow, tmp, T, AW);  td_sync();
    boxN_adjoint(tmp, low_g, T, AW); td_sync();
    for (int t = threadIdx.x; t < T; t += 256) dx_gen[((size_t)b * T + t) * Cc + c] = rect_g[t] + low_g[t];
}

// calculate_time_domain_features of ONE signal: one block per (b, c) row; row buffers (x, low / tmp, rect) in global scratch
__global__ __launch_bounds__(256) void td_features_kernel(const float* __restrict__ x, int T, int Cc, int W, int S, int P, int F,
                                                          int AW, float* __restrict__ feats, float* __restrict__ scratch) {
    float* xr = scratch + (size_t)blockIdx.x * 3 * T;
    float* low = xr + T;
    float* tmp = low + T;
    const int row = blockIdx.x, b = row / Cc, c = row % Cc;
    for (int t = threadIdx.x; t < T; t += 256) xr[t] = x[((size_t)b * T + t) * Cc + c];
    td_sync();
    boxN(xr, tmp, T, AW); td_sync();
    boxN(tmp, low, T, AW); td_sync();
    for (int t = threadIdx.x; t < T; t += 256) tmp[t] = fabsf(xr[t] - low[t]);       // rectified high-frequency part
    td_sync();
    for (int f = threadIdx.x; f < F; f += 256) {
        float ml = 0.f, pl = 0.f, pr = 0.f, mr = 0.f;
        for (int j = 0; j < W; ++j) {
            const int t = refl(f * S + j - P, T);
            const float a = low[t], r = tmp[t];
            ml += a; pl += a * a; pr += r * r; mr += r;
        }
        float* o = feats + (((size_t)b * F + f) * Cc + c) * 4;
        o[0] = ml / (float)W; o[1] = pl; o[2] = pr; o[3] = mr / (float)W;
    }
}

// AverageFilter forward (adjoint == 0): y[r][t] = (1/AW) sum_j xpad[r][t + j], xpad = reflect pad by P (or x itself);
// adjoint: dx[r][s] = sum over the padded positions that mirror to s of (1/AW) sum_{t : t <= tau < t + AW} dy[r][t]
__global__ __launch_bounds__(256) void avg_filter_kernel(const float* __restrict__ src, float* __restrict__ dst, long rows, int T,
                                                         int To, int AW, int P, int adjoint) {
    const float inv = 1.f / (float)AW;
    if (!adjoint) {
        const long n = rows * To;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
            const long r = i / To;
            const int t = (int)(i - r * To);
            float s = 0.f;
            for (int j = 0; j < AW; ++j) s += src[r * T + refl(t + j - P, T)];
            dst[i] = s * inv;
        }
        return;
    }
    const long n = rows * T;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / T;
        const int s_ = (int)(i - r * T);
        const float* dy = src + r * To;
        int taus[3] = {s_ + P, (s_ >= 1 && s_ <= P) ? P - s_ : -1, (s_ >= T - 1 - P && s_ <= T - 2) ? P + 2 * (T - 1) - s_ : -1};
        float g = 0.f;
        for (int q = 0; q < 3; ++q) {
            const int tau = taus[q];
            if (tau < 0) continue;
            const int lo = max(0, tau - (AW - 1)), hi = min(To - 1, tau);
            for (int t = lo; t <= hi; ++t) g += dy[t];
        }
        dst[i] = g * inv;
    }
}

// out[slot_base + s] += sum_rows partial[row][s]   (fixed order -> deterministic)
__global__ __launch_bounds__(256) void td_finish_kernel(const float* __restrict__ partial, int rows, float weight,
                                                        float* __restrict__ loss_slot, float* __restrict__ per_scale) {
    __shared__ float red[4];
    float tot = 0.f;
    for (int s = 0; s < 3; ++s) {
        float a = 0.f;
        for (int r = threadIdx.x; r < rows; r += 256) a += partial[(size_t)r * 3 + s];
        a = block_sum_256(a, red);
        if (threadIdx.x == 0 && per_scale) per_scale[s] = a;
        tot += a;
    }
    if (threadIdx.x == 0 && loss_slot) *loss_slot += tot * weight;
}

// speech-unit loss: mean_rows || t - p + eps ||_2 ; one wave per row
__global__ __launch_bounds__(256) void su_loss_kernel(const float* __restrict__ tgt, const float* __restrict__ pred,
                                                      int rows, int D, float eps, float w_over_rows,
                                                      float* __restrict__ partial, float* __restrict__ dpred,
                                                      const float* __restrict__ gscale) {
    if (gscale) w_over_rows *= *gscale;  // (gradient only: sum_partials_kernel weighs the value)
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* t = tgt + (size_t)row * D;
    const float* p = pred + (size_t)row * D;
    float ss = 0.f;
    for (int j = lane; j < D; j += 64) { const float d = t[j] - p[j] + eps; ss += d * d; }
    ss = wave_sum(ss);
    const float nrm = sqrtf(ss);
    if (lane == 0) partial[row] = nrm;
    if (dpred) {
        const float k = nrm > 0.f ? -w_over_rows / nrm : 0.f;
        for (int j = lane; j < D; j += 64) dpred[(size_t)row * D + j] = (t[j] - p[j] + eps) * k;
    }
}

// cross entropy + argmax counters; one wave per row, P <= 64 classes
__global__ __launch_bounds__(256) void phoneme_ce_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                         int rows, int P, int silence, float w_over_rows,
                                                         float* __restrict__ partial, int* __restrict__ counters,
                                                         float* __restrict__ dlogits, long long* __restrict__ argmax_out,
                                                         const float* __restrict__ gscale) {
    if (gscale) w_over_rows *= *gscale;  // (gradient only)
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float x = lane < P ? logits[(size_t)row * P + lane] : -INFINITY;
    // max + first-index argmax (torch.argmax returns the first maximal index)
    float m = x;
    int am = lane < P ? lane : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64);
        const int oa = __shfl_xor(am, o, 64);
        if (om > m || (om == m && oa < am)) { m = om; am = oa; }
    }
    const float e = lane < P ? expf(x - m) : 0.f;
    const float se = wave_sum(e);
    const int tg = (int)target[row];
    const float lse = logf(se) + m;
    const float xt = __shfl(x, tg, 64);
    if (lane == 0) {
        partial[row] = lse - xt;
        if (argmax_out) argmax_out[row] = am;
        if (counters) {
            atomicAdd(&counters[0], 1);
            if (tg == silence) atomicAdd(&counters[1], 1);
            if (am == tg) {
                atomicAdd(&counters[2], 1);
                if (tg != silence) atomicAdd(&counters[3], 1);
            }
        }
    }
    if (dlogits && lane < P) dlogits[(size_t)row * P + lane] = (e / se - (lane == tg ? 1.f : 0.f)) * w_over_rows;
}

// loss_slot += scale * sum(partial[0..n))  in fixed order (single block)
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, int n, float scale,
                                                           float* __restrict__ loss_slot) {
    __shared__ float red[4];
    float a = 0.f;
    for (int r = threadIdx.x; r < n; r += 256) a += partial[r];
    a = block_sum_256(a, red);
    if (threadIdx.x == 0) *loss_slot += a * scale;
}

// Every block ends in ONE atomic on the shared loss word, and same-address atomics serialise in L2 at ~13 ns each: with
// the former cap of 2048 blocks a 6 MB feature map took 23 us, 5x its HBM time (round-2 kernel trace).  512 blocks of
// grid-stride loops keep >= 128 KiB of loads in flight per pass, enough for the HBM-bound body.
inline int nblk(long n) {
    long b = (n + 255) / 256;
    return (int)(b > 512 ? 512 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int stegan_loss_grad_scale(const float* scale_dev) {
    g_gscale = scale_dev;
    return STEGAN_OK;
}

extern "C" int stegan_pair_loss_fwd_bwd(const float* a, const float* b, float target, int64_t n, int32_t mode,
                                        float weight, float* loss_slot, float* da, void* stream) {
    STEGAN_CHECK_ARG(a && n > 0 && (mode == 0 || mode == 1));
    const int vec = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)da) % 16) == 0;
    hipLaunchKernelGGL(pair_loss_kernel, dim3(nblk(vec ? (n + 3) / 4 : n)), dim3(256), 0, (hipStream_t)stream, a, b, target,
                       (long)n, mode, weight / (float)n, loss_slot, da, vec, g_gscale);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

static int pair_loss_multi_impl(int32_t n, const float* const* a, const float* const* b, const int64_t* numel,
                                const float* weight, const float* target, const float* target2, const int64_t* split,
                                int32_t mode, float* loss_slot, float* const* da, float* partials, int64_t partials_floats,
                                void* stream) {
    STEGAN_CHECK_ARG(n >= 1 && a && numel && weight && loss_slot && (mode == 0 || mode == 1));
    int64_t used = 0;
    for (int base = 0; base < n; base += PL_MULTI) {
        PairMulti m;
        m.n = n - base < PL_MULTI ? n - base : PL_MULTI;
        m.mode = mode; m.loss_slot = loss_slot; m.gscale = g_gscale;
        int blocks = 0;
        for (int i = 0; i < m.n; ++i) {
            const int j = base + i;
            STEGAN_CHECK_ARG(a[j] && numel[j] > 0);
            PairMulti::E& e = m.e[i];
            e.a = a[j]; e.b = b ? b[j] : nullptr; e.da = da ? da[j] : nullptr; e.n = numel[j];
            e.w_over_n = weight[j] / (float)numel[j];
            e.target = target ? target[j] : 0.f;
            e.target2 = target2 ? target2[j] : e.target;
            e.split = (split && split[j] > 0) ? split[j] : (1l << 62);
            e.vec = (((uintptr_t)e.a | (uintptr_t)e.b | (uintptr_t)e.da) % 16) == 0;
            m.first[i] = blocks;
            int nb = nblk(e.vec ? (e.n + 3) / 4 : e.n);
            if (nb > 128) nb = 128;                   // the launch as a whole fills the chip
            blocks += nb;
        }
        m.first[m.n] = blocks;
        // the blocks' shares of the value: slots of their own when there is room (then a fixed-order sum), else atomics
        m.part = (partials && used + blocks <= partials_floats) ? partials + used : nullptr;
        hipLaunchKernelGGL(pair_loss_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, m);
        if (m.part) {
            hipLaunchKernelGGL(pair_loss_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)m.part, blocks,
                               loss_slot);
            used += blocks;
        }
    }
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_pair_loss_multi_ex(int32_t n, const float* const* a, const float* const* b, const int64_t* numel,
                                         const float* weight, const float* target, const float* target2,
                                         const int64_t* split, int32_t mode, float* loss_slot, float* const* da,
                                         void* stream) {
    return pair_loss_multi_impl(n, a, b, numel, weight, target, target2, split, mode, loss_slot, da, nullptr, 0, stream);
}

extern "C" int stegan_pair_loss_multi_det(int32_t n, const float* const* a, const float* const* b, const int64_t* numel,
                                          const float* weight, const float* target, const float* target2,
                                          const int64_t* split, int32_t mode, float* loss_slot, float* const* da,
                                          float* partials, int64_t partials_floats, void* stream) {
    STEGAN_CHECK_ARG(partials && partials_floats > 0);
    return pair_loss_multi_impl(n, a, b, numel, weight, target, target2, split, mode, loss_slot, da, partials, partials_floats,
                                stream);
}

extern "C" int stegan_pair_loss_multi(int32_t n, const float* const* a, const float* const* b, const int64_t* numel,
                                      float target, int32_t mode, float weight, float* loss_slot, float* const* da,
                                      void* stream) {
    STEGAN_CHECK_ARG(n >= 1 && n <= 4096);
    std::vector<float> w((size_t)n, weight), t((size_t)n, target);
    return stegan_pair_loss_multi_ex(n, a, b, numel, w.data(), t.data(), nullptr, nullptr, mode, loss_slot, da, stream);
}

extern "C" int stegan_scale_by_scalar(const float* x, const float* scalar_dev, float* out, int64_t n, void* stream) {
    STEGAN_CHECK_ARG(x && scalar_dev && out && n > 0);
    hipLaunchKernelGGL(scale_by_scalar_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, x, scalar_dev, out, (long)n);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

// out = in[0] + in[1] (+ in[2] + in[3]), left to right: the sum of the gradients of a tensor with several consumers (the
// generated EMG feeds the period branch, the scale branch and the D-independent losses; a scale input feeds its
// discriminator and the next pooling; the encoder's last layer feeds both heads) -- what the autograd engine would do with one
// ATen add per extra consumer, as ONE launch in a fixed order (ops.fan_out).
namespace {
struct SumN { const float* in[4]; int n; };
__global__ __launch_bounds__(256) void sum_n_kernel(const SumN a, float* __restrict__ out, long n, int vec) {
    if (vec) {
        const long n4 = n >> 2;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
            float4 s = reinterpret_cast<const float4*>(a.in[0])[i];
            for (int j = 1; j < a.n; ++j) {
                const float4 v = reinterpret_cast<const float4*>(a.in[j])[i];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            reinterpret_cast<float4*>(out)[i] = s;
        }
        for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
            float s = a.in[0][i];
            for (int j = 1; j < a.n; ++j) s += a.in[j][i];
            out[i] = s;
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
            float s = a.in[0][i];
            for (int j = 1; j < a.n; ++j) s += a.in[j][i];
            out[i] = s;
        }
    }
}
}  // namespace

extern "C" int stegan_sum_n(int32_t n_in, const float* const* in, float* out, int64_t numel, void* stream) {
    STEGAN_CHECK_ARG(n_in >= 2 && n_in <= 4 && in && out && numel > 0);
    SumN a;
    a.n = n_in;
    uintptr_t al = (uintptr_t)out;
    for (int j = 0; j < 4; ++j) {
        a.in[j] = j < n_in ? in[j] : nullptr;
        if (j < n_in) { STEGAN_CHECK_ARG(in[j]); al |= (uintptr_t)in[j]; }
    }
    const int vec = (al % 16) == 0;
    hipLaunchKernelGGL(sum_n_kernel, dim3(nblk(vec ? (numel + 3) / 4 : numel)), dim3(256), 0, (hipStream_t)stream, a, out,
                       (long)numel, vec);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

// out[i] = sum_j coeff[i * n_in + j] * *in[j]: the loss terms of a step combined into its reported scalars (the weighted
// total, the unweighted terms) in ONE one-thread launch instead of a chain of 0-dim torch mul / add / div kernels
// (train.py:213-264: loss_G += w * term).  n_in, n_out <= 8.
namespace {
struct Combine { const float* in[8]; float* out[8]; float coeff[64]; int n_in, n_out; };
__global__ void scalar_combine_kernel(const Combine c) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float v[8];
    for (int j = 0; j < c.n_in; ++j) v[j] = *c.in[j];
    for (int i = 0; i < c.n_out; ++i) {
        float a = 0.f;
        for (int j = 0; j < c.n_in; ++j) a += c.coeff[i * c.n_in + j] * v[j];
        *c.out[i] = a;
    }
}
}  // namespace

extern "C" int stegan_scalar_combine(int32_t n_in, const float* const* in, int32_t n_out, float* const* out, const float* coeff,
                                     void* stream) {
    STEGAN_CHECK_ARG(n_in >= 1 && n_in <= 8 && n_out >= 1 && n_out <= 8 && in && out && coeff);
    Combine c;
    c.n_in = n_in; c.n_out = n_out;
    for (int j = 0; j < n_in; ++j) { STEGAN_CHECK_ARG(in[j]); c.in[j] = in[j]; }
    for (int i = 0; i < n_out; ++i) { STEGAN_CHECK_ARG(out[i]); c.out[i] = out[i]; }
    for (int k = 0; k < n_in * n_out; ++k) c.coeff[k] = coeff[k];
    hipLaunchKernelGGL(scalar_combine_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, c);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int64_t stegan_multi_td_workspace_floats(int32_t B, int32_t C) { return (int64_t)B * C * 3; }

namespace {
constexpr size_t TD_LDS_MAX = 150 * 1024;
inline int td_frames(int T, int W, int S, int pad) { return (T + 2 * (pad ? W / 2 : 0) - W) / S + 1; }
}

// row buffer of the time-domain kernels in floats (9 T + 4 maxF); rows that exceed the LDS budget need a global scratch
// of B * C such rows (stegan_multi_td_fwd_bwd_ex)
extern "C" int64_t stegan_multi_td_scratch_floats(int32_t B, int32_t T, int32_t C, int32_t n_scales, const int32_t* win_host,
                                                  const int32_t* shift_host, int32_t pad_windowing) {
    if (B <= 0 || T <= 0 || C <= 0 || n_scales < 1 || n_scales > 3 || !win_host || !shift_host) return -1;
    int maxF = 1;
    for (int i = 0; i < n_scales; ++i) {
        if (win_host[i] < 1 || shift_host[i] < 1) return -1;
        const int F = td_frames(T, win_host[i], shift_host[i], pad_windowing);
        if (F > maxF) maxF = F;
    }
    const size_t row = (size_t)9 * T + 4 * maxF;
    return row * sizeof(float) <= TD_LDS_MAX ? 0 : (int64_t)row * B * C;
}

// ste_gan/losses/time_domain_loss.py:20-41 in full: any odd average-filter width, with or without the reflect-padded
// windowing.  scratch (stegan_multi_td_scratch_floats; may be null when that returns 0) holds the row buffers of rows
// too long for LDS (whole-utterance validation).
extern "C" int stegan_multi_td_fwd_bwd_ex(const float* x_real, const float* x_gen, int32_t B, int32_t T, int32_t C,
                                          int32_t n_scales, const int32_t* win_host, const int32_t* shift_host,
                                          int32_t avg_window, int32_t pad_windowing, float weight, float* loss_slot,
                                          float* per_scale, float* dx_gen, float* workspace, float* scratch,
                                          int64_t scratch_floats, void* stream) {
    STEGAN_CHECK_ARG(x_real && x_gen && workspace && B > 0 && T > 0 && C > 0);
    STEGAN_CHECK_ARG(n_scales >= 1 && n_scales <= 3 && win_host && shift_host);
    STEGAN_CHECK_ARG(avg_window >= 1 && (avg_window & 1) && avg_window / 2 < T);      // reflect padding needs pad < length
    TDScale sc[3] = {TDScale{1, 1, 0, 0, 0.f}, TDScale{1, 1, 0, 0, 0.f}, TDScale{1, 1, 0, 0, 0.f}};
    int maxF = 1;
    for (int i = 0; i < n_scales; ++i) {
        const int W = win_host[i], S = shift_host[i], P = pad_windowing ? W / 2 : 0;
        STEGAN_CHECK_ARG(W >= 1 && S >= 1);
        STEGAN_CHECK_ARG(P < T && W <= T + 2 * P);          // (ATen raises likewise: pad >= length, window > signal)
        const int F = td_frames(T, W, S, pad_windowing);
        STEGAN_CHECK_ARG(F >= 1);
        sc[i] = TDScale{W, S, P, F, 1.f / ((float)B * F * C * 4.f)};
        if (F > maxF) maxF = F;
    }
    const size_t row = (size_t)9 * T + 4 * maxF;
    size_t lds = row * sizeof(float);
    float* scr = nullptr;
    if (lds > TD_LDS_MAX) {
        STEGAN_CHECK_ARG(scratch && scratch_floats >= (int64_t)row * B * C);
        scr = scratch; lds = 0;
    }
    hipStream_t s = (hipStream_t)stream;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)multi_td_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TD_LDS_MAX);
        attr_set = true;
    }
    hipLaunchKernelGGL(multi_td_kernel, dim3(B * C), dim3(256), lds, s, x_real, x_gen, T, C, sc[0], sc[1], sc[2], weight,
                       workspace, dx_gen, maxF, avg_window, scr, g_gscale);
    hipLaunchKernelGGL(td_finish_kernel, dim3(1), dim3(256), 0, s, workspace, B * C, weight, loss_slot, per_scale);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_multi_td_fwd_bwd(const float* x_real, const float* x_gen, int32_t B, int32_t T, int32_t C,
                                       int32_t n_scales, const int32_t* win_host, const int32_t* shift_host,
                                       float weight, float* loss_slot, float* per_scale, float* dx_gen,
                                       float* workspace, void* stream) {
    return stegan_multi_td_fwd_bwd_ex(x_real, x_gen, B, T, C, n_scales, win_host, shift_host, 9, 1, weight, loss_slot,
                                      per_scale, dx_gen, workspace, nullptr, 0, stream);
}

// TimeDomainFeatureLoss.calculate_time_domain_features (time_domain_loss.py:58-69): x (B, T, C) -> (B, F, C, 4) =
// [mean(low), sum(low^2), sum(rect^2), mean(rect)] per frame; scratch: B * C rows of 3 T floats.
extern "C" int stegan_td_features(const float* x, int32_t B, int32_t T, int32_t C, int32_t win, int32_t shift,
                                  int32_t avg_window, int32_t pad_windowing, float* feats, float* scratch, void* stream) {
    STEGAN_CHECK_ARG(x && feats && scratch && B > 0 && T > 0 && C > 0 && win >= 1 && shift >= 1);
    STEGAN_CHECK_ARG(avg_window >= 1 && (avg_window & 1) && avg_window / 2 < T);
    const int P = pad_windowing ? win / 2 : 0;
    STEGAN_CHECK_ARG(P < T && win <= T + 2 * P);
    const int F = td_frames(T, win, shift, pad_windowing);
    hipLaunchKernelGGL(td_features_kernel, dim3(B * C), dim3(256), 0, (hipStream_t)stream, x, T, C, win, shift, P, F,
                       avg_window, feats, scratch);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

// AverageFilter (layers/average_filter.py:10-28) on rows of length T: reflect pad by window / 2 (pad_signal) + AvgPool1d
// (window, stride 1).  Output length T (padded) or T - window + 1.  The backward is the adjoint (dx from dy).
extern "C" int stegan_average_filter_fwd(const float* x, float* y, int64_t rows, int32_t T, int32_t window, int32_t pad_signal,
                                         void* stream) {
    STEGAN_CHECK_ARG(x && y && rows > 0 && T > 0 && window >= 1);
    STEGAN_CHECK_ARG(pad_signal ? (window / 2 < T) : (window <= T));
    const int To = pad_signal ? T + 2 * (window / 2) - window + 1 : T - window + 1;
    const long n = rows * To;
    hipLaunchKernelGGL(avg_filter_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, x, y, rows, T, To, window,
                       pad_signal ? window / 2 : 0, 0);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_average_filter_bwd(const float* dy, float* dx, int64_t rows, int32_t T, int32_t window, int32_t pad_signal,
                                         void* stream) {
    STEGAN_CHECK_ARG(dy && dx && rows > 0 && T > 0 && window >= 1);
    STEGAN_CHECK_ARG(pad_signal ? (window / 2 < T) : (window <= T));
    const int To = pad_signal ? T + 2 * (window / 2) - window + 1 : T - window + 1;
    const long n = rows * T;
    hipLaunchKernelGGL(avg_filter_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, dy, dx, rows, T, To, window,
                       pad_signal ? window / 2 : 0, 1);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_su_loss_fwd_bwd(const float* target, const float* pred, int32_t rows, int32_t D, float weight,
                                      float* loss_slot, float* dpred, float* workspace, void* stream) {
    STEGAN_CHECK_ARG(target && pred && loss_slot && workspace && rows > 0 && D > 0);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(su_loss_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, s, target, pred, rows, D, 1e-6f,
                       weight / (float)rows, workspace, dpred, g_gscale);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, workspace, rows, weight / (float)rows, loss_slot);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_phoneme_ce_fwd_bwd(const float* logits, const int64_t* targets, int32_t rows, int32_t P,
                                         int32_t silence_index, float weight, float* loss_slot, int32_t* counters,
                                         float* dlogits, int64_t* argmax_out, float* workspace, void* stream) {
    STEGAN_CHECK_ARG(logits && targets && loss_slot && workspace && rows > 0 && P > 0 && P <= 64);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(phoneme_ce_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, s, logits, (const long long*)targets,
                       rows, P, silence_index, weight / (float)rows, workspace, counters, dlogits, (long long*)argmax_out,
                       g_gscale);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, workspace, rows, weight / (float)rows, loss_slot);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}
