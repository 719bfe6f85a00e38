// Discriminator-side data movement kernels: MPD reflect-pad + period fold, (B,T,C)<->(B,C,T),
// AvgPool1d(4,2,1) and the spectral-norm power iteration.  All HBM-bound, tiny next to the convs.
#include "common.h"

namespace {

// out[(b*p + w), c, h] = xpad[b, h*p + w, c],  xpad = right reflect pad of x (B,T,C) to H*p samples
__global__ __launch_bounds__(256) void mpd_fold_fwd_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                           int B, int T, int Cc, int p, int H) {
    const long total = (long)B * p * Cc * H;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int h = (int)(i % H);
        long r = i / H;
        const int c = (int)(r % Cc);
        r /= Cc;
        const int w = (int)(r % p), b = (int)(r / p);
        int t = h * p + w;
        if (t >= T) t = 2 * T - 2 - t;          // reflect (no edge repeat): xpad[T+i] = x[T-2-i]
        out[i] = x[((long)b * T + t) * Cc + c];
    }
}

// dx[b,t,c] (+)= sum over the folded positions that read x[b,t,c]
__global__ __launch_bounds__(256) void mpd_fold_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx,
                                                           int B, int T, int Cc, int p, int H, int accumulate) {
    const long total = (long)B * T * Cc;
    const int Tp = H * p;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % Cc);
        long r = i / Cc;
        const int t = (int)(r % T), b = (int)(r / T);
        float g = dout[(((long)b * p + t % p) * Cc + c) * H + t / p];
        const int tr = 2 * T - 2 - t;            // padded position that mirrors onto t
        if (tr >= T && tr < Tp) g += dout[(((long)b * p + tr % p) * Cc + c) * H + tr / p];
        dx[i] = accumulate ? dx[i] + g : g;
    }
}

// The folds of ALL period discriminators in one launch, and their backward in one launch that also forms the sum of the
// discriminators' gradients w.r.t. the signal (autograd added the five dx tensors with four launches of its own).
constexpr int MPD_MAX = 8;
struct FoldMulti {
    int n, B, T, Cc;
    int p[MPD_MAX], H[MPD_MAX];
    long first[MPD_MAX + 1];            // first linear output element of each period
    float* out[MPD_MAX];
};

__global__ __launch_bounds__(256) void mpd_fold_fwd_multi_kernel(const float* __restrict__ x, const FoldMulti a) {
    const long total = a.first[a.n];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int k = 0;
#pragma unroll
        for (int j = 1; j < MPD_MAX; ++j)
            if (j < a.n && i >= a.first[j]) k = j;
        const int p = a.p[k], H = a.H[k];
        const long li = i - a.first[k];
        const int h = (int)(li % H);
        long r = li / H;
        const int c = (int)(r % a.Cc);
        r /= a.Cc;
        const int w = (int)(r % p), b = (int)(r / p);
        int t = h * p + w;
        if (t >= a.T) t = 2 * a.T - 2 - t;
        a.out[k][li] = x[((long)b * a.T + t) * a.Cc + c];
    }
}

struct FoldMultiB {
    int n, B, T, Cc;
    int p[MPD_MAX], H[MPD_MAX];
    const float* dout[MPD_MAX];
};

__global__ __launch_bounds__(256) void mpd_fold_bwd_multi_kernel(const FoldMultiB a, float* __restrict__ dx, int accumulate) {
    const long total = (long)a.B * a.T * a.Cc;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % a.Cc);
        long r = i / a.Cc;
        const int t = (int)(r % a.T), b = (int)(r / a.T);
        const int tr = 2 * a.T - 2 - t;          // padded position that mirrors onto t
        float g = 0.f;
#pragma unroll
        for (int k = 0; k < MPD_MAX; ++k) {
            if (k >= a.n) break;
            const int p = a.p[k], H = a.H[k];
            const float* d = a.dout[k];
            float gk = d[(((long)b * p + t % p) * a.Cc + c) * H + t / p];          // (per period in the order of the single
            if (tr >= a.T && tr < H * p) gk += d[(((long)b * p + tr % p) * a.Cc + c) * H + tr / p];   //  kernel, then summed)
            g += gk;
        }
        dx[i] = accumulate ? dx[i] + g : g;
    }
}

// (B,T,C) -> (B,C,T)   (C is tiny: 8)
__global__ __launch_bounds__(256) void btc_to_bct_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         int B, int T, int Cc) {
    const long total = (long)B * T * Cc;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int t = (int)(i % T);
        long r = i / T;
        const int c = (int)(r % Cc), b = (int)(r / Cc);
        y[i] = x[((long)b * T + t) * Cc + c];
    }
}

__global__ __launch_bounds__(256) void bct_to_btc_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         int B, int T, int Cc, int accumulate) {
    const long total = (long)B * T * Cc;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % Cc);
        long r = i / Cc;
        const int t = (int)(r % T), b = (int)(r / T);
        const float v = x[((long)b * Cc + c) * T + t];
        y[i] = accumulate ? y[i] + v : v;
    }
}

// AvgPool1d(kernel 4, stride 2, pad 1, count_include_pad): y[t] = (x[2t-1]+x[2t]+x[2t+1]+x[2t+2])/4
__global__ __launch_bounds__(256) void avgpool421_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             long rows, int T, int To) {
    const long total = rows * To;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int t = (int)(i % To);
        const float* xr = x + (i / To) * T;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ti = 2 * t - 1 + k;
            if (ti >= 0 && ti < T) s += xr[ti];
        }
        y[i] = 0.25f * s;
    }
}

// dx[ti] = 0.25 * sum_{t : 2t-1 <= ti <= 2t+2} dy[t]  (+ add[ti])
__global__ __launch_bounds__(256) void avgpool421_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ add,
                                                             float* __restrict__ dx, long rows, int T, int To) {
    const long total = rows * T;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ti = (int)(i % T);
        const float* dr = dy + (i / T) * To;
        // t ranges over ceil((ti-2)/2) .. floor((ti+1)/2)
        int lo = (ti - 2 + 1) >> 1, hi = (ti + 1) >> 1;
        if (ti - 2 < 0) lo = 0;
        float s = 0.f;
        for (int t = lo; t <= hi; ++t)
            if (t >= 0 && t < To) s += dr[t];
        dx[i] = 0.25f * s + (add ? add[i] : 0.f);
    }
}

// ---------------- spectral norm: one power iteration + sigma --------------------------------
// stage 1: partial[rc][j] = sum_{i in row chunk rc} W[i][j] * u[i]       (deterministic two-stage column sums)
constexpr int SN_ROWS = 32;
__device__ __forceinline__ void sn_wtu_body(const float* __restrict__ W, const float* __restrict__ u,
                                            float* __restrict__ partial, int R, int Cn, int bx, int by) {
    const int j = bx * 256 + threadIdx.x;
    const int r0 = by * SN_ROWS;
    if (j >= Cn || r0 >= R) return;
    // four independent partial sums: the loop below otherwise waits for every load before it issues the next one (a
    // column of 32 rows took 32 memory round trips: 18 us for a 10 MB weight)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const int r1 = min(R, r0 + SN_ROWS);
    int i = r0;
#pragma unroll 2
    for (; i + 4 <= r1; i += 4) {
        s0 += W[(size_t)i * Cn + j] * u[i];
        s1 += W[(size_t)(i + 1) * Cn + j] * u[i + 1];
        s2 += W[(size_t)(i + 2) * Cn + j] * u[i + 2];
        s3 += W[(size_t)(i + 3) * Cn + j] * u[i + 3];
    }
    for (; i < r1; ++i) s0 += W[(size_t)i * Cn + j] * u[i];
    partial[(size_t)by * Cn + j] = (s0 + s1) + (s2 + s3);
}

__global__ __launch_bounds__(256) void sn_wtu_kernel(const float* __restrict__ W, const float* __restrict__ u,
                                                     float* __restrict__ partial, int R, int Cn) {
    sn_wtu_body(W, u, partial, R, Cn, blockIdx.x, blockIdx.y);
}

// stage 2 (single block): t = sum partial; v = t / max(||t||, eps)
__device__ __forceinline__ void sn_vnorm_body(const float* __restrict__ partial, float* __restrict__ v,
                                              float* __restrict__ v_save, int nchunks, int Cn, float eps) {
    __shared__ float red[16];
    __shared__ float inv;
    float ss = 0.f;
    for (int j = threadIdx.x; j < Cn; j += 1024) {
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;                 // (independent loads in flight, as in stage 1)
        int c = 0;
#pragma unroll 2
        for (; c + 4 <= nchunks; c += 4) {
            t0 += partial[(size_t)c * Cn + j];
            t1 += partial[(size_t)(c + 1) * Cn + j];
            t2 += partial[(size_t)(c + 2) * Cn + j];
            t3 += partial[(size_t)(c + 3) * Cn + j];
        }
        for (; c < nchunks; ++c) t0 += partial[(size_t)c * Cn + j];
        const float t = (t0 + t1) + (t2 + t3);
        v[j] = t;
        ss += t * t;
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    block_sync();
    if (threadIdx.x == 0) {
        float a = 0.f;
        for (int k = 0; k < 16; ++k) a += red[k];
        inv = 1.f / fmaxf(sqrtf(a), eps);
    }
    block_sync();
    for (int j = threadIdx.x; j < Cn; j += 1024) {
        const float t = v[j] * inv;
        v[j] = t;
        if (v_save) v_save[j] = t;
    }
}

__global__ __launch_bounds__(1024) void sn_vnorm_kernel(const float* __restrict__ partial, float* __restrict__ v,
                                                        int nchunks, int Cn, float eps) {
    sn_vnorm_body(partial, v, nullptr, nchunks, Cn, eps);
}

// stage 3: s[i] = <W[i,:], v>   (one wave per row)
__device__ __forceinline__ void sn_wv_body(const float* __restrict__ W, const float* __restrict__ v,
                                           float* __restrict__ s, int R, int Cn, int bx) {
    const int row = bx * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const int lane = threadIdx.x & 63;
    const float* wr = W + (size_t)row * Cn;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int j = lane;
#pragma unroll 2
    for (; j + 192 < Cn; j += 256) {
        a0 += wr[j] * v[j];
        a1 += wr[j + 64] * v[j + 64];
        a2 += wr[j + 128] * v[j + 128];
        a3 += wr[j + 192] * v[j + 192];
    }
    for (; j < Cn; j += 64) a0 += wr[j] * v[j];
    const float a = wave_sum((a0 + a1) + (a2 + a3));
    if (lane == 0) s[row] = a;
}

__global__ __launch_bounds__(256) void sn_wv_kernel(const float* __restrict__ W, const float* __restrict__ v,
                                                    float* __restrict__ s, int R, int Cn) {
    sn_wv_body(W, v, s, R, Cn, blockIdx.x);
}

// stage 4 (single block): u = s / max(||s||,eps) (if update_u); sigma = <u, s>; scale[:] = 1/sigma; sigma_out
__device__ __forceinline__ void sn_finish_body(const float* __restrict__ s, float* __restrict__ u,
                                               float* __restrict__ u_save, float* __restrict__ scale,
                                               float* __restrict__ sigma_out, int R, float eps, int update_u) {
    __shared__ float red[16];
    __shared__ float bc;
    float ss = 0.f;
    for (int i = threadIdx.x; i < R; i += 1024) ss += s[i] * s[i];
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    block_sync();
    if (threadIdx.x == 0) {
        float a = 0.f;
        for (int k = 0; k < 16; ++k) a += red[k];
        bc = 1.f / fmaxf(sqrtf(a), eps);
    }
    block_sync();
    float dot = 0.f;
    for (int i = threadIdx.x; i < R; i += 1024) {
        float ui = update_u ? s[i] * bc : u[i];
        if (update_u) u[i] = ui;
        if (u_save) u_save[i] = ui;
        dot += ui * s[i];
    }
    dot = wave_sum(dot);
    block_sync();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
    block_sync();
    if (threadIdx.x == 0) {
        float a = 0.f;
        for (int k = 0; k < 16; ++k) a += red[k];
        bc = a;
        *sigma_out = a;
    }
    block_sync();
    const float inv_sigma = 1.f / bc;
    for (int i = threadIdx.x; i < R; i += 1024) scale[i] = inv_sigma;
}

__global__ __launch_bounds__(1024) void sn_finish_kernel(const float* __restrict__ s, float* __restrict__ u,
                                                         float* __restrict__ scale, float* __restrict__ sigma_out,
                                                         int R, float eps, int update_u) {
    sn_finish_body(s, u, nullptr, scale, sigma_out, R, eps, update_u);
}

// All spectral-norm layers of one discriminator in ONE launch per stage (the layers are independent: 4 launches per
// forward instead of 4 per layer); same per-layer arithmetic as the single-layer kernels above.
constexpr int SN_MULTI = 8;
struct SNMulti {
    int n, update_u;
    float eps;
    struct E {
        const float* W;
        float *u, *v, *scale, *sigma, *partial, *svec, *u_save, *v_save;
        int R, Cn, nch;
    } e[SN_MULTI];
};

__global__ __launch_bounds__(256) void sn_wtu_multi_kernel(const SNMulti a) {
    const SNMulti::E& e = a.e[blockIdx.z];
    sn_wtu_body(e.W, e.u, e.partial, e.R, e.Cn, blockIdx.x, blockIdx.y);
}
__global__ __launch_bounds__(1024) void sn_vnorm_multi_kernel(const SNMulti a) {
    const SNMulti::E& e = a.e[blockIdx.x];
    sn_vnorm_body(e.partial, e.v, e.v_save, e.nch, e.Cn, a.eps);
}
__global__ __launch_bounds__(256) void sn_wv_multi_kernel(const SNMulti a) {
    const SNMulti::E& e = a.e[blockIdx.y];
    sn_wv_body(e.W, e.v, e.svec, e.R, e.Cn, blockIdx.x);
}
__global__ __launch_bounds__(1024) void sn_finish_multi_kernel(const SNMulti a) {
    const SNMulti::E& e = a.e[blockIdx.x];
    sn_finish_body(e.svec, e.u, e.u_save, e.scale, e.sigma, e.R, a.eps, a.update_u);
}

// dW_orig = dw/sigma - (<dw, W>/sigma^2) u v^T      two kernels: row dots -> total, then elementwise
__device__ __forceinline__ void sn_bwd_dot_body(const float* __restrict__ W, const float* __restrict__ dw,
                                                float* __restrict__ rowdot, int R, int Cn, int bx) {
    const int row = bx * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const int lane = threadIdx.x & 63;
    const float* wr = W + (size_t)row * Cn;
    const float* dr = dw + (size_t)row * Cn;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int j = lane;
#pragma unroll 2
    for (; j + 192 < Cn; j += 256) {
        a0 += wr[j] * dr[j];
        a1 += wr[j + 64] * dr[j + 64];
        a2 += wr[j + 128] * dr[j + 128];
        a3 += wr[j + 192] * dr[j + 192];
    }
    for (; j < Cn; j += 64) a0 += wr[j] * dr[j];
    const float a = wave_sum((a0 + a1) + (a2 + a3));
    if (lane == 0) rowdot[row] = a;
}

__global__ __launch_bounds__(256) void sn_bwd_dot_kernel(const float* __restrict__ W, const float* __restrict__ dw,
                                                         float* __restrict__ rowdot, int R, int Cn) {
    sn_bwd_dot_body(W, dw, rowdot, R, Cn, blockIdx.x);
}

__device__ __forceinline__ void sn_bwd_apply_body(const float* __restrict__ dw, const float* __restrict__ rowdot,
                                                  const float* __restrict__ u, const float* __restrict__ v,
                                                  const float* __restrict__ sigma, float* __restrict__ dW,
                                                  int R, int Cn, int accumulate, int bx, int nbx) {
    __shared__ float tot;
    if (threadIdx.x < 64) {            // every block re-reduces the R row dots (R <= 1024): deterministic, no atomics
        float a = 0.f;
        for (int i = threadIdx.x; i < R; i += 64) a += rowdot[i];
        a = wave_sum(a);
        if (threadIdx.x == 0) tot = a;
    }
    block_sync();
    const float sg = *sigma;
    const float k = tot / (sg * sg), inv = 1.f / sg;
    const long total = (long)R * Cn;
    for (long i = (long)bx * 256 + threadIdx.x; i < total; i += (long)nbx * 256) {
        const int j = (int)(i % Cn), r = (int)(i / Cn);
        const float gv = dw[i] * inv - k * u[r] * v[j];
        dW[i] = accumulate ? dW[i] + gv : gv;               // accumulate: straight into the parameter's .grad arena view
    }
}

__global__ __launch_bounds__(256) void sn_bwd_apply_kernel(const float* __restrict__ dw, const float* __restrict__ rowdot,
                                                           const float* __restrict__ u, const float* __restrict__ v,
                                                           const float* __restrict__ sigma, float* __restrict__ dW,
                                                           int R, int Cn, int accumulate) {
    sn_bwd_apply_body(dw, rowdot, u, v, sigma, dW, R, Cn, accumulate, blockIdx.x, gridDim.x);
}

// the same for up to SNB_MAX layers in one launch per stage (round 5: the scale discriminator's eight spectral-norm backward
// transforms of a step -- four layers, fake and real pass -- used to be 16 launches inside its backward pass; the weight
// gradients they produce are read by the optimiser only, so they run as two launches when the gradients are finalized)
constexpr int SNB_MAX = 8;
struct SNBwdMulti {
    int n;
    int fdot[SNB_MAX + 1], fapp[SNB_MAX + 1];
    struct E { const float* W; const float* dw; const float* u; const float* v; const float* sigma; float* rowdot; float* dW;
               int R, Cn, accumulate; } e[SNB_MAX];
};
__global__ __launch_bounds__(256) void sn_bwd_dot_multi_kernel(const SNBwdMulti a) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < SNB_MAX; ++i)
        if (i < a.n && (int)blockIdx.x >= a.fdot[i]) pi = i;
    const SNBwdMulti::E& e = a.e[pi];
    sn_bwd_dot_body(e.W, e.dw, e.rowdot, e.R, e.Cn, (int)blockIdx.x - a.fdot[pi]);
}
__global__ __launch_bounds__(256) void sn_bwd_apply_multi_kernel(const SNBwdMulti a) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < SNB_MAX; ++i)
        if (i < a.n && (int)blockIdx.x >= a.fapp[i]) pi = i;
    const SNBwdMulti::E& e = a.e[pi];
    sn_bwd_apply_body(e.dw, e.rowdot, e.u, e.v, e.sigma, e.dW, e.R, e.Cn, e.accumulate, (int)blockIdx.x - a.fapp[pi],
                      a.fapp[pi + 1] - a.fapp[pi]);
}

inline int nblocks(long total) {
    long b = (total + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int stegan_mpd_fold_fwd(const float* x_btc, float* out, int32_t B, int32_t T, int32_t Cc, int32_t period,
                                   void* stream) {
    STEGAN_CHECK_ARG(x_btc && out && B > 0 && T > 0 && Cc > 0 && period > 0);
    const int n_pad = period - T % period;
    STEGAN_CHECK_ARG(n_pad < T);                     // reflect pad needs pad < length (ATen raises likewise)
    const int H = (T + n_pad) / period;
    hipLaunchKernelGGL(mpd_fold_fwd_kernel, dim3(nblocks((long)B * period * Cc * H)), dim3(256), 0, (hipStream_t)stream,
                       x_btc, out, B, T, Cc, period, H);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_mpd_fold_bwd(const float* dout, float* dx_btc, int32_t B, int32_t T, int32_t Cc, int32_t period,
                                   int32_t accumulate, void* stream) {
    STEGAN_CHECK_ARG(dout && dx_btc && B > 0 && T > 0 && Cc > 0 && period > 0);
    const int n_pad = period - T % period;
    STEGAN_CHECK_ARG(n_pad < T);
    const int H = (T + n_pad) / period;
    hipLaunchKernelGGL(mpd_fold_bwd_kernel, dim3(nblocks((long)B * T * Cc)), dim3(256), 0, (hipStream_t)stream, dout,
                       dx_btc, B, T, Cc, period, H, accumulate);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_mpd_fold_fwd_multi(const float* x_btc, float* const* outs, int32_t B, int32_t T, int32_t Cc,
                                         const int32_t* periods, int32_t n, void* stream) {
    STEGAN_CHECK_ARG(x_btc && outs && periods && B > 0 && T > 0 && Cc > 0 && n >= 1 && n <= MPD_MAX);
    FoldMulti a;
    a.n = n; a.B = B; a.T = T; a.Cc = Cc;
    a.first[0] = 0;
    for (int k = 0; k < n; ++k) {
        STEGAN_CHECK_ARG(outs[k] && periods[k] > 0);
        const int n_pad = periods[k] - T % periods[k];
        STEGAN_CHECK_ARG(n_pad < T);
        a.p[k] = periods[k]; a.H[k] = (T + n_pad) / periods[k]; a.out[k] = outs[k];
        a.first[k + 1] = a.first[k] + (long)B * a.p[k] * Cc * a.H[k];
    }
    hipLaunchKernelGGL(mpd_fold_fwd_multi_kernel, dim3(nblocks(a.first[n])), dim3(256), 0, (hipStream_t)stream, x_btc, a);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_mpd_fold_bwd_multi(const float* const* douts, float* dx_btc, int32_t B, int32_t T, int32_t Cc,
                                         const int32_t* periods, int32_t n, int32_t accumulate, void* stream) {
    STEGAN_CHECK_ARG(douts && dx_btc && periods && B > 0 && T > 0 && Cc > 0 && n >= 1 && n <= MPD_MAX);
    FoldMultiB a;
    a.n = n; a.B = B; a.T = T; a.Cc = Cc;
    for (int k = 0; k < n; ++k) {
        STEGAN_CHECK_ARG(douts[k] && periods[k] > 0);
        const int n_pad = periods[k] - T % periods[k];
        STEGAN_CHECK_ARG(n_pad < T);
        a.p[k] = periods[k]; a.H[k] = (T + n_pad) / periods[k]; a.dout[k] = douts[k];
    }
    hipLaunchKernelGGL(mpd_fold_bwd_multi_kernel, dim3(nblocks((long)B * T * Cc)), dim3(256), 0, (hipStream_t)stream, a,
                       dx_btc, accumulate);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_btc_to_bct(const float* x, float* y, int32_t B, int32_t T, int32_t Cc, void* stream) {
    STEGAN_CHECK_ARG(x && y && B > 0 && T > 0 && Cc > 0);
    hipLaunchKernelGGL(btc_to_bct_kernel, dim3(nblocks((long)B * T * Cc)), dim3(256), 0, (hipStream_t)stream, x, y, B, T, Cc);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_bct_to_btc(const float* x, float* y, int32_t B, int32_t T, int32_t Cc, int32_t accumulate,
                                 void* stream) {
    STEGAN_CHECK_ARG(x && y && B > 0 && T > 0 && Cc > 0);
    hipLaunchKernelGGL(bct_to_btc_kernel, dim3(nblocks((long)B * T * Cc)), dim3(256), 0, (hipStream_t)stream, x, y, B, T,
                       Cc, accumulate);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_avgpool421_fwd(const float* x, float* y, int64_t rows, int32_t T, void* stream) {
    STEGAN_CHECK_ARG(x && y && rows > 0 && T >= 2);
    const int To = (T + 2 - 4) / 2 + 1;
    hipLaunchKernelGGL(avgpool421_fwd_kernel, dim3(nblocks(rows * To)), dim3(256), 0, (hipStream_t)stream, x, y, (long)rows, T, To);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_avgpool421_bwd(const float* dy, const float* add, float* dx, int64_t rows, int32_t T, void* stream) {
    STEGAN_CHECK_ARG(dy && dx && rows > 0 && T >= 2);
    const int To = (T + 2 - 4) / 2 + 1;
    hipLaunchKernelGGL(avgpool421_bwd_kernel, dim3(nblocks(rows * T)), dim3(256), 0, (hipStream_t)stream, dy, add, dx, (long)rows, T, To);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int64_t stegan_spectral_norm_workspace_floats(int32_t rows, int32_t cols) {
    return (int64_t)ceil_div(rows, SN_ROWS) * cols + rows;
}

extern "C" int stegan_spectral_norm_fwd(const float* w_orig, float* u, float* v, float* scale, float* sigma,
                                        float* workspace, int32_t rows, int32_t cols, int32_t power_iterate,
                                        void* stream) {
    STEGAN_CHECK_ARG(w_orig && u && v && scale && sigma && workspace && rows > 0 && cols > 0);
    hipStream_t s = (hipStream_t)stream;
    const int nch = ceil_div(rows, SN_ROWS);
    float* partial = workspace;
    float* svec = workspace + (size_t)nch * cols;
    const float eps = 1e-12f;
    if (power_iterate) {
        hipLaunchKernelGGL(sn_wtu_kernel, dim3(ceil_div(cols, 256), nch), dim3(256), 0, s, w_orig, u, partial, rows, cols);
        hipLaunchKernelGGL(sn_vnorm_kernel, dim3(1), dim3(1024), 0, s, partial, v, nch, cols, eps);
    }
    hipLaunchKernelGGL(sn_wv_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, s, w_orig, v, svec, rows, cols);
    hipLaunchKernelGGL(sn_finish_kernel, dim3(1), dim3(1024), 0, s, svec, u, scale, sigma, rows, eps, power_iterate);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_spectral_norm_fwd_multi(int32_t n, const float* const* w_orig, float* const* u, float* const* v,
                                              float* const* scale, float* const* sigma, float* const* u_save,
                                              float* const* v_save, float* const* workspace, const int32_t* rows,
                                              const int32_t* cols, int32_t power_iterate, void* stream) {
    STEGAN_CHECK_ARG(n >= 1 && n <= SN_MULTI && w_orig && u && v && scale && sigma && workspace && rows && cols);
    SNMulti a;
    a.n = n; a.update_u = power_iterate; a.eps = 1e-12f;
    int max_cb = 0, max_nch = 0, max_rb = 0;
    for (int i = 0; i < n; ++i) {
        STEGAN_CHECK_ARG(w_orig[i] && u[i] && v[i] && scale[i] && sigma[i] && workspace[i] && rows[i] > 0 && cols[i] > 0);
        SNMulti::E& e = a.e[i];
        e.W = w_orig[i]; e.u = u[i]; e.v = v[i]; e.scale = scale[i]; e.sigma = sigma[i];
        e.R = rows[i]; e.Cn = cols[i]; e.nch = ceil_div(rows[i], SN_ROWS);
        e.partial = workspace[i]; e.svec = workspace[i] + (size_t)e.nch * e.Cn;
        e.u_save = u_save ? u_save[i] : nullptr; e.v_save = v_save ? v_save[i] : nullptr;
        max_cb = max(max_cb, ceil_div(e.Cn, 256)); max_nch = max(max_nch, e.nch); max_rb = max(max_rb, ceil_div(e.R, 4));
    }
    hipStream_t s = (hipStream_t)stream;
    if (power_iterate) {
        hipLaunchKernelGGL(sn_wtu_multi_kernel, dim3(max_cb, max_nch, n), dim3(256), 0, s, a);
        hipLaunchKernelGGL(sn_vnorm_multi_kernel, dim3(n), dim3(1024), 0, s, a);
    }
    hipLaunchKernelGGL(sn_wv_multi_kernel, dim3(max_rb, n), dim3(256), 0, s, a);
    hipLaunchKernelGGL(sn_finish_multi_kernel, dim3(n), dim3(1024), 0, s, a);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_spectral_norm_bwd_acc(const float* w_orig, const float* dw, const float* u, const float* v,
                                            const float* sigma, float* workspace, float* dw_orig, int32_t rows,
                                            int32_t cols, int32_t accumulate, void* stream) {
    STEGAN_CHECK_ARG(w_orig && dw && u && v && sigma && workspace && dw_orig && rows > 0 && cols > 0);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sn_bwd_dot_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, s, w_orig, dw, workspace, rows, cols);
    hipLaunchKernelGGL(sn_bwd_apply_kernel, dim3(nblocks((long)rows * cols)), dim3(256), 0, s, dw, workspace, u, v, sigma,
                       dw_orig, rows, cols, accumulate);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_spectral_norm_bwd_acc_multi(int32_t n, const float* const* w_orig, const float* const* dw,
                                                  const float* const* u, const float* const* v, const float* const* sigma,
                                                  float* const* workspace, float* const* dw_orig, const int32_t* rows,
                                                  const int32_t* cols, int32_t accumulate, void* stream) {
    STEGAN_CHECK_ARG(n >= 1 && w_orig && dw && u && v && sigma && workspace && dw_orig && rows && cols);
    hipStream_t s = (hipStream_t)stream;
    for (int base = 0; base < n; base += SNB_MAX) {
        SNBwdMulti a;
        a.n = n - base < SNB_MAX ? n - base : SNB_MAX;
        a.fdot[0] = 0; a.fapp[0] = 0;
        for (int i = 0; i < a.n; ++i) {
            const int j = base + i;
            STEGAN_CHECK_ARG(w_orig[j] && dw[j] && u[j] && v[j] && sigma[j] && workspace[j] && dw_orig[j] && rows[j] > 0 && cols[j] > 0);
            SNBwdMulti::E& e = a.e[i];
            e.W = w_orig[j]; e.dw = dw[j]; e.u = u[j]; e.v = v[j]; e.sigma = sigma[j]; e.rowdot = workspace[j]; e.dW = dw_orig[j];
            e.R = rows[j]; e.Cn = cols[j]; e.accumulate = accumulate;
            int nb = nblocks((long)rows[j] * cols[j]);
            if (nb > 1024) nb = 1024;
            a.fdot[i + 1] = a.fdot[i] + ceil_div(rows[j], 4);
            a.fapp[i + 1] = a.fapp[i] + nb;
        }
        for (int i = a.n; i < SNB_MAX; ++i) { a.fdot[i + 1] = a.fdot[a.n]; a.fapp[i + 1] = a.fapp[a.n]; }
        hipLaunchKernelGGL(sn_bwd_dot_multi_kernel, dim3((unsigned)a.fdot[a.n]), dim3(256), 0, s, a);
        hipLaunchKernelGGL(sn_bwd_apply_multi_kernel, dim3((unsigned)a.fapp[a.n]), dim3(256), 0, s, a);
    }
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_spectral_norm_bwd(const float* w_orig, const float* dw, const float* u, const float* v,
                                        const float* sigma, float* workspace, float* dw_orig, int32_t rows,
                                        int32_t cols, void* stream) {
    return stegan_spectral_norm_bwd_acc(w_orig, dw, u, v, sigma, workspace, dw_orig, rows, cols, 0, stream);
}
