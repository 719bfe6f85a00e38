// Frozen EMG-encoder pieces that are not convolutions: relative-position multi-head attention and
// post-LayerNorm, both on the (B, C, T) layout the conv kernels use (C = model dim, T = frames).
// T is tiny on this path (12 frames at chunk 200, 128 at chunk 2048), so these are latency-bound
// VALU kernels: one block per (batch, head) / per (batch, frame tile).
#include "common.h"

namespace {

// qkv: (B, 3*C, T), channel = which*C + h*A + a.   emb: (H, 2R-1, A).   P out: (B,H,T,T).  O out: (B, C, T)
// O_h / dqkv_h (nullable): the output also as a bf16 activation image [B][C / 8][T][8] for the 1 x 1 conv behind it
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ qkv, const float* __restrict__ emb,
                                                       float* __restrict__ P, float* __restrict__ O, int H, int A,
                                                       int T, int R, float inv_sqrt_d, unsigned short* __restrict__ O_h) {
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int Cm = H * A;
    const float* q = qkv + ((size_t)b * 3 * Cm + 0 * Cm + h * A) * T;
    const float* k = qkv + ((size_t)b * 3 * Cm + 1 * Cm + h * A) * T;
    const float* v = qkv + ((size_t)b * 3 * Cm + 2 * Cm + h * A) * T;
    const float* e = emb + (size_t)h * (2 * R - 1) * A;
    float* Pb = P + (size_t)blockIdx.x * T * T;
    // phase 1: logits
    for (int ij = threadIdx.x; ij < T * T; ij += 256) {
        const int i = ij / T, j = ij % T;
        const int rel = j - i;
        float s;
        if (rel > R - 1 || rel < -(R - 1)) {
            float qk = 0.f;
            for (int a = 0; a < A; ++a) qk += q[a * T + i] * k[a * T + j];
            s = qk * inv_sqrt_d + (-1e8f);      // zero embedding, then the -1e8 mask (transformer.py:256-263)
        } else {
            const float* er = e + (size_t)(rel + R - 1) * A;
            float qk = 0.f, qe = 0.f;
            for (int a = 0; a < A; ++a) {
                const float qa = q[a * T + i];
                qk += qa * k[a * T + j];
                qe += qa * er[a];
            }
            s = qk * inv_sqrt_d + qe;
        }
        Pb[ij] = s;
    }
    block_sync();
    // phase 2: row softmax, one wave per row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = wave; i < T; i += 4) {
        float m = -INFINITY;
        for (int j = lane; j < T; j += 64) m = fmaxf(m, Pb[i * T + j]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float se = 0.f;
        for (int j = lane; j < T; j += 64) { const float ex = expf(Pb[i * T + j] - m); Pb[i * T + j] = ex; se += ex; }
        se = wave_sum(se);
        const float inv = 1.f / se;
        for (int j = lane; j < T; j += 64) Pb[i * T + j] *= inv;
    }
    block_sync();
    // phase 3: O[a][i] = sum_j P[i][j] v[a][j]
    float* Ob = O + ((size_t)b * Cm + h * A) * T;
    for (int ai = threadIdx.x; ai < A * T; ai += 256) {
        const int a = ai / T, i = ai % T;
        float s = 0.f;
        for (int j = 0; j < T; ++j) s += Pb[i * T + j] * v[a * T + j];
        Ob[ai] = s;
        if (O_h) img_store(O_h, Cm >> 3, T, b, h * A + a, i, s);
    }
}

// dO: (B,C,T).  Outputs dqkv (B,3C,T).  dS is staged in the workspace (B,H,T,T).
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ emb,
                                                       const float* __restrict__ P, const float* __restrict__ dO,
                                                       float* __restrict__ dS, float* __restrict__ dqkv, int H, int A,
                                                       int T, int R, float inv_sqrt_d, unsigned short* __restrict__ dqkv_h) {
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int Cm = H * A;
    const size_t base = ((size_t)b * 3 * Cm + h * A) * T;
    const float* q = qkv + base;
    const float* k = qkv + base + (size_t)Cm * T;
    const float* v = qkv + base + (size_t)2 * Cm * T;
    float* dq = dqkv + base;
    float* dk = dqkv + base + (size_t)Cm * T;
    float* dv = dqkv + base + (size_t)2 * Cm * T;
    const float* e = emb + (size_t)h * (2 * R - 1) * A;
    const float* Pb = P + (size_t)blockIdx.x * T * T;
    float* dSb = dS + (size_t)blockIdx.x * T * T;
    const float* dOb = dO + ((size_t)b * Cm + h * A) * T;
    // dP[i][j] = sum_a dO[a][i] v[a][j]
    for (int ij = threadIdx.x; ij < T * T; ij += 256) {
        const int i = ij / T, j = ij % T;
        float s = 0.f;
        for (int a = 0; a < A; ++a) s += dOb[a * T + i] * v[a * T + j];
        dSb[ij] = s;
    }
    block_sync();
    // dS = P * (dP - rowsum(dP*P)), one wave per row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = wave; i < T; i += 4) {
        float r = 0.f;
        for (int j = lane; j < T; j += 64) r += dSb[i * T + j] * Pb[i * T + j];
        r = wave_sum(r);
        for (int j = lane; j < T; j += 64) dSb[i * T + j] = Pb[i * T + j] * (dSb[i * T + j] - r);
    }
    block_sync();
    for (int ai = threadIdx.x; ai < A * T; ai += 256) {
        const int a = ai / T, i = ai % T;
        // dv[a][i] = sum_r P[r][i] dO[a][r];  dk[a][i] = sum_r dS[r][i] q[a][r] / sqrt(d)
        // dq[a][i] = sum_j dS[i][j] (k[a][j]/sqrt(d) + emb[rel(j-i)][a])
        float sv = 0.f, sk = 0.f, sq = 0.f;
        for (int r = 0; r < T; ++r) {
            sv += Pb[r * T + i] * dOb[a * T + r];
            sk += dSb[r * T + i] * q[a * T + r];
            const int rel = r - i;
            float kk = k[a * T + r] * inv_sqrt_d;
            if (rel <= R - 1 && rel >= -(R - 1)) kk += e[(size_t)(rel + R - 1) * A + a];
            sq += dSb[i * T + r] * kk;
        }
        dv[ai] = sv;
        dk[ai] = sk * inv_sqrt_d;
        dq[ai] = sq;
        if (dqkv_h) {
            img_store(dqkv_h, (3 * Cm) >> 3, T, b, h * A + a, i, sq);
            img_store(dqkv_h, (3 * Cm) >> 3, T, b, Cm + h * A + a, i, sk * inv_sqrt_d);
            img_store(dqkv_h, (3 * Cm) >> 3, T, b, 2 * Cm + h * A + a, i, sv);
        }
    }
}

// ---- short sequences (T <= 32: the training chunks): q / k / v / dO of the (batch, head) and the live rows of the
// relative-position table are staged in LDS once, coalesced, and every phase then runs out of LDS.  (The general kernels
// above read them from global memory inside their inner loops -- ~100 dependent L1 round trips per thread: 36 us forward
// and 57 us backward per call at T = 12 in the round-2 profile, for 2 MFLOP of work per block.)
constexpr int ATT_TMAX = 32;

// LDS floats: q, k, v, (dO) as [A][T] | emb rows rel = -(T-1)..(T-1) as [2T-1][A + 1] (odd pitch: the T x T threads of
// the logits phase read one column of DIFFERENT rows at a time) | scores [T][T] (| dS [T][T])
__device__ __forceinline__ void att_stage(float* dst, const float* src, int n) {
    for (int i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
}

// the live rows of the relative-position table, 8 loads in flight per thread (same reason as att_stage_many)
__device__ __forceinline__ void att_stage_emb(float* es, const float* __restrict__ emb, int h, int A, int T, int R, int EP) {
    const int n = (2 * T - 1) * A;
    for (int base = 0; base < n; base += 2048) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * 256 + threadIdx.x;
            const int r = i / A, a = i - r * A, rel = r - (T - 1);
            v[u] = (i < n && rel <= R - 1 && rel >= -(R - 1)) ? emb[((size_t)h * (2 * R - 1) + rel + R - 1) * A + a] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * 256 + threadIdx.x;
            const int r = i / A, a = i - r * A;
            if (i < n) es[r * EP + a] = v[u];
        }
    }
}

// NA equally long arrays at once, all loads of a round of 4 x 256 elements per array issued before the first LDS store (the
// loop above is load, wait, store, load, ...: staging q, k, v took 14 memory round trips in sequence, most of the kernel)
template <int NA>
__device__ __forceinline__ void att_stage_many(float* const (&dst)[NA], const float* const (&src)[NA], int n) {
    for (int base = 0; base < n; base += 1024) {
        float r[NA][4];
#pragma unroll
        for (int k = 0; k < NA; ++k)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = base + u * 256 + threadIdx.x;
                r[k][u] = i < n ? src[k][i] : 0.f;
            }
#pragma unroll
        for (int k = 0; k < NA; ++k)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = base + u * 256 + threadIdx.x;
                if (i < n) dst[k][i] = r[k][u];
            }
    }
}

__global__ __launch_bounds__(256) void attn_fwd_small_kernel(const float* __restrict__ qkv, const float* __restrict__ emb,
                                                             float* __restrict__ P, float* __restrict__ O, int H, int A,
                                                             int T, int R, float inv_sqrt_d, unsigned short* __restrict__ O_h) {
    extern __shared__ float sm[];
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int Cm = H * A, AT = A * T;
    const int EP = A + 1;
    float* qs = sm; float* ks = qs + AT; float* vs = ks + AT; float* es = vs + AT; float* Ps = es + (2 * T - 1) * EP;
    const size_t base = ((size_t)b * 3 * Cm + h * A) * T;
    {
        float* const d3[3] = {qs, ks, vs};
        const float* const s3[3] = {qkv + base, qkv + base + (size_t)Cm * T, qkv + base + (size_t)2 * Cm * T};
        att_stage_many<3>(d3, s3, AT);
    }
    // table row for rel = r - (T - 1): zero when |rel| > R - 1 (masked below)
    att_stage_emb(es, emb, h, A, T, R, EP);
    block_sync();
    for (int ij = threadIdx.x; ij < T * T; ij += 256) {
        const int i = ij / T, j = ij - i * T, rel = j - i;
        const float* er = es + (rel + T - 1) * EP;
        float qk = 0.f, qe = 0.f;
        for (int a = 0; a < A; ++a) {
            const float qa = qs[a * T + i];
            qk += qa * ks[a * T + j];
            qe += qa * er[a];
        }
        Ps[ij] = (rel > R - 1 || rel < -(R - 1)) ? qk * inv_sqrt_d + (-1e8f) : qk * inv_sqrt_d + qe;
    }
    block_sync();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* Pb = P + (size_t)blockIdx.x * T * T;
    for (int i = wave; i < T; i += 4) {              // T <= 32 < 64: one element per lane
        const float sv = lane < T ? Ps[i * T + lane] : -INFINITY;
        float m = sv;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        const float ex = lane < T ? expf(sv - m) : 0.f;
        const float se = wave_sum(ex);
        const float pr = ex * (1.f / se);
        if (lane < T) { Ps[i * T + lane] = pr; Pb[i * T + lane] = pr; }
    }
    block_sync();
    float* Ob = O + ((size_t)b * Cm + h * A) * T;
    for (int ai = threadIdx.x; ai < AT; ai += 256) {
        const int a = ai / T, i = ai - a * T;
        float s = 0.f;
        for (int j = 0; j < T; ++j) s += Ps[i * T + j] * vs[a * T + j];
        Ob[ai] = s;
        if (O_h) img_store(O_h, Cm >> 3, T, b, h * A + a, i, s);
    }
}

__global__ __launch_bounds__(256) void attn_bwd_small_kernel(const float* __restrict__ qkv, const float* __restrict__ emb,
                                                             const float* __restrict__ P, const float* __restrict__ dO,
                                                             float* __restrict__ dqkv, int H, int A, int T, int R,
                                                             float inv_sqrt_d, unsigned short* __restrict__ dqkv_h) {
    extern __shared__ float sm[];
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int Cm = H * A, AT = A * T;
    const int EP = A + 1;
    float* qs = sm; float* ks = qs + AT; float* vs = ks + AT; float* ds = vs + AT; float* es = ds + AT;
    float* Ps = es + (2 * T - 1) * EP; float* Ss = Ps + T * T;
    const size_t base = ((size_t)b * 3 * Cm + h * A) * T;
    {
        float* const d4[4] = {qs, ks, vs, ds};
        const float* const s4[4] = {qkv + base, qkv + base + (size_t)Cm * T, qkv + base + (size_t)2 * Cm * T,
                                    dO + ((size_t)b * Cm + h * A) * T};
        att_stage_many<4>(d4, s4, AT);
    }
    att_stage(Ps, P + (size_t)blockIdx.x * T * T, T * T);
    att_stage_emb(es, emb, h, A, T, R, EP);
    block_sync();
    // dP[i][j] = sum_a dO[a][i] v[a][j]
    for (int ij = threadIdx.x; ij < T * T; ij += 256) {
        const int i = ij / T, j = ij - i * T;
        float s = 0.f;
        for (int a = 0; a < A; ++a) s += ds[a * T + i] * vs[a * T + j];
        Ss[ij] = s;
    }
    block_sync();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = wave; i < T; i += 4) {
        const float pv = lane < T ? Ps[i * T + lane] : 0.f, dp = lane < T ? Ss[i * T + lane] : 0.f;
        const float r = wave_sum(dp * pv);
        if (lane < T) Ss[i * T + lane] = pv * (dp - r);
    }
    block_sync();
    float* dq = dqkv + base;
    float* dk = dqkv + base + (size_t)Cm * T;
    float* dv = dqkv + base + (size_t)2 * Cm * T;
    for (int ai = threadIdx.x; ai < AT; ai += 256) {
        const int a = ai / T, i = ai - a * T;
        float sv = 0.f, sk = 0.f, sq = 0.f;
        for (int r = 0; r < T; ++r) {
            sv += Ps[r * T + i] * ds[a * T + r];
            sk += Ss[r * T + i] * qs[a * T + r];
            sq += Ss[i * T + r] * (ks[a * T + r] * inv_sqrt_d + es[(r - i + T - 1) * EP + a]);    // zero row when masked
        }
        dv[ai] = sv;
        dk[ai] = sk * inv_sqrt_d;
        dq[ai] = sq;
        if (dqkv_h) {
            img_store(dqkv_h, (3 * Cm) >> 3, T, b, h * A + a, i, sq);
            img_store(dqkv_h, (3 * Cm) >> 3, T, b, Cm + h * A + a, i, sk * inv_sqrt_d);
            img_store(dqkv_h, (3 * Cm) >> 3, T, b, 2 * Cm + h * A + a, i, sv);
        }
    }
}

// y = LayerNorm_C(x + r) * gamma + beta on (B, C, T); one wave per (b, t) column; saves xhat and rstd
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ y, float* __restrict__ xhat,
                                                     float* __restrict__ rstd, int B, int Cm, int T, float eps,
                                                     unsigned short* __restrict__ y_h) {
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= B * T) return;
    const int lane = threadIdx.x & 63;
    const int b = col / T, t = col % T;
    const size_t base = (size_t)b * Cm * T + t;
    float s = 0.f;
    for (int c = lane; c < Cm; c += 64) s += x[base + (size_t)c * T] + (r ? r[base + (size_t)c * T] : 0.f);
    s = wave_sum(s);
    const float mean = s / (float)Cm;
    float vs = 0.f;
    for (int c = lane; c < Cm; c += 64) {
        const float d = x[base + (size_t)c * T] + (r ? r[base + (size_t)c * T] : 0.f) - mean;
        vs += d * d;
    }
    vs = wave_sum(vs);
    const float rs = rsqrtf(vs / (float)Cm + eps);
    for (int c = lane; c < Cm; c += 64) {
        const float xh = (x[base + (size_t)c * T] + (r ? r[base + (size_t)c * T] : 0.f) - mean) * rs;
        if (xhat) xhat[base + (size_t)c * T] = xh;
        const float yv = xh * gamma[c] + beta[c];
        y[base + (size_t)c * T] = yv;
        if (y_h) img_store(y_h, Cm >> 3, T, b, c, t, yv);
    }
    if (lane == 0 && rstd) rstd[col] = rs;
}

// dz = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = gamma * dy     (z = x + r; dx = dr = dz)
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     float* __restrict__ dz, int B, int Cm, int T,
                                                     unsigned short* __restrict__ dz_h) {
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= B * T) return;
    const int lane = threadIdx.x & 63;
    const int b = col / T, t = col % T;
    const size_t base = (size_t)b * Cm * T + t;
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < Cm; c += 64) {
        const float g = gamma[c] * dy[base + (size_t)c * T];
        s1 += g;
        s2 += g * xhat[base + (size_t)c * T];
    }
    s1 = wave_sum(s1) / (float)Cm;
    s2 = wave_sum(s2) / (float)Cm;
    const float rs = rstd[col];
    for (int c = lane; c < Cm; c += 64) {
        const float g = gamma[c] * dy[base + (size_t)c * T];
        const float dv_ = rs * (g - s1 - xhat[base + (size_t)c * T] * s2);
        dz[base + (size_t)c * T] = dv_;
        if (dz_h) img_store(dz_h, Cm >> 3, T, b, c, t, dv_);
    }
}

// ---- the same two operators with one block per BATCH ELEMENT (short rows: the encoder runs them at T = 12) -------------
// The kernels above give a wave one (b, t) column and walk it with a stride of T floats: every lane of a load touches its
// own 64-byte segment (17 us for a 2.4 MB tensor).  Here a block owns the contiguous (C, T) slab of one batch element,
// reads it with unit stride into registers (LN_EPT values per thread) and reduces per t through LDS: the block has T * P
// threads, so thread i only ever sees t = i % T; the P partials of a column are added by one wave (two barriers).
constexpr int LN_EPT = 16;

__device__ __forceinline__ float ln_column_sum(float v, float* red, int tid, int T, int P) {
    // partials of column t sit at red[t + T k], k < P: the waves share the columns, a wave adds one with shuffles
    float* tot = red + T * P;
    red[tid] = v;
    block_sync();
    const int wave = tid >> 6, lane = tid & 63, nw = (T * P) >> 6;          // whole waves only (wave_sum needs 64 lanes)
    for (int t = wave; t < T && wave < nw; t += nw) {
        float a = 0.f;
        for (int k = lane; k < P; k += 64) a += red[t + T * k];
        a = wave_sum(a);
        if (lane == 0) tot[t] = a;
    }
    block_sync();
    return tot[tid % T];          // (red / tot are rewritten only behind the next reduction's first barrier)
}

// Round 5: a block owns Tg of the T columns of its batch element (grid = B x T / Tg): at T = 12, B = 64 the 64 blocks of the
// first form left three quarters of the chip idle on the encoder branch's critical chain (13.6 / 11.4 us per launch, 24 launches
// a step); four column groups make 256 blocks with a quarter of the loads each.  Thread i sees column t0 + i % Tg and channels
// i / Tg + e P.  Tg == T is the first form.
__global__ __launch_bounds__(1024) void ln_fwd_slab_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ y, float* __restrict__ xhat,
                                                           float* __restrict__ rstd, int Cm, int T, int Tg, int P, float eps,
                                                           unsigned short* __restrict__ y_h) {
    extern __shared__ float red[];
    const int G = T / Tg, b = (int)blockIdx.x / G, t0 = ((int)blockIdx.x - b * G) * Tg;
    const int tid = threadIdx.x, c0 = tid / Tg, tl = tid - c0 * Tg;
    const size_t off = (size_t)b * Cm * T + t0 + tl;
    float z[LN_EPT], gm[LN_EPT], bt[LN_EPT];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < LN_EPT; ++e) {
        const int c = c0 + e * P;
        const bool in = c < Cm;
        z[e] = in ? x[off + (size_t)c * T] + (r ? r[off + (size_t)c * T] : 0.f) : 0.f;
        gm[e] = in ? gamma[c] : 0.f;
        bt[e] = in ? beta[c] : 0.f;
        s += z[e];
    }
    const float mean = ln_column_sum(s, red, tid, Tg, P) / (float)Cm;
    float vs = 0.f;
#pragma unroll
    for (int e = 0; e < LN_EPT; ++e) {
        const float d = (c0 + e * P < Cm) ? z[e] - mean : 0.f;
        vs += d * d;
    }
    const float rs = rsqrtf(ln_column_sum(vs, red, tid, Tg, P) / (float)Cm + eps);
#pragma unroll
    for (int e = 0; e < LN_EPT; ++e) {
        const int c = c0 + e * P;
        if (c < Cm) {
            const float xh = (z[e] - mean) * rs;
            if (xhat) xhat[off + (size_t)c * T] = xh;
            const float yv = xh * gm[e] + bt[e];
            y[off + (size_t)c * T] = yv;
            if (y_h) img_store(y_h, Cm >> 3, T, b, c, t0 + tl, yv);
        }
    }
    if (tid < Tg && rstd) rstd[b * T + t0 + tid] = rs;
}

__global__ __launch_bounds__(1024) void ln_bwd_slab_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           float* __restrict__ dz, int Cm, int T, int Tg, int P,
                                                           unsigned short* __restrict__ dz_h) {
    extern __shared__ float red[];
    const int G = T / Tg, b = (int)blockIdx.x / G, t0 = ((int)blockIdx.x - b * G) * Tg;
    const int tid = threadIdx.x, c0 = tid / Tg, tl = tid - c0 * Tg;
    const size_t off = (size_t)b * Cm * T + t0 + tl;
    const float rs = rstd[b * T + t0 + tl];
    float g[LN_EPT], xh[LN_EPT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < LN_EPT; ++e) {
        const int c = c0 + e * P;
        const bool in = c < Cm;
        g[e] = in ? gamma[c] * dy[off + (size_t)c * T] : 0.f;
        xh[e] = in ? xhat[off + (size_t)c * T] : 0.f;
        s1 += g[e];
        s2 += g[e] * xh[e];
    }
    s1 = ln_column_sum(s1, red, tid, Tg, P) / (float)Cm;
    s2 = ln_column_sum(s2, red, tid, Tg, P) / (float)Cm;
#pragma unroll
    for (int e = 0; e < LN_EPT; ++e) {
        const int c = c0 + e * P;
        if (c < Cm) {
            const float dv_ = rs * (g[e] - s1 - xh[e] * s2);
            dz[off + (size_t)c * T] = dv_;
            if (dz_h) img_store(dz_h, Cm >> 3, T, b, c, t0 + tl, dv_);
        }
    }
}

// threads per batch element for the slab kernels: T * P, P the largest power of two with T * P <= 1024 (0: rows too long or
// the slab does not fit LN_EPT values per thread -- the per-column kernels take it)
inline int ln_slab_p(int Cm, int T) {
    static const bool off = getenv("STEGAN_LN_COLUMNS") != nullptr;          // experiments: the first version's kernels
    if (off || T > 256) return 0;
    int P = 1;
    while (T * P * 2 <= 1024) P *= 2;
    if ((long)Cm > (long)LN_EPT * P || T * P < 64) return 0;
    return P;
}

// columns per block: the divisor Tg of T that brings the grid B * T / Tg closest to one block per CU from below (Tg == T when
// the batch alone fills the chip), with whole waves in the block
inline int ln_slab_tg(int B, int Cm, int T) {
    static const char* e = getenv("STEGAN_LN_SPLIT");                        // experiments: column groups per batch element
    const int want = e ? atoi(e) : 0;
    int best = T;
    for (int G = 1; G <= T; ++G) {
        if (T % G) continue;
        const int Tg = T / G;
        if (!ln_slab_p(Cm, Tg)) continue;
        if (want ? G == want : (long)B * G <= 320) best = Tg;
    }
    return best;
}

}  // namespace

extern "C" int stegan_relpos_attention_fwd_img(const float* qkv, const float* rel_emb, float* probs, float* out, uint16_t* out_h,
                                               int32_t B, int32_t H, int32_t A, int32_t T, int32_t max_rel, void* stream) {
    STEGAN_CHECK_ARG(qkv && rel_emb && probs && out && B > 0 && H > 0 && A > 0 && T > 0 && max_rel > 0);
    STEGAN_CHECK_ARG(!out_h || (H * A) % 8 == 0);
    const size_t lds_s = ((size_t)3 * A * T + (size_t)(2 * T - 1) * (A + 1) + (size_t)T * T) * sizeof(float);
    if (T <= ATT_TMAX && lds_s <= 64 * 1024) {
        hipLaunchKernelGGL(attn_fwd_small_kernel, dim3(B * H), dim3(256), lds_s, (hipStream_t)stream, qkv, rel_emb, probs, out, H,
                           A, T, max_rel, 1.f / sqrtf((float)A), out_h);
        STEGAN_LAUNCH_CHECK();
        return STEGAN_OK;
    }
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(B * H), dim3(256), 0, (hipStream_t)stream, qkv, rel_emb, probs, out, H, A, T,
                       max_rel, 1.f / sqrtf((float)A), out_h);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_relpos_attention_fwd(const float* qkv, const float* rel_emb, float* probs, float* out, int32_t B,
                                           int32_t H, int32_t A, int32_t T, int32_t max_rel, void* stream) {
    return stegan_relpos_attention_fwd_img(qkv, rel_emb, probs, out, nullptr, B, H, A, T, max_rel, stream);
}

extern "C" int stegan_relpos_attention_bwd_img(const float* qkv, const float* rel_emb, const float* probs, const float* dout,
                                               float* workspace, float* dqkv, uint16_t* dqkv_h, int32_t B, int32_t H, int32_t A,
                                               int32_t T, int32_t max_rel, void* stream) {
    STEGAN_CHECK_ARG(qkv && rel_emb && probs && dout && workspace && dqkv && B > 0 && H > 0 && A > 0 && T > 0);
    STEGAN_CHECK_ARG(!dqkv_h || (H * A) % 8 == 0);
    const size_t lds_s = ((size_t)4 * A * T + (size_t)(2 * T - 1) * (A + 1) + (size_t)2 * T * T) * sizeof(float);
    if (T <= ATT_TMAX && lds_s <= 64 * 1024) {
        hipLaunchKernelGGL(attn_bwd_small_kernel, dim3(B * H), dim3(256), lds_s, (hipStream_t)stream, qkv, rel_emb, probs, dout,
                           dqkv, H, A, T, max_rel, 1.f / sqrtf((float)A), dqkv_h);
        STEGAN_LAUNCH_CHECK();
        return STEGAN_OK;
    }
    hipLaunchKernelGGL(attn_bwd_kernel, dim3(B * H), dim3(256), 0, (hipStream_t)stream, qkv, rel_emb, probs, dout,
                       workspace, dqkv, H, A, T, max_rel, 1.f / sqrtf((float)A), dqkv_h);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_relpos_attention_bwd(const float* qkv, const float* rel_emb, const float* probs, const float* dout,
                                           float* workspace, float* dqkv, int32_t B, int32_t H, int32_t A, int32_t T,
                                           int32_t max_rel, void* stream) {
    return stegan_relpos_attention_bwd_img(qkv, rel_emb, probs, dout, workspace, dqkv, nullptr, B, H, A, T, max_rel, stream);
}

extern "C" int stegan_add_layernorm_fwd_img(const float* x, const float* residual, const float* gamma, const float* beta,
                                            float* y, uint16_t* y_h, float* xhat, float* rstd, int32_t B, int32_t C, int32_t T,
                                            float eps, void* stream) {
    STEGAN_CHECK_ARG(x && gamma && beta && y && B > 0 && C > 0 && T > 0);
    STEGAN_CHECK_ARG(!y_h || C % 8 == 0);
    if (ln_slab_p(C, T)) {
        const int Tg = ln_slab_tg(B, C, T), P = ln_slab_p(C, Tg);
        hipLaunchKernelGGL(ln_fwd_slab_kernel, dim3(B * (T / Tg)), dim3(Tg * P), (size_t)Tg * (P + 1) * sizeof(float),
                           (hipStream_t)stream, x, residual, gamma, beta, y, xhat, rstd, C, T, Tg, P, eps, y_h);
        STEGAN_LAUNCH_CHECK();
        return STEGAN_OK;
    }
    hipLaunchKernelGGL(ln_fwd_kernel, dim3(ceil_div(B * T, 4)), dim3(256), 0, (hipStream_t)stream, x, residual, gamma,
                       beta, y, xhat, rstd, B, C, T, eps, y_h);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_add_layernorm_fwd(const float* x, const float* residual, const float* gamma, const float* beta,
                                        float* y, float* xhat, float* rstd, int32_t B, int32_t C, int32_t T, float eps,
                                        void* stream) {
    return stegan_add_layernorm_fwd_img(x, residual, gamma, beta, y, nullptr, xhat, rstd, B, C, T, eps, stream);
}

extern "C" int stegan_add_layernorm_bwd_img(const float* dy, const float* xhat, const float* rstd, const float* gamma,
                                            float* dz, uint16_t* dz_h, int32_t B, int32_t C, int32_t T, void* stream) {
    STEGAN_CHECK_ARG(dy && xhat && rstd && gamma && dz && B > 0 && C > 0 && T > 0);
    STEGAN_CHECK_ARG(!dz_h || C % 8 == 0);
    if (ln_slab_p(C, T)) {
        const int Tg = ln_slab_tg(B, C, T), P = ln_slab_p(C, Tg);
        hipLaunchKernelGGL(ln_bwd_slab_kernel, dim3(B * (T / Tg)), dim3(Tg * P), (size_t)Tg * (P + 1) * sizeof(float),
                           (hipStream_t)stream, dy, xhat, rstd, gamma, dz, C, T, Tg, P, dz_h);
        STEGAN_LAUNCH_CHECK();
        return STEGAN_OK;
    }
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(ceil_div(B * T, 4)), dim3(256), 0, (hipStream_t)stream, dy, xhat, rstd, gamma,
                       dz, B, C, T, dz_h);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_add_layernorm_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma,
                                        float* dz, int32_t B, int32_t C, int32_t T, void* stream) {
    return stegan_add_layernorm_bwd_img(dy, xhat, rstd, gamma, dz, nullptr, B, C, T, stream);
}
