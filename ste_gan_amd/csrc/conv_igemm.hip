// Direct 1-D convolution as an implicit GEMM on the fp32 matrix cores of gfx950.
//
//   C[m][n] = sum_r A[m][r] * B[r][n]
//
//   mode        m          n            r (reduction)      A                    B
//   FWD         co         (b, t_out)   (ci, k)            w[co][ci][k]         act(up(x))[b][ci][t*s+k*d-p]
//   BWD_DATA    ci         (b, t_eff)   (co, k)            w[co][ci][k]         scale[co]*dy[b][co][(t+p-k*d)/s]
//   BWD_WEIGHT  co         (ci, k)      (b, t_out)         dy[b][co][t]         act(up(x))[b][ci][t*s+k*d-p]
//
// The im2col matrix never exists in HBM: every block gathers its B tile straight from the
// (B,C,T) activation into LDS (zero padding, dilation, stride, nearest-x2 upsample, the input
// ReLU and dead-tap skipping are all resolved in that gather), stages the weight tile next to
// it, and four wave64s run v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain) over the LDS tiles.
// Weight-norm / spectral-norm / BatchNorm factors enter as a per-output-channel scale in the
// epilogue (FWD) or on the dy operand (BWD_DATA), so normalised weights are never written out.
//
// Tile: block = 256 threads = 2x2 waves, wave tile = (32*TM) x (32*TN), K-chunk = 32.
// LDS image: [r][m] and [r][n] with +1 padding; operand reads are ds_read_b32 with the 32 lanes
// of each half-wave on consecutive dwords (conflict-free), lanes 32-63 on the next r row, which
// is exactly the A[i=l&31][k=l>>5] / B[k=l>>5][j=l&31] operand map of the 32x32x2 MFMA.
#include "common.h"

namespace {

enum { MODE_FWD = 0, MODE_BWD_DATA = 1, MODE_BWD_WEIGHT = 2 };
constexpr int KC = 32;

struct ConvK {
    const float* x;      // FWD/BWD_WEIGHT: conv input (B,Cin,Tin).  BWD_DATA: unused
    const float* w;      // (Cout, Cin_g, K) raw direction tensor
    const float* dy;     // BWD_*: (B,Cout,Tout)
    const float* scale;  // (Cout) or null
    const float* bias;   // (Cout) or null           (FWD)
    const float* res;    // residual / add_src or null
    const float* mask;   // BWD_DATA: pre-activation input (B,Cin,Tin) or null
    float* out;          // FWD: y, BWD_DATA: dx (B,Cin,Te), BWD_WEIGHT: dw (accumulated; with `partial` the slices' slabs)
    float* db;           // BWD_WEIGHT: bias gradient (accumulated) or null -- row sums of the dy tile
    float* partial;      // BWD_WEIGHT with > 1 slice of the reduction: [splits][dw numel] slabs + [splits][Cout] bias sums,
                         // summed in a fixed order by igemm_wgrad_reduce_kernel (no float atomics: reproducible gradients)
    long dw_numel;
    int B, Cin, Cout, K, stride, dil, pad, Tin, Tout, Te, ups;  // ups = log2(upsample)
    int in_act, out_act, out_btc;
    int Cin_g, Cout_g;
    int k_lo, K_eff;     // live tap range: taps outside it only ever see zero padding
    int M, N, R;         // GEMM extents per group
    int m_tiles;         // M tiles per group
    int r_chunks_per_split;
    FastDiv fdK;         // / K_eff
    FastDiv fdT;         // / Tout (FWD, BWD_WEIGHT) or / Te (BWD_DATA)
    FastDiv fdS;         // / stride
};

// Address generation is separated from the loads: every element yields (pointer, valid) and the kernel then issues all
// loads of a chunk back to back on clamped, always-in-bounds addresses and zeroes the invalid ones with a select when
// it writes LDS.  (A branch around a load makes hipcc wait for each load in turn: 16 serialised round trips per chunk.)
template <int MODE>
__device__ __forceinline__ const float* addr_a(const ConvK& p, int g, int m, int r, bool& ok) {
    ok = m < p.M && r < p.R;
    const int mm = ok ? m : 0, rr = ok ? r : 0;
    if (MODE == MODE_FWD) {
        uint32_t ci, ke;
        fd_divmod((uint32_t)rr, p.fdK, ci, ke);
        return p.w + ((size_t)(g * p.Cout_g + mm) * p.Cin_g + ci) * p.K + ke + p.k_lo;
    } else if (MODE == MODE_BWD_DATA) {
        uint32_t co, ke;
        fd_divmod((uint32_t)rr, p.fdK, co, ke);
        return p.w + ((size_t)(g * p.Cout_g + co) * p.Cin_g + mm) * p.K + ke + p.k_lo;
    } else {
        uint32_t b, t;
        fd_divmod((uint32_t)rr, p.fdT, b, t);
        return p.dy + ((size_t)b * p.Cout + g * p.Cout_g + mm) * p.Tout + t;
    }
}

// (activated, upsampled, zero padded) conv input for group-local channel ci, live-tap index ke
__device__ __forceinline__ const float* addr_x(const ConvK& p, int g, uint32_t b, uint32_t t, uint32_t ci, uint32_t ke,
                                               bool& ok) {
    const int tin = (int)t * p.stride + (int)(ke + p.k_lo) * p.dil - p.pad;
    ok = ok && tin >= 0 && tin < p.Te;
    const int tt = ok ? tin : 0;
    return p.x + ((size_t)b * p.Cin + g * p.Cin_g + ci) * p.Tin + (tt >> p.ups);
}

template <int MODE>
__device__ __forceinline__ const float* addr_b(const ConvK& p, int g, int r, int n, bool& ok, const float*& scale_ptr) {
    ok = r < p.R && n < p.N;
    const int rr = ok ? r : 0, nn = ok ? n : 0;
    scale_ptr = nullptr;
    if (MODE == MODE_FWD) {
        uint32_t ci, ke, b, t;
        fd_divmod((uint32_t)rr, p.fdK, ci, ke);
        fd_divmod((uint32_t)nn, p.fdT, b, t);
        return addr_x(p, g, b, t, ci, ke, ok);
    } else if (MODE == MODE_BWD_DATA) {
        uint32_t co, ke, b, te;
        fd_divmod((uint32_t)rr, p.fdK, co, ke);
        fd_divmod((uint32_t)nn, p.fdT, b, te);
        const int tt = (int)te + p.pad - (int)(ke + p.k_lo) * p.dil;
        uint32_t q, rem;
        fd_divmod((uint32_t)(tt < 0 ? 0 : tt), p.fdS, q, rem);
        ok = ok && tt >= 0 && rem == 0 && (int)q < p.Tout;
        const int cog = g * p.Cout_g + co;
        if (p.scale) scale_ptr = p.scale + cog;
        return p.dy + ((size_t)b * p.Cout + cog) * p.Tout + (ok ? q : 0);
    } else {
        uint32_t ci, ke, b, t;
        fd_divmod((uint32_t)nn, p.fdK, ci, ke);
        fd_divmod((uint32_t)rr, p.fdT, b, t);
        return addr_x(p, g, b, t, ci, ke, ok);
    }
}

template <int MODE, int TM, int TN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvK p) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int LDA = BM + 1, LDB = BN + 1;
    // staging thread maps: "r fast" = lanes along r (global contiguity along r), else along m / n
    constexpr bool A_RFAST = (MODE != MODE_BWD_DATA);
    constexpr bool B_RFAST = (MODE == MODE_BWD_WEIGHT);
    constexpr int A_PER = KC * BM / 256;
    constexpr int B_PER = KC * BN / 256;

    __shared__ float lds[KC * LDA + KC * LDB];
    float* As = lds;
    float* Bs = lds + KC * LDA;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * BN;
    const int g = blockIdx.y / p.m_tiles;
    const int m0 = (blockIdx.y % p.m_tiles) * BM;

    int chunk_lo = 0, chunk_hi = (p.R + KC - 1) / KC;
    if (MODE == MODE_BWD_WEIGHT) {
        chunk_lo = blockIdx.z * p.r_chunks_per_split;
        chunk_hi = min(chunk_hi, chunk_lo + p.r_chunks_per_split);
        if (chunk_lo >= chunk_hi) return;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float ra[A_PER], rb[B_PER], rs[MODE == MODE_BWD_DATA ? B_PER : 1];
    unsigned ok_a = 0, ok_b = 0;          // validity bits of the staged elements (applied when LDS is written)

    auto load_chunk = [&](int r0) {
        ok_a = 0; ok_b = 0;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int m, r;
            if (A_RFAST) { r = tid & 31; m = (tid >> 5) + 8 * i; }
            else         { m = tid % BM; r = tid / BM + (256 / BM) * i; }
            bool ok;
            const float* ptr = addr_a<MODE>(p, g, m0 + m, r0 + r, ok);
            ra[i] = *ptr;
            ok_a |= (ok ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            int n, r;
            if (B_RFAST) { r = tid & 31; n = (tid >> 5) + 8 * i; }
            else         { n = tid % BN; r = tid / BN + (256 / BN) * i; }
            bool ok;
            const float* sp;
            const float* ptr = addr_b<MODE>(p, g, r0 + r, n0 + n, ok, sp);
            rb[i] = *ptr;
            if (MODE == MODE_BWD_DATA) rs[i] = sp ? *sp : 1.f;
            ok_b |= (ok ? 1u : 0u) << i;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int m, r;
            if (A_RFAST) { r = tid & 31; m = (tid >> 5) + 8 * i; }
            else         { m = tid % BM; r = tid / BM + (256 / BM) * i; }
            As[r * LDA + m] = ((ok_a >> i) & 1u) ? ra[i] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            int n, r;
            if (B_RFAST) { r = tid & 31; n = (tid >> 5) + 8 * i; }
            else         { n = tid % BN; r = tid / BN + (256 / BN) * i; }
            float v = rb[i];
            if (MODE != MODE_BWD_DATA && p.in_act == STEGAN_ACT_RELU) v = fmaxf(v, 0.f);
            if (MODE == MODE_BWD_DATA) v *= rs[i];
            Bs[r * LDB + n] = ((ok_b >> i) & 1u) ? v : 0.f;
        }
    };

    float bsum = 0.f;                     // BWD_WEIGHT, first N tile only: sum over (b,t) of dy[co = m0 + tid]
    const bool do_bias = MODE == MODE_BWD_WEIGHT && p.db != nullptr && blockIdx.x == 0 && tid < BM;
    load_chunk(chunk_lo * KC);
    for (int c = chunk_lo; c < chunk_hi; ++c) {
        store_chunk();
        block_sync();
        if (do_bias) {
#pragma unroll 8
            for (int r = 0; r < KC; ++r) bsum += As[r * LDA + tid];
        }
        if (c + 1 < chunk_hi) load_chunk((c + 1) * KC);   // global loads in flight under the MFMAs
        const int half = lane >> 5, l31 = lane & 31;
#pragma unroll 4
        for (int kk = 0; kk < KC; kk += 2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[(kk + half) * LDA + (wm * TM + i) * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[(kk + half) * LDB + (wn * TN + j) * 32 + l31];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        block_sync();
    }

    if (do_bias && m0 + tid < p.M) {                     // (first N tile of every (M tile, slice): the channel's only writer)
        const int ch = g * p.Cout_g + m0 + tid;
        if (p.partial) p.partial[(size_t)gridDim.z * p.dw_numel + (size_t)blockIdx.z * p.Cout + ch] = bsum;
        else p.db[ch] += bsum;
    }
    // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + l31;
        if (n >= p.N) continue;
        uint32_t q0, q1;   // FWD/BWD_DATA: (b, t)   BWD_WEIGHT: (ci, ke)
        fd_divmod((uint32_t)n, (MODE == MODE_BWD_WEIGHT) ? p.fdK : p.fdT, q0, q1);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (m >= p.M) continue;
                float v = acc[i][j][e];
                if (MODE == MODE_FWD) {
                    const int co = g * p.Cout_g + m;
                    if (p.scale) v *= p.scale[co];
                    if (p.bias) v += p.bias[co];
                    const size_t idx = ((size_t)q0 * p.Cout + co) * p.Tout + q1;
                    if (p.res) v += p.res[idx];
                    if (p.out_act == STEGAN_ACT_LRELU) v = v > 0.f ? v : 0.1f * v;
                    else if (p.out_act == STEGAN_ACT_RELU) v = fmaxf(v, 0.f);
                    else if (p.out_act == STEGAN_ACT_TANH) v = tanhf(v);
                    if (p.out_btc) p.out[((size_t)q0 * p.Tout + q1) * p.Cout + co] = v;
                    else p.out[idx] = v;
                } else if (MODE == MODE_BWD_DATA) {
                    const int ci = g * p.Cin_g + m;
                    if (p.mask) {
                        const float xv = p.mask[((size_t)q0 * p.Cin + ci) * p.Tin + (q1 >> p.ups)];
                        if (!(xv > 0.f)) v = 0.f;
                    }
                    const size_t idx = ((size_t)q0 * p.Cin + ci) * p.Te + q1;
                    if (p.res) v += p.res[idx];
                    p.out[idx] = v;
                } else {
                    const size_t idx = ((size_t)(g * p.Cout_g + m) * p.Cin_g + q0) * p.K + q1 + p.k_lo;
                    if (p.partial) p.partial[(size_t)blockIdx.z * p.dw_numel + idx] = v;      // this slice's slab
                    else p.out[idx] += v;                    // ONE slice: every element has a single writer
                }
            }
        }
    }
}

// dW += the slices' slabs, db += their bias sums, each in slice order (live taps only: dead taps were never written)
__global__ __launch_bounds__(256) void igemm_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                 long numel, int splits, int K, int k_lo, int K_eff,
                                                                 float* __restrict__ db, int Cout) {
    if (db) {
        const float* bp = partial + (size_t)splits * numel;
        for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < Cout; c += (long)gridDim.x * 256) {
            float v = 0.f;
            for (int s = 0; s < splits; ++s) v += bp[(size_t)s * Cout + c];
            db[c] += v;
        }
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < numel; i += (long)gridDim.x * 256) {
        const int k = (int)(i % K);
        if (k < k_lo || k >= k_lo + K_eff) continue;
        const float* src = partial + i;
        // 16 independent loads in flight per round (these dW are tiny -- a few blocks -- so the kernel is a chain of load
        // round trips: 64 slabs in 4 rounds); fixed summation order
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = 0.f;
        int s = 0;
        for (; s + 16 <= splits; s += 16) {
            float t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = src[(size_t)(s + u) * numel];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] += t[u];
        }
        for (; s < splits; ++s) v[0] += src[(size_t)s * numel];
        float a = 0.f;
#pragma unroll
        for (int u = 0; u < 16; u += 4) a += (v[u] + v[u + 1]) + (v[u + 2] + v[u + 3]);
        dw[i] += a;
    }
}

__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ dy, float* __restrict__ db,
                                                        int B, int C, int T, int per, FastDiv fdT, int vec) {
    bias_grad_body(dy, db, B, C, T, per, fdT, vec, blockIdx.x, blockIdx.y);
}

inline void launch_bias_grad(const float* dy, float* db, int B, int C, int T, hipStream_t s) {
    const int sl = bias_grad_slices(B, C, T), per = (B + sl - 1) / sl;
    const int vec = (T % 4 == 0) && ((uintptr_t)dy % 16 == 0);
    hipLaunchKernelGGL(bias_grad_kernel, dim3(C, (B + per - 1) / per), dim3(256), 0, s, dy, db, B, C, T, per,
                       make_fastdiv((uint32_t)(vec ? T / 4 : T)), vec);
}

int fill_common(const stegan_conv_desc* d, ConvK& p) {
    STEGAN_CHECK_ARG(d != nullptr);
    STEGAN_CHECK_ARG(d->B > 0 && d->Cin > 0 && d->Cout > 0 && d->groups > 0 && d->K > 0);
    STEGAN_CHECK_ARG(d->Cin % d->groups == 0 && d->Cout % d->groups == 0);
    STEGAN_CHECK_ARG(d->stride >= 1 && d->dilation >= 1 && d->pad >= 0 && d->Tin > 0);
    STEGAN_CHECK_ARG(d->upsample == 1 || d->upsample == 2);
    STEGAN_CHECK_ARG(d->in_act == STEGAN_ACT_NONE || d->in_act == STEGAN_ACT_RELU);
    const int Te = d->Tin * d->upsample;
    const int span = Te + 2 * d->pad - d->dilation * (d->K - 1) - 1;
    STEGAN_CHECK_ARG(span >= 0);
    STEGAN_CHECK_ARG(d->Tout == span / d->stride + 1);
    STEGAN_CHECK_ARG((int64_t)d->B * (d->Cin > d->Cout ? d->Cin : d->Cout) * (Te > d->Tout ? Te : d->Tout) < (1ll << 31));
    p.B = d->B; p.Cin = d->Cin; p.Cout = d->Cout; p.K = d->K; p.stride = d->stride; p.dil = d->dilation;
    p.pad = d->pad; p.Tin = d->Tin; p.Tout = d->Tout; p.Te = Te; p.ups = d->upsample == 2 ? 1 : 0;
    p.in_act = d->in_act; p.out_act = d->out_act; p.out_btc = d->out_btc;
    p.Cin_g = d->Cin / d->groups; p.Cout_g = d->Cout / d->groups;
    // live taps: tap k touches real data for some t_out iff  [k*d - pad, (Tout-1)*s + k*d - pad] meets [0, Te)
    int k_lo = d->K, k_hi = -1;
    for (int k = 0; k < d->K; ++k) {
        const int lo = k * d->dilation - d->pad, hi = (d->Tout - 1) * d->stride + lo;
        if (hi >= 0 && lo < Te) { if (k < k_lo) k_lo = k; k_hi = k; }
    }
    STEGAN_CHECK_ARG(k_hi >= k_lo);
    p.k_lo = k_lo; p.K_eff = k_hi - k_lo + 1;
    p.fdK = make_fastdiv((uint32_t)p.K_eff);
    p.fdS = make_fastdiv((uint32_t)d->stride);
    p.r_chunks_per_split = 1 << 30;
    p.x = p.w = p.dy = p.scale = p.bias = p.res = p.mask = nullptr;
    p.out = nullptr;
    p.db = nullptr;
    p.partial = nullptr;
    p.dw_numel = (long)d->Cout * p.Cin_g * d->K;
    return STEGAN_OK;
}

// tile choice: 128x128 when that still gives every CU >= 2 blocks, else 64x64
inline bool use_big_tile(const ConvK& p, int groups, int splits) {
    const long tiles128 = (long)ceil_div(p.M, 128) * ceil_div(p.N, 128) * groups * splits;
    return tiles128 >= 512 && p.M >= 96;
}

inline int bwd_weight_splits(ConvK& p, int groups, long max_splits = 1 << 30) {
    // split the (b,t) reduction so the grid fills the chip; the slices write slabs that a second kernel sums in order
    const int chunks = ceil_div(p.R, KC);
    const long tiles = (long)ceil_div(p.M, 64) * ceil_div(p.N, 64) * groups;
    static const char* env_tgt = getenv("STEGAN_WGRAD_TARGET");          // blocks to aim for (experiments)
    const long target = env_tgt ? atol(env_tgt) : 256;        // (1024 at first: 7.49 vs 7.46 ms/step -- these run beside other streams)
    int splits = (int)((target + tiles - 1) / tiles);
    // every split costs a slab written and read back (these layers' dW are small: 8 -> 128 k15 is 61 KB)
    static const char* env_cap = getenv("STEGAN_WGRAD_SPLIT_CAP");
    const int cap = env_cap ? atoi(env_cap) : 64;
    if (splits > cap) splits = cap;
    if (splits > max_splits) splits = (int)max_splits;
    if (splits > chunks) splits = chunks;
    if (splits < 1) splits = 1;
    p.r_chunks_per_split = ceil_div(chunks, splits);
    return ceil_div(chunks, p.r_chunks_per_split);
}

template <int MODE>
int launch(const ConvK& p, int groups, int splits, hipStream_t s) {
    ConvK q = p;
    if (use_big_tile(p, groups, splits)) {
        q.m_tiles = ceil_div(p.M, 128);
        dim3 grid(ceil_div(p.N, 128), q.m_tiles * groups, splits);
        hipLaunchKernelGGL((conv_igemm_kernel<MODE, 2, 2>), grid, dim3(256), 0, s, q);
    } else {
        q.m_tiles = ceil_div(p.M, 64);
        dim3 grid(ceil_div(p.N, 64), q.m_tiles * groups, splits);
        hipLaunchKernelGGL((conv_igemm_kernel<MODE, 1, 1>), grid, dim3(256), 0, s, q);
    }
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

}  // namespace

extern "C" int stegan_conv1d_fwd(const stegan_conv_desc* d, const float* x, const float* w, const float* scale,
                                 const float* bias, const float* residual, float* y, void* stream) {
    ConvK p;
    int rc = fill_common(d, p);
    if (rc) return rc;
    STEGAN_CHECK_ARG(x && w && y);
    STEGAN_CHECK_ARG(d->out_act >= STEGAN_ACT_NONE && d->out_act <= STEGAN_ACT_TANH);
    STEGAN_CHECK_ARG(!(d->out_btc && residual));
    p.x = x; p.w = w; p.scale = scale; p.bias = bias; p.res = residual; p.out = y;
    p.M = p.Cout_g; p.N = d->B * d->Tout; p.R = p.Cin_g * p.K_eff;
    p.fdT = make_fastdiv((uint32_t)d->Tout);
    return launch<MODE_FWD>(p, d->groups, 1, (hipStream_t)stream);
}

extern "C" int stegan_conv1d_bwd_data(const stegan_conv_desc* d, const float* dy, const float* w, const float* scale,
                                      const float* mask_src, const float* add_src, float* dx, void* stream) {
    ConvK p;
    int rc = fill_common(d, p);
    if (rc) return rc;
    STEGAN_CHECK_ARG(dy && w && dx);
    STEGAN_CHECK_ARG(d->in_act != STEGAN_ACT_RELU || mask_src);
    p.dy = dy; p.w = w; p.scale = scale; p.res = add_src; p.out = dx;
    p.mask = d->in_act == STEGAN_ACT_RELU ? mask_src : nullptr;
    p.M = p.Cin_g; p.N = d->B * p.Te; p.R = p.Cout_g * p.K_eff;
    p.fdT = make_fastdiv((uint32_t)p.Te);
    return launch<MODE_BWD_DATA>(p, d->groups, 1, (hipStream_t)stream);
}

static int bwd_weight_impl(const stegan_conv_desc* d, const float* x, const float* dy, float* dw, float* db,
                           float* workspace, int64_t workspace_floats, void* stream) {
    ConvK p;
    int rc = fill_common(d, p);
    if (rc) return rc;
    STEGAN_CHECK_ARG(x && dy && dw);
    rc = stegan_wgrad_small(d, x, dy, dw, db, workspace, workspace_floats, (hipStream_t)stream);
    if (rc != 1) return rc;
    rc = stegan_wgrad_direct(d, x, dy, dw, db, workspace, workspace_floats, (hipStream_t)stream);
    if (rc != 1) return rc;
    p.x = x; p.dy = dy; p.out = dw; p.db = db;
    p.M = p.Cout_g; p.N = p.Cin_g * p.K_eff; p.R = d->B * d->Tout;
    p.fdT = make_fastdiv((uint32_t)d->Tout);
    // slices of the reduction need slabs: without a workspace ONE slice adds straight into dW (slow, still deterministic)
    const long per_slice = p.dw_numel + d->Cout;
    const long max_splits = (workspace && workspace_floats >= 2 * per_slice) ? workspace_floats / per_slice : 1;
    const int splits = bwd_weight_splits(p, d->groups, max_splits);
    if (splits > 1) p.partial = workspace;
    rc = launch<MODE_BWD_WEIGHT>(p, d->groups, splits, (hipStream_t)stream);
    if (rc || splits == 1) return rc;
    long rb = (p.dw_numel + 255) / 256;
    if (rb > 1024) rb = 1024;
    hipLaunchKernelGGL(igemm_wgrad_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, (hipStream_t)stream, p.partial, dw,
                       p.dw_numel, splits, d->K, p.k_lo, p.K_eff, db, d->Cout);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_conv1d_bwd_weight(const stegan_conv_desc* d, const float* x, const float* dy, float* dw,
                                        float* db, void* stream) {
    return bwd_weight_impl(d, x, dy, dw, db, nullptr, 0, stream);
}

extern "C" int stegan_conv1d_bwd_weight_ws(const stegan_conv_desc* d, const float* x, const float* dy, float* dw,
                                           float* db, float* workspace, int64_t workspace_floats, void* stream) {
    return bwd_weight_impl(d, x, dy, dw, db, workspace, workspace_floats, stream);
}

extern "C" int stegan_conv1d_plan(const stegan_conv_desc* d, int32_t mode, int32_t* tile, int32_t* splits,
                                  int64_t* live_macs) {
    ConvK p;
    int rc = fill_common(d, p);
    if (rc) return rc;
    STEGAN_CHECK_ARG(mode >= 0 && mode <= 2 && tile && splits && live_macs);
    int sp = 1;
    if (mode == MODE_FWD) { p.M = p.Cout_g; p.N = d->B * d->Tout; p.R = p.Cin_g * p.K_eff; }
    else if (mode == MODE_BWD_DATA) { p.M = p.Cin_g; p.N = d->B * p.Te; p.R = p.Cout_g * p.K_eff; }
    else { p.M = p.Cout_g; p.N = p.Cin_g * p.K_eff; p.R = d->B * d->Tout; sp = bwd_weight_splits(p, d->groups); }
    *tile = use_big_tile(p, d->groups, sp) ? 128 : 64;
    *splits = sp;
    *live_macs = (int64_t)p.M * p.N * p.R * d->groups;      // MACs the MFMA loop actually issues (dead taps skipped)
    return STEGAN_OK;
}

extern "C" int stegan_bias_grad(const float* dy, float* db, int32_t B, int32_t C, int32_t T, void* stream) {
    STEGAN_CHECK_ARG(dy && db && B > 0 && C > 0 && T > 0);
    launch_bias_grad(dy, db, B, C, T, (hipStream_t)stream);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}
