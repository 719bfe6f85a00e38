// Direct weight-gradient convolution on the fp32 matrix cores.
//
//     dW[co, ci, k] += sum_{b, t}  dy[b, co, t] * xin[b, ci, t*stride + k*dil - pad]       xin = relu?(upsample?(x))
//
// As a GEMM the reduction runs over (b, t) -- the axis along which BOTH operands are contiguous in HBM -- so the tiles
// are staged as plain row segments (coalesced along time, no im2col gather, no per-element index arithmetic):
//
//   * block = 64 output channels (co) x 32 input channels (ci) x KT taps (KT <= 8 per block, grid.y walks tap groups)
//     over a slice of the batch (grid.z); 4 wave64s = 2 (co halves) x 2 (even / odd reduction pairs);
//   * per chunk of <= 32 reduction positions: dy tile [64][33] and the halo'd input rows [32][XP] go global -> registers
//     -> LDS (double buffered, one barrier per chunk, the loads of chunk c+1 are in flight under the MFMAs of chunk c);
//     short sequences (T < 32) pack several batch items into one chunk, each with its own halo'd row segment;
//   * v_mfma_f32_32x32x2_f32: A[m][r] = dy_s[m][r] (pitch 33: conflict free), B[r][ci] = x_s[ci][t(r)*s + k*d] (odd
//     pitch: conflict free) -- one A read feeds KT MFMAs, the tap shift is an address offset into the halo'd row;
//   * the batch slices write slabs of a workspace and a second kernel adds them into dW (and the slices' bias sums into
//     db) in a fixed order; a single slice adds straight into dW with plain read-modify-writes (every element has one
//     writer).  No float atomics: the gradient path is bit-reproducible from launch to launch (round 4).
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int WG_XPMAX = 192;    // longest halo'd x row (floats): three 64-lane passes

struct WgP {
    const float* x;
    const float* dy;
    float* dw;
    float* db;
    float* partial;              // [splits][dW numel] slabs, then [splits][Cout] bias sums (NULL: ONE slice, adds into dW / db)
    long dw_numel;
    int B, Cin, Cout, Cin_g, Cout_g, K, stride, dil, pad, Tin, Te, Tout, ups, relu;
    int k_lo, K_eff;             // live taps [k_lo, k_lo + K_eff)
    int RCH, RV;                 // positions per chunk (even, <= 32) and how many of them are real
    int nct, nbc;                // T >= 32: chunks per item (nbc = 1);  T < 32: items per chunk (nct = 0)
    int XSEG, XP;                // halo'd row segment per item, padded (odd) row pitch
    int total_chunks, chunks_per_block;
    int m_tiles, ci_tiles;
    FastDiv fdXSEG, fdT, fdNCT, fdCIT, fdMT;
};

__device__ __forceinline__ float floor_max(float x, float floor_) {      // one VALU op, no canonicalisation
    return __builtin_amdgcn_fmed3f(x, floor_, INFINITY);
}

template <int KT, int NJ, int XS>       // NJ = reduction pairs per wave and chunk: 8 (RCH <= 32) or 6 (RCH <= 24);
                                       // XS = 64-lane passes over an x row (XP <= 64 XS)
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgP p) {
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int half = lane >> 5, l31 = lane & 31;

    // descriptor words in one clustered batch of kernarg loads; block-id decode by magic division (see conv_direct.hip)
    asm volatile("" ::"s"(p.B), "s"(p.Cin), "s"(p.Cout), "s"(p.Cin_g), "s"(p.Cout_g), "s"(p.K), "s"(p.stride), "s"(p.dil),
                 "s"(p.pad), "s"(p.Tin), "s"(p.Te), "s"(p.Tout), "s"(p.ups), "s"(p.relu), "s"(p.k_lo), "s"(p.K_eff),
                 "s"(p.RCH), "s"(p.RV), "s"(p.nct), "s"(p.nbc), "s"(p.XSEG), "s"(p.XP), "s"(p.total_chunks),
                 "s"(p.chunks_per_block), "s"(p.m_tiles), "s"(p.ci_tiles));
    const int tile = blockIdx.x;
    const int t1 = (int)fd_div((uint32_t)tile, p.fdCIT);
    const int cit = tile - t1 * p.ci_tiles;
    const int g = (int)fd_div((uint32_t)t1, p.fdMT);
    const int mt = t1 - g * p.m_tiles;
    const int m0 = mt * 64, ci0 = cit * 32;
    const int tg = blockIdx.y;
    const int ktaps = min(KT, p.K_eff - tg * KT);
    const int k_first = p.k_lo + tg * KT;
    // this block's slice of the reduction: global chunks [c_lo, c_lo + nchunks) (chunk -> item, time: see WG_LOAD)
    const int c_lo = blockIdx.z * p.chunks_per_block;
    const int nchunks = min(p.total_chunks, c_lo + p.chunks_per_block) - c_lo;
    if (nchunks <= 0) return;
    const int b_end = p.B;

    const int buf_floats = 64 * 33 + 32 * p.XP;

    // ---- chunk-invariant staging maps.  Staging is ROW oriented: a wave-instruction loads one row segment, its base
    // address is wave-uniform (scalar registers, computed on the scalar unit) and the per-lane byte offset + validity
    // are the same for every row of a chunk -- so the vector ALU does ~2 operations per staged element instead of a
    // full address + bounds computation each.
    // dy: wave w owns rows 16 w .. 16 w + 15 as 8 row pairs; lane <-> (row of the pair = lane / 32, position = lane % 32)
    const int hrow = lane >> 5, pos = lane & 31;
    int d_it = 0, d_t = pos;
    if (!p.nct) { d_it = (int)fd_div((uint32_t)pos, p.fdT); d_t = pos - d_it * p.Tout; }
    const bool pos_ok = pos < p.RV;
    const uint32_t dvoff = (uint32_t)(((d_it * p.Cout + hrow) * p.Tout + d_t) * 4);
    const int dy_lds = (wave * 16 + hrow) * 33 + pos;
    // x: wave w owns input channels 8 w .. 8 w + 7; lane <-> slots lane + 64 s of the row (all items of the chunk side by
    // side: slot = item * XSEG + q_rel)
    int xs_it[XS], xs_ql[XS];
#pragma unroll
    for (int s2 = 0; s2 < XS; ++s2) {
        const int slot = lane + 64 * s2;
        xs_it[s2] = 0; xs_ql[s2] = 1 << 28;                      // 1 << 28: never a valid source position
        if (slot < p.nbc * p.XSEG) {
            xs_it[s2] = (int)fd_div((uint32_t)slot, p.fdXSEG);
            xs_ql[s2] = slot - xs_it[s2] * p.XSEG;
        }
    }
    const int x_lds = wave * 8 * p.XP + lane;
    // B-operand offsets of this lane's reduction positions r = 2 (2 j + wn) + half
    int xo[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int r = 2 * (2 * j + wn) + half;
        int it = 0, t = r;
        if (!p.nct) { it = (int)fd_div((uint32_t)r, p.fdT); t = r - it * p.Tout; }
        xo[j] = r < p.RV ? it * p.XSEG + t * p.stride : 0;
    }

    f32x16 acc[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;

    float rdy[8], rx[8][XS];
    const float relu_floor = p.relu ? 0.f : -INFINITY;
    const bool do_bias = p.db != nullptr && cit == 0 && tg == 0 && tid < 64;
    float bsum = 0.f;

    // The loaded words stay untouched in registers (validity is applied when LDS is written), so the loads of chunk c+1
    // are in flight under the MFMAs of chunk c.  Rows past the tile edge are loaded from the clamped last row and
    // zeroed by the (scalar) row mask.
    bool dok = false, xok[XS];
    const char* const dyb = reinterpret_cast<const char*>(p.dy);
    const char* const xb = reinterpret_cast<const char*>(p.x);
    const int co_last = p.Cout_g - 1, ci_last = p.Cin_g - 1;
#define WG_LOAD(c_)                                                                                            \
    {                                                                                                          \
        int item_, t0_;                                                                                        \
        const int cg_ = c_lo + (c_);                                                                           \
        if (p.nct) { item_ = (int)fd_div((uint32_t)cg_, p.fdNCT); t0_ = (cg_ - item_ * p.nct) * p.RV; }        \
        else { item_ = cg_ * p.nbc; t0_ = 0; }                                                                 \
        const int left_ = b_end - item_;                                                                       \
        dok = pos_ok && d_it < left_ && (t0_ + d_t < p.Tout);                                                  \
        const uint32_t dvo_ = dok ? dvoff : 0u;                                                                \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                        \
            const int row_ = min(m0 + wave * 16 + 2 * i, co_last - 1);                                         \
            const char* base_ = dyb + ((size_t)(item_ * p.Cout + g * p.Cout_g + row_) * p.Tout + t0_) * 4;     \
            rdy[i] = *reinterpret_cast<const float*>(base_ + dvo_);                                            \
        }                                                                                                      \
        const int q0_ = t0_ * p.stride - p.pad + k_first * p.dil;                                              \
        uint32_t xvo_[XS];                                                                                     \
        _Pragma("unroll") for (int s2 = 0; s2 < XS; ++s2) {                                                    \
            const int q_ = xs_ql[s2] + q0_;                                                                    \
            xok[s2] = xs_it[s2] < left_ && (uint32_t)q_ < (uint32_t)p.Te;                                      \
            xvo_[s2] = xok[s2] ? (uint32_t)((xs_it[s2] * p.Cin * p.Tin + (q_ >> p.ups)) * 4) : 0u;             \
        }                                                                                                      \
        _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                                        \
            const int ci_ = min(ci0 + wave * 8 + r, ci_last);                                                  \
            const char* base_ = xb + (size_t)(item_ * p.Cin + g * p.Cin_g + ci_) * p.Tin * 4;                  \
            _Pragma("unroll") for (int s2 = 0; s2 < XS; ++s2)                                                  \
                rx[r][s2] = *reinterpret_cast<const float*>(base_ + xvo_[s2]);                                 \
        }                                                                                                      \
    }
#define WG_STORE(buf_)                                                                                         \
    {                                                                                                          \
        float* D_ = smem + (buf_) * buf_floats;                                                                \
        float* X_ = D_ + 64 * 33;                                                                              \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                        \
            const bool rok_ = m0 + wave * 16 + 2 * i + hrow < p.Cout_g;                                        \
            D_[dy_lds + i * 66] = (dok && rok_) ? rdy[i] : 0.f;                                                \
        }                                                                                                      \
        _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                                        \
            const bool rok_ = ci0 + wave * 8 + r < p.Cin_g;                                                    \
            _Pragma("unroll") for (int s2 = 0; s2 < XS; ++s2)                                                  \
                if (lane + 64 * s2 < p.XP) X_[x_lds + r * p.XP + 64 * s2] = (xok[s2] && rok_) ? rx[r][s2] : 0.f; \
        }                                                                                                      \
    }

    WG_LOAD(0)
    WG_STORE(0)
    block_sync();
    for (int c = 0; c < nchunks; ++c) {
        const int cur = c & 1;
#ifndef WG_EXP_NOSTAGE
        if (c + 1 < nchunks) WG_LOAD(c + 1)
#endif
        const float* Ds = smem + cur * buf_floats;
        const float* Xs = Ds + 64 * 33;
        if (do_bias) {
            float s = 0.f;
            for (int r = 0; r < p.RCH; ++r) s += Ds[tid * 33 + r];
            bsum += s;
        }
        // positions past RV carry zero dy, so every wave runs all NJ pairs (no per-pair branches); operands of pair
        // j+1 are read while the KT MFMAs of pair j run
        const float* arow = Ds + (wm * 32 + l31) * 33 + half + 2 * wn;
        const float* xrow = Xs + l31 * p.XP;
        float a0, a1, b0[KT], b1[KT];
#define WG_OPS(a_, b_, j_)                                                                                     \
        {                                                                                                      \
            a_ = arow[4 * (j_)];                                                                               \
            _Pragma("unroll") for (int k = 0; k < KT; ++k) b_[k] = xrow[xo[j_] + k * p.dil];                   \
        }
#ifdef WG_EXP_NORELU
#define WG_BVAL(x_) (x_)
#else
#define WG_BVAL(x_) floor_max(x_, relu_floor)
#endif
#define WG_MFMA(a_, b_)                                                                                        \
        _Pragma("unroll") for (int k = 0; k < KT; ++k)                                                         \
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, WG_BVAL(b_[k]), acc[k], 0, 0, 0);
#ifndef WG_EXP_NOMFMA
        WG_OPS(a0, b0, 0)
#pragma unroll
        for (int j = 0; j < NJ; j += 2) {
            WG_OPS(a1, b1, j + 1)
            __builtin_amdgcn_sched_barrier(0);             // reads of pair j+1 go out BEFORE the MFMAs of pair j
            WG_MFMA(a0, b0)
            __builtin_amdgcn_sched_barrier(0);
            if (j + 2 < NJ) WG_OPS(a0, b0, j + 2)
            __builtin_amdgcn_sched_barrier(0);
            WG_MFMA(a1, b1)
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
#undef WG_OPS
#undef WG_MFMA
#ifndef WG_EXP_NOSTAGE
        if (c + 1 < nchunks) WG_STORE(cur ^ 1)
#endif
        block_sync();
    }
#undef WG_LOAD
#undef WG_STORE

    if (do_bias && m0 + tid < p.Cout_g) {            // one block per (channel tile, slice) holds the bias sum: single writer
        const int ch = g * p.Cout_g + m0 + tid;
        if (p.partial) p.partial[(size_t)gridDim.z * p.dw_numel + (size_t)blockIdx.z * p.Cout + ch] = bsum;
        else p.db[ch] += bsum;
    }
    // ---- epilogue.  The block first folds its two reduction-parity halves together in LDS and then writes its
    // 64 x (32 ci x ktaps) tile with lanes running along the (ci, k) axis of dW -- contiguous floats -- instead of one
    // strided element per lane straight out of the MFMA registers.  C/D map of the 32x32 MFMA: col = lane & 31 (ci), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (co).
    constexpr int OP = 32 * KT + 1;
    float* out_s = smem;                                  // [64][OP], column = kt * 32 + ci  (staging buffers are dead)
    if (wn == 1) {
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                out_s[(wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * half) * OP + k * 32 + l31] = acc[k][e];
    }
    block_sync();
    if (wn == 0) {
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float* o = out_s + (wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * half) * OP + k * 32 + l31;
                *o += acc[k][e];
            }
    }
    block_sync();
    const int run = 32 * ktaps;                           // floats per output row of this block
    const uint32_t inv = (65536u + ktaps - 1) / ktaps;    // j / ktaps == (j * inv) >> 16 for j < 256, ktaps <= 8
    float* dst = p.partial ? p.partial + (size_t)blockIdx.z * p.dw_numel : p.dw;
    for (int row = wave; row < 64; row += 4) {
        const int co = m0 + row;
        if (co >= p.Cout_g) break;
        float* drow = dst + ((long)(g * p.Cout_g + co) * p.Cin_g + ci0) * p.K + k_first;
        for (int j = lane; j < run; j += 64) {
            const int cil = (int)(((uint32_t)j * inv) >> 16), kt = j - cil * ktaps;
            if (ci0 + cil < p.Cin_g) {
                const float v = out_s[row * OP + kt * 32 + cil];
                if (p.partial) drow[cil * p.K + kt] = v;
                else drow[cil * p.K + kt] += v;           // single slice: this block is the element's only writer
            }
        }
    }
}

// dW += sum over the batch slices' slabs (fixed order: deterministic); dead taps were never written and are skipped
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                long numel, int splits, int K, int k_lo, int K_eff,
                                                                float* __restrict__ db, int Cout) {
    if (db) {                                             // the slices' bias sums: [splits][Cout] behind the slabs
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
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int s = 0;
        for (; s + 8 <= splits; s += 8) {                 // eight independent loads in flight, fixed summation order
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] += src[(size_t)(s + u) * numel];
        }
        for (; s < splits; ++s) v[0] += src[(size_t)s * numel];
        dw[i] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
}

int wg_live_taps(const stegan_conv_desc* d, int& k_lo) {
    const int Te = d->Tin * d->upsample;
    int lo = d->K, hi = -1;
    for (int k = 0; k < d->K; ++k) {
        const int a = k * d->dilation - d->pad, b = (d->Tout - 1) * d->stride + a;
        if (b >= 0 && a < Te) { if (k < lo) lo = k; hi = k; }
    }
    k_lo = lo;
    return hi - lo + 1;
}

}  // namespace

// returns 1 when the shape is better served by (or only fits) the generic implicit-GEMM kernel
int stegan_wgrad_direct(const stegan_conv_desc* d, const float* x, const float* dy, float* dw, float* db,
                        float* workspace, int64_t workspace_floats, hipStream_t stream) {
    static const char* off = getenv("STEGAN_NO_WGRAD_DIRECT");
    if (off) return 1;
    WgP p;
    p.x = x; p.dy = dy; p.dw = dw; p.db = db;
    p.dw_numel = (long)d->Cout * (d->Cin / d->groups) * d->K;
    p.B = d->B; p.Cin = d->Cin; p.Cout = d->Cout; p.Cin_g = d->Cin / d->groups; p.Cout_g = d->Cout / d->groups;
    p.K = d->K; p.stride = d->stride; p.dil = d->dilation; p.pad = d->pad; p.Tin = d->Tin; p.Tout = d->Tout;
    p.Te = d->Tin * d->upsample; p.ups = d->upsample == 2 ? 1 : 0; p.relu = d->in_act == STEGAN_ACT_RELU;
    // mostly-empty MFMA tiles (narrow groups) and single-tap convs (one MFMA per operand pair read): the generic kernel,
    // which packs (ci, k) densely along N, measured faster there (scripts/conv_microbench.py)
    {
        static const char* e1 = getenv("STEGAN_WG_MIN_CIN");
        static const char* e2 = getenv("STEGAN_WG_MIN_COUT");
        const int min_cin = e1 ? atoi(e1) : 64, min_cout = e2 ? atoi(e2) : 64;
        if (p.Cout_g < min_cout || p.Cin_g < min_cin) return 1;
    }
    p.K_eff = wg_live_taps(d, p.k_lo);
    if (p.K_eff <= 1) return 1;
    const int KT = p.K_eff <= 1 ? 1 : p.K_eff <= 3 ? 3 : p.K_eff <= 5 ? 5 : 8;
    const int T = d->Tout;
    int tseg;
    if (T >= 32) {
        p.nct = (T + 31) / 32; p.nbc = 1;
        p.RV = (T + p.nct - 1) / p.nct;
        tseg = p.RV;
    } else {
        p.nct = 0; p.nbc = 32 / T;
        p.RV = p.nbc * T;
        tseg = T;
    }
    p.RCH = (p.RV + 1) & ~1;
    p.XSEG = (tseg - 1) * d->stride + (KT - 1) * d->dilation + 1;
    p.XP = (p.nbc * p.XSEG) | 1;
    if (p.XP > WG_XPMAX || p.Cout_g % 2 != 0) return 1;
    p.fdXSEG = make_fastdiv((uint32_t)p.XSEG); p.fdT = make_fastdiv((uint32_t)T);
    p.fdNCT = make_fastdiv((uint32_t)(p.nct ? p.nct : 1));
    p.m_tiles = ceil_div(p.Cout_g, 64); p.ci_tiles = ceil_div(p.Cin_g, 32);
    p.fdCIT = make_fastdiv((uint32_t)p.ci_tiles); p.fdMT = make_fastdiv((uint32_t)p.m_tiles);
    const int tapgroups = ceil_div(p.K_eff, KT);
    const long tiles = (long)d->groups * p.m_tiles * p.ci_tiles * tapgroups;
    // Slices of the reduction (whole chunks, not whole items, so any slice count works): aim at two co-resident blocks
    // per CU -- a multiple of the CU count keeps the CUs evenly loaded -- with at least 3 chunks per block.  The slices
    // write slabs of `workspace` (plain stores + one deterministic reduction pass); with a short workspace use fewer,
    // longer slices; without one run ONE slice (slow, still deterministic).
    p.total_chunks = p.nct ? d->B * p.nct : ceil_div(d->B, p.nbc);
    long max_splits = 1;
    p.partial = nullptr;
    if (workspace && workspace_floats >= 2 * (p.dw_numel + d->Cout)) {
        max_splits = workspace_floats / (p.dw_numel + d->Cout);
        p.partial = workspace;
    }
    static const char* env_cap = getenv("STEGAN_WG_MAX_SPLITS");         // experiments
    const long cap_splits = env_cap ? atol(env_cap) : 64;
    if (max_splits > cap_splits) max_splits = cap_splits;
    // cost model in units of one chunk of one block: the blocks on a CU share its matrix pipes, so a launch takes about
    // (blocks on the busiest CU) x (chunks per block + ~3 for prologue / first loads / epilogue), stretched when only
    // one or two blocks share a CU; every extra slice adds a slab of dw_numel floats to the reduction pass
    long want = 1;
    double best = 1e30;
    for (long sp = 1; sp <= max_splits && sp <= p.total_chunks; ++sp) {
        const int cps = ceil_div(p.total_chunks, (int)sp);
        const long blocks = tiles * ceil_div(p.total_chunks, cps);
        const long load = (blocks + 255) / 256;
        const double eff = load >= 3 ? 1.0 : load == 2 ? 0.8 : 0.55;      // few resident waves hide little latency
        static const char* env_sc = getenv("STEGAN_WG_SLAB_COST");       // experiments: weight of the slab traffic
        const double cost = (double)load * (cps + 3.0) / eff + (sp > 1 ? sp * (double)p.dw_numel * 1.3e-6 * (env_sc ? atof(env_sc) : 1.0) : 0.0);
        if (cost < best - 1e-9) { best = cost; want = sp; }
    }
    {
        static const char* env = getenv("STEGAN_WG_SPLITS");        // experiment: force the slice count
        if (env && p.partial) { want = atol(env); if (want > max_splits) want = max_splits; if (want < 1) want = 1; }
    }
    p.chunks_per_block = ceil_div(p.total_chunks, (int)want);
    const int splits = ceil_div(p.total_chunks, p.chunks_per_block);
    if (splits == 1) p.partial = nullptr;                       // a single slice adds straight into dW
    if (tiles > 65535L * 32768 || splits > 65535) return 1;
    size_t lds = 2 * ((size_t)64 * 33 + (size_t)32 * p.XP) * sizeof(float);
    const size_t lds_out = (size_t)64 * (32 * KT + 1) * sizeof(float);            // epilogue transpose tile
    if (lds_out > lds) lds = lds_out;
    dim3 grid((unsigned)(d->groups * p.m_tiles * p.ci_tiles), (unsigned)tapgroups, (unsigned)splits);
#define WG_LAUNCH3(KT_, NJ_, XS_)                                                                                \
    {                                                                                                            \
        static bool attr_ = false;                                                                               \
        if (lds > 64 * 1024 && !attr_) {                                                                         \
            (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<KT_, NJ_, XS_>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); \
            attr_ = true;                                                                                        \
        }                                                                                                        \
        hipLaunchKernelGGL((conv_wgrad_kernel<KT_, NJ_, XS_>), grid, dim3(256), lds, stream, p);                 \
    }
#define WG_LAUNCH2(KT_, NJ_) { if (p.XP <= 128) WG_LAUNCH3(KT_, NJ_, 2) else WG_LAUNCH3(KT_, NJ_, 3) }
#define WG_LAUNCH(KT_) { if (p.RCH <= 24) WG_LAUNCH2(KT_, 6) else WG_LAUNCH2(KT_, 8) }
    if (KT == 1) WG_LAUNCH(1) else if (KT == 3) WG_LAUNCH(3) else if (KT == 5) WG_LAUNCH(5) else WG_LAUNCH(8)
#undef WG_LAUNCH3
#undef WG_LAUNCH2
    if (p.partial) {
        long rb = (p.dw_numel + 255) / 256;
        if (rb > 2048) rb = 2048;
        hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, stream, p.partial, dw, p.dw_numel,
                           splits, d->K, p.k_lo, p.K_eff, db, d->Cout);
    }
#undef WG_LAUNCH
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}
