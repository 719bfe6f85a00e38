// Direct (im2col-free) 1-D convolution on the fp32 matrix cores, v2: halo'd input rows in LDS.
//
// Forward and input-gradient convolutions are the same computation
//
//     out[b, m, to(u)] = sum_{c2, tap}  W[tap][c2][m] * src[b, c2, u*cs + off[tap]]        (zero outside [0, Tq))
//
//   FWD        m = co, c2 = ci, src = relu?(upsample?(x)), cs = stride, off[k] = k*dil - pad, to(u) = u
//   BWD_DATA   m = ci, c2 = co, src = scale[co]*dy; for each output phase phi = te mod stride only the taps with
//              (phi + pad - k*dil) % stride == 0 contribute, off[k] = (phi+pad-k*dil)/stride, cs = 1,
//              to(u) = phi + u*stride  -> a strided conv's input gradient costs no wasted MACs (grid.z = phase)
//
// Per block (256 threads = 2x2 wave64, tile 64 or 128 (m) x 64 or 128 (n = flattened (b,u)); narrow groups: 1x4 waves,
// tile 32 x 128):
//   * for a chunk of CC source channels, every (channel, batch-segment) input row the tile needs is copied ONCE, with
//     its halo, into LDS by LDS-DMA (global_load_lds, 16 B per lane where rows are float4 aligned): zero padding, dead
//     rows and the nearest-x2 upsample are resolved by the per-lane source address (a zero row sits behind the
//     transposed weights), the input ReLU is one v_max at the operand read, the per-channel dy scale is folded into the
//     transposed weights -- each sample is read from HBM/L2 once per chunk instead of K times and the per-element index
//     arithmetic of an im2col gather disappears (the lane -> source offsets are chunk invariant and live in registers);
//   * the weight tile comes from a K-major, output-channel-contiguous copy of the direction tensor
//     (stegan_conv_prep: Wf[g][k][ci][co], Wb[g][k][co][ci], leading dimension padded to 64) so staging is a
//     pure 16-byte-per-lane copy into an unpadded [row][64] LDS image;
//   * v_mfma_f32_32x32x2_f32 operands: A = Ws[row][m] (32 consecutive dwords per half-wave), B = Xs[channel * XROW +
//     tap * step + colbase(lane)] -- the live taps form an arithmetic progression, so the tap shift is a computed
//     address offset into the halo'd row (no tables);
//   * two LDS buffers, one barrier per chunk; many-tap layers use tap groups (TG taps x 64/TG channels per W chunk, the X
//     tile staged once per channel chunk); small grids split the reduction (uniform split-K, or only the tail tiles
//     beyond a multiple of the CU count) with a fixed-order epilogue kernel; up to five independent problems can share
//     one launch (MultiP).  DESIGN.md section 4.1 has the measurements behind each of these.
#include <stdio.h>
#include <stdlib.h>

#include "conv_direct_common.h"

namespace {

// TAPG: tap groups (p.NTG > 1) -- compiled separately so the common all-taps-in-one-chunk kernels keep their straight-line
// DMA issue (a data-dependent branch around the X-tile DMA costs every chunk of every layer otherwise).
template <int TM, int TN, bool FAST, bool XVEC, bool W14 = false, bool TAPG = false>
__device__ __forceinline__ void conv_direct_body(const DirP& p, const unsigned bid_x, const unsigned bid_y, const unsigned bid_z) {
    constexpr int BN = W14 ? 128 : 64 * TN;
    constexpr int BM = 64 * TM;                 // 2x2 waves, wave tile (32*TM) x (32*TN)
    constexpr int WEt = WE * TM;                // W-tile float4 per thread and chunk
    constexpr int WROW4 = 16 * TM;              // float4 per W row
    extern __shared__ __align__(16) float smem[];      // 2 x { Ws[WR*BM] | Xs[CC*XROW] }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef STEGAN_STAMPS
#define STAMP(i_) if (p.stamps && tid == 0) p.stamps[((size_t)(bid_z * gridDim.y + bid_y) * gridDim.x + bid_x) * 8 + (i_)] = __builtin_amdgcn_s_memtime();
#else
#define STAMP(i_)
#endif
    STAMP(0)
    // Pull the descriptor words the prologue needs into scalar registers in ONE clustered batch of kernarg loads: left
    // alone, hipcc sinks each s_load next to its first use -- ~20 dependent load + wait round trips (100-200 cycles each)
    // strung through the index setup in front of the first DMA.
    asm volatile("" ::"s"(p.B), "s"(p.Csrc), "s"(p.Cdst), "s"(p.Tsrc), "s"(p.Tq), "s"(p.Tdst), "s"(p.C2_g), "s"(p.M_g),
                 "s"(p.LD), "s"(p.K), "s"(p.stride), "s"(p.mode), "s"(p.m_tiles), "s"(p.CC), "s"(p.logCC), "s"(p.SEG),
                 "s"(p.NSEG), "s"(p.XROW), "s"(p.WR), "s"(p.U0), "s"(p.splits), "s"(p.chunks_per_split), "s"(p.mix_S),
                 "s"(p.mix_full), "s"(p.nbx), "s"(p.ny), "s"(p.ups), "s"(p.in_relu));
    const int wm = W14 ? 0 : wave >> 1, wn = W14 ? wave : wave & 1;
    int bx = (int)bid_x, by = (int)bid_y, phase, split, nsplits = p.splits, cps = p.chunks_per_split;
    if (p.mix_S) {
        int tile = bx;
        split = 0; nsplits = 1;
        if (bx >= p.mix_full) {
            const int r = bx - p.mix_full;
            const int rq = (int)fd_div((uint32_t)r, p.fdMS);
            tile = p.mix_full + rq;
            split = r - rq * p.mix_S;
            nsplits = p.mix_S; cps = p.mix_cps;
        }
        const int rest = (int)fd_div((uint32_t)tile, p.fdNBX);
        bx = tile - rest * p.nbx;
        phase = (int)fd_div((uint32_t)rest, p.fdNY);
        by = rest - phase * p.ny;
    } else {
        phase = (int)fd_div(bid_z, p.fdSP);
        split = (int)bid_z - phase * p.splits;
    }
    const int g = (int)fd_div((uint32_t)by, p.fdMT);
    const int m0 = (by - g * p.m_tiles) * BM;

    // ---- per-phase geometry (wave-uniform scalar work; the per-phase values come precomputed from the host)
    int U, cs, os;
    if (p.mode == 0) { U = p.U0; cs = p.stride; os = 1; }
    else { U = p.ph_U[phase]; cs = 1; os = p.stride; }
    const int N = p.B * U;
    const int n0 = bx * BN;
    if (n0 >= N) return;
    const FastDiv fdU = (U == p.U0) ? p.fdU0 : p.fdU1;
    const Taps tp = phase_taps(p, phase);
#ifdef STEGAN_STAMPS_PROLOGUE
    STAMP(3)
#endif
    const int ntaps = tp.ntaps, off_min = tp.off_min;
    const int CC = p.CC;
    // rows per W chunk: (taps of the chunk) x CC.  With tap groups every chunk has TG * CC rows; the rows of taps past the
    // phase's last live tap are zero weight rows (their B operands read real halo data, so the products are exact zeros)
    const int RCV = (TAPG ? p.TG : ntaps) * CC;
    const int RC = (RCV + 7) & ~7;                  // padded to the MFMA loop's unroll (zero weight rows)

    // ---- chunk-invariant part of the X staging.  Scalar form: thread element e <-> tile index tid + 256 e =
    // ci * XROW + slot.  Vector form (XVEC: row length and valid extent are multiples of 4 floats, no upsample): every
    // segment starts at the source position rounded DOWN to a multiple of 4, so one 16-byte DMA moves 4 slots that
    // are all valid or all zero; the rounding (0..3 slots) is added to the lane's column base instead.
    const int b_lo = (int)fd_div((uint32_t)n0, fdU), u0 = n0 - b_lo * U;
    const int qbase0 = u0 * cs + off_min, qbaseN = off_min;                 // first / later segments
    const int qal0 = XVEC ? (qbase0 & ~3) : qbase0, qalN = XVEC ? (qbaseN & ~3) : qbaseN;
    constexpr int XN = XVEC ? XE / 4 : XE;
    int x_src[XN];      // source offset (without the chunk's channel base); x_ci = big marks zero fill
    int x_ci[XN];       // channel within the chunk (1<<20 = never valid)
#pragma unroll
    for (int e = 0; e < XN; ++e) {
        const int idx = (tid + 256 * e) * (XVEC ? 4 : 1);
        x_src[e] = 0;
        x_ci[e] = 1 << 20;
        if (idx < CC * p.XROW) {
            const int ci = (int)fd_div((uint32_t)idx, p.fdXROW), slot = idx - ci * p.XROW;
            const int sg = (int)fd_div((uint32_t)slot, p.fdSEG), sl = slot - sg * p.SEG;
            const int b = b_lo + sg;
            const int q = (sg == 0 ? qal0 : qalN) + sl;
            if (sg < p.NSEG && b < p.B && q >= 0 && q < p.Tq) {
                x_ci[e] = ci;
                x_src[e] = (b * p.Csrc + ci) * p.Tsrc + (p.mode == 0 ? (q >> p.ups) : q);
            }
        }
    }
#ifdef STEGAN_STAMPS_PROLOGUE
    STAMP(4)
#endif
    // ---- W staging: thread element i <-> float4 index tid + 256 i = row * 16 + c4
    int w_off[WEt];     // offset of the float4 inside the group's transposed weights, without the chunk term
    int w_ci[WEt];      // channel within the chunk + 2^16 * tap within the chunk (1 << 30 = never valid)
#pragma unroll
    for (int i = 0; i < WEt; ++i) {
        const int idx = tid + 256 * i;
        const int r = idx / WROW4, c4 = idx % WROW4;
        const int tap = r >> p.logCC, ci = r & (CC - 1);
        w_ci[i] = 1 << 30;
        w_off[i] = 0;
        if (r < RCV) {
            w_ci[i] = TAPG ? (ci | (tap << 16)) : ci;
            w_off[i] = ((tp.kbase + tap * tp.kstep) * p.C2_g + ci) * p.LD + m0 + c4 * 4;
        }
    }
#ifdef STEGAN_STAMPS_PROLOGUE
    STAMP(5)
#endif
    // ---- per-lane column bases for the B operand
    int colbase[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int jn = (wn * TN + j) * 32 + (lane & 31);
        int cb = 0;
        if (n0 + jn < N) {
            const int uu = u0 + jn;
            const int sg = (int)fd_div((uint32_t)uu, fdU), u = uu - sg * U;
            cb = sg * p.SEG + (u - (sg == 0 ? u0 : 0)) * cs + (sg == 0 ? qbase0 - qal0 : qbaseN - qalN);
        }
        colbase[j] = cb;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#ifdef STEGAN_STAMPS_PROLOGUE
    STAMP(6)
#endif
    const int half = lane >> 5, l31 = lane & 31;
    const int nchunks = ntaps > 0 ? ((p.C2_g + CC - 1) >> p.logCC) * (TAPG ? p.NTG : 1) : 0;   // c = (channel chunk, tap group)
    const int c_begin = split * cps;
    const int c_end = min(nchunks, c_begin + cps);
    const float* wt_g = p.wt + (size_t)g * p.K * p.C2_g * p.LD;

    // Staging is LDS-DMA (global_load_lds): every lane hands the DMA engine its own global address and the data lands
    // at wave-uniform LDS base + lane * size -- no staging registers, no ds_write, no selects.  Elements that must be
    // zero (halo, out-of-range rows, channels past C2_g) read a zero word / a zero row kept behind the transposed
    // weights (stegan_conv_prep leaves >= 64 zero floats of slack), so the copy stays branch-free.
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    const float* zero_row = p.wt + p.zero_off;
    // LDS: two W buffers (one per chunk, alternating) followed by two X buffers (one per CHANNEL chunk, alternating)
    const int wbuf_floats = p.WR * BM, xbuf_floats = CC * p.XROW;
    float* const Xbufs = smem + 2 * wbuf_floats;
    const int cc_begin = TAPG ? (int)fd_div((uint32_t)c_begin, p.fdNTG) : c_begin;
#define STEGAN_DMA_CHUNK(c_, buf_)                                                                             \
    {                                                                                                          \
        const int cc_ = TAPG ? (int)fd_div((uint32_t)(c_), p.fdNTG) : (c_), tg_ = TAPG ? (c_) - cc_ * p.NTG : 0; \
        const int c0_ = cc_ * CC;                                                                              \
        const int lim_ = p.C2_g - c0_;                          /* live channels of this chunk        */        \
        const int tlim_ = (ntaps - tg_ * p.TG) << 16;           /* live taps of this chunk, << 16     */        \
        float* Wd_ = smem + (buf_) * wbuf_floats;                                                              \
        const float* wc_ = wt_g + ((long)c0_ + (long)tg_ * p.TG * tp.kstep * p.C2_g) * p.LD;                   \
        _Pragma("unroll") for (int i = 0; i < WEt; ++i) {                                                      \
            const int idx = tid + 256 * i;                                                                     \
            if ((idx / WROW4) < RC) {                                                                          \
                const bool ok_ = TAPG ? ((w_ci[i] & 0xffff) < lim_ && (w_ci[i] & ~0xffff) < tlim_) : (w_ci[i] < lim_); \
                const float* src_ = ok_ ? wc_ + w_off[i] : zero_row + (idx % WROW4) * 4;                        \
                __builtin_amdgcn_global_load_lds((gptr_t)src_, (lds_ptr_t)(Wd_ + (wave * 64 + 256 * i) * 4), 16, 0, 0); \
            }                                                                                                  \
        }                                                                                                      \
        if (!TAPG || tg_ == 0 || (c_) == c_begin) {             /* first chunk of a channel chunk: its X tile */ \
            float* Xd_ = Xbufs + ((cc_ - cc_begin) & 1) * xbuf_floats;                                         \
            const float* xc_ = p.src + (size_t)(g * p.C2_g + c0_) * p.Tsrc;                                    \
            _Pragma("unroll") for (int e = 0; e < XN; ++e) {                                                   \
                const int idx = (tid + 256 * e) * (XVEC ? 4 : 1);                                              \
                if (idx < CC * p.XROW) {                                                                       \
                    const float* src_ = (x_ci[e] < lim_) ? xc_ + x_src[e] : zero_row;                          \
                    if (XVEC) __builtin_amdgcn_global_load_lds((gptr_t)src_, (lds_ptr_t)(Xd_ + (wave * 64 + 256 * e) * 4), 16, 0, 0); \
                    else __builtin_amdgcn_global_load_lds((gptr_t)src_, (lds_ptr_t)(Xd_ + wave * 64 + 256 * e), 4, 0, 0); \
                }                                                                                              \
            }                                                                                                  \
        }                                                                                                      \
    }
    const float relu_floor = p.in_relu ? 0.f : -INFINITY;     // B operand = max(x, floor): the input ReLU costs one v_max

#define STEGAN_LOAD_BODY(a_, b_, r2_)                                                                          \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                            \
        const int r = (r2_) + 2 * k + half;                                                                    \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) a_[k][i] = Wcur[r * BM + (wm * TM + i) * 32 + l31];     \
        const int tap = r >> p.logCC, ci = r & (CC - 1);                                                       \
        const int xo = r < RCV ? ci * p.XROW + tap * tp.step : 0;                                              \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) b_[k][j] = Xcur[xo + colbase[j]];                       \
    }
#define STEGAN_MFMA_BODY(a_, b_)                                                                               \
    _Pragma("unroll") for (int k = 0; k < 4; ++k)                                                              \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                         \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                     \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[k][i], relu_max(b_[k][j], relu_floor),     \
                                                                 acc[i][j], 0, 0, 0);

    // Two LDS buffers, ONE barrier per chunk: the DMA of chunk c+1 is issued right behind the barrier and stays in
    // flight while the matrix cores run over chunk c (LDS reads of the next 4-MFMA body interleaved with the MFMAs).
    STAMP(1)
    if (c_begin < c_end) STEGAN_DMA_CHUNK(c_begin, 0)
    STAMP(2)
    for (int c = c_begin; c < c_end; ++c) {
        const int cur = (c - c_begin) & 1;
#ifndef STEGAN_STAMPS_PROLOGUE
        if (c == c_begin + 3) { STAMP(3) }
#endif
        block_sync();        // hipcc drains vmcnt(0) here: this wave's DMA of chunk c has landed, and after the
                                // barrier so has everyone's; nobody still reads buffer cur ^ 1
#ifndef STEGAN_STAMPS_PROLOGUE
        if (c == c_begin + 3) { STAMP(4) }
#endif
#ifdef STEGAN_EXP_NODMA
        if (false) {
#else
        if (c + 1 < c_end) {
#endif
            // the short DMA-issue phase runs at raised priority so it is not starved by the co-resident waves' MFMA
            // streams (they keep the matrix pipe busy meanwhile)
            __builtin_amdgcn_s_setprio(2);
            STEGAN_DMA_CHUNK(c + 1, cur ^ 1)
            __builtin_amdgcn_s_setprio(0);
        }
#ifndef STEGAN_STAMPS_PROLOGUE
        if (c == c_begin + 3) { STAMP(6) }
#endif
        const int cc_cur = TAPG ? (int)fd_div((uint32_t)c, p.fdNTG) : c, tg_cur = TAPG ? c - cc_cur * p.NTG : 0;
        const float* Wcur = smem + cur * wbuf_floats;
        // the chunk's taps start tg * TG taps into the halo'd rows
        const float* Xcur = Xbufs + ((cc_cur - cc_begin) & 1) * xbuf_floats + tg_cur * p.TG * tp.step;
        if constexpr (FAST) {
            // Fast path (host guarantees CC >= 8) (K_eff*CC is a multiple of 8, a body of 8 rows never straddles a tap): the B address is
            // colb2 + scalar, the A address abase + immediate.  Loads of the NEXT body are interleaved one row pair at
            // a time with the MFMAs of the current body, so LDS latency and address math hide under the matrix pipe.
            const float* abase = Wcur + half * BM + wm * TM * 32 + l31;
            int colb2[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) colb2[j] = colbase[j] + half * p.XROW;
            // Operands are kept RAW in registers and the ReLU floor is applied at the point of use, one body (4 MFMAs,
            // >= 256 matrix-pipe cycles) after the ds_reads were issued: the wait for them is then free.  (Applying it
            // right behind the loads made hipcc drain lgkmcnt(0) in front of every MFMA -- the LDS latency of each
            // step was exposed.)  The loop is peeled so its steady state carries no branches.
            float a0[4][TM], b0[4][TN], a1[4][TM], b1[4][TN];
#define STEGAN_FAST_LOAD(a_, b_, r2_, k_)                                                                      \
            {                                                                                                  \
                const int xb_ = ((r2_) & (CC - 1)) * p.XROW + ((r2_) >> p.logCC) * tp.step + 2 * (k_) * p.XROW; \
                _Pragma("unroll") for (int i = 0; i < TM; ++i) a_[k_][i] = abase[((r2_) + 2 * (k_)) * BM + i * 32]; \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) b_[k_][j] = Xcur[xb_ + colb2[j]];               \
            }
#define STEGAN_FAST_MFMA(a_, b_, k_)                                                                           \
            {                                                                                                  \
                float bb_[TN];                                                                                 \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) bb_[j] = relu_max(b_[k_][j], relu_floor);       \
                _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                 \
                    _Pragma("unroll") for (int j = 0; j < TN; ++j)                                             \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[k_][i], bb_[j], acc[i][j], 0, 0, 0); \
            }
#define STEGAN_FAST_STEP(an_, bn_, rn_, ac_, bc_)                                                              \
            _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                    \
                STEGAN_FAST_LOAD(an_, bn_, rn_, k)                                                             \
                STEGAN_FAST_MFMA(ac_, bc_, k)                                                                  \
                __builtin_amdgcn_sched_barrier(0);                                                             \
            }
#define STEGAN_FAST_LAST(ac_, bc_)                                                                             \
            _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                    \
                STEGAN_FAST_MFMA(ac_, bc_, k)                                                                  \
                __builtin_amdgcn_sched_barrier(0);                                                             \
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) STEGAN_FAST_LOAD(a0, b0, 0, k)
            int r2 = 0;
            for (; r2 + 16 < RC; r2 += 16) {
                STEGAN_FAST_STEP(a1, b1, r2 + 8, a0, b0)
                STEGAN_FAST_STEP(a0, b0, r2 + 16, a1, b1)
            }
            if (r2 + 8 < RC) {
                STEGAN_FAST_STEP(a1, b1, r2 + 8, a0, b0)
                STEGAN_FAST_LAST(a1, b1)
            } else {
                STEGAN_FAST_LAST(a0, b0)
            }
#undef STEGAN_FAST_STEP
#undef STEGAN_FAST_LAST
#undef STEGAN_FAST_LOAD
#undef STEGAN_FAST_MFMA
        } else {
            float a0[4][TM], b0[4][TN], a1[4][TM], b1[4][TN];
            STEGAN_LOAD_BODY(a0, b0, 0)
            for (int r2 = 0; r2 < RC; r2 += 16) {
                if (r2 + 8 < RC) STEGAN_LOAD_BODY(a1, b1, r2 + 8)
                STEGAN_MFMA_BODY(a0, b0)
                if (r2 + 8 < RC) {
                    if (r2 + 16 < RC) STEGAN_LOAD_BODY(a0, b0, r2 + 16)
                    STEGAN_MFMA_BODY(a1, b1)
                }
            }
        }
#ifndef STEGAN_STAMPS_PROLOGUE
        if (c == c_begin + 3) { STAMP(7) }
#endif
    }

#undef STEGAN_LOAD_BODY
#undef STEGAN_MFMA_BODY
#undef STEGAN_DMA_CHUNK

    // ---- epilogue.  Every global read (scale / bias / residual / mask) is issued unconditionally on a clamped
    // index and in batches of 16 before it is used, so the loads overlap instead of serialising on per-element
    // branches; only the stores are predicated.
    const size_t out_elems = (size_t)p.B * p.Cdst * p.Tdst;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int jn = (wn * TN + j) * 32 + l31;
        const bool n_ok = n0 + jn < N;
        const int n = n_ok ? n0 + jn : N - 1;
        const int b = (int)fd_div((uint32_t)n, fdU), u = n - b * U;
        const int to = p.mode == 0 ? u : phase + u * os;
#pragma unroll
     for (int im = 0; im < TM; ++im) {
        const int mrow0 = m0 + (wm * TM + im) * 32 + 4 * half;
        if (nsplits > 1) {
            float* dst = p.partial + (size_t)split * out_elems;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = mrow0 + (e & 3) + 8 * (e >> 2);
                if (n_ok && m < p.M_g) dst[((size_t)b * p.Cdst + g * p.M_g + m) * p.Tdst + to] = acc[im][j][e];
            }
            continue;
        }
#pragma unroll
      for (int e0 = 0; e0 < 16; e0 += 8) {
        float sc[8], bi[8], rs[8];
        int chn[8], oidx[8];                         // element counts are checked < 2^31 on the host
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int m = mrow0 + ((e0 + e) & 3) + 8 * ((e0 + e) >> 2);
            chn[e] = g * p.M_g + (m < p.M_g ? m : p.M_g - 1);
            oidx[e] = (b * p.Cdst + chn[e]) * p.Tdst + to;
            sc[e] = 1.f; bi[e] = 0.f; rs[e] = 0.f;
        }
        if (p.mode == 0) {
            if (p.scale) {
#pragma unroll
                for (int e = 0; e < 8; ++e) sc[e] = p.scale[chn[e]];
            }
            if (p.bias) {
#pragma unroll
                for (int e = 0; e < 8; ++e) bi[e] = p.bias[chn[e]];
            }
        } else {
            if (p.scale) {
                const float us = p.scale[0];
#pragma unroll
                for (int e = 0; e < 8; ++e) sc[e] = us;
            }
            if (p.mask) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xv = p.mask[(b * p.Cdst + chn[e]) * p.Tmask + (to >> p.ups)];
                    if (!(xv > 0.f)) sc[e] = 0.f;
                }
            }
        }
        if (p.res) {
#pragma unroll
            for (int e = 0; e < 8; ++e) rs[e] = p.res[oidx[e]];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int m = mrow0 + ((e0 + e) & 3) + 8 * ((e0 + e) >> 2);
            float v = acc[im][j][e0 + e] * sc[e] + bi[e] + rs[e];
            if (p.out_act == STEGAN_ACT_LRELU) v = v > 0.f ? v : 0.1f * v;
            else if (p.out_act == STEGAN_ACT_RELU) v = fmaxf(v, 0.f);
            else if (p.out_act == STEGAN_ACT_TANH) v = tanhf(v);
            if (n_ok && m < p.M_g) {
                if (p.out_btc) p.out[((size_t)b * p.Tdst + to) * p.Cdst + g * p.M_g + m] = v;
                else p.out[oidx[e]] = v;
            }
        }
      }
     }
    }
#undef STAMP
}

template <int TM, int TN, bool FAST, bool XVEC, bool W14 = false, bool TAPG = false>
__global__ __launch_bounds__(256) void conv_direct_kernel(const DirP p) {
    conv_direct_body<TM, TN, FAST, XVEC, W14, TAPG>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Multi-problem launch: up to MAX_MULTI independent convolutions that share a kernel variant (the same layer of the
// period discriminators: one channel structure, different weights, batch and length) in ONE grid -- their ~6 us of
// per-launch fixed cost is paid once and the blocks of all problems fill the CUs together.  Every problem runs in the
// 1-D "tail split" decode with no split tiles (mix_full = all its tiles).
constexpr int MAX_MULTI = 5;
struct MultiP {
    int n;
    int first[MAX_MULTI + 1];       // first linear block of each problem
    DirP p[MAX_MULTI];
};

static_assert(sizeof(MultiP) <= 4096, "kernel argument segment limit");

template <bool FAST, bool XVEC>
__global__ __launch_bounds__(256) void conv_direct_multi_kernel(const MultiP a) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAX_MULTI; ++i)
        if (i < a.n && (int)blockIdx.x >= a.first[i]) pi = i;
    conv_direct_body<1, 1, FAST, XVEC, false, false>(a.p[pi], blockIdx.x - a.first[pi], 0, 0);
}

// ---- weight transposition: v (Cout, Cin_g, K) -> Wf[g][k][ci][co_pad], Wb[g][k][co][ci_pad]
__global__ __launch_bounds__(256) void conv_prep_kernel(const float* __restrict__ v, const float* __restrict__ scale,
                                                        float* __restrict__ wf, float* __restrict__ wb, int Cout_g,
                                                        int Cin_g, int K, int G, int LDf, int LDb) {
    // tile: 32 co x 32 e, e = ci*K + k   (reads coalesced along e, Wf writes coalesced along co)
    __shared__ float t[32][33];
    const int E = Cin_g * K;
    const int e0 = blockIdx.x * 32, co0 = blockIdx.y * 32, g = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 8 rows per pass
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, e = e0 + tx;
        t[r][tx] = (co < Cout_g && e < E) ? v[((size_t)(g * Cout_g + co)) * E + e] : 0.f;
    }
    block_sync();
    for (int r = ty; r < 32; r += 8) {
        // Wf: row = e index r, column = co tx
        const int e = e0 + r, co = co0 + tx;
        if (e < E && co < Cout_g) {
            const int ci = e / K, k = e - ci * K;
            wf[(((size_t)g * K + k) * Cin_g + ci) * LDf + co] = t[tx][r];
        }
        // Wb: element (co = r-th row of tile, e = tx)
        const int co2 = co0 + r, e2 = e0 + tx;
        if (e2 < E && co2 < Cout_g) {
            const int ci = e2 / K, k = e2 - ci * K;
            wb[(((size_t)g * K + k) * Cout_g + co2) * LDb + ci] = t[r][tx] * (scale ? scale[g * Cout_g + co2] : 1.f);
        }
    }
}

// multi-tensor form: one launch transposes every conv of a network (grid.y = conv, grid.x = tile of the largest conv)
struct PrepEntry {
    const float* v;
    const float* scale;
    float* wf;
    float* wb;
    int Cout_g, Cin_g, K, G, LDf, LDb, tiles_e, tiles_co;
};

__global__ __launch_bounds__(256) void conv_prep_multi_kernel(const PrepEntry* __restrict__ table) {
    const PrepEntry t0 = table[blockIdx.y];
    const int per_group = t0.tiles_e * t0.tiles_co;
    if ((int)blockIdx.x >= per_group * t0.G) return;
    __shared__ float t[32][33];
    const int g = blockIdx.x / per_group, rem = blockIdx.x - g * per_group;
    const int e0 = (rem % t0.tiles_e) * 32, co0 = (rem / t0.tiles_e) * 32;
    const int E = t0.Cin_g * t0.K;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, e = e0 + tx;
        t[r][tx] = (co < t0.Cout_g && e < E) ? t0.v[((size_t)(g * t0.Cout_g + co)) * E + e] : 0.f;
    }
    block_sync();
    for (int r = ty; r < 32; r += 8) {
        const int e = e0 + r, co = co0 + tx;
        if (e < E && co < t0.Cout_g) {
            const int ci = e / t0.K, k = e - ci * t0.K;
            t0.wf[(((size_t)g * t0.K + k) * t0.Cin_g + ci) * t0.LDf + co] = t[tx][r];
        }
        const int co2 = co0 + r, e2 = e0 + tx;
        if (e2 < E && co2 < t0.Cout_g) {
            const int ci = e2 / t0.K, k = e2 - ci * t0.K;
            t0.wb[(((size_t)g * t0.K + k) * t0.Cout_g + co2) * t0.LDb + ci] =
                t[r][tx] * (t0.scale ? t0.scale[g * t0.Cout_g + co2] : 1.f);
        }
    }
}

struct Geo { int CC, logCC, SEG, NSEG, XROW, BN, WR; size_t lds; bool ok; };

// geometry for one launch; ntaps_max bounds the taps of any phase
Geo plan_geo(int ntaps_max, int span_off, int cs, int U, int U_min, long N, long other_blocks, bool xvec,
             int force_bn = 0) {
    Geo gm;
    gm.ok = false;
    if (ntaps_max > MAX_TAPS) return gm;
    // BN: 128 when the grid still fills the chip, else 64
    for (int BN : {128, 64}) {
        if (force_bn && BN != force_bn) continue;
        const int UT = U < BN ? U : BN;
        int SEG = (UT - 1) * cs + span_off + 1;
        if (xvec) SEG = (SEG + 3 + 3) & ~3;          // + up to 3 slots of down-rounding, padded to whole float4s
        // batch elements one tile can touch: worst case over the per-phase column counts U_min..U
        int NSEG = 1;
        for (int uu = (U_min < 1 ? 1 : U_min); uu <= U; ++uu) {
            const int ns = (uu % BN == 0) ? 1 : ((BN + uu - 2) / uu + 1);
            if (ns > NSEG) NSEG = ns;
        }
        const int XROW = SEG * NSEG;
        int CC = 32;
        while (CC > 1 && (ntaps_max * CC > 64 || CC * XROW > 256 * XE)) CC >>= 1;
        if (CC * XROW > 256 * XE) continue;
        const int WR = round_up(ntaps_max * CC, 8);
        const size_t lds = 2 * ((size_t)WR * 64 * 4 + (size_t)CC * XROW * 4);
        const long blocks = (N + BN - 1) / BN;
        static const char* env128 = getenv("STEGAN_BN128_MIN_BLOCKS");
        const long min128 = env128 ? atol(env128) : 768;
        if (BN == 64 || force_bn || blocks * other_blocks >= min128) {
            gm.CC = CC;
            gm.logCC = 0;
            while ((1 << gm.logCC) < CC) ++gm.logCC;
            gm.SEG = SEG; gm.NSEG = NSEG; gm.XROW = XROW; gm.BN = BN; gm.WR = WR; gm.lds = lds; gm.ok = true;
            return gm;
        }
    }
    return gm;
}

}  // namespace

extern "C" int64_t stegan_conv_prep_floats(const stegan_conv_desc* d, int32_t which) {
    if (!d || d->groups <= 0) return -1;
    const int Cin_g = d->Cin / d->groups, Cout_g = d->Cout / d->groups;
    if (which == 0) return (int64_t)d->groups * d->K * Cin_g * round_up(Cout_g, 64) + 128;
    return (int64_t)d->groups * d->K * Cout_g * round_up(Cin_g, 64) + 128;
}

extern "C" int stegan_conv_prep(const stegan_conv_desc* d, const float* v, const float* scale, float* wf, float* wb,
                                void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    STEGAN_CHECK_ARG(v && wf && wb);
    const int Cin_g = d->Cin / d->groups, Cout_g = d->Cout / d->groups;
    const int E = Cin_g * d->K;
    // padding columns are never read into valid outputs (rows m >= M are masked), but keep them finite
    hipStream_t s = (hipStream_t)stream;


    hipLaunchKernelGGL(conv_prep_kernel, dim3(ceil_div(E, 32), ceil_div(Cout_g, 32), d->groups), dim3(256), 0, s, v, scale, wf,
                       wb, Cout_g, Cin_g, d->K, d->groups, round_up(Cout_g, 64), round_up(Cin_g, 64));
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_conv_prep_multi(int32_t n, const void* table_dev, int32_t max_tiles, void* stream) {
    STEGAN_CHECK_ARG(n > 0 && table_dev && max_tiles > 0);
    hipLaunchKernelGGL(conv_prep_multi_kernel, dim3(max_tiles, n), dim3(256), 0, (hipStream_t)stream,
                       (const PrepEntry*)table_dev);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

// returns 1 when the v2 kernel cannot take this shape (caller falls back to the generic implicit-GEMM kernel)
struct MultiCollect {             // host side of a multi-problem launch being assembled
    MultiP mp;
    bool fast, xvec;
    bool allow_xvec;              // false when some problem's rows are not float4 aligned: all use the 4-byte X DMA
    size_t lds;
    long blocks;
};

static int launch_direct(const stegan_conv_desc* d, int mode, DirP& p, float* workspace, int64_t workspace_floats,
                         hipStream_t s, MultiCollect* mc = nullptr) {
    int k_lo, k_hi;
    const int K_eff = live_taps(d, k_lo, k_hi);
    if (K_eff <= 0) return STEGAN_E_ARG;
    p.K = d->K; p.stride = d->stride; p.dil = d->dilation; p.pad = d->pad; p.k_lo = k_lo; p.k_hi = k_hi;
    p.B = d->B; p.mode = mode;
    p.fdS = make_fastdiv((uint32_t)d->stride);
    const int Te = d->Tin * d->upsample;
    int ntaps_max, span_off, cs, U, U_min, phases;
    if (mode == 0) {
        ntaps_max = K_eff; span_off = (k_hi - k_lo) * d->dilation; cs = d->stride; U = d->Tout; U_min = U; phases = 1;
    } else {
        // per phase: taps k with (phi + pad - k*dil) % s == 0; offsets (phi+pad-k*dil)/s
        ntaps_max = 0; span_off = 0;
        for (int phi = 0; phi < d->stride; ++phi) {
            int nt = 0, omin = 1 << 30, omax = -(1 << 30);
            for (int k = k_lo; k <= k_hi; ++k) {
                const int a = phi + d->pad - k * d->dilation;
                int q = a / d->stride, r = a % d->stride;
                if (r < 0) { r += d->stride; q -= 1; }
                if (r != 0) continue;
                ++nt; omin = q < omin ? q : omin; omax = q > omax ? q : omax;
            }
            if (nt > ntaps_max) ntaps_max = nt;
            if (nt > 0 && omax - omin > span_off) span_off = omax - omin;
        }
        if (ntaps_max == 0) ntaps_max = 1;
        cs = 1; U = (Te + d->stride - 1) / d->stride; U_min = Te / d->stride; phases = d->stride;
    }
    const long N = (long)d->B * U;
    // Tap groups: with more than 8 live taps a chunk of ALL taps leaves room for only 1-4 channels (rows <= 64) and the
    // slow generic MFMA loop; instead a W chunk takes TG = 4 or 8 taps x (64 / TG) channels -- the fast loop's shape --
    // and the halo'd X tile of a channel chunk is staged once and reused by its NTG tap groups.
    int TG = ntaps_max, NTG = 1;
    const int span_all = span_off;
    if (mc && ntaps_max > 8) return 1;                  // multi-problem launches: plain variants only
    if (ntaps_max > 8 && !STEGAN_ENV("STEGAN_NO_TAPGROUPS")) {
        const int pad8 = ceil_div(ntaps_max, 8) * 8, pad4 = ceil_div(ntaps_max, 4) * 4;
        TG = pad4 < pad8 ? 4 : 8;
        NTG = ceil_div(ntaps_max, TG);
        int step = d->dilation;
        if (mode == 1) {
            int gg = d->dilation % d->stride, hh = d->stride;
            while (gg) { const int tt = hh % gg; hh = gg; gg = tt; }
            step = d->dilation / hh;
        }
        span_off = (NTG * TG - 1) * step;              // the dead taps of the last group read real halo data
    }
    // 16-byte X DMA needs float4-aligned rows: row length and valid extent multiples of 4, no upsample, aligned base
    const bool xvec = p.Tsrc % 4 == 0 && p.Tq % 4 == 0 && (mode == 1 || p.ups == 0) && ((uintptr_t)p.src % 16) == 0 &&
                      !STEGAN_ENV("STEGAN_NO_XVEC") && (!mc || mc->allow_xvec);
    if (ntaps_max > MAX_TAPS) return 1;
    Geo gm = plan_geo(TG, span_off, cs, U, U_min, N, (long)ceil_div(p.M_g, BM2) * d->groups * phases, xvec, mc ? 64 : 0);
    if (!gm.ok) return 1;
    // (narrow problems join a multi launch in the plain 2 x 2 layout: tiny layers, the shared fixed cost matters more)
    if (NTG > 1 && gm.CC < 8) {                        // X tile too wide for 8 channels: back to all taps in one chunk
        TG = ntaps_max; NTG = 1;
        span_off = span_all;
        gm = plan_geo(TG, span_off, cs, U, U_min, N, (long)ceil_div(p.M_g, BM2) * d->groups * phases, xvec);
        if (!gm.ok) return 1;
    }
    // narrow groups (<= 32 output rows per group / phase): four waves along N, see W14.  Measured on the K = 37 grouped
    // layers: not multiplying padding rows matters more than the fast loop, so when the 128-column X tile leaves no room
    // for 8-channel tap groups the layer keeps W14 with all taps in one chunk (generic loop).
    static const char* no_w14 = getenv("STEGAN_NO_W14");
    bool w14 = false;
    if (!mc && !no_w14 && p.M_g <= 32) {
        Geo g2 = plan_geo(TG, span_off, cs, U, U_min, N, (long)d->groups * phases, xvec, 128);
        if (g2.ok && (NTG == 1 ? g2.CC >= gm.CC : g2.CC >= 8)) { gm = g2; w14 = true; }
        else if (NTG > 1) {
            Geo g3 = plan_geo(ntaps_max, span_all, cs, U, U_min, N, (long)d->groups * phases, xvec, 128);
            if (g3.ok) { gm = g3; w14 = true; TG = ntaps_max; NTG = 1; span_off = span_all; }
        }
    }
    p.CC = gm.CC; p.logCC = gm.logCC; p.SEG = gm.SEG; p.NSEG = gm.NSEG; p.XROW = gm.XROW; p.WR = gm.WR; p.U0 = U;
    p.fdXROW = make_fastdiv((uint32_t)gm.XROW); p.fdSEG = make_fastdiv((uint32_t)gm.SEG);
    p.fdU0 = make_fastdiv((uint32_t)U); p.fdU1 = make_fastdiv((uint32_t)(U > 1 ? U - 1 : 1));
    const long nb = (N + gm.BN - 1) / gm.BN;
    p.TG = TG; p.NTG = NTG; p.fdNTG = make_fastdiv((uint32_t)NTG);
    const int nchunks = ceil_div(p.C2_g, gm.CC) * NTG;
    const bool fast = gm.CC >= 8;
    // M tile: 128 rows (two MFMA row tiles per wave: twice the matrix work per staged X element, per barrier and per
    // block prologue / epilogue) when the channel count allows it and the cost model does not lose to 64-row tiles
    double cost1 = 0, cost2 = 1e30;
    int splits = choose_splits(nb * ceil_div(p.M_g, 64) * d->groups * phases, nchunks, 1.0, &cost1);
    int tm = 1;
    static const char* env_tm = getenv("STEGAN_TM");
    // measured (scripts/conv_microbench.py): 128-row tiles win only on the small-N layers (few output tiles, deep
    // reductions); with plenty of tiles the smaller LDS footprint of 64-row tiles (3 resident blocks per CU) wins
    const long base1 = nb * ceil_div(p.M_g, 64) * d->groups * phases;
    if (mc) splits = 1;
    if (!mc && !w14 && NTG == 1 && fast && gm.BN == 64 && p.M_g % 128 == 0 && ((!env_tm && base1 <= 288) || (env_tm && atoi(env_tm) == 2))) {
        const int s2 = choose_splits(nb * (p.M_g / 128) * d->groups * phases, nchunks, 1.7, &cost2);
        tm = 2; splits = s2;
    }
    if (tm == 2) gm.lds = 2 * ((size_t)gm.WR * 128 * 4 + (size_t)gm.CC * gm.XROW * 4);
    p.m_tiles = ceil_div(p.M_g, 64 * tm);
    p.tile_m = 64 * tm; p.tile_n = gm.BN;
    const int64_t out_elems = (int64_t)d->B * p.Cdst * p.Tdst;
    if (splits > 1 && (!workspace || workspace_floats < out_elems * splits)) {
        splits = workspace ? (int)(workspace_floats / out_elems) : 1;     // use what the caller provided
        if (splits < 2) splits = 1;
    }
    p.splits = splits;
    p.chunks_per_split = ceil_div(nchunks, splits);
    p.splits = ceil_div(nchunks, p.chunks_per_split);
    p.partial = workspace;
    // Tail split: with T tiles on 256 CUs, T mod 256 CUs carry one whole block more than the rest and set the launch
    // time (every generator layer has 576 tiles: 3 blocks on 64 CUs, 2 on 192).  Cutting just those tail tiles into S
    // reduction slices gives every CU the same number of whole tiles plus one short block.
    p.mix_S = 0; p.mix_full = 0; p.mix_cps = 0; p.nbx = (int)nb; p.ny = p.m_tiles * d->groups;
    {
        static const char* no_tail = getenv("STEGAN_NO_TAILSPLIT");
        const long tiles = nb * p.ny * phases;
        const long q = tiles / 256, r = tiles % 256;
        if (mc) {                                              // 1-D decode, every tile whole
            p.mix_S = 1; p.mix_full = (int)tiles; p.mix_cps = nchunks;
        } else if (!no_tail && tm == 1 && gm.BN == 64 && q >= 2 && q <= 8 && r > 0 && nchunks >= 4 && workspace &&
            workspace_floats >= 2 * out_elems && tiles < (1L << 30)) {
            long S = 256 / r;
            if (S < 2) S = 2;
            if (S > nchunks / 2) S = nchunks / 2;
            if (S > 8) S = 8;
            if (S > workspace_floats / out_elems) S = workspace_floats / out_elems;
            if (S >= 2) {
                // same cost units as choose_splits: whole tiles on every CU, then the short blocks of the tail tiles
                const int cps = ceil_div(nchunks, (int)S);
                const long tail_rounds = (r * ceil_div(nchunks, cps) + 255) / 256;
                const double cost_tail = (double)q * (nchunks + 2.5) + (double)tail_rounds * (cps + 2.5) + 1.0;
                if (cost_tail < cost1 - 1e-9) {
                    p.splits = 1; p.chunks_per_split = nchunks;
                    p.mix_cps = cps;
                    p.mix_S = ceil_div(nchunks, cps);
                    p.mix_full = (int)(tiles - r);
                }
            }
        }
    }
#ifdef STEGAN_STAMPS             // diagnostic builds only: the stamp buffer's device address (decimal) from the environment
    {
        static const char* se = getenv("STEGAN_STAMP_PTR");
        p.stamps = se ? (unsigned long long*)strtoull(se, nullptr, 10) : nullptr;
    }
#else
    p.stamps = nullptr;
#endif
    {
        static const char* dbg = getenv("STEGAN_PLAN_DEBUG");
        if (dbg) fprintf(stderr, "plan mode=%d B=%d C2=%d M=%d K=%d s=%d d=%d T=%d | BN=%d tm=%d CC=%d nchunks=%d tiles=%ld splits=%d mix_S=%d full=%d lds=%zu TG=%d NTG=%d w14=%d XROW=%d\n",
                         mode, d->B, p.C2_g, p.M_g, d->K, d->stride, d->dilation, p.Tdst, gm.BN, tm, gm.CC, nchunks,
                         nb * p.ny * phases, p.splits, p.mix_S, p.mix_full, gm.lds, TG, NTG, (int)w14, gm.XROW);
    }
    if (phases > MAX_PHASES) return 1;
    p.fdMT = make_fastdiv((uint32_t)p.m_tiles); p.fdSP = make_fastdiv((uint32_t)p.splits);
    p.fdNBX = make_fastdiv((uint32_t)p.nbx); p.fdNY = make_fastdiv((uint32_t)p.ny);
    p.fdMS = make_fastdiv((uint32_t)(p.mix_S > 0 ? p.mix_S : 1));
    for (int ph = 0; ph < MAX_PHASES; ++ph) {
        const Taps t = phase_taps_host(p, ph < phases ? ph : 0);
        p.ph_ntaps[ph] = t.ntaps; p.ph_kbase[ph] = t.kbase; p.ph_kstep[ph] = t.kstep; p.ph_offmin[ph] = t.off_min;
        p.ph_step[ph] = t.step;
        p.ph_U[ph] = mode == 0 ? U : (p.Tdst - ph + d->stride - 1) / d->stride;
    }
    if (mc) {
        MultiP& mp = mc->mp;
        if (mp.n >= MAX_MULTI) return 1;
        if (mp.n == 0) { mc->fast = fast; mc->xvec = xvec; mc->lds = gm.lds; mc->blocks = 0; mp.first[0] = 0; }
        else if (mc->fast != fast || mc->xvec != xvec) return 1;
        if (gm.lds > mc->lds) mc->lds = gm.lds;
        mp.p[mp.n] = p;
        mc->blocks += nb * p.ny * phases;
        mp.first[++mp.n] = (int)mc->blocks;
        return STEGAN_OK;
    }
    dim3 grid((unsigned)nb, (unsigned)(p.m_tiles * d->groups), (unsigned)(phases * p.splits));
    const long tail_tiles = p.mix_S ? nb * p.ny * phases - p.mix_full : 0;
    if (p.mix_S) grid = dim3((unsigned)(p.mix_full + tail_tiles * p.mix_S), 1, 1);
    if (gm.lds > 64 * 1024) {
        static bool attr_done = false;
        if (!attr_done) {
            (void)hipFuncSetAttribute((const void*)conv_direct_kernel<2, 1, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
            (void)hipFuncSetAttribute((const void*)conv_direct_kernel<2, 1, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
            attr_done = true;
        }
    }
#define STEGAN_LAUNCH(TM_, TN_, F_, V_) hipLaunchKernelGGL((conv_direct_kernel<TM_, TN_, F_, V_>), grid, dim3(256), gm.lds, s, p)
#define STEGAN_LAUNCH_W14(F_, V_) hipLaunchKernelGGL((conv_direct_kernel<1, 1, F_, V_, true>), grid, dim3(256), gm.lds, s, p)
    if (NTG > 1) {                                           // tap groups: fast loop by construction (CC >= 8), 64-row tiles
        if (w14) {
            if (xvec) hipLaunchKernelGGL((conv_direct_kernel<1, 1, true, true, true, true>), grid, dim3(256), gm.lds, s, p);
            else hipLaunchKernelGGL((conv_direct_kernel<1, 1, true, false, true, true>), grid, dim3(256), gm.lds, s, p);
        } else {
            if (xvec) hipLaunchKernelGGL((conv_direct_kernel<1, 1, true, true, false, true>), grid, dim3(256), gm.lds, s, p);
            else hipLaunchKernelGGL((conv_direct_kernel<1, 1, true, false, false, true>), grid, dim3(256), gm.lds, s, p);
        }
    } else if (w14) {
        if (fast) { if (xvec) STEGAN_LAUNCH_W14(true, true); else STEGAN_LAUNCH_W14(true, false); }
        else      { if (xvec) STEGAN_LAUNCH_W14(false, true); else STEGAN_LAUNCH_W14(false, false); }
    } else if (tm == 2) {
        if (xvec) STEGAN_LAUNCH(2, 1, true, true); else STEGAN_LAUNCH(2, 1, true, false);
    } else if (gm.BN == 128) {
        if (fast) { if (xvec) STEGAN_LAUNCH(1, 2, true, true); else STEGAN_LAUNCH(1, 2, true, false); }
        else      { if (xvec) STEGAN_LAUNCH(1, 2, false, true); else STEGAN_LAUNCH(1, 2, false, false); }
    } else {
        if (fast) { if (xvec) STEGAN_LAUNCH(1, 1, true, true); else STEGAN_LAUNCH(1, 1, true, false); }
        else      { if (xvec) STEGAN_LAUNCH(1, 1, false, true); else STEGAN_LAUNCH(1, 1, false, false); }
    }
#undef STEGAN_LAUNCH
#undef STEGAN_LAUNCH_W14
    if (p.mix_S) hipLaunchKernelGGL(conv_tail_epilogue_kernel, dim3((unsigned)tail_tiles, 16), dim3(256), 0, s, p);
    if (p.splits > 1) {
        long eb = (((p.Tdst & 3) == 0 ? out_elems / 4 : out_elems) + 255) / 256;
        if (eb > 4096) eb = 4096;
        hipLaunchKernelGGL(conv_splitk_epilogue_kernel, dim3((unsigned)eb), dim3(256), 0, s, p);
    }
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int64_t stegan_conv_workspace_floats(const stegan_conv_desc* d, int32_t mode) {
    if (!d) return -1;
    const int64_t out_elems = mode == 0 ? (int64_t)d->B * d->Cout * d->Tout : (int64_t)d->B * d->Cin * d->Tin * d->upsample;
    return out_elems * 16;      // upper bound: at most 16 splits
}

extern "C" int stegan_conv1d_fwd_t(const stegan_conv_desc* d, const float* x, const float* wf, const float* scale,
                                   const float* bias, const float* residual, float* y, float* workspace,
                                   int64_t workspace_floats, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    STEGAN_CHECK_ARG(x && wf && y);
    STEGAN_CHECK_ARG(d->in_act == STEGAN_ACT_NONE || d->in_act == STEGAN_ACT_RELU);
    STEGAN_CHECK_ARG(d->out_act >= STEGAN_ACT_NONE && d->out_act <= STEGAN_ACT_TANH);
    STEGAN_CHECK_ARG(!(d->out_btc && residual));
    DirP p;
    p.src = x; p.wt = wf; p.scale = scale; p.bias = bias; p.res = residual; p.mask = nullptr; p.out = y;
    p.Csrc = d->Cin; p.Cdst = d->Cout; p.Tsrc = d->Tin; p.Tq = d->Tin * d->upsample; p.Tdst = d->Tout;
    p.Tmask = 0; p.ups = d->upsample == 2 ? 1 : 0;
    p.C2_g = d->Cin / d->groups; p.M_g = d->Cout / d->groups; p.LD = round_up(p.M_g, 64);
    p.zero_off = (long)d->groups * d->K * p.C2_g * p.LD;
    p.in_relu = d->in_act == STEGAN_ACT_RELU; p.out_act = d->out_act; p.out_btc = d->out_btc;
    return launch_direct(d, 0, p, workspace, workspace_floats, (hipStream_t)stream);
}

static int launch_multi(MultiCollect& mc, hipStream_t s) {
    dim3 grid((unsigned)mc.blocks);
    if (mc.fast) {
        if (mc.xvec) hipLaunchKernelGGL((conv_direct_multi_kernel<true, true>), grid, dim3(256), mc.lds, s, mc.mp);
        else hipLaunchKernelGGL((conv_direct_multi_kernel<true, false>), grid, dim3(256), mc.lds, s, mc.mp);
    } else {
        if (mc.xvec) hipLaunchKernelGGL((conv_direct_multi_kernel<false, true>), grid, dim3(256), mc.lds, s, mc.mp);
        else hipLaunchKernelGGL((conv_direct_multi_kernel<false, false>), grid, dim3(256), mc.lds, s, mc.mp);
    }
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_conv1d_bwd_data_multi(int32_t n, const stegan_conv_desc* const* descs, const float* const* dy,
                                            const float* const* wb, const float* const* mask_src,
                                            const float* const* add_src, float* const* dx, void* stream) {
    STEGAN_CHECK_ARG(n >= 1 && n <= MAX_MULTI && descs && dy && wb && dx);
    MultiCollect mc;
    mc.mp.n = 0;
    mc.allow_xvec = true;
    for (int i = 0; i < n; ++i)
        if (!descs[i] || descs[i]->Tout % 4 != 0 || ((uintptr_t)dy[i] % 16) != 0) mc.allow_xvec = false;
    for (int i = 0; i < n; ++i) {
        const stegan_conv_desc* d = descs[i];
        int rc = check_desc(d);
        if (rc) return rc;
        STEGAN_CHECK_ARG(dy[i] && wb[i] && dx[i]);
        const float* mk = mask_src ? mask_src[i] : nullptr;
        STEGAN_CHECK_ARG(d->in_act != STEGAN_ACT_RELU || mk);
        DirP p;
        p.src = dy[i]; p.wt = wb[i]; p.scale = nullptr; p.bias = nullptr; p.res = add_src ? add_src[i] : nullptr; p.out = dx[i];
        p.mask = d->in_act == STEGAN_ACT_RELU ? mk : nullptr;
        p.Csrc = d->Cout; p.Cdst = d->Cin; p.Tsrc = d->Tout; p.Tq = d->Tout; p.Tdst = d->Tin * d->upsample;
        p.Tmask = d->Tin; p.ups = d->upsample == 2 ? 1 : 0;
        p.C2_g = d->Cout / d->groups; p.M_g = d->Cin / d->groups; p.LD = round_up(p.M_g, 64);
        p.zero_off = (long)d->groups * d->K * p.C2_g * p.LD;
        p.in_relu = 0; p.out_act = 0; p.out_btc = 0;
        rc = launch_direct(d, 1, p, nullptr, 0, (hipStream_t)stream, &mc);
        if (rc) return rc;
    }
    return launch_multi(mc, (hipStream_t)stream);
}

extern "C" int stegan_conv1d_fwd_multi(int32_t n, const stegan_conv_desc* const* descs, const float* const* x,
                                       const float* const* wf, const float* const* scale, const float* const* bias,
                                       const float* const* residual, float* const* y, void* stream) {
    STEGAN_CHECK_ARG(n >= 1 && n <= MAX_MULTI && descs && x && wf && y);
    MultiCollect mc;
    mc.mp.n = 0;
    mc.allow_xvec = true;
    for (int i = 0; i < n; ++i)
        if (!descs[i] || descs[i]->Tin % 4 != 0 || descs[i]->upsample != 1 || ((uintptr_t)x[i] % 16) != 0) mc.allow_xvec = false;
    for (int i = 0; i < n; ++i) {
        const stegan_conv_desc* d = descs[i];
        int rc = check_desc(d);
        if (rc) return rc;
        STEGAN_CHECK_ARG(x[i] && wf[i] && y[i]);
        STEGAN_CHECK_ARG(d->in_act == STEGAN_ACT_NONE || d->in_act == STEGAN_ACT_RELU);
        STEGAN_CHECK_ARG(d->out_act >= STEGAN_ACT_NONE && d->out_act <= STEGAN_ACT_TANH);
        const float* res = residual ? residual[i] : nullptr;
        STEGAN_CHECK_ARG(!(d->out_btc && res));
        DirP p;
        p.src = x[i]; p.wt = wf[i]; p.scale = scale ? scale[i] : nullptr; p.bias = bias ? bias[i] : nullptr; p.res = res;
        p.mask = nullptr; p.out = y[i];
        p.Csrc = d->Cin; p.Cdst = d->Cout; p.Tsrc = d->Tin; p.Tq = d->Tin * d->upsample; p.Tdst = d->Tout;
        p.Tmask = 0; p.ups = d->upsample == 2 ? 1 : 0;
        p.C2_g = d->Cin / d->groups; p.M_g = d->Cout / d->groups; p.LD = round_up(p.M_g, 64);
        p.zero_off = (long)d->groups * d->K * p.C2_g * p.LD;
        p.in_relu = d->in_act == STEGAN_ACT_RELU; p.out_act = d->out_act; p.out_btc = d->out_btc;
        rc = launch_direct(d, 0, p, nullptr, 0, (hipStream_t)stream, &mc);
        if (rc) return rc;                               // 1: not expressible as one launch, caller launches one by one
    }
    return launch_multi(mc, (hipStream_t)stream);
}

extern "C" int stegan_conv1d_bwd_data_t(const stegan_conv_desc* d, const float* dy, const float* wb, const float* scale,
                                        const float* mask_src, const float* add_src, float* dx, float* workspace,
                                        int64_t workspace_floats, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    STEGAN_CHECK_ARG(dy && wb && dx);
    STEGAN_CHECK_ARG(d->in_act != STEGAN_ACT_RELU || mask_src);
    DirP p;
    p.src = dy; p.wt = wb; p.scale = scale; p.bias = nullptr; p.res = add_src; p.out = dx;
    p.mask = d->in_act == STEGAN_ACT_RELU ? mask_src : nullptr;
    p.Csrc = d->Cout; p.Cdst = d->Cin; p.Tsrc = d->Tout; p.Tq = d->Tout; p.Tdst = d->Tin * d->upsample;
    p.Tmask = d->Tin; p.ups = d->upsample == 2 ? 1 : 0;
    p.C2_g = d->Cout / d->groups; p.M_g = d->Cin / d->groups; p.LD = round_up(p.M_g, 64);
    p.zero_off = (long)d->groups * d->K * p.C2_g * p.LD;
    p.in_relu = 0; p.out_act = 0; p.out_btc = 0;
    return launch_direct(d, 1, p, workspace, workspace_floats, (hipStream_t)stream);
}
