// Direct 1-D convolution on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate): the throughput variant
// of conv_direct.hip for BASELINE configs 3-5 ("bf16 operands, fp32 master weights / accumulators / activations").
//
// Same computation, same launch descriptor (DirP), same fused epilogue as the fp32 kernel
//
//     out[b, m, to(u)] = sum_{c2, tap}  W[tap][m][c2] * src[b, c2, u*cs + off[tap]]
//
// but the operands are rounded to bf16 ONCE, while they are staged, and the LDS images are channel-contiguous so that
// one MFMA operand (8 consecutive reduction channels of one row / column) is ONE 16-byte ds_read_b128:
//
//   * X tile  Xs[slot][8 NCB channels] bf16 -- activations stay fp32 (B, C, T) in HBM; a thread loads the 8 channels of
//     one (8-channel block, time slot) with 8 buffer_load_dword (lanes along time: coalesced; the channel stride is a
//     scalar offset; halo / padding / out-of-range slots carry an out-of-range offset and read 0 from the buffer unit),
//     applies the input ReLU on the packed halves (v_pk_max_i16 x, 0), converts with v_cvt_pk_bf16_f32 and stores ONE
//     ds_write_b128.  Rows are padded by 16 bytes (odd pitch in 16-byte units): operand reads of 32 consecutive rows are
//     bank-conflict free.  A tap shift is a ROW offset, so any dilation / stride is an address offset, never a gather;
//     the nearest-x2 upsample is resolved by the per-slot source offset.  Loads for chunk c+1 are issued before the MFMA
//     loop of chunk c and written to LDS after it (registers carry them across): global latency hides under the MFMAs.
//   * W tile  Ws[tap][m][8 NCB channels] bf16 from a bf16 copy of the direction tensor laid out [g][k][m][c] (c
//     contiguous, zero padded to 128 rows / 64 channels, plus one all-zero tap plane), produced once per weight
//     version by stegan_conv_prep_bf16: staged by LDS-DMA (global_load_lds, 16 B per lane) with the XOR swizzle applied
//     to the per-lane SOURCE address (linear LDS image), the same swizzle on the operand read: conflict free.
//   * the reduction runs over "atoms" = (tap, 8-channel block); one MFMA consumes two atoms (lane halves): with >= 16
//     channels per chunk the halves take neighbouring channel blocks of one tap, with 8-channel chunks (first layers,
//     narrow groups, many-tap layers where the W tile must stay small) they take neighbouring TAPS.
//   * block = 4 waves (2 x 2), wave tile 32 TM x 32 TN, block tile up to 128 x 128; two LDS buffers, one barrier per
//     chunk; split-K / tail split and their fixed-order epilogue kernels are shared with the fp32 kernel.
//
// Numerics: products of bf16-rounded operands accumulated in fp32 (k-ordered chain inside the MFMA); everything around
// the contraction (scale, bias, residual, activations, masks) is fp32 exactly as in conv_direct.hip.
#include "conv_direct_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int HX_MAX = 6;      // X items (8 channels x 1 slot -> one 16-byte LDS store) per thread and chunk
constexpr int HW_MAX = 12;     // W 16-byte DMA pieces per thread and chunk
constexpr unsigned OOB = 0x7ffff000u;   // buffer offset beyond any source tensor (host checks bytes < 2^31 - 2^13): reads 0

struct HP {
    DirP d;                       // CC = channels per chunk (8 NCB); XROW etc. as in the fp32 kernel
    const unsigned short* wh;     // bf16 [G][K + 1][Mpad][Cpad]; plane K is all zero
    int Mpad, Cpad;
    int XROWP;                    // X rows rounded up to 64 (staging items are laid out per 64-slot wave row)
    int XP;                       // X row pitch in bytes
    int ntapsW;                   // taps in the W tile = max over phases (rounded up to even when NCB == 1)
    int xit, wit;                 // staging items per thread
    unsigned src_bytes;
    FastDiv fdXROWP;
};

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b, bool relu) {
    const f32x2 f = {a, b};
    s16x2 s = __builtin_bit_cast(s16x2, __builtin_convertvector(f, bf16x2));
    if (relu) s = __builtin_elementwise_max(s, (s16x2)(0));       // sign bit set <=> negative int16: relu on packed halves
    return __builtin_bit_cast(unsigned, s);
}

template <int TM, int TN, int LOGNCB>
__device__ __forceinline__ void conv_bf16_body(const HP& h, const unsigned bid_x, const unsigned bid_y, const unsigned bid_z) {
    const DirP& p = h.d;
    constexpr int NCB = 1 << LOGNCB;
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int KS = NCB >= 2 ? NCB / 2 : 1;        // MFMA k-steps per tap (NCB >= 2) / per tap pair (NCB == 1)
    constexpr int WROWB = NCB * 16;                   // bytes per W row
    extern __shared__ __align__(16) char smem[];      // W buf 0 | W buf 1 | X buf 0 | X buf 1

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
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

    int U, cs, os;
    if (p.mode == 0) { U = p.U0; cs = p.stride; os = 1; }
    else { U = p.ph_U[phase]; cs = 1; os = p.stride; }
    const int N = p.B * U;
    const int n0 = bx * BN;
    if (n0 >= N) return;
    const FastDiv fdU = (U == p.U0) ? p.fdU0 : p.fdU1;
    const Taps tp = phase_taps(p, phase);
    const int ntaps = tp.ntaps, off_min = tp.off_min;
    const int ntapsW = NCB == 1 ? ((ntaps + 1) & ~1) : ntaps;       // NCB == 1: MFMA halves take neighbouring taps
    const int S = NCB == 1 ? (ntapsW >> 1) : ntaps * KS;            // MFMA k-steps per chunk

    // ---- chunk-invariant part of the X staging: item = (8-channel block cg, slot); a wave's 64 items share cg
    const int b_lo = (int)fd_div((uint32_t)n0, fdU), u0 = n0 - b_lo * U;
    const int qbase0 = u0 * cs + off_min, qbaseN = off_min;
    unsigned x_voff[HX_MAX];
    int x_lds[HX_MAX], x_cg[HX_MAX];
#pragma unroll
    for (int it = 0; it < HX_MAX; ++it) {
        const int wbase = it * 256 + wave * 64;                     // wave-uniform
        const int cg = (int)fd_div((uint32_t)wbase, h.fdXROWP);
        const int slot = wbase - cg * h.XROWP + lane;
        x_cg[it] = (it < h.xit && cg < NCB) ? cg : -1;
        x_voff[it] = OOB;
        x_lds[it] = slot * h.XP + cg * 16;
        if (slot < p.XROW) {
            const int sg = (int)fd_div((uint32_t)slot, p.fdSEG), sl = slot - sg * p.SEG;
            const int b = b_lo + sg;
            const int q = (sg == 0 ? qbase0 : qbaseN) + sl;
            if (sg < p.NSEG && b < p.B && q >= 0 && q < p.Tq)
                x_voff[it] = (unsigned)(((b * p.Csrc + g * p.C2_g) * p.Tsrc + (p.mode == 0 ? (q >> p.ups) : q)) * 4);
        }
    }
    // ---- W staging: piece = it * 256 + tid -> LDS byte 16 * piece (linear image); row = piece >> LOGNCB = tap * BM + m,
    // physical 16-byte slot = piece & (NCB - 1) holds logical channel block slot ^ swz(m)
    int w_off[HW_MAX];            // byte offset inside the group's weights, without the chunk's channel term; -1: no piece
#pragma unroll
    for (int it = 0; it < HW_MAX; ++it) {
        const int piece = it * 256 + tid;
        const int row = piece >> LOGNCB, phys = piece & (NCB - 1);
        const int tap = row / BM, m = row - tap * BM;
        const int cb = phys ^ ((m >> (4 - LOGNCB)) & (NCB - 1));
        const int k = tap < ntaps ? tp.kbase + tap * tp.kstep : p.K;            // dead tap of an odd pair: the zero plane
        w_off[it] = (it < h.wit && tap < ntapsW) ? ((k * h.Mpad + m0 + m) * h.Cpad + cb * 8) * 2 : -1;
    }

    // ---- per-lane operand addresses
    const int hl = lane >> 5, l31 = lane & 31;
    int a_byte[TM], b_byte[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_byte[i] = ((wm * TM + i) * 32 + l31) * WROWB + (NCB == 1 ? hl * BM * 16 : 0);
    const int a_swz = (l31 >> (4 - LOGNCB)) & (NCB - 1);           // the same for every 32-row sub-tile
    int a_xo[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) a_xo[kk] = NCB == 1 ? 0 : (((2 * kk + hl) ^ a_swz) * 16);
    const int tapB = tp.step * h.XP;                               // bytes between the rows of neighbouring taps
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int jn = (wn * TN + j) * 32 + l31;
        int cb = 0;
        if (n0 + jn < N) {
            const int uu = u0 + jn;
            const int sg = (int)fd_div((uint32_t)uu, fdU), u = uu - sg * U;
            cb = sg * p.SEG + (u - (sg == 0 ? u0 : 0)) * cs;
        }
        b_byte[j] = cb * h.XP + (NCB == 1 ? hl * tapB : hl * 16);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int CCH = 8 * NCB;
    const int nchunks = ntaps > 0 ? (p.C2_g + CCH - 1) / CCH : 0;
    const int c_begin = split * cps;
    const int c_end = min(nchunks, c_begin + cps);
    const char* wt_g = (const char*)h.wh + (size_t)g * (p.K + 1) * h.Mpad * h.Cpad * 2;
    const int wbuf = h.ntapsW * BM * WROWB, xbuf = h.XROWP * h.XP;
    char* const Xbufs = smem + 2 * wbuf;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, (int)h.src_bytes, 0x00020000);
    const bool relu = p.in_relu != 0;
    const int row_bytes = p.Tsrc * 4;

    float xr[HX_MAX][8];
#define HB_LOAD_X(c_)                                                                                           \
    {                                                                                                           \
        const int c0_ = (c_) * CCH;                                                                             \
        _Pragma("unroll") for (int it = 0; it < HX_MAX; ++it) {                                                 \
            if (x_cg[it] >= 0) {                                                                                \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                 \
                    int ch_ = c0_ + x_cg[it] * 8 + j;                    /* channels past the group read a live */ \
                    ch_ = ch_ < p.C2_g ? ch_ : p.C2_g - 1;               /* one: their weights are zero         */ \
                    xr[it][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)x_voff[it], ch_ * row_bytes, 0)); \
                }                                                                                               \
            }                                                                                                   \
        }                                                                                                       \
    }
#define HB_STORE_X(buf_)                                                                                        \
    {                                                                                                           \
        char* Xd_ = Xbufs + (buf_) * xbuf;                                                                      \
        _Pragma("unroll") for (int it = 0; it < HX_MAX; ++it) {                                                 \
            if (x_cg[it] >= 0) {                                                                                \
                const u32x4 v_ = {pack_bf16x2(xr[it][0], xr[it][1], relu), pack_bf16x2(xr[it][2], xr[it][3], relu), \
                                  pack_bf16x2(xr[it][4], xr[it][5], relu), pack_bf16x2(xr[it][6], xr[it][7], relu)}; \
                *(u32x4*)(Xd_ + x_lds[it]) = v_;                                                                \
            }                                                                                                   \
        }                                                                                                       \
    }
#define HB_DMA_W(c_, buf_)                                                                                      \
    {                                                                                                           \
        const char* wc_ = wt_g + (c_) * CCH * 2;                                                                \
        char* Wd_ = smem + (buf_) * wbuf;                                                                       \
        _Pragma("unroll") for (int it = 0; it < HW_MAX; ++it) {                                                 \
            if (it < h.wit && (it * 256 + wave * 64) < ntapsW * BM * NCB) {                                     \
                const char* src_ = wc_ + (w_off[it] >= 0 ? w_off[it] : 0);                                      \
                __builtin_amdgcn_global_load_lds((gptr_t)src_, (lds_ptr_t)(Wd_ + (it * 256 + wave * 64) * 16), 16, 0, 0); \
            }                                                                                                   \
        }                                                                                                       \
    }

    // prologue: first chunk
    if (c_begin < c_end) {
        HB_LOAD_X(c_begin)
        HB_DMA_W(c_begin, 0)
        HB_STORE_X(0)
    }
    for (int c = c_begin; c < c_end; ++c) {
        const int cur = (c - c_begin) & 1;
        __syncthreads();                    // chunk c's X stores and W DMA (vmcnt(0) + barrier) are visible; nobody reads cur ^ 1
        const bool more = c + 1 < c_end;
        if (more) {
            HB_LOAD_X(c + 1)                // into registers; written to LDS behind the MFMA loop
            HB_DMA_W(c + 1, cur ^ 1)
        }
        const char* Wcur = smem + cur * wbuf;
        const char* Xcur = Xbufs + cur * xbuf;
#define HB_LDS(ptr_) (*(const bf16x8*)(ptr_))
        if constexpr (KS >= 2) {
            bf16x8 aA[TM], bA[TN], aB[TM], bB[TN];
#define HB_LOAD(a_, b_, tap_, KK_)                                                                              \
            {                                                                                                   \
                const char* wa_ = Wcur + (tap_) * (BM * WROWB) + a_xo[KK_];                                     \
                const char* xb_ = Xcur + (tap_) * tapB + (KK_) * 32;                                            \
                _Pragma("unroll") for (int i = 0; i < TM; ++i) a_[i] = HB_LDS(wa_ + a_byte[i]);                 \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) b_[j] = HB_LDS(xb_ + b_byte[j]);                 \
            }
#define HB_MFMA(a_, b_)                                                                                         \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                      \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                  \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[i], b_[j], acc[i][j], 0, 0, 0);
            HB_LOAD(aA, bA, 0, 0)
            for (int tap = 0; tap < ntaps; ++tap) {
                HB_LOAD(aB, bB, tap, 1)
                HB_MFMA(aA, bA)
                if constexpr (KS == 4) {
                    HB_LOAD(aA, bA, tap, 2)
                    HB_MFMA(aB, bB)
                    HB_LOAD(aB, bB, tap, 3)
                    HB_MFMA(aA, bA)
                }
                if (tap + 1 < ntaps) HB_LOAD(aA, bA, tap + 1, 0)
                HB_MFMA(aB, bB)
            }
#undef HB_LOAD
        } else {
            // one k-step per tap (NCB == 2) or per tap pair (NCB == 1): step s reads W rows at s * stepA, X rows at s * stepB
            const int stepA = (NCB == 1 ? 2 : 1) * BM * WROWB, stepB = (NCB == 1 ? 2 : 1) * tapB;
            bf16x8 aA[TM], bA[TN], aB[TM], bB[TN];
#define HB_LOAD(a_, b_, s_)                                                                                     \
            {                                                                                                   \
                const char* wa_ = Wcur + (s_) * stepA + a_xo[0];                                                \
                const char* xb_ = Xcur + (s_) * stepB;                                                          \
                _Pragma("unroll") for (int i = 0; i < TM; ++i) a_[i] = HB_LDS(wa_ + a_byte[i]);                 \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) b_[j] = HB_LDS(xb_ + b_byte[j]);                 \
            }
            if (S > 0) HB_LOAD(aA, bA, 0)
            int s = 0;
            for (; s + 1 < S; s += 2) {
                HB_LOAD(aB, bB, s + 1)
                HB_MFMA(aA, bA)
                if (s + 2 < S) HB_LOAD(aA, bA, s + 2)
                HB_MFMA(aB, bB)
            }
            if (s < S) { HB_MFMA(aA, bA) }
#undef HB_LOAD
        }
#undef HB_MFMA
#undef HB_LDS
        if (more) HB_STORE_X(cur ^ 1)
    }
#undef HB_LOAD_X
#undef HB_STORE_X
#undef HB_DMA_W

    // ---- epilogue (same as conv_direct.hip: the accumulator layout of the 32x32 MFMAs does not depend on the dtype)
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
            const int mrow0 = m0 + (wm * TM + im) * 32 + 4 * hl;
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
                int chn[8], oidx[8];
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
}

template <int TM, int TN, int LOGNCB>
__global__ __launch_bounds__(256) void conv_bf16_kernel(const HP h) {
    conv_bf16_body<TM, TN, LOGNCB>(h, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ---- bf16 weight copies: v (Cout, Cin_g, K) fp32 -> wf[g][k][co][ci], wb[g][k][ci][co] (x scale[co]); k = K: zero plane
struct PrepH {
    const float* v;
    const float* scale;
    unsigned short* wf;
    unsigned short* wb;
    int Cout_g, Cin_g, K, G;
    int Mf, Cf, Mb, Cb;           // padded rows / channels of wf and wb
    long nf, nb;                  // elements of wf / wb
};

__device__ __forceinline__ unsigned short to_bf16(float f) {
    return __builtin_bit_cast(unsigned short, (__bf16)f);
}

__device__ __forceinline__ void prep_bf16_elems(const PrepH& t, long e0, long stride) {
    const int K = t.K;
    for (long e = e0; e < t.nf + t.nb; e += stride) {
        if (e < t.nf) {
            const int c = (int)(e % t.Cf);
            long r = e / t.Cf;
            const int m = (int)(r % t.Mf); r /= t.Mf;
            const int k = (int)(r % (K + 1)), g = (int)(r / (K + 1));
            float v = 0.f;
            if (k < K && m < t.Cout_g && c < t.Cin_g) v = t.v[((size_t)(g * t.Cout_g + m) * t.Cin_g + c) * K + k];
            t.wf[e] = to_bf16(v);
        } else {
            const long eb = e - t.nf;
            const int c = (int)(eb % t.Cb);
            long r = eb / t.Cb;
            const int m = (int)(r % t.Mb); r /= t.Mb;
            const int k = (int)(r % (K + 1)), g = (int)(r / (K + 1));
            float v = 0.f;
            if (k < K && m < t.Cin_g && c < t.Cout_g) {
                const int co = g * t.Cout_g + c;
                v = t.v[((size_t)co * t.Cin_g + m) * K + k] * (t.scale ? t.scale[co] : 1.f);
            }
            t.wb[eb] = to_bf16(v);
        }
    }
}

__global__ __launch_bounds__(256) void conv_prep_bf16_kernel(const PrepH t) {
    prep_bf16_elems(t, (long)blockIdx.x * 256 + threadIdx.x, (long)gridDim.x * 256);
}

__global__ __launch_bounds__(256) void conv_prep_bf16_multi_kernel(const PrepH* __restrict__ table) {
    const PrepH t = table[blockIdx.y];
    prep_bf16_elems(t, (long)blockIdx.x * 256 + threadIdx.x, (long)gridDim.x * 256);
}

void fill_prep(PrepH& t, const stegan_conv_desc* d) {
    t.G = d->groups; t.K = d->K;
    t.Cout_g = d->Cout / d->groups; t.Cin_g = d->Cin / d->groups;
    t.Mf = round_up(t.Cout_g, 128); t.Cf = round_up(t.Cin_g, 64);
    t.Mb = round_up(t.Cin_g, 128); t.Cb = round_up(t.Cout_g, 64);
    t.nf = (long)t.G * (t.K + 1) * t.Mf * t.Cf;
    t.nb = (long)t.G * (t.K + 1) * t.Mb * t.Cb;
}

struct GeoH { int TM, TN, logNCB, SEG, NSEG, XROW, XROWP, XP, ntW, xit, wit, span; size_t lds; bool ok; };

// geometry for one (TM, TN): the largest channel chunk whose tiles fit
GeoH plan_bf16(int TM, int TN, int C2_g, int ntaps_max, int span_taps, int step_max, int cs, int U, int U_min) {
    GeoH gm;
    gm.ok = false; gm.TM = TM; gm.TN = TN;
    const int BM = 64 * TM, BN = 64 * TN;
    GeoH fallback = gm;
    for (int logNCB = 3; logNCB >= 0; --logNCB) {
        const int NCB = 1 << logNCB, CCH = 8 * NCB;
        if (logNCB > 0 && CCH > round_up(C2_g, 8) && CCH > 16) continue;       // chunk wider than the group: only padding
        if (logNCB > 0 && C2_g <= 8) continue;
        const int ntW = NCB == 1 ? round_up(ntaps_max, 2) : ntaps_max;
        const int span = NCB == 1 ? span_taps + (ntW - ntaps_max) * step_max : span_taps;   // the dead tap reads real rows
        const int UT = U < BN ? U : BN;
        const int SEG = (UT - 1) * cs + span + 1;
        int NSEG = 1;
        for (int uu = (U_min < 1 ? 1 : U_min); uu <= U; ++uu) {
            const int ns = (uu % BN == 0) ? 1 : ((BN + uu - 2) / uu + 1);
            if (ns > NSEG) NSEG = ns;
        }
        const int XROW = SEG * NSEG, XROWP = round_up(XROW, 64);
        const int XP = NCB == 1 ? 16 : (NCB + 1) * 16;
        const int xit = ceil_div(NCB * XROWP, 256), wit = ceil_div(ntW * BM * NCB, 256);
        if (xit > HX_MAX || wit > HW_MAX) continue;
        const size_t lds = 2 * ((size_t)ntW * BM * NCB * 16 + (size_t)XROWP * XP);
        if (lds > 160 * 1024) continue;
        GeoH c = gm;
        c.logNCB = logNCB; c.SEG = SEG; c.NSEG = NSEG; c.XROW = XROW; c.XROWP = XROWP; c.XP = XP; c.ntW = ntW;
        c.xit = xit; c.wit = wit; c.span = span; c.lds = lds; c.ok = true;
        if (lds <= 80 * 1024) return c;                  // two blocks per CU
        if (!fallback.ok) fallback = c;
    }
    return fallback;
}

template <int TM, int TN>
void launch_variant(int logNCB, dim3 grid, size_t lds, hipStream_t s, const HP& h) {
    switch (logNCB) {
        case 0: hipLaunchKernelGGL((conv_bf16_kernel<TM, TN, 0>), grid, dim3(256), lds, s, h); break;
        case 1: hipLaunchKernelGGL((conv_bf16_kernel<TM, TN, 1>), grid, dim3(256), lds, s, h); break;
        case 2: hipLaunchKernelGGL((conv_bf16_kernel<TM, TN, 2>), grid, dim3(256), lds, s, h); break;
        default: hipLaunchKernelGGL((conv_bf16_kernel<TM, TN, 3>), grid, dim3(256), lds, s, h); break;
    }
}

template <int TM, int TN>
void set_lds_attr() {
    (void)hipFuncSetAttribute((const void*)conv_bf16_kernel<TM, TN, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_bf16_kernel<TM, TN, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_bf16_kernel<TM, TN, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_bf16_kernel<TM, TN, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// returns 1 when this kernel does not take the shape (caller uses the fp32 kernels)
int launch_bf16(const stegan_conv_desc* d, int mode, HP& h, float* workspace, int64_t workspace_floats, hipStream_t s) {
    DirP& p = h.d;
    int k_lo, k_hi;
    const int K_eff = live_taps(d, k_lo, k_hi);
    if (K_eff <= 0) return STEGAN_E_ARG;
    p.K = d->K; p.stride = d->stride; p.dil = d->dilation; p.pad = d->pad; p.k_lo = k_lo; p.k_hi = k_hi;
    p.B = d->B; p.mode = mode;
    p.fdS = make_fastdiv((uint32_t)d->stride);
    // shapes the matrix cores cannot fill stay on the exact fp32 kernels: fewer than 8 reduction channels (the
    // input gradient of the Cout = 1 output convs) or fewer than 16 output rows per group (Cout = 1 / 8)
    static const char* env_minm = getenv("STEGAN_BF16_MIN_M");
    const int min_m = env_minm ? atoi(env_minm) : 16;
    if (p.C2_g < 8 || p.M_g < min_m) return 1;
    if ((int64_t)d->B * p.Csrc * p.Tsrc * 4 >= (int64_t)OOB) return 1;
    const int Te = d->Tin * d->upsample;
    int ntaps_max, span_off, step_max, cs, U, U_min, phases;
    if (mode == 0) {
        ntaps_max = K_eff; span_off = (k_hi - k_lo) * d->dilation; step_max = d->dilation; cs = d->stride; U = d->Tout; U_min = U;
        phases = 1;
    } else {
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
        int gg = d->dilation % d->stride, hh = d->stride;
        while (gg) { const int tt = hh % gg; hh = gg; gg = tt; }
        step_max = d->dilation / hh;
        cs = 1; U = (Te + d->stride - 1) / d->stride; U_min = Te / d->stride; phases = d->stride;
    }
    if (ntaps_max > MAX_TAPS || phases > MAX_PHASES) return 1;
    const long N = (long)d->B * U;
    // tile: 128 rows when the padding it adds stays small, 128 columns when the grid still fills the chip
    static const char* env_tile = getenv("STEGAN_BF16_TILE");          // "MN", e.g. "22", "11" (experiments)
    int TM = (p.M_g >= 112 && round_up(p.M_g, 128) - p.M_g <= p.M_g / 6) ? 2 : 1;
    int TN = 2;
    {
        const long tiles2 = ceil_div(p.M_g, 64 * TM) * (long)d->groups * phases * ((N + 127) / 128);
        if (tiles2 < 512) TN = 1;
        if (TN == 1 && TM == 2) {
            const long tiles1 = ceil_div(p.M_g, 128) * (long)d->groups * phases * ((N + 63) / 64);
            if (tiles1 < 256) TM = 1;
        }
    }
    if (env_tile && env_tile[0] && env_tile[1]) { TM = env_tile[0] - '0'; TN = env_tile[1] - '0'; }
    GeoH gm = plan_bf16(TM, TN, p.C2_g, ntaps_max, span_off, step_max, cs, U, U_min);
    if (!gm.ok && TN == 2) { TN = 1; gm = plan_bf16(TM, TN, p.C2_g, ntaps_max, span_off, step_max, cs, U, U_min); }
    if (!gm.ok && TM == 2) { TM = 1; gm = plan_bf16(TM, TN, p.C2_g, ntaps_max, span_off, step_max, cs, U, U_min); }
    if (!gm.ok) return 1;
    const int BM = 64 * TM, BN = 64 * TN, NCB = 1 << gm.logNCB, CCH = 8 * NCB;
    p.CC = CCH; p.logCC = gm.logNCB + 3; p.SEG = gm.SEG; p.NSEG = gm.NSEG; p.XROW = gm.XROW; p.WR = 0; p.U0 = U;
    p.TG = ntaps_max; p.NTG = 1; p.fdNTG = make_fastdiv(1);
    p.fdXROW = make_fastdiv((uint32_t)gm.XROW); p.fdSEG = make_fastdiv((uint32_t)gm.SEG);
    p.fdU0 = make_fastdiv((uint32_t)U); p.fdU1 = make_fastdiv((uint32_t)(U > 1 ? U - 1 : 1));
    h.XROWP = gm.XROWP; h.XP = gm.XP; h.ntapsW = gm.ntW; h.xit = gm.xit; h.wit = gm.wit;
    h.fdXROWP = make_fastdiv((uint32_t)gm.XROWP);
    h.src_bytes = (unsigned)((int64_t)d->B * p.Csrc * p.Tsrc * 4);
    const long nb = (N + BN - 1) / BN;
    const int nchunks = ceil_div(p.C2_g, CCH);
    p.m_tiles = ceil_div(p.M_g, BM);
    p.tile_m = BM; p.tile_n = BN;
    const long base = nb * p.m_tiles * d->groups * phases;
    // split-K: a chunk of the bf16 kernel is short (16x the fp32 MFMA rate), so the fixed per-block cost weighs more
    double cost1 = 0;
    int splits = choose_splits(base, nchunks, 0.5, &cost1);
    const int64_t out_elems = (int64_t)d->B * p.Cdst * p.Tdst;
    if (splits > 1 && (!workspace || workspace_floats < out_elems * splits)) {
        splits = workspace ? (int)(workspace_floats / out_elems) : 1;
        if (splits < 2) splits = 1;
    }
    p.splits = splits;
    p.chunks_per_split = ceil_div(nchunks, splits);
    p.splits = ceil_div(nchunks, p.chunks_per_split);
    p.partial = workspace;
    p.mix_S = 0; p.mix_full = 0; p.mix_cps = 0; p.nbx = (int)nb; p.ny = p.m_tiles * d->groups;
    p.stamps = nullptr;
    p.fdMT = make_fastdiv((uint32_t)p.m_tiles); p.fdSP = make_fastdiv((uint32_t)p.splits);
    p.fdNBX = make_fastdiv((uint32_t)p.nbx); p.fdNY = make_fastdiv((uint32_t)p.ny);
    p.fdMS = make_fastdiv(1);
    for (int ph = 0; ph < MAX_PHASES; ++ph) {
        const Taps t = phase_taps_host(p, ph < phases ? ph : 0);
        p.ph_ntaps[ph] = t.ntaps; p.ph_kbase[ph] = t.kbase; p.ph_kstep[ph] = t.kstep; p.ph_offmin[ph] = t.off_min;
        p.ph_step[ph] = t.step;
        p.ph_U[ph] = mode == 0 ? U : (p.Tdst - ph + d->stride - 1) / d->stride;
    }
    {
        static const char* dbg = getenv("STEGAN_PLAN_DEBUG");
        if (dbg) fprintf(stderr, "plan-bf16 mode=%d B=%d C2=%d M=%d K=%d s=%d d=%d T=%d | BM=%d BN=%d CCH=%d nchunks=%d tiles=%ld splits=%d lds=%zu ntW=%d XROW=%d xit=%d wit=%d\n",
                         mode, d->B, p.C2_g, p.M_g, d->K, d->stride, d->dilation, p.Tdst, BM, BN, CCH, nchunks, base, p.splits,
                         gm.lds, gm.ntW, gm.XROW, gm.xit, gm.wit);
    }
    static bool attr_done = false;
    if (!attr_done) {
        set_lds_attr<1, 1>(); set_lds_attr<1, 2>(); set_lds_attr<2, 1>(); set_lds_attr<2, 2>();
        attr_done = true;
    }
    dim3 grid((unsigned)nb, (unsigned)(p.m_tiles * d->groups), (unsigned)(phases * p.splits));
    if (TM == 2 && TN == 2) launch_variant<2, 2>(gm.logNCB, grid, gm.lds, s, h);
    else if (TM == 2) launch_variant<2, 1>(gm.logNCB, grid, gm.lds, s, h);
    else if (TN == 2) launch_variant<1, 2>(gm.logNCB, grid, gm.lds, s, h);
    else launch_variant<1, 1>(gm.logNCB, grid, gm.lds, s, h);
    if (p.splits > 1) {
        long eb = (((p.Tdst & 3) == 0 ? out_elems / 4 : out_elems) + 255) / 256;
        if (eb > 4096) eb = 4096;
        hipLaunchKernelGGL(conv_splitk_epilogue_kernel, dim3((unsigned)eb), dim3(256), 0, s, p);
    }
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

}  // namespace

extern "C" int64_t stegan_conv_prep_bf16_elems(const stegan_conv_desc* d, int32_t which) {
    if (!d || d->groups <= 0 || d->K <= 0) return -1;
    PrepH t;
    fill_prep(t, d);
    return which == 0 ? t.nf : t.nb;
}

extern "C" int stegan_conv_prep_bf16(const stegan_conv_desc* d, const float* v, const float* scale, uint16_t* wf,
                                     uint16_t* wb, void* stream) {
    STEGAN_CHECK_ARG(d && v && wf && wb && d->groups > 0 && d->K > 0 && d->Cin % d->groups == 0 && d->Cout % d->groups == 0);
    PrepH t;
    fill_prep(t, d);
    t.v = v; t.scale = scale; t.wf = wf; t.wb = wb;
    long blocks = (t.nf + t.nb + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(conv_prep_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_conv_prep_bf16_multi(int32_t n, const void* table_dev, int32_t blocks_per_entry, void* stream) {
    STEGAN_CHECK_ARG(n > 0 && table_dev && blocks_per_entry > 0);
    hipLaunchKernelGGL(conv_prep_bf16_multi_kernel, dim3((unsigned)blocks_per_entry, (unsigned)n), dim3(256), 0,
                       (hipStream_t)stream, (const PrepH*)table_dev);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_conv1d_fwd_h(const stegan_conv_desc* d, const float* x, const uint16_t* wfh, const float* scale,
                                   const float* bias, const float* residual, float* y, float* workspace,
                                   int64_t workspace_floats, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    STEGAN_CHECK_ARG(x && wfh && y);
    STEGAN_CHECK_ARG(d->in_act == STEGAN_ACT_NONE || d->in_act == STEGAN_ACT_RELU);
    STEGAN_CHECK_ARG(d->out_act >= STEGAN_ACT_NONE && d->out_act <= STEGAN_ACT_TANH);
    STEGAN_CHECK_ARG(!(d->out_btc && residual));
    HP h;
    DirP& p = h.d;
    p.src = x; p.wt = nullptr; p.scale = scale; p.bias = bias; p.res = residual; p.mask = nullptr; p.out = y;
    p.Csrc = d->Cin; p.Cdst = d->Cout; p.Tsrc = d->Tin; p.Tq = d->Tin * d->upsample; p.Tdst = d->Tout;
    p.Tmask = 0; p.ups = d->upsample == 2 ? 1 : 0;
    p.C2_g = d->Cin / d->groups; p.M_g = d->Cout / d->groups; p.LD = 0; p.zero_off = 0;
    p.in_relu = d->in_act == STEGAN_ACT_RELU; p.out_act = d->out_act; p.out_btc = d->out_btc;
    h.wh = wfh; h.Mpad = round_up(p.M_g, 128); h.Cpad = round_up(p.C2_g, 64);
    return launch_bf16(d, 0, h, workspace, workspace_floats, (hipStream_t)stream);
}

extern "C" int stegan_conv1d_bwd_data_h(const stegan_conv_desc* d, const float* dy, const uint16_t* wbh, const float* scale,
                                        const float* mask_src, const float* add_src, float* dx, float* workspace,
                                        int64_t workspace_floats, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    STEGAN_CHECK_ARG(dy && wbh && dx);
    STEGAN_CHECK_ARG(d->in_act != STEGAN_ACT_RELU || mask_src);
    HP h;
    DirP& p = h.d;
    p.src = dy; p.wt = nullptr; p.scale = scale; p.bias = nullptr; p.res = add_src; p.out = dx;
    p.mask = d->in_act == STEGAN_ACT_RELU ? mask_src : nullptr;
    p.Csrc = d->Cout; p.Cdst = d->Cin; p.Tsrc = d->Tout; p.Tq = d->Tout; p.Tdst = d->Tin * d->upsample;
    p.Tmask = d->Tin; p.ups = d->upsample == 2 ? 1 : 0;
    p.C2_g = d->Cout / d->groups; p.M_g = d->Cin / d->groups; p.LD = 0; p.zero_off = 0;
    p.in_relu = 0; p.out_act = 0; p.out_btc = 0;
    h.wh = wbh; h.Mpad = round_up(p.M_g, 128); h.Cpad = round_up(p.C2_g, 64);
    return launch_bf16(d, 1, h, workspace, workspace_floats, (hipStream_t)stream);
}
