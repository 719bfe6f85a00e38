// One tile per CU: forward / input-gradient convolution of the stride-1, ungrouped layers (round 5).
//
// Same computation and epilogue as conv_bf16.hip,
//
//     out[b, m, t] = sum_{tap, c}  W[tap][m][c] * src[b, c, (t + off[tap]) >> ups]
//
// but a different launch geometry.  The round-4 profile showed what bounds those launches: 144-576 blocks of 64 x 64 or
// 64 x 128 outputs, i.e. half a chip of blocks that each stream the whole reduction depth through their CU's fetch path, or
// 1.1-2.2 blocks per CU (a second, mostly idle round), behind ~5 us of per-launch set-up.  Here the host cuts every layer
// into AT MOST 256 EQUAL tiles -- one per CU, 48 / 96 / 192 rows or columns, chosen per layer so that the bytes a CU must
// fetch, (Mc + Nc) x K x 2, are smallest -- and the kernel is a lean ring:
//
//   * operands: the activation image [B][C / 8][T][8] and the weight copy [k][C / 8][Mpad][8] of stegan_conv_prep_bf16, both
//     staged by LDS-DMA (buffer_load ... lds, 16 B per lane, 1 KiB per wave-instruction) into [cb][slot] / [tap][cb][m]
//     stages of 16-byte pieces: an MFMA operand is one ds_read_b128, conflict-free (plane strides are multiples of 256 B);
//     padding, halos, ragged channel blocks and rows past the tensor carry an out-of-range buffer offset and land as zeros;
//   * every wave stages AND multiplies (no producer / consumer split: an LDS-DMA item is four instructions, and with one
//     tile per CU there is no second block to overlap with); NS-stage ring, counted vmcnt, ONE barrier per chunk;
//   * v_mfma_f32_16x16x32 (bf16 | f16): a wave owns 48 x 48 outputs = 3 x 3 MFMA tiles (6 fragment reads feed 9 MFMAs), the
//     activation fragment is the A operand, so a lane ends up with ONE output channel and four runs of 4 consecutive time
//     steps: 16-byte residual / mask loads and output stores;
//   * small layers (48-wide tiles: T = 12 ... 48) split the reduction over the 2 or 4 waves of a tile position inside each
//     chunk; the partial accumulators meet in LDS in a fixed order (bit-reproducible);
//   * block -> tile placement: each XCD takes a gm x gn rectangle of the tile grid (host: fewest bytes through the fabric).
//
// Numerics: products of operands rounded to the 2-byte type, fp32 accumulation; everything behind the sum in fp32, exactly as
// conv_bf16.hip (the summation ORDER over channel chunks differs from that kernel's: last-bit differences).
#include "conv_tile.h"

#include <stdio.h>
#include <stdlib.h>

namespace {

typedef hop_t hx8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef hop_t hx2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#ifdef STEGAN_OPERAND_F16
#define TILE_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x32_f16(a_, b_, c_, 0, 0, 0)
#else
#define TILE_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, b_, c_, 0, 0, 0)
#endif

constexpr unsigned T_OOB = 0x7ffff000u;      // buffer offset beyond any operand (host: operand bytes < 2^31 - 2^13): reads 0
constexpr int TW_MAX = 12;                   // W DMA items per producer wave and chunk
constexpr int TX_MAX = 8;                    // X DMA items per producer wave and chunk
constexpr int WT = 48;                       // wave tile: 48 x 48 outputs

struct TK {
    const unsigned short* xh;
    const unsigned short* wh;
    const float* scale;
    const float* bias;
    const float* res;
    const float* mask;
    float* out;
    unsigned short* out_h;
    unsigned xh_bytes, wh_bytes, out_bytes, mask_bytes, outh_bytes;
    int B, Tsrc, Tq, Tdst, Tmask, ups, xups, mode;   // ups: forward = nearest-x2 on the source read (xups), input gradient = the mask's fold
    int CBx;                      // channel blocks of the source image per batch element
    int Cdst, out_cb, M;
    int Mpad, CB;                 // weight copy: rows per (tap, channel block) plane, channel blocks per tap
    int ntaps, kbase, kstep, off_min, step;
    int in_relu, out_act, mask_lrelu;
    int Mc, Nc, WM, WN, WK, lgWN, lgWK, ncons, nthreads;
    int NCB, nch, KS, lgKS, S;    // channel blocks per chunk, chunks, k-steps per tap and per chunk (32 channels each)
    int Tt, nseg, SEGW, XROW, NTS;   // columns per segment, segments per tile, slots per segment, slots per plane (x 64), tiles per sample
    int ipt, xseg, nW, nX;        // W items (64 pieces) per tap, X items per channel block, items per chunk
    int w_adv, plane_k, chunk_w;  // bytes: per 3-item period of a tap's planes, per tap of the weight copy, per chunk
    int NS, wstage, stage, rot;   // ring stages, bytes of the W part / of a whole stage, rotated chunk walk
    int MT, NTl, lgm, mper;       // tile grid; XCD rectangles: 2^lgm parts along M of mper tiles each (lgm < 0: identity)
    FastDiv fdMc, fdSEGW, fdTt, fdNTS, fdMper, fdMT, fdIpt, fdXseg;
#ifdef STEGAN_STAMPS
    unsigned long long* stamps;   // diagnostic builds: 16 time stamps per block (scripts/tile_stamps.py)
#endif
};

__device__ __forceinline__ void tile_wait_vmcnt(int n) {
#define TV_CASE(k_) case k_: asm volatile("s_waitcnt vmcnt(" #k_ ")" ::: "memory"); break;
    switch (n) {
        TV_CASE(0) TV_CASE(1) TV_CASE(2) TV_CASE(3) TV_CASE(4) TV_CASE(5) TV_CASE(6) TV_CASE(7) TV_CASE(8) TV_CASE(9)
        TV_CASE(10) TV_CASE(11) TV_CASE(12) TV_CASE(13) TV_CASE(14) TV_CASE(15) TV_CASE(16) TV_CASE(17) TV_CASE(18)
        TV_CASE(19) TV_CASE(20) TV_CASE(21) TV_CASE(22) TV_CASE(23) TV_CASE(24) TV_CASE(25) TV_CASE(26) TV_CASE(27)
        TV_CASE(28) TV_CASE(29) TV_CASE(30) TV_CASE(31) TV_CASE(32) TV_CASE(33) TV_CASE(34) TV_CASE(35) TV_CASE(36)
        TV_CASE(37) TV_CASE(38) TV_CASE(39) TV_CASE(40) TV_CASE(41) TV_CASE(42) TV_CASE(43) TV_CASE(44) TV_CASE(45)
        TV_CASE(46) TV_CASE(47) TV_CASE(48)
        default: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;     // (waiting for fewer is always safe)
    }
#undef TV_CASE
}

__device__ __forceinline__ unsigned sdiv(unsigned n, const FastDiv& f) {     // scalar operands: s_mul_hi_u32
    return (unsigned)((((unsigned long long)n * f.mul) >> 32) + n) >> f.shift;
}

#ifdef STEGAN_TILE_PAIR
// diagnostic build (scripts/build_tile_pair.sh): blocks of at most 8 waves, 128 registers and 78 KB of LDS, so that two blocks --
// of two launches on two streams -- are resident on a CU.  Measures what co-residency would buy.
__global__ __launch_bounds__(512, 4) void conv_tile_kernel(const TK k) {
#else
__global__ __launch_bounds__(768) void conv_tile_kernel(const TK k) {
#endif
    extern __shared__ __align__(16) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= k.ncons;
#ifdef STEGAN_STAMPS
#define TSTAMP(i_) if (k.stamps && threadIdx.x == 0) k.stamps[(size_t)blockIdx.x * 16 + (i_)] = __builtin_amdgcn_s_memtime();
#define PSTAMP(i_) if (k.stamps && (int)threadIdx.x == k.ncons * 64) k.stamps[(size_t)blockIdx.x * 16 + (i_)] = __builtin_amdgcn_s_memtime();
    if (k.stamps && threadIdx.x == 0) k.stamps[(size_t)blockIdx.x * 16 + 8] = __builtin_amdgcn_s_memrealtime();
#else
#define TSTAMP(i_)
#define PSTAMP(i_)
#endif
    TSTAMP(0)

    // ---- block -> tile.  Blocks are dealt round-robin over the XCDs; XCD x takes rectangle (x & (gm - 1), x >> lgm) of the
    // tile grid, M tiles fastest inside it (placement is speed only: any mapping gives the same results)
    int mt, nt;
    {
        const unsigned bid = blockIdx.x;
        if (k.lgm >= 0) {
            const unsigned xcd = bid & 7u, local = bid >> 3;
            const unsigned xi = xcd & ((1u << k.lgm) - 1u), xj = xcd >> k.lgm;
            const unsigned ln = sdiv(local, k.fdMper), lm = local - ln * (unsigned)k.mper;
            const unsigned nper = (unsigned)(k.NTl >> (3 - k.lgm));
            mt = (int)(xi * (unsigned)k.mper + lm);
            nt = (int)(xj * nper + ln);
        } else {
            const unsigned q = sdiv(bid, k.fdMT);
            mt = (int)(bid - q * (unsigned)k.MT);
            nt = (int)q;
        }
    }
    const int m0 = mt * k.Mc;
    const int smp = (int)sdiv((unsigned)nt, k.fdNTS);                 // (group of) sample(s) of this tile
    const int b0 = smp * k.nseg, t0 = (nt - smp * k.NTS) * k.Nc;
    const int nch = k.nch, NS = k.NS;

    if (producer) {
        // ================= producers (4 waves): LDS-DMA of chunk j + NS - 1 while the consumers multiply chunk j =============
        // An item = one wave-instruction = 64 pieces of 16 bytes landing as 1 KiB of the stage.  The per-lane part of every
        // source offset is one of THREE patterns (W: the 64 x 3 pieces of a period are 192 / Mc whole planes of Mc rows) or of
        // the row's <= 4 segments of 64 slots (X); tap, period, channel block and chunk are a SCALAR offset: an item costs a
        // handful of scalar instructions, two selects and the DMA itself.
        const int pw = wave - k.ncons;
        // The ring's loop invariants are pinned in scalar registers behind an empty asm: left as kernel-argument fields, hipcc
        // -- out of scalar registers with the whole argument struct in flight -- RE-LOADED the descriptor's base from the
        // kernarg segment in front of EVERY DMA (s_load_dwordx16 + s_waitcnt lgkmcnt(0): a ~200-cycle stall per item, which
        // is what bounded the first versions of this kernel: 175 cycles per item whatever it moved).  An asm result cannot be
        // rematerialised; under pressure it is spilled to a vector lane (v_readlane: one instruction, no wait).
        unsigned long long wbase = (unsigned long long)k.wh, xbase = (unsigned long long)k.xh;
        int wbytes = (int)k.wh_bytes, xbytes = (int)k.xh_bytes;
        asm volatile("" : "+s"(wbase), "+s"(xbase), "+s"(wbytes), "+s"(xbytes));
        const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, wbytes, 0x00020000);
        const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, xbytes, 0x00020000);
        unsigned vw[3], vx[4];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const unsigned q = (unsigned)(r * 64 + lane);
            const unsigned cb = fd_div(q, k.fdMc), m = q - cb * (unsigned)k.Mc;
            vw[r] = (m0 + (int)m < k.Mpad) ? (unsigned)(((int)cb * k.Mpad + m0 + (int)m) * 16) : T_OOB;
        }
        const int xlive = k.nseg * k.SEGW;                            // slots of the plane that exist (the rest is padding)
#pragma unroll
        for (int sgi = 0; sgi < 4; ++sgi) {
            const unsigned slot = (unsigned)(sgi * 64 + lane);
            const unsigned g = fd_div(slot, k.fdSEGW), sl = slot - g * (unsigned)k.SEGW;
            const int b = b0 + (int)g, q = t0 + k.off_min + (int)sl;
            const bool ok = ((int)slot < xlive) & (b < k.B) & (q >= 0) & (q < k.Tq);
            vx[sgi] = ok ? (unsigned)((b * k.CBx * k.Tsrc + (q >> k.xups)) * 16) : T_OOB;
        }
        // this wave's items: i = pw + 4 n.  Everything that does not depend on the chunk is computed ONCE -- the per-lane
        // offset (a copy of its pattern), the scalar offset inside the chunk -- into register arrays indexed by compile-time
        // constants: an item of the ring is then s_add (chunk) + M0 + the DMA (a lone wave issues one instruction per ~4-5
        // cycles: the first version recomputed tap / period / channel block per item, ~25 scalar instructions = 120 of the
        // 175 cycles an item cost)
        const int nW = k.nW, nX = k.nX;
        const int nWw = nW > pw ? (nW - pw + 3) >> 2 : 0, nXw = nX > pw ? (nX - pw + 3) >> 2 : 0;
        const int ipc = nWw + nXw;                                    // this wave's DMA instructions per chunk
        unsigned wv[TW_MAX], xv[TX_MAX];                              // per-lane offsets WITH the item's scalar part folded in
#pragma unroll
        for (int n = 0; n < TW_MAX; ++n) {
            const unsigned i = (unsigned)(pw + 4 * n);
            const unsigned tap = sdiv(i, k.fdIpt), r = i - tap * (unsigned)k.ipt;
            const unsigned per = (r * 171u) >> 9, ph = r - 3u * per;              // r / 3, r % 3 (r < 256)
            const unsigned so = (unsigned)((k.kbase + (int)tap * k.kstep) * k.plane_k + (int)per * k.w_adv);
            const unsigned v = ph == 0 ? vw[0] : (ph == 1 ? vw[1] : vw[2]);
            wv[n] = v == T_OOB ? T_OOB : v + so;                      // (an out-of-range lane stays out of range)
        }
        const int row_x = k.Tsrc * 16;
#pragma unroll
        for (int n = 0; n < TX_MAX; ++n) {
            const unsigned i = (unsigned)(pw + 4 * n);
            const unsigned cb = sdiv(i, k.fdXseg), sg = i - cb * (unsigned)k.xseg;
            const unsigned v = sg == 0 ? vx[0] : (sg == 1 ? vx[1] : (sg == 2 ? vx[2] : vx[3]));
            xv[n] = v == T_OOB ? T_OOB : v + cb * (unsigned)row_x;
        }
        int chunk_x = k.NCB * row_x, chunk_w = k.chunk_w, stage_b = k.stage, wstage_b = k.wstage, ncb = k.NCB, cbx = k.CBx;
        int nWi = nWw, nXi = nXw;
        asm volatile("" : "+s"(chunk_x), "+s"(chunk_w), "+s"(stage_b), "+s"(wstage_b), "+s"(ncb), "+s"(cbx));
        PSTAMP(6)
#define TL_W_IT(n_) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(Wd_ + (n_) * 4096), 16, (int)wv[n_], cw_, 0, 0);
#define TL_X_IT(n_, XO_) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(Xd_ + (n_) * 4096), 16, XO_(n_), cx_, 0, 0);
#define TL_XO_FULL(n_) (int)xv[n_]
#define TL_XO_RAG(n_) ((int)sdiv((unsigned)(pw + 4 * (n_)), k.fdXseg) < xlive_ ? (int)xv[n_] : (int)T_OOB)
#define TL_X_SW(XO_)                                                                                            \
            if (nXw > 0) { TL_X_IT(0, XO_) if (nXw > 1) { TL_X_IT(1, XO_) if (nXw > 2) { TL_X_IT(2, XO_)                \
            if (nXw > 3) { TL_X_IT(3, XO_) if (nXw > 4) { TL_X_IT(4, XO_) if (nXw > 5) { TL_X_IT(5, XO_)                \
            if (nXw > 6) { TL_X_IT(6, XO_) if (nXw > 7) { TL_X_IT(7, XO_) } } } } } } } }
#define TL_ISSUE(c_, st_)                                                                                       \
        {                                                                                                       \
            char* Wd_ = smem + (st_) * stage_b + pw * 1024;                                                     \
            char* Xd_ = Wd_ + wstage_b;                                                                         \
            const int cw_ = (c_) * chunk_w, cx_ = (c_) * chunk_x;                                               \
            const int xlive_ = cbx - (c_) * ncb;              /* channel blocks of the image left for this chunk */ \
            /* (opaque per issue: the item guards stay ONE compare + branch on a live scalar instead of a table of     \
               hoisted mask registers that get spilled) */                                                      \
            int nWw = nWi, nXw = nXi;                                                                           \
            asm volatile("" : "+s"(nWw), "+s"(nXw));                                                            \
            if (nWw > 0) { TL_W_IT(0) if (nWw > 1) { TL_W_IT(1) if (nWw > 2) { TL_W_IT(2) if (nWw > 3) { TL_W_IT(3)     \
            if (nWw > 4) { TL_W_IT(4) if (nWw > 5) { TL_W_IT(5) if (nWw > 6) { TL_W_IT(6) if (nWw > 7) { TL_W_IT(7)     \
            if (nWw > 8) { TL_W_IT(8) if (nWw > 9) { TL_W_IT(9) if (nWw > 10) { TL_W_IT(10) if (nWw > 11) { TL_W_IT(11) \
            } } } } } } } } } } } }                                                                             \
            if (xlive_ >= ncb) { TL_X_SW(TL_XO_FULL) }                                                        \
            else { TL_X_SW(TL_XO_RAG) }               /* ragged last chunk: blocks past the image read zeros */  \
        }
        static_assert(TW_MAX == 12 && TX_MAX == 8, "the item switches are written out for 12 / 8 items");
        // the tiles of an XCD that share a W or X tile walk the chunks from different starting points: at any moment they ask
        // for DIFFERENT lines, so all but the first request for a line hit in the XCD's L2 (same terms, rotated order)
        const int rot = (k.rot && nch > 1) ? (int)((unsigned)(mt + nt) % (unsigned)nch) : 0;
#define TL_CID(j_) ((j_) + rot - (((j_) + rot) >= nch ? nch : 0))
        // prologue: chunk 0 alone -- the consumers start as soon as it has landed; the ring is filled behind barrier 0
        if (nch > 0) TL_ISSUE(TL_CID(0), 0)
        PSTAMP(7)
        int st_in = 1 % NS, issued = 1;
        for (int j = 0; j < nch; ++j) {
            tile_wait_vmcnt((issued - 1 - j) * ipc);                  // chunk j has landed; the younger ones stay in flight
            __builtin_amdgcn_s_barrier();                             // chunk j is visible; the stage of chunk j - 1 is free
            asm volatile("" ::: "memory");
            while (issued < nch && issued < j + NS) {                 // (iteration 0 issues chunks 1 .. NS - 1)
                TL_ISSUE(TL_CID(issued), st_in)
                st_in = st_in + 1 == NS ? 0 : st_in + 1;
                ++issued;
            }
        }
#undef TL_W_IT
#undef TL_X_IT
#undef TL_XO_FULL
#undef TL_XO_RAG
#undef TL_X_SW
#undef TL_ISSUE
#undef TL_CID
        PSTAMP(14)
        return;                                                       // (barriers behind this point count the surviving waves)
    }

    // ================= consumers: operand reads + MFMAs, epilogue =================
    // wave -> (tile position, reduction part)
    const int wk = wave & (k.WK - 1), sp = wave >> k.lgWK;
    const int wn = sp & (k.WN - 1), wm = sp >> k.lgWN;
    const int wm0 = wm * WT, wn0 = wn * WT;
    const int l15 = lane & 15, kg = lane >> 4;
    const int mode = k.mode, M = k.M;

    // per-channel epilogue factors of this lane's three output channels: requested first, used last
    float sc[3], bi[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int m = m0 + wm0 + 16 * j + l15;
        const int ch = m < M ? m : M - 1;
        sc[j] = k.scale ? (mode == 0 ? k.scale[ch] : k.scale[0]) : 1.f;
        bi[j] = (mode == 0 && k.bias) ? k.bias[ch] : 0.f;
    }

    // ---- operand addresses.  A = activations: row r = lane & 15 is a column (time step) of the tile, lane >> 4 picks the
    // 8-channel block inside the 32-channel k-step; B = weights: column = output channel
    int a_off[3], b_off[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const unsigned nl = (unsigned)(wn0 + 16 * i + l15);
        const unsigned g = fd_div(nl, k.fdTt), u = nl - g * (unsigned)k.Tt;
        a_off[i] = (kg * k.XROW + (int)g * k.SEGW + (int)u) * 16;
        b_off[i] = (kg * k.Mc + wm0 + 16 * i + l15) * 16;
    }
    const int stepA = 4 * k.XROW * 16, stepB = 4 * k.Mc * 16, tapA = k.step * 16;
    const int KS = k.KS, lgKS = k.lgKS, S = k.S, WK = k.WK;

    f32x4 acc[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#define TL_LDS(ptr_) (*(const hx8*)(ptr_))
#define TL_LOAD(a_, b_, s_)                                                                                     \
    {                                                                                                           \
        const int tap_ = (s_) >> lgKS, ks_ = (s_) & (KS - 1);                                                   \
        const char* xa_ = Xcur + ks_ * stepA + tap_ * tapA;                                                     \
        const char* wb_ = Wcur + (s_) * stepB;                                                                  \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) a_[i] = TL_LDS(xa_ + a_off[i]);                           \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) b_[i] = TL_LDS(wb_ + b_off[i]);                           \
    }
    // (RELU_: the input ReLU on the packed halves of the activation fragment -- sign bit set <=> negative int16 -- 12 extra
    // vector instructions per k-step, so the layers without one run a loop without them)
#define TL_MFMA(a_, b_, RELU_)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                             \
        const hx8 ai_ = RELU_ ? __builtin_bit_cast(hx8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, a_[i]), (s16x8)(0))) : a_[i]; \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) acc[i][j] = TILE_MFMA(ai_, b_[j], acc[i][j]);             \
    }
#define TL_RING(RELU_)                                                                                          \
    for (int j = 0; j < nch; ++j) {                                                                             \
        if (j == 2) { TSTAMP(11) }                                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
        __builtin_amdgcn_s_barrier();                 /* (the producers' side: counted vmcnt before it) */      \
        asm volatile("" ::: "memory");                                                                          \
        if (j == 0) { TSTAMP(2) }                                                                               \
        if (j == 2) { TSTAMP(12) }                                                                              \
        {                                                                                                       \
            const char* Wcur = smem + st_c * k.stage;                                                           \
            const char* Xcur = Wcur + k.wstage;                                                                 \
            hx8 aA[3], bA[3], aB[3], bB[3];                                                                     \
            int s = wk;                                                                                         \
            if (s < S) TL_LOAD(aA, bA, s)                                                                       \
            for (; s + WK < S; s += 2 * WK) {                                                                   \
                TL_LOAD(aB, bB, s + WK)                                                                         \
                TL_MFMA(aA, bA, RELU_)                                                                          \
                if (s + 2 * WK < S) TL_LOAD(aA, bA, s + 2 * WK)                                                 \
                TL_MFMA(aB, bB, RELU_)                                                                          \
            }                                                                                                   \
            if (s < S) { TL_MFMA(aA, bA, RELU_) }                                                               \
        }                                                                                                       \
        if (j == 2) { TSTAMP(13) }                                                                              \
        st_c = st_c + 1 == NS ? 0 : st_c + 1;                                                                   \
    }
    TSTAMP(1)
    int st_c = 0;
    if (k.in_relu) { TL_RING(true) } else { TL_RING(false) }
#undef TL_RING
#undef TL_MFMA
#undef TL_LOAD
#undef TL_LDS
    TSTAMP(3)

    // ---- epilogue.  A 16 x 16 MFMA tile holds, per lane, output channel (lane & 15) and the 4 consecutive time steps
    // 4 * (lane >> 4) + 0..3 of its 16 columns: 16-byte accesses on the fp32 side.  ONE wave per SIMD is left by now and issues
    // an instruction every ~4-5 cycles, so what this part costs is its instruction COUNT (the first version: ~1 500 executed
    // instructions, 6-7 k cycles -- as much as the main loop of a T = 12 layer): every access goes through a buffer
    // descriptor with a 32-bit offset (rows / columns / samples past the tensor carry bit 31: dropped or zero, no branches,
    // no 64-bit index arithmetic); the image (8 CHANNELS of one column per 16-byte piece) is transposed by LDS in one batch
    // per wave: one ds_write_b64 per tile into a [channel][column] scratch, read back with the transposing ds_read_b64_tr_b16.
    const int ncw = k.ncons;
    const float* const resp = k.res;
    const float* const maskp = mode == 1 ? k.mask : nullptr;
    const bool mlr = k.mask_lrelu != 0;
    const int act = k.out_act, Cdst = k.Cdst, Tdst = k.Tdst, ups = k.ups;
    unsigned short* const outh = k.out_h;
    const auto rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)k.out, 0, (int)k.out_bytes, 0x00020000);
    const auto rs_r = __builtin_amdgcn_make_buffer_rsrc((void*)k.res, 0, resp ? (int)k.out_bytes : 0, 0x00020000);
    const auto rs_m = __builtin_amdgcn_make_buffer_rsrc((void*)k.mask, 0, maskp ? (int)k.mask_bytes : 0, 0x00020000);
    const auto rs_h = __builtin_amdgcn_make_buffer_rsrc((void*)k.out_h, 0, outh ? (int)k.outh_bytes : 0, 0x00020000);
    constexpr unsigned BAD = 0x80000000u;         // (every tensor of the launch is < 2 GiB: host)

    // byte offsets of the lane's 4 columns of row group i (fp32 output / residual; mask), of its image column, and of its
    // three channels; bit 31 marks "outside"
    const int Tm = ups ? k.Tmask : Tdst;
    unsigned colf[3], colm[3], colh[3], chf[3], chm[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const unsigned nl = (unsigned)(wn0 + 16 * i + 4 * kg);
        const unsigned g = fd_div(nl, k.fdTt), u = nl - g * (unsigned)k.Tt;
        const int b = b0 + (int)g, t = t0 + (int)u;
        const bool ok = ((int)g < k.nseg) & (b < k.B) & (t < Tdst);
        colf[i] = ok ? (unsigned)((b * Cdst * Tdst + t) * 4) : BAD;
        colm[i] = ok ? (unsigned)((b * Cdst * Tm + (t >> ups)) * 4) : BAD;
        const unsigned nl2 = (unsigned)(wn0 + 16 * i + l15);
        const unsigned g2 = fd_div(nl2, k.fdTt), u2 = nl2 - g2 * (unsigned)k.Tt;
        const int b2 = b0 + (int)g2, t2 = t0 + (int)u2;
        const bool ok2 = ((int)g2 < k.nseg) & (b2 < k.B) & (t2 < Tdst);
        colh[i] = ok2 ? (unsigned)((b2 * k.out_cb * Tdst + t2) * 16) : BAD;
        const int m = m0 + wm0 + 16 * i + l15;                       // (i doubles as the channel group j here)
        chf[i] = m < M ? (unsigned)(m * Tdst * 4) : BAD;
        chm[i] = m < M ? (unsigned)(m * Tm * 4) : BAD;
    }
#define TL_OFF(a_, b_) ((((a_) | (b_)) & BAD) | ((a_) + (b_)))       /* sum of two offsets, bit 31 if either is outside */
    // the fused epilogue on one tile's 4 values: mask -> scale, bias -> + residual -> leaky mask -> activation
#define TL_FUSE(x_, mk_, rs_, sc_, bi_)                                                                         \
    {                                                                                                           \
        if (maskp && !mlr) { _Pragma("unroll") for (int e = 0; e < 4; ++e) x_[e] = mk_[e] > 0.f ? x_[e] : 0.f; }        \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) x_[e] = x_[e] * (sc_) + (bi_);                            \
        if (resp) { _Pragma("unroll") for (int e = 0; e < 4; ++e) x_[e] += rs_[e]; }                            \
        if (maskp && mlr) { _Pragma("unroll") for (int e = 0; e < 4; ++e) x_[e] *= mk_[e] > 0.f ? 1.f : 0.1f; } \
        if (act != STEGAN_ACT_NONE) {                                                                           \
            if (act == STEGAN_ACT_LRELU) { _Pragma("unroll") for (int e = 0; e < 4; ++e) x_[e] = x_[e] > 0.f ? x_[e] : 0.1f * x_[e]; } \
            else { _Pragma("unroll") for (int e = 0; e < 4; ++e) x_[e] = fmaxf(x_[e], 0.f); }       /* (host: no tanh) */  \
        }                                                                                                       \
    }
    // loads of one tile's fused inputs (the mask of an upsampled layer: two source positions per four columns)
#define TL_FETCH(mk_, rs_, of_, om_)                                                                            \
    {                                                                                                           \
        if (maskp) {                                                                                            \
            if (ups == 0) mk_ = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_m, (int)(om_), 0, 0)); \
            else {                                                                                              \
                const f32x2 m2_ = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_m, (int)(om_), 0, 0)); \
                mk_ = (f32x4){m2_[0], m2_[0], m2_[1], m2_[1]};                                                  \
            }                                                                                                   \
        }                                                                                                       \
        if (resp) rs_ = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_r, (int)(of_), 0, 0)); \
    }
#define TL_PACK2(a_, b_) __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){hop_src(a_), hop_src(b_)}, hx2))
    typedef __attribute__((address_space(3))) s16x4* tr_ptr_t;
    const int q4 = l15 >> 2, p4 = l15 & 3;

    if (WK == 1) {
        unsigned of[3][3];
        f32x4 mk[3][3], rs[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                of[i][j] = TL_OFF(colf[i], chf[j]);
                mk[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; rs[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                TL_FETCH(mk[i][j], rs[i][j], of[i][j], TL_OFF(colm[i], chm[j]))
            }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                TL_FUSE(acc[i][j], mk[i][j], rs[i][j], sc[j], bi[j])
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs_o, (int)of[i][j], 0, 0);
            }
        TSTAMP(15)
        if (outh) {
            // scratch of this wave: [64 channel rows (48 used)][48 columns] x 2 B, rows of 104 B (conflict-free 8-byte stores)
            char* const scr = smem + wave * (64 * 104);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                             // every consumer has read its last operands: LDS is free
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const u32x2 pk = {TL_PACK2(acc[i][j][0], acc[i][j][1]), TL_PACK2(acc[i][j][2], acc[i][j][3])};
                    *(u32x2*)(scr + (16 * j + l15) * 104 + (16 * i + 4 * kg) * 2) = pk;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // 6 channel blocks x 3 column groups = 18 blocks of (8 channels x 16 columns), one per 16-lane group and round:
            // lane 4 q + p of a group addresses row q (+ 4), columns 4 p .. 4 p + 3 and receives column (lane & 15), rows 0..3
            const int cb0 = (m0 + wm0) >> 3;
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const int blk = 4 * r + kg;
                const int blc = blk < 18 ? blk : 17;                  // (pad, don't mask: every lane reads a valid block)
                const int cb = (blc * 11) >> 5, tb = blc - 3 * cb;    // blk / 3, blk % 3 (blk < 32)
                const char* src = scr + (8 * cb + q4) * 104 + (16 * tb + 4 * p4) * 2;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(src));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(src + 4 * 104));
                const s16x8 pc = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const unsigned ch_ = tb == 0 ? colh[0] : (tb == 1 ? colh[1] : colh[2]);
                const bool okb = (blk < 18) & (8 * (cb0 + cb) < M);
                const unsigned oh = okb ? TL_OFF(ch_, (unsigned)((cb0 + cb) * Tdst * 16)) : BAD;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pc), rs_h, (int)oh, 0, 0);
            }
        }
    } else {
        // partial accumulators of the WK waves of a tile position meet in LDS; wave wk then finishes tiles wk, wk + WK, ...
        // Its tiles' fused inputs are requested first: their round trip runs under the exchange.
        // (WK = 2: tiles wk, wk + 2, ... -- up to 5 per wave; WK = 4: up to 3)
        constexpr int NOWN = 5;
        f32x4 mk[NOWN], rs[NOWN];
        unsigned of[NOWN], oh_col[NOWN];
        int jj[NOWN];
#pragma unroll
        for (int n = 0; n < NOWN; ++n) {
            const int tau = wk + n * WK;
            const int i = (tau * 11) >> 5, j = tau - 3 * i;           // tau / 3, tau % 3
            const unsigned cf = i == 0 ? colf[0] : (i == 1 ? colf[1] : colf[2]), cm = i == 0 ? colm[0] : (i == 1 ? colm[1] : colm[2]);
            const unsigned hf = j == 0 ? chf[0] : (j == 1 ? chf[1] : chf[2]), hm = j == 0 ? chm[0] : (j == 1 ? chm[1] : chm[2]);
            of[n] = tau < 9 ? TL_OFF(cf, hf) : BAD;
            oh_col[n] = i == 0 ? colh[0] : (i == 1 ? colh[1] : colh[2]);
            jj[n] = j;
            mk[n] = (f32x4){0.f, 0.f, 0.f, 0.f}; rs[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (tau < 9) TL_FETCH(mk[n], rs[n], of[n], TL_OFF(cm, hm))
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                 // every consumer has read its last operands: LDS is free
        asm volatile("" ::: "memory");
        // [wave][tile][lane] x 16 B: one ds_write_b128 per tile
        f32x4* const red = (f32x4*)smem;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) red[(wave * 9 + i * 3 + j) * 64 + lane] = acc[i][j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        TSTAMP(15)
        f32x4 v[NOWN];
#pragma unroll
        for (int n = 0; n < NOWN; ++n) {
            const int tau = wk + n * WK;
            v[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (tau >= 9) continue;                                   // (wave-uniform)
            for (int pw = 0; pw < WK; ++pw) {                          // fixed order: bit-reproducible
                const f32x4 t = red[((sp * WK + pw) * 9 + tau) * 64 + lane];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[n][e] += t[e];
            }
            const int j = jj[n];
            const float scj = j == 0 ? sc[0] : (j == 1 ? sc[1] : sc[2]), bij = j == 0 ? bi[0] : (j == 1 ? bi[1] : bi[2]);
            TL_FUSE(v[n], mk[n], rs[n], scj, bij)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[n]), rs_o, (int)of[n], 0, 0);
        }
        if (outh) {
            // scratch behind the exchange area: [16 rows per tile][16 columns] x 2 B, rows of 40 B
            char* const scr = smem + ncw * (9 * 64 * 16) + wave * (16 * NOWN * 40);
#pragma unroll
            for (int n = 0; n < NOWN; ++n) {
                const u32x2 pk = {TL_PACK2(v[n][0], v[n][1]), TL_PACK2(v[n][2], v[n][3])};
                *(u32x2*)(scr + (16 * n + l15) * 40 + 4 * kg * 2) = pk;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // NOWN tiles x 2 channel blocks of (8 channels x 16 columns), one per 16-lane group and round
            const int cbw = (m0 + wm0) >> 3;
#pragma unroll
            for (int r = 0; r < (2 * NOWN + 3) / 4; ++r) {
                const int blk = 4 * r + kg;
                const int blc = blk < 2 * NOWN ? blk : 2 * NOWN - 1;
                const int n = blc >> 1, cbh = blc & 1;
                const char* src = scr + (16 * n + 8 * cbh + q4) * 40 + 4 * p4 * 2;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(src));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(src + 4 * 40));
                const s16x8 pc = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                int j = jj[0];
                unsigned ch_ = oh_col[0];
#pragma unroll
                for (int t = 1; t < NOWN; ++t) { j = n == t ? jj[t] : j; ch_ = n == t ? oh_col[t] : ch_; }
                const int cb = cbw + 2 * j + cbh;
                const bool okb = (blk < 2 * NOWN) & (wk + n * WK < 9) & (8 * cb < M);
                const unsigned oh = okb ? TL_OFF(ch_, (unsigned)(cb * Tdst * 16)) : BAD;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pc), rs_h, (int)oh, 0, 0);
            }
        }
    }
#undef TL_FUSE
#undef TL_FETCH
#undef TL_PACK2
#undef TL_OFF
    TSTAMP(4)
#ifdef STEGAN_STAMPS
    if (k.stamps && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        k.stamps[(size_t)blockIdx.x * 16 + 5] = __builtin_amdgcn_s_memtime();
        k.stamps[(size_t)blockIdx.x * 16 + 9] = __builtin_amdgcn_s_memrealtime();
        unsigned xcc_;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));
        k.stamps[(size_t)blockIdx.x * 16 + 10] = xcc_;
    }
#endif
#undef TSTAMP
#undef PSTAMP
}

// ---- host side ------------------------------------------------------------------------------------------------------------

struct TileOpts {
    int enable;         // STEGAN_TILE=0 switches the kernel off (comparisons)
    int ncb1;           // STEGAN_TILE_NCB: channel blocks per chunk of the unsplit tiles (4 | 8; 0 = by stage size)
    int ns;             // STEGAN_TILE_NS: ring stages (0 = as many as fit, at most 4)
    int cfg;            // STEGAN_TILE_CFG: force one tile configuration (index + 1; 0 = cost model)
    int debug;          // STEGAN_PLAN_DEBUG
    int rot;            // STEGAN_TILE_ROT=0: every tile walks the chunks from chunk 0
};

const TileOpts& tile_opts() {
    static const TileOpts o = [] {
        TileOpts t;
        const char* e;
        t.enable = (e = getenv("STEGAN_TILE")) ? atoi(e) : 1;
        t.ncb1 = (e = getenv("STEGAN_TILE_NCB")) ? atoi(e) : 0;
        t.ns = (e = getenv("STEGAN_TILE_NS")) ? atoi(e) : 0;
        t.cfg = (e = getenv("STEGAN_TILE_CFG")) ? atoi(e) : 0;
        t.debug = getenv("STEGAN_PLAN_DEBUG") != nullptr;
        t.rot = (e = getenv("STEGAN_TILE_ROT")) ? atoi(e) : 1;
        return t;
    }();
    return o;
}

struct TileCfg { int WM, WN, WK; };
// rows (output channels) x columns per block = 48 WM x 48 WN; WK waves share a tile position and split each chunk's k-steps
const TileCfg kTileCfgs[] = {{1, 1, 4}, {1, 2, 2}, {2, 1, 2}, {2, 2, 1}, {2, 4, 1}, {4, 2, 1}};

inline int t_round_up(int a, int b) { return (a + b - 1) / b * b; }

// fills ``k`` for the cheapest admissible configuration; false: the kernel does not take the launch
bool tile_plan(const TileConvArgs& a, TK& k, size_t& lds) {
    const TileOpts& o = tile_opts();
    if (!o.enable) return false;
    if (!a.xh || !a.wh || !a.out) return false;
    if (a.C2 % 8 != 0 || a.C2 < 32 || a.M < 32) return false;
    if (a.Tdst % 4 != 0 || a.ntaps < 1 || a.ntaps > 7) return false;
    if (a.out_act == STEGAN_ACT_TANH) return false;       // (the generator's 8-channel output conv: not a shape for this kernel)
    if (a.mode == 1 && a.ups && (a.Tmask * 2 != a.Tdst || a.Tmask % 2 != 0)) return false;
    if (((uintptr_t)a.out % 16) || (a.res && ((uintptr_t)a.res % 16)) || (a.mask && ((uintptr_t)a.mask % 8))) return false;
    if (a.mask && !a.ups && ((uintptr_t)a.mask % 16)) return false;
    if (a.out_h && (a.Cdst % 8 != 0 || ((uintptr_t)a.out_h % 16))) return false;
    const int T = a.Tdst;                                 // output columns per batch element (stride 1: = Tq)
    const int64_t xbytes = (int64_t)a.B * ((a.Csrc + 7) / 8) * a.Tsrc * 16;
    const int64_t wbytes = (int64_t)(a.K + 1) * a.Mpad * a.Cpad * 2;
    const int64_t obytes = (int64_t)a.B * a.Cdst * a.Tdst * 4, mbytes = (int64_t)a.B * a.Cdst * (a.ups ? a.Tmask : a.Tdst) * 4;
    if (xbytes >= (int64_t)T_OOB || wbytes >= (int64_t)T_OOB || obytes >= (int64_t)T_OOB) return false;
    const int span = (a.ntaps - 1) * a.step;              // slots between the first and the last tap
    double best = 1e30;
    bool found = false;
    const int ncfg = (int)(sizeof(kTileCfgs) / sizeof(kTileCfgs[0]));
    for (int ci = 0; ci < ncfg; ++ci) {
        if (o.cfg && ci != o.cfg - 1) continue;
        const TileCfg& c = kTileCfgs[ci];
        const int Mc = WT * c.WM, Nc = WT * c.WN, ncons = c.WM * c.WN * c.WK, nthreads = 64 * (ncons + 4);
        int nseg, NTS, Tt;
        if (T >= Nc) { nseg = 1; NTS = (T + Nc - 1) / Nc; Tt = Nc; }
        else { if (Nc % T != 0) continue; nseg = Nc / T; NTS = 1; Tt = T; }
        const int MT = (a.M + Mc - 1) / Mc;
        const int NTl = NTS * ((a.B + nseg - 1) / nseg);
        const long tiles = (long)MT * NTl;
        // padding waste: rows past M, columns past T, samples past B
        const double waste = ((double)MT * Mc / a.M) * ((double)NTl * Nc / ((double)a.B * T));
        if (waste > 1.4 && !o.cfg) continue;
        const int SEGW = Tt + span;
        const int XROW = t_round_up(nseg * SEGW, 64);           // whole DMA items per channel block
        if (XROW > 256) continue;
        // channel blocks per chunk: the reduction parts of a tile position split each chunk's k-steps evenly
        int NCB = c.WK == 4 ? 16 : (c.WK == 2 ? 8 : (o.ncb1 ? o.ncb1 : 8));
#ifdef STEGAN_TILE_PAIR
        constexpr int STAGE_MAX = 38 * 1024;                    // two stages in half a CU's LDS
#else
        constexpr int STAGE_MAX = 72 * 1024;
#endif
        while (NCB > 4 && (NCB * a.ntaps * Mc + NCB * XROW) * 16 > STAGE_MAX) NCB >>= 1;
        if ((a.ntaps * (NCB / 4)) % c.WK != 0) continue;        // (uneven reduction parts)
        const int ipt = NCB * Mc / 64, xseg = XROW / 64;        // (NCB * Mc is a multiple of 192: whole periods of 3 items)
        const int nW = a.ntaps * ipt, nX = NCB * xseg;
        if (ipt >= 256 || (nW + 3) / 4 > TW_MAX || (nX + 3) / 4 > TX_MAX) continue;
        const int wstage = nW * 1024, stage = wstage + nX * 1024;
        const int nch = (a.C2 + 8 * NCB - 1) / (8 * NCB);
        const size_t epi = (size_t)ncons * (9 * 64 * 16 + 80 * 40 > 64 * 104 ? 9 * 64 * 16 + 80 * 40 : 64 * 104);
        // one block per CU may use the whole LDS; more tiles than CUs: two co-resident blocks
#ifdef STEGAN_TILE_PAIR
        const bool pair = true;
        if (nthreads > 512) continue;
        const size_t cap = 78 * 1024;
#else
        const bool pair = false;    // (157 VGPRs: one block per CU; two co-resident blocks need <= 128)
        const size_t cap = pair ? 80 * 1024 : 160 * 1024;
#endif
        const int ipc = (nW + 3) / 4 + (nX + 3) / 4;            // DMA instructions of the busiest producer wave per chunk
        // ring stages: TWO.  Four (as many as fit the LDS) are no faster per layer (sum of the 24 hot-path layers 302 vs 303 us)
        // and cost the STEP 1.7 % (5.59 -> 5.50 ms): with half the LDS left free, blocks of the small kernels of the other
        // streams (LayerNorm, attention, activation derivatives, the strided convs) are resident beside a tile block
        int NS = o.ns ? o.ns : 2;
        if (NS > nch) NS = nch;
        if (NS < 2) NS = 2;
        while (NS > 2 && ((size_t)NS * stage > cap || (NS - 1) * ipc > 48)) --NS;
        if ((size_t)NS * stage > cap || epi > cap || (NS - 1) * ipc > 48) continue;
        // cost in CU cycles: the producers' DMA rate (~32 B / clk / CU) against the MFMA time of the tile, per round of tiles
        const double bytes = ((double)Mc * a.ntaps + (double)(XROW)) * a.C2 * 2.0;
        const double mfma = (double)Mc * Nc * a.C2 * a.ntaps * 2.0 / 4096.0;
        const double per_tile = (bytes / 32.0 > mfma ? bytes / 32.0 : mfma) + 5000.0 + nch * 100.0;
        const long slots = pair ? 512 : 256;
        const double rounds = (double)((tiles + slots - 1) / slots) * (pair ? 1.7 : 1.0);
        const double cost = rounds * per_tile;
        if (o.debug)
            fprintf(stderr, "  tile-cand %dx%d wk%d tiles=%ld NCB=%d NS=%d stage=%d waste=%.2f cost=%.0f\n", Mc, Nc, c.WK, tiles,
                    NCB, NS, stage, waste, cost);
        if (cost < best) {
            best = cost; found = true;
            k.Mc = Mc; k.Nc = Nc; k.WM = c.WM; k.WN = c.WN; k.WK = c.WK; k.ncons = ncons; k.nthreads = nthreads;
            k.NCB = NCB; k.nch = nch; k.KS = NCB / 4; k.lgKS = NCB == 4 ? 0 : (NCB == 8 ? 1 : 2); k.S = a.ntaps * (NCB / 4);
            k.lgWN = c.WN == 1 ? 0 : (c.WN == 2 ? 1 : 2); k.lgWK = c.WK == 1 ? 0 : (c.WK == 2 ? 1 : 2);
            k.Tt = Tt; k.nseg = nseg; k.SEGW = SEGW; k.XROW = XROW; k.NTS = NTS;
            k.ipt = ipt; k.xseg = xseg; k.nW = nW; k.nX = nX;
            k.w_adv = (192 / Mc) * a.Mpad * 16; k.plane_k = (a.Cpad / 8) * a.Mpad * 16; k.chunk_w = NCB * a.Mpad * 16;
            k.NS = NS; k.wstage = wstage; k.stage = stage;
            k.MT = MT; k.NTl = NTl;
            lds = (size_t)NS * stage > epi ? (size_t)NS * stage : epi;
        }
    }
    if (!found) return false;
    k.xh = a.xh; k.wh = a.wh; k.scale = a.scale; k.bias = a.bias; k.res = a.res; k.mask = a.mask; k.out = a.out; k.out_h = a.out_h;
    k.xh_bytes = (unsigned)xbytes; k.wh_bytes = (unsigned)wbytes;
    k.out_bytes = (unsigned)obytes; k.mask_bytes = (unsigned)mbytes; k.outh_bytes = (unsigned)(obytes / 2);
    k.B = a.B; k.Tsrc = a.Tsrc; k.Tq = a.Tq; k.Tdst = a.Tdst; k.Tmask = a.Tmask; k.ups = a.ups; k.xups = a.mode == 0 ? a.ups : 0; k.mode = a.mode;
    k.CBx = (a.Csrc + 7) / 8; k.Cdst = a.Cdst; k.out_cb = a.Cdst / 8; k.M = a.M;
    k.Mpad = a.Mpad; k.CB = a.Cpad / 8;
    k.ntaps = a.ntaps; k.kbase = a.kbase; k.kstep = a.kstep; k.off_min = a.off_min; k.step = a.step;
    k.in_relu = a.in_relu; k.out_act = a.out_act; k.mask_lrelu = a.mask_lrelu;
    // XCD rectangles: gm parts along M x 8 / gm along N, both dividing the grid; fewest bytes through the fabric:
    // every XCD reads its rows of W and its columns of X once
    k.rot = o.rot;
    k.lgm = -1; k.mper = 1;
    {
        const double xb = (double)a.B * a.Tsrc * a.C2 * 2.0, wb = (double)a.M * a.C2 * a.ntaps * 2.0;
        double bb = 0;
        for (int l = 0; l <= 3; ++l) {
            const int gm = 1 << l, gn = 8 >> l;
            if (k.MT % gm || k.NTl % gn) continue;
            const double cst = xb * gm + wb * gn;
            if (k.lgm < 0 || cst < bb) { bb = cst; k.lgm = l; k.mper = k.MT / gm; }
        }
    }
    k.fdMc = make_fastdiv((uint32_t)k.Mc); k.fdIpt = make_fastdiv((uint32_t)k.ipt); k.fdXseg = make_fastdiv((uint32_t)k.xseg);
    k.fdSEGW = make_fastdiv((uint32_t)k.SEGW); k.fdTt = make_fastdiv((uint32_t)k.Tt); k.fdNTS = make_fastdiv((uint32_t)k.NTS);
    k.fdMper = make_fastdiv((uint32_t)k.mper); k.fdMT = make_fastdiv((uint32_t)k.MT);
    return true;
}

}  // namespace

bool stegan_tile_conv_takes(const TileConvArgs& a) {
    TK k;
    size_t lds;
    return tile_plan(a, k, lds);
}

#ifdef STEGAN_STAMPS
static unsigned long long* g_tile_stamps = nullptr;
extern "C" int stegan_tile_debug_stamps(void* p) { g_tile_stamps = (unsigned long long*)p; return 0; }
#endif

int stegan_tile_conv(const TileConvArgs& a, hipStream_t stream) {
    TK k;
    size_t lds = 0;
    if (!tile_plan(a, k, lds)) return 1;
#ifdef STEGAN_STAMPS
    k.stamps = g_tile_stamps;
#endif
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)conv_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
        attr = true;
    }
    if (tile_opts().debug)
        fprintf(stderr, "plan-tile mode=%d B=%d C2=%d M=%d taps=%d step=%d T=%d | %dx%d wk=%d threads=%d NCB=%d nch=%d NS=%d tiles=%dx%d "
                        "lds=%zu nW=%d nX=%d XROW=%d lgm=%d\n", a.mode, a.B, a.C2, a.M, a.ntaps, a.step, a.Tdst, k.Mc, k.Nc, k.WK,
                k.nthreads, k.NCB, k.nch, k.NS, k.MT, k.NTl, lds, k.nW, k.nX, k.XROW, k.lgm);
    STEGAN_TIMED_LAUNCH(conv_tile_kernel, dim3((unsigned)(k.MT * k.NTl)), dim3((unsigned)k.nthreads), lds, stream, k);
    { const hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return -(1000 + (int)e__); }   // (never 1: that means "declined")
    return STEGAN_OK;
}
