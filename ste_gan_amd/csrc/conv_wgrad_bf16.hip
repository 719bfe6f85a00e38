// Weight gradient of the 1-D convolution on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate):
//
//     dW[co, ci, k] = sum_{b, u}  dy[b, co, u] * xin[b, ci, u*s + k*d - p],       xin = in_act(upsample(x))
//
// The reduction runs over the flattened (b, u) axis -- the axis along which BOTH operands are contiguous in HBM -- but a
// bf16 MFMA operand needs 8 CONSECUTIVE reduction indices per lane, and a tap shift k*d - p moves the x operand by an
// arbitrary number of 2-byte elements: no aligned 16-byte LDS read can deliver that.  So both tiles are staged
// channel-block-major, Xs[cb][slot][8 channels] bf16 (the image of conv_bf16.hip: fp32 (B, C, T) loads with lanes along
// time, converted once, ReLU / nearest-x2 upsample / zero padding resolved while staging), and the operands are read with
// the transposing LDS read ds_read_b64_tr_b16: every lane supplies the address of one 8-byte piece (4 channels of one
// slot), a 16-lane group gathers 4 slots x 16 channels and each lane receives 4 consecutive SLOTS of its own channel.
// A tap shift or a stride is then just a different slot address per lane -- any dilation, any stride, no alignment rule.
//
//   * block = 4 waves as 2 (co) x 2 (ci); wave tile 32 TM co x 32 TN ci x KT taps: one dy fragment (2 transposed reads)
//     feeds TN * KT MFMAs; more than KT live taps run as tap groups (grid dimension).
//   * a chunk of the reduction = NB batch elements x RU output positions (R = NB * RUp columns, RUp = RU rounded up to 4 so
//     that the 4 slots of one transposed read never straddle two batch elements; padding columns carry dy = 0);
//     chunk c + 1 is loaded into registers while chunk c is multiplied, two LDS buffers, one barrier per chunk; there is
//     no LDS-DMA in the loop, so hipcc's counted vmcnt waits stay exact.
//   * planes are padded to (slots = 4 mod 16) so that the 32 lanes of a half-wave, which read 4 slots x 2 halves x 2
//     planes x 2 channel groups, cover all 64 banks once.
//   * the batch is cut into slices (grid dimension); every slice writes its partial dW to a slab [slice][k][co][ci]
//     with plain coalesced stores and a second kernel adds the slabs into dW in a fixed order (deterministic).
//
// Numerics: bf16-rounded operands, fp32 accumulation; the bias gradient stays fp32 (stegan_bias_grad).
#include "conv_direct_common.h"
#include "conv_tile.h"

namespace {

typedef hop_t bf16x8 __attribute__((ext_vector_type(8)));
typedef hop_t bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int WS_MAX = 4;      // staging items (8 channels x 1 slot) per thread, tile and chunk, scalar loads
constexpr int WSV_MAX = 2;     // ... (8 channels x 4 slots) with 16-byte loads
constexpr int NST_MAX = 8;     // MFMA k-steps (16 columns) per chunk: R <= 128
constexpr unsigned OOBW = 0x7ffff000u;

struct WH {
    const float* x;            // (B, Cin, Tin)
    const float* dy;           // (B, Cout, Tout)
    float* partial;            // slabs [slices][K_eff][Cout][Cin_g]
    int B, Cin, Cout, Cin_g, Cout_g, K, stride, dil, pad, Tin, Te, Tout, ups, relu;
    int k_lo, K_eff, NTG;      // live taps, tap groups of KT
    int NB, RU, RUp, R, NST;   // chunk: batch elements, output positions (and padded), columns, k-steps
    int SEG, RSA, RSX;         // x slots per batch element; slots per channel-block plane of the dy / x image
    int NGA, NGX, ait, xit;    // staging groups per plane (slots or runs of 4), items per thread
    int u_chunks;              // chunks along time = ceil(Tout / RU)
    int co_tiles, ci_tiles, slices, bps;     // bps = batch elements per slice (multiple of NB)
    int xcd = 1, rot = 1;                    // image kernel: XCD-contiguous tile ids; rotated chunk walk (experiments: 0)
    unsigned x_bytes, dy_bytes;
    FastDiv fdNGA, fdNGX, fdRUp, fdSEG, fdUC, fdCI, fdCO, fdTG;
    // round 3: both operands as bf16 activation images [B][C / 8][T][8] (conv_bf16.hip's DirP::out_h): staged by LDS-DMA
    const unsigned short* xh = nullptr;
    const unsigned short* dyh = nullptr;
    unsigned xh_bytes = 0, dyh_bytes = 0;
    int xslots = 0;            // staged slots of an x plane
    FastDiv fdRSA, fdRSX;      // image path: piece -> (plane, slot) of the padded LDS images
    unsigned long long* stamps = nullptr;      // diagnostic builds (-DSTEGAN_STAMPS, scripts/stamps_wgrad.py)
};
#ifdef STEGAN_STAMPS
#define WSTAMP(i_) if (h.stamps && threadIdx.x == 0) h.stamps[(size_t)blockIdx.x * 8 + (i_)] = __builtin_amdgcn_s_memtime();
#define WPSTAMP(i_) if (h.stamps && threadIdx.x == 256) h.stamps[(1u << 19) + (size_t)blockIdx.x * 8 + (i_)] = __builtin_amdgcn_s_memtime();
#else
#define WSTAMP(i_)
#define WPSTAMP(i_)
#endif

constexpr int WDI = 8;         // image path: LDS-DMA wave-instructions per producer wave, operand and chunk

__device__ __forceinline__ unsigned wpack(float a, float b, bool relu) {
    const f32x2 f = {a, b};
    s16x2 s = __builtin_bit_cast(s16x2, __builtin_convertvector(f, bf16x2));
    if (relu) s = __builtin_elementwise_max(s, (s16x2)(0));
    return __builtin_bit_cast(unsigned, s);
}

// one 32 x 16 MFMA operand (lane = channel, 8 consecutive reduction columns per half-wave) from a [cb][slot][8] image:
// two transposed reads of 4 slots each; `a0` = this lane's byte address for the first read, `d1` = byte distance of the second
__device__ __forceinline__ bf16x8 tr_operand(const char* a0, int d1) {
    typedef __attribute__((address_space(3))) s16x4* lp;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(a0 + d1));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// SIV: staging items per PRODUCER thread and operand with 16-byte loads (long rows, T >= 64: one; short rows, several
// batch elements and their halos per chunk: two).
//
// Round 3: 8 waves with fixed roles, as in conv_bf16.hip.  Waves 0-3 (consumers, 2 co x 2 ci) only read operands and issue
// MFMAs; waves 4-7 (producers) only stage: fp32 loads two chunks ahead into two register sets (no accumulators live in a
// producer, so the registers are free), conversion + LDS stores one chunk ahead, one barrier per chunk.  The first
// version had every wave do both in sequence -- load, MFMA, convert, store -- with one wave per SIMD and mostly one block
// per CU: a chunk of the 512 -> 1024 k5 layer took 2.6 us for 640 cycles of MFMA work (124 us per launch, 130 TFLOP/s).
// BiasW: the bias gradient rides in the same grid (blocks past the tile blocks), so a layer whose reduction is not
// sliced needs ONE launch: its blocks add their tile straight into dW (no slab, no reduce launch).
struct BiasW { const float* dy; float* db; int B, C, T, per, slices, vec, first; FastDiv fdT; };

#ifndef STEGAN_WGRAD_WAVES
#define STEGAN_WGRAD_WAVES 2
#endif
template <int TM, int TN, int KT, bool XV, int SIV, bool XH = false>
__device__ __forceinline__ void conv_wgrad_bf16_body(const WH& h, float* __restrict__ dw_direct, unsigned tile_id) {
    static_assert(!(XV && XH), "the image path has no use for the 16-byte fp32 loads");
    constexpr int BM = 64 * TM, BN = 64 * TN;          // co x ci tile of the block
    constexpr int CBA = BM / 8, CBX = BN / 8;          // channel-block planes of the two images
    constexpr int SI = XV ? SIV : WS_MAX;
    extern __shared__ __align__(16) char smem[];       // dy image 0 | x image 0 | dy image 1 | x image 1

    const int tid = threadIdx.x & 255, lane = tid & 63;               // index inside the role's 256 threads
    const bool producer = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) != 0;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    WSTAMP(0)
    // block -> (ci tile fastest, co tile, group, tap group, slice): neighbours share the dy tile
    unsigned t = tile_id;
    unsigned q_ = fd_div(t, h.fdCI);
    const int ci_t = (int)(t - q_ * h.ci_tiles); t = q_;
    q_ = fd_div(t, h.fdCO);
    const int co_t = (int)(t - q_ * h.co_tiles); t = q_;
    const int G = h.Cin / h.Cin_g;
    const int g = (int)(t % (unsigned)G); t /= (unsigned)G;
    q_ = fd_div(t, h.fdTG);
    const int tg = (int)(t - q_ * h.NTG);
    const int slice = (int)q_;
    const int co0 = co_t * BM, ci0 = ci_t * BN;
    const int kk0 = h.k_lo + tg * KT;                  // first tap of this pass
    const int b_begin = slice * h.bps, b_end = min(h.B, b_begin + h.bps);
    const int b_chunks = (b_end - b_begin + h.NB - 1) / h.NB;
    const int nchunks = b_chunks * h.u_chunks;

    // (each image is padded to whole wave-instructions of 64 pieces: the image path's DMA lanes past the last plane are
    // not masked, they write zeros there)
    const int abuf = ((CBA * h.RSA + 63) & ~63) * 16, xbuf = ((CBX * h.RSX + 63) & ~63) * 16, bufsz = abuf + xbuf;
    const int xsh = XV ? 2 + h.ups : 0;
    const int xgm = (1 << xsh) - 1;

    // =========================== producers (waves 4-7) ===========================
    if (producer) {
    if constexpr (XH) {
        // ---- both operands are bf16 images [B][C / 8][T][8]: a 16-byte piece of the LDS images IS 16 contiguous bytes of
        // them.  One wave-instruction copies up to 64 consecutive slots of ONE channel-block plane (LDS-DMA lands at
        // wave-uniform base + 16 * lane, and the planes are padded to the conflict-free pitch); instruction ii of a plane
        // list goes to wave ii % 4.  Slots past the row / batch slice / padding carry an out-of-range offset (the buffer
        // unit writes zeros); lanes past a plane's staged slots are switched off.
        typedef __attribute__((address_space(3))) void* lds_ptr_t;
        const auto rs_xh = __builtin_amdgcn_make_buffer_rsrc((void*)h.xh, 0, (int)h.xh_bytes, 0x00020000);
        const auto rs_yh = __builtin_amdgcn_make_buffer_rsrc((void*)h.dyh, 0, (int)h.dyh_bytes, 0x00020000);
        const int CBy = h.Cout >> 3, CBx = h.Cin >> 3;              // planes per batch element of the two images
        // pieces run LINEARLY over the padded LDS images ([plane][RSA | RSX slots]): instruction ii covers pieces
        // 64 ii .. 64 ii + 63 whatever plane they fall in (the first image build gave every plane its own runs of 64: a
        // 70-slot x plane cost two instructions, the second moving 96 bytes); the padding slots between a plane's staged
        // slots and its pitch are written as zeros and never read
        const int nA = (CBA * h.RSA + 63) >> 6, nX = (CBX * h.RSX + 63) >> 6;       // wave-instructions per chunk
        const int ylive = (h.Cout_g - co0 + 7) >> 3, xlive = (h.Cin_g - ci0 + 7) >> 3;     // live planes of the tiles
        // chunk-invariant per-lane parts
        int a_lane[WDI], a_seg[WDI], a_uu[WDI], a_ldsb[WDI];        // dy: byte offset without the chunk term | -1 | -2, row, position
        int x_lane[WDI], x_seg[WDI], x_sl[WDI], x_ldsb[WDI];
#pragma unroll
        for (int it = 0; it < WDI; ++it) {
            const int ii = it * 4 + wave;
            // (branch-free: items past nA / nX are never issued -- WG_NEST counts -- so their fields may hold anything; guarded
            // writes into these register arrays turn into whole-tuple copies, see conv_bf16.hip HB_W_OFFSETS)
            {
                const int piece = ii * 64 + lane;
                const int cg = (int)fd_div((uint32_t)piece, h.fdRSA), slot = piece - cg * h.RSA;
                const int seg = (int)fd_div((uint32_t)slot, h.fdRUp), uu = slot - seg * h.RUp;
                a_seg[it] = seg; a_uu[it] = uu;
                a_ldsb[it] = ii * 1024;
                a_lane[it] = (cg < CBA && slot < h.R && cg < ylive && seg < h.NB && uu < h.RU)
                                 ? ((seg * CBy + ((g * h.Cout_g + co0) >> 3) + cg) * h.Tout + uu) * 16 : -2;
            }
            {
                const int piece = ii * 64 + lane;
                const int cg = (int)fd_div((uint32_t)piece, h.fdRSX), slot = piece - cg * h.RSX;
                const int seg = (int)fd_div((uint32_t)slot, h.fdSEG), sl = slot - seg * h.SEG;
                x_seg[it] = seg; x_sl[it] = sl;
                x_ldsb[it] = ii * 1024;
                x_lane[it] = (cg < CBX && slot < h.xslots && cg < xlive) ? ((seg * CBx + ((g * h.Cin_g + ci0) >> 3) + cg) * h.Tin) * 16 : -2;
            }
        }
        // -2: staged as zeros (padding slots, dead planes, and the lanes past the last plane: they land in the image's padding)
        const int na_ = nA > wave ? (nA - wave + 3) >> 2 : 0, nx_ = nX > wave ? (nX - wave + 3) >> 2 : 0;
        static_assert(WDI == 8, "WG_NEST is written out for 8 items");
#define WG_NEST(n_, IT_) if (n_ > 0) { IT_(0) if (n_ > 1) { IT_(1) if (n_ > 2) { IT_(2) if (n_ > 3) { IT_(3)            \
                         if (n_ > 4) { IT_(4) if (n_ > 5) { IT_(5) if (n_ > 6) { IT_(6) if (n_ > 7) { IT_(7) } } } } } } } }
#define WG_A_IT(i_)                                                                                             \
            {                                                                                                   \
                const bool ok_ = a_lane[i_] >= 0 && b0_ + a_seg[i_] < b_end && u0_ + a_uu[i_] < h.Tout;         \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_yh, (lds_ptr_t)(A_ + a_ldsb[i_]), 16,               \
                                                         ok_ ? a_lane[i_] : (int)OOBW, so_y, 0, 0);             \
            }
#define WG_X_IT(i_)                                                                                             \
            {                                                                                                   \
                const int q_ = q0_ + x_sl[i_];                                                                  \
                const bool ok_ = x_lane[i_] >= 0 && b0_ + x_seg[i_] < b_end && q_ >= 0 && q_ < h.Te;            \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_xh, (lds_ptr_t)(X_ + x_ldsb[i_]), 16,               \
                                                         ok_ ? x_lane[i_] + (q_ >> h.ups) * 16 : (int)OOBW, so_x, 0, 0); \
            }
#define WG_DMA(c_, buf_)                                                                                        \
        {                                                                                                       \
            const int bc_ = (int)fd_div((uint32_t)(c_), h.fdUC), uc_ = (c_) - bc_ * h.u_chunks;                 \
            const int b0_ = b_begin + bc_ * h.NB, u0_ = uc_ * h.RU;                                             \
            const int q0_ = u0_ * h.stride - h.pad + kk0 * h.dil;                                               \
            char* A_ = smem + (buf_) * bufsz;                                                                   \
            char* X_ = A_ + abuf;                                                                               \
            const int so_y = (b0_ * CBy * h.Tout + u0_) * 16, so_x = b0_ * CBx * h.Tin * 16;                    \
            /* lean issue (conv_bf16.hip): the items of this wave are ONE scalar count each, nested ifs run through them in \
               a straight line, no lane is masked */                                                           \
            WG_NEST(na_, WG_A_IT)                                                                               \
            WG_NEST(nx_, WG_X_IT)                                                                               \
        }
        // the blocks that share an operand slice walk its chunks from different starting points (conv_bf16.hip: at any moment
        // the sharers of a line fetch different chunks; all but the first touch are L2 hits); the sum is the same set of
        // terms in a rotated order
        const int rot = (nchunks > 1 && h.rot) ? (int)((unsigned)(ci_t + co_t) % (unsigned)nchunks) : 0;
#define WG_CID(c_) ((c_) + rot - (((c_) + rot) >= nchunks ? nchunks : 0))
        WPSTAMP(0)
        if (nchunks > 0) WG_DMA(WG_CID(0), 0)
        WPSTAMP(1)
        for (int c = 0; c < nchunks; ++c) {
            if (c == 5) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                WPSTAMP(4)                         // chunk 5 landed
            }
            block_sync();                          // barrier c: chunk c has landed (vmcnt(0) in front of the barrier)
            if (c == 4) { WPSTAMP(2) }
            if (c + 1 < nchunks) WG_DMA(WG_CID(c + 1), (c + 1) & 1)
            if (c == 4) { WPSTAMP(3) }
        }
        return;
#undef WG_CID
#undef WG_DMA
#undef WG_NEST
#undef WG_A_IT
#undef WG_X_IT
    } else {
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)h.x, 0, (int)h.x_bytes, 0x00020000);
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)h.dy, 0, (int)h.dy_bytes, 0x00020000);
    const bool relu = h.relu != 0;
    // ---- staging items (chunk invariant part): item = (channel block, slot or run of 4 slots), slots fastest.  With the
    // nearest-x2 upsample one 16-byte load of 4 source positions of x feeds 8 slots (as in conv_bf16.hip)
    int a_cg[SI], a_seg[SI], a_uu[SI], a_lds[SI];       // dy: plane, batch element within the chunk, position, LDS byte (-1: none)
    int x_cg[SI], x_seg[SI], x_sl[SI], x_lds[SI];
#pragma unroll
    for (int it = 0; it < SI; ++it) {
        const int idx = it * 256 + tid;
        {
            const int cg = (int)fd_div((uint32_t)idx, h.fdNGA), grp = idx - cg * h.NGA;
            const int slot = XV ? grp * 4 : grp;
            const int seg = (int)fd_div((uint32_t)slot, h.fdRUp);
            a_cg[it] = cg; a_seg[it] = seg; a_uu[it] = slot - seg * h.RUp;
            a_lds[it] = (it < h.ait && cg < CBA) ? (cg * h.RSA + slot) * 16 : -1;
        }
        {
            const int cg = (int)fd_div((uint32_t)idx, h.fdNGX), grp = idx - cg * h.NGX;
            const int slot = XV ? (grp << xsh) : grp;
            const int seg = (int)fd_div((uint32_t)slot, h.fdSEG);
            x_cg[it] = cg; x_seg[it] = seg; x_sl[it] = slot - seg * h.SEG;
            x_lds[it] = (it < h.xit && cg < CBX) ? (cg * h.RSX + slot) * 16 : -1;
        }
    }
    // two register sets: chunk c + 2 is loading while chunk c + 1 is converted and stored.  (The two-item 16-byte variant
    // would need 256 staging registers for that -- it spilled 65-121 VGPRs when built so: it keeps ONE set and loads one
    // chunk ahead, its load latency beside the consumers' MFMAs instead of behind a whole chunk.)
    constexpr bool DEEP = !XV || SIV == 1;
    float ar0[XV ? 1 : WS_MAX][8], xr0[XV ? 1 : WS_MAX][8], ar1[(XV || !DEEP) ? 1 : WS_MAX][8], xr1[(XV || !DEEP) ? 1 : WS_MAX][8];
    f32x4 av0[XV ? SIV : 1][8], xv0[XV ? SIV : 1][8], av1[(XV && DEEP) ? SIV : 1][8], xv1[(XV && DEEP) ? SIV : 1][8];
    const int rowb_y = h.Tout * 4, rowb_x = h.Tin * 4;
    const int ylim = h.Cout_g - co0, xlim = h.Cin_g - ci0;       // live channels of the tiles

#define WG_LOAD(c_, ar, av, xr, xv)                                                                             \
    {                                                                                                           \
        const int bc_ = (int)fd_div((uint32_t)(c_), h.fdUC), uc_ = (c_) - bc_ * h.u_chunks;                     \
        const int b0_ = b_begin + bc_ * h.NB, u0_ = uc_ * h.RU;                                                 \
        const int q0_ = u0_ * h.stride - h.pad + kk0 * h.dil;                                                   \
        const int qal_ = XV ? (q0_ & ~xgm) : q0_;                                                               \
        _Pragma("unroll") for (int it = 0; it < SI; ++it) {                                                     \
            if (it < h.ait) {                                                                                   \
                const int b_ = b0_ + a_seg[it], u_ = u0_ + a_uu[it];                                            \
                const bool ok_ = a_lds[it] >= 0 && a_seg[it] < h.NB && b_ < b_end && a_uu[it] < h.RU && u_ < h.Tout; \
                const int vo_ = ok_ ? ((b_ * h.Cout + g * h.Cout_g + co0 + a_cg[it] * 8) * h.Tout + u_) * 4 : (int)OOBW; \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                 \
                    const int vj_ = (a_cg[it] * 8 + j < ylim) ? vo_ : (int)OOBW;                                \
                    if constexpr (XV) av[it][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_y, vj_, j * rowb_y, 0)); \
                    else ar[it][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_y, vj_, j * rowb_y, 0)); \
                }                                                                                               \
            }                                                                                                   \
            if (it < h.xit) {                                                                                   \
                const int b_ = b0_ + x_seg[it], q_ = qal_ + x_sl[it];                                           \
                const bool ok_ = x_lds[it] >= 0 && b_ < b_end && q_ >= 0 && q_ < h.Te;                          \
                const int vo_ = ok_ ? ((b_ * h.Cin + g * h.Cin_g + ci0 + x_cg[it] * 8) * h.Tin + (q_ >> h.ups)) * 4 : (int)OOBW; \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                 \
                    const int vj_ = (x_cg[it] * 8 + j < xlim) ? vo_ : (int)OOBW;                                \
                    if constexpr (XV) xv[it][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, vj_, j * rowb_x, 0)); \
                    else xr[it][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, vj_, j * rowb_x, 0)); \
                }                                                                                               \
            }                                                                                                   \
        }                                                                                                       \
    }
#define WG_STORE_ONE(dst_, lds_, r_, v_, relu_, dup_)                                                           \
    if constexpr (XV) {                                                                                         \
        _Pragma("unroll") for (int s4 = 0; s4 < 4; ++s4) {                                                      \
            const u32x4 w_ = {wpack(v_[0][s4], v_[1][s4], relu_), wpack(v_[2][s4], v_[3][s4], relu_),            \
                              wpack(v_[4][s4], v_[5][s4], relu_), wpack(v_[6][s4], v_[7][s4], relu_)};           \
            if (dup_) {                               /* nearest x2: source position s4 -> slots 2 s4, 2 s4 + 1 */ \
                *(u32x4*)(dst_ + lds_ + s4 * 32) = w_;                                                          \
                *(u32x4*)(dst_ + lds_ + s4 * 32 + 16) = w_;                                                     \
            } else {                                                                                            \
                *(u32x4*)(dst_ + lds_ + s4 * 16) = w_;                                                          \
            }                                                                                                   \
        }                                                                                                       \
    } else {                                                                                                    \
        const u32x4 w_ = {wpack(r_[0], r_[1], relu_), wpack(r_[2], r_[3], relu_), wpack(r_[4], r_[5], relu_), wpack(r_[6], r_[7], relu_)}; \
        *(u32x4*)(dst_ + lds_) = w_;                                                                            \
    }
#define WG_STORE(buf_, ar, av, xr, xv)                                                                          \
    {                                                                                                           \
        char* A_ = smem + (buf_) * bufsz;                                                                       \
        char* X_ = A_ + abuf;                                                                                   \
        _Pragma("unroll") for (int it = 0; it < SI; ++it) {                                                     \
            if (it < h.ait && a_lds[it] >= 0) { WG_STORE_ONE(A_, a_lds[it], ar[XV ? 0 : it], av[XV ? it : 0], false, false) } \
            if (it < h.xit && x_lds[it] >= 0) { WG_STORE_ONE(X_, x_lds[it], xr[XV ? 0 : it], xv[XV ? it : 0], relu, xsh == 3) }  \
        }                                                                                                       \
    }
        // chunk c lives in register set c & 1 and LDS buffer c & 1.  Barrier c (the c-th block_sync of BOTH roles) says:
        // chunk c is staged and visible, and the consumers have finished chunk c - 1 (its buffer may be overwritten).
        if constexpr (!DEEP) {
            if (nchunks > 0) {
                WG_LOAD(0, ar0, av0, xr0, xv0)
                WG_STORE(0, ar0, av0, xr0, xv0)
            }
            for (int c = 0; c < nchunks; ++c) {
                block_sync();                                         // barrier c
                if (c + 1 < nchunks) {
                    WG_LOAD(c + 1, ar0, av0, xr0, xv0)
                    WG_STORE((c + 1) & 1, ar0, av0, xr0, xv0)
                }
            }
            return;
        }
        if (nchunks > 0) {
            WG_LOAD(0, ar0, av0, xr0, xv0)
            if (nchunks > 1) WG_LOAD(1, ar1, av1, xr1, xv1)
            WG_STORE(0, ar0, av0, xr0, xv0)
        }
        for (int c = 0; c < nchunks; c += 2) {
            block_sync();                                             // barrier c
            if (c + 2 < nchunks) WG_LOAD(c + 2, ar0, av0, xr0, xv0)   // set 0 held chunk c: stored before barrier c
            if (c + 1 < nchunks) WG_STORE(1, ar1, av1, xr1, xv1)      // chunk c + 1 -> buffer 1 (chunk c - 1 is done with it)
            if (c + 1 >= nchunks) break;
            block_sync();                                             // barrier c + 1
            if (c + 3 < nchunks) WG_LOAD(c + 3, ar1, av1, xr1, xv1)
            if (c + 2 < nchunks) WG_STORE(0, ar0, av0, xr0, xv0)
        }
        return;                                         // the accumulators live in the consumer waves
    }   // !XH
    }
#undef WG_LOAD
#undef WG_STORE
#undef WG_STORE_ONE

    // =========================== consumers (waves 0-3) ===========================
    // ---- operand addresses.  Lane l: half hl, channel l & 31 of the wave's 32-channel tile; inside its 16-lane group
    // i = l & 15 it supplies the piece (row q = i >> 2, channels 4 p .. 4 p + 3 with p = i & 3) of each transposed read
    const int hl = lane >> 5, g16 = (lane >> 4) & 1, qrow = (lane >> 2) & 3, pp = lane & 3;
    int a_byte[TM], b_byte[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
        a_byte[i] = (((wm * TM + i) * 4 + 2 * g16 + (pp >> 1)) * h.RSA + 8 * hl + qrow) * 16 + (pp & 1) * 8;
#pragma unroll
    for (int j = 0; j < TN; ++j)
        b_byte[j] = (((wn * TN + j) * 4 + 2 * g16 + (pp >> 1)) * h.RSX + qrow * h.stride) * 16 + (pp & 1) * 8;
    // x slot of the first column of each transposed read: columns r4 = 16 st + 8 hl + 4 rd -> (batch element, position)
    int xoff[NST_MAX][2];
#pragma unroll
    for (int st = 0; st < NST_MAX; ++st)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const int r4 = st * 16 + 8 * hl + 4 * rd;
            const int seg = (int)fd_div((uint32_t)r4, h.fdRUp), uu = r4 - seg * h.RUp;
            xoff[st][rd] = (seg * h.SEG + uu * h.stride) * 16;
        }
    const int tapB = h.dil * 16;

    f32x16 acc[TM][TN][KT];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int k = 0; k < KT; ++k)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][k][e] = 0.f;
    const short rfl = (XH && h.relu) ? (short)0 : (short)-32768;       // max(x, 0) on packed halves, or identity
    const s16x8 relu_fl = {rfl, rfl, rfl, rfl, rfl, rfl, rfl, rfl};
    (void)relu_fl;

    WSTAMP(1)
    for (int c = 0; c < nchunks; ++c) {
        const int cur = c & 1;
        block_sync();                                                 // barrier c: chunk c is staged
        if (c == 1) { WSTAMP(2) }
        if (c == 5) { WSTAMP(3) }
        int adj = 0;                                                  // down-rounding of the x window start (XV)
        if (XV) {
            const int bc = (int)fd_div((uint32_t)c, h.fdUC), uc = c - bc * h.u_chunks;
            const int q0 = uc * h.RU * h.stride - h.pad + kk0 * h.dil;
            adj = q0 - (q0 & ~xgm);
        }
        const char* Acur = smem + cur * bufsz;
        const char* Xcur = Acur + abuf + adj * 16;
        // k-steps of 16 columns, software-pipelined over two operand register sets: the transposed reads of step st + 1 are
        // in flight while the MFMAs of step st run (the first version read, waited and multiplied step by step: ~200 cycles
        // of exposed LDS latency for 64-160 cycles of MFMA per step)
#define WG_READ(a_, b_, st_)                                                                                    \
        {                                                                                                       \
            int xo0 = 0, xo1 = 0;       /* xoff[st_]: selected through registers (an indexed array would go to scratch) */ \
            _Pragma("unroll") for (int s2 = 0; s2 < NST_MAX; ++s2)                                              \
                if (s2 == (st_)) { xo0 = xoff[s2][0]; xo1 = xoff[s2][1]; }                                      \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) a_[i] = tr_operand(Acur + a_byte[i] + (st_) * 256, 64); \
            _Pragma("unroll") for (int k = 0; k < KT; ++k)                                                      \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                  \
                    b_[k][j] = tr_operand(Xcur + b_byte[j] + xo0 + k * tapB, xo1 - xo0);                        \
        }
#define WG_MFMA(a_, b_)                                                                                         \
        _Pragma("unroll") for (int k = 0; k < KT; ++k)                                                          \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                    \
                const bf16x8 bx_ = XH ? __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, b_[k][j]), relu_fl)) \
                                      : b_[k][j];      /* image path: the input ReLU at the operand */         \
                _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                  \
                    acc[i][j][k] = STEGAN_MFMA_H(a_[i], bx_, acc[i][j][k]);                                   \
            }
        bf16x8 aA[TM], bA[KT][TN], aB[TM], bB[KT][TN];
        if (h.NST > 0) WG_READ(aA, bA, 0)
        for (int st = 0; st < h.NST; st += 2) {
            if (st + 1 < h.NST) WG_READ(aB, bB, st + 1)
            WG_MFMA(aA, bA)
            if (st + 1 < h.NST) {
                if (st + 2 < h.NST) WG_READ(aA, bA, st + 2)
                WG_MFMA(aB, bB)
            }
        }
#undef WG_READ
#undef WG_MFMA
    }

    WSTAMP(4)
    // ---- store: accumulator tile = 32 co (registers: 4 runs of 4 rows) x 32 ci (lanes), KT taps per element.  dW is
    // (Cout, Cin_g, K) with the tap innermost, so a lane's KT values of one (co, ci) are CONTIGUOUS there, and the lanes of
    // a half-wave (consecutive ci) cover 32 x KT contiguous floats: one or two wide accesses per (co, ci) instead of KT
    // scattered dwords (the first version, tap-outermost with one dword per access, spent 14-40 k cycles per block here:
    // store-issue bound).  Sliced reductions write their partial dW to a slab in the same [co][ci][kk] order (kk over the
    // live taps), summed by the reduce kernel in a fixed order; an unsliced one adds straight into dW: this block is
    // the only writer of its elements.
    // (the wide accesses go through a buffer descriptor: a (co, ci) group is only dword aligned -- 12 or 20 bytes apart --
    // and hipcc splits an under-aligned vector access of a flat pointer into dwords, while buffer_load / _store_dwordx4
    // take any dword-aligned offset)
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const int l31 = lane & 31;
    const int kq0 = tg * KT;                                // first live tap of this block
    const int nk = min(KT, h.K_eff - kq0);                  // taps it owns
    const int KK = dw_direct ? h.K : h.K_eff;               // taps per (co, ci) in the destination
    float* const dst0 = dw_direct ? dw_direct + h.k_lo + kq0
                                  : h.partial + (size_t)slice * h.K_eff * h.Cout * h.Cin_g + kq0;
    const auto rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)dst0, 0, 0x7ffffff0, 0x00020000);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int ci = ci0 + (wn * TN + j) * 32 + l31;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int cob = co0 + (wm * TM + i) * 32 + 4 * hl;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = cob + (e & 3) + 8 * (e >> 2);
                if (ci >= h.Cin_g || co >= h.Cout_g) continue;
                const int off = (int)((((size_t)(g * h.Cout_g + co) * h.Cin_g + ci) * KK) * 4);       // bytes from dst0
                float v[KT];
#pragma unroll
                for (int k = 0; k < KT; ++k) v[k] = acc[i][j][k][e];
                if (nk == KT) {
                    if (dw_direct) {                        // read-modify-write: all loads first, then the stores
                        // (dword loads: a (co, ci) group is only dword aligned, and a buffer_load_dwordx2 at a 4-byte-but-
                        // not-8-byte aligned offset returned its FIRST DWORD TWICE on gfx950 -- found in round 4 when an
                        // experiment read slabs back this way (tap 1 == tap 0); every accumulating call so far had met a
                        // zeroed or constant dW, which hides it.  The wide STORES are fine: the slabs they write have always
                        // been read back correctly by the reduce kernel.)
                        float o[KT];
#pragma unroll
                        for (int k = 0; k < KT; ++k)
                            o[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_o, off + 4 * k, 0, 0));
#pragma unroll
                        for (int q = 0; q < KT; ++q) v[q] += o[q];
                    }
                    int k = 0;
                    if constexpr (KT >= 4) {
#pragma unroll
                        for (; k + 4 <= KT; k += 4) {
                            const u32x4 t4 = {__builtin_bit_cast(unsigned, v[k]), __builtin_bit_cast(unsigned, v[k + 1]),
                                              __builtin_bit_cast(unsigned, v[k + 2]), __builtin_bit_cast(unsigned, v[k + 3])};
                            __builtin_amdgcn_raw_buffer_store_b128(t4, rs_o, off + 4 * k, 0, 0);
                        }
                    }
                    if constexpr ((KT & 3) >= 2) {
                        const u32x2 t2 = {__builtin_bit_cast(unsigned, v[k]), __builtin_bit_cast(unsigned, v[k + 1])};
                        __builtin_amdgcn_raw_buffer_store_b64(t2, rs_o, off + 4 * k, 0, 0);
                        k += 2;
                    }
                    if constexpr (KT & 1) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[k]), rs_o, off + 4 * k, 0, 0);
                } else {                                    // ragged last tap group
                    float* ptr = dst0 + (off >> 2);
#pragma unroll
                    for (int k = 0; k < KT; ++k)
                        if (k < nk) ptr[k] = dw_direct ? ptr[k] + v[k] : v[k];
                }
            }
        }
    }
    WSTAMP(5)
}

// bias-gradient blocks of the weight-gradient grid (512 threads): db[c] += sum over a batch slice of dy[b, c, :]
__device__ __forceinline__ void wg_bias_block(const BiasW& ba, int j) {
    const int c = j / ba.slices, slice = j - c * ba.slices;
    const int b0 = slice * ba.per, b1 = min(ba.B, b0 + ba.per);
    float s = 0.f;
    if (ba.vec) {
        const int T4 = ba.T >> 2, n = (b1 - b0) * T4;
        for (int r = threadIdx.x; r < n; r += 512) {
            const int bi = (int)fd_div((uint32_t)r, ba.fdT), t4 = r - bi * T4;
            const float4 v = *reinterpret_cast<const float4*>(ba.dy + ((size_t)(b0 + bi) * ba.C + c) * ba.T + 4 * t4);
            s += (v.x + v.y) + (v.z + v.w);
        }
    } else {
        const int n = (b1 - b0) * ba.T;
        for (int r = threadIdx.x; r < n; r += 512) {
            const int bi = (int)fd_div((uint32_t)r, ba.fdT), t_ = r - bi * ba.T;
            s += ba.dy[((size_t)(b0 + bi) * ba.C + c) * ba.T + t_];
        }
    }
    extern __shared__ __align__(16) char smem_bias[];       // the grid's dynamic LDS (>= 32 bytes: the tile blocks' images)
    float* part = reinterpret_cast<float*>(smem_bias);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    block_sync();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int w = 0; w < 8; ++w) tot += part[w];
        ba.db[c] += tot;               // ba.slices == 1: this block is the only writer of db[c] (fixed order, no atomics)
    }
}

#if STEGAN_WGRAD_WAVES > 0
#define WG_OCC __attribute__((amdgpu_waves_per_eu(STEGAN_WGRAD_WAVES, STEGAN_WGRAD_WAVES)))
#else
#define WG_OCC
#endif
template <int TM, int TN, int KT, bool XV>
__global__ __launch_bounds__(512) WG_OCC void conv_wgrad_bf16_kernel(const WH h, float* dw_direct, const BiasW ba) {
    if ((int)blockIdx.x >= ba.first) { wg_bias_block(ba, (int)blockIdx.x - ba.first); return; }
    conv_wgrad_bf16_body<TM, TN, KT, XV, WSV_MAX>(h, dw_direct, blockIdx.x);
}

template <int TM, int TN, int KT>
__global__ __launch_bounds__(512) WG_OCC void conv_wgrad_bf16_one_kernel(const WH h, float* dw_direct, const BiasW ba) {
    if ((int)blockIdx.x >= ba.first) { wg_bias_block(ba, (int)blockIdx.x - ba.first); return; }
    conv_wgrad_bf16_body<TM, TN, KT, true, 1>(h, dw_direct, blockIdx.x);
}

template <int TM, int TN, int KT>
__global__ __launch_bounds__(512) WG_OCC void conv_wgrad_bf16_img_kernel(const WH h, float* dw_direct, const BiasW ba) {
    if ((int)blockIdx.x >= ba.first) { wg_bias_block(ba, (int)blockIdx.x - ba.first); return; }
    // Blocks are dealt round-robin over the 8 XCDs: every XCD takes ONE contiguous range of tile ids (ids run ci tile, co tile,
    // group, tap group, slice: a range is a slice, or part of one), so the x / dy rows of a batch slice are fetched into one
    // L2 instead of eight (round 2: 4.8x the operand bytes through the fabric per launch).
    unsigned t = blockIdx.x;
    if (h.xcd) {
        const unsigned total = (unsigned)ba.first, L = blockIdx.x;
        const unsigned q8 = total >> 3, r8 = total & 7, xcd = L & 7;
        t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (L >> 3);
    }
    conv_wgrad_bf16_body<TM, TN, KT, false, 1, true>(h, dw_direct, t);
}

// dW[co][ci][k_lo + kk] += sum over slices of slab[slice][kk][co][ci]   (fixed order: deterministic)
// The bias gradient rides in the same launch: blocks past the first `rb` are bias_grad_body (channel, batch slice) blocks
// over the fp32 dy (one launch less per layer: 52 a step).
struct BiasArgs { const float* dy; float* db; int B, C, T, per, slices, vec; FastDiv fdT; };

__global__ __launch_bounds__(256) void conv_wgrad_bf16_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                     long per_tap, int slices, int K, int k_lo, int K_eff,
                                                                     int rb, const BiasArgs ba) {
    if ((int)blockIdx.x >= rb) {
        const int j = (int)blockIdx.x - rb;                  // channel-major: the slices of a channel are neighbours
        bias_grad_body(ba.dy, ba.db, ba.B, ba.C, ba.T, ba.per, ba.fdT, ba.vec, j / ba.slices, j % ba.slices);
        return;
    }
    // slabs are [co][ci][kk] (kk over the K_eff live taps): element i = r * K_eff + kk adds into dw[r * K + k_lo + kk]
    const long total = per_tap * K_eff;
    if (K_eff == K && (total & 3) == 0 && (((uintptr_t)partial | (uintptr_t)dw) & 15) == 0) {
        // every tap live: slab element i IS dw element i -- 16 bytes per lane, every slab's load in flight at once, no division
        // (the scalar loop below spent 7-9 us on a 442 k-element layer: a 64-bit division per element and 4-byte accesses)
        const long n4 = total >> 2;
        const f32x4* src4 = (const f32x4*)partial;
        f32x4* dw4 = (f32x4*)dw;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)rb * 256) {
            f32x4 o = dw4[i];
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            int s = 0;
            for (; s + 4 <= slices; s += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] += src4[(size_t)(s + u) * n4 + i];
            }
            for (; s < slices; ++s) v[0] += src4[(size_t)s * n4 + i];
            dw4[i] = o + ((v[0] + v[1]) + (v[2] + v[3]));
        }
        return;
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)rb * 256) {
        const long r = i / K_eff;                          // co * Cin_g + ci
        const int kk = (int)(i - r * K_eff);
        const float* src = partial + i;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        int s = 0;
        for (; s + 4 <= slices; s += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] += src[(size_t)(s + u) * total];
        }
        for (; s < slices; ++s) v[0] += src[(size_t)s * total];
        dw[r * K + k_lo + kk] += (v[0] + v[1]) + (v[2] + v[3]);
    }
}

int wgh_live_taps(const stegan_conv_desc* d, int& k_lo) {
    const int Te = d->Tin * d->upsample;
    int lo = d->K, hi = -1;
    for (int k = 0; k < d->K; ++k) {
        const int a = k * d->dilation - d->pad, b = (d->Tout - 1) * d->stride + a;
        if (b >= 0 && a < Te) { if (k < lo) lo = k; hi = k; }
    }
    k_lo = lo;
    return hi - lo + 1;
}

template <int TM, int TN, int KT>
void wgh_launch(bool xv, dim3 grid, size_t lds, hipStream_t s, const WH& h, float* dw_direct, const BiasW& bw) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_bf16_kernel<TM, TN, KT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_wgrad_bf16_kernel<TM, TN, KT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_wgrad_bf16_one_kernel<TM, TN, KT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_wgrad_bf16_img_kernel<TM, TN, KT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    if (h.xh) {
        STEGAN_TIMED_LAUNCH((conv_wgrad_bf16_img_kernel<TM, TN, KT>), grid, dim3(512), lds, s, h, dw_direct, bw);
        return;
    }
    static const bool no_one = getenv("STEGAN_BF16_WGRAD_NO_ONE") != nullptr;       // experiments
    if (xv && h.ait <= 1 && h.xit <= 1 && !no_one)
        STEGAN_TIMED_LAUNCH((conv_wgrad_bf16_one_kernel<TM, TN, KT>), grid, dim3(512), lds, s, h, dw_direct, bw);
    else if (xv) STEGAN_TIMED_LAUNCH((conv_wgrad_bf16_kernel<TM, TN, KT, true>), grid, dim3(512), lds, s, h, dw_direct, bw);
    else STEGAN_TIMED_LAUNCH((conv_wgrad_bf16_kernel<TM, TN, KT, false>), grid, dim3(512), lds, s, h, dw_direct, bw);
}

}  // namespace

// Accumulates into dw (and db, through the fp32 bias-gradient kernel).  Returns 1 (nothing enqueued) for shapes left to
// the fp32 kernels: fewer than 16 channels per group on either side, or a workspace too small for one slab.
static int wgh_run(const stegan_conv_desc* d, const float* x, const uint16_t* xh, const float* dy, const uint16_t* dyh,
                   float* dw, float* db, float* workspace, int64_t workspace_floats, stegan_wgrad_reduce* deferred,
                   void* stream, const stegan_wgrad_reduce* prev = nullptr, int* prev_done = nullptr) {
    if (deferred) { deferred->partial = nullptr; deferred->dw = nullptr; deferred->per_tap = 0; deferred->slices = 0;
                    deferred->K = 0; deferred->k_lo = 0; deferred->K_eff = 0; deferred->used_floats = 0; }
    int rc = check_desc(d);
    if (rc) return rc;
    STEGAN_CHECK_ARG(x && dy && dw);
    STEGAN_CHECK_ARG(d->in_act == STEGAN_ACT_NONE || d->in_act == STEGAN_ACT_RELU);
    hipStream_t s = (hipStream_t)stream;
    WH h;
    h.x = x; h.dy = dy;
    // both operands as bf16 images (LDS-DMA staging): whole 8-channel blocks per group on both sides
    static const bool no_xh = getenv("STEGAN_BF16_NO_XH") != nullptr;
    const bool img = xh && dyh && !no_xh && (d->Cin / d->groups) % 8 == 0 && (d->Cout / d->groups) % 8 == 0 &&
                     (int64_t)d->B * (d->Cin / 8) * d->Tin * 16 < (int64_t)OOBW && (int64_t)d->B * (d->Cout / 8) * d->Tout * 16 < (int64_t)OOBW;
    if (img) {
        h.xh = xh; h.dyh = dyh;
        h.xh_bytes = (unsigned)((int64_t)d->B * (d->Cin / 8) * d->Tin * 16);
        h.dyh_bytes = (unsigned)((int64_t)d->B * (d->Cout / 8) * d->Tout * 16);
    }
    h.B = d->B; h.Cin = d->Cin; h.Cout = d->Cout; h.Cin_g = d->Cin / d->groups; h.Cout_g = d->Cout / d->groups;
    h.K = d->K; h.stride = d->stride; h.dil = d->dilation; h.pad = d->pad; h.Tin = d->Tin; h.Tout = d->Tout;
    h.Te = d->Tin * d->upsample; h.ups = d->upsample == 2 ? 1 : 0; h.relu = d->in_act == STEGAN_ACT_RELU;
    static const char* env_min = getenv("STEGAN_BF16_WGRAD_MIN_C");
    const int min_c = env_min ? atoi(env_min) : 16;
    if (h.Cin_g < min_c || h.Cout_g < min_c) return 1;
    if ((int64_t)d->B * d->Cin * d->Tin * 4 >= (int64_t)OOBW || (int64_t)d->B * d->Cout * d->Tout * 4 >= (int64_t)OOBW) return 1;
    h.x_bytes = (unsigned)((int64_t)d->B * d->Cin * d->Tin * 4);
    h.dy_bytes = (unsigned)((int64_t)d->B * d->Cout * d->Tout * 4);
    h.K_eff = wgh_live_taps(d, h.k_lo);
    if (h.K_eff <= 0) return STEGAN_OK;                  // every tap reads only padding: dW += 0
    // round 5: stride-1 ungrouped layers with both operands as images go to the one-tile-per-CU kernel (conv_tile_wgrad.hip)
    // when its plan takes the shape; its slabs are summed by the same ordered reduce as this file's
    if (img && d->stride == 1 && d->groups == 1) {
        TileWgradArgs ta;
        ta.xh = xh; ta.dyh = dyh; ta.dy = dy; ta.dw = dw; ta.db = db; ta.partial = workspace; ta.partial_floats = workspace_floats;
        ta.B = d->B; ta.Cin = d->Cin; ta.Cout = d->Cout; ta.K = d->K; ta.dil = d->dilation; ta.pad = d->pad; ta.Tin = d->Tin;
        ta.Tout = d->Tout; ta.ups = h.ups; ta.relu = h.relu; ta.k_lo = h.k_lo; ta.Keff = h.K_eff;
        ta.bias_in_reduce = deferred ? 0 : 1;
        // a pending slab sum of the layer before rides in this launch when it is the 16-byte case of the reduce kernel
        ta.prev_partial = nullptr; ta.prev_dw = nullptr; ta.prev_n4 = 0; ta.prev_slices = 0;
        if (prev && prev->slices > 0 && prev->K_eff == prev->K && prev->k_lo == 0 && ((prev->per_tap * prev->K_eff) & 3) == 0 &&
            (((uintptr_t)prev->partial | (uintptr_t)prev->dw) & 15) == 0) {
            ta.prev_partial = prev->partial; ta.prev_dw = prev->dw; ta.prev_n4 = prev->per_tap * prev->K_eff / 4;
            ta.prev_slices = prev->slices;
        }
        int sl = 0, bias_done = 0;
        const int trc = stegan_tile_wgrad(ta, &sl, &bias_done, s);
        if (trc != 1) {
            if (trc) return trc;
            if (prev_done && ta.prev_n4 > 0) *prev_done = 1;
            const int64_t slab_t = (int64_t)h.K_eff * h.Cout * h.Cin_g;
            if (sl > 1 && deferred) {
                deferred->partial = workspace; deferred->dw = dw; deferred->per_tap = (int64_t)h.Cout * h.Cin_g;
                deferred->slices = sl; deferred->K = d->K; deferred->k_lo = h.k_lo; deferred->K_eff = h.K_eff;
                deferred->used_floats = slab_t * sl;
            } else if (sl > 1) {
                const long per_tap = (long)h.Cout * h.Cin_g;
                long rb = (per_tap * h.K_eff + 255) / 256;
                if (rb > 2048) rb = 2048;
                BiasArgs ba;                            // the bias gradient rides here: blocks of 16 bytes of LDS
                const bool bias_here = db && !bias_done;
                ba.dy = dy; ba.db = bias_here ? db : nullptr; ba.B = d->B; ba.C = d->Cout; ba.T = d->Tout;
                ba.slices = bias_here ? 1 : 0; ba.per = d->B;
                ba.vec = (d->Tout % 4 == 0) && ((uintptr_t)dy % 16 == 0);
                ba.fdT = make_fastdiv((uint32_t)(ba.vec ? d->Tout / 4 : d->Tout));
                hipLaunchKernelGGL(conv_wgrad_bf16_reduce_kernel, dim3((unsigned)(rb + (bias_here ? d->Cout : 0))), dim3(256), 0, s,
                                   (const float*)workspace, dw, per_tap, sl, d->K, h.k_lo, h.K_eff, (int)rb, ba);
            }
            STEGAN_LAUNCH_CHECK();
            return STEGAN_OK_TILE;
        }
    }
    // tile: 128 x 64 x 3 taps for the dense k = 3 layers; single taps take 128 x 128; more taps 64 x 64 x (5 | 8)
    int TM, TN, KT;
    if (h.K_eff == 1) { KT = 1; TM = h.Cout_g >= 96 ? 2 : 1; TN = h.Cin_g >= 96 ? 2 : 1; }
    else if (h.K_eff <= 3) {
        // (128 x 128 x 3 taps -- more MFMA work per staged byte: 32 KiB per 64-column chunk for 1 536 MFMA cycles per wave
        // against 24 KiB for 768 -- was built for the image path and measured SLOWER: 245 registers, 41.7 vs 34 us on the
        // 384 -> 384 layers, train step 7.19 vs 6.98 ms; opt-in for experiments: STEGAN_BF16_WGRAD_223=1)
        static const bool use_223 = getenv("STEGAN_BF16_WGRAD_223") != nullptr;
        KT = 3; TM = h.Cout_g >= 96 ? 2 : 1; TN = (img && use_223 && TM == 2 && h.Cin_g >= 128) ? 2 : 1;
    }
    else if (h.K_eff <= 5) { KT = 5; TM = 1; TN = 1; }
    else { KT = 8; TM = 1; TN = 1; }
    h.NTG = ceil_div(h.K_eff, KT);
    const int BM = 64 * TM, BN = 64 * TN;
    const bool xv = !img && h.Tin % 4 == 0 && h.Tout % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0 &&
                    !STEGAN_ENV("STEGAN_BF16_NO_XV") && !(h.ups && STEGAN_ENV("STEGAN_BF16_NO_XVUP"));
    const int xg = 4 << h.ups;                           // x slots per 16-byte item
    // chunk of the reduction: up to RMAX columns (LDS budget / halo share), whole batch elements when the rows are short
    static const char* env_r = getenv("STEGAN_BF16_WGRAD_R");
    const int RMAX0 = env_r ? atoi(env_r) : 64;
    size_t lds = 0;
    auto pad4 = [](int v) { int r = v % 16; return v + ((4 - r) + 16) % 16; };     // slots = 4 (mod 16): conflict-free reads
    auto plan = [&](int RMAX) -> bool {
        h.RU = h.Tout < RMAX ? h.Tout : RMAX;
        h.RUp = round_up(h.RU, 4);
        h.NB = h.RUp < RMAX ? RMAX / h.RUp : 1;
        if (h.NB > h.B) h.NB = h.B;
        h.R = round_up(h.NB * h.RUp, 16);
        h.NST = h.R / 16;
        if (h.NST > NST_MAX) return false;
        h.u_chunks = ceil_div(h.Tout, h.RU);
        h.SEG = (h.RUp - 1) * h.stride + (KT - 1) * h.dil + 1;
        if (xv) h.SEG = (h.SEG + 2 * (xg - 1)) & ~(xg - 1);
        // columns of the last k-step past NB * RUp belong to "batch element NB": its slots must exist (they hold zeros)
        const int segs = ceil_div(h.R, h.RUp);
        // staged slots: the R columns of dy; for x the segments plus the slots of down-rounding and the reach of the last
        // transposed read; the planes are then padded (never staged, never read) to the conflict-free pitch
        const int xslots = xv ? round_up(segs * h.SEG + xg, xg) : round_up(segs * h.SEG + 4, 4);
        h.RSA = pad4(h.R);
        h.RSX = pad4(xslots);
        h.NGA = xv ? h.R / 4 : h.R;
        h.NGX = xv ? xslots / xg : xslots;
        h.ait = ceil_div((BM / 8) * h.NGA, 256);
        h.xit = ceil_div((BN / 8) * h.NGX, 256);
        h.xslots = xslots;
        if (img) {                                     // DMA pieces, not register items: <= WDI wave-instructions per wave
            if (ceil_div(ceil_div((BM / 8) * h.RSA, 64), 4) > WDI || ceil_div(ceil_div((BN / 8) * h.RSX, 64), 4) > WDI) return false;
        } else if (h.ait > (xv ? WSV_MAX : WS_MAX) || h.xit > (xv ? WSV_MAX : WS_MAX)) return false;
        lds = 2 * ((size_t)(((BM / 8) * h.RSA + 63) & ~63) + (size_t)(((BN / 8) * h.RSX + 63) & ~63)) * 16;
        return lds <= 160 * 1024;
    };
    if (!plan(RMAX0)) return 1;
    // short rows with wide halos (T = 12, d = 3 | 9: five batch elements and their halos per chunk) need two x items per
    // thread and 93 KB of LDS: one block per CU.  Half the chunk fits the one-item variant and twice on a CU
    static const bool no_half = getenv("STEGAN_BF16_WGRAD_NO_HALF") != nullptr;      // experiments
    if (xv && !no_half && !env_r && (h.ait > 1 || h.xit > 1)) {
        const WH keep = h;
        const size_t keep_lds = lds;
        if (!(plan(RMAX0 / 2) && h.ait <= 1 && h.xit <= 1 && lds <= 80 * 1024)) { h = keep; lds = keep_lds; }
    }
    h.co_tiles = ceil_div(h.Cout_g, BM); h.ci_tiles = ceil_div(h.Cin_g, BN);
    const long tiles = (long)h.co_tiles * h.ci_tiles * d->groups * h.NTG;
    // slices of the batch: about 128 blocks per launch, bounded by the workspace and by one chunk per slice.  Every slice
    // costs a slab written and read back (7 MB for a 768 x 768 x 3 layer) and the weight gradients run beside other streams
    // (the generator's backward chain, the other discriminators), so HALF a chip of longer blocks beats two blocks per CU:
    // train step 8.37 / 8.09 / 7.95 / 7.85 / 7.92 ms for a target of 1024 / 512 / 256 / 128 / 64 blocks (10.4 unsliced)
    const int64_t slab = (int64_t)h.K_eff * h.Cout * h.Cin_g;
    if (!workspace || workspace_floats < slab) return 1;
    const int b_chunks = ceil_div(h.B, h.NB);
    static const char* env_t = getenv("STEGAN_BF16_WGRAD_TARGET");       // blocks to aim for (experiments)
    const long target = env_t ? atol(env_t) : 128;
    long want = (target + tiles - 1) / tiles;
    if (const char* es = STEGAN_ENV("STEGAN_BF16_WGRAD_SLICES")) want = atol(es);
    if (want > b_chunks) want = b_chunks;
    if (want > workspace_floats / slab) want = workspace_floats / slab;
    if (want < 1) want = 1;
    h.bps = ceil_div(b_chunks, (int)want) * h.NB;
    h.slices = ceil_div(h.B, h.bps);
    h.partial = workspace;
    { const char* e1 = STEGAN_ENV("STEGAN_BF16_WGRAD_XCD"); h.xcd = e1 ? atoi(e1) : 1; }
    { const char* e2 = STEGAN_ENV("STEGAN_BF16_WGRAD_ROT"); h.rot = e2 ? atoi(e2) : 1; }
    h.fdNGA = make_fastdiv((uint32_t)h.NGA); h.fdNGX = make_fastdiv((uint32_t)h.NGX);
    h.fdRUp = make_fastdiv((uint32_t)h.RUp); h.fdSEG = make_fastdiv((uint32_t)h.SEG);
#ifdef STEGAN_STAMPS             // diagnostic builds only
    { const char* se = getenv("STEGAN_STAMP_PTR"); h.stamps = se ? (unsigned long long*)strtoull(se, nullptr, 10) : nullptr; }
#else
    h.stamps = nullptr;
#endif
    h.fdRSA = make_fastdiv((uint32_t)h.RSA); h.fdRSX = make_fastdiv((uint32_t)h.RSX);
    h.fdUC = make_fastdiv((uint32_t)h.u_chunks); h.fdCI = make_fastdiv((uint32_t)h.ci_tiles);
    h.fdCO = make_fastdiv((uint32_t)h.co_tiles); h.fdTG = make_fastdiv((uint32_t)h.NTG);
    {
        static const char* dbg = getenv("STEGAN_PLAN_DEBUG");
        if (dbg) fprintf(stderr, "plan-wgrad-bf16 B=%d Cin_g=%d Cout_g=%d K=%d(live %d) s=%d d=%d T=%d->%d | tile %dx%dx%d NTG=%d NB=%d RU=%d R=%d SEG=%d RSA=%d RSX=%d ait=%d xit=%d lds=%zu tiles=%ld slices=%d xv=%d\n",
                         d->B, h.Cin_g, h.Cout_g, d->K, h.K_eff, d->stride, d->dilation, d->Tin, d->Tout, BM, BN, KT, h.NTG, h.NB,
                         h.RU, h.R, h.SEG, h.RSA, h.RSX, h.ait, h.xit, lds, tiles, h.slices, (int)xv);
    }
    // the bias gradient rides in the same grid (blocks past the tile blocks); an unsliced reduction adds straight into dW
    BiasW bw;
    bw.dy = dy; bw.db = db; bw.B = d->B; bw.C = d->Cout; bw.T = d->Tout;
    bw.slices = db ? bias_grad_slices(d->B, d->Cout, d->Tout) : 0;
    bw.per = db ? ceil_div(d->B, bw.slices) : 1;
    bw.slices = db ? ceil_div(d->B, bw.per) : 0;
    bw.vec = (d->Tout % 4 == 0) && ((uintptr_t)dy % 16 == 0);
    bw.fdT = make_fastdiv((uint32_t)(bw.vec ? d->Tout / 4 : d->Tout));
    bw.first = (int)(tiles * h.slices);
    static const bool no_direct = getenv("STEGAN_BF16_WGRAD_NO_DIRECT") != nullptr;      // experiments
    float* dw_direct = (h.slices == 1 && !no_direct) ? dw : nullptr;
    dim3 grid((unsigned)(tiles * h.slices + (long)bw.slices * d->Cout));
    if (KT == 1) {
        if (TM == 2 && TN == 2) wgh_launch<2, 2, 1>(xv, grid, lds, s, h, dw_direct, bw);
        else if (TM == 2) wgh_launch<2, 1, 1>(xv, grid, lds, s, h, dw_direct, bw);
        else if (TN == 2) wgh_launch<1, 2, 1>(xv, grid, lds, s, h, dw_direct, bw);
        else wgh_launch<1, 1, 1>(xv, grid, lds, s, h, dw_direct, bw);
    } else if (KT == 3) {
        if (TM == 2 && TN == 2) {
            static bool attr223 = false;
            if (!attr223) {
                (void)hipFuncSetAttribute((const void*)conv_wgrad_bf16_img_kernel<2, 2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                attr223 = true;
            }
            STEGAN_TIMED_LAUNCH((conv_wgrad_bf16_img_kernel<2, 2, 3>), grid, dim3(512), lds, s, h, dw_direct, bw);
        }
        else if (TM == 2) wgh_launch<2, 1, 3>(xv, grid, lds, s, h, dw_direct, bw);
        else wgh_launch<1, 1, 3>(xv, grid, lds, s, h, dw_direct, bw);
    } else if (KT == 5) wgh_launch<1, 1, 5>(xv, grid, lds, s, h, dw_direct, bw);
    else wgh_launch<1, 1, 8>(xv, grid, lds, s, h, dw_direct, bw);
    if (!dw_direct && deferred) {
        // the caller sums the slabs later, together with those of other layers (stegan_wgrad_reduce_multi)
        deferred->partial = h.partial; deferred->dw = dw; deferred->per_tap = (int64_t)h.Cout * h.Cin_g;
        deferred->slices = h.slices; deferred->K = d->K; deferred->k_lo = h.k_lo; deferred->K_eff = h.K_eff;
        deferred->used_floats = slab * h.slices;
    } else if (!dw_direct) {
        const long per_tap = (long)h.Cout * h.Cin_g;
        long rb = (per_tap * h.K_eff + 255) / 256;
        if (rb > 2048) rb = 2048;
        BiasArgs ba;
        ba.dy = dy; ba.db = nullptr; ba.B = d->B; ba.C = d->Cout; ba.T = d->Tout; ba.slices = 0; ba.per = 1; ba.vec = 0;
        ba.fdT = make_fastdiv(1u);
        hipLaunchKernelGGL(conv_wgrad_bf16_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, s,
                           h.partial, dw, per_tap, h.slices, d->K, h.k_lo, h.K_eff, (int)rb, ba);
    }
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_conv1d_bwd_weight_hx(const stegan_conv_desc* d, const float* x, const uint16_t* xh, const float* dy,
                                           const uint16_t* dyh, float* dw, float* db, float* workspace,
                                           int64_t workspace_floats, void* stream) {
    return wgh_run(d, x, xh, dy, dyh, dw, db, workspace, workspace_floats, nullptr, stream);
}

extern "C" int stegan_conv1d_bwd_weight_hxd(const stegan_conv_desc* d, const float* x, const uint16_t* xh, const float* dy,
                                            const uint16_t* dyh, float* dw, float* db, float* workspace,
                                            int64_t workspace_floats, stegan_wgrad_reduce* deferred, void* stream) {
    STEGAN_CHECK_ARG(deferred);
    return wgh_run(d, x, xh, dy, dyh, dw, db, workspace, workspace_floats, deferred, stream);
}

// the chained form: `prev` (nullable / slices == 0: none) is the pending sum an earlier call on the SAME stream left in its
// `next`; it is run at the head of this launch (inside the tile kernel when it takes the shape, else as a reduce launch of its
// own), so a chain of n sliced layers costs one reduce launch -- its tail's -- instead of n.  Return 1: nothing was enqueued,
// `prev` is still pending.
extern "C" int stegan_conv1d_bwd_weight_hxc(const stegan_conv_desc* d, const float* x, const uint16_t* xh, const float* dy,
                                            const uint16_t* dyh, float* dw, float* db, float* workspace,
                                            int64_t workspace_floats, const stegan_wgrad_reduce* prev, stegan_wgrad_reduce* next,
                                            void* stream) {
    STEGAN_CHECK_ARG(next && next != prev);
    int prev_done = 0;
    const int rc = wgh_run(d, x, xh, dy, dyh, dw, db, workspace, workspace_floats, next, stream, prev, &prev_done);
    if ((rc == STEGAN_OK || rc == STEGAN_OK_TILE) && prev && prev->slices > 0 && !prev_done) {
        const int r2 = stegan_wgrad_reduce_multi(1, prev, stream);
        if (r2) return r2;
    }
    return rc;
}

// Up to 5 layers' weight gradients in ONE launch of the tile kernel (the five convs of a generator block, issued together by
// ops._wgrad_flush).  All or nothing: return 1 (nothing enqueued, `prev` untouched) unless EVERY layer is a shape for the tile
// kernel (stride 1, ungrouped, both operands as images, plan admits it); the caller then goes layer by layer.  On success
// next[i] describes layer i's pending slab sum (slices == 0: none) and every prev[j] has been dealt with: prev[j], j < n, rides
// in problem j's prologue when it is the 16-byte case, the others are summed by one reduce launch.
extern "C" int stegan_conv1d_bwd_weight_multi_hxc(int32_t n, const stegan_conv_desc* const* descs, const float* const* x,
                                                  const uint16_t* const* xh, const float* const* dy, const uint16_t* const* dyh,
                                                  float* const* dw, float* const* db, float* const* workspace,
                                                  const int64_t* workspace_floats, int32_t n_prev,
                                                  const stegan_wgrad_reduce* prev, stegan_wgrad_reduce* next, void* stream) {
    STEGAN_CHECK_ARG(n >= 1 && n <= 5 && descs && x && xh && dy && dyh && dw && workspace && workspace_floats && next);
    STEGAN_CHECK_ARG(n_prev >= 0 && (n_prev == 0 || prev));
    static const bool no_xh = getenv("STEGAN_BF16_NO_XH") != nullptr;
    if (no_xh) return 1;
    TileWgradArgs ta[5];
    int k_lo[5], K_eff[5];
    bool rides[5] = {false, false, false, false, false};
    for (int i = 0; i < n; ++i) {
        const stegan_conv_desc* d = descs[i];
        STEGAN_CHECK_ARG(d && x[i] && dy[i] && dw[i]);
        const int rc = check_desc(d);
        if (rc) return rc;
        if (!xh[i] || !dyh[i] || d->stride != 1 || d->groups != 1 || d->Cin % 8 || d->Cout % 8) return 1;
        if (d->in_act != STEGAN_ACT_NONE && d->in_act != STEGAN_ACT_RELU) return 1;
        if ((int64_t)d->B * (d->Cin / 8) * d->Tin * 16 >= (int64_t)OOBW || (int64_t)d->B * (d->Cout / 8) * d->Tout * 16 >= (int64_t)OOBW) return 1;
        K_eff[i] = wgh_live_taps(d, k_lo[i]);
        if (K_eff[i] <= 0) return 1;
        TileWgradArgs& t = ta[i];
        t.xh = xh[i]; t.dyh = dyh[i]; t.dy = dy[i]; t.dw = dw[i]; t.db = db ? db[i] : nullptr;
        t.partial = workspace[i]; t.partial_floats = workspace_floats[i];
        t.B = d->B; t.Cin = d->Cin; t.Cout = d->Cout; t.K = d->K; t.dil = d->dilation; t.pad = d->pad; t.Tin = d->Tin;
        t.Tout = d->Tout; t.ups = d->upsample == 2 ? 1 : 0; t.relu = d->in_act == STEGAN_ACT_RELU; t.k_lo = k_lo[i]; t.Keff = K_eff[i];
        t.bias_in_reduce = 0;
        t.prev_partial = nullptr; t.prev_dw = nullptr; t.prev_n4 = 0; t.prev_slices = 0;
        if (i < n_prev) {
            const stegan_wgrad_reduce& pv = prev[i];
            if (pv.slices > 0 && pv.K_eff == pv.K && pv.k_lo == 0 && ((pv.per_tap * pv.K_eff) & 3) == 0 &&
                (((uintptr_t)pv.partial | (uintptr_t)pv.dw) & 15) == 0) {
                t.prev_partial = pv.partial; t.prev_dw = pv.dw; t.prev_n4 = pv.per_tap * pv.K_eff / 4; t.prev_slices = pv.slices;
                rides[i] = true;
            }
        }
    }
    int sl[5] = {0, 0, 0, 0, 0}, bias_done[5] = {0, 0, 0, 0, 0};
    const int trc = stegan_tile_wgrad_multi(n, ta, sl, bias_done, (hipStream_t)stream);
    if (trc) return trc;                                    // 1: declined as a whole; < 0: launch error
    for (int i = 0; i < n; ++i) {
        stegan_wgrad_reduce& nx = next[i];
        nx.partial = nullptr; nx.dw = nullptr; nx.per_tap = 0; nx.used_floats = 0; nx.slices = 0; nx.K = 0; nx.k_lo = 0; nx.K_eff = 0;
        if (sl[i] > 1) {
            nx.partial = workspace[i]; nx.dw = dw[i]; nx.per_tap = (int64_t)descs[i]->Cout * descs[i]->Cin;
            nx.slices = sl[i]; nx.K = descs[i]->K; nx.k_lo = k_lo[i]; nx.K_eff = K_eff[i];
            nx.used_floats = (int64_t)K_eff[i] * descs[i]->Cout * descs[i]->Cin * sl[i];
        }
        if (db && db[i] && !bias_done[i]) return STEGAN_E_ARG;      // (cannot happen: the multi plan refuses bias blocks)
    }
    // the pending sums that did not ride
    stegan_wgrad_reduce rest[64];
    int nr = 0;
    for (int j = 0; j < n_prev && nr < 64; ++j)
        if (prev[j].slices > 0 && !(j < n && rides[j])) rest[nr++] = prev[j];
    if (nr) {
        const int r2 = stegan_wgrad_reduce_multi(nr, rest, stream);
        if (r2) return r2;
    }
    return STEGAN_OK_TILE;
}

// ---- the deferred slab sums of up to WRM_MAX layers in ONE launch (52 reduce launches a step become one per phase)
namespace {
constexpr int WRM_MAX = 48;
struct WRMulti {
    const float* partial[WRM_MAX];
    float* dw[WRM_MAX];
    long per_tap[WRM_MAX];
    int slices[WRM_MAX], K[WRM_MAX], k_lo[WRM_MAX], K_eff[WRM_MAX];
    int first[WRM_MAX + 1];           // first block of every item; first[n] = grid size
    int n;
};

__global__ __launch_bounds__(256) void conv_wgrad_bf16_reduce_multi_kernel(const WRMulti a) {
    int it = 0;
#pragma unroll 1
    for (int i = 1; i < a.n; ++i)
        if ((int)blockIdx.x >= a.first[i]) it = i;
    const int rb = a.first[it + 1] - a.first[it], blk = (int)blockIdx.x - a.first[it];
    const float* __restrict__ partial = a.partial[it];
    float* __restrict__ dw = a.dw[it];
    const int K_eff = a.K_eff[it], K = a.K[it], k_lo = a.k_lo[it], slices = a.slices[it];
    const long total = a.per_tap[it] * K_eff;
    if (K_eff == K && (total & 3) == 0 && (((uintptr_t)partial | (uintptr_t)dw) & 15) == 0) {     // as the per-layer kernel
        const long n4 = total >> 2;
        const f32x4* src4 = (const f32x4*)partial;
        f32x4* dw4 = (f32x4*)dw;
        for (long i = (long)blk * 256 + threadIdx.x; i < n4; i += (long)rb * 256) {
            const f32x4 o = dw4[i];
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            int s = 0;
            for (; s + 4 <= slices; s += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] += src4[(size_t)(s + u) * n4 + i];
            }
            for (; s < slices; ++s) v[0] += src4[(size_t)s * n4 + i];
            dw4[i] = o + ((v[0] + v[1]) + (v[2] + v[3]));
        }
        return;
    }
    for (long i = (long)blk * 256 + threadIdx.x; i < total; i += (long)rb * 256) {
        const long r = i / K_eff;
        const int kk = (int)(i - r * K_eff);
        const float* src = partial + i;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        int s = 0;
        for (; s + 4 <= slices; s += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] += src[(size_t)(s + u) * total];
        }
        for (; s < slices; ++s) v[0] += src[(size_t)s * total];
        dw[r * K + k_lo + kk] += (v[0] + v[1]) + (v[2] + v[3]);       // the summation order of the per-layer reduce kernel
    }
}
}  // namespace

extern "C" int stegan_wgrad_reduce_multi(int32_t n, const stegan_wgrad_reduce* items, void* stream) {
    STEGAN_CHECK_ARG(n >= 0 && (n == 0 || items));
    hipStream_t s = (hipStream_t)stream;
    if (n == 1 && items[0].slices > 0) {                   // one layer: the per-layer kernel (16 bytes per lane when every tap is live)
        const stegan_wgrad_reduce& it = items[0];
        STEGAN_CHECK_ARG(it.partial && it.dw && it.per_tap > 0 && it.K_eff > 0);
        long rb = (it.per_tap * it.K_eff + 255) / 256;
        if (rb > 2048) rb = 2048;
        BiasArgs ba;
        ba.dy = nullptr; ba.db = nullptr; ba.B = 0; ba.C = 0; ba.T = 0; ba.slices = 0; ba.per = 1; ba.vec = 0;
        ba.fdT = make_fastdiv(1u);
        hipLaunchKernelGGL(conv_wgrad_bf16_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, s, it.partial, it.dw, (long)it.per_tap,
                           (int)it.slices, (int)it.K, (int)it.k_lo, (int)it.K_eff, (int)rb, ba);
        STEGAN_LAUNCH_CHECK();
        return STEGAN_OK;
    }
    for (int base = 0; base < n; base += WRM_MAX) {
        WRMulti a;
        a.n = 0; a.first[0] = 0;
        for (int i = base; i < n && a.n < WRM_MAX; ++i) {
            const stegan_wgrad_reduce& it = items[i];
            if (it.slices <= 0) continue;
            STEGAN_CHECK_ARG(it.partial && it.dw && it.per_tap > 0 && it.K_eff > 0);
            long rb = (it.per_tap * it.K_eff + 255) / 256;
            if (rb > 1024) rb = 1024;
            const int j = a.n++;
            a.partial[j] = it.partial; a.dw[j] = it.dw; a.per_tap[j] = (long)it.per_tap; a.slices[j] = it.slices;
            a.K[j] = it.K; a.k_lo[j] = it.k_lo; a.K_eff[j] = it.K_eff;
            a.first[j + 1] = a.first[j] + (int)rb;
        }
        if (a.n == 0) continue;
        hipLaunchKernelGGL(conv_wgrad_bf16_reduce_multi_kernel, dim3((unsigned)a.first[a.n]), dim3(256), 0, s, a);
    }
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_conv1d_bwd_weight_h(const stegan_conv_desc* d, const float* x, const float* dy, float* dw, float* db,
                                          float* workspace, int64_t workspace_floats, void* stream) {
    return stegan_conv1d_bwd_weight_hx(d, x, nullptr, dy, nullptr, dw, db, workspace, workspace_floats, stream);
}
