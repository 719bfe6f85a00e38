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
//   * X tile  Xs[cb][slot][8 channels] bf16 (channel-block-major: one 8-channel block of all slots, then the next) --
//     activations stay fp32 (B, C, T) in HBM; a thread loads the 8 channels of one (8-channel block, run of 4 time slots)
//     with 8 buffer_load_dwordx4 (lanes along time: whole 128-byte lines; the channel stride is a scalar offset; halo /
//     padding / out-of-range slots carry an out-of-range offset and read 0 from the buffer unit; the forward's nearest-x2
//     upsample loads 4 SOURCE positions and stores every converted piece twice; rows that are not float4-aligned take 8
//     buffer_load_dword per slot, or -- opt-in -- the same 16-byte loads with the row's tail zeroed), applies the input ReLU on the
//     packed halves (v_pk_max_i16 x, 0), converts with v_cvt_pk_bf16_f32 and stores ds_write_b128.  A tap shift is a SLOT
//     offset, so any dilation / stride is an address offset, never a gather; operand reads of 32 consecutive slots are 32
//     consecutive 16-byte pieces: bank-conflict free without padding or swizzle.
//   * W tile  Ws[tap][cb][m][8 channels] bf16 from a bf16 copy of the direction tensor laid out [g][k][cb][m][8] (zero
//     padded to 128 rows / 64 channels, plus one all-zero tap plane), produced once per weight version by
//     stegan_conv_prep_bf16: staged by LDS-DMA (global_load_lds, 16 B per lane), every wave-instruction a contiguous
//     1 KiB of global memory landing as a contiguous 1 KiB of the LDS image.
//   * pipeline: chunk c + 1 is in flight while chunk c is multiplied (X loads held in registers and converted / stored
//     behind the MFMA loop, W by LDS-DMA), two LDS buffers per operand, one barrier per chunk; two blocks share a CU.
//   * the reduction runs over "atoms" = (tap, 8-channel block); one MFMA consumes two atoms (lane halves): with >= 16
//     channels per chunk the halves take neighbouring channel blocks of one tap, with 8-channel chunks (first layers,
//     narrow groups, many-tap layers where the W tile must stay small) they take neighbouring TAPS.
//   * block = 8 waves with fixed roles: 4 consumer waves (2 x 2, wave tile 32 TM x 32 TN, block tile up to 128 x 128) only
//     multiply, 4 producer waves only stage, one of each per SIMD (see the kernel body); the X fragment is the MFMA's A
//     operand, so an accumulator tile holds its output channel on the lane and 4 runs of 4 consecutive time steps in
//     its registers: the epilogue loads / stores 16 bytes per lane; split-K and its fixed-order epilogue kernel are
//     shared with the fp32 kernel; blocks are mapped to tiles XCD-aware (M tiles fastest inside one XCD's range).
//
// Numerics: products of bf16-rounded operands accumulated in fp32 (k-ordered chain inside the MFMA); everything around
// the contraction (scale, bias, residual, activations, masks) is fp32 exactly as in conv_direct.hip.
#include "conv_direct_common.h"
#include "conv_tile.h"

namespace {

typedef hop_t bf16x8 __attribute__((ext_vector_type(8)));
typedef hop_t bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int HX_MAX = 6;      // X items (8 channels x 1 slot -> one 16-byte LDS store) per thread and chunk, scalar loads
constexpr int HXV_MAX = 2;     // X items (8 channels x 4 slots -> four LDS stores) per thread and chunk, 16-byte loads
constexpr int HW_MAX = 12;     // W 16-byte DMA pieces per thread and chunk
constexpr unsigned OOB = 0x7ffff000u;   // buffer offset beyond any source tensor (host checks bytes < 2^31 - 2^13): reads 0

struct HP {
    DirP d;                       // CC = channels per chunk (8 NCB); XROW etc. as in the fp32 kernel
    const unsigned short* wh;     // bf16 [G][K + 1][Mpad][Cpad]; plane K is all zero
    int Mpad, Cpad;
    int NG;                       // X staging items per 8-channel block: XROW (scalar loads) or XROW / 4 (16-byte loads)
    FastDiv fdNG;
    int ntapsW;                   // taps in the W tile = max over phases (rounded up to even when NCB == 1)
    int xit, wit;                 // staging items per thread
    unsigned src_bytes;
    int vec_out;                  // 16-byte epilogue accesses: unit output stride, columns and rows multiples of 4, aligned
    int dbg;                      // diagnostic ablations (STEGAN_BF16_DBG bits: 1 no X loads, 2 no W DMA, 4 no MFMA loop, 8 no
                                  // X LDS stores); results are wrong when set
    // round 3: the source as a bf16 activation image [B][xh_cb][Tsrc][8 channels] (written by the producing kernel's
    // epilogue, DirP::out_h): the X tile is then staged by LDS-DMA like the weights -- no loads into registers, no
    // conversion, no LDS stores; null: the fp32 source is converted while it is staged (kernel template parameter XH)
    const unsigned short* xh = nullptr;
    int xh_cb = 0;                // channel blocks per batch element of the image = ceil(Csrc / 8)
    unsigned xh_bytes = 0;
    int xnit = 0;                   // X DMA wave-instructions per producer wave and chunk = ceil(NCB * XROW / 256)
    int nstage = 2;               // image path: LDS stages of the chunk ring (nstage - 1 chunks in flight)
    // block -> tile placement over the 8 XCDs: the tile grid (ny M tiles x xR "rest" ids = N tiles x phases x splits) is cut
    // into 2^xlgm x 2^(3 - xlgm) rectangles, one per XCD; xlgm = 0 is the round-2 placement (all M tiles of a few N tiles per
    // XCD: X once, W by every XCD), larger xlgm trades re-reads of X for fewer re-reads of W.  xlgm < 0: identity (experiments)
    int xlgm = 0, xR = 1;
    // input gradient of a strided layer: the `stride` output phases of a tile write interleaved columns of the SAME lines of dx
    // and read the same dy tile / mask rows.  With the phase as the SLOWEST part of the id (rounds 2-4) an XCD worked on one
    // phase: every line of dx left three L2s as three partial writes (2.7 x the algorithmic write bytes, 4-6 x the reads:
    // profiles/r04_pmc_by_shape.txt).  nph > 1: the phase is the FASTEST part of the "rest" id, so a tile's phases run on one
    // XCD within a few blocks of each other and meet in its L2
    int nph = 1;
    FastDiv fdPH;
    int rot = 1;                  // image path: rotated chunk walk per block (STEGAN_BF16_ROT=0: every block starts at chunk 0)
    int cw0 = 1;                  // image path: the consumer waves request chunk 0's weights (STEGAN_BF16_CW0=0: the producers)
};

// a scalar moved into a vector register through an opaque instruction: the value cannot be re-derived (re-loaded) later
__device__ __forceinline__ int vg(int s) { int v; asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s)); return v; }
__device__ __forceinline__ unsigned vg(unsigned s) { unsigned v; asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s)); return v; }

#ifndef STEGAN_BF16_SCHED
#define STEGAN_BF16_SCHED 1
#endif
constexpr bool SCHED_ON = STEGAN_BF16_SCHED != 0;
constexpr int HXD_MAX = 8;     // X DMA wave-instructions per producer wave and chunk (8 x 256 pieces of 16 bytes = 32 KiB)

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b, bool relu) {
    const f32x2 f = {a, b};
    s16x2 s = __builtin_bit_cast(s16x2, __builtin_convertvector(f, bf16x2));
    if (relu) s = __builtin_elementwise_max(s, (s16x2)(0));       // sign bit set <=> negative int16: relu on packed halves
    return __builtin_bit_cast(unsigned, s);
}

// XV: 16-byte X loads (4 consecutive source positions per lane; float4-aligned rows by default, see xv_eligible: a load
// that straddles the end of a ragged row has its tail zeroed before the conversion) (with the
// forward's nearest-x2 upsample one load feeds 8 slots: the first version sent those convs to 4-byte loads, 49 vs ~17 us);
// every segment then starts at the source position rounded DOWN to a multiple of 4 and the rounding (0..3 slots) is
// added to the lanes' column bases instead (the fp32 kernel's XVEC idea).
#ifdef STEGAN_BF16_ABLATE       // diagnostic builds: STEGAN_BF16_DBG bits switch parts of the kernel off (results wrong)
#define HB_DBG(bit_) (h.dbg & (bit_))
#else
#define HB_DBG(bit_) false
#endif

// s_waitcnt vmcnt(n) with a run-time (wave-uniform) n: the instruction takes an immediate, so one case per value.  Waiting for
// FEWER outstanding operations than asked is always safe (it only waits longer): counts above the table wait for 36.
__device__ __forceinline__ void wait_vmcnt(int n) {
#define WV_CASE(k_) case k_: asm volatile("s_waitcnt vmcnt(" #k_ ")" ::: "memory"); break;
    switch (n) {
        WV_CASE(0) WV_CASE(1) WV_CASE(2) WV_CASE(3) WV_CASE(4) WV_CASE(5) WV_CASE(6) WV_CASE(7) WV_CASE(8) WV_CASE(9)
        WV_CASE(10) WV_CASE(11) WV_CASE(12) WV_CASE(13) WV_CASE(14) WV_CASE(15) WV_CASE(16) WV_CASE(17) WV_CASE(18)
        WV_CASE(19) WV_CASE(20) WV_CASE(21) WV_CASE(22) WV_CASE(23) WV_CASE(24) WV_CASE(25) WV_CASE(26) WV_CASE(27)
        WV_CASE(28) WV_CASE(29) WV_CASE(30) WV_CASE(31) WV_CASE(32) WV_CASE(33) WV_CASE(34) WV_CASE(35)
        default: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
    }
#undef WV_CASE
}

template <int TM, int TN, int LOGNCB, bool XV, bool XH = false>
__device__ __forceinline__ void conv_bf16_body(const HP& h, const unsigned bid_x, const unsigned bid_y, const unsigned bid_z) {
    static_assert(!(XV && XH), "the image path has no use for the 16-byte fp32 loads");
    const DirP& p = h.d;
    constexpr int NCB = 1 << LOGNCB;
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int KS = NCB >= 2 ? NCB / 2 : 1;        // MFMA k-steps per tap (NCB >= 2) / per tap pair (NCB == 1)
    extern __shared__ __align__(16) char smem[];      // W ring (2 chunks) | X ring (2 chunks)

    // 512 threads = 8 waves, two per SIMD, with fixed roles: waves 0-3 ("consumers") only read MFMA operands from LDS and
    // multiply; waves 4-7 ("producers") only stage -- X loads, W LDS-DMA, bf16 conversion, LDS stores.  One wave alone on a
    // SIMD runs its instruction streams back to back: measured per iteration of a 128 x 128 x 32-channel chunk
    // (scripts/bf16_stamps.py, STEGAN_STAMPS_ITER) 912 cycles to issue the 14 vector-memory instructions, 792 for the 24
    // MFMAs (exactly their pipe time), 332 waiting for the loads and 788 to convert and store: the matrix pipe idled 72 % of
    // the loop.  With the roles on different waves of each SIMD the staging of chunk c + 1 runs beside the MFMAs of chunk c.
    const int tid = threadIdx.x & 255, lane = tid & 63;           // index inside the role's 256 threads
    const bool producer = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) != 0;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave inside the role
    const int wm = wave >> 1, wn = wave & 1;
#ifdef STEGAN_STAMPS          // diagnostic builds only (scripts/bf16_stamps.py): 8 s_memtime stamps per block
#define HSTAMP(i_) if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)bid_x * 8 + (i_)] = __builtin_amdgcn_s_memtime();
// producer wave 0 (thread 256): a second table of 8 stamps per block, 1 Mi entries behind the consumers' table
#define PSTAMP(i_) if (p.stamps && threadIdx.x == 256) p.stamps[(1u << 19) + (size_t)bid_x * 8 + (i_)] = __builtin_amdgcn_s_memtime();
// ... and a third one for the stretch between kernel entry and the first DMA
#define QSTAMP(i_) if (p.stamps && threadIdx.x == 256) p.stamps[(1u << 19) + (1u << 18) + (size_t)bid_x * 8 + (i_)] = __builtin_amdgcn_s_memtime();
#else
#define HSTAMP(i_)
#define PSTAMP(i_)
#define QSTAMP(i_)
#endif
    HSTAMP(0)
    PSTAMP(6)                                                        // producer wave 0 enters
    // one clustered batch of kernarg loads (left alone, hipcc sinks each s_load next to its first use: a string of
    // dependent load + wait round trips through the index setup)
    asm volatile("" ::"s"(p.B), "s"(p.Csrc), "s"(p.Cdst), "s"(p.Tsrc), "s"(p.Tq), "s"(p.Tdst), "s"(p.C2_g), "s"(p.M_g), "s"(p.K),
                 "s"(p.stride), "s"(p.mode), "s"(p.m_tiles), "s"(p.SEG), "s"(p.NSEG), "s"(p.XROW), "s"(p.U0), "s"(p.splits),
                 "s"(p.chunks_per_split), "s"(p.nbx), "s"(p.ny), "s"(p.ups), "s"(p.in_relu), "s"(h.Mpad), "s"(h.Cpad), "s"(h.NG),
                 "s"(h.ntapsW), "s"(h.xit), "s"(h.wit), "s"(h.src_bytes), "s"(h.vec_out));
    // round 3: the fields of the image path and every magic divisor of the index setup in the same batch (the ISA of the
    // first image build showed eight dependent s_load / s_waitcnt round trips between kernel entry and the first DMA)
    asm volatile("" ::"s"(h.xh), "s"(h.wh), "s"(h.xh_cb), "s"(h.xh_bytes), "s"(h.xnit), "s"(h.nstage), "s"(p.out_h), "s"(p.out_cb),
                 "s"(p.fdNY.mul), "s"(p.fdNY.shift), "s"(p.fdNBX.mul), "s"(p.fdNBX.shift), "s"(p.fdSP.mul), "s"(p.fdSP.shift),
                 "s"(p.fdMT.mul), "s"(p.fdMT.shift), "s"(p.fdXROW.mul), "s"(p.fdXROW.shift), "s"(p.fdSEG.mul), "s"(p.fdSEG.shift),
                 "s"(p.fdU0.mul), "s"(p.fdU0.shift), "s"(p.fdU1.mul), "s"(p.fdU1.shift), "s"(p.src), "s"(p.out), "s"(p.partial));
    // 1-D grid.  Blocks are dealt round-robin over the 8 XCDs (each with its own 4 MiB L2), so linear block L is first
    // mapped to a logical tile id such that every XCD works on ONE contiguous range of ids, and ids run M-tile fastest:
    // the blocks resident on an XCD at any time then cover a few N tiles x all M tiles of a layer -- each X chunk is
    // fetched into that L2 once and reused by the M tiles, the W chunks of all M tiles stay resident across its N tiles
    // (with the natural x-fastest order every XCD streams the whole of X and W through its L2: measured 3-4 TB/s of
    // Infinity-Cache traffic bound the kernel).  Placement only affects speed, never results.
    QSTAMP(0)                                                        // kernel arguments have arrived
    int bx, by, phase, split;
    const int nsplits = p.splits, cps = p.chunks_per_split;
    {
        const unsigned total = bid_y /* blocks of this problem */, L = bid_x;
        const unsigned q8 = total >> 3, r8 = total & 7, xcd = L & 7;
        unsigned t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (L >> 3);
        unsigned rest;
        if (h.xlgm <= 0) {
            if (h.xlgm < 0) t = L;
            rest = fd_div(t, p.fdNY);
            by = (int)(t - rest * p.ny);
        } else {
            // ids enumerate the XCD rectangles one after the other (rest part j slowest, M part i, then the rectangle's tiles
            // M fastest); the contiguous id range of an XCD is then (up to the remainders) one rectangle.  All scalar work.
            const unsigned lgm = (unsigned)h.xlgm, lgn = 3u - lgm, ny = (unsigned)p.ny, R = (unsigned)h.xR;
            unsigned j = 0, i = 0;
            for (unsigned jj = 1; jj < (1u << lgn); ++jj) if (((jj * R) >> lgn) * ny <= t) j = jj;
            const unsigned rb0 = (j * R) >> lgn, hr = (((j + 1) * R) >> lgn) - rb0, t2 = t - rb0 * ny;
            for (unsigned ii = 1; ii < (1u << lgm); ++ii) if (hr * ((ii * ny) >> lgm) <= t2) i = ii;
            const unsigned mb0 = (i * ny) >> lgm, hm = (((i + 1) * ny) >> lgm) - mb0, t3 = t2 - hr * mb0;
            const unsigned q = t3 / hm;
            by = (int)(mb0 + (t3 - q * hm));
            rest = rb0 + q;
        }
        if (h.nph > 1) {
            const unsigned r2 = fd_div(rest, h.fdPH);
            phase = (int)(rest - r2 * (unsigned)h.nph);
            const unsigned z = fd_div(r2, p.fdNBX);
            bx = (int)(r2 - z * p.nbx);
            split = (int)z;
        } else {
            const unsigned z = fd_div(rest, p.fdNBX);
            bx = (int)(rest - z * p.nbx);
            phase = (int)fd_div(z, p.fdSP);
            split = (int)z - phase * p.splits;
        }
    }
    const int g = (int)fd_div((uint32_t)by, p.fdMT);
    const int m0 = (by - g * p.m_tiles) * BM;
    QSTAMP(1)                                                        // tile decoded

    // the phase's six table entries in ONE batch of scalar loads (they sat behind two branches: two more round trips)
    Taps tp;
    tp.ntaps = p.ph_ntaps[phase]; tp.kbase = p.ph_kbase[phase]; tp.kstep = p.ph_kstep[phase];
    tp.off_min = p.ph_offmin[phase]; tp.step = p.ph_step[phase];
    const int phU = p.ph_U[phase];
    asm volatile("" ::"s"(tp.ntaps), "s"(tp.kbase), "s"(tp.kstep), "s"(tp.off_min), "s"(tp.step), "s"(phU));
    int U, cs, os;
    if (p.mode == 0) { U = p.U0; cs = p.stride; os = 1; }
    else { U = phU; cs = 1; os = p.stride; }
    const int N = p.B * U;
    const int n0 = bx * BN;
    if (n0 >= N) return;
    const FastDiv fdU = (U == p.U0) ? p.fdU0 : p.fdU1;
    const int ntaps = tp.ntaps, off_min = tp.off_min;
    const int ntapsW = NCB == 1 ? ((ntaps + 1) & ~1) : ntaps;       // NCB == 1: MFMA halves take neighbouring taps
    const int S = NCB == 1 ? (ntapsW >> 1) : ntaps * KS;            // MFMA k-steps per chunk

    // ---- chunk-invariant part of the X staging.  Item = (8-channel block cg, slot or run of 4 slots), slots fastest:
    // the lanes of a wave-instruction read consecutive time positions of one channel row (whole 128-byte lines) and store
    // consecutive 16-byte pieces of the [cb][slot] LDS image.  The item's source offset (batch row, first channel of its
    // block, time) is a per-lane constant; the chunk's channel base and the channel within the block are a SCALAR offset.
    const int b_lo = (int)fd_div((uint32_t)n0, fdU), u0 = n0 - b_lo * U;
    const int qbase0 = u0 * cs + off_min, qbaseN = off_min;
    // XV with the nearest-x2 upsample (forward only): one aligned 16-byte load of 4 SOURCE positions feeds 8 slots, so the
    // granule of the segment alignment and of an item is 8 slots instead of 4
    const int xsh = XV ? 2 + ((p.mode == 0 && p.ups) ? 1 : 0) : 0;       // log2(slots per X item)
    const int xgm = (1 << xsh) - 1;
    const int qal0 = XV ? (qbase0 & ~xgm) : qbase0, qalN = XV ? (qbaseN & ~xgm) : qbaseN;
    const int CCH = 8 * NCB;
    const int nchunks = ntaps > 0 ? (p.C2_g + CCH - 1) / CCH : 0;
    const int c_begin = split * cps;
    const int c_end = min(nchunks, c_begin + cps);
    const char* wt_g = (const char*)h.wh + (size_t)g * (p.K + 1) * h.Mpad * h.Cpad * 2;
    // (the X stage is padded to whole waves of 64 pieces: the lean DMA issue of the image path lets the lanes past the tile
    // write their zeros there instead of masking them)
    const int wbuf = h.ntapsW * NCB * BM * 16, xbuf = ((NCB * p.XROW + 63) & ~63) * 16;
    const int NS = XH ? h.nstage : 2;                              // stages of the chunk ring
    char* const Xbufs = smem + NS * wbuf;                          // W ring | X ring
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, (int)h.src_bytes, 0x00020000);
    const bool relu = p.in_relu != 0;
    const int row_bytes = p.Tsrc * 4;
    const int chunk_w = NCB * h.Mpad * 16;                         // bytes between the weights of neighbouring chunks


    // image path, two-stage ring: the consumers request chunk 0's weights (h.cw0; STEGAN_BF16_CW0=0: the producers do)
    const bool cw0 = XH && h.cw0 != 0 && h.nstage == 2;
    // accumulators: multiplied in the consumer waves; for the epilogue each consumer hands the second half of every tile to
    // the producer wave at its position (round 4, below), so both roles own the variable
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int hl = lane >> 5, l31 = lane & 31;
    // ================= producers (waves 4-7): stage chunk c + 1 while the consumers multiply chunk c =================
    // (each role computes only its own per-lane state inside its branch: register demand is the maximum of the two
    // roles, not their sum)
    if (producer) {
    QSTAMP(2)                                                        // phase table, chunk range, ... : role-independent setup done
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    // ---- W staging: piece = it * 256 + tid -> LDS byte 16 * piece; piece = (tap * NCB + cb) * BM + m.  The bf16 weights
    // are stored [g][k][cb][m][8 channels], so the 64 lanes of a wave-instruction copy 1 KiB of CONTIGUOUS global memory
    const int CB = h.Cpad >> 3;
    int w_off[HW_MAX];            // byte offset inside the group's weights, without the chunk's channel-block term; -1: no piece
    /* BRANCH-FREE on purpose (round 4).  The first form guarded every item with `if (it < h.wit)`: hipcc keeps w_off[] as  \
       ONE 12-register tuple, so every guarded write became a phi of whole tuples -- ~300 v_mov copies of 12-register       \
       groups and 72-144 spilled registers (scratch round trips) in front of the first DMA.  Items past the tile are never \
       issued (HB_DMA_W2 counts), so their offsets may hold anything: computing all twelve costs ~100 VALU instructions. */ \
#define HB_W_OFFSETS                                                                                            \
    _Pragma("unroll") for (int it = 0; it < HW_MAX; ++it) {                                                     \
        const int piece = it * 256 + tid;                                                                       \
        const int m = piece & (BM - 1), r = piece / BM;                                                         \
        const int cb = r & (NCB - 1), tap = r >> LOGNCB;                                                        \
        const int k = tap < ntaps ? tp.kbase + tap * tp.kstep : p.K;        /* dead tap of an odd pair: the zero plane */ \
        w_off[it] = tap < ntapsW ? ((k * CB + cb) * h.Mpad + m0 + m) * 16 : 0;      /* (unused pieces: whole waves, skipped) */ \
    }
    if (!cw0) { HB_W_OFFSETS }   // cw0: chunk 0's weights are the consumers' business; the offsets follow the first X DMAs
#define HB_DMA_W(c_, buf_)                                                                                      \
    {                                                                                                           \
        const char* wc_ = wt_g + (size_t)(c_) * chunk_w;                                                        \
        char* Wd_ = smem + (buf_) * wbuf;                                                                       \
        _Pragma("unroll") for (int it = 0; it < HW_MAX; ++it) {                                                 \
            if (it < h.wit && (it * 256 + wave * 64) < ntapsW * BM * NCB && !HB_DBG(2)) {                     \
                const char* src_ = wc_ + w_off[it];                                                             \
                __builtin_amdgcn_global_load_lds((gptr_t)src_, (lds_ptr_t)(Wd_ + (it * 256 + wave * 64) * 16), 16, 0, 0); \
            }                                                                                                   \
        }                                                                                                       \
    }
    if constexpr (XH) {
        // ---- the source is a bf16 image [B][xh_cb][Tsrc][8]: piece (cb, slot) of the LDS image [cb][slot] IS 16 contiguous
        // bytes of it, so the whole X tile is staged by LDS-DMA through a buffer descriptor -- piece = it * 256 + tid lands
        // at LDS byte 16 * piece (wave-uniform base + 16 * lane); per-lane offset = (batch row, channel block of the
        // chunk, time); the chunk's first channel block is a SCALAR offset; halo / padding / slots past the tile and
        // channel blocks past the group carry an out-of-range offset: the buffer unit writes zeros for them
        // (scripts/micro/bufload_lds_oob.hip); lanes past the image (piece >= NCB * XROW) are switched off.  No
        // alignment rule: any tap offset, stride or dilation is a slot address, as before; the nearest-x2 upsample
        // reads source position q >> 1.
        const auto rsrc_h = __builtin_amdgcn_make_buffer_rsrc((void*)h.xh, 0, (int)h.xh_bytes, 0x00020000);
        const int npieces = NCB * p.XROW;
        const int cbg0 = g * (p.C2_g >> 3);                       // first channel block of the group inside the image
        const int nch = c_end - c_begin;
        // The blocks of an XCD that share an operand tile (the N tiles of one M tile share its W chunks, the M tiles of one N
        // tile its X chunks) start together and walk the chunks in step: every line is requested by all of them at the same
        // moment, the first request misses the L2 and the others queue behind the same fill -- everybody sees the miss
        // latency.  Each block therefore starts its walk at a different chunk (rot) and wraps around: at any moment the
        // sharers fetch DIFFERENT chunks, and all but the first touch of a line are L2 hits.  The sum over the chunks is the
        // same set of terms in a rotated order (fp32 rounding differs in the last bit; parity bounds unchanged).
        const int rot = (nch > 1 && h.rot) ? (int)((unsigned)(bx + by) % (unsigned)nch) : 0;
#define HB_CID(j_) (c_begin + (j_) + rot - (((j_) + rot) >= nch ? nch : 0))
        // the first chunk's weights go out BEFORE the X offsets are computed: their round trip runs under that arithmetic
        QSTAMP(3)                                                    // W offsets, X descriptor done
        // (chunk 0's weights are requested by the CONSUMER waves, which otherwise idle until the first barrier: the W path
        // and the X path to the first chunk then run side by side -- see the consumers' side below)
        if (nch > 0 && !cw0) HB_DMA_W(HB_CID(0), 0)      // (old form: before the descriptor / counts exist)
        PSTAMP(7)                                                    // tile decode + W offsets done, first W DMA issued
        unsigned xd_voff[HXD_MAX];
        int xd_cb[HXD_MAX];
        {
            // The loop-invariant scalars of the item arithmetic, moved into VECTOR registers once (vg: an opaque v_mov).  Left
            // as kernel-argument fields the compiler -- out of scalar registers with the whole HP struct live -- re-loaded
            // them from the kernarg segment inside EVERY item (two dependent s_load / s_waitcnt round trips per item) and
            // spilled scalars to vector lanes around them: the producers' setup took 7 500 cycles to the first X DMA.  Vector
            // registers are plentiful here (52 of 512 in use), and an asm result cannot be rematerialised.
            const unsigned xr_mul = vg(p.fdXROW.mul), xr_sh = vg(p.fdXROW.shift), xrow = vg((unsigned)p.XROW);
            const unsigned sg_mul = vg(p.fdSEG.mul), sg_sh = vg(p.fdSEG.shift), seg = vg((unsigned)p.SEG);
            const int blo = vg(b_lo), q0 = vg(qbase0), qN = vg(qbaseN), nseg = vg(p.NSEG), nb = vg(p.B), tq = vg(p.Tq);
            const int rowcb = vg(h.xh_cb), cb0 = vg(cbg0), tsrc = vg(p.Tsrc), npc = vg(npieces);
            const int upsh = vg(p.mode == 0 ? p.ups : 0);
#pragma unroll
            for (int it = 0; it < HXD_MAX; ++it) {              // branch-free, as HB_W_OFFSETS: items past h.xnit are never issued
                const unsigned piece = (unsigned)(it * 256 + tid);
                const unsigned cb = (__umulhi(piece, xr_mul) + piece) >> xr_sh, slot = piece - cb * xrow;
                const unsigned sg = (__umulhi(slot, sg_mul) + slot) >> sg_sh, sl = slot - sg * seg;
                const int b = blo + (int)sg;
                const int q = (sg == 0 ? q0 : qN) + (int)sl;
                const bool in = (int)piece < npc;
                const bool ok = in & ((int)sg < nseg) & (b < nb) & (q >= 0) & (q < tq);
                xd_cb[it] = in ? (int)cb : -1;                  // -1: lane switched off
                xd_voff[it] = ok ? (unsigned)(((b * rowcb + cb0 + (int)cb) * tsrc + (q >> upsh)) * 16) : OOB;
            }
        }
        const int chunk_x = NCB * p.Tsrc * 16;                    // bytes between the image planes of neighbouring chunks
        // ---- lean issue.  The first version guarded every item with its own conditions; hipcc turned them into a mask
        // register and TWO taken branches per item (out of line and back), a 64-bit address add and an exec save / restore
        // around each X item: 130 cycles per DMA instruction, 1 300 per chunk -- the issue, not the memory system, set the
        // iteration time (scripts/micro/stage_rate.hip: the same ring staged with straight-line issue takes 625 / 890 cycles
        // per 16 / 32 KiB chunk on L2 hits, 1 150 / 1 540 on misses; the conv's iterations took 1 470 / 1 940).  Now: the number
        // of items of THIS wave is one scalar (nw, nx), nested ifs run through that many straight-line items, the weights go
        // through a buffer descriptor too (offset register + scalar chunk offset: no address arithmetic per item), and the
        // lanes past the X tile are not masked: their offset is out of range, so they write zeros into the stage's padding.
        const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)wt_g, 0, (p.K + 1) * h.Mpad * h.Cpad * 2, 0x00020000);
        const int wpieces = ntapsW * BM * NCB;
        const int nw = wpieces > wave * 64 ? min(h.wit, (wpieces - wave * 64 + 255) >> 8) : 0;
        const int nx = npieces > wave * 64 ? min(h.xnit, (npieces - wave * 64 + 255) >> 8) : 0;
#define HB_W_IT(i_) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(Wd_ + ((i_) * 256 + wave * 64) * 16), 16, w_off[i_], so_, 0, 0);
#define HB_DMA_W2(c_, buf_)                                                                                     \
        if (!HB_DBG(2)) {                                                                                       \
            char* Wd_ = smem + (buf_) * wbuf;                                                                   \
            const int so_ = (c_) * chunk_w;                                                                     \
            /* nested ifs on ONE scalar: compare + not-taken branch + M0 + load per item, one taken branch at the end */ \
            if (nw > 0) { HB_W_IT(0) if (nw > 1) { HB_W_IT(1) if (nw > 2) { HB_W_IT(2) if (nw > 3) { HB_W_IT(3)         \
            if (nw > 4) { HB_W_IT(4) if (nw > 5) { HB_W_IT(5) if (nw > 6) { HB_W_IT(6) if (nw > 7) { HB_W_IT(7)         \
            if (nw > 8) { HB_W_IT(8) if (nw > 9) { HB_W_IT(9) if (nw > 10) { HB_W_IT(10) if (nw > 11) { HB_W_IT(11)     \
            } } } } } } } } } } } }                                                                             \
        }
        static_assert(HW_MAX == 12 && HXD_MAX == 8, "the item switches of the lean DMA issue are written out for 12 / 8 items");
#define HB_X_IT(i_, vo_) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_h, (lds_ptr_t)(Xd_ + ((i_) * 256 + wave * 64) * 16), 16, vo_, sx_, 0, 0);
#define HB_X_SW(VO_)                                                                                            \
            if (nx > 0) { HB_X_IT(0, VO_(0)) if (nx > 1) { HB_X_IT(1, VO_(1)) if (nx > 2) { HB_X_IT(2, VO_(2))          \
            if (nx > 3) { HB_X_IT(3, VO_(3)) if (nx > 4) { HB_X_IT(4, VO_(4)) if (nx > 5) { HB_X_IT(5, VO_(5))          \
            if (nx > 6) { HB_X_IT(6, VO_(6)) if (nx > 7) { HB_X_IT(7, VO_(7)) } } } } } } } }
#define HB_VO_FULL(i_) (int)xd_voff[i_]
#define HB_VO_RAG(i_) (xd_cb[i_] < live_ ? (int)xd_voff[i_] : (int)OOB)
#define HB_DMA_X(c_, buf_)                                                                                      \
        if (!HB_DBG(1)) {                                                                                       \
            const int live_ = (p.C2_g - (c_) * CCH + 7) >> 3;   /* channel blocks of the group left for this chunk */ \
            char* Xd_ = Xbufs + (buf_) * xbuf;                                                                  \
            const int sx_ = (c_) * chunk_x;                                                                     \
            if (live_ >= NCB) { HB_X_SW(HB_VO_FULL) }           /* every channel block of the chunk exists */   \
            else { HB_X_SW(HB_VO_RAG) }                         /* ragged last chunk: the blocks past the group read zeros */ \
        }
        // Ring of NS stages, NS - 1 chunks in flight.  An iteration of a small tile is 128-768 cycles of MFMA against a
        // DMA round trip of ~1 000 (measured with the stamps: with ONE chunk in flight -- two buffers, the barrier's
        // vmcnt(0) -- every iteration of a 64 x 64 x 64-channel chunk took 1 030 cycles for 128 cycles of MFMA).  So the
        // producers wait with a COUNTED vmcnt -- until chunk c + 1 has landed, leaving the younger chunks in flight --
        // and meet the consumers at a raw s_barrier (__syncthreads() would add vmcnt(0) while a DMA is pending).  A wave
        // issues the same number of DMA instructions for every chunk (ipc), so the count is (chunks left in flight) x ipc.
        const int ipc = nw + nx;
        PSTAMP(0)                                                    // setup done
        if (cw0) {                                                   // (two stages: the prologue is chunk 0's X tile)
            if (nch > 0) HB_DMA_X(HB_CID(0), 0)
            HB_W_OFFSETS                                             // ... and under its round trip, the W offsets of chunks 1 ..
        } else
        for (int j = 0; j < NS - 1 && j < nch; ++j) {               // prologue: chunks 0 .. NS - 2 (chunk 0's weights are out)
            if (j > 0) HB_DMA_W2(HB_CID(j), j)
            HB_DMA_X(HB_CID(j), j)
        }
        PSTAMP(1)                                                    // prologue issued
        int stage_in = (NS - 1) % NS;                                // stage the next issued chunk goes to
        for (int j = 0; j < nch; ++j) {
            // chunk j must have landed before barrier j: the chunks issued after it may stay in flight
            const int younger = min(NS - 1, nch - j) - 1;            // chunks j + 1 .. issued so far
            if (NS == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the common case without the switch of wait_vmcnt)
            else wait_vmcnt(younger * ipc);
            if (j == 0) { PSTAMP(2) }                                // chunk 0 landed
            if (j == 5) { PSTAMP(5) }                                // chunk 5 landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                            // barrier j: chunk j is visible; stage of chunk j - 1 is free
            asm volatile("" ::: "memory");
            if (j == 4) { PSTAMP(3) }                                // past barrier 4
            if (j + NS - 1 < nch) {
                HB_DMA_W2(HB_CID(j + NS - 1), stage_in)
                HB_DMA_X(HB_CID(j + NS - 1), stage_in)
                stage_in = stage_in + 1 == NS ? 0 : stage_in + 1;
            }
            if (j == 4) { PSTAMP(4) }                                // iteration 4's DMAs issued
        }
        // (no return: the producers take half of the epilogue, below)
#undef HB_DMA_X
#undef HB_DMA_W2
#undef HB_W_IT
#undef HB_X_IT
#undef HB_X_SW
#undef HB_VO_FULL
#undef HB_VO_RAG
#undef HB_CID
    } else {
    constexpr int XI = XV ? HXV_MAX : HX_MAX;
    unsigned x_voff[XI];
    int x_lds[XI], x_c8[XI];             // LDS byte offset of the item's first slot (-1: no item); 8 * cg
    int x_nv[XV ? XI : 1];               // XV: source positions of the item inside the row (< 4: the row's ragged tail)
#pragma unroll
    for (int it = 0; it < XI; ++it) {
        const int idx = it * 256 + tid;
        const int cg = (int)fd_div((uint32_t)idx, h.fdNG), grp = idx - cg * h.NG;
        const int slot = XV ? (grp << xsh) : grp;
        x_voff[it] = OOB;
        x_c8[it] = cg * 8;
        x_lds[it] = (it < h.xit && cg < NCB) ? (cg * p.XROW + slot) * 16 : -1;
        if (cg < NCB) {
            const int sg = (int)fd_div((uint32_t)slot, p.fdSEG), sl = slot - sg * p.SEG;
            const int b = b_lo + sg;
            const int q = (sg == 0 ? qal0 : qalN) + sl;
            const int qs = p.mode == 0 ? (q >> p.ups) : q;
            if (sg < p.NSEG && b < p.B && q >= 0 && q < p.Tq)
                x_voff[it] = (unsigned)(((b * p.Csrc + g * p.C2_g + cg * 8) * p.Tsrc + qs) * 4);
            if constexpr (XV) x_nv[it] = p.Tsrc - qs;
        }
    }
    // staged X values of the chunk in flight: scalar form xAr[it][j] = channel j of the item's slot; vector form
    // xAv[it][j] = 4 slots of channel j
    float xAr[XV ? 1 : HX_MAX][8];
    f32x4 xAv[XV ? HXV_MAX : 1][8];
#define HB_LOAD_X(c_, xr, xv)                                                                                   \
    {                                                                                                           \
        const int c0_ = (c_) * CCH;                                                                             \
        const int lim_ = p.C2_g - c0_;                  /* live channels of this chunk (ragged last chunk)     */ \
        const bool full_ = lim_ >= CCH;                 /* scalar: the per-load select folds away              */ \
        _Pragma("unroll") for (int it = 0; it < XI; ++it) {                                                     \
            if (it < h.xit && !HB_DBG(1)) {                                                                   \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                 \
                    /* channels past the group: out-of-range offset, the buffer unit returns 0 */               \
                    const int vo_ = (full_ || x_c8[it] + j < lim_) ? (int)x_voff[it] : (int)OOB;                \
                    const int so_ = (c0_ + j) * row_bytes;                                                      \
                    if constexpr (XV) xv[it][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo_, so_, 0)); \
                    else xr[it][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, vo_, so_, 0)); \
                }                                                                                               \
            }                                                                                                   \
        }                                                                                                       \
    }
#define HB_STORE_X(buf_, xr, xv)                                                                                \
    if (relu) { HB_STORE_X_(buf_, xr, xv, true) } else { HB_STORE_X_(buf_, xr, xv, false) }
#define HB_STORE_X_(buf_, xr, xv, relu)                                                                         \
    {                                                                                                           \
        char* Xd_ = Xbufs + (buf_) * xbuf;                                                                      \
        _Pragma("unroll") for (int it = 0; it < XI; ++it) {                                                     \
            if (it < h.xit && x_lds[it] >= 0 && !HB_DBG(8)) {                                                 \
                if constexpr (XV) {                                                                             \
                    _Pragma("unroll") for (int s4 = 0; s4 < 4; ++s4) {                                          \
                        u32x4 v_ = {pack_bf16x2(xv[it][0][s4], xv[it][1][s4], relu), pack_bf16x2(xv[it][2][s4], xv[it][3][s4], relu), \
                                    pack_bf16x2(xv[it][4][s4], xv[it][5][s4], relu), pack_bf16x2(xv[it][6][s4], xv[it][7][s4], relu)}; \
                        /* the row ends inside the load (ragged rows): the rest belongs to the next row */      \
                        if (s4 > 0 && s4 >= x_nv[it]) v_ = u32x4{0u, 0u, 0u, 0u};                               \
                        if (xsh == 3) {               /* nearest x2: source position s4 -> slots 2 s4, 2 s4 + 1 */ \
                            *(u32x4*)(Xd_ + x_lds[it] + s4 * 32) = v_;                                          \
                            *(u32x4*)(Xd_ + x_lds[it] + s4 * 32 + 16) = v_;                                     \
                        } else {                                                                                \
                            *(u32x4*)(Xd_ + x_lds[it] + s4 * 16) = v_;                                          \
                        }                                                                                       \
                    }                                                                                           \
                } else {                                                                                        \
                    const u32x4 v_ = {pack_bf16x2(xr[it][0], xr[it][1], relu), pack_bf16x2(xr[it][2], xr[it][3], relu), \
                                      pack_bf16x2(xr[it][4], xr[it][5], relu), pack_bf16x2(xr[it][6], xr[it][7], relu)}; \
                    *(u32x4*)(Xd_ + x_lds[it]) = v_;                                                            \
                }                                                                                               \
            }                                                                                                   \
        }                                                                                                       \
    }
    // Pipeline, one chunk ahead, two LDS buffers per operand, ONE barrier per chunk:
    //   iteration c: [barrier] issue W(c+1) LDS-DMA + X(c+1) loads -> MFMAs of chunk c -> convert + store X(c+1)
    // Two blocks share a CU where the grid allows, so one block's global round trip hides under the other's MFMAs.  (A
    // two-chunks-ahead variant -- ring of 3 W buffers, two X register sets -- was built and measured: hipcc drains
    // vmcnt(0) at every use of an ordinary load while an LDS-DMA is in flight, and with both operands staged through
    // registers the per-item guards around the loads make its wait counting fall back to vmcnt(0) as well; the
    // iteration time did not move and the doubled registers cost the second block per CU.)
#ifndef STEGAN_STAMPS_ITER
        HSTAMP(1)
#endif
        if (c_begin < c_end) {
            HB_LOAD_X(c_begin, xAr, xAv)
            HB_DMA_W(c_begin, 0)
            HB_STORE_X(0, xAr, xAv)
        }
        for (int c = c_begin; c < c_end; ++c) {
            const int cur = (c - c_begin) & 1;
            block_sync();                // all 8 waves: chunk c is staged; nobody reads buffer cur ^ 1 any more
            if (c + 1 < c_end) {
                HB_LOAD_X(c + 1, xAr, xAv)
                HB_DMA_W(c + 1, cur ^ 1)
                HB_STORE_X(cur ^ 1, xAr, xAv)
            }
        }
        // (no return: the producers take half of the epilogue, below)
    }   // !XH
    }
#undef HB_LOAD_X
#undef HB_STORE_X
#undef HB_STORE_X_
#undef HB_DMA_W
#undef HB_W_OFFSETS

    if (!producer) {
    // ================= consumers (waves 0-3): LDS operand reads + MFMAs =================
    // ---- chunk 0's weights (cw0): the same pieces the producers would request (piece = it * 256 + tid of the role), issued
    // here FIRST, before the consumers' own index setup: their round trip runs under it, and the producers go straight to
    // the X offsets.  The first barrier is preceded by this role's vmcnt(0).
    if constexpr (XH) {
        if (cw0 && c_begin < c_end) {
            typedef __attribute__((address_space(3))) void* lds_ptr_t;
            typedef const __attribute__((address_space(1))) void* gptr_t;
            const int nch0 = c_end - c_begin;
            const int rot0 = (nch0 > 1 && h.rot) ? (int)((unsigned)(bx + by) % (unsigned)nch0) : 0;
            const char* wc = wt_g + (size_t)(c_begin + rot0) * chunk_w;
            const int CBw = h.Cpad >> 3;
#pragma unroll
            for (int it = 0; it < HW_MAX; ++it) {
                if (it < h.wit && (it * 256 + wave * 64) < ntapsW * BM * NCB && !HB_DBG(2)) {
                    const int piece = it * 256 + tid;
                    const int m = piece & (BM - 1), r = piece / BM;
                    const int cb = r & (NCB - 1), tap = r >> LOGNCB;
                    const int k = tap < ntaps ? tp.kbase + tap * tp.kstep : p.K;
                    const int wo = tap < ntapsW ? ((k * CBw + cb) * h.Mpad + m0 + m) * 16 : 0;
                    __builtin_amdgcn_global_load_lds((gptr_t)(wc + wo), (lds_ptr_t)(smem + (it * 256 + wave * 64) * 16), 16, 0, 0);
                }
            }
        }
    }
    // ---- per-lane operand addresses.  Reduction "atoms" (tap, cb) are BM (W) / XROW or step (X) 16-byte pieces apart; the
    // two lane halves of an MFMA take atoms 2 s and 2 s + 1 of k-step s: W offset s * 2 BM * 16 for every chunk shape
    int a_byte[TM], b_byte[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_byte[i] = ((wm * TM + i) * 32 + l31 + hl * BM) * 16;
    const int tapB = tp.step * 16;                                 // bytes between the X pieces of neighbouring taps
    const int cbB = p.XROW * 16;                                   // ... of neighbouring channel blocks
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int jn = (wn * TN + j) * 32 + l31;
        int cb = 0;
        if (n0 + jn < N) {
            const int uu = u0 + jn;
            const int sg = (int)fd_div((uint32_t)uu, fdU), u = uu - sg * U;
            cb = sg * p.SEG + (u - (sg == 0 ? u0 : 0)) * cs + (sg == 0 ? qbase0 - qal0 : qbaseN - qalN);
        }
        b_byte[j] = cb * 16 + hl * (NCB == 1 ? tapB : cbB);
    }

#define HB_LDS(ptr_) (*(const bf16x8*)(ptr_))
#define HB_LOAD(a_, b_, s_)                                                                                     \
    {                                                                                                           \
        const char* wa_ = Wcur + (s_) * (BM * 32);                                                              \
        const char* xb_ = Xcur + (NCB == 1 ? (s_) * 2 * tapB : ((s_) / KS) * tapB + ((s_) % KS) * 2 * cbB);     \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) a_[i] = HB_LDS(wa_ + a_byte[i]);                         \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) b_[j] = HB_LDS(xb_ + b_byte[j]);                         \
    }
    // image path: the input ReLU cannot ride on the staging (LDS-DMA moves bytes), so it is applied to the X fragment at
    // its point of use -- max with 0 on the packed halves (sign bit set <=> negative int16), or with -32768 (identity)
    // when the layer has no input activation: one wave-uniform operand, no branch
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    const short relu_floor = (XH && relu) ? (short)0 : (short)-32768;
    const s16x8_t relu_fl = {relu_floor, relu_floor, relu_floor, relu_floor, relu_floor, relu_floor, relu_floor, relu_floor};
#define HB_XFRAG(b_) (XH ? __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8_t, b_), relu_fl)) : (b_))
#define HB_MFMA(a_, b_)                                                                                         \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                            \
        const bf16x8 bj_ = HB_XFRAG(b_[j]);                                                                     \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                          \
            acc[i][j] = STEGAN_MFMA_H(bj_, a_[i], acc[i][j]);                                                 \
    }
    // one chunk: S k-steps, operands of step s + 1 read from LDS while the MFMAs of step s run
#define HB_SCHED if constexpr (TM * TN == 1 && SCHED_ON) __builtin_amdgcn_sched_barrier(0);
#define HB_COMPUTE(c_)                                                                                          \
    if (!HB_DBG(4)) {                                                                                           \
        const char* Wcur = smem + stage_c * wbuf;                                                               \
        const char* Xcur = Xbufs + stage_c * xbuf;                                                              \
        bf16x8 aA[TM], bA[TN], aB[TM], bB[TN];                                                                  \
        if (S > 0) HB_LOAD(aA, bA, 0)                                                                           \
        int s = 0;                                                                                              \
        for (; s + 1 < S; s += 2) {                                                                             \
            HB_LOAD(aB, bB, s + 1)                                                                              \
            HB_SCHED      /* single-tile waves: the reads of step s + 1 go out BEFORE the MFMA of step s (hipcc sinks them behind it) */ \
            HB_MFMA(aA, bA)                                                                                     \
            HB_SCHED                                                                                            \
            if (s + 2 < S) HB_LOAD(aA, bA, s + 2)                                                               \
            HB_SCHED                                                                                            \
            HB_MFMA(aB, bB)                                                                                     \
            HB_SCHED                                                                                            \
        }                                                                                                       \
        if (s < S) { HB_MFMA(aA, bA) }                                                                          \
    }
#ifndef STEGAN_STAMPS_ITER
    HSTAMP(2)
#endif
    int stage_c = 0;                                              // ring stage of the chunk being multiplied
    for (int c = c_begin; c < c_end; ++c) {
        if constexpr (XH) {                                       // (the producers' side of this barrier: counted vmcnt, above)
            if (cw0 && c == c_begin) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this role's DMAs of chunk 0
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else {
            block_sync();
        }
#ifdef STEGAN_STAMPS_ITER                   // diagnostic: tops of iterations 1, 5, 9, 13, 17, 21 in stamps 1..6
        if (c == c_begin + 1) { HSTAMP(1) }
        if (c == c_begin + 5) { HSTAMP(2) }
        if (c == c_begin + 9) { HSTAMP(3) }
        if (c == c_begin + 13) { HSTAMP(4) }
        if (c == c_begin + 17) { HSTAMP(5) }
        if (c == c_begin + 21) { HSTAMP(6) }
#else
        if (c == c_begin + 1) { HSTAMP(3) }
        if (c == c_begin + 5) { HSTAMP(4) }
#endif
        HB_COMPUTE(c)
        stage_c = stage_c + 1 == NS ? 0 : stage_c + 1;
    }
#undef HB_COMPUTE
#undef HB_SCHED
#undef HB_MFMA
#undef HB_XFRAG
#undef HB_LOAD
#undef HB_LDS
    }   // consumers

    // ---- epilogue.  The MFMAs ran with the X fragment as the A operand, so a 32 x 32 accumulator tile has its output
    // CHANNEL on the lane and 16 output COLUMNS (time) in its registers, as four runs of 4 consecutive columns
    // (n = 8 q + 4 half + 0..3): scale / bias are ONE value per lane and tile, and wherever 4 consecutive columns are 4
    // consecutive, 16-byte aligned floats of one output row (unit output stride, column count a multiple of 4) a run is ONE
    // 16-byte load per fused input (residual, mask) and ONE 16-byte store.  (The first version kept channels in the
    // registers: 64 dword stores and 128 dependent scalar-per-lane loads per lane cost as much as the whole main loop.)
#ifndef STEGAN_STAMPS_ITER
    HSTAMP(5)
#endif
    // ---- ALL EIGHT WAVES run the epilogue (round 4).  It is bound by instruction issue, not by memory: with every load and
    // store switched off it still cost 1.2 ms of a 6.1 ms train step (ablation builds) -- a few hundred
    // VALU / LDS / address instructions per 32 x 32 tile and wave, issued by ONE wave per SIMD while the four producer
    // waves had already left.  Now each consumer wave hands the SECOND half of every tile it owns (column runs q = 2, 3:
    // 8 of its 16 values per lane) to the producer wave at the same (wm, wn) position through LDS, and both process 8
    // values per lane and tile: compile-time register indices in both roles (the producer receives the values into
    // elements 0..7), only the column base differs (+16).
    block_sync();                        // every wave: the last chunk's operands have been read, all DMAs have landed
    const int half = producer ? 1 : 0;
    {
        float* const hand = (float*)smem + ((wave * TM * TN) * 64 + lane) * 8;          // [position][tile][lane][8 floats]
        if (!producer) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float* hp = hand + (i * TN + j) * 64 * 8;
                    *(f32x4*)(hp) = (f32x4){acc[i][j][8], acc[i][j][9], acc[i][j][10], acc[i][j][11]};
                    *(f32x4*)(hp + 4) = (f32x4){acc[i][j][12], acc[i][j][13], acc[i][j][14], acc[i][j][15]};
                }
        }
        block_sync();
        if (producer) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float* hp = hand + (i * TN + j) * 64 * 8;
                    const f32x4 lo = *(const f32x4*)(hp), hi = *(const f32x4*)(hp + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { acc[i][j][e] = lo[e]; acc[i][j][4 + e] = hi[e]; }
                }
        }
    }
    if (HB_DBG(16)) {
        if (acc[0][0][0] == 123.456f) p.out[0] = 1.f;        // keep the accumulators alive
        return;
    }
    const size_t out_elems = (size_t)p.B * p.Cdst * p.Tdst;
    const bool vec4 = h.vec_out != 0;                          // host: os == 1, U % 4 == 0, Tdst % 4 == 0, aligned pointers
    const bool raw = nsplits > 1;                              // split-K: raw partial sums, the epilogue kernel fuses the rest
    float* const outp = raw ? p.partial + (size_t)split * out_elems : p.out;
    // every launch-uniform condition is resolved ONCE here (a single wave per SIMD pays every branch and scalar load
    // of the epilogue in full: the first version, with the conditions inside the element loops, took 20k cycles)
    const int act = raw ? STEGAN_ACT_NONE : p.out_act;
    // (HB_DBG: diagnostic ablations of the epilogue's parts, -DSTEGAN_BF16_ABLATE builds only: 32 no image, 64 no mask /
    // residual loads, 128 no scale / bias loads, 256 no fp32 stores -- results wrong when set)
    const float* const resp = (raw || HB_DBG(64)) ? nullptr : p.res;
    const float* const maskp = (raw || p.mode != 1 || HB_DBG(64)) ? nullptr : p.mask;
    const float* const scalep = (raw || HB_DBG(128)) ? nullptr : p.scale;
    const float* const biasp = (raw || p.mode != 0 || HB_DBG(128)) ? nullptr : p.bias;
    const int mode = p.mode, ups = p.ups, Cdst = p.Cdst, Tdst = p.Tdst, Tmask = p.Tmask, M_g = p.M_g;
    const bool mlr = p.mask_lrelu != 0;                        // the mask is a leaky-ReLU output: factor AFTER scale and add
#define HB_ACT8(v_)                                                                                             \
    if (act == STEGAN_ACT_LRELU) { _Pragma("unroll") for (int e = 0; e < 8; ++e) v_[e] = v_[e] > 0.f ? v_[e] : 0.1f * v_[e]; } \
    else if (act == STEGAN_ACT_RELU) { _Pragma("unroll") for (int e = 0; e < 8; ++e) v_[e] = fmaxf(v_[e], 0.f); }       \
    else if (act == STEGAN_ACT_TANH) { _Pragma("unroll") for (int e = 0; e < 8; ++e) v_[e] = tanhf(v_[e]); }
    // ---- bf16 image of the output (DirP::out_h), for the kernel that consumes it next: [B][out_cb][Tdst][8 channels].
    // A 16-byte piece needs 8 CHANNELS of one column -- 8 lanes of this layout -- so each half tile (32 channels x 16
    // columns) goes through a 1 KiB LDS scratch of its wave: 8 two-byte stores per lane into [column][channel], ONE 16-byte
    // read per lane of a (column, 8-channel block) piece, 16 consecutive lanes on consecutive columns: 256 contiguous bytes
    // of the image per quarter of a store instruction.  The scratch sits behind the hand-over area of the W / X ring.
    unsigned short* const outh = (raw || HB_DBG(32)) ? nullptr : p.out_h;
    char* const scr = smem + 4 * TM * TN * 64 * 32 + (wave + 4 * half) * 1024;
#define HB_IMAGE_TILE(v_, i_, j_, m_ok_)                                                                        \
    if (outh) {                                                                                                 \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                         \
            const int nl_ = 8 * (e >> 2) + 4 * hl + (e & 3);                     /* local column 0..15 */       \
            *(unsigned short*)(scr + nl_ * 64 + l31 * 2) = (m_ok_) ? f32_to_bf16_bits(v_[e]) : (unsigned short)0; \
        }                                                                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
        {                                                                                                       \
            const int cbq_ = lane >> 4, nl_ = lane & 15;                                                        \
            const u32x4 pc_ = *(const u32x4*)(scr + nl_ * 64 + cbq_ * 16);                                      \
            const int mb_ = m0 + (wm * TM + (i_)) * 32 + cbq_ * 8;                                              \
            const int n_ = n0 + (wn * TN + (j_)) * 32 + 16 * half + nl_;                                        \
            if (mb_ < M_g && n_ < N) {                                                                          \
                const int b_ = (int)fd_div((uint32_t)n_, fdU), u_ = n_ - b_ * U;                                \
                const int to_ = mode == 0 ? u_ : phase + u_ * os;                                               \
                *(u32x4*)(outh + (((size_t)b_ * p.out_cb + ((g * M_g + mb_) >> 3)) * Tdst + to_) * 8) = pc_;    \
            }                                                                                                   \
        }                                                                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* the read is done before the next tile's stores */ \
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + (wm * TM + i) * 32 + l31;
        const bool m_ok = m < M_g;
        const int ch = g * M_g + (m_ok ? m : M_g - 1);
        float sc = 1.f, bi = 0.f;
        if (scalep) sc = mode == 0 ? scalep[ch] : scalep[0];
        if (biasp) bi = biasp[ch];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nb0 = n0 + (wn * TN + j) * 32 + 4 * hl + 16 * half;       // this wave's half of the tile's columns
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[i][j][e];
            if (vec4) {
                int oidx[2], midx[2];
                bool ok[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int n = nb0 + 8 * q;
                    ok[q] = m_ok && n < N;
                    const int nn = n < N ? n : 0;
                    const int b = (int)fd_div((uint32_t)nn, fdU), u = nn - b * U;
                    oidx[q] = (b * Cdst + ch) * Tdst + (mode == 0 ? u : phase + u);
                    midx[q] = (b * Cdst + ch) * Tmask + (u >> 1);
                }
                if (maskp && !mlr) {
                    if (ups == 0) {
                        f32x4 mk[2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) mk[q] = *(const f32x4*)(maskp + oidx[q]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = mk[e >> 2][e & 3] > 0.f ? v[e] : 0.f;
                    } else {
                        f32x2 mk[2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) mk[q] = *(const f32x2*)(maskp + midx[q]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = mk[e >> 2][(e & 3) >> 1] > 0.f ? v[e] : 0.f;
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * sc + bi;
                if (resp) {
                    f32x4 rs[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) rs[q] = *(const f32x4*)(resp + oidx[q]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += rs[e >> 2][e & 3];
                }
                if (maskp && mlr) {                        // leaky-ReLU derivative of the layer before (host: ups == 0)
                    f32x4 mk[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) mk[q] = *(const f32x4*)(maskp + oidx[q]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= mk[e >> 2][e & 3] > 0.f ? 1.f : 0.1f;
                }
                HB_ACT8(v)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    if (ok[q] && !HB_DBG(256)) *(f32x4*)(outp + oidx[q]) = (f32x4){v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
                HB_IMAGE_TILE(v, i, j, m_ok)
            } else {
                int oi[8];
                bool okk[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int n = nb0 + 8 * (e >> 2) + (e & 3);
                    okk[e] = m_ok && n < N;
                    const int nn = n < N ? n : 0;
                    const int b = (int)fd_div((uint32_t)nn, fdU), u = nn - b * U;
                    const int to = mode == 0 ? u : phase + u * os;
                    oi[e] = (b * Cdst + ch) * Tdst + to;
                    if (maskp && !mlr) {
                        const float xv = maskp[(b * Cdst + ch) * Tmask + (to >> ups)];
                        v[e] = xv > 0.f ? v[e] : 0.f;
                    }
                    if (!raw && p.out_btc) oi[e] = (b * Tdst + to) * Cdst + ch;       // (B, T, C) store; never with a residual
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * sc + bi;
                if (resp) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += resp[oi[e]];
                }
                if (maskp && mlr) {                        // (ups == 0, no (B, T, C) store: the mask sits at the output's index)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= maskp[oi[e]] > 0.f ? 1.f : 0.1f;
                }
                HB_ACT8(v)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (okk[e]) outp[oi[e]] = v[e];
                HB_IMAGE_TILE(v, i, j, m_ok)
            }
        }
    }
#undef HB_ACT8
#undef HB_IMAGE_TILE
#ifndef STEGAN_STAMPS_ITER
    HSTAMP(6)
#endif
#if defined(STEGAN_STAMPS) && !defined(STEGAN_STAMPS_ITER)
    if (p.stamps && threadIdx.x == 0) {
        __builtin_amdgcn_s_waitcnt(0);                         // stores drained (vmcnt 0)
        p.stamps[(size_t)bid_x * 8 + 7] = __builtin_amdgcn_s_memtime();
    }
#endif
#undef HSTAMP
#undef PSTAMP
}

// Two 512-thread blocks per CU need 4 waves per SIMD: 128 VGPRs.  Left to itself hipcc gives the 4-byte-load variants
// 135-141 and the 128 x 128 tile 129-131 registers (ONE block per CU); told the target it fits 128 with no VGPR spill.
#ifndef STEGAN_BF16_WAVES
#define STEGAN_BF16_WAVES 4
#endif
#if STEGAN_BF16_WAVES > 0
#define HB_OCC __attribute__((amdgpu_waves_per_eu(STEGAN_BF16_WAVES, STEGAN_BF16_WAVES)))
#else
#define HB_OCC
#endif

template <int TM, int TN, int LOGNCB, bool XV, bool XH = false>
__global__ __launch_bounds__(512) HB_OCC void conv_bf16_kernel(const HP h) {
    conv_bf16_body<TM, TN, LOGNCB, XV, XH>(h, blockIdx.x, gridDim.x, 0);
}

// Multi-problem launch: up to HMULTI independent convolutions that share a kernel variant (the same layer of the five
// period discriminators: one channel structure; own weights, batch and length) in ONE grid, as in conv_direct.hip.
constexpr int HMULTI = 5;
struct MultiH {
    int n;
    int first[HMULTI + 1];          // first linear block of each problem
    HP p[HMULTI];
};
static_assert(sizeof(MultiH) <= 4096, "kernel argument segment limit");

template <int TM, int TN, int LOGNCB, bool XV, bool XH = false>
__global__ __launch_bounds__(512) HB_OCC void conv_bf16_multi_kernel(const MultiH a) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < HMULTI; ++i)
        if (i < a.n && (int)blockIdx.x >= a.first[i]) pi = i;
    conv_bf16_body<TM, TN, LOGNCB, XV, XH>(a.p[pi], blockIdx.x - a.first[pi], a.first[pi + 1] - a.first[pi], 0);
}

// ---- bf16 weight copies: v (Cout, Cin_g, K) fp32 -> wf[g][k][ci/8][co][ci%8], wb[g][k][co/8][ci][co%8] (x scale[co]);
// k = K: zero plane.  Channel-block-major: the rows of one 8-channel block are contiguous 16-byte pieces, the order of the
// kernel's LDS image, so a W tile is staged by fully contiguous 1 KiB LDS-DMA wave-instructions.
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
    return __builtin_bit_cast(unsigned short, (hop_t)hop_src(f));
}

// Tiled copy: a block reads ROWS output channels x CIT input channels x K taps of the (Cout, Cin_g, K) tensor -- ROWS
// contiguous runs of CIT K floats -- into LDS once and emits the 16-byte pieces of BOTH layouts from there:
//   wf[g][k][ci / 8][co][8 ci]   : piece (k, 8-channel block, co),  consecutive threads consecutive co  -> 1 KiB runs
//   wb[g][k][co / 8][ci][8 co]   : piece (k, 8-row block, ci) x scale[co], consecutive threads consecutive ci
// (element e = (((g (K+1) + k) CB + cb) M + m) 8 + j).  Every weight is read once and written twice; the first version
// gathered element by element with a K-float stride, twice per weight: 146 us for the generator's 24 M weights.  The
// buffers are zero-filled when they are allocated and only live elements are ever written: the zero tap plane k = K and
// the padding rows / channels stay zero.
// 25 KiB of LDS per block (six blocks per CU; the first version's 48 KiB allowed three): this kernel only moves memory
#ifndef STEGAN_PREP_LDS_FLOATS
#define STEGAN_PREP_LDS_FLOATS 6400
#endif
constexpr int PREP_LDS_FLOATS = STEGAN_PREP_LDS_FLOATS;
__host__ __device__ inline int prep_cit(int K) {               // input channels per tile: CIT K <= ~192 floats per row
    int c = (192 / K) & ~7;
    return c < 8 ? 8 : (c > 64 ? 64 : c);
}
__host__ __device__ inline int prep_rows(int K) {              // output rows per tile: the most that fit
    const int P = prep_cit(K) * K + 1;
    return 64 * P <= PREP_LDS_FLOATS ? 64 : (32 * P <= PREP_LDS_FLOATS ? 32 : (16 * P <= PREP_LDS_FLOATS ? 16 : 8));
}
__host__ __device__ inline long prep_tiles(int G, int Cout_g, int Cin_g, int K) {
    return (long)G * ((Cout_g + prep_rows(K) - 1) / prep_rows(K)) * ((Cin_g + prep_cit(K) - 1) / prep_cit(K));
}

__device__ __forceinline__ void prep_bf16_tiles(const PrepH& t, long tile0, long tstride) {
    __shared__ float lds[PREP_LDS_FLOATS];
    const int K = t.K, CIT = prep_cit(K), ROWS = prep_rows(K), P = CIT * K + 1;
    if (ROWS * P > PREP_LDS_FLOATS) return;                    // K > 47: not produced by the host (fill_prep rejects it)
    const int mt_n = (t.Cout_g + ROWS - 1) / ROWS, ct_n = (t.Cin_g + CIT - 1) / CIT;
    const long ntiles = (long)t.G * mt_n * ct_n;
    const int CBf = t.Cf >> 3, CBb = t.Cb >> 3;
    for (long tile = tile0; tile < ntiles; tile += tstride) {
        const int ct = (int)(tile % ct_n);
        const int mt = (int)((tile / ct_n) % mt_n);
        const int g = (int)(tile / ((long)ct_n * mt_n));
        const int m0 = mt * ROWS, c0 = ct * CIT;
        const int rows = min(ROWS, t.Cout_g - m0), cols = min(CIT, t.Cin_g - c0) * K;      // live part of the tile
        block_sync();                                        // previous tile's readers are done
        for (int idx = threadIdx.x; idx < ROWS * CIT * K; idx += 256) {
            const int r = idx / (CIT * K), j = idx - r * (CIT * K);
            float v = 0.f;
            if (r < rows && j < cols) v = t.v[((size_t)(g * t.Cout_g + m0 + r) * t.Cin_g + c0) * K + j];
            lds[r * P + j] = v;
        }
        block_sync();
        // forward layout: pieces (k, cbl, r), r fastest
        const int nwf = K * (CIT / 8) * ROWS;
        for (int pi = threadIdx.x; pi < nwf; pi += 256) {
            const int r = pi % ROWS, q = pi / ROWS, cbl = q % (CIT / 8), k = q / (CIT / 8);
            if (r >= rows || cbl * 8 * K >= cols) continue;     // padding stays zero
            const float* src = lds + r * P + cbl * 8 * K + k;
            u32x4 o;
#pragma unroll
            for (int h2 = 0; h2 < 4; ++h2)
                o[h2] = (unsigned)to_bf16(src[(2 * h2) * K]) | ((unsigned)to_bf16(src[(2 * h2 + 1) * K]) << 16);
            const size_t e = ((((size_t)g * (K + 1) + k) * CBf + (c0 >> 3) + cbl) * t.Mf + m0 + r) * 8;
            *reinterpret_cast<u32x4*>(t.wf + e) = o;
        }
        // input-gradient layout: pieces (k, rb, cil), cil fastest; scale[co] folded in
        const int nwb = K * (ROWS / 8) * CIT;
        for (int pi = threadIdx.x; pi < nwb; pi += 256) {
            const int cil = pi % CIT, q = pi / CIT, rb = q % (ROWS / 8), k = q / (ROWS / 8);
            if (rb * 8 >= rows || cil * K >= cols) continue;
            float f[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int r = rb * 8 + jj;
                const float sc = (t.scale && r < rows) ? t.scale[g * t.Cout_g + m0 + r] : 1.f;
                f[jj] = lds[r * P + cil * K + k] * sc;          // rows past the group hold zeros
            }
            u32x4 o;
#pragma unroll
            for (int h2 = 0; h2 < 4; ++h2) o[h2] = (unsigned)to_bf16(f[2 * h2]) | ((unsigned)to_bf16(f[2 * h2 + 1]) << 16);
            const size_t e = ((((size_t)g * (K + 1) + k) * CBb + (m0 >> 3) + rb) * t.Mb + c0 + cil) * 8;
            *reinterpret_cast<u32x4*>(t.wb + e) = o;
        }
    }
}

__global__ __launch_bounds__(256) void conv_prep_bf16_kernel(const PrepH t) {
    prep_bf16_tiles(t, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(256) void conv_prep_bf16_multi_kernel(const PrepH* __restrict__ table) {
    const PrepH t = table[blockIdx.y];
    prep_bf16_tiles(t, blockIdx.x, gridDim.x);
}

void fill_prep(PrepH& t, const stegan_conv_desc* d) {
    t.G = d->groups; t.K = d->K;
    t.Cout_g = d->Cout / d->groups; t.Cin_g = d->Cin / d->groups;
    t.Mf = round_up(t.Cout_g, 128); t.Cf = round_up(t.Cin_g, 64);
    t.Mb = round_up(t.Cin_g, 128); t.Cb = round_up(t.Cout_g, 64);
    t.nf = (long)t.G * (t.K + 1) * t.Mf * t.Cf;
    t.nb = (long)t.G * (t.K + 1) * t.Mb * t.Cb;
}

// 16-byte X loads: float4-aligned rows (the forward's upsample included).  Rows that are NOT float4-aligned can take them
// too (a dwordx4 load needs dword alignment only; the load straddling a row's end has its tail zeroed) -- built, parity
// green, and measured 1 % SLOWER on the train step (the period discriminators' rows: misaligned 16-byte loads split in
// the memory pipeline), so it is opt-in: STEGAN_BF16_XVRAG=1.  Experiment switches back to the first version's 4-byte
// loads: STEGAN_BF16_NO_XV (all), STEGAN_BF16_NO_XVUP (the forward's upsample).
bool xv_eligible(int Tsrc, const void* src, bool fwd_ups) {
    if (STEGAN_ENV("STEGAN_BF16_NO_XV") || (fwd_ups && STEGAN_ENV("STEGAN_BF16_NO_XVUP"))) return false;
    return (Tsrc % 4 == 0 && ((uintptr_t)src % 16) == 0) || STEGAN_ENV("STEGAN_BF16_XVRAG");
}

struct GeoH { int TM, TN, logNCB, SEG, NSEG, XROW, NG, ntW, xit, wit, span; size_t lds; bool ok; };

// geometry for one (TM, TN): the largest channel chunk whose tiles fit (cch_max caps it: experiments)
GeoH plan_bf16(int TM, int TN, int C2_g, int ntaps_max, int span_taps, int step_max, int cs, int U, int U_min, bool xv,
               int cch_max, int xgran = 4, bool dma_x = false) {
    GeoH gm;
    gm.ok = false; gm.TM = TM; gm.TN = TN;
    const int BM = 64 * TM, BN = 64 * TN;
    GeoH fallback = gm;
    // (128-channel chunks -- logNCB 4 -- exist for the image path only: half as many iterations for the 1 x 1 layers, whose
    // 16 KiB chunks pay the landing latency and the barrier every 64 channels)
    for (int logNCB = dma_x ? 4 : 3; logNCB >= 0; --logNCB) {
        const int NCB = 1 << logNCB, CCH = 8 * NCB;
        if (CCH > cch_max && logNCB > 0) continue;
        if (logNCB > 0 && CCH > round_up(C2_g, 8) && CCH > 16) continue;       // chunk wider than the group: only padding
        if (logNCB > 0 && C2_g <= 8) continue;
        const int ntW = NCB == 1 ? round_up(ntaps_max, 2) : ntaps_max;
        const int span = NCB == 1 ? span_taps + (ntW - ntaps_max) * step_max : span_taps;   // the dead tap reads real rows
        const int UT = U < BN ? U : BN;
        int SEG = (UT - 1) * cs + span + 1;
        if (xv) SEG = (SEG + 2 * (xgran - 1)) & ~(xgran - 1);   // up to xgran - 1 slots of down-rounding, whole items
        int NSEG = 1;
        for (int uu = (U_min < 1 ? 1 : U_min); uu <= U; ++uu) {
            const int ns = (uu % BN == 0) ? 1 : ((BN + uu - 2) / uu + 1);
            if (ns > NSEG) NSEG = ns;
        }
        const int XROW = SEG * NSEG;
        const int NG = xv ? XROW / xgran : XROW;
        const int xit = ceil_div(NCB * NG, 256), wit = ceil_div(ntW * BM * NCB, 256);
        // image source: the X tile is DMA pieces, not register items -- bounded by HXD_MAX wave-instructions per wave
        if ((dma_x ? xit > HXD_MAX : xit > (xv ? HXV_MAX : HX_MAX)) || wit > HW_MAX) continue;
        const size_t lds = 2 * ((size_t)ntW * BM * NCB * 16 + (size_t)((XROW * NCB + 63) & ~63) * 16);       // W ring of 2, X ring of 2 (X stage padded to 64 pieces)
        if (lds > 160 * 1024) continue;
        GeoH c = gm;
        c.logNCB = logNCB; c.SEG = SEG; c.NSEG = NSEG; c.XROW = XROW; c.NG = NG; c.ntW = ntW;
        c.xit = xit; c.wit = wit; c.span = span; c.lds = lds; c.ok = true;
        static const char* env_pref = getenv("STEGAN_BF16_LDS_PREF");       // KiB; experiments
        if (lds <= (size_t)(env_pref ? atoi(env_pref) : 80) * 1024) return c;   // default: two blocks per CU
        if (!fallback.ok) fallback = c;
    }
    return fallback;
}

}  // namespace

// ---- kernel-timestamp timing (common.h) -----------------------------------------------------------------------------------
#include <vector>
static bool g_timing = false;
static std::vector<hipEvent_t> g_timing_events;        // start, stop, start, stop, ...
bool stegan_timing_on() { return g_timing; }
void stegan_timing_pair(hipEvent_t* start, hipEvent_t* stop) {
    (void)hipEventCreate(start);
    (void)hipEventCreate(stop);
    g_timing_events.push_back(*start);
    g_timing_events.push_back(*stop);
}
extern "C" int stegan_kernel_timing(int on) {
    g_timing = on != 0;
    return STEGAN_OK;
}
extern "C" int64_t stegan_kernel_timing_count() { return (int64_t)(g_timing_events.size() / 2); }
extern "C" int stegan_kernel_timing_collect(float* ms, int64_t cap) {
    const int64_t n = (int64_t)(g_timing_events.size() / 2);
    if (n > cap) return STEGAN_E_ARG;
    for (int64_t i = 0; i < n; ++i) {
        hipEvent_t e0 = g_timing_events[2 * i], e1 = g_timing_events[2 * i + 1];
        hipError_t e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms[i], e0, e1);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        if (e != hipSuccess) { g_timing_events.clear(); return (int)e; }
    }
    g_timing_events.clear();
    return STEGAN_OK;
}

namespace {

template <int TM, int TN, bool XV, bool XH = false>
void launch_variant(int logNCB, dim3 grid, size_t lds, hipStream_t s, const HP& h) {
    switch (logNCB) {
        case 0: STEGAN_TIMED_LAUNCH((conv_bf16_kernel<TM, TN, 0, XV, XH>), grid, dim3(512), lds, s, h); break;
        case 1: STEGAN_TIMED_LAUNCH((conv_bf16_kernel<TM, TN, 1, XV, XH>), grid, dim3(512), lds, s, h); break;
        case 2: STEGAN_TIMED_LAUNCH((conv_bf16_kernel<TM, TN, 2, XV, XH>), grid, dim3(512), lds, s, h); break;
        case 4: if constexpr (XH) { STEGAN_TIMED_LAUNCH((conv_bf16_kernel<TM, TN, 4, XV, XH>), grid, dim3(512), lds, s, h); break; }
                [[fallthrough]];
        default: STEGAN_TIMED_LAUNCH((conv_bf16_kernel<TM, TN, 3, XV, XH>), grid, dim3(512), lds, s, h); break;
    }
}

template <int TM, int TN, bool XV, bool XH = false>
void set_lds_attr() {
    (void)hipFuncSetAttribute((const void*)conv_bf16_kernel<TM, TN, 0, XV, XH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_bf16_kernel<TM, TN, 1, XV, XH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_bf16_kernel<TM, TN, 2, XV, XH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_bf16_kernel<TM, TN, 3, XV, XH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if constexpr (XH)
        (void)hipFuncSetAttribute((const void*)conv_bf16_kernel<TM, TN, 4, XV, XH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

struct MultiCollectH {              // host side of a multi-problem launch being assembled
    MultiH mh;
    int TM, TN, logNCB;             // variant of the first problem, forced on the others (0 / -1: not chosen yet)
    bool xv;                        // 16-byte X loads only when every problem's rows are float4 aligned
    bool xh = false;                // every problem's source comes as a bf16 image (LDS-DMA staging)
    size_t lds;
    long blocks;
};

template <int TM, int TN, bool XV, bool XH = false>
void launch_variant_multi(int logNCB, dim3 grid, size_t lds, hipStream_t s, const MultiH& a) {
    switch (logNCB) {
        case 0: hipLaunchKernelGGL((conv_bf16_multi_kernel<TM, TN, 0, XV, XH>), grid, dim3(512), lds, s, a); break;
        case 1: hipLaunchKernelGGL((conv_bf16_multi_kernel<TM, TN, 1, XV, XH>), grid, dim3(512), lds, s, a); break;
        case 2: hipLaunchKernelGGL((conv_bf16_multi_kernel<TM, TN, 2, XV, XH>), grid, dim3(512), lds, s, a); break;
        case 4: if constexpr (XH) { hipLaunchKernelGGL((conv_bf16_multi_kernel<TM, TN, 4, XV, XH>), grid, dim3(512), lds, s, a); break; }
                [[fallthrough]];
        default: hipLaunchKernelGGL((conv_bf16_multi_kernel<TM, TN, 3, XV, XH>), grid, dim3(512), lds, s, a); break;
    }
}

// returns 1 when this kernel does not take the shape (caller uses the fp32 kernels)
int launch_bf16(const stegan_conv_desc* d, int mode, HP& h, float* workspace, int64_t workspace_floats, hipStream_t s,
                MultiCollectH* mc = nullptr) {
    DirP& p = h.d;
    int k_lo, k_hi;
    const int K_eff = live_taps(d, k_lo, k_hi);
    if (K_eff <= 0) return STEGAN_E_ARG;
    p.K = d->K; p.stride = d->stride; p.dil = d->dilation; p.pad = d->pad; p.k_lo = k_lo; p.k_hi = k_hi;
    p.B = d->B; p.mode = mode;
    p.fdS = make_fastdiv((uint32_t)d->stride);
    // shapes the matrix cores cannot fill stay on the exact fp32 kernels: fewer than 8 reduction channels (the
    // input gradient of the Cout = 1 output convs) or fewer than 8 output rows per group (Cout = 1).  8 rows of a 64-row
    // tile still win: the input gradient of the 8 -> 128 k15 first layer takes 10.6 us here against 40.5 us in fp32.
    static const char* env_minm = getenv("STEGAN_BF16_MIN_M");
    const int min_m = env_minm ? atoi(env_minm) : 8;
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
    // the source as a bf16 image (staged by LDS-DMA): whole 8-channel blocks per group, an image within the buffer
    // descriptor's 2 GiB; a multi launch takes it when EVERY problem has one
    static const bool no_xh = getenv("STEGAN_BF16_NO_XH") != nullptr;
    const int64_t img_bytes = (int64_t)d->B * ceil_div(p.Csrc, 8) * p.Tsrc * 16;
    const bool xh = (mc ? mc->xh : (h.xh != nullptr)) && h.xh && !no_xh && p.C2_g % 8 == 0 && img_bytes < (int64_t)OOB;
    if (mc && mc->xh && !xh) return 1;
    if (!xh) h.xh = nullptr;
    h.xh_cb = ceil_div(p.Csrc, 8); h.xh_bytes = (unsigned)img_bytes;
    const bool xv = xh ? false : (mc ? mc->xv : xv_eligible(p.Tsrc, p.src, mode == 0 && p.ups));
    // round 5: the stride-1, ungrouped layers whose source comes as an image go to the one-tile-per-CU kernel (conv_tile.hip)
    // when its plan takes the shape: generator blocks, encoder convs and linears, forward and input gradient
    if (xh && !mc && d->stride == 1 && d->groups == 1 && !p.out_btc) {
        const Taps t0 = phase_taps_host(p, 0);
        TileConvArgs ta;
        ta.xh = h.xh; ta.wh = h.wh; ta.scale = p.scale; ta.bias = mode == 0 ? p.bias : nullptr; ta.res = p.res;
        ta.mask = mode == 1 ? p.mask : nullptr; ta.out = p.out; ta.out_h = p.out_h;
        ta.B = d->B; ta.Csrc = p.Csrc; ta.Cdst = p.Cdst; ta.Tsrc = p.Tsrc; ta.Tq = p.Tq; ta.Tdst = p.Tdst; ta.Tmask = p.Tmask;
        ta.ups = p.ups; ta.mode = mode; ta.C2 = p.C2_g; ta.M = p.M_g; ta.Mpad = h.Mpad; ta.Cpad = h.Cpad; ta.K = d->K;
        ta.ntaps = t0.ntaps; ta.kbase = t0.kbase; ta.kstep = t0.kstep; ta.off_min = t0.off_min; ta.step = t0.step;
        ta.in_relu = p.in_relu; ta.out_act = p.out_act; ta.mask_lrelu = p.mask_lrelu;
        const int rc = stegan_tile_conv(ta, s);
        if (rc != 1) return rc == STEGAN_OK ? STEGAN_OK_TILE : rc;
    }
    const int xgran = (mode == 0 && p.ups) ? 8 : 4;               // slots per 16-byte X item
    // Tile and split-K factor from a small cost model in microseconds, constants measured with the in-kernel stamps and the
    // graph-timed microbenchmark (scripts/bf16_stamps.py, scripts/bf16_microbench.py) on the hot-path layers:
    //   block time   = 5 (index setup + epilogue) + chunks x (0.35 + staged bytes per chunk / 68 KB per us): an iteration is
    //                  bound by the producer waves, whose time grows with the bytes they stage;
    //   co-residency : two blocks share a CU when their LDS allows it and the tile is not 128 x 128 (registers); each then
    //                  runs ~20 % slower;
    //   launch time  = rounds x block time (+ 6 + (splits + 1) x output bytes / 3 TB/s for the split-K partials and their
    //                  epilogue kernel).
    const char* env_tile = STEGAN_ENV("STEGAN_BF16_TILE");                 // "MN", e.g. "22", "11" (experiments)
    const char* env_cch = STEGAN_ENV("STEGAN_BF16_CCH_MAX");
    const char* env_spl = STEGAN_ENV("STEGAN_BF16_SPLITS");
    // (128-channel chunks, image path: the 1 x 1 layers alone get 10-17 % faster -- 768 -> 768: 10.9 -> 9.1 us -- but their
    // 68 KiB of LDS per block crowd the kernels of the other streams: 6.40 against 6.31 ms per step.  STEGAN_BF16_CCH_MAX=128)
    int cch_max = env_cch ? atoi(env_cch) : 64;
    if (mc && mc->mh.n > 0) cch_max = 8 << mc->logNCB;            // later problems of a multi launch: the first one's variant
    const int64_t out_elems = (int64_t)d->B * p.Cdst * p.Tdst;
    const int max_splits = (workspace && !mc) ? (int)(workspace_floats / out_elems < 16 ? workspace_floats / out_elems : 16) : 1;
    int TM = 0, TN = 0, splits = 1;
    GeoH gm;
    gm.ok = false;
    double best = 1e30;
    for (int cand = 0; cand < 4; ++cand) {
        const int tm = cand < 2 ? 2 : 1, tn = (cand & 1) ? 1 : 2;      // (2,2) (2,1) (1,2) (1,1)
        if (env_tile && env_tile[0] && env_tile[1] && (tm != env_tile[0] - '0' || tn != env_tile[1] - '0')) continue;
        if (mc && mc->mh.n > 0 && (tm != mc->TM || tn != mc->TN)) continue;
        if (mc && tm == 2 && tn == 2) continue;                      // (keeps the (2,2) variants out of the multi kernels)
        if (tm == 2 && p.M_g <= 64) continue;
        const GeoH g2 = plan_bf16(tm, tn, p.C2_g, ntaps_max, span_off, step_max, cs, U, U_min, xv, cch_max, xgran, xh);
        if (!g2.ok) continue;
        if (mc && mc->mh.n > 0 && g2.logNCB != mc->logNCB) continue;
        const int cch = 8 << g2.logNCB, nch = ceil_div(p.C2_g, cch);
        const long blocks1 = ceil_div(p.M_g, 64 * tm) * (long)d->groups * phases * ((N + 64 * tn - 1) / (64 * tn));
        // bytes the producers move per chunk; an image source is half the bytes and needs no conversion / LDS stores:
        // DMA pieces cost the producers about half of what register-staged bytes do (STEGAN_BF16_XH_RATE: experiments)
        static const char* env_xr = getenv("STEGAN_BF16_XH_RATE");
        const double xcost = xh ? (env_xr ? atof(env_xr) : 0.5) * 2.0 : 4.0;
        const double bytes = (double)g2.ntW * 64 * tm * cch * 2 + (double)g2.XROW * cch * xcost;
        static const char* env_p22 = getenv("STEGAN_BF16_PAIR22");
        const bool pair = g2.lds <= 80 * 1024 && (!(tm == 2 && tn == 2) || (env_p22 && atoi(env_p22)));
        for (int sp = 1; sp <= (max_splits < nch ? max_splits : nch); ++sp) {
            if (env_spl && sp != atoi(env_spl) && !(atoi(env_spl) > nch && sp == nch)) continue;
            const int per = ceil_div(nch, sp);
            if (sp > 1 && ceil_div(nch, per) != sp) continue;          // same per-block chain as a smaller factor
            const long blocks = blocks1 * sp;
            const long cap = pair && blocks > 256 ? 512 : 256;
            const double t_iter = (0.35 + bytes / 68e3) * (cap == 512 ? 1.2 : 1.0);
            const double rounds = (double)((blocks + cap - 1) / cap);
            double t = rounds * (5.0 + per * t_iter);
            static const char* env_pen = getenv("STEGAN_BF16_SPLIT_PEN");
            if (sp > 1) t += (env_pen ? atof(env_pen) : 1.0) * (6.0 + (sp + 1) * (double)out_elems * 4.0 / 3e6);
            // In the train step three or four streams keep every CU's block slots taken, so what a launch costs the step is
            // not only its own latency but the slot time it occupies: blocks x block time over the chip's ~512 slots
            // (STEGAN_BF16_SLOT_W weighs it; 0 = latency only, the round-2 model)
            static const char* env_sw = getenv("STEGAN_BF16_SLOT_W");
            static const double slot_w = env_sw ? atof(env_sw) : 0.0;
            t += slot_w * (double)blocks * (5.0 + per * t_iter) / 512.0;
            if (t < best - 1e-9) { best = t; TM = tm; TN = tn; splits = sp; gm = g2; }
        }
    }
    if (!gm.ok) return 1;
    const int BM = 64 * TM, BN = 64 * TN, NCB = 1 << gm.logNCB, CCH = 8 * NCB;
    p.CC = CCH; p.logCC = gm.logNCB + 3; p.SEG = gm.SEG; p.NSEG = gm.NSEG; p.XROW = gm.XROW; p.WR = 0; p.U0 = U;
    p.TG = ntaps_max; p.NTG = 1; p.fdNTG = make_fastdiv(1);
    p.fdXROW = make_fastdiv((uint32_t)gm.XROW); p.fdSEG = make_fastdiv((uint32_t)gm.SEG);
    p.fdU0 = make_fastdiv((uint32_t)U); p.fdU1 = make_fastdiv((uint32_t)(U > 1 ? U - 1 : 1));
    h.NG = gm.NG; h.fdNG = make_fastdiv((uint32_t)gm.NG); h.ntapsW = gm.ntW; h.xit = gm.xit; h.wit = gm.wit;
    h.xnit = xh ? ceil_div(NCB * gm.XROW, 256) : 0;
    if (p.out_h && (p.out_btc || p.Cdst % 8 != 0 || p.M_g % 8 != 0)) p.out_h = nullptr;      // whole 8-channel blocks only
    p.out_cb = p.Cdst / 8;
    h.nstage = 2;
    if (xh) {
        // stages of the chunk ring.  TWO by default: with 3 / 4 stages (two / three chunks in flight, counted vmcnt) every
        // layer got SLOWER -- the train step 6.90 / 7.49 / 7.75 ms with 2 / 3 / 4 stages -- because a chunk is not waiting
        // for its own round trip: the CU's fetch path from L2 delivers ~17 bytes per cycle whatever is queued (producer
        // stamps: a wave-instruction of 1 KiB takes ~120-170 cycles to issue once two chunks are outstanding; an iteration
        // is issue 1 000-1 250 + landing 500 cycles for 16-33 KiB), so more chunks in flight only land later
        // (STEGAN_BF16_STAGES: experiments)
        static const char* env_st = getenv("STEGAN_BF16_STAGES");
        const size_t stage = gm.lds / 2;
        const int per_block = ceil_div(ceil_div(p.C2_g, CCH), splits);
        int ns = 2;
        if (env_st) ns = atoi(env_st);
        if (ns > 4) ns = 4;
        if (ns > per_block) ns = per_block;
        if (ns < 2) ns = 2;
        while (ns > 2 && (size_t)ns * stage > 160 * 1024) --ns;
        h.nstage = ns;
        gm.lds = (size_t)ns * stage;
    }
    {   // the epilogue's LDS: hand-over of the tiles' second halves (4 positions x TM TN tiles x 2 KiB) + 8 x 1 KiB of image scratch
        const size_t epi = (size_t)4 * TM * TN * 2048 + 8192;
        if (gm.lds < epi) gm.lds = epi;
    }
#ifdef STEGAN_BF16_ABLATE       // diagnostic builds only (scripts/build_stamps.sh): parts of the kernel switched off, WRONG results
    { const char* ed = getenv("STEGAN_BF16_DBG"); h.dbg = ed ? atoi(ed) : 0; }
#else
    h.dbg = 0;
#endif
    {
        const bool al = ((uintptr_t)p.out % 16) == 0 && (!p.res || ((uintptr_t)p.res % 16) == 0) &&
                        (!p.mask || ((uintptr_t)p.mask % 16) == 0) && (!workspace || ((uintptr_t)workspace % 16) == 0);
        // unit output stride (forward, or the input gradient of a stride-1 conv), whole runs of 4 columns per row
        h.vec_out = al && !p.out_btc && (mode == 0 || d->stride == 1) && U % 4 == 0 && p.Tdst % 4 == 0 &&
                    (p.ups == 0 || p.Tmask * 2 == p.Tdst) && !STEGAN_ENV("STEGAN_BF16_NO_VEC_OUT");
    }
    h.src_bytes = (unsigned)((int64_t)d->B * p.Csrc * p.Tsrc * 4);
    const long nb = (N + BN - 1) / BN;
    const int nchunks = ceil_div(p.C2_g, CCH);
    p.m_tiles = ceil_div(p.M_g, BM);
    p.tile_m = BM; p.tile_n = BN;
    const long base = nb * p.m_tiles * d->groups * phases;
    p.splits = splits;
    p.chunks_per_split = ceil_div(nchunks, splits);
    p.splits = ceil_div(nchunks, p.chunks_per_split);
    p.partial = workspace;
    p.mix_S = 0; p.mix_full = 0; p.mix_cps = 0; p.nbx = (int)nb; p.ny = p.m_tiles * d->groups;
#ifdef STEGAN_STAMPS             // diagnostic builds only: the stamp buffer's device address (decimal) from the environment
    { const char* se = getenv("STEGAN_STAMP_PTR"); p.stamps = se ? (unsigned long long*)strtoull(se, nullptr, 10) : nullptr; }
#else
    p.stamps = nullptr;
#endif
    p.fdMT = make_fastdiv((uint32_t)p.m_tiles); p.fdSP = make_fastdiv((uint32_t)p.splits);
    p.fdNBX = make_fastdiv((uint32_t)p.nbx); p.fdNY = make_fastdiv((uint32_t)p.ny);
    p.fdMS = make_fastdiv(1);
    {
        // fewest bytes through the fabric: an XCD rectangle of ny / gm M tiles x R / gn rest ids reads its share of W once
        // and its share of X once, so over the chip X is read gm times (M parts inside one group share no X: / groups) and W
        // gn times (rest parts that hold different K splits / phases share no W: / nz).  Measured with FETCH_SIZE on eight
        // layers (scripts/traffic_probe.py): the model is within 5 % of the counter; the layers' run times do NOT follow
        // it (+-3 % over a 4x range of traffic: the Infinity Cache absorbs the re-reads) -- the choice lowers fabric load for
        // whatever runs beside the kernel, it is not a speed-up of the kernel itself.
        const char* env_x = STEGAN_ENV("STEGAN_BF16_XLGM");                  // experiments: -1 identity, 0 .. 3 fixed
        h.xR = p.nbx * p.splits * phases;
        static const bool ph_slow = STEGAN_ENV("STEGAN_BF16_PHASE_SLOW") != nullptr;      // experiments: the round-2 order
        h.nph = ph_slow ? 1 : phases;
        h.fdPH = make_fastdiv((uint32_t)phases);
        const double xb = (double)N * p.C2_g * d->groups * (xh ? 2 : 4), wb = (double)p.M_g * d->groups * p.C2_g * d->K * 2;
        const int nz = p.splits * phases;
        double best = 0;
        h.xlgm = 0;
        for (int l = 0; l <= 3; ++l) {
            const int gm_ = 1 << l, gn_ = 8 >> l;
            if (l > 0 && (gm_ > p.ny || gn_ > h.xR)) continue;
            const double c = xb * (gm_ > d->groups ? (double)gm_ / d->groups : 1.0) + wb * (gn_ > nz ? (double)gn_ / nz : 1.0);
            if (l == 0 || c < best) { best = c; h.xlgm = l; }
        }
        if (env_x) h.xlgm = atoi(env_x);
        { const char* er = STEGAN_ENV("STEGAN_BF16_ROT"); h.rot = er ? atoi(er) : 1; }
        { const char* ec = STEGAN_ENV("STEGAN_BF16_CW0"); h.cw0 = ec ? atoi(ec) : 1; }
        while (h.xlgm > 0 && ((1 << h.xlgm) > p.ny || (8 >> h.xlgm) > h.xR)) --h.xlgm;
    }
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
        if (dbg) fprintf(stderr, "          xv=%d\n", (int)xv);
    }
    if (mc) {
        MultiH& mh = mc->mh;
        if (mh.n >= HMULTI) return 1;
        if (mh.n == 0) { mc->TM = TM; mc->TN = TN; mc->logNCB = gm.logNCB; mc->lds = gm.lds; mc->blocks = 0; mh.first[0] = 0; mc->xh = xh; }
        if (gm.lds > mc->lds) mc->lds = gm.lds;
        mh.p[mh.n] = h;
        mc->blocks += nb * p.m_tiles * d->groups * phases;
        mh.first[++mh.n] = (int)mc->blocks;
        return STEGAN_OK;
    }
    static bool attr_done = false;
    if (!attr_done) {
        set_lds_attr<1, 1, false>(); set_lds_attr<1, 2, false>(); set_lds_attr<2, 1, false>(); set_lds_attr<2, 2, false>();
        set_lds_attr<1, 1, true>(); set_lds_attr<1, 2, true>(); set_lds_attr<2, 1, true>(); set_lds_attr<2, 2, true>();
        set_lds_attr<1, 1, false, true>(); set_lds_attr<1, 2, false, true>(); set_lds_attr<2, 1, false, true>(); set_lds_attr<2, 2, false, true>();
        attr_done = true;
    }
    dim3 grid((unsigned)(nb * p.m_tiles * d->groups * phases * p.splits));
#define HB_GO(TM_, TN_) { if (xh) launch_variant<TM_, TN_, false, true>(gm.logNCB, grid, gm.lds, s, h); else if (xv) launch_variant<TM_, TN_, true>(gm.logNCB, grid, gm.lds, s, h); else launch_variant<TM_, TN_, false>(gm.logNCB, grid, gm.lds, s, h); }
    if (TM == 2 && TN == 2) HB_GO(2, 2)
    else if (TM == 2) HB_GO(2, 1)
    else if (TN == 2) HB_GO(1, 2)
    else HB_GO(1, 1)
#undef HB_GO
#ifdef STEGAN_BF16_ABLATE       // diagnostic builds only: launches skipped to see what they cost -- WRONG results
    static const int whatif = getenv("STEGAN_WHATIF_C") ? atoi(getenv("STEGAN_WHATIF_C")) : 0;
#else
    constexpr int whatif = 0;
#endif
    if (p.splits > 1 && !(whatif & 1)) {
        long eb = (((p.Tdst & 3) == 0 ? out_elems / 4 : out_elems) + 255) / 256;
        if (eb > 4096) eb = 4096;
        hipLaunchKernelGGL(conv_splitk_epilogue_kernel, dim3((unsigned)eb), dim3(256), 0, s, p);
    }
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

template <int TM, int TN>
void multi_go(MultiCollectH& mc, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
#define HB_ATTR(L_, V_, H_) (void)hipFuncSetAttribute((const void*)conv_bf16_multi_kernel<TM, TN, L_, V_, H_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        HB_ATTR(0, true, false) HB_ATTR(1, true, false) HB_ATTR(2, true, false) HB_ATTR(3, true, false)
        HB_ATTR(0, false, false) HB_ATTR(1, false, false) HB_ATTR(2, false, false) HB_ATTR(3, false, false)
        HB_ATTR(0, false, true) HB_ATTR(1, false, true) HB_ATTR(2, false, true) HB_ATTR(3, false, true) HB_ATTR(4, false, true)
#undef HB_ATTR
        attr = true;
    }
    dim3 grid((unsigned)mc.blocks);
    if (mc.xh) launch_variant_multi<TM, TN, false, true>(mc.logNCB, grid, mc.lds, s, mc.mh);
    else if (mc.xv) launch_variant_multi<TM, TN, true>(mc.logNCB, grid, mc.lds, s, mc.mh);
    else launch_variant_multi<TM, TN, false>(mc.logNCB, grid, mc.lds, s, mc.mh);
}

int launch_multi_h(MultiCollectH& mc, hipStream_t s) {
    if (mc.TM == 2) multi_go<2, 1>(mc, s);
    else if (mc.TN == 2) multi_go<1, 2>(mc, s);
    else multi_go<1, 1>(mc, s);
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

extern "C" int64_t stegan_conv_prep_bf16_tiles(const stegan_conv_desc* d) {
    if (!d || d->groups <= 0 || d->K <= 0) return -1;
    return prep_tiles(d->groups, d->Cout / d->groups, d->Cin / d->groups, d->K);
}

extern "C" int stegan_conv_prep_bf16(const stegan_conv_desc* d, const float* v, const float* scale, uint16_t* wf,
                                     uint16_t* wb, void* stream) {
    STEGAN_CHECK_ARG(d && v && wf && wb && d->groups > 0 && d->K > 0 && d->Cin % d->groups == 0 && d->Cout % d->groups == 0);
    PrepH t;
    fill_prep(t, d);
    t.v = v; t.scale = scale; t.wf = wf; t.wb = wb;
    STEGAN_CHECK_ARG(prep_rows(t.K) * (prep_cit(t.K) * t.K + 1) <= PREP_LDS_FLOATS);
    long blocks = prep_tiles(t.G, t.Cout_g, t.Cin_g, t.K);
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

// Forward / input-gradient conv with bf16 activation images (round 3): ``xh`` (nullable) = the source as an image
// [B][ceil(Cin / 8)][Tin][8] written by the kernel that produced it -- staged by LDS-DMA instead of being converted from
// fp32; ``yh`` (nullable) = where to write the image of the output beside the fp32 tensor (whole 8-channel blocks only:
// ignored otherwise).  The fp32 arguments are as for stegan_conv1d_fwd_h / stegan_conv1d_bwd_data_h.
extern "C" int stegan_conv1d_fwd_hx(const stegan_conv_desc* d, const float* x, const uint16_t* xh, const uint16_t* wfh,
                                    const float* scale, const float* bias, const float* residual, float* y, uint16_t* yh,
                                    float* workspace, int64_t workspace_floats, void* stream) {
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
    h.xh = xh; p.out_h = yh;
    return launch_bf16(d, 0, h, workspace, workspace_floats, (hipStream_t)stream);
}

extern "C" int stegan_conv1d_bwd_data_hx(const stegan_conv_desc* d, const float* dy, const uint16_t* dyh, const uint16_t* wbh,
                                         const float* scale, const float* mask_src, const float* add_src, float* dx,
                                         uint16_t* dxh, float* workspace, int64_t workspace_floats, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    STEGAN_CHECK_ARG(dy && wbh && dx);
    const bool lrelu_mask = d->in_act == STEGAN_ACT_LRELU;         // x is a leaky-ReLU output: its derivative rides here
    STEGAN_CHECK_ARG((d->in_act != STEGAN_ACT_RELU && !lrelu_mask) || mask_src);
    STEGAN_CHECK_ARG(!lrelu_mask || d->upsample == 1);
    HP h;
    DirP& p = h.d;
    p.src = dy; p.wt = nullptr; p.scale = scale; p.bias = nullptr; p.res = add_src; p.out = dx;
    p.mask = (d->in_act == STEGAN_ACT_RELU || lrelu_mask) ? mask_src : nullptr;
    p.mask_lrelu = lrelu_mask ? 1 : 0;
    p.Csrc = d->Cout; p.Cdst = d->Cin; p.Tsrc = d->Tout; p.Tq = d->Tout; p.Tdst = d->Tin * d->upsample;
    p.Tmask = d->Tin; p.ups = d->upsample == 2 ? 1 : 0;
    p.C2_g = d->Cout / d->groups; p.M_g = d->Cin / d->groups; p.LD = 0; p.zero_off = 0;
    p.in_relu = 0; p.out_act = 0; p.out_btc = 0;
    h.wh = wbh; h.Mpad = round_up(p.M_g, 128); h.Cpad = round_up(p.C2_g, 64);
    h.xh = dyh; p.out_h = dxh;
    return launch_bf16(d, 1, h, workspace, workspace_floats, (hipStream_t)stream);
}

extern "C" int stegan_conv1d_fwd_h(const stegan_conv_desc* d, const float* x, const uint16_t* wfh, const float* scale,
                                   const float* bias, const float* residual, float* y, float* workspace,
                                   int64_t workspace_floats, void* stream) {
    return stegan_conv1d_fwd_hx(d, x, nullptr, wfh, scale, bias, residual, y, nullptr, workspace, workspace_floats, stream);
}

extern "C" int stegan_conv1d_bwd_data_h(const stegan_conv_desc* d, const float* dy, const uint16_t* wbh, const float* scale,
                                        const float* mask_src, const float* add_src, float* dx, float* workspace,
                                        int64_t workspace_floats, void* stream) {
    return stegan_conv1d_bwd_data_hx(d, dy, nullptr, wbh, scale, mask_src, add_src, dx, nullptr, workspace, workspace_floats,
                                     stream);
}

// Up to 5 independent convolutions in ONE launch (the same layer of the period discriminators), bf16 operands: see
// stegan_conv1d_fwd_multi.  Returns 1 when the problems cannot share a kernel variant (the caller launches them one by one).
extern "C" int stegan_conv1d_fwd_multi_hx(int32_t n, const stegan_conv_desc* const* descs, const float* const* x,
                                          const uint16_t* const* xh, const uint16_t* const* wfh, const float* const* scale,
                                          const float* const* bias, const float* const* residual, float* const* y,
                                          uint16_t* const* yh, void* stream) {
    STEGAN_CHECK_ARG(n >= 1 && n <= HMULTI && descs && x && wfh && y);
    MultiCollectH mc;
    mc.mh.n = 0; mc.TM = 0; mc.TN = 0; mc.logNCB = -1; mc.xv = true; mc.xh = xh != nullptr;
    for (int i = 0; i < n; ++i) {
        if (!descs[i] || descs[i]->upsample != 1 || !xv_eligible(descs[i]->Tin, x[i], false)) mc.xv = false;   // (one granule per launch)
        if (xh && (!xh[i] || !descs[i] || (descs[i]->Cin / descs[i]->groups) % 8 != 0)) mc.xh = false;
    }
    for (int i = 0; i < n; ++i) {
        const stegan_conv_desc* d = descs[i];
        int rc = check_desc(d);
        if (rc) return rc;
        STEGAN_CHECK_ARG(x[i] && wfh[i] && y[i]);
        STEGAN_CHECK_ARG(d->in_act == STEGAN_ACT_NONE || d->in_act == STEGAN_ACT_RELU);
        STEGAN_CHECK_ARG(d->out_act >= STEGAN_ACT_NONE && d->out_act <= STEGAN_ACT_TANH);
        const float* res = residual ? residual[i] : nullptr;
        STEGAN_CHECK_ARG(!(d->out_btc && res));
        HP h;
        DirP& p = h.d;
        p.src = x[i]; p.wt = nullptr; p.scale = scale ? scale[i] : nullptr; p.bias = bias ? bias[i] : nullptr; p.res = res;
        p.mask = nullptr; p.out = y[i];
        p.Csrc = d->Cin; p.Cdst = d->Cout; p.Tsrc = d->Tin; p.Tq = d->Tin * d->upsample; p.Tdst = d->Tout;
        p.Tmask = 0; p.ups = d->upsample == 2 ? 1 : 0;
        p.C2_g = d->Cin / d->groups; p.M_g = d->Cout / d->groups; p.LD = 0; p.zero_off = 0;
        p.in_relu = d->in_act == STEGAN_ACT_RELU; p.out_act = d->out_act; p.out_btc = d->out_btc;
        h.wh = wfh[i]; h.Mpad = round_up(p.M_g, 128); h.Cpad = round_up(p.C2_g, 64);
        h.xh = mc.xh ? xh[i] : nullptr; p.out_h = yh ? yh[i] : nullptr;
        rc = launch_bf16(d, 0, h, nullptr, 0, (hipStream_t)stream, &mc);
        if (rc) return rc;
    }
    return launch_multi_h(mc, (hipStream_t)stream);
}

extern "C" int stegan_conv1d_fwd_multi_h(int32_t n, const stegan_conv_desc* const* descs, const float* const* x,
                                         const uint16_t* const* wfh, const float* const* scale, const float* const* bias,
                                         const float* const* residual, float* const* y, void* stream) {
    return stegan_conv1d_fwd_multi_hx(n, descs, x, nullptr, wfh, scale, bias, residual, y, nullptr, stream);
}

extern "C" int stegan_conv1d_bwd_data_multi_hx(int32_t n, const stegan_conv_desc* const* descs, const float* const* dy,
                                               const uint16_t* const* dyh, const uint16_t* const* wbh,
                                               const float* const* mask_src, const float* const* add_src, float* const* dx,
                                               uint16_t* const* dxh, void* stream) {
    STEGAN_CHECK_ARG(n >= 1 && n <= HMULTI && descs && dy && wbh && dx);
    MultiCollectH mc;
    mc.mh.n = 0; mc.TM = 0; mc.TN = 0; mc.logNCB = -1; mc.xv = true; mc.xh = dyh != nullptr;
    for (int i = 0; i < n; ++i) {
        if (!descs[i] || !xv_eligible(descs[i]->Tout, dy[i], false)) mc.xv = false;
        if (dyh && (!dyh[i] || !descs[i] || (descs[i]->Cout / descs[i]->groups) % 8 != 0)) mc.xh = false;
    }
    for (int i = 0; i < n; ++i) {
        const stegan_conv_desc* d = descs[i];
        int rc = check_desc(d);
        if (rc) return rc;
        STEGAN_CHECK_ARG(dy[i] && wbh[i] && dx[i]);
        const float* mk = mask_src ? mask_src[i] : nullptr;
        const bool lrelu_mask = d->in_act == STEGAN_ACT_LRELU;
        STEGAN_CHECK_ARG((d->in_act != STEGAN_ACT_RELU && !lrelu_mask) || mk);
        STEGAN_CHECK_ARG(!lrelu_mask || d->upsample == 1);
        HP h;
        DirP& p = h.d;
        p.src = dy[i]; p.wt = nullptr; p.scale = nullptr; p.bias = nullptr; p.res = add_src ? add_src[i] : nullptr; p.out = dx[i];
        p.mask = (d->in_act == STEGAN_ACT_RELU || lrelu_mask) ? mk : nullptr;
        p.mask_lrelu = lrelu_mask ? 1 : 0;
        p.Csrc = d->Cout; p.Cdst = d->Cin; p.Tsrc = d->Tout; p.Tq = d->Tout; p.Tdst = d->Tin * d->upsample;
        p.Tmask = d->Tin; p.ups = d->upsample == 2 ? 1 : 0;
        p.C2_g = d->Cout / d->groups; p.M_g = d->Cin / d->groups; p.LD = 0; p.zero_off = 0;
        p.in_relu = 0; p.out_act = 0; p.out_btc = 0;
        h.wh = wbh[i]; h.Mpad = round_up(p.M_g, 128); h.Cpad = round_up(p.C2_g, 64);
        h.xh = mc.xh ? dyh[i] : nullptr; p.out_h = dxh ? dxh[i] : nullptr;
        rc = launch_bf16(d, 1, h, nullptr, 0, (hipStream_t)stream, &mc);
        if (rc) return rc;
    }
    return launch_multi_h(mc, (hipStream_t)stream);
}

extern "C" int stegan_conv1d_bwd_data_multi_h(int32_t n, const stegan_conv_desc* const* descs, const float* const* dy,
                                              const uint16_t* const* wbh, const float* const* mask_src,
                                              const float* const* add_src, float* const* dx, void* stream) {
    return stegan_conv1d_bwd_data_multi_hx(n, descs, dy, nullptr, wbh, mask_src, add_src, dx, nullptr, stream);
}
