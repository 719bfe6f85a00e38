// Weight gradient of the stride-1, ungrouped layers, one tile per CU (round 5) -- the companion of conv_tile.hip.
//
//     dW[co][ci][k] += sum_{b, t}  dy[b, co, t] * act(x[b, ci, (t + k dil - pad) >> ups])
//
// Why: the generator's backward pass is bound by its weight gradients, not by the input-gradient chain -- 41 layers x
// (22 us launch + 7 us slab sum) on four side streams.  Same recipe as conv_tile.hip:
//
//   * a block owns 48 output channels x 48 input channels x all live taps (<= 5) of dW for ONE slice of the batch; tiles x
//     slices ~ one block per CU.  Slices > 1 write slabs [slice][co][ci][tap] that the caller's ordered reduce adds into dW
//     (conv_wgrad_bf16.hip); one slice adds straight into dW;
//   * both operands come as activation images [b][C / 8][T][8] and are staged by 4 producer waves with LDS-DMA into
//     [channel block][slot] planes: dy dense (the chunk's columns one after the other), x with every sample's halo (padding =
//     out-of-range offset = zeros; nearest-x2 upsample = source position >> 1);
//   * the reduction runs over (b, t): an MFMA operand needs 8 consecutive COLUMNS of one channel per lane, i.e. the transpose of
//     the image's 16-byte pieces -- read with ds_read_b64_tr_b16: lane 4 q + p of a 16-lane group addresses row (slot) q,
//     channels 4 p .. 4 p + 3 and receives 4 consecutive slots of channel (lane & 15); every row has its own address, so a tap
//     is a slot offset and the x rows of a k-step may sit in different samples' segments (T % 4 == 0: a 4-row block never
//     straddles two samples);
//   * a consumer wave owns ONE tap and every nkg-th k-step (32 columns) of the chunk: 9 accumulator tiles, 12 transposed reads
//     per 9 MFMAs.  (The first version gave every wave all taps of its k-steps -- 27 tiles, 108 accumulator registers, one
//     k-step per wave and barrier, so each k-step's LDS latency lay open -- and met tap by tap in LDS behind six barriers: the
//     stamps (scripts/tile_wgrad_stamps.py) showed 2700-3500 cycles per chunk against 1400 of DMA, and 8000 cycles for the
//     meeting.)  The waves of a tap meet once in LDS, in a fixed order (bit-reproducible); the summed tile is laid out as dW's
//     rows and added with 16-byte read-modify-writes of whole rows (48 ci x K floats are contiguous in dW), the old values
//     fetched before the meeting;
//   * the bias gradient rides in the same grid (blocks behind the tiles), as in conv_wgrad_bf16.hip.
#include "conv_tile.h"

#include <stdio.h>
#include <stdlib.h>

namespace {

typedef hop_t hx8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifdef STEGAN_OPERAND_F16
#define TILE_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x32_f16(a_, b_, c_, 0, 0, 0)
#else
#define TILE_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, b_, c_, 0, 0, 0)
#endif

constexpr unsigned G_OOB = 0x7ffff000u;
constexpr int GX_MAX = 12;                   // x DMA items per producer wave and chunk
constexpr int GY_MAX = 8;                    // dy DMA items per producer wave and chunk
constexpr int GT = 48;                       // tile: 48 x 48 x taps
constexpr int KT_MAX = 5;
constexpr int G_PRE = 5;                     // 16-byte pieces of the tile per consumer thread (48 x 12 Keff pieces over ncons x 64 threads)
constexpr int G_LDS_MAX = 160 * 1024 - 256;

struct GK {
    const unsigned short* xh;
    const unsigned short* dyh;
    const float* dy;              // bias gradient
    float* dw;                    // slices == 1: accumulate here
    float* db;
    float* partial;               // slices > 1: slabs [slice][Cout][Cin][Keff]
    const float* prev_partial;    // pending slab sum of an earlier launch (conv_tile.h)
    float* prev_dw;
    long prev_n4;
    int prev_slices;
    unsigned xh_bytes, dyh_bytes;
    int B, Cin, Cout, K, k_lo, Keff, Tin, Tout, ups, relu;
    int CBx, CBy;                 // channel blocks per sample of the two images
    int off_min, step;            // x slot 0 of a segment is source position off_min; taps are `step` slots apart
    int ncons, nthreads, nkg;     // consumer waves = Keff taps x nkg k-step groups
    int CS, cols, KSC;            // samples per chunk, dense columns per chunk, k-steps per chunk
    int SEGW, XSLOT, DSLOT;       // slots per x segment, per x plane, per dy plane (multiples of 64)
    int nX, nY, xseg, yseg;       // DMA items per chunk, items per plane
    int NS, ystage, stage;
    int slices, bps, nch;         // batch slices, samples per slice, chunks per slice
    int MT, NTl, tiles;
    int x_samp, y_samp;           // bytes between samples of the images
    int bias_first, bias_vec, bias_mode;   // 0 none | 1 inside the tile blocks (one slice) | 2 blocks behind the tiles
    FastDiv fdT, fdSEGW, fdXseg, fdYseg, fdMT, fdTiles, fdTb, fdNkg, fdRow4;
#ifdef STEGAN_STAMPS
    unsigned long long* stamps;
#endif
};

__device__ __forceinline__ void g_wait_vmcnt(int n) {
#define GV_CASE(k_) case k_: asm volatile("s_waitcnt vmcnt(" #k_ ")" ::: "memory"); break;
    switch (n) {
        GV_CASE(0) GV_CASE(1) GV_CASE(2) GV_CASE(3) GV_CASE(4) GV_CASE(5) GV_CASE(6) GV_CASE(7) GV_CASE(8) GV_CASE(9)
        GV_CASE(10) GV_CASE(11) GV_CASE(12) GV_CASE(13) GV_CASE(14) GV_CASE(15) GV_CASE(16) GV_CASE(17) GV_CASE(18)
        GV_CASE(19) GV_CASE(20) GV_CASE(21) GV_CASE(22) GV_CASE(23) GV_CASE(24) GV_CASE(25) GV_CASE(26) GV_CASE(27)
        GV_CASE(28) GV_CASE(29) GV_CASE(30) GV_CASE(31) GV_CASE(32) GV_CASE(33) GV_CASE(34) GV_CASE(35) GV_CASE(36)
        GV_CASE(37) GV_CASE(38) GV_CASE(39) GV_CASE(40)
        default: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
    }
#undef GV_CASE
}

__device__ __forceinline__ unsigned g_sdiv(unsigned n, const FastDiv& f) {
    return (unsigned)((((unsigned long long)n * f.mul) >> 32) + n) >> f.shift;
}

// (body shared by the single-problem kernel and the multi-problem one: `bid0` is the block's index inside its problem)
__device__ __forceinline__ void conv_tile_wgrad_body(const GK& k, const unsigned bid0) {
    extern __shared__ __align__(16) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef __attribute__((address_space(3))) s16x4* tr_ptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wave >= k.ncons + 4) return;          // (multi-problem launches size the block for the widest problem)

#ifdef STEGAN_STAMPS
#define GSTAMP(i_) if (k.stamps && threadIdx.x == 0) k.stamps[(size_t)blockIdx.x * 16 + (i_)] = __builtin_amdgcn_s_memtime();
#define GPSTAMP(i_) if (k.stamps && (int)threadIdx.x == k.ncons * 64) k.stamps[(size_t)blockIdx.x * 16 + (i_)] = __builtin_amdgcn_s_memtime();
    if (k.stamps && threadIdx.x == 0) k.stamps[(size_t)blockIdx.x * 16 + 8] = __builtin_amdgcn_s_memrealtime();
#else
#define GSTAMP(i_)
#define GPSTAMP(i_)
#endif
    GSTAMP(0)
    if ((int)bid0 >= k.bias_first) {                                  // bias-gradient blocks (bias_mode 2): one output channel each
        if (tid >= 256) return;
        bias_grad_body(k.dy, k.db, k.B, k.Cout, k.Tout, k.B, k.fdTb, k.bias_vec, (int)bid0 - k.bias_first, 0);
        return;
    }
    // block -> (slice, tile): tiles of one slice are neighbours (they share the slice's columns of dy / x in the L2s)
    const unsigned bid = bid0;
    const unsigned slice = g_sdiv(bid, k.fdTiles), tile = bid - slice * (unsigned)k.tiles;
    const unsigned nt = g_sdiv(tile, k.fdMT), mt = tile - nt * (unsigned)k.MT;
    const int m0 = (int)mt * GT, n0 = (int)nt * GT;                  // first output channel, first input channel
    const int bs0 = (int)slice * k.bps;                               // first sample of the slice
    const int nch = k.nch, NS = k.NS;
    const bool producer = wave >= k.ncons;

    if (producer) {
        const int pw = wave - k.ncons;
        unsigned long long xbase = (unsigned long long)k.xh, ybase = (unsigned long long)k.dyh;
        int xbytes = (int)k.xh_bytes, ybytes = (int)k.dyh_bytes;
        asm volatile("" : "+s"(xbase), "+s"(ybase), "+s"(xbytes), "+s"(ybytes));
        const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, xbytes, 0x00020000);
        const auto rsrc_y = __builtin_amdgcn_make_buffer_rsrc((void*)ybase, 0, ybytes, 0x00020000);
        // per-lane offsets of this wave's items (i = pw + 4 n), relative to the chunk's first sample; the chunk is a scalar.
        // The first chunk's loads leave as soon as their offsets exist (dy, then x): the set-up of the other half runs under them
        unsigned xv[GX_MAX], yv[GY_MAX];
        const int xlive = k.CS * k.SEGW, Te = k.Tin << k.ups;
        const int nXw = k.nX > pw ? (k.nX - pw + 3) >> 2 : 0, nYw = k.nY > pw ? (k.nY - pw + 3) >> 2 : 0;
        const int ipc = nXw + nYw;
        int chunk_x = k.CS * k.x_samp, chunk_y = k.CS * k.y_samp, stage_b = k.stage, ystage_b = k.ystage;
        int x0 = bs0 * k.x_samp, y0 = bs0 * k.y_samp;
        int nXi = nXw, nYi = nYw;
        asm volatile("" : "+s"(chunk_x), "+s"(chunk_y), "+s"(stage_b), "+s"(ystage_b), "+s"(x0), "+s"(y0));
#define GL_X_IT(n_) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(Xd_ + (n_) * 4096), 16, (int)xv[n_], cx_, 0, 0);
#define GL_Y_IT(n_) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (lds_ptr_t)(Yd_ + (n_) * 4096), 16, (int)yv[n_], cy_, 0, 0);
#define GL_ISSUE_Y(c_, st_)                                                                                     \
        {                                                                                                       \
            char* Yd_ = smem + (st_) * stage_b + pw * 1024;                                                     \
            const int cy_ = y0 + (c_) * chunk_y;                                                                \
            int nYw_ = nYi;                                                                                     \
            asm volatile("" : "+s"(nYw_));                                                                      \
            if (nYw_ > 0) { GL_Y_IT(0) if (nYw_ > 1) { GL_Y_IT(1) if (nYw_ > 2) { GL_Y_IT(2) if (nYw_ > 3) { GL_Y_IT(3) \
            if (nYw_ > 4) { GL_Y_IT(4) if (nYw_ > 5) { GL_Y_IT(5) if (nYw_ > 6) { GL_Y_IT(6) if (nYw_ > 7) { GL_Y_IT(7) \
            } } } } } } } }                                                                                     \
        }
#define GL_ISSUE_X(c_, st_)                                                                                     \
        {                                                                                                       \
            char* Xd_ = smem + (st_) * stage_b + pw * 1024 + ystage_b;                                          \
            const int cx_ = x0 + (c_) * chunk_x;                                                                \
            int nXw_ = nXi;                                                                                     \
            asm volatile("" : "+s"(nXw_));                                                                      \
            if (nXw_ > 0) { GL_X_IT(0) if (nXw_ > 1) { GL_X_IT(1) if (nXw_ > 2) { GL_X_IT(2) if (nXw_ > 3) { GL_X_IT(3) \
            if (nXw_ > 4) { GL_X_IT(4) if (nXw_ > 5) { GL_X_IT(5) if (nXw_ > 6) { GL_X_IT(6) if (nXw_ > 7) { GL_X_IT(7) \
            if (nXw_ > 8) { GL_X_IT(8) if (nXw_ > 9) { GL_X_IT(9) if (nXw_ > 10) { GL_X_IT(10) if (nXw_ > 11) { GL_X_IT(11) \
            } } } } } } } } } } } }                                                                             \
        }
#define GL_ISSUE(c_, st_) GL_ISSUE_Y(c_, st_) GL_ISSUE_X(c_, st_)
        static_assert(GX_MAX == 12 && GY_MAX == 8, "the item switches are written out for 12 / 8 items");
#pragma unroll
        for (int n = 0; n < GY_MAX; ++n) {
            const unsigned i = (unsigned)(pw + 4 * n);
            const unsigned cb = g_sdiv(i, k.fdYseg), sg = i - cb * (unsigned)k.yseg;
            const unsigned slot = sg * 64u + (unsigned)lane;
            const unsigned g = fd_div(slot, k.fdT), t = slot - g * (unsigned)k.Tout;
            const bool ok = ((int)slot < k.cols) & ((int)cb < 6) & (m0 / 8 + (int)cb < k.CBy);
            yv[n] = ok ? (unsigned)((((int)g * k.CBy + m0 / 8 + (int)cb) * k.Tout + (int)t) * 16) : G_OOB;
        }
        if (nch > 0) GL_ISSUE_Y(0, 0)
#pragma unroll
        for (int n = 0; n < GX_MAX; ++n) {
            const unsigned i = (unsigned)(pw + 4 * n);
            const unsigned cb = g_sdiv(i, k.fdXseg), sg = i - cb * (unsigned)k.xseg;
            const unsigned slot = sg * 64u + (unsigned)lane;
            const unsigned g = fd_div(slot, k.fdSEGW), sl = slot - g * (unsigned)k.SEGW;
            const int q = k.off_min + (int)sl;
            const bool ok = ((int)slot < xlive) & (q >= 0) & (q < Te) & ((int)cb < 6) & (n0 / 8 + (int)cb < k.CBx);
            xv[n] = ok ? (unsigned)((((int)g * k.CBx + n0 / 8 + (int)cb) * k.Tin + (q >> k.ups)) * 16) : G_OOB;
        }
        GPSTAMP(6)
        if (nch > 0) GL_ISSUE_X(0, 0)
        int st_in = 1 % NS, issued = 1;
        for (int j = 0; j < nch; ++j) {
            g_wait_vmcnt((issued - 1 - j) * ipc);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            while (issued < nch && issued < j + NS) {
                GL_ISSUE(issued, st_in)
                st_in = st_in + 1 == NS ? 0 : st_in + 1;
                ++issued;
            }
        }
        GPSTAMP(7)
#undef GL_ISSUE
#undef GL_ISSUE_X
#undef GL_ISSUE_Y
#undef GL_X_IT
#undef GL_Y_IT
        return;
    }

    // ================= consumers =================
    const int l15 = lane & 15, kg = lane >> 4, q4 = l15 >> 2, p4 = l15 & 3;
    const int ncw = k.ncons, KSC = k.KSC, ntaps = k.Keff, nkg = k.nkg;
    const int tp = (int)g_sdiv((unsigned)wave, k.fdNkg), kgp = wave - tp * nkg;        // this wave's tap, its k-step group
    // lane-constant parts of the transposed reads: channel quad p4 of group i sits in plane 2 i + (p4 >> 1), bytes 8 (p4 & 1)
    const int pl = (p4 >> 1), sub = (p4 & 1) * 8;
    const int a_pl = pl * k.DSLOT * 16 + sub, b_pl = pl * k.XSLOT * 16 + sub + tp * k.step * 16;
    const int a_grp = 2 * k.DSLOT * 16, b_grp = 2 * k.XSLOT * 16;      // bytes between the planes of neighbouring channel groups

    // ---- the pending slab sum of the layer before (same stream: its launch is complete).  Same order of adds as
    // conv_wgrad_bf16_reduce_kernel: identical bits
    if (k.prev_n4 > 0) {
        const f32x4* src4 = (const f32x4*)k.prev_partial;
        f32x4* dw4 = (f32x4*)k.prev_dw;
        const long n4 = k.prev_n4, stride = (long)k.tiles * k.slices * (ncw * 64);
        const int ps = k.prev_slices;
        // three pieces a thread per trip, every load of the trip in flight at once (one memory latency, under the producers'
        // set-up; a piece per trip was 2-3 dependent latencies: as long as the reduce launch it replaces)
        for (long i0 = (long)bid * (ncw * 64) + tid; i0 < n4; i0 += 3 * stride) {
            f32x4 o[3], v[3][4];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const long i = i0 + q * stride;
#pragma unroll
                for (int u = 0; u < 4; ++u) v[q][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (i < n4) {
                    o[q] = dw4[i];
                    if (ps == 4) {                                    // the usual slicing: four loads, no loop
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[q][u] = src4[(size_t)u * n4 + i];
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const long i = i0 + q * stride;
                if (i < n4) {
                    if (ps != 4) {
                        int sidx = 0;
                        for (; sidx + 4 <= ps; sidx += 4) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) v[q][u] += src4[(size_t)(sidx + u) * n4 + i];
                        }
                        for (; sidx < ps; ++sidx) v[q][0] += src4[(size_t)sidx * n4 + i];
                    }
                    dw4[i] = o[q] + ((v[q][0] + v[q][1]) + (v[q][2] + v[q][3]));
                }
            }
        }
    }
    // ---- while the producers set up and the first chunk is in flight (about 4500 cycles): the tile's old values (one slice:
    // dW += tile) are fetched -- they meet the sums after the LDS exchange -- and the bias gradient is summed.  A row of the tile is
    // 48 * Keff consecutive floats of its row of dW when every tap is live and the tile is whole: 12 Keff 16-byte pieces
    const int nthr = ncw * 64;
    const int Keff = ntaps, K = k.K;
    const int rowf = GT * ntaps;
    float* const dst = k.slices > 1 ? k.partial + (size_t)slice * k.Cout * k.Cin * Keff : k.dw;
    const int dK = k.slices > 1 ? Keff : K, dk0 = k.slices > 1 ? 0 : k.k_lo;       // taps per (co, ci) of the destination
    const bool acc_into = k.slices == 1;
    const int co_n = min(GT, k.Cout - m0), ci_n = min(GT, k.Cin - n0);
    const bool vec = dK == Keff && ci_n == GT && ((k.Cin * dK) & 3) == 0 && ((n0 * dK) & 3) == 0;
    const int v4row = GT * Keff / 4, total4 = co_n * v4row;
    f32x4 old[G_PRE];
    if (vec && acc_into) {
#pragma unroll
        for (int u = 0; u < G_PRE; ++u) {
            const int e = tid + u * nthr;
            const int r = (int)fd_div((uint32_t)e, k.fdRow4), c4 = e - r * v4row;
            if (e < total4) old[u] = *(const f32x4*)(dst + ((size_t)(m0 + r) * k.Cin + n0) * dK + c4 * 4);
        }
    }
    // ---- bias gradient inside the tile blocks (one slice: B * T is small -- the T = 12 ... 48 layers): the NTl tiles of an output
    // channel group share its 48 channels; consumer wave w of tile nt sums channel m0 + nt * cpb + w over the whole batch in
    // fp32 (fixed order) and adds it into db -- no extra blocks (which would each hold a whole block's LDS: 768 of them cost
    // three more rounds of blocks)
    if (k.bias_mode == 1 && slice == 0) {
        const int cpb = (GT + k.NTl - 1) / k.NTl;                     // channels per tile block (<= 2 ncons: host)
        for (int cw = wave; cw < cpb; cw += ncw) {
            const int cl = (int)nt * cpb + cw;                        // channel inside the group
            if (cl >= GT || m0 + cl >= k.Cout) break;
            const int c = m0 + cl, T4 = k.Tout >> 2, n4 = k.B * T4;
            const float* base = k.dy + (size_t)c * k.Tout;
            const size_t bstr = (size_t)k.Cout * k.Tout;
            float sacc = 0.f;
            int r = lane;
            for (; r + 7 * 64 < n4; r += 8 * 64) {                    // eight loads in flight (B * T up to 16 k: 64 a lane)
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int ru = r + u * 64;
                    const int b = (int)fd_div((uint32_t)ru, k.fdTb), t4 = ru - b * T4;
                    v[u] = *reinterpret_cast<const float4*>(base + (size_t)b * bstr + 4 * t4);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) sacc += (v[u].x + v[u].y) + (v[u].z + v[u].w);
            }
            for (; r < n4; r += 64) {
                const int b = (int)fd_div((uint32_t)r, k.fdTb), t4 = r - b * T4;
                const float4 v = *reinterpret_cast<const float4*>(base + (size_t)b * bstr + 4 * t4);
                sacc += (v.x + v.y) + (v.z + v.w);
            }
            sacc = wave_sum(sacc);
            if (lane == 0) k.db[c] += sacc;
        }
    }
    f32x4 acc[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bool relu = k.relu != 0;
    int st_c = 0;
    for (int jc = 0; jc < nch; ++jc) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (jc == 0) { GSTAMP(1) }
        const char* Ycur = smem + st_c * k.stage;
        const char* Xcur = Ycur + k.ystage;
        for (int s = kgp; s < KSC; s += nkg) {
            // rows (columns of the layer) 32 s + 8 kg + q4 and + 4: dense slot for dy, (sample segment, t) for x
            const unsigned c0 = (unsigned)(32 * s + 8 * kg + q4), c1 = c0 + 4u;
            const unsigned g0 = fd_div(c0, k.fdT), g1 = fd_div(c1, k.fdT);
            const int xs0 = (int)g0 * k.SEGW + (int)(c0 - g0 * (unsigned)k.Tout), xs1 = (int)g1 * k.SEGW + (int)(c1 - g1 * (unsigned)k.Tout);
            const char* ya0 = Ycur + a_pl + (int)c0 * 16;
            const char* xb0 = Xcur + b_pl + xs0 * 16;
            const int dx1 = (xs1 - xs0) * 16;
            hx8 af[3], bf[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(ya0 + i * a_grp));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(ya0 + i * a_grp + 64));
                const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                af[i] = __builtin_bit_cast(hx8, v);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const char* pb = xb0 + j * b_grp;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(pb));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(pb + dx1));
                s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                if (relu) v = __builtin_elementwise_max(v, (s16x8)(0));
                bf[j] = __builtin_bit_cast(hx8, v);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i][j] = TILE_MFMA(af[i], bf[j], acc[i][j]);
        }
        st_c = st_c + 1 == NS ? 0 : st_c + 1;
    }
    GSTAMP(2)
    GSTAMP(3)

    // ---- the waves of a tap meet in LDS, in a fixed order; the sums are laid out as the rows of dW ([co][ci][tap],
    // 48 x 48 x Keff floats) behind the exchange area
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                     // every consumer has read its last operands: LDS is free
    asm volatile("" ::: "memory");
    f32x4* const red = (f32x4*)smem;                                  // [wave][9][64] x 16 B
    float* const rows = (float*)(smem + ncw * 9 * 64 * 16);           // [48][48 * Keff]
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) red[(wave * 9 + i * 3 + j) * 64 + lane] = acc[i][j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int u = wave; u < 9 * ntaps; u += ncw) {
        const int tpu = (u * 57) >> 9, tau = u - 9 * tpu;              // u / 9, u % 9 (u < 45)
        f32x4 v = red[((tpu * nkg) * 9 + tau) * 64 + lane];
        for (int g = 1; g < nkg; ++g) {
            const f32x4 t = red[((tpu * nkg + g) * 9 + tau) * 64 + lane];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += t[e];
        }
        const int i = (tau * 11) >> 5, j = tau - 3 * i;               // tau / 3, tau % 3
        // D[row = co 16 i + 4 kg + e][col = ci 16 j + l15]
#pragma unroll
        for (int e = 0; e < 4; ++e) rows[(16 * i + 4 * kg + e) * rowf + (16 * j + l15) * ntaps + tpu] = v[e];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    GSTAMP(4)
    // ---- rows of the tile -> dW (+=) or the slice's slab (=)
    if (vec) {
#pragma unroll
        for (int u = 0; u < G_PRE; ++u) {
            const int e = tid + u * nthr;
            const int r = (int)fd_div((uint32_t)e, k.fdRow4), c4 = e - r * v4row;
            if (e < total4) {
                f32x4 v = *(const f32x4*)(rows + r * rowf + c4 * 4);
                if (acc_into) { v[0] += old[u][0]; v[1] += old[u][1]; v[2] += old[u][2]; v[3] += old[u][3]; }
                *(f32x4*)(dst + ((size_t)(m0 + r) * k.Cin + n0) * dK + c4 * 4) = v;
            }
        }
    } else {
        const int per = ci_n * Keff, total = co_n * per;
        for (int e = tid; e < total; e += nthr) {
            const int r = e / per, c = e - r * per;
            const int ci = c / Keff, tpe = c - ci * Keff;
            const float v = rows[r * rowf + ci * Keff + tpe];
            float* p = dst + ((size_t)(m0 + r) * k.Cin + n0 + ci) * dK + dk0 + tpe;
            if (acc_into) *p += v; else *p = v;
        }
    }
#ifdef STEGAN_STAMPS
    if (k.stamps && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        k.stamps[(size_t)blockIdx.x * 16 + 5] = __builtin_amdgcn_s_memtime();
        k.stamps[(size_t)blockIdx.x * 16 + 9] = __builtin_amdgcn_s_memrealtime();
    }
#endif
#undef GSTAMP
#undef GPSTAMP
}

__global__ __launch_bounds__(1024) void conv_tile_wgrad_kernel(const GK k) { conv_tile_wgrad_body(k, blockIdx.x); }

// The weight gradients of up to GM_MAX layers in ONE grid (the five convs of a generator block: VERDICT r04 item 3).  Layer by
// layer a launch is 256 blocks that all pay their fixed phases at the same time and a tail in which the chip drains; in one
// grid the blocks of different layers follow each other on a CU, so set-up, first fetch and epilogue of one block overlap the
// K loops of its neighbours' blocks on other CUs, and the batch can be cut into 2 slices instead of 4 (half the slabs).
constexpr int GM_MAX = 5;
struct GKMulti { int n; int first[GM_MAX + 1]; GK p[GM_MAX]; };
static_assert(sizeof(GKMulti) <= 4096, "kernel argument segment limit");

__global__ __launch_bounds__(1024) void conv_tile_wgrad_multi_kernel(const GKMulti m) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < GM_MAX; ++i)
        if (i < m.n && (int)blockIdx.x >= m.first[i]) pi = i;
    conv_tile_wgrad_body(m.p[pi], blockIdx.x - (unsigned)m.first[pi]);
}

struct GOpts { int enable, debug, ns, target, nkg; };
const GOpts& g_opts() {
    static const GOpts o = [] {
        GOpts t;
        const char* e;
        t.enable = (e = getenv("STEGAN_TILE_WGRAD")) ? atoi(e) : 1;
        t.debug = getenv("STEGAN_PLAN_DEBUG") != nullptr;
        t.ns = (e = getenv("STEGAN_TILE_WGRAD_NS")) ? atoi(e) : 0;
        t.target = (e = getenv("STEGAN_TILE_WGRAD_TARGET")) ? atoi(e) : 256;
        t.nkg = (e = getenv("STEGAN_TILE_WGRAD_NKG")) ? atoi(e) : 0;
        return t;
    }();
    return o;
}

inline int g_round_up(int a, int b) { return (a + b - 1) / b * b; }

bool wgrad_plan(const TileWgradArgs& a, GK& k, size_t& lds, int target = 0) {
    const GOpts& o = g_opts();
    if (target <= 0) target = o.target;
    if (!o.enable || !a.xh || !a.dyh || !a.dw) return false;
    if (a.Cin % 8 || a.Cout % 8 || a.Cin < 32 || a.Cout < 32) return false;
    if (a.Keff < 1 || a.Keff > KT_MAX) return false;
    if (a.Tout % 4 != 0 || a.Tout > 192 || a.Tout != (a.Tin << a.ups)) return false;
    const int64_t xbytes = (int64_t)a.B * (a.Cin / 8) * a.Tin * 16, ybytes = (int64_t)a.B * (a.Cout / 8) * a.Tout * 16;
    if (xbytes >= (int64_t)G_OOB || ybytes >= (int64_t)G_OOB) return false;
    if ((int64_t)a.Cout * a.Cin * a.K * 4 >= (int64_t)G_OOB) return false;
    const int T = a.Tout;
    int CS = 192 / T;                                     // whole samples per chunk, <= 192 dense columns
    if (CS < 1) CS = 1;
    if (CS > a.B) CS = a.B;
    const int cols = CS * T, KSC = (cols + 31) / 32;
    const int MT = (a.Cout + GT - 1) / GT, NTl = (a.Cin + GT - 1) / GT, tiles = MT * NTl;
    if ((double)MT * GT / a.Cout * ((double)NTl * GT / a.Cin) > 1.5) return false;
    // batch slices: about one block per CU, whole chunks per slice, a workspace that holds the slabs
    const int chunks_all = a.B / CS;
    if (a.B % CS) return false;
    int slices = (target + tiles - 1) / tiles;
    if (slices > chunks_all) slices = chunks_all;
    const int64_t slab = (int64_t)a.Cout * a.Cin * a.Keff;
    if (slices > 1 && (!a.partial || a.partial_floats < slab)) return false;
    if (slices > 1 && slices > a.partial_floats / slab) slices = (int)(a.partial_floats / slab);
    if (slices < 1) slices = 1;
    while (chunks_all % slices) --slices;                 // equal slices (B, CS powers of two in practice)
    const int nch = chunks_all / slices, bps = nch * CS;
    const int span = (a.Keff - 1) * a.dil;
    const int SEGW = T + span;
    // (a partial last k-step reads rows past the chunk's columns: dy is zero there, and x must be -- never stale LDS, which may
    // hold NaN patterns -- so the x plane covers every segment a k-step row can fall into; slots past CS * SEGW land as zeros)
    const int XSLOT = g_round_up(((KSC * 32 + T - 1) / T) * SEGW, 64), DSLOT = g_round_up(KSC * 32, 64);
    const int xseg = XSLOT / 64, yseg = DSLOT / 64;
    const int nX = 6 * xseg, nY = 6 * yseg;
    if ((nX + 3) / 4 > GX_MAX || (nY + 3) / 4 > GY_MAX) return false;
    const int ystage = nY * 1024, stage = ystage + nX * 1024;
    // consumer waves: one tap each, nkg k-step groups per tap, at most 12 (+ 4 producers = 1024 threads)
    int nkg = 12 / a.Keff;
    if (o.nkg > 0 && nkg > o.nkg) nkg = o.nkg;
    if (nkg > KSC) nkg = KSC;
    const int ncons = a.Keff * nkg;
    if ((size_t)GT * (GT * a.Keff / 4) > (size_t)G_PRE * ncons * 64) return false;
    const size_t epi = (size_t)ncons * 9 * 64 * 16 + (size_t)GT * GT * a.Keff * 4;
    int NS = o.ns ? o.ns : 3;
    if (NS > nch) NS = nch;
    if (NS < 2) NS = 2;
    const int ipc = (nX + 3) / 4 + (nY + 3) / 4;
    while (NS > 2 && ((size_t)NS * stage > (size_t)G_LDS_MAX || (NS - 1) * ipc > 40)) --NS;
    if ((size_t)NS * stage > (size_t)G_LDS_MAX || epi > (size_t)G_LDS_MAX || (NS - 1) * ipc > 40) return false;
    lds = (size_t)NS * stage > epi ? (size_t)NS * stage : epi;
    k.xh = a.xh; k.dyh = a.dyh; k.dy = a.dy; k.dw = a.dw; k.db = a.db; k.partial = a.partial;
    k.prev_partial = a.prev_partial; k.prev_dw = a.prev_dw; k.prev_n4 = (long)a.prev_n4; k.prev_slices = a.prev_slices;
    k.xh_bytes = (unsigned)xbytes; k.dyh_bytes = (unsigned)ybytes;
    k.B = a.B; k.Cin = a.Cin; k.Cout = a.Cout; k.K = a.K; k.k_lo = a.k_lo; k.Keff = a.Keff; k.Tin = a.Tin; k.Tout = a.Tout;
    k.ups = a.ups; k.relu = a.relu;
    k.CBx = a.Cin / 8; k.CBy = a.Cout / 8;
    k.off_min = a.k_lo * a.dil - a.pad; k.step = a.dil;
    k.ncons = ncons; k.nthreads = 64 * (ncons + 4); k.nkg = nkg;
    k.CS = CS; k.cols = cols; k.KSC = KSC; k.SEGW = SEGW; k.XSLOT = XSLOT; k.DSLOT = DSLOT;
    k.nX = nX; k.nY = nY; k.xseg = xseg; k.yseg = yseg; k.NS = NS; k.ystage = ystage; k.stage = stage;
    k.slices = slices; k.bps = bps; k.nch = nch; k.MT = MT; k.NTl = NTl; k.tiles = tiles;
    k.x_samp = (a.Cin / 8) * a.Tin * 16; k.y_samp = (a.Cout / 8) * a.Tout * 16;
    k.bias_vec = (a.Tout % 4 == 0) && ((uintptr_t)a.dy % 16 == 0);
    // the bias gradient: inside the tile blocks when the launch is one slice; in the caller's reduce launch (small blocks) when it
    // promises that; else blocks behind the tiles
    k.bias_mode = 0;
    if (a.db) {
        const int cpb = (GT + NTl - 1) / NTl;
        if (slices == 1 && cpb <= ncons && k.bias_vec && (int64_t)a.B * T <= 8192) k.bias_mode = 1;
        else if (slices > 1 && a.bias_in_reduce) k.bias_mode = 0;
        else if (slices > 1 && cpb <= 2 * ncons && k.bias_vec && (int64_t)a.B * T <= 16384) k.bias_mode = 1;   // slice 0's blocks
        else k.bias_mode = 2;
    }
    k.bias_first = k.bias_mode == 2 ? tiles * slices : 0x7fffffff;
    k.fdT = make_fastdiv((uint32_t)T); k.fdSEGW = make_fastdiv((uint32_t)SEGW); k.fdXseg = make_fastdiv((uint32_t)xseg);
    k.fdYseg = make_fastdiv((uint32_t)yseg); k.fdMT = make_fastdiv((uint32_t)MT); k.fdTiles = make_fastdiv((uint32_t)tiles);
    k.fdTb = make_fastdiv((uint32_t)(k.bias_vec ? a.Tout / 4 : a.Tout));
    k.fdNkg = make_fastdiv((uint32_t)nkg); k.fdRow4 = make_fastdiv((uint32_t)(GT * a.Keff / 4));
    return true;
}

}  // namespace

#ifdef STEGAN_STAMPS
static unsigned long long* g_wgrad_stamps = nullptr;
extern "C" int stegan_tile_wgrad_debug_stamps(void* p) { g_wgrad_stamps = (unsigned long long*)p; return 0; }
#endif

// n layers in one launch; 1: not every layer is a shape for the tile kernel (nothing enqueued: the caller goes layer by layer)
int stegan_tile_wgrad_multi(int n, const TileWgradArgs* a, int* slices_out, int* bias_done, hipStream_t stream) {
    if (n < 1 || n > GM_MAX) return 1;
    static const int mtarget = [] { const char* e = getenv("STEGAN_TILE_WGRAD_MULTI_TARGET"); return e ? atoi(e) : 128; }();
    GKMulti m;
    m.n = n; m.first[0] = 0;
    size_t lds = 0;
    int nthreads = 0;
    for (int i = 0; i < n; ++i) {
        size_t l = 0;
        if (!wgrad_plan(a[i], m.p[i], l, mtarget)) return 1;
        if (m.p[i].bias_mode == 2) return 1;          // (blocks behind the tiles: single launches only)
        if (l > lds) lds = l;
        if (m.p[i].nthreads > nthreads) nthreads = m.p[i].nthreads;
#ifdef STEGAN_STAMPS
        m.p[i].stamps = nullptr;
#endif
        m.first[i + 1] = m.first[i] + m.p[i].tiles * m.p[i].slices;
    }
    for (int i = n; i < GM_MAX; ++i) m.first[i + 1] = m.first[n];
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)conv_tile_wgrad_multi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_MAX);
        (void)hipGetLastError();
        attr = true;
    }
    if (g_opts().debug)
        for (int i = 0; i < n; ++i)
            fprintf(stderr, "plan-tile-wgrad-multi[%d/%d] Cin=%d Cout=%d K=%d T=%d | tiles=%d slices=%d nch=%d ncons=%d bias_mode=%d\n", i, n,
                    a[i].Cin, a[i].Cout, a[i].Keff, a[i].Tout, m.p[i].tiles, m.p[i].slices, m.p[i].nch, m.p[i].ncons, m.p[i].bias_mode);
    STEGAN_TIMED_LAUNCH(conv_tile_wgrad_multi_kernel, dim3((unsigned)m.first[n]), dim3((unsigned)nthreads), lds, stream, m);
    { const hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return -(1000 + (int)e__); }
    for (int i = 0; i < n; ++i) {
        if (slices_out) slices_out[i] = m.p[i].slices;
        if (bias_done) bias_done[i] = (!a[i].db || m.p[i].bias_mode != 0) ? 1 : 0;
    }
    return STEGAN_OK;
}

int stegan_tile_wgrad(const TileWgradArgs& a, int* slices_out, int* bias_done, hipStream_t stream) {
    GK k;
    size_t lds = 0;
    if (!wgrad_plan(a, k, lds)) return 1;
    static bool attr = false;
    if (!attr) {
        // (the bias-gradient body keeps 16 bytes of static LDS: the dynamic part must leave room for them)
        (void)hipFuncSetAttribute((const void*)conv_tile_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_MAX);
        (void)hipGetLastError();
        attr = true;
    }
    if (g_opts().debug)
        fprintf(stderr, "plan-tile-wgrad B=%d Cin=%d Cout=%d K=%d(live %d) d=%d T=%d ups=%d | tiles=%dx%d slices=%d nch=%d CS=%d KSC=%d ncons=%d "
                        "NS=%d lds=%zu nX=%d nY=%d XSLOT=%d\n", a.B, a.Cin, a.Cout, a.K, a.Keff, a.dil, a.Tout, a.ups, k.MT, k.NTl, k.slices,
                k.nch, k.CS, k.KSC, k.ncons, k.NS, lds, k.nX, k.nY, k.XSLOT);
#ifdef STEGAN_STAMPS
    k.stamps = g_wgrad_stamps;
#endif
    const unsigned grid = (unsigned)(k.tiles * k.slices + (k.bias_mode == 2 ? a.Cout : 0));
    if (bias_done) *bias_done = (!a.db || k.bias_mode != 0) ? 1 : 0;
    STEGAN_TIMED_LAUNCH(conv_tile_wgrad_kernel, dim3(grid), dim3((unsigned)k.nthreads), lds, stream, k);
    { const hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return -(1000 + (int)e__); }   // (never 1: that means "declined")
    if (slices_out) *slices_out = k.slices;
    return STEGAN_OK;
}
