// Shared between conv_direct.hip (fp32 operands) and conv_bf16.hip (bf16 operands): launch descriptor, tap progressions,
// fused epilogue and the split-K / tail-split epilogue kernels.  Everything lives in an anonymous namespace: each
// translation unit gets its own copy.
#pragma once
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace {


constexpr int MAX_TAPS = 48;
constexpr int MAX_PHASES = 8;    // BWD_DATA output phases = stride (larger strides take the generic kernel)
constexpr int XE = 16;           // X-tile elements staged per thread and chunk: CC * XROW <= 256 * XE
constexpr int WE = 4;            // W-tile float4 per thread and chunk: RC * 16 <= 256 * WE
constexpr int BM2 = 64;

struct DirP {
    const float* src;     // (B, Csrc, Tsrc)
    const float* wt;      // transposed weights: [G][K][C2_g][LD]
    const float* scale;   // FWD: epilogue scale[co];  BWD: nullable pointer to ONE uniform factor (per-channel factors
                          // are folded into wb by stegan_conv_prep)
    const float* bias;    // FWD
    const float* res;     // FWD residual / BWD add_src
    const float* mask;    // BWD relu mask source (B, Cdst, Tmask); with mask_lrelu: the leaky-ReLU OUTPUT the gradient passes
    float* out;
    int B, Csrc, Cdst, Tsrc, Tq;        // Tq = valid source extent in q units (FWD: Tin*up, BWD: Tout)
    int Tdst;                           // stored output length per (b, channel)  (FWD: Tout, BWD: Te)
    int Tmask, ups;                     // BWD: mask length (Tin) and log2(upsample) for mask index; FWD: ups for src index
    int C2_g, M_g, LD;                  // reduction channels / output channels per group, padded leading dim of wt
    long zero_off;                      // float offset (from wt) of >= 64 zero floats behind the transposed weights
    int K, stride, dil, pad, k_lo, k_hi;
    int mode;                           // 0 FWD, 1 BWD_DATA
    int in_relu, out_act, out_btc;
    int m_tiles, CC, logCC;
    int SEG, NSEG, XROW;
    int WR;                             // W-tile rows held in LDS per buffer (>= taps * CC of any phase, multiple of 8)
    int TG, NTG;                        // tap groups: a W chunk holds TG taps x CC channels, NTG chunks per channel chunk
                                        // (NTG = 1: every live tap in one chunk); the X tile is staged once per channel chunk
    FastDiv fdNTG;
    int U0;                             // columns per batch element for phase 0 (max over phases)
    int splits, chunks_per_split;       // split-K over channel chunks: grid.z = phase*splits + split
    // Tail split (mix_S > 0, 1-D grid): the first mix_full tiles run whole; the remaining tiles -- the ones that would
    // make a quarter of the CUs carry one block more than the rest -- are cut into mix_S reduction slices each, so every
    // CU gets the same number of whole tiles plus one short block.  Tile id = (phase * ny + y) * nbx + x.
    int mix_S, mix_full, mix_cps, nbx, ny;
    int tile_m, tile_n;                 // block tile (rows, columns) of the launching kernel: what a tail tile covers
    float* partial;                     // splits > 1: raw accumulators [split][B*Cdst*Tdst], epilogue runs separately
    unsigned long long* stamps;         // diagnostic builds only (STEGAN_STAMPS): 8 s_memtime stamps per block
    FastDiv fdS;
    FastDiv fdXROW, fdSEG, fdU0, fdU1;  // magic division by XROW, SEG, U0 and U0-1 (the two per-phase column counts)
    FastDiv fdMT, fdSP, fdNBX, fdNY, fdMS;   // ... by m_tiles, splits, nbx, ny, mix_S (block-id decode: no software division
                                             // chains, ~35 scalar instructions each, in front of the first DMA)
    int ph_ntaps[MAX_PHASES], ph_kbase[MAX_PHASES], ph_kstep[MAX_PHASES], ph_offmin[MAX_PHASES], ph_step[MAX_PHASES];
    int ph_U[MAX_PHASES];               // per output phase: tap progression and column count, precomputed on the host
    // round 3: bf16 activation image of the output, [B][out_cb][Tdst][8 channels] (channel-block-major, the layout the bf16
    // kernels stage by LDS-DMA); null = not wanted.  Written beside the fp32 output by the fused epilogues.
    unsigned short* out_h = nullptr;
    int out_cb = 0;
    // round 4: BWD with in_act == STEGAN_ACT_LRELU -- the conv's input is the leaky-ReLU(0.1) output of the layer before it, and
    // this launch also applies THAT layer's activation derivative: dx = (acc * scale + res) * (mask > 0 ? 1 : 0.1), `res`
    // being the feature-matching gradient of the same tensor.  (The separate act_bwd launch per layer goes away.)
    int mask_lrelu = 0;
};

__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    return __builtin_bit_cast(unsigned short, (hop_t)hop_src(f));
}

__device__ __forceinline__ void epilogue_store(const DirP& p, float v, int b, int ch, int to) {
    if (p.mode == 0) {
        if (p.scale) v *= p.scale[ch];
        if (p.bias) v += p.bias[ch];
        const size_t idx = ((size_t)b * p.Cdst + ch) * p.Tdst + to;
        if (p.res) v += p.res[idx];
        if (p.out_act == STEGAN_ACT_LRELU) v = v > 0.f ? v : 0.1f * v;
        else if (p.out_act == STEGAN_ACT_RELU) v = fmaxf(v, 0.f);
        else if (p.out_act == STEGAN_ACT_TANH) v = tanhf(v);
        if (p.out_btc) p.out[((size_t)b * p.Tdst + to) * p.Cdst + ch] = v;
        else p.out[idx] = v;
        if (p.out_h) p.out_h[(((size_t)b * p.out_cb + (ch >> 3)) * p.Tdst + to) * 8 + (ch & 7)] = f32_to_bf16_bits(v);
    } else {
        float lr = 1.f;
        if (p.mask) {
            const float xv = p.mask[((size_t)b * p.Cdst + ch) * p.Tmask + (to >> p.ups)];
            if (p.mask_lrelu) lr = xv > 0.f ? 1.f : 0.1f;
            else if (!(xv > 0.f)) v = 0.f;
        }
        if (p.scale) v *= p.scale[0];
        const size_t idx = ((size_t)b * p.Cdst + ch) * p.Tdst + to;
        if (p.res) v += p.res[idx];
        v *= lr;
        p.out[idx] = v;
        if (p.out_h) p.out_h[(((size_t)b * p.out_cb + (ch >> 3)) * p.Tdst + to) * 8 + (ch & 7)] = f32_to_bf16_bits(v);
    }
}

// max(x, floor) as ONE VALU op (fmaxf also canonicalises both inputs): median(x, floor, +inf); floor = 0 (ReLU) or
// -inf (identity)
__device__ __forceinline__ float relu_max(float x, float floor_) {
    return __builtin_amdgcn_fmed3f(x, floor_, INFINITY);
}

// sums the split-K partials and applies the fused epilogue; HBM-bound: a thread takes 4 consecutive outputs (t fastest)
// with 16-byte loads from every slab, all slabs' loads issued before the first add
__global__ __launch_bounds__(256) void conv_splitk_epilogue_kernel(const DirP p) {
    const uint32_t total = (uint32_t)p.B * p.Cdst * p.Tdst;           // < 2^31, checked on the host
    if ((p.Tdst & 3) == 0) {
        const uint32_t total4 = total >> 2;
        for (uint32_t i4 = blockIdx.x * 256 + threadIdx.x; i4 < total4; i4 += gridDim.x * 256) {
            const float4* src = reinterpret_cast<const float4*>(p.partial) + i4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            int s = 0;
            for (; s + 4 <= p.splits; s += 4) {
                float4 t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) t[u] = src[(size_t)(s + u) * total4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { v.x += t[u].x; v.y += t[u].y; v.z += t[u].z; v.w += t[u].w; }
            }
            for (; s < p.splits; ++s) {
                const float4 t = src[(size_t)s * total4];
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            const uint32_t i = i4 << 2;
            const uint32_t r = i / (uint32_t)p.Tdst, to = i - r * p.Tdst;
            const uint32_t b = r / (uint32_t)p.Cdst, ch = r - b * p.Cdst;
            epilogue_store(p, v.x, (int)b, (int)ch, (int)to);
            epilogue_store(p, v.y, (int)b, (int)ch, (int)to + 1);
            epilogue_store(p, v.z, (int)b, (int)ch, (int)to + 2);
            epilogue_store(p, v.w, (int)b, (int)ch, (int)to + 3);
        }
        return;
    }
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        float v = 0.f;
        for (int s = 0; s < p.splits; ++s) v += p.partial[(size_t)s * total + i];
        const uint32_t r = i / (uint32_t)p.Tdst, to = i - r * p.Tdst;
        const uint32_t b = r / (uint32_t)p.Cdst, ch = r - b * p.Cdst;
        epilogue_store(p, v, (int)b, (int)ch, (int)to);
    }
}

// tail-split launches: grid (tail tiles * tile_n / 64, tile_m / 4); a block sums 4 rows x 64 columns of one split tile
// (one output per thread, all mix_S partial loads in flight at once, fixed summation order) and applies the fused epilogue
__global__ __launch_bounds__(256) void conv_tail_epilogue_kernel(const DirP p) {
    const int nsub = p.tile_n >> 6;
    const int tile = p.mix_full + (int)blockIdx.x / nsub, sub = (int)blockIdx.x % nsub;
    const int bx = tile % p.nbx, rest = tile / p.nbx;
    const int by = rest % p.ny, phase = rest / p.ny;
    const int g = by / p.m_tiles, m0 = (by % p.m_tiles) * p.tile_m;
    int U, os;
    if (p.mode == 0) { U = p.U0; os = 1; }
    else { U = (p.Tdst - phase + p.stride - 1) / p.stride; os = p.stride; }
    const int N = p.B * U;
    const int n = bx * p.tile_n + sub * 64 + (threadIdx.x & 63);
    const int m = m0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (n >= N || m >= p.M_g) return;
    const int b = n / U, u = n - b * U;
    const int to = p.mode == 0 ? u : phase + u * os;
    const size_t total = (size_t)p.B * p.Cdst * p.Tdst;
    const int ch = g * p.M_g + m;
    const float* src = p.partial + ((size_t)b * p.Cdst + ch) * p.Tdst + to;
    float t[8];
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) t[s2] = s2 < p.mix_S ? src[(size_t)s2 * total] : 0.f;
    float v = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) v += t[s2];
    epilogue_store(p, v, b, ch, to);
}

// Per-phase tap progression.  The live taps of a launch (FWD) or of one output phase (BWD) form an arithmetic
// progression: tap j uses weight index k = kbase + j*kstep and reads the source at offset off_min + j*step, so no
// lookup tables are needed in the inner loop.
struct Taps { int ntaps, kbase, kstep, off_min, step; };

inline Taps phase_taps_host(const DirP& p, int phase) {
    Taps t;
    if (p.mode == 0) {
        t.ntaps = p.k_hi - p.k_lo + 1; t.kbase = p.k_lo; t.kstep = 1;
        t.off_min = p.k_lo * p.dil - p.pad; t.step = p.dil;
        return t;
    }
    // BWD: taps k (descending) with (phase + pad - k*dil) % stride == 0; offsets (phase+pad-k*dil)/stride ascend
    int gg = p.dil % p.stride, hh = p.stride;            // gcd(dil, stride)
    while (gg) { const int tt = hh % gg; hh = gg; gg = tt; }
    const int dk = p.stride / hh;
    t.ntaps = 0; t.kbase = 0; t.kstep = -dk; t.off_min = 0; t.step = p.dil / hh;
    for (int k = p.k_hi; k > p.k_hi - dk && k >= p.k_lo; --k) {
        const int a = phase + p.pad - k * p.dil;
        int q = a / p.stride, r = a - q * p.stride;
        if (r < 0) { r += p.stride; q -= 1; }
        if (r == 0) {
            t.kbase = k; t.off_min = q; t.ntaps = (k - p.k_lo) / dk + 1;
            break;
        }
    }
    return t;
}

__device__ __forceinline__ Taps phase_taps(const DirP& p, int phase) {
    Taps t;
    t.ntaps = p.ph_ntaps[phase]; t.kbase = p.ph_kbase[phase]; t.kstep = p.ph_kstep[phase];
    t.off_min = p.ph_offmin[phase]; t.step = p.ph_step[phase];
    return t;
}

inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

int live_taps(const stegan_conv_desc* d, int& k_lo, int& k_hi) {
    const int Te = d->Tin * d->upsample;
    k_lo = d->K; k_hi = -1;
    for (int k = 0; k < d->K; ++k) {
        const int lo = k * d->dilation - d->pad, hi = (d->Tout - 1) * d->stride + lo;
        if (hi >= 0 && lo < Te) { if (k < k_lo) k_lo = k; k_hi = k; }
    }
    return k_hi - k_lo + 1;
}

int check_desc(const stegan_conv_desc* d) {
    STEGAN_CHECK_ARG(d != nullptr);
    STEGAN_CHECK_ARG(d->B > 0 && d->Cin > 0 && d->Cout > 0 && d->groups > 0 && d->K > 0);
    STEGAN_CHECK_ARG(d->Cin % d->groups == 0 && d->Cout % d->groups == 0);
    STEGAN_CHECK_ARG(d->stride >= 1 && d->dilation >= 1 && d->pad >= 0 && d->Tin > 0);
    STEGAN_CHECK_ARG(d->upsample == 1 || d->upsample == 2);
    const int Te = d->Tin * d->upsample;
    const int span = Te + 2 * d->pad - d->dilation * (d->K - 1) - 1;
    STEGAN_CHECK_ARG(span >= 0 && d->Tout == span / d->stride + 1);
    STEGAN_CHECK_ARG((int64_t)d->B * (d->Cin > d->Cout ? d->Cin : d->Cout) * (Te > d->Tout ? Te : d->Tout) < (1ll << 31));
    return STEGAN_OK;
}


// split-K factor from a small cost model.  All blocks of a launch are co-resident (<= ~3 per CU) and share their CU's
// matrix pipes, so time ~ (max blocks on one CU) x (chunks per block + fixed per-block overhead, ~2.5 chunk
// equivalents for prologue + first DMA latency + epilogue) + the extra pass that sums the partial tiles.
inline int choose_splits(long base_blocks, int nchunks, double chunk_cost = 1.0, double* cost_out = nullptr) {
    static const char* env = getenv("STEGAN_SPLITK");
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= 16 && s <= nchunks; ++s) {
        const int per = (nchunks + s - 1) / s;
        if (s > 1 && per < 2) break;
        const long blocks = base_blocks * ((nchunks + per - 1) / per);
        const double rounds = (double)((blocks + 255) / 256);
        const double cost = rounds * (per * chunk_cost + 2.5) + (s > 1 ? 1.5 : 0.0);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
    }
    if (env) { int f = atoi(env); best = f < 1 ? 1 : (f > nchunks ? nchunks : f); }
    if (cost_out) *cost_out = best_cost;
    return best;
}

}  // namespace
