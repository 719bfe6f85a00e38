// "Skinny" convolutions: the single-output-channel convs that end every sub-discriminator (discriminator.py:31,60,81,
// 113: 512 -> 1 / 1024 -> 1, k = 3) in all three directions, on the vector ALUs in exact fp32.
//
// A 1-row GEMM has nothing for a matrix core: in the 64-row MFMA tiles of conv_direct.hip / conv_igemm.hip 63 rows are
// padding, and those kernels' per-block chains (stage a chunk, barrier, multiply, next chunk) made each of these tiny
// layers cost 15-30 us -- 24 forward, 16 input-gradient and 8 weight-gradient launches a step.  Each direction is a
// streaming pass over ONE tensor (x, dx, x again), i.e. HBM-bound at a few microseconds:
//
//   forward   y[b, t]       = act(scale * sum_{ci, k} w[ci, k] * x[b, ci, t s + k d - p] + bias)
//   d/dx      dx[b, ci, q]  = scale * sum_k w[ci, k] * dy[b, (q + p - k d) / s]          (terms with s | (q + p - k d))
//   d/dw      dw[ci, k]    += sum_{b, t} dy[b, t] * x[b, ci, t s + k d - p]               (raw direction gradient)
//
// Every entry point returns 1 for a shape it does not take (caller falls through to the general kernels).
#include "common.h"

namespace {

constexpr int SK_KMAX = 8;

struct SkP {
    const float* x;       // (B, Cin, Tin)
    const float* w;       // (1, Cin, K)
    const float* dy;      // (B, 1, Tout)
    const float* scale;   // 1 value or null
    const float* bias;    // 1 value or null
    float* out;           // y (B, 1, Tout) | dx (B, Cin, Tin) | dw (1, Cin, K)
    int B, Cin, K, stride, dil, pad, Tin, Tout, out_act;
    int TT, CL;           // forward: time-tile width, channel lanes per block (TT * CL <= 256)
    int per;              // weight gradient: batch rows per block
    int nqt;              // weight gradient: time tiles per batch slice (slab index = slice * nqt + time tile)
    float* part;          // weight gradient: slabs [slices * nqt][Cin * K], then [slices] bias sums
    // input gradient (round 4): x is the leaky-ReLU output of the layer before -- dx = (conv_T(dy) * scale + add) * lrelu'(mask)
    // applies that layer's activation derivative here (`add`: the feature-matching gradient of x, may be null), and the
    // result also goes out as a bf16 activation image [B][Cin / 8][Tin][8] for the layer's own backward kernels
    const float* mask;
    const float* add;
    unsigned short* out_h;
    FastDiv fdTT, fdS, fdTin, fdCT, fdCL;
};

__device__ __forceinline__ float sk_act(float v, int act) {
    if (act == STEGAN_ACT_LRELU) return v > 0.f ? v : 0.1f * v;
    if (act == STEGAN_ACT_RELU) return fmaxf(v, 0.f);
    if (act == STEGAN_ACT_TANH) return tanhf(v);
    return v;
}

// forward: block = (batch row, tile of TT <= 8 output steps); thread = (output step, channel lane); a channel lane walks
// channels lane, lane + CL, ...  The (ci, t) planes are dense, so the TT lanes of a channel lane read one contiguous run
// and neighbouring channel lanes the next rows.  The kernel is bound by load round trips, not bytes: all 2 K loads of
// FOUR channels are issued before their first use, and short tiles give many channel lanes (few trips each) and many
// blocks -- the first version (64-step tiles, one load per multiply-add) took 35 us for 1024 channels x 48 steps.
// Partial sums meet in LDS, in fixed order.
__device__ __forceinline__ void conv_cout1_fwd_body(const SkP& p, int bx, int by) {
    __shared__ float red[256];
    const int b = bx, tid = threadIdx.x;
    const int cl = (int)fd_div((uint32_t)tid, p.fdTT), tl = tid - cl * p.TT;
    const int t = by * p.TT + tl;
    const bool live = cl < p.CL && t < p.Tout;
    int qk[SK_KMAX];
    bool ok[SK_KMAX];
#pragma unroll
    for (int k = 0; k < SK_KMAX; ++k) {
        const int q = t * p.stride + k * p.dil - p.pad;
        ok[k] = live && k < p.K && q >= 0 && q < p.Tin;
        qk[k] = ok[k] ? q : 0;
    }
    const float* xb = p.x + (size_t)b * p.Cin * p.Tin;
    constexpr int NCH = 4;                                  // channels in flight per thread: NCH x 2 K independent loads
    float acc[NCH];
#pragma unroll
    for (int u = 0; u < NCH; ++u) acc[u] = 0.f;
    if (live) {
        for (int c = cl; c < p.Cin; c += NCH * p.CL) {
            float xv[NCH][SK_KMAX], wv[NCH][SK_KMAX];
            bool has[NCH];
#pragma unroll
            for (int u = 0; u < NCH; ++u) {
                has[u] = c + u * p.CL < p.Cin;
                const int cc = has[u] ? c + u * p.CL : c;
#pragma unroll
                for (int k = 0; k < SK_KMAX; ++k) {
                    if (k < p.K) {                          // uniform
                        xv[u][k] = xb[(size_t)cc * p.Tin + qk[k]];
                        wv[u][k] = p.w[cc * p.K + k];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < NCH; ++u)
#pragma unroll
                for (int k = 0; k < SK_KMAX; ++k)
                    if (k < p.K) acc[u] += (ok[k] && has[u]) ? xv[u][k] * wv[u][k] : 0.f;
        }
    }
    red[tid] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    block_sync();
    if (cl == 0 && t < p.Tout) {
        float s = 0.f;
        for (int j = 0; j < p.CL; ++j) s += red[j * p.TT + tl];       // fixed order: deterministic
        if (p.scale) s *= p.scale[0];
        if (p.bias) s += p.bias[0];
        p.out[(size_t)b * p.Tout + t] = sk_act(s, p.out_act);
    }
}

__global__ __launch_bounds__(256) void conv_cout1_fwd_kernel(const SkP p) {
    conv_cout1_fwd_body(p, blockIdx.x, blockIdx.y);
}

// Up to SK_MULTI problems in one launch (the output convs of the five period discriminators: same layer, own batch
// B * period and length each): blockIdx.z picks the problem, blocks outside its extent leave at once.
constexpr int SK_MULTI = 5;
struct SkMulti { int n; SkP p[SK_MULTI]; long ne[SK_MULTI]; };

__global__ __launch_bounds__(256) void conv_cout1_fwd_multi_kernel(const SkMulti a) {
    const SkP& p = a.p[blockIdx.z];
    if ((int)blockIdx.x >= p.B || (int)blockIdx.y * p.TT >= p.Tout) return;
    conv_cout1_fwd_body(p, blockIdx.x, blockIdx.y);
}

// input gradient: one thread per dx element, consecutive threads consecutive addresses; dy (B x Tout floats) and w stay
// in cache.
__device__ __forceinline__ void conv_cout1_bwd_data_body(const SkP& p, long n, int bx, int nbx) {
    const float sc = p.scale ? p.scale[0] : 1.f;
    for (long e = (long)bx * 256 + threadIdx.x; e < n; e += (long)nbx * 256) {
        const uint32_t b = fd_div((uint32_t)e, p.fdCT);
        const uint32_t r = (uint32_t)e - b * (uint32_t)(p.Cin * p.Tin);
        const uint32_t ci = fd_div(r, p.fdTin);
        const int q = (int)(r - ci * p.Tin);
        const float* dyb = p.dy + (size_t)b * p.Tout;
        const float* wc = p.w + (size_t)ci * p.K;
        float s = 0.f;
        for (int k = 0; k < p.K; ++k) {
            const int a = q + p.pad - k * p.dil;
            if (a < 0) continue;
            const int t = (int)fd_div((uint32_t)a, p.fdS);
            if (t * p.stride != a || t >= p.Tout) continue;
            s += wc[k] * dyb[t];
        }
        float v = s * sc;
        if (p.mask) {
            if (p.add) v += p.add[e];
            v *= p.mask[e] > 0.f ? 1.f : 0.1f;
        }
        p.out[e] = v;
        if (p.out_h) img_store(p.out_h, p.Cin >> 3, p.Tin, (int)b, (int)ci, q, v);
    }
}

__global__ __launch_bounds__(256) void conv_cout1_bwd_data_kernel(const SkP p, long n) {
    conv_cout1_bwd_data_body(p, n, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(256) void conv_cout1_bwd_data_multi_kernel(const SkMulti a) {
    conv_cout1_bwd_data_body(a.p[blockIdx.y], a.ne[blockIdx.y], blockIdx.x, gridDim.x);
}

// weight gradient: block = (CL channels, slice of `per` batch rows, tile of TT <= 64 input steps); thread = (input step
// q, channel): it walks the batch rows of its slice with x[b, c, q] -- consecutive lanes consecutive addresses, the CL
// rows of a block contiguous -- against the <= K dy values that pair with q, then the TT steps of a channel meet in LDS
// and one thread per (channel, tap) stores the block's sum into the block's slab; conv_cout1_wgrad_reduce_kernel adds the
// slabs into dw in a fixed order (rounds 1-3 met in dw through fp32 atomics, whose order changed from launch to launch).  (A lane-per-channel version read x with a row stride between lanes: 64 cache lines per
// load instruction, 32 us.)
__device__ __forceinline__ void conv_cout1_wgrad_body(const SkP& p, int bx, int by, int bz) {
    __shared__ float red[SK_KMAX][256];
    const int tid = threadIdx.x;
    const int cl = (int)fd_div((uint32_t)tid, p.fdTT), tl = tid - cl * p.TT;
    const int c = bx * p.CL + cl, q = bz * p.TT + tl;
    const bool live = cl < p.CL && c < p.Cin && q < p.Tin;
    const int b0 = by * p.per, b1 = min(p.B, b0 + p.per);
    int tk[SK_KMAX];
    bool ok[SK_KMAX];
#pragma unroll
    for (int k = 0; k < SK_KMAX; ++k) {
        const int a = q + p.pad - k * p.dil;
        const int t = (int)fd_div((uint32_t)(a < 0 ? 0 : a), p.fdS);
        ok[k] = live && k < p.K && a >= 0 && t * p.stride == a && t < p.Tout;
        tk[k] = ok[k] ? t : 0;
    }
    float acc[SK_KMAX];
#pragma unroll
    for (int k = 0; k < SK_KMAX; ++k) acc[k] = 0.f;
    if (live) {
        const float* xr = p.x + (size_t)c * p.Tin + q;
        const size_t xs = (size_t)p.Cin * p.Tin;
#pragma unroll 4
        for (int b = b0; b < b1; ++b) {
            const float xv = xr[b * xs];
            const float* dyb = p.dy + (size_t)b * p.Tout;
#pragma unroll
            for (int k = 0; k < SK_KMAX; ++k)
                if (k < p.K) acc[k] += ok[k] ? dyb[tk[k]] * xv : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < SK_KMAX; ++k) red[k][tid] = acc[k];
    block_sync();
    // thread (k, channel lane): sum of the channel's TT steps
    if (tid < p.CL * p.K) {
        const int k = (int)fd_div((uint32_t)tid, p.fdCL), cj = tid - k * p.CL;
        const int cc = bx * p.CL + cj;
        if (cc < p.Cin) {
            float s = 0.f;
            for (int j = 0; j < p.TT; ++j) s += red[k][cj * p.TT + j];
            p.part[(size_t)(by * p.nqt + bz) * p.Cin * p.K + (size_t)cc * p.K + k] = s;
        }
    }
}

// the weight gradients of up to SK_MULTI problems in one launch; blockIdx.x runs over the problems' channel blocks
// (first[]); the first channel block / first time tile of a problem also sums dy of its batch slice into the bias
// gradient (one output channel: db is ONE number per problem)
struct SkMultiW { int n; int first[SK_MULTI + 1]; SkP p[SK_MULTI]; float* db[SK_MULTI]; int nsl[SK_MULTI]; };

// dw[i] += sum over the slabs, db += sum over the slices' bias sums, both in slab order; blockIdx.y = problem
__global__ __launch_bounds__(256) void conv_cout1_wgrad_reduce_kernel(const SkMultiW a) {
    const SkP& p = a.p[blockIdx.y];
    const int nslab = a.nsl[blockIdx.y] * p.nqt, n = p.Cin * p.K;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float v = 0.f;
        for (int j = 0; j < nslab; ++j) v += p.part[(size_t)j * n + i];
        p.out[i] += v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.db[blockIdx.y]) {
        const float* bp = p.part + (size_t)nslab * n;
        float v = 0.f;
        for (int j = 0; j < a.nsl[blockIdx.y]; ++j) v += bp[j];
        *a.db[blockIdx.y] += v;
    }
}

__global__ __launch_bounds__(256) void conv_cout1_wgrad_multi_kernel(const SkMultiW a) {
    int k = 0;
#pragma unroll
    for (int j = 1; j < SK_MULTI; ++j)
        if (j < a.n && (int)blockIdx.x >= a.first[j]) k = j;
    const SkP& p = a.p[k];
    const int bx = blockIdx.x - a.first[k];
    if ((int)blockIdx.y * p.per >= p.B || (int)blockIdx.z * p.TT >= p.Tin) return;
    if (bx == 0 && blockIdx.z == 0 && a.db[k]) {
        __shared__ float bred[4];
        const int b0 = blockIdx.y * p.per, b1 = min(p.B, b0 + p.per);
        const int n = (b1 - b0) * p.Tout;
        const float* dyb = p.dy + (size_t)b0 * p.Tout;
        float s0 = 0.f, s1 = 0.f;
        int i = threadIdx.x;
        for (; i + 256 < n; i += 512) { s0 += dyb[i]; s1 += dyb[i + 256]; }
        if (i < n) s0 += dyb[i];
        float s = wave_sum(s0 + s1);
        if ((threadIdx.x & 63) == 0) bred[threadIdx.x >> 6] = s;
        block_sync();
        if (threadIdx.x == 0)
            p.part[(size_t)a.nsl[k] * p.nqt * p.Cin * p.K + blockIdx.y] = (bred[0] + bred[1]) + (bred[2] + bred[3]);
        block_sync();
    }
    conv_cout1_wgrad_body(p, bx, blockIdx.y, blockIdx.z);
}

// Cout == 1, ungrouped, no upsample / input activation / residual / (B, T, C) output: everything else returns 1
int fill(const stegan_conv_desc* d, SkP& p, bool lrelu_mask_ok = false) {
    if (!d) return STEGAN_E_ARG;
    if (d->Cout != 1 || d->groups != 1 || d->upsample != 1 || d->out_btc) return 1;
    if (d->in_act != STEGAN_ACT_NONE && !(lrelu_mask_ok && d->in_act == STEGAN_ACT_LRELU)) return 1;
    if (d->K < 1 || d->K > SK_KMAX || d->B <= 0 || d->Cin <= 0 || d->Tin <= 0 || d->Tout <= 0 || d->stride < 1 ||
        d->dilation < 1 || d->pad < 0)
        return 1;
    if ((int64_t)d->B * d->Cin * d->Tin >= (1ll << 31)) return 1;
    const int span = d->Tin + 2 * d->pad - d->dilation * (d->K - 1) - 1;
    if (span < 0 || d->Tout != span / d->stride + 1) return STEGAN_E_ARG;
    p.B = d->B; p.Cin = d->Cin; p.K = d->K; p.stride = d->stride; p.dil = d->dilation; p.pad = d->pad;
    p.Tin = d->Tin; p.Tout = d->Tout; p.out_act = d->out_act;
    p.fdS = make_fastdiv((uint32_t)d->stride);
    p.fdTin = make_fastdiv((uint32_t)d->Tin);
    p.fdCT = make_fastdiv((uint32_t)(d->Cin * d->Tin));
    p.TT = p.CL = p.per = p.nqt = 1;
    p.part = nullptr;
    p.mask = p.add = nullptr;
    p.out_h = nullptr;
    p.fdTT = make_fastdiv(1); p.fdCL = make_fastdiv(1);
    p.x = p.w = p.dy = p.scale = p.bias = nullptr;
    p.out = nullptr;
    return STEGAN_OK;
}

}  // namespace

extern "C" int stegan_conv1d_fwd_skinny(const stegan_conv_desc* d, const float* x, const float* w, const float* scale,
                                        const float* bias, const float* residual, float* y, void* stream) {
    SkP p;
    int rc = fill(d, p);
    if (rc) return rc;
    if (residual) return 1;
    STEGAN_CHECK_ARG(x && w && y);
    STEGAN_CHECK_ARG(d->out_act >= STEGAN_ACT_NONE && d->out_act <= STEGAN_ACT_TANH);
    p.x = x; p.w = w; p.scale = scale; p.bias = bias; p.out = y;
    p.TT = d->Tout < 8 ? d->Tout : 8;
    p.CL = 256 / p.TT;
    if (p.CL > d->Cin) p.CL = d->Cin;
    p.fdTT = make_fastdiv((uint32_t)p.TT);
    hipLaunchKernelGGL(conv_cout1_fwd_kernel, dim3((unsigned)d->B, (unsigned)ceil_div(d->Tout, p.TT)), dim3(256), 0,
                       (hipStream_t)stream, p);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

// The forward of up to 5 single-output-channel convs in one launch; returns 1 (nothing enqueued) unless every problem is
// one stegan_conv1d_fwd_skinny takes.
extern "C" int stegan_conv1d_fwd_skinny_multi(int32_t n, const stegan_conv_desc* const* descs, const float* const* x,
                                              const float* const* w, const float* const* scale, const float* const* bias,
                                              float* const* y, void* stream) {
    STEGAN_CHECK_ARG(n >= 1 && descs && x && w && y);
    if (n > SK_MULTI) return 1;
    SkMulti a;
    a.n = n;
    unsigned gx = 0, gy = 0;
    for (int i = 0; i < n; ++i) {
        SkP& p = a.p[i];
        int rc = fill(descs[i], p);
        if (rc) return rc;
        const stegan_conv_desc* d = descs[i];
        STEGAN_CHECK_ARG(x[i] && w[i] && y[i]);
        STEGAN_CHECK_ARG(d->out_act >= STEGAN_ACT_NONE && d->out_act <= STEGAN_ACT_TANH);
        p.x = x[i]; p.w = w[i]; p.scale = scale ? scale[i] : nullptr; p.bias = bias ? bias[i] : nullptr; p.out = y[i];
        p.TT = d->Tout < 8 ? d->Tout : 8;
        p.CL = 256 / p.TT;
        if (p.CL > d->Cin) p.CL = d->Cin;
        p.fdTT = make_fastdiv((uint32_t)p.TT);
        a.ne[i] = 0;
        if ((unsigned)d->B > gx) gx = (unsigned)d->B;
        if ((unsigned)ceil_div(d->Tout, p.TT) > gy) gy = (unsigned)ceil_div(d->Tout, p.TT);
    }
    if (gy > 65535u) return 1;
    hipLaunchKernelGGL(conv_cout1_fwd_multi_kernel, dim3(gx, gy, (unsigned)n), dim3(256), 0, (hipStream_t)stream, a);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_conv1d_bwd_data_skinny_multi(int32_t n, const stegan_conv_desc* const* descs, const float* const* dy,
                                                   const float* const* w, const float* const* scale,
                                                   const float* const* mask_src, const float* const* add_src,
                                                   float* const* dx, uint16_t* const* dxh, void* stream) {
    STEGAN_CHECK_ARG(n >= 1 && descs && dy && w && dx);
    if (n > SK_MULTI) return 1;
    SkMulti a;
    a.n = n;
    long blocks = 1;
    for (int i = 0; i < n; ++i) {
        SkP& p = a.p[i];
        int rc = fill(descs[i], p, true);
        if (rc) return rc;
        STEGAN_CHECK_ARG(dy[i] && w[i] && dx[i]);
        p.dy = dy[i]; p.w = w[i]; p.scale = scale ? scale[i] : nullptr; p.out = dx[i];
        const bool lm = descs[i]->in_act == STEGAN_ACT_LRELU;
        STEGAN_CHECK_ARG(!lm || (mask_src && mask_src[i]));
        p.mask = lm ? mask_src[i] : nullptr;
        p.add = (lm && add_src) ? add_src[i] : nullptr;
        if (!lm && add_src && add_src[i]) return 1;           // a plain add without the mask: not this kernel's case
        p.out_h = (dxh && descs[i]->Cin % 8 == 0) ? dxh[i] : nullptr;
        a.ne[i] = (long)descs[i]->B * descs[i]->Cin * descs[i]->Tin;
        const long b = (a.ne[i] + 255) / 256;
        if (b > blocks) blocks = b;
    }
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv_cout1_bwd_data_multi_kernel, dim3((unsigned)blocks, (unsigned)n), dim3(256), 0,
                       (hipStream_t)stream, a);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_conv1d_bwd_data_skinny(const stegan_conv_desc* d, const float* dy, const float* w, const float* scale,
                                             const float* mask_src, const float* add_src, float* dx, uint16_t* dxh,
                                             void* stream) {
    SkP p;
    int rc = fill(d, p, true);
    if (rc) return rc;
    const bool lm = d->in_act == STEGAN_ACT_LRELU;
    if (!lm && (mask_src || add_src)) return 1;
    STEGAN_CHECK_ARG(dy && w && dx && (!lm || mask_src));
    p.dy = dy; p.w = w; p.scale = scale; p.out = dx;
    p.mask = lm ? mask_src : nullptr; p.add = lm ? add_src : nullptr;
    p.out_h = (dxh && d->Cin % 8 == 0) ? dxh : nullptr;
    const long n = (long)d->B * d->Cin * d->Tin;
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv_cout1_bwd_data_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, n);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

// wgrad geometry of one problem (shared by the single and the multi launch)
static void sk_wgrad_geometry(const stegan_conv_desc* d, SkP& p, int& cblocks, int& slices, int& qtiles) {
    p.TT = d->Tin < 64 ? d->Tin : 64;
    p.CL = 256 / p.TT;
    if (p.CL > d->Cin) p.CL = d->Cin;
    if (p.CL * d->K > 256) p.CL = 256 / d->K;
    p.fdTT = make_fastdiv((uint32_t)p.TT); p.fdCL = make_fastdiv((uint32_t)p.CL);
    cblocks = ceil_div(d->Cin, p.CL); qtiles = ceil_div(d->Tin, p.TT);
    slices = ceil_div(512, cblocks * qtiles);             // ~2 blocks per CU; every slice writes a slab of Cin x K sums
    if (slices > d->B) slices = d->B;
    if (slices > 64) slices = 64;
    p.per = ceil_div(d->B, slices);
    slices = ceil_div(d->B, p.per);
    p.nqt = qtiles;
}

static int64_t sk_wgrad_floats(const stegan_conv_desc* d, int slices, int qtiles) {
    return ((int64_t)slices * qtiles * d->Cin * d->K + slices + 3) & ~(int64_t)3;
}

extern "C" int64_t stegan_conv1d_bwd_weight_skinny_floats(const stegan_conv_desc* d) {
    SkP p;
    if (fill(d, p)) return 0;
    int cb, sl, qt;
    sk_wgrad_geometry(d, p, cb, sl, qt);
    return sk_wgrad_floats(d, sl, qt);
}

// dw[i] (+ db[i], may be NULL) of n <= 5 single-output-channel convs in one launch; returns 1 (nothing enqueued) unless
// stegan_conv1d_bwd_weight_skinny would take every problem
extern "C" int stegan_conv1d_bwd_weight_skinny_multi(int32_t n, const stegan_conv_desc* const* descs, const float* const* x,
                                                     const float* const* dy, float* const* dw, float* const* db,
                                                     float* workspace, int64_t workspace_floats, void* stream) {
    STEGAN_CHECK_ARG(n >= 1 && descs && x && dy && dw);
    if (n > SK_MULTI || !workspace) return 1;
    SkMultiW a;
    a.n = n; a.first[0] = 0;
    int gy = 1, gz = 1;
    int64_t used = 0;
    int most = 1;
    for (int i = 0; i < n; ++i) {
        SkP& p = a.p[i];
        int rc = fill(descs[i], p);
        if (rc) return rc;
        STEGAN_CHECK_ARG(x[i] && dy[i] && dw[i]);
        p.x = x[i]; p.dy = dy[i]; p.out = dw[i];
        a.db[i] = db ? db[i] : nullptr;
        int cb, sl, qt;
        sk_wgrad_geometry(descs[i], p, cb, sl, qt);
        a.first[i + 1] = a.first[i] + cb;
        a.nsl[i] = sl;
        p.part = workspace + used;
        used += sk_wgrad_floats(descs[i], sl, qt);
        if (used > workspace_floats) return 1;              // (nothing enqueued yet)
        if (sl > gy) gy = sl;
        if (qt > gz) gz = qt;
        if (descs[i]->Cin * descs[i]->K > most) most = descs[i]->Cin * descs[i]->K;
    }
    for (int i = n; i < SK_MULTI; ++i) { a.db[i] = nullptr; a.nsl[i] = 0; }
    hipLaunchKernelGGL(conv_cout1_wgrad_multi_kernel, dim3((unsigned)a.first[n], (unsigned)gy, (unsigned)gz), dim3(256), 0,
                       (hipStream_t)stream, a);
    hipLaunchKernelGGL(conv_cout1_wgrad_reduce_kernel, dim3((unsigned)ceil_div(most, 256), (unsigned)n), dim3(256), 0,
                       (hipStream_t)stream, a);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}

extern "C" int stegan_conv1d_bwd_weight_skinny(const stegan_conv_desc* d, const float* x, const float* dy, float* dw,
                                               float* db, float* workspace, int64_t workspace_floats, void* stream) {
    const stegan_conv_desc* descs[1] = {d};
    const float* xs[1] = {x};
    const float* dys[1] = {dy};
    float* dws[1] = {dw};
    float* dbs[1] = {db};
    return stegan_conv1d_bwd_weight_skinny_multi(1, descs, xs, dys, dws, dbs, workspace, workspace_floats, stream);
}
