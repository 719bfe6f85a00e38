// Weight gradient of the first layers of the discriminators (8 EMG channels in): exact fp32, register-blocked (round 5).
//
//     dW[co][ci][k] += sum_{b, t} dy[b, co, t] * x[b, ci, t + k - pad]          (stride 1, ungrouped, Cin = 8, K = 3 | 15)
//
// These layers have a few hundred to 15 360 outputs and a reduction of 10^4 ... 10^5 columns: a shape no GEMM tile fills.  The
// generic implicit-GEMM kernel took 28-36 us per launch plus a 17-24 us slab sum for 0.04-0.8 GFLOP -- ten such pairs sat at the
// serial tail of the discriminator's backward pass (profiles/r05_step_timeline.txt).  Here:
//
//   * dW in matrix form [co][(ci, k)] is a GEMM with a tiny output (32 x 24 | 128 x 120) and the columns (b, t) as the
//     reduction: v_mfma_f32_16x16x4_f32 with the dy tile as A and the x windows as B, both read from LDS tiles of the chunk
//     (64 columns of one sample); an extra column of ones yields the bias gradient in the same MFMAs;
//   * every block writes ONE slab, summed in a fixed order by the reduce kernel (bit-reproducible).
//
// The operands stay fp32 (the oracle's operand emulation leaves weight gradients with fewer than 16 channels per group exact).
#include "common.h"

#include <stdio.h>
#include <stdlib.h>

namespace {

constexpr int SW_CIN = 8;
constexpr int SW_L = 64;                     // columns per chunk
constexpr int SW_XP = 84;                    // pitch of an x row in LDS (floats): 64 + 14 halo, padded
constexpr int SW_YP = 65;                    // pitch of a dy row in LDS (floats)

struct SWK {
    const float* x;
    const float* dy;
    float* partial;               // [nblocks][Cout * 8 * K + Cout]
    int B, Cout, Tin, Tout, pad, K;
    int NK;                       // 8 K: columns of dW's matrix form [co][(ci, k)]; column NK carries the bias sums
    int wnw;                      // waves along the N tiles (the rest along M)
    int cpr, chunks;              // chunks per sample row, chunks in all
    long per;                     // floats per slab
};

// exact fp32 on the matrix cores: D[16 co x 16 n] += A[16 co x 4 t] * B[4 t x 16 n] (v_mfma_f32_16x16x4_f32: the products and
// the accumulation are the fmaf chain, bit for bit), n = ci * K + k.  A = dy tile (LDS), B = x windows (LDS): lane (n, t)
// reads xs[ci(n)][t + k(n)]; the extra column n = 8 K reads a row of ones, so its sums are the bias gradient.  One MFMA does
// the work of 16 vector FMAs per lane: a single wave per SIMD keeps the fp32 pipe busy (the first version of this file did
// the same sums with v_fma and one wave per SIMD: issue-bound at a quarter of the rate, 68 us for the k = 15 layer).
template <int WM, int WNT, int COUT>
__global__ __launch_bounds__(256) void wgrad_small_kernel(const SWK k) {
    extern __shared__ __align__(16) float sm[];
    float* const xs = sm;                              // [8][SW_XP] + one row of ones + one row of zeros
    float* const ys = sm + (SW_CIN + 2) * SW_XP;       // [COUT][SW_YP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, tq = lane >> 4;
    const int mt0 = (wave / k.wnw) * WM, nt0 = (wave % k.wnw) * WNT;
    // B operand: column n of N tile j -> LDS offset of x[ci][k] (t = 0); n = NK: the ones row; n > NK: the zeros row
    int boff[WNT];
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
        const int n = (nt0 + j) * 16 + l15;
        const int ci = n / k.K, kk = n - ci * k.K;
        boff[j] = n < k.NK ? ci * SW_XP + kk : (n == k.NK ? SW_CIN * SW_XP : (SW_CIN + 1) * SW_XP);
    }
    f32x4 acc[WM][WNT];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int e = tid; e < 2 * SW_XP; e += 256) xs[SW_CIN * SW_XP + e] = e < SW_XP ? 1.f : 0.f;

    // the tiles of a chunk travel global -> registers -> LDS: the loads of chunk i + 1 are issued (all at once: a load per loop
    // trip with its LDS store behind it was a chain of 32 memory latencies, 18 us a chunk) before the MFMAs of chunk i
    constexpr int NY = COUT / 4;                 // dy tile: COUT x 64 floats over 256 threads; thread: column tid & 63, rows 4 u + wave
    constexpr int NX = (SW_CIN * SW_XP + 255) / 256;
    float yv[NY], xv[NX];
#define SW_LOAD(ch_)                                                                                                     \
    do {                                                                                                                 \
        const int b_ = (ch_) / k.cpr, t0_ = ((ch_) - b_ * k.cpr) * SW_L;                                                 \
        const float* yp_ = k.dy + ((size_t)b_ * COUT + wave) * k.Tout + t0_ + lane;                                      \
        const bool yin_ = t0_ + lane < k.Tout;                                                                           \
        _Pragma("unroll") for (int u = 0; u < NY; ++u) yv[u] = yin_ ? yp_[(size_t)(4 * u) * k.Tout] : 0.f;               \
        _Pragma("unroll") for (int u = 0; u < NX; ++u) {                                                                 \
            const int e_ = tid + 256 * u;                                                                                \
            const int c_ = e_ / SW_XP, j_ = e_ - c_ * SW_XP;                                                             \
            const int q_ = t0_ - k.pad + j_;                                                                             \
            xv[u] = (e_ < SW_CIN * SW_XP && q_ >= 0 && q_ < k.Tin) ? k.x[((size_t)b_ * SW_CIN + c_) * k.Tin + q_] : 0.f; \
        }                                                                                                                \
    } while (0)
    int ch = blockIdx.x;
    if (ch < k.chunks) SW_LOAD(ch);
    for (; ch < k.chunks; ch += gridDim.x) {
        __syncthreads();                          // the previous chunk's reads are done
#pragma unroll
        for (int u = 0; u < NY; ++u) ys[(4 * u + wave) * SW_YP + lane] = yv[u];
#pragma unroll
        for (int u = 0; u < NX; ++u)
            if (tid + 256 * u < SW_CIN * SW_XP) xs[tid + 256 * u] = xv[u];
        block_sync();
        if (ch + (int)gridDim.x < k.chunks) SW_LOAD(ch + (int)gridDim.x);
#pragma unroll 4
        for (int t = 0; t < SW_L; t += 4) {
            float a[WM], bv[WNT];
#pragma unroll
            for (int i = 0; i < WM; ++i) a[i] = ys[((mt0 + i) * 16 + l15) * SW_YP + t + tq];
#pragma unroll
            for (int j = 0; j < WNT; ++j) bv[j] = xs[boff[j] + t + tq];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WNT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bv[j], acc[i][j], 0, 0, 0);
        }
    }
#undef SW_LOAD
    // D[row co = 4 tq + e][col n = l15] -> the block's slab [co][NK] | bias sums [co]
    float* slab = k.partial + (size_t)blockIdx.x * k.per;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j) {
            const int n = (nt0 + j) * 16 + l15;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = (mt0 + i) * 16 + 4 * tq + e;
                if (n < k.NK) slab[(size_t)co * k.NK + n] = acc[i][j][e];
                else if (n == k.NK) slab[(size_t)COUT * k.NK + co] = acc[i][j][e];
            }
        }
}

// dw[i] += sum_s slab[s][i]; db likewise from the slabs' tails.  The sum runs over up to 256 slabs that lie 62 KB apart: with a
// thread per output it is a chain of 64 dependent trips to L2 (50 us measured, more than the GEMM).  Here 16 threads share an
// output -- thread g takes slabs g, g + 16, ... with every load of its chain in flight at once -- and meet in LDS; the order of
// the adds is fixed by (g, s), so two launches give the same bits.
constexpr int SR_G = 16;                                // slab groups
__global__ __launch_bounds__(256) void wgrad_small_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                 float* __restrict__ db, int nw, int nb, long per, int slabs) {
    __shared__ float red[SR_G][17];
    const int o = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + o;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < nw + nb) {
        int s = g;
        for (; s + 3 * SR_G < slabs; s += 4 * SR_G) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] += partial[(size_t)(s + u * SR_G) * per + i];
        }
        for (; s < slabs; s += SR_G) v[0] += partial[(size_t)s * per + i];
    }
    red[g][o] = (v[0] + v[1]) + (v[2] + v[3]);
    block_sync();
    if (g == 0 && i < nw + nb) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < SR_G; ++q) t += red[q][o];
        if (i < nw) dw[i] += t;
        else if (db) db[i - nw] += t;
    }
}

}  // namespace

// 0: launched; 1: not a shape for this kernel
int stegan_wgrad_small(const stegan_conv_desc* d, const float* x, const float* dy, float* dw, float* db, float* workspace,
                       int64_t workspace_floats, hipStream_t stream) {
    static const bool off = STEGAN_ENV("STEGAN_WGRAD_SMALL") && atoi(STEGAN_ENV("STEGAN_WGRAD_SMALL")) == 0;
    if (off || !workspace) return 1;
    if (d->Cin != SW_CIN || d->groups != 1 || d->stride != 1 || d->dilation != 1 || d->upsample != 1) return 1;
    if (d->in_act != STEGAN_ACT_NONE || (d->K != 3 && d->K != 15) || d->Cout > 128) return 1;
    if (d->pad > 14 || d->Tout != d->Tin + 2 * d->pad - d->K + 1) return 1;
    if (((uintptr_t)x % 4) || ((uintptr_t)dy % 4)) return 1;
    if (d->Cout % 16 != 0) return 1;
    SWK k;
    k.x = x; k.dy = dy; k.partial = workspace;
    k.B = d->B; k.Cout = d->Cout; k.Tin = d->Tin; k.Tout = d->Tout; k.pad = d->pad; k.K = d->K;
    k.NK = SW_CIN * d->K;
    const int MT = d->Cout / 16, NT = (k.NK + 1 + 15) / 16;
    k.cpr = (d->Tout + SW_L - 1) / SW_L;
    k.chunks = d->B * k.cpr;
    const int nw = d->Cout * k.NK;
    k.per = nw + d->Cout;
    long nblocks = k.chunks < 256 ? k.chunks : 256;
    if (nblocks > workspace_floats / k.per) nblocks = workspace_floats / k.per;
    if (nblocks < 1) return 1;
    const size_t lds = ((size_t)(SW_CIN + 2) * SW_XP + (size_t)d->Cout * SW_YP) * sizeof(float);
    // 4 waves: M tiles x N tiles split so that every wave owns whole tiles
    if (MT == 8 && NT == 8) { k.wnw = 1; hipLaunchKernelGGL((wgrad_small_kernel<2, 8, 128>), dim3((unsigned)nblocks), dim3(256), lds, stream, k); }
    else if (MT == 2 && NT == 2) { k.wnw = 2; hipLaunchKernelGGL((wgrad_small_kernel<1, 1, 32>), dim3((unsigned)nblocks), dim3(256), lds, stream, k); }
    else if (MT == 2 && NT == 8) { k.wnw = 2; hipLaunchKernelGGL((wgrad_small_kernel<1, 4, 32>), dim3((unsigned)nblocks), dim3(256), lds, stream, k); }
    else if (MT == 8 && NT == 2) { k.wnw = 1; hipLaunchKernelGGL((wgrad_small_kernel<2, 2, 128>), dim3((unsigned)nblocks), dim3(256), lds, stream, k); }
    else return 1;
    const int total = nw + d->Cout;
    hipLaunchKernelGGL(wgrad_small_reduce_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, stream,
                       (const float*)workspace, dw, db, nw, d->Cout, k.per, (int)nblocks);
    { const hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return -(1000 + (int)e__); }
    return STEGAN_OK;
}
