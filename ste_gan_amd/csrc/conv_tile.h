// One-tile-per-CU convolution for the stride-1, ungrouped layers (conv_tile.hip): interface towards conv_bf16.hip, which
// offers every forward / input-gradient launch whose source comes as an activation image to it first.
#pragma once
#include "common.h"

// What conv_bf16.hip knows about a launch when it asks (launch_bf16): the operands, the fused epilogue and the tap progression
// of the (single) output phase.  Pointers as in DirP; ``mode`` 0 forward, 1 input gradient (stride 1: one phase).
struct TileConvArgs {
    const unsigned short* xh;     // source image [B][ceil(Csrc / 8)][Tsrc][8]
    const unsigned short* wh;     // weight copy [K + 1][Cpad / 8][Mpad][8] of the direction (stegan_conv_prep_bf16)
    const float* scale;           // mode 0: per output channel; mode 1: ONE factor (nullable)
    const float* bias;            // mode 0
    const float* res;             // mode 0 residual / mode 1 add_src
    const float* mask;            // mode 1: ReLU mask source (B, Cdst, Tmask), or (mask_lrelu) the leaky-ReLU output
    float* out;                   // (B, Cdst, Tdst)
    unsigned short* out_h;        // image of the output or null
    int B, Csrc, Cdst, Tsrc, Tq, Tdst, Tmask, ups, mode;
    int C2, M, Mpad, Cpad, K;
    int ntaps, kbase, kstep, off_min, step;
    int in_relu, out_act, mask_lrelu;
};

// 0: launched; 1: the kernel does not take the shape (the caller goes on to its own plans); other: error status
int stegan_tile_conv(const TileConvArgs& a, hipStream_t stream);
// whether stegan_tile_conv would take the launch (no launch): bench.py names the kernel family by it
bool stegan_tile_conv_takes(const TileConvArgs& a);

// ---- weight gradient (conv_tile_wgrad.hip), offered every stride-1 ungrouped launch with both operands as images by
// conv_wgrad_bf16.hip (wgh_run)
struct TileWgradArgs {
    const unsigned short* xh;     // [B][Cin / 8][Tin][8]
    const unsigned short* dyh;    // [B][Cout / 8][Tout][8]
    const float* dy;              // fp32 dy (bias gradient)
    float* dw;                    // (Cout, Cin, K), accumulated into when the launch runs in one slice
    float* db;                    // nullable
    float* partial;               // slabs [slice][Cout][Cin][Keff] when the launch is sliced over the batch
    int64_t partial_floats;
    int B, Cin, Cout, K, dil, pad, Tin, Tout, ups, relu;
    int k_lo, Keff;               // live taps
    int bias_in_reduce;           // the caller computes db in its slab reduce when the launch is sliced
    // a PENDING slab sum of an earlier launch on the same stream (another layer: dw_prev[i] += sum_s slab_prev[s][i] over 16-byte
    // pieces, every tap live), run by the consumer waves while the producers set up -- the launch boundary is its hand-off, so
    // no fences and no reduce launch (prev_n4 == 0: none)
    const float* prev_partial;
    float* prev_dw;
    int64_t prev_n4;
    int prev_slices;
};
// 0: launched (*slices_out slabs to sum when > 1; 1: dW has been accumulated; *bias_done: db has been taken care of);
// 1: declined; other: error status
int stegan_tile_wgrad(const TileWgradArgs& a, int* slices_out, int* bias_done, hipStream_t stream);
// n <= 5 layers in ONE launch (per-layer outputs in the arrays); 1: declined as a whole (nothing enqueued)
int stegan_tile_wgrad_multi(int n, const TileWgradArgs* a, int* slices_out, int* bias_done, hipStream_t stream);
