/*
 * stegan_hip.h -- C ABI of libstegan_hip.so: the MI355X (gfx950) kernels behind the STE-GAN
 * train / inference hot path.
 *
 * The reference (scheck-k/ste-gan) has no FFI layer: its device boundary is ATen, reached
 * through torch.nn modules.  Each entry point below therefore names the ATen call site(s) of
 * the reference hot path that it replaces (paths relative to the reference root); the Python
 * host side (ste_gan_amd/) binds them with ctypes and mirrors the reference's module API.
 *
 * Conventions:
 *   - plain pointers + POD descriptors, no torch types; all pointers are DEVICE pointers unless
 *     the parameter name ends in _host;
 *   - the caller owns every buffer (inputs, outputs, workspaces);
 *   - functions only ENQUEUE work on `stream` (a hipStream_t passed as void*): no allocation,
 *     no synchronisation, no state kept between calls, so they are re-entrant on distinct
 *     streams and can be captured into a hipGraph.  Two things live outside a call, both read-only
 *     on the launch path: (1) the planners' experiment knobs (environment variables STEGAN_*,
 *     read ONCE per process at first use; none of them changes results -- the switches that do
 *     exist in the diagnostic builds only, scripts/build_stamps.sh); (2) the opt-in measurement
 *     facility stegan_kernel_timing* (a process-wide list of event pairs, off unless bench.py's
 *     instrumented pass switches it on; never under stream capture);
 *   - return value: 0 = ok, STEGAN_E_ARG = invalid argument / unsupported shape, 1 = the entry
 *     point's kernels do not take the shape (nothing enqueued: the *_h / *_hx / *_t / *_skinny
 *     families, whose callers then use the generic entry point), otherwise a hipError_t; no C++
 *     exception crosses the ABI;
 *   - activations are fp32, layout (B, C, T) with T contiguous unless stated otherwise.
 */
#ifndef STEGAN_HIP_H
#define STEGAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STEGAN_OK 0
#define STEGAN_E_ARG (-1)
/* informational success of the *_hx conv entry points: the launch went to the one-tile-per-CU kernels of round 5
 * (conv_tile.hip / conv_tile_wgrad.hip) instead of the round 2-4 kernels -- same results contract; bench.py names the
 * kernel family of a launch by it.  Callers that only want success treat it as STEGAN_OK. */
#define STEGAN_OK_TILE 2

#define STEGAN_ABI_VERSION 1
int stegan_abi_version(void);
/* The 2-byte operand type this build of the library rounds matrix-core operands to: "bf16" (libstegan_hip.so) or "f16"
 * (libstegan_hip_f16.so, the same sources compiled with -DSTEGAN_OPERAND_F16).  Every entry point below whose name or
 * comment says "bf16" -- the *_h / *_hx conv entry points, stegan_conv_prep_bf16*, the activation images -- means THIS type:
 * the reference reaches the path through ``torch.autocast(dtype=torch.float16, enabled=cfg.train.mixed_precision)``
 * (ste_gan/train.py:181,204); BASELINE configs 3 / 4 name bf16, config 5 fp16. */
const char* stegan_operand_type(void);
/* Host-only self check of the index helpers (magic division); returns 0 when correct. */
int stegan_selftest_host(void);

/* ---- activation / layout codes ---------------------------------------------------------- */
#define STEGAN_ACT_NONE 0
#define STEGAN_ACT_RELU 1   /* input-side: ReLU applied while gathering x (pre-activation blocks);
                               output-side: F.relu of the encoder ResBlock / FFN (conv.py:125,133)   */
#define STEGAN_ACT_LRELU 2  /* output-side: leaky_relu(0.1)   (discriminator.py:38,64,90,116)    */
#define STEGAN_ACT_TANH 3   /* output-side: tanh              (generator.py:160)                 */

/*
 * One 1-D convolution y = out_act( scale[co] * (w (*) in_act(upsample(x))) + bias[co] + residual ).
 *
 * Replaces, fused: nn.ReLU -> nn.Upsample(nearest x2) -> F.conv1d (groups / stride / dilation /
 * zero padding) -> + bias -> + residual -> leaky_relu | tanh, i.e. the ATen sequence behind
 *   ste_gan/layers/conv.py:16-17 (WNConv1d), :38-75 (GBlock conv1 / res1 / conv2), :82-84,
 *   ste_gan/layers/conv.py:89-101 (NormedConv1d / NormedConv2d with (k,1) kernels),
 *   ste_gan/layers/conv.py:111-133 (encoder ResBlock conv + eval BatchNorm folded into scale/bias),
 *   nn.Linear in models/emg_encoder.py:59,67-68,79,88 and layers/transformer.py:33-35 (K = 1).
 * `w` is the RAW direction tensor (weight_v / weight_orig / weight), layout (Cout, Cin/groups, K);
 * the weight-norm / spectral-norm / BatchNorm factor is the per-output-channel `scale`
 * (NULL = 1), so the re-parametrised weight is never materialised in HBM.
 */
typedef struct stegan_conv_desc {
    int32_t B;          /* batch (MPD: batch * period, the period axis is folded into the batch) */
    int32_t Cin, Cout;  /* total channels */
    int32_t groups;
    int32_t K, stride, dilation, pad;
    int32_t Tin;        /* stored input length, BEFORE the optional nearest upsample */
    int32_t Tout;       /* output length = (Tin*upsample + 2*pad - dilation*(K-1) - 1)/stride + 1 */
    int32_t upsample;   /* 1 or 2 */
    int32_t in_act;     /* STEGAN_ACT_NONE | STEGAN_ACT_RELU */
    int32_t out_act;    /* STEGAN_ACT_NONE | STEGAN_ACT_RELU | STEGAN_ACT_LRELU | STEGAN_ACT_TANH */
    int32_t out_btc;    /* 0: y is (B,Cout,Tout); 1: y is (B,Tout,Cout)  (generator.py:157) */
} stegan_conv_desc;

int stegan_conv1d_fwd(const stegan_conv_desc* d, const float* x, const float* w, const float* scale,
                      const float* bias, const float* residual, float* y, void* stream);

/*
 * Gradient w.r.t. the conv input (autograd of the call sites above; ATen convolution_backward
 * input-grad).  dy is the gradient w.r.t. the conv output BEFORE out_act (the caller applies
 * the activation derivative).  dx has the EFFECTIVE input resolution (B, Cin, Tin*upsample):
 *   dx[b,ci,te] = relu'(mask_src[b,ci,te/upsample]) * sum_{co,k} scale[co] w[co,ci,k] dy[b,co,q]
 *                 + add_src[b,ci,te],      q*stride = te + pad - k*dilation
 * mask_src (pre-activation input, (B,Cin,Tin)) is used iff in_act == RELU; add_src may be NULL.
 */
int stegan_conv1d_bwd_data(const stegan_conv_desc* d, const float* dy, const float* w, const float* scale,
                           const float* mask_src, const float* add_src, float* dx, void* stream);

/*
 * Gradient w.r.t. the raw direction tensor, ACCUMULATED (atomic fp32 add) into dw, which the
 * caller zeroes once per step:  dw[co,ci,k] += sum_{b,t} dy[b,co,t] * in_act(upsample(x))[b,ci,t*s+k*d-p].
 * (ATen convolution_backward weight-grad; the scale / norm factors are applied by the
 * re-parametrisation backward below.)   db (nullable): db[co] += sum_{b,t} dy[b,co,t].
 */
int stegan_conv1d_bwd_weight(const stegan_conv_desc* d, const float* x, const float* dy, float* dw,
                             float* db, void* stream);
/*
 * Same contract with a caller-owned workspace: the reduction over (batch, time) is cut into batch slices that write
 * their partial dw to slabs of `workspace` with plain stores, and a second kernel adds the slabs into dw in a fixed
 * order (deterministic, and far cheaper than float atomics, which cost one memory transaction per element).
 * Cout * Cin/groups * K floats per slice; any size works (a short workspace means fewer, longer slices; NULL / too
 * small for one slab falls back to the atomic path above).
 */
int stegan_conv1d_bwd_weight_ws(const stegan_conv_desc* d, const float* x, const float* dy, float* dw, float* db,
                                float* workspace, int64_t workspace_floats, void* stream);

/*
 * Fast path of stegan_conv1d_fwd / stegan_conv1d_bwd_data (same semantics, same descriptors): the direct
 * convolution kernel that keeps halo'd input rows in LDS and takes K-major transposed copies of the direction
 * tensor, produced once per weight version by stegan_conv_prep:
 *     wf[g][k][ci][co] (leading dimension round_up(Cout/groups, 64))  for the forward pass,
 *     wb[g][k][co][ci] (leading dimension round_up(Cin/groups, 64))   for the input gradient, pre-multiplied by the
 *                      per-output-channel `scale` (nullable) -- the weight-norm / BatchNorm factor; in
 *                      stegan_conv1d_bwd_data_t `scale` is then only a nullable pointer to ONE uniform factor
 *                      (spectral norm's 1/sigma, which changes on every forward).
 * Buffers of stegan_conv_prep_floats(d, 0 | 1) floats; they must be ZERO-initialised once by the caller (prep only
 * writes the unpadded part).  The *_t launches return 1 (nothing enqueued) for a shape they do not cover (more
 * than 48 live taps per phase, or an input window that does not fit the LDS budget); the caller then uses the
 * generic entry points.  A strided convolution's input gradient is decomposed into `stride` dense phases, so no
 * multiply-accumulates are spent on the zero-inserted positions.
 */
int64_t stegan_conv_prep_floats(const stegan_conv_desc* d, int32_t which);
int stegan_conv_prep(const stegan_conv_desc* d, const float* v, const float* scale, float* wf, float* wb, void* stream);
/*
 * Multi-tensor stegan_conv_prep: one launch for all n convs of a network.  table_dev = device array of n
 *   struct { const float* v; const float* scale; float* wf; float* wb;
 *            int32 Cout_g, Cin_g, K, G, LDf, LDb, tiles_e, tiles_co; }      (LD* = round_up(.,64), tiles_* = ceil(./32))
 * max_tiles = max over entries of G * tiles_e * tiles_co.
 */
int stegan_conv_prep_multi(int32_t n, const void* table_dev, int32_t max_tiles, void* stream);
/*
 * When a launch would leave most CUs idle (small B*T), the reduction over channels is split over up to 16 blocks
 * per output tile; the raw partial sums go to `workspace` (caller-owned, stegan_conv_workspace_floats(d, mode)
 * floats is always enough; a smaller or NULL workspace just lowers / disables the split) and a second kernel sums
 * them in a fixed order and applies the fused epilogue -- results stay bit-reproducible (no float atomics).
 */
int64_t stegan_conv_workspace_floats(const stegan_conv_desc* d, int32_t mode);
int stegan_conv1d_fwd_t(const stegan_conv_desc* d, const float* x, const float* wf, const float* scale,
                        const float* bias, const float* residual, float* y, float* workspace,
                        int64_t workspace_floats, void* stream);
/*
 * Up to 5 independent forward convolutions in ONE launch (the same layer of the period discriminators: one channel
 * structure, different weights, batch and length).  Arrays of n host pointers to descriptors / device buffers
 * (scale, bias, residual arrays or entries may be NULL).  Returns 1 when the problems cannot share a launch (different
 * kernel variants, many-tap or narrow-group layers): the caller then issues stegan_conv1d_fwd_t per problem.
 */
int stegan_conv1d_fwd_multi(int32_t n, const stegan_conv_desc* const* descs, const float* const* x,
                            const float* const* wf, const float* const* scale, const float* const* bias,
                            const float* const* residual, float* const* y, void* stream);
/* Input-gradient counterpart of stegan_conv1d_fwd_multi (per-channel scales folded into wb; no uniform scale). */
int stegan_conv1d_bwd_data_multi(int32_t n, const stegan_conv_desc* const* descs, const float* const* dy,
                                 const float* const* wb, const float* const* mask_src, const float* const* add_src,
                                 float* const* dx, void* stream);
int stegan_conv1d_bwd_data_t(const stegan_conv_desc* d, const float* dy, const float* wb, const float* scale,
                             const float* mask_src, const float* add_src, float* dx, float* workspace,
                             int64_t workspace_floats, void* stream);

/*
 * bf16-operand variants of the two launches above (BASELINE configs 3-5; the CUDA reference reaches its reduced-
 * precision path through torch.autocast, ste_gan/train.py:181,204): identical semantics, descriptors, epilogues and
 * fp32 activations / accumulators, but the two operands of the contraction are rounded to bf16 while they are staged
 * (v_mfma_f32_32x32x16_bf16, 16x the fp32 matrix rate).  They read bf16 copies of the direction tensor written by
 * stegan_conv_prep_bf16 (only LIVE elements are written: the caller zero-fills both buffers ONCE when it allocates
 * them; padding rows / channels and the extra tap plane then stay zero; K < 48):
 *     wfh[g][k][ci / 8][co][8 ci]  co padded to 128, ci to 64, k = 0..K with plane K all zero   (forward)
 *     wbh[g][k][co / 8][ci][8 co]  ci padded to 128, co to 64, pre-multiplied by scale[co]      (input gradient)
 * of stegan_conv_prep_bf16_elems(d, 0 | 1) uint16 elements.  Return 1 (nothing enqueued) for shapes left to the exact
 * fp32 kernels: fewer than 8 reduction channels or fewer than 8 output rows per group, more than 48 live taps,
 * windows that do not fit LDS.  The *_multi form takes a device table of n
 *   struct { const float* v; const float* scale; uint16_t* wfh; uint16_t* wbh;
 *            int32 Cout_g, Cin_g, K, G, Mf, Cf, Mb, Cb; int64 nf, nb; }     (M* / C* the padded dims, n* the elements)
 * and runs blocks_per_entry blocks of 256 threads per entry (grid-stride over the entry's tiles,
 * stegan_conv_prep_bf16_tiles).
 */
int64_t stegan_conv_prep_bf16_elems(const stegan_conv_desc* d, int32_t which);
/* work items (tiles) of the copy kernel for one conv: the grid of stegan_conv_prep_bf16_multi should cover the largest */
int64_t stegan_conv_prep_bf16_tiles(const stegan_conv_desc* d);
int stegan_conv_prep_bf16(const stegan_conv_desc* d, const float* v, const float* scale, uint16_t* wfh, uint16_t* wbh,
                          void* stream);
int stegan_conv_prep_bf16_multi(int32_t n, const void* table_dev, int32_t blocks_per_entry, void* stream);
int stegan_conv1d_fwd_h(const stegan_conv_desc* d, const float* x, const uint16_t* wfh, const float* scale,
                        const float* bias, const float* residual, float* y, float* workspace,
                        int64_t workspace_floats, void* stream);
int stegan_conv1d_bwd_data_h(const stegan_conv_desc* d, const float* dy, const uint16_t* wbh, const float* scale,
                             const float* mask_src, const float* add_src, float* dx, float* workspace,
                             int64_t workspace_floats, void* stream);
/*
 * Round 3: bf16 ACTIVATION IMAGES.  A conv kernel can write, beside its fp32 output (B, C, T), the same values rounded to
 * bf16 in the channel-block-major layout its consumers stage by LDS-DMA: image[b][c / 8][t][c % 8], i.e. (B, ceil(C/8), T, 8)
 * uint16 (whole 8-channel blocks per group only; otherwise the request is ignored), and can take its input as such an
 * image instead of converting the fp32 tensor while it is staged.  xh / dyh and yh / dxh are nullable; every other
 * argument is as for stegan_conv1d_fwd_h / _bwd_data_h / _fwd_multi_h / _bwd_data_multi_h (which are these with null
 * images).  Reference: the same layers/conv.py convolutions; the images carry no values of their own.
 */
int stegan_conv1d_fwd_hx(const stegan_conv_desc* d, const float* x, const uint16_t* xh, const uint16_t* wfh,
                         const float* scale, const float* bias, const float* residual, float* y, uint16_t* yh,
                         float* workspace, int64_t workspace_floats, void* stream);
int stegan_conv1d_bwd_data_hx(const stegan_conv_desc* d, const float* dy, const uint16_t* dyh, const uint16_t* wbh,
                              const float* scale, const float* mask_src, const float* add_src, float* dx, uint16_t* dxh,
                              float* workspace, int64_t workspace_floats, void* stream);
/* stegan_adamw_step / stegan_wn_backward that ZERO the gradient they consume (zero_grads / zero_dw != 0): the arenas then
 * need no fill before the next backward pass (torch.optim semantics keep .grad until zero_grad(); the graph-replayed
 * train step switches this on, GraphedTrainStep) */
int stegan_adamw_step_z(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, const float* hyper_dev,
                        float beta1, float beta2, float eps, float weight_decay, float grad_scale, int32_t zero_grads,
                        void* stream);
int stegan_wn_backward_z(int32_t n, const float* const* v, const float* const* g, float* const* dw, float* const* dv,
                         float* const* dg, const int32_t* rows, const int32_t* cols, int32_t max_rows, int32_t zero_dw,
                         void* stream);
/* kernel-timestamp timing of the bf16 conv / weight-gradient launches (bench.py's roofline pass): while on, each such launch
 * carries a start / stop event pair holding the dispatch's own begin / end timestamps (hipExtLaunchKernelGGL), so a duration is
 * the kernel's, without the marker packets and dispatch gaps of an event bracket.  _count: pairs recorded so far;
 * _collect: waits for them, writes their durations (ms, launch order) and empties the list.  Not for use under stream capture. */
int stegan_kernel_timing(int32_t on);
int64_t stegan_kernel_timing_count(void);
int stegan_kernel_timing_collect(float* ms, int64_t cap);
/* diagnostic time mark: a one-lane kernel stores the 100 MHz real-time counter into *slot (scripts/step_marks.py) */
int stegan_timestamp(uint64_t* slot, void* stream);
/* the elementwise producers of the path with an image of their output beside it (out_h / dx_h / y_h / dz_h / dqkv_h nullable;
 * C % 8 == 0; act_bwd additionally T % 4 == 0): as the functions of the same name without `_img` */
int stegan_act_bwd_add_img(const float* dy, const float* add, const float* y, float* out, uint16_t* out_h, int64_t n, int32_t C,
                           int32_t T, int32_t act, void* stream);
int stegan_act_bwd_add_multi_img(int32_t m, const float* const* dy, const float* const* add, const float* const* y,
                                 float* const* out, uint16_t* const* out_h, const int64_t* n, const int32_t* Cs,
                                 const int32_t* Ts, int32_t act, void* stream);
int stegan_fold_upsample2_img(const float* dxe, float* dx, uint16_t* dx_h, int64_t rows, int32_t C, int32_t T, void* stream);
int stegan_relpos_attention_fwd_img(const float* qkv, const float* rel_emb, float* probs, float* out, uint16_t* out_h, int32_t B,
                                    int32_t H, int32_t A, int32_t T, int32_t max_rel, void* stream);
int stegan_relpos_attention_bwd_img(const float* qkv, const float* rel_emb, const float* probs, const float* dout,
                                    float* workspace, float* dqkv, uint16_t* dqkv_h, int32_t B, int32_t H, int32_t A, int32_t T,
                                    int32_t max_rel, void* stream);
int stegan_add_layernorm_fwd_img(const float* x, const float* residual, const float* gamma, const float* beta, float* y,
                                 uint16_t* y_h, float* xhat, float* rstd, int32_t B, int32_t C, int32_t T, float eps,
                                 void* stream);
int stegan_add_layernorm_bwd_img(const float* dy, const float* xhat, const float* rstd, const float* gamma, float* dz,
                                 uint16_t* dz_h, int32_t B, int32_t C, int32_t T, void* stream);
/* weight gradient with both operands as images (nullable; taken only when BOTH are given): as stegan_conv1d_bwd_weight_h */
int stegan_conv1d_bwd_weight_hx(const stegan_conv_desc* d, const float* x, const uint16_t* xh, const float* dy,
                                const uint16_t* dyh, float* dw, float* db, float* workspace, int64_t workspace_floats,
                                void* stream);
/* the same with the sum of the batch-slice slabs DEFERRED: when the launch sliced the batch, *deferred describes the pending sum
 * (dw[r * K + k_lo + kk] += sum over slices of partial[slice][r * K_eff + kk]; used_floats of the workspace hold the slabs and must
 * stay untouched until it ran) and no reduce kernel is launched; slices == 0: nothing is pending (dW was accumulated directly, or
 * the shape went elsewhere: return 1 as stegan_conv1d_bwd_weight_hx).  stegan_wgrad_reduce_multi runs the pending sums of any number of
 * layers as one launch per 48 (same summation order per element as the per-layer kernel: identical results); the caller orders it
 * after every weight-gradient launch it covers (ops.wgrad_join). */
typedef struct stegan_wgrad_reduce {
    const float* partial;
    float* dw;
    int64_t per_tap, used_floats;
    int32_t slices, K, k_lo, K_eff;
} stegan_wgrad_reduce;
int stegan_conv1d_bwd_weight_hxd(const stegan_conv_desc* d, const float* x, const uint16_t* xh, const float* dy,
                                 const uint16_t* dyh, float* dw, float* db, float* workspace, int64_t workspace_floats,
                                 stegan_wgrad_reduce* deferred, void* stream);
int stegan_wgrad_reduce_multi(int32_t n, const stegan_wgrad_reduce* items, void* stream);
/* the CHAINED form (round 5): `prev` (nullable, or slices == 0: none) is the pending sum that an earlier call ON THE SAME STREAM
 * left in its `next`; it is run at the head of this launch -- by the tile kernel's consumer waves while its producers set up,
 * with the launch boundary as the hand-off (no fences), when the kernel takes the shape; else as a reduce launch of its own --
 * so a chain of n sliced layers costs one reduce launch (its tail's: stegan_wgrad_reduce_multi(1, &next)) instead of n.  The
 * slabs of `prev` must stay untouched until this call's kernels have run: give consecutive calls different workspaces.
 * Same adds in the same order as the reduce kernel: identical results.  Return 1: nothing was enqueued and `prev` is still
 * pending. */
int stegan_conv1d_bwd_weight_hxc(const stegan_conv_desc* d, const float* x, const uint16_t* xh, const float* dy,
                                 const uint16_t* dyh, float* dw, float* db, float* workspace, int64_t workspace_floats,
                                 const stegan_wgrad_reduce* prev, stegan_wgrad_reduce* next, void* stream);
/* Up to 5 layers' weight gradients in ONE launch of the one-tile-per-CU kernel (round 5: the five convs of a generator block;
 * the blocks of different layers follow each other on a CU, so their fixed phases overlap and the batch is cut in 2 slices
 * instead of 4).  All or nothing: 1 = nothing enqueued and `prev` untouched unless EVERY layer is stride 1, ungrouped, with both
 * operands as images and admitted by the kernel's plan.  On success (STEGAN_OK_TILE) next[i] is layer i's pending slab sum
 * (slices == 0: none; run it with stegan_wgrad_reduce_multi or hand it on as `prev`), and every prev[j], j < n_prev, has been
 * run: inside problem j's prologue when it is the 16-byte case, else by one reduce launch.  workspace[i]: layer i's slabs. */
int stegan_conv1d_bwd_weight_multi_hxc(int32_t n, const stegan_conv_desc* const* descs, const float* const* x,
                                       const uint16_t* const* xh, const float* const* dy, const uint16_t* const* dyh,
                                       float* const* dw, float* const* db, float* const* workspace,
                                       const int64_t* workspace_floats, int32_t n_prev, const stegan_wgrad_reduce* prev,
                                       stegan_wgrad_reduce* next, void* stream);
int stegan_conv1d_fwd_multi_hx(int32_t n, const stegan_conv_desc* const* descs, const float* const* x,
                               const uint16_t* const* xh, const uint16_t* const* wfh, const float* const* scale,
                               const float* const* bias, const float* const* residual, float* const* y,
                               uint16_t* const* yh, void* stream);
int stegan_conv1d_bwd_data_multi_hx(int32_t n, const stegan_conv_desc* const* descs, const float* const* dy,
                                    const uint16_t* const* dyh, const uint16_t* const* wbh, const float* const* mask_src,
                                    const float* const* add_src, float* const* dx, uint16_t* const* dxh, void* stream);
/* bf16 forms of stegan_conv1d_fwd_multi / stegan_conv1d_bwd_data_multi (same contract: up to 5 problems, return 1 when they
 * cannot share one kernel variant or one of them is left to the fp32 kernels). */
int stegan_conv1d_fwd_multi_h(int32_t n, const stegan_conv_desc* const* descs, const float* const* x,
                              const uint16_t* const* wfh, const float* const* scale, const float* const* bias,
                              const float* const* residual, float* const* y, void* stream);
int stegan_conv1d_bwd_data_multi_h(int32_t n, const stegan_conv_desc* const* descs, const float* const* dy,
                                   const uint16_t* const* wbh, const float* const* mask_src, const float* const* add_src,
                                   float* const* dx, void* stream);

/*
 * bf16-operand variant of stegan_conv1d_bwd_weight_ws (same accumulate-into-dw contract, fp32 accumulation, fp32 bias
 * gradient): both operands are rounded to bf16 while they are staged and read with the transposing LDS load, so any tap
 * shift / stride is a per-lane address.  The batch is cut into slices that write slabs of `workspace`
 * ([slice][live tap][Cout][Cin/groups] floats) and a second kernel adds them into dw in a fixed order.  Returns 1 (nothing
 * enqueued) for shapes left to the fp32 kernels: fewer than 16 channels per group on either side, or no room for one slab.
 */
int stegan_conv1d_bwd_weight_h(const stegan_conv_desc* d, const float* x, const float* dy, float* dw, float* db,
                               float* workspace, int64_t workspace_floats, void* stream);

/* db[c] += sum_{b,t} dy[b,c,t] -- the bias gradient alone (also folded into stegan_conv1d_bwd_weight). */
int stegan_bias_grad(const float* dy, float* db, int32_t B, int32_t C, int32_t T, void* stream);

/*
 * Launch plan of the three conv entry points (mode 0 = fwd, 1 = bwd_data, 2 = bwd_weight): the block tile edge
 * (64 or 128 -> kernel instantiation conv_igemm_kernel<mode, tile/64, tile/64>), the number of reduction splits
 * and the multiply-accumulates the matrix cores issue once dead (all-padding) taps are skipped.  Host only;
 * used by bench.py to attribute measured kernel time to algorithmic work.
 */
int stegan_conv1d_plan(const stegan_conv_desc* d, int32_t mode, int32_t* tile, int32_t* splits, int64_t* live_macs);

/* ---- weight re-parametrisations ---------------------------------------------------------- */
/*
 * Multi-tensor weight-norm scale: for every conv i, scale_i[co] = g_i[co] / ||v_i[co,:]||_2
 * (torch._weight_norm, dim 0, no eps -- ste_gan/layers/conv.py:16-17,91,98).
 * The tables are DEVICE arrays of n entries: v / g / scale pointers, rows (=Cout), cols (=Cin/g*K).
 */
int stegan_wn_scales(int32_t n, const float* const* v, const float* const* g, float* const* scale,
                     const int32_t* rows, const int32_t* cols, int32_t max_rows, void* stream);

/* Single-tensor form of stegan_wn_scales (direct pointers, used by the per-module autograd path). */
int stegan_wn_scale1(const float* v, const float* g, float* scale, int32_t rows, int32_t cols, void* stream);

/*
 * Weight-norm backward, multi-tensor: from the raw-direction gradient dw (as produced by
 * stegan_conv1d_bwd_weight) compute  ds[co] = <dw[co,:], v[co,:]>,  dg[co] = ds/||v||,
 * dv[co,:] = (g/||v||) dw[co,:] - (ds * g / ||v||^3) v[co,:]     (autograd of torch._weight_norm).
 */
int stegan_wn_backward(int32_t n, const float* const* v, const float* const* g, const float* const* dw,
                       float* const* dv, float* const* dg, const int32_t* rows, const int32_t* cols,
                       int32_t max_rows, void* stream);

/* Single-tensor form of stegan_wn_backward. */
int stegan_wn_backward1(const float* v, const float* g, const float* dw, float* dv, float* dg,
                        int32_t rows, int32_t cols, void* stream);

/*
 * Spectral norm (old-style torch.nn.utils.spectral_norm, n_power_iterations=1, eps=1e-12, dim=0) of a
 * (rows=Cout, cols=Cin/g*K) matrix -- ste_gan/layers/conv.py:92-94 via discriminator.py:136,172.
 * power_iterate != 0 (module in training mode): v <- normalize(W^T u); u <- normalize(W v), IN PLACE,
 * exactly once per call (the reference runs it on each of the 4 D forwards of a step).  Then
 * sigma = u^T W v, *sigma is written and scale[0..rows) = 1/sigma (the conv epilogue factor).
 * workspace: stegan_spectral_norm_workspace_floats(rows, cols) floats.
 */
int64_t stegan_spectral_norm_workspace_floats(int32_t rows, int32_t cols);
int stegan_spectral_norm_fwd(const float* w_orig, float* u, float* v, float* scale, float* sigma,
                             float* workspace, int32_t rows, int32_t cols, int32_t power_iterate, void* stream);
/* The same for n <= 8 independent layers (all spectral-norm convs of one scale discriminator, discriminator.py:100-113)
 * in one launch per stage: per-layer arrays of the single-layer arguments; u_save / v_save (arrays, or NULL, or NULL
 * entries) receive copies of the updated u / v for the backward pass (torch clones them for the same reason). */
int stegan_spectral_norm_fwd_multi(int32_t n, const float* const* w_orig, float* const* u, float* const* v,
                                   float* const* scale, float* const* sigma, float* const* u_save,
                                   float* const* v_save, float* const* workspace, const int32_t* rows,
                                   const int32_t* cols, int32_t power_iterate, void* stream);
/* dw_orig = dw/sigma - (<dw,W>/sigma^2) u v^T, dw = raw-direction gradient of stegan_conv1d_bwd_weight
 * (autograd of weight_orig / sigma with u, v held constant).  workspace: >= rows floats. */
int stegan_spectral_norm_bwd(const float* w_orig, const float* dw, const float* u, const float* v,
                             const float* sigma, float* workspace, float* dw_orig, int32_t rows, int32_t cols,
                             void* stream);
/* the same for n layers (or n calls on one layer: fake and real pass) as ONE launch per stage -- the transforms feed the
 * optimiser only, so a step can run all of them when its gradients are finalized (round 5: 16 launches -> 2). */
int stegan_spectral_norm_bwd_acc_multi(int32_t n, const float* const* w_orig, const float* const* dw, const float* const* u,
                                       const float* const* v, const float* const* sigma, float* const* workspace,
                                       float* const* dw_orig, const int32_t* rows, const int32_t* cols, int32_t accumulate,
                                       void* stream);
/* accumulate != 0: dw_orig += ... (the parameter's gradient arena view: no autograd accumulation kernel afterwards) */
int stegan_spectral_norm_bwd_acc(const float* w_orig, const float* dw, const float* u, const float* v,
                                 const float* sigma, float* workspace, float* dw_orig, int32_t rows, int32_t cols,
                                 int32_t accumulate, void* stream);

/* ---- discriminator input plumbing ---------------------------------------------------------- */
/*
 * Multi-period fold: right reflect-pad x (B,T,C) by n = period - T % period (a FULL period when
 * divisible), view as (B,C,H,period), and emit it with the period axis folded into the batch:
 * out[(b*period + w), c, h] = xpad[b, h*period + w, c], shape (B*period, C, H), H = (T+n)/period.
 * Replaces F.pad(reflect) + view of discriminator.py:36-37,86-87 (+ the rearrange of :145,181); the
 * (k,1) Conv2d stack then runs as Conv1d over H.  _bwd scatters the gradient back (optionally +=).
 */
int stegan_mpd_fold_fwd(const float* x_btc, float* out, int32_t B, int32_t T, int32_t C, int32_t period, void* stream);
int stegan_mpd_fold_bwd(const float* dout, float* dx_btc, int32_t B, int32_t T, int32_t C, int32_t period,
                        int32_t accumulate, void* stream);
/*
 * The folds of all n <= 8 period discriminators of a MultiPeriodDiscriminator (discriminator.py:174-183: every
 * sub-discriminator pads and folds the same x) in ONE launch: outs[k] receives the fold for periods[k].  _bwd_multi
 * forms dx = sum_k fold_k^T(douts[k]) in one launch -- the sum autograd accumulates over the sub-discriminators'
 * input gradients.  outs / douts / periods are HOST arrays of n entries.
 */
int stegan_mpd_fold_fwd_multi(const float* x_btc, float* const* outs, int32_t B, int32_t T, int32_t C,
                              const int32_t* periods, int32_t n, void* stream);
int stegan_mpd_fold_bwd_multi(const float* const* douts, float* dx_btc, int32_t B, int32_t T, int32_t C,
                              const int32_t* periods, int32_t n, int32_t accumulate, void* stream);
/* rearrange 'b t c -> b c t' (discriminator.py:145,181) and its transpose (optionally +=). */
int stegan_btc_to_bct(const float* x, float* y, int32_t B, int32_t T, int32_t C, void* stream);
int stegan_bct_to_btc(const float* x, float* y, int32_t B, int32_t T, int32_t C, int32_t accumulate, void* stream);
/* nn.AvgPool1d(kernel_size=4, stride=2, padding=1) (discriminator.py:140,153) on (rows, T); bwd: dx = pool^T dy + add. */
int stegan_avgpool421_fwd(const float* x, float* y, int64_t rows, int32_t T, void* stream);
int stegan_avgpool421_bwd(const float* dy, const float* add, float* dx, int64_t rows, int32_t T, void* stream);

/* ---- frozen EMG encoder: attention + LayerNorm on the (B, C, T) layout ----------------------- */
/*
 * Multi-head self attention with learned relative-position logits
 * (ste_gan/layers/transformer.py:88-113 + :168-306, eval mode: no dropout).
 * qkv: (B, 3*H*A, T), channel = {q,k,v}*H*A + h*A + a (the fused projection conv's output);
 * rel_emb: (H, 2*max_rel-1, A) (= relative_positional.embeddings[...,0]);
 * logits[i,j] = q_i.k_j/sqrt(A) + q_i.emb[h, max_rel-1+(j-i)]  for |j-i| <= max_rel-1, -1e8 added beyond;
 * probs: (B,H,T,T) softmax (saved for backward); out: (B, H*A, T) = sum_j probs[i,j] v_j.
 * _bwd: gradient w.r.t. qkv only (the encoder is frozen); workspace: B*H*T*T floats.
 */
int stegan_relpos_attention_fwd(const float* qkv, const float* rel_emb, float* probs, float* out, int32_t B, int32_t H,
                                int32_t A, int32_t T, int32_t max_rel, void* stream);
int stegan_relpos_attention_bwd(const float* qkv, const float* rel_emb, const float* probs, const float* dout,
                                float* workspace, float* dqkv, int32_t B, int32_t H, int32_t A, int32_t T,
                                int32_t max_rel, void* stream);
/*
 * y = LayerNorm_C(x + residual) * gamma + beta over the channel axis of (B,C,T)
 * (post-LN residual blocks, transformer.py:56-60; nn.LayerNorm eps 1e-5).  xhat (B,C,T) / rstd (B*T)
 * are saved for _bwd, which returns dz = d/d(x+residual) (the same tensor is the gradient of both).
 */
int stegan_add_layernorm_fwd(const float* x, const float* residual, const float* gamma, const float* beta, float* y,
                             float* xhat, float* rstd, int32_t B, int32_t C, int32_t T, float eps, void* stream);
int stegan_add_layernorm_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, float* dz,
                             int32_t B, int32_t C, int32_t T, void* stream);

/* ---- losses: scalar (accumulated into *loss_slot, a device float) + input gradient in one pass ---- */
/*
 * mode 0: weight * mean|a - b|      (F.l1_loss: feature matching, ste_gan/train.py:257-262)
 * mode 1: weight * mean (a - b)^2   (F.mse_loss: LSGAN train.py:192-196,209-211; waveform :251-254)
 * b == NULL compares against the constant `target` (torch.zeros_like / ones_like targets).
 * da (nullable) receives d(weight*loss)/da.  loss_slot may be NULL (gradient only).
 */
int stegan_pair_loss_fwd_bwd(const float* a, const float* b, float target, int64_t n, int32_t mode, float weight,
                             float* loss_slot, float* da, void* stream);
/* n pairs in one launch (the feature-matching sum, train.py:257-262): weight * sum_i mean(|a_i - b_i|) (mode 0) or
 * weight * sum_i mean((a_i - b_i)^2) (mode 1) is ADDED to *loss_slot and each da[i] (may be NULL) receives pair i's
 * gradient; b == NULL or b[i] == NULL: the constant `target`. */
int stegan_pair_loss_multi(int32_t n, const float* const* a, const float* const* b, const int64_t* numel, float target,
                           int32_t mode, float weight, float* loss_slot, float* const* da, void* stream);
/* the same with per-pair HOST arrays: weight[i]; constant targets (b NULL / b[i] NULL) target[i] for elements
 * [0, split[i]) and target2[i] from split[i] on (split NULL or split[i] <= 0: one target) -- the discriminator step's
 * batched fake | real logits against zeros | ones (train.py:192-196) without slicing the batch. */
int stegan_pair_loss_multi_ex(int32_t n, const float* const* a, const float* const* b, const int64_t* numel,
                              const float* weight, const float* target, const float* target2, const int64_t* split,
                              int32_t mode, float* loss_slot, float* const* da, void* stream);
/* the same with a REPRODUCIBLE value (round 5): every block writes its share of the sum into `partials` (at most 128 floats per
 * pair; with fewer than that many floats left the launch falls back to atomic adds) and a one-block kernel adds the shares
 * into *loss_slot in a fixed order -- two launches on the same inputs report the same bits.  The gradients never went through
 * atomics in either form.  `partials` is scratch: free again when the call's kernels have run. */
int stegan_pair_loss_multi_det(int32_t n, const float* const* a, const float* const* b, const int64_t* numel,
                               const float* weight, const float* target, const float* target2, const int64_t* split,
                               int32_t mode, float* loss_slot, float* const* da, float* partials, int64_t partials_floats,
                               void* stream);
/* out = x * (*scalar_dev): chains an upstream 0-dim gradient into a stored unit gradient without a host sync. */
int stegan_scale_by_scalar(const float* x, const float* scalar_dev, float* out, int64_t n, void* stream);
/*
 * Multi-resolution time-domain feature loss (losses/time_domain_loss.py:35-73,87-107 with
 * layers/average_filter.py:22-28): per scale (win, shift): low = avg9(avg9(x)) (reflect pad 4),
 * rect = |x - low|, frames over a reflect-padded (win/2) axis, features [mean low, sum low^2,
 * sum rect^2, mean rect], L1 mean against the real signal's features; summed over n_scales (<= 3).
 * x_* are (B,T,C).  *loss_slot += weight*loss; per_scale[n_scales] (nullable) = unweighted losses;
 * dx_gen (nullable) = d(weight*loss)/dx_gen.  workspace: stegan_multi_td_workspace_floats(B,C) floats.
 * stegan_multi_td_fwd_bwd is the reference's default configuration (9-tap filter, padded windowing) with rows short
 * enough for LDS (9*T floats per row in 150 KiB); stegan_multi_td_fwd_bwd_ex takes every option of
 * TimeDomainFeatureLoss.__init__ (time_domain_loss.py:20-33): avg_window = average_filter_window_size (odd),
 * pad_windowing = apply_padding_windowing, and rows of any length -- when stegan_multi_td_scratch_floats() returns
 * n > 0 the row buffers do not fit LDS and the caller passes a scratch of n floats (whole-utterance validation,
 * train.py:329-336).
 */
int64_t stegan_multi_td_workspace_floats(int32_t B, int32_t C);
int stegan_multi_td_fwd_bwd(const float* x_real, const float* x_gen, int32_t B, int32_t T, int32_t C,
                            int32_t n_scales, const int32_t* win_host, const int32_t* shift_host, float weight,
                            float* loss_slot, float* per_scale, float* dx_gen, float* workspace, void* stream);
int64_t stegan_multi_td_scratch_floats(int32_t B, int32_t T, int32_t C, int32_t n_scales, const int32_t* win_host,
                                       const int32_t* shift_host, int32_t pad_windowing);
int stegan_multi_td_fwd_bwd_ex(const float* x_real, const float* x_gen, int32_t B, int32_t T, int32_t C,
                               int32_t n_scales, const int32_t* win_host, const int32_t* shift_host, int32_t avg_window,
                               int32_t pad_windowing, float weight, float* loss_slot, float* per_scale, float* dx_gen,
                               float* workspace, float* scratch, int64_t scratch_floats, void* stream);
/* TimeDomainFeatureLoss.calculate_time_domain_features (time_domain_loss.py:58-69): x (B,T,C) -> feats (B,F,C,4),
 * F = (T + 2*(pad_windowing ? win/2 : 0) - win) / shift + 1.  scratch: 3*B*C*T floats. */
int stegan_td_features(const float* x, int32_t B, int32_t T, int32_t C, int32_t win, int32_t shift, int32_t avg_window,
                       int32_t pad_windowing, float* feats, float* scratch, void* stream);
/* AverageFilter.forward (layers/average_filter.py:22-28) on `rows` rows of length T: reflect pad by window/2 when
 * pad_signal, AvgPool1d(window, stride 1); y has T (padded) or T - window + 1 columns.  _bwd: dx (rows,T) from dy. */
int stegan_average_filter_fwd(const float* x, float* y, int64_t rows, int32_t T, int32_t window, int32_t pad_signal,
                              void* stream);
int stegan_average_filter_bwd(const float* dy, float* dx, int64_t rows, int32_t T, int32_t window, int32_t pad_signal,
                              void* stream);
/* weight * mean_rows ||target - pred + 1e-6||_2 (F.pairwise_distance, losses/emg_encoder_loss.py:63-67).
 * workspace: rows floats. */
int stegan_su_loss_fwd_bwd(const float* target, const float* pred, int32_t rows, int32_t D, float weight,
                           float* loss_slot, float* dpred, float* workspace, void* stream);
/*
 * weight * cross_entropy(logits (rows,P<=64), targets) (losses/emg_encoder_loss.py:80-83) plus the
 * EMGEncoderLossOutput counters (:27-53), bit-exact integers accumulated into counters[4] =
 * {num_phones, num_silence, num_correct, num_correct_no_silence}; argmax uses torch's first-maximum
 * tie rule.  dlogits / counters / argmax_out are nullable.  workspace: rows floats.
 */
int stegan_phoneme_ce_fwd_bwd(const float* logits, const int64_t* targets, int32_t rows, int32_t P,
                              int32_t silence_index, float weight, float* loss_slot, int32_t* counters,
                              float* dlogits, int64_t* argmax_out, float* workspace, void* stream);

/* ---- batch assembly ---------------------------------------------------------------------- */
/*
 * The corpus resident in HBM: every utterance's fields concatenated along time, with row-offset tables
 * (n_utts + 1 entries, in rows) beside them.  Field shapes per utterance follow EMGDataset.__getitem__
 * (ste_gan/data/emg_dataset.py:277-294): EMG (T, emg_channels) fp32, soft speech units (F, unit_dim) fp32,
 * phoneme ids (F,) int64, MFCCs (M, mfcc_dim) fp32 (NULL when the corpus has none), session / speaking-mode
 * indices int64 scalars.
 */
typedef struct stegan_corpus {
    const float* emg;
    const int64_t* emg_row_off;
    const float* units;
    const int64_t* phonemes;
    const int64_t* frame_off;      /* shared by units and phonemes */
    const float* mfcc;
    const int64_t* mfcc_row_off;
    const int64_t* session_idx;
    const int64_t* mode_idx;
    int32_t n_utts, emg_channels, unit_dim, mfcc_dim;
} stegan_corpus;

/*
 * Crop-and-stack of one batch -- replaces ste_gan_collate's per-utterance slicing + torch.stack
 * (ste_gan/data/collate.py:77-83 EMG, :124-129 units / phonemes / MFCCs, :101-102 indices) and the
 * DataLoader worker hop in front of it (data/loader.py:66-73).  For batch row b, utterance utt[b] and
 * window start start_frame[b] (feature frames; the host draws it as collate.py:118-121 does):
 *   emg_out[b]      (n_emg, emg_channels)      = emg[start*hopsize : start*hopsize + n_emg]
 *   units_out[b]    (n_frames, unit_dim)       = units[start : start + n_frames]
 *   phonemes_out[b] (n_frames)                 = phonemes[start : start + n_frames]
 *   mfcc_out[b]     (2*n_frames, mfcc_dim)     = mfcc[2*start : 2*start + 2*n_frames]   (skipped when NULL)
 *   session_out[b], mode_out[b]                = the utterance's indices
 * utt / start_frame are device int32[B].  The host must have checked that every window lies inside its
 * utterance (the reference's torch.stack raises otherwise); the kernel never reads outside an utterance
 * and zero-fills what a bad window would have read.
 */
int stegan_assemble_batch(const stegan_corpus* corpus, const int32_t* utt, const int32_t* start_frame, int32_t B,
                          int32_t n_frames, int32_t n_emg, int32_t hopsize, float* emg_out, float* units_out,
                          int64_t* phonemes_out, float* mfcc_out, int64_t* session_out, int64_t* mode_out,
                          void* stream);

/* ---- optimiser -------------------------------------------------------------------------------- */
/*
 * One AdamW step (decoupled weight decay; torch.optim.AdamW single-tensor arithmetic) over a flat
 * arena of n fp32 parameters -- replaces ste_gan.OPTIMIZER(...).step() (constants.py:57, train.py:199,267).
 * hyper_dev = device float[4] {lr, 1-beta1^t, 1-beta2^t, t as int32 bits} (kept on the device so a
 * captured graph can be replayed with new values).  grads are multiplied by grad_scale first
 * (1/world_size after a sum all-reduce).  All arenas 16-byte aligned.
 */
int stegan_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                      const float* hyper_dev, float beta1, float beta2, float eps, float weight_decay,
                      float grad_scale, void* stream);

/*
 * t += 1 and the two bias corrections 1-beta^t (evaluated in double, like torch's Python-side
 * `1 - beta ** step`, optim/adamw.py) on the device, stream-ordered ahead of stegan_adamw_step: the host never
 * writes step-dependent scalars, so eager steps and graph replays may queue arbitrarily far ahead.
 */
int stegan_adamw_advance(float* hyper_dev, double beta1, double beta2, void* stream);

/*
 * ---- loss scaling on the device (fp16 operand build; usable with any) -----------------------------------------------------
 * torch.cuda.amp.GradScaler as the reference drives it -- ONE scaler for both optimisers: scaler.scale(loss_D).backward();
 * scaler.step(optD); ... scaler.scale(loss_G).backward(); scaler.step(optG); scaler.update()  (ste_gan/train.py:151,198-199,
 * 266-268) -- with every decision taken on the device, so that a captured step replays it:
 *   state        device float[8]: [0] scale S, [1] 1 / S, [2] growth tracker (int32 bits), [3] non-finite gradients seen by
 *                the check of the current optimiser step, [4] ... by any check since the last update, [5] skipped optimiser
 *                steps so far (int32 bits).  The host initialises {S, 1 / S, 0, ...} once.
 *   stegan_loss_grad_scale(scale_dev)
 *                from now on every GRADIENT written by the loss entry points (stegan_pair_loss_*, stegan_multi_td_fwd_bwd*,
 *                stegan_su_loss_fwd_bwd, stegan_phoneme_ce_fwd_bwd) is multiplied by *scale_dev, read when the kernel runs;
 *                the loss values are not.  NULL turns it off.  Host-side state of the library (not thread-safe).
 *   stegan_grad_scaler_check(grads, n, state)
 *                = the inf / NaN test of scaler.step(): state[3] = state[4] = 1 when any of the n gradients is not finite
 *                (state[3] is cleared first).  Run it on the gradients the optimiser is about to read, behind the
 *                data-parallel all-reduce (which spreads a non-finite value to every rank).
 *   stegan_adamw_advance_s / stegan_adamw_step_zs(..., state)
 *                as stegan_adamw_advance / stegan_adamw_step_z with gradients multiplied by state[1] (unscale) -- or, when
 *                state[3] != 0, a skipped step: no parameter, moment or step-count change; the gradients are still zeroed
 *                when zero_grads is set.  state == NULL: the plain entry points.
 *   stegan_grad_scaler_update(state, growth_factor, backoff_factor, growth_interval)
 *                = scaler.update(): S *= backoff_factor and tracker = 0 when state[4] != 0, else ++tracker and S *=
 *                growth_factor every growth_interval clean iterations (torch defaults: 65536, 2, 0.5, 2000); clears [3], [4].
 */
int stegan_loss_grad_scale(const float* scale_dev);
int stegan_grad_scaler_check(const float* grads, int64_t n, float* state, void* stream);
int stegan_grad_scaler_update(float* state, float growth_factor, float backoff_factor, int32_t growth_interval, void* stream);
int stegan_adamw_advance_s(float* hyper_dev, double beta1, double beta2, const float* scaler_state, void* stream);
int stegan_adamw_step_zs(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, const float* hyper_dev,
                         float beta1, float beta2, float eps, float weight_decay, float grad_scale, int32_t zero_grads,
                         const float* scaler_state, void* stream);

/* ---- elementwise helpers of the backward pass --------------------------------------------- */
/* dx[b,c,t] = dxe[b,c,2t] + dxe[b,c,2t+1]   (backward of nn.Upsample(scale_factor=2), conv.py:40,55) */
/*
 * Single-output-channel convs (the output layer of every sub-discriminator, discriminator.py:31,60,81,113) as streaming
 * fp32 VALU kernels -- same arguments and results as stegan_conv1d_fwd / _bwd_data / _bwd_weight with the RAW direction
 * tensor w (1, Cin, K) and scale (1 value or NULL).  Return 1 (nothing launched) unless Cout == 1, groups == 1,
 * upsample == 1, K <= 8, no input activation / residual / mask / add / (B, T, C) output.
 */
int stegan_conv1d_fwd_skinny(const stegan_conv_desc* d, const float* x, const float* w, const float* scale,
                             const float* bias, const float* residual, float* y, void* stream);
/* (input gradient with d->in_act == STEGAN_ACT_LRELU, round 4 -- also stegan_conv1d_bwd_data_hx / _multi_hx: x is the
 * leaky-ReLU(0.1) OUTPUT of the layer before (discriminator.py:38,64,90,116), and the launch applies that layer's
 * activation derivative itself: dx = (conv_T(dy) * scale + add_src) * (mask_src > 0 ? 1 : 0.1), mask_src = x, add_src =
 * the feature-matching gradient of x or NULL (train.py:257-262); dxh (nullable): dx as a bf16 activation image.  The
 * separate activation-backward launch per layer goes away.) */
int stegan_conv1d_bwd_data_skinny(const stegan_conv_desc* d, const float* dy, const float* w, const float* scale,
                                  const float* mask_src, const float* add_src, float* dx, uint16_t* dxh, void* stream);
/* (weight gradient: the blocks' partial sums go to `workspace` -- stegan_conv1d_bwd_weight_skinny_floats() floats at most
 * -- and a second launch adds them into dw / db in a fixed order: no float atomics, reproducible gradients; returns 1
 * when the workspace is missing or too small) */
int stegan_conv1d_bwd_weight_skinny(const stegan_conv_desc* d, const float* x, const float* dy, float* dw, float* db,
                                    float* workspace, int64_t workspace_floats, void* stream);
int64_t stegan_conv1d_bwd_weight_skinny_floats(const stegan_conv_desc* d);
/* The forward / input gradient of n <= 5 single-output-channel convs in ONE launch (the output convs of the period
 * discriminators, discriminator.py:31: one per sub-discriminator, own batch B * period and length each).  Arrays of n
 * HOST entries; returns 1 (nothing enqueued) unless stegan_conv1d_fwd_skinny would take every problem. */
int stegan_conv1d_fwd_skinny_multi(int32_t n, const stegan_conv_desc* const* descs, const float* const* x,
                                   const float* const* w, const float* const* scale, const float* const* bias,
                                   float* const* y, void* stream);
int stegan_conv1d_bwd_data_skinny_multi(int32_t n, const stegan_conv_desc* const* descs, const float* const* dy,
                                        const float* const* w, const float* const* scale, const float* const* mask_src,
                                        const float* const* add_src, float* const* dx, uint16_t* const* dxh, void* stream);
/* ... and their weight gradients: dw[i] += raw direction gradient, db[i] (array or entries may be NULL) += sum of dy[i]. */
int stegan_conv1d_bwd_weight_skinny_multi(int32_t n, const stegan_conv_desc* const* descs, const float* const* x,
                                          const float* const* dy, float* const* dw, float* const* db, float* workspace,
                                          int64_t workspace_floats, void* stream);

int stegan_fold_upsample2(const float* dxe, float* dx, int64_t rows, int32_t T, void* stream);

/*
 * out[i] = dy[i] * act'(pre[i]) written through the stored OUTPUT y: leaky_relu(0.1) keeps the
 * sign (y>0 ? 1 : 0.1), tanh' = 1 - y^2.  (autograd of F.leaky_relu discriminator.py:38,64,90,116 and
 * torch.tanh generator.py:160).  Pointers must be 16-byte aligned.
 */
int stegan_act_bwd(const float* dy, const float* y, float* out, int64_t n, int32_t act, void* stream);
/* out[i] = (dy[i] + add[i]) * act'(.): the sum autograd forms when a tensor has two consumers (a discriminator feature
 * map: the next conv and the feature-matching L1 term, train.py:236-239), folded into the activation backward.
 * add may be NULL (= stegan_act_bwd). */
int stegan_act_bwd_add(const float* dy, const float* add, const float* y, float* out, int64_t n, int32_t act,
                       void* stream);
/* stegan_act_bwd_add for m <= 8 tensors in one launch (layer L of the period discriminators); HOST arrays of m entries,
 * add (the array, or single entries) may be NULL. */
int stegan_act_bwd_add_multi(int32_t m, const float* const* dy, const float* const* add, const float* const* y,
                             float* const* out, const int64_t* n, int32_t act, void* stream);

/* out = in[0] + in[1] (+ in[2] + in[3]) elementwise, summed left to right (n_in in 2..4; `in` a HOST array of device
 * pointers; out may alias in[0]): the gradient of a tensor with several autograd consumers in one launch and a fixed order,
 * where the reference's autograd engine issues one ATen add per extra consumer (ops.fan_out). */
int stegan_sum_n(int32_t n_in, const float* const* in, float* out, int64_t numel, void* stream);
/* out[i] = sum_j coeff[i * n_in + j] * *in[j] (n_in, n_out <= 8; HOST arrays of device pointers / coefficients): the loss
 * terms of a step combined into its reported scalars in one launch (train.py:213-264, loss_G += weight * term). */
int stegan_scalar_combine(int32_t n_in, const float* const* in, int32_t n_out, float* const* out, const float* coeff,
                          void* stream);

/*
 * Generator input (generator.py:143-154): out[b, c, t] = c < F ? x[b, t, c] : w1[id1[b], c - F] (: w2[id2[b], c - F - E])
 * -- the session (and optional speaking-mode) embedding gathered, broadcast over time, concatenated behind the speech
 * features and stored 'b c t' -- in one launch; out_h (nullable, F + E (+ E) a multiple of 8): the bf16 image of out.
 * id2 / w2 NULL: one embedding.  Ids outside their table read zeros.  _bwd: the table's gradient,
 * dw[s, j] += sum_{b: ids[b] == s} sum_t dout[b, c_off + j, t], in a fixed order (deterministic).
 */
int stegan_embed_concat_fwd(const float* x, const int64_t* id1, const float* w1, const int64_t* id2, const float* w2,
                            float* out, uint16_t* out_h, int32_t B, int32_t T, int32_t F, int32_t E, int32_t S1, int32_t S2,
                            void* stream);
int stegan_embed_concat_bwd(const float* dout, const int64_t* ids, float* dw, int32_t B, int32_t T, int32_t Ct, int32_t c_off,
                            int32_t E, int32_t S, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STEGAN_HIP_H */
