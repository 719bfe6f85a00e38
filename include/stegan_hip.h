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
 *     no synchronisation, no global state, so they are re-entrant on distinct streams and can
 *     be captured into a hipGraph;
 *   - return value: 0 = ok, STEGAN_E_ARG = invalid argument / unsupported shape,
 *     otherwise a positive hipError_t; no C++ exception crosses the ABI;
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

#define STEGAN_ABI_VERSION 1
int stegan_abi_version(void);
/* Host-only self check of the index helpers (magic division); returns 0 when correct. */
int stegan_selftest_host(void);

/* ---- activation / layout codes ---------------------------------------------------------- */
#define STEGAN_ACT_NONE 0
#define STEGAN_ACT_RELU 1   /* input-side: ReLU applied while gathering x (pre-activation blocks) */
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
    int32_t out_act;    /* STEGAN_ACT_NONE | STEGAN_ACT_LRELU | STEGAN_ACT_TANH */
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

/* ---- elementwise helpers of the backward pass --------------------------------------------- */
/* dx[b,c,t] = dxe[b,c,2t] + dxe[b,c,2t+1]   (backward of nn.Upsample(scale_factor=2), conv.py:40,55) */
int stegan_fold_upsample2(const float* dxe, float* dx, int64_t rows, int32_t T, void* stream);

/*
 * out[i] = dy[i] * act'(pre[i]) written through the stored OUTPUT y: leaky_relu(0.1) keeps the
 * sign (y>0 ? 1 : 0.1), tanh' = 1 - y^2.  (autograd of F.leaky_relu discriminator.py:38,64,90,116 and
 * torch.tanh generator.py:160).  Pointers must be 16-byte aligned.
 */
int stegan_act_bwd(const float* dy, const float* y, float* out, int64_t n, int32_t act, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STEGAN_HIP_H */
