"""Thin host wrappers + autograd glue over the C ABI of libstegan_hip.so.

PyTorch is used for device memory, streams and autograd bookkeeping only; all arithmetic of
the hot path runs in the hand-written gfx950 kernels.  Every wrapper fails loudly when a
tensor is not on the GPU or the library is missing -- there is no eager / CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, replace
from typing import Optional

import torch

from . import _lib
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH, ConvDesc, check

Tensor = torch.Tensor


@dataclass(frozen=True)
class ConvCfg:
    """Static configuration of one fused conv launch (mirrors stegan_conv_desc)."""
    cin: int
    cout: int
    k: int
    stride: int = 1
    dilation: int = 1
    pad: int = 0
    groups: int = 1
    upsample: int = 1
    in_act: int = ACT_NONE
    out_act: int = ACT_NONE
    out_btc: bool = False

    def tout(self, tin: int) -> int:
        span = tin * self.upsample + 2 * self.pad - self.dilation * (self.k - 1) - 1
        if span < 0:
            raise RuntimeError(
                f"Calculated padded input size per channel: ({tin * self.upsample + 2 * self.pad}). "
                f"Kernel size: ({self.dilation * (self.k - 1) + 1}). Kernel size can't be greater than actual input size")
        return span // self.stride + 1


def _p(t: Optional[Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _gpu_f32(t: Tensor, name: str) -> Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"ste_gan_amd: '{name}' lives on {t.device}; the HIP kernels need a cuda (MI355X) tensor "
                           "and there is no CPU fallback")
    if t.dtype != torch.float32:
        raise RuntimeError(f"ste_gan_amd: '{name}' must be float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _desc(cfg: ConvCfg, B: int, tin: int, tout: int) -> ConvDesc:
    return ConvDesc(B, cfg.cin, cfg.cout, cfg.groups, cfg.k, cfg.stride, cfg.dilation, cfg.pad, tin, tout,
                    cfg.upsample, cfg.in_act, cfg.out_act, int(cfg.out_btc))


# ---------------------------------------------------------------------------------------------
# raw launches
# ---------------------------------------------------------------------------------------------
def conv1d_fwd(x: Tensor, w: Tensor, scale: Optional[Tensor], bias: Optional[Tensor],
               residual: Optional[Tensor], cfg: ConvCfg) -> Tensor:
    x = _gpu_f32(x, "x")
    w = _gpu_f32(w, "weight")
    if x.dim() != 3 or x.shape[1] != cfg.cin:
        raise RuntimeError(f"Given groups={cfg.groups}, weight of size {list(w.shape)}, expected input[B, {cfg.cin}, T] "
                           f"but got input of size {list(x.shape)}")
    B, _, tin = x.shape
    tout = cfg.tout(tin)
    y = torch.empty((B, tout, cfg.cout) if cfg.out_btc else (B, cfg.cout, tout), device=x.device, dtype=torch.float32)
    if residual is not None:
        residual = _gpu_f32(residual, "residual")
        if residual.shape != y.shape:
            raise RuntimeError(f"residual shape {list(residual.shape)} does not match conv output {list(y.shape)}")
    d = _desc(cfg, B, tin, tout)
    check(_lib.lib().stegan_conv1d_fwd(C.byref(d), _p(x), _p(w), _p(scale), _p(bias), _p(residual), _p(y), _stream()),
          "stegan_conv1d_fwd")
    return y


def conv1d_bwd_data(dy: Tensor, w: Tensor, scale: Optional[Tensor], mask_src: Optional[Tensor],
                    add_src: Optional[Tensor], cfg: ConvCfg, tin: int) -> Tensor:
    dy = _gpu_f32(dy, "dy")
    B, _, tout = dy.shape
    dx = torch.empty((B, cfg.cin, tin * cfg.upsample), device=dy.device, dtype=torch.float32)
    d = _desc(cfg, B, tin, tout)
    check(_lib.lib().stegan_conv1d_bwd_data(C.byref(d), _p(dy), _p(w), _p(scale), _p(mask_src), _p(add_src), _p(dx),
                                            _stream()), "stegan_conv1d_bwd_data")
    return dx


def conv1d_bwd_weight(x: Tensor, dy: Tensor, cfg: ConvCfg, dw: Tensor, db: Optional[Tensor]) -> None:
    """Accumulates into dw (and db)."""
    B, _, tin = x.shape
    d = _desc(cfg, B, tin, dy.shape[2])
    check(_lib.lib().stegan_conv1d_bwd_weight(C.byref(d), _p(x), _p(dy), _p(dw), _p(db), _stream()),
          "stegan_conv1d_bwd_weight")


def wn_scale(v: Tensor, g: Tensor) -> Tensor:
    v = _gpu_f32(v, "weight_v")
    g = _gpu_f32(g, "weight_g")
    rows = v.shape[0]
    scale = torch.empty(rows, device=v.device, dtype=torch.float32)
    check(_lib.lib().stegan_wn_scale1(_p(v), _p(g), _p(scale), rows, v.numel() // rows, _stream()), "stegan_wn_scale1")
    return scale


def fold_upsample2(dxe: Tensor) -> Tensor:
    B, Cc, Te = dxe.shape
    dx = torch.empty((B, Cc, Te // 2), device=dxe.device, dtype=torch.float32)
    check(_lib.lib().stegan_fold_upsample2(_p(dxe), _p(dx), B * Cc, Te // 2, _stream()), "stegan_fold_upsample2")
    return dx


# ---------------------------------------------------------------------------------------------
# autograd: weight-normalised fused conv
# ---------------------------------------------------------------------------------------------
class _WNConvFn(torch.autograd.Function):
    """y = out_act( (g/||v||)[co] * (v (*) in_act(up(x))) + bias + residual ) with hand-written backward."""

    @staticmethod
    def forward(ctx, x, v, g, bias, residual, cfg: ConvCfg):
        scale = wn_scale(v, g)
        w3 = v.reshape(v.shape[0], v.shape[1], -1)
        y = conv1d_fwd(x, w3, scale, bias, residual, cfg)
        ctx.cfg = cfg
        ctx.has_res = residual is not None
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, v, g, scale, y if cfg.out_act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        cfg: ConvCfg = ctx.cfg
        x, v, g, scale, y = ctx.saved_tensors
        dy = _gpu_f32(dy, "grad_output")
        if cfg.out_btc:
            dy = dy.transpose(1, 2).contiguous()
            if y is not None:
                y = y.transpose(1, 2).contiguous()
        if cfg.out_act != ACT_NONE:
            dy = act_bwd(dy, y, cfg.out_act)
        w3 = v.reshape(v.shape[0], v.shape[1], -1)
        dx = dv = dg = db = dres = None
        if ctx.needs_input_grad[0]:
            dxe = conv1d_bwd_data(dy, w3, scale, x if cfg.in_act == ACT_RELU else None, None, cfg, x.shape[2])
            dx = fold_upsample2(dxe) if cfg.upsample == 2 else dxe
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2] or (ctx.has_bias and ctx.needs_input_grad[3]):
            dw = torch.zeros_like(w3)
            db = torch.zeros(cfg.cout, device=dy.device, dtype=torch.float32) if ctx.has_bias else None
            conv1d_bwd_weight(x, dy, cfg, dw, db)
            dv = torch.empty_like(w3)
            dg = torch.empty(cfg.cout, device=dy.device, dtype=torch.float32)
            rows = cfg.cout
            check(_lib.lib().stegan_wn_backward1(_p(w3), _p(g), _p(dw), _p(dv), _p(dg), rows, w3.numel() // rows,
                                                 _stream()), "stegan_wn_backward1")
            dv = dv.view_as(v)
            dg = dg.view_as(g)
        if ctx.has_res and ctx.needs_input_grad[4]:
            dres = dy
        return dx, dv, dg, db, dres, None


def wn_conv1d(x, v, g, bias, residual, cfg: ConvCfg) -> Tensor:
    return _WNConvFn.apply(x, v, g, bias, residual, cfg)


def act_bwd(dy: Tensor, y: Tensor, act: int) -> Tensor:
    """dy * act'(.) expressed through the stored OUTPUT y (lrelu: sign(y) = sign(pre); tanh: 1 - y^2)."""
    out = torch.empty_like(dy)
    check(_lib.lib().stegan_act_bwd(_p(dy), _p(y), _p(out), dy.numel(), act, _stream()), "stegan_act_bwd")
    return out
