"""Convolution layers of the generator / discriminators, host side.

Mirrors ``ste_gan/layers/conv.py`` (WNConv1d :16-17, get_padding :24-25, GBlock :29-84,
NormedConv1d/2d :89-101) but every layer is ONE fused launch of the implicit-GEMM kernel
(ste_gan_amd/csrc/conv_igemm.hip): the ReLU / nearest-x2 upsample in front of a conv, the
weight-norm factor, bias, residual add and output activation are folded into it.
Parameter names and shapes equal the reference's old-style ``weight_norm`` layout
(``weight_g`` (Cout,1,1), ``weight_v`` (Cout,Cin/groups,K), ``bias``) so checkpoints interchange.
"""
from __future__ import annotations

import math
from dataclasses import replace

import torch
import torch.nn as nn

from .. import ops
from .._lib import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH


def get_padding(kernel_size, dilation=1):
    """"same" padding for odd kernels -- ste_gan/layers/conv.py:24-25."""
    return int((kernel_size * dilation - dilation) / 2)


def _owner_epoch(m) -> int:
    """Epoch of the optimiser arena that owns this layer's parameters (optim.FlatParams sets ``_owner``): the fused AdamW
    kernel writes the weights through raw pointers, which torch's version counters cannot see, and bumps its OWN network's
    epoch -- an update of the generator does not make the discriminator's prepared copies stale."""
    o = getattr(m, "_owner", None)
    return o.epoch if o is not None else 0


class WNConv1d(nn.Module):
    """Weight-normalised Conv1d (old-style ``weight_norm``: parameters weight_g / weight_v / bias).

    Extra keyword arguments describe what is fused around the convolution:
    ``in_act`` (ReLU on the input), ``upsample`` (nearest x2 on the input), ``out_act``
    (leaky_relu(0.1) / tanh on the output), ``out_btc`` (store (B,T,C)).  ``kernel_dims`` = 2
    keeps the trailing singleton of the reference's (k,1) Conv2d weights in the state dict.
    """

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 in_act=ACT_NONE, upsample=1, out_act=ACT_NONE, out_btc=False, kernel_dims=1):
        super().__init__()
        self.cfg = ops.ConvCfg(in_channels, out_channels, kernel_size, stride, dilation, padding, groups,
                               upsample, in_act, out_act, out_btc)
        kshape = (kernel_size,) + (1,) * (kernel_dims - 1)
        v = torch.empty(out_channels, in_channels // groups, *kshape)
        # same initial distribution as nn.Conv1d.reset_parameters + weight_norm (g = ||v||)
        nn.init.kaiming_uniform_(v, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(v[0].numel())
        self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))
        self.weight_g = nn.Parameter(v.flatten(1).norm(dim=1).view(-1, 1, *([1] * kernel_dims)))
        self.weight_v = nn.Parameter(v)
        self._prep = None
        self._pack = None           # set by optim.ConvPack: (pack, scale, wf, wb, dw_slice)

    def prepared(self):
        """(scale, wf, wb) for the current weights: the weight-norm factor g/||v|| and the K-major transposed copies
        the direct-conv kernel reads.  Recomputed (3 small launches) only when the weights changed: torch-side writes
        bump ``_version``; the fused AdamW kernel writes through raw pointers and bumps ``ops.WEIGHT_EPOCH`` instead."""
        if self._pack is not None:
            pack, scale, wf, wb, _ = self._pack
            pack.prepare()
            return scale, wf, wb
        v, g = self.weight_v, self.weight_g
        stamp = (ops.WEIGHT_EPOCH, _owner_epoch(self), v._version, g._version, v.data_ptr(), g.data_ptr())
        if self._prep is None or self._prep[0] != stamp:
            w3 = v.detach().reshape(v.shape[0], v.shape[1], -1)
            scale = ops.wn_scale(v.detach(), g.detach())
            bufs = ops.prep_weights(w3, self.cfg, self._prep[1][1:] if self._prep is not None and
                                    self._prep[1][1].device == v.device else None, scale)
            self._prep = (stamp, (scale, bufs[0], bufs[1]))
        return self._prep[1]

    def forward(self, x, residual=None, hand=None, fuse=None):
        """``fuse`` = (ops.ActFuse of x's producer | None, ops.ActFuse of this layer's output activation | None)."""
        sink = None
        if self._pack is not None and torch.is_grad_enabled() and self.weight_v.requires_grad:
            sink = (self._pack[4], self.bias.grad)          # raw dw slice, bias-gradient view
        return ops.wn_conv1d(x, self.weight_v, self.weight_g, self.bias, residual, self.cfg, self.prepared(), sink, hand, fuse)

    def extra_repr(self):
        return str(self.cfg)


class _Slot(nn.Module):
    """Parameter-free placeholder keeping nn.Sequential indices equal to the reference's
    (its ReLU / Upsample modules are fused into the neighbouring conv launch here)."""

    def __init__(self, what):
        super().__init__()
        self.what = what

    def extra_repr(self):
        return f"{self.what} (fused into the next conv launch)"


class GBlock(nn.Module):
    """GAN-TTS residual block -- ste_gan/layers/conv.py:29-84.

    y = conv_d3(relu(conv_d1(up(relu(x))))) + conv1x1(up(x));  out = y + conv_d27(relu(conv_d9(relu(y)))).
    Five fused launches; the two residual adds ride in the epilogues of conv_d3 / conv_d27.
    Sequential indices (conv1.1/.3 or .2/.4, res1.0 or .1, conv2.1/.3) match the reference.
    """

    def __init__(self, input_dim, output_dim, upsample=1, kernel_size=3):
        super().__init__()
        k = kernel_size
        up = upsample if upsample > 1 else 1
        c1 = [_Slot("ReLU")] + ([_Slot("Upsample")] if upsample > 1 else [])
        c1 += [WNConv1d(input_dim, output_dim, k, padding=get_padding(k), in_act=ACT_RELU, upsample=up),
               _Slot("ReLU"),
               WNConv1d(output_dim, output_dim, k, dilation=3, padding=get_padding(k, 3), in_act=ACT_RELU)]
        r1 = ([_Slot("Upsample")] if upsample > 1 else []) + [WNConv1d(input_dim, output_dim, 1, upsample=up)]
        c2 = [_Slot("ReLU"),
              WNConv1d(output_dim, output_dim, k, dilation=9, padding=get_padding(k, 9), in_act=ACT_RELU),
              _Slot("ReLU"),
              WNConv1d(output_dim, output_dim, k, dilation=27, padding=get_padding(k, 27), in_act=ACT_RELU)]
        self.conv1 = nn.Sequential(*c1)
        self.res1 = nn.Sequential(*r1)
        self.conv2 = nn.Sequential(*c2)
        self._o = 1 if upsample > 1 else 0

    def forward(self, x):
        o = self._o
        if not (ops.HANDOVER and torch.is_grad_enabled() and x.requires_grad):
            h = self.conv1[1 + o](x)
            r = self.res1[o](x)
            y = self.conv1[3 + o](h, residual=r)
            h = self.conv2[1](y)
            return self.conv2[3](h, residual=y)
        # x and y each feed two nodes: their gradients are summed inside the input-gradient kernels (ops.GradHandover)
        # instead of by two autograd adds per block.  Backward order: conv_d27 (parks d out for y) -> conv_d9 (takes) ->
        # conv_d3 -> res1 (parks its unfolded dx) -> conv_d1 (takes, folds the upsample once).
        hx, hy = ops.GradHandover(), ops.GradHandover()
        h = self.conv1[1 + o](x, hand=(ops.TAKE, hx))
        r = self.res1[o](x, hand=(ops.PARK_DX, hx))
        y = self.conv1[3 + o](h, residual=r)
        h = self.conv2[1](y, hand=(ops.TAKE, hy))
        return self.conv2[3](h, residual=y, hand=(ops.PARK_RES, hy))


class SNConv1d(nn.Module):
    """Spectral-normalised Conv1d (old-style ``spectral_norm``: parameter ``weight_orig``, buffers
    ``weight_u`` (Cout,) / ``weight_v`` (Cin/g*K,), ``bias``) -- conv.py:92-94 via discriminator.py:136.

    In training mode every forward power-iterates u / v in place once, as the reference does.
    """

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 out_act=ACT_NONE):
        super().__init__()
        self.cfg = ops.ConvCfg(in_channels, out_channels, kernel_size, stride, dilation, padding, groups,
                               1, ACT_NONE, out_act, False)
        w = torch.empty(out_channels, in_channels // groups, kernel_size)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(w[0].numel())
        self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))
        self.weight_orig = nn.Parameter(w)
        self.register_buffer("weight_u", nn.functional.normalize(torch.randn(out_channels), dim=0, eps=1e-12))
        self.register_buffer("weight_v", nn.functional.normalize(torch.randn(w[0].numel()), dim=0, eps=1e-12))
        self._prep = None
        self.grad_sink = False      # set by optim.FlatParams: .grad tensors are persistent, zeroed arena views

    def _stamp(self):
        w = self.weight_orig
        return (ops.WEIGHT_EPOCH, _owner_epoch(self), w._version, w.data_ptr())

    def prepared(self):
        w = self.weight_orig
        stamp = self._stamp()
        if self._prep is None or self._prep[0] != stamp:
            self._prep = (stamp, ops.prep_weights(w.detach(), self.cfg, self._prep[1] if self._prep is not None and
                                                  self._prep[1][0].device == w.device else None))
        return self._prep[1]

    def forward(self, x, pre=None, fuse=None):
        """``pre``: this forward's spectral-norm state from :func:`ops.spectral_norm_multi` (all layers of the
        discriminator in one call); None: the layer runs its own power iteration.  ``fuse``: see WNConv1d.forward."""
        sink = None
        w, b = self.weight_orig, self.bias
        if (self.grad_sink and torch.is_grad_enabled() and w.requires_grad and b.requires_grad and w.grad is not None
                and b.grad is not None and w.grad.is_contiguous() and w.grad.is_cuda):
            sink = (w.grad, b.grad)              # FlatParams arena views (optim.FlatParams sets ``grad_sink``)
        return ops.sn_conv1d(x, w, self.weight_u, self.weight_v, b, self.cfg, self.training, self.prepared(), pre, sink, fuse)


def NormedConv1d(*args, **kwargs):
    """Weight- or spectral-norm Conv1d constructor -- ste_gan/layers/conv.py:89-94."""
    norm = kwargs.pop("norm", "weight_norm")
    if norm == "weight_norm":
        return WNConv1d(*args, **kwargs)
    elif norm == "spectral_norm":
        kwargs.pop("kernel_dims", None)
        return SNConv1d(*args, **kwargs)


def NormedConv2d(in_channels, out_channels, kernel_size, stride=1, padding=0, norm="weight_norm", **kwargs):
    """(k,1)-kernel Conv2d of the period discriminators -- ste_gan/layers/conv.py:96-101.

    Only (k,1) kernels with (s,1) strides occur on the path (discriminator.py:25-31,76-81); they run
    as Conv1d over H with the period axis folded into the batch, keeping the 4-D weight shapes.
    """
    k = kernel_size[0] if isinstance(kernel_size, (tuple, list)) else kernel_size
    s = stride[0] if isinstance(stride, (tuple, list)) else stride
    p = padding[0] if isinstance(padding, (tuple, list)) else padding
    if isinstance(kernel_size, (tuple, list)) and kernel_size[1] != 1:
        raise ValueError("ste_gan_amd.NormedConv2d supports (k,1) kernels only")
    if norm != "weight_norm":
        raise ValueError("ste_gan_amd.NormedConv2d supports weight_norm only (the reference never uses another)")
    return WNConv1d(in_channels, out_channels, k, stride=s, padding=p, kernel_dims=2, **kwargs)
