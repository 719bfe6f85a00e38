"""Multi-period / multi-scale discriminators, host side -- mirrors ``ste_gan/models/discriminator.py``.

Class names, constructor arguments, ``forward((B,T,C)) -> list[list[Tensor]]`` (5 MPD periods
2,3,5,7,11 then 3 MSD scales; last entry of each inner list = logits), ``discriminator_names`` and
state-dict keys equal the reference's.  Each conv + leaky_relu(0.1) is one fused launch; the MPD
reflect-pad/fold, the 'b t c -> b c t' rearrange and AvgPool1d(4,2,1) are small HIP kernels.
MPD feature maps are returned as (B,C,H,p) *views* of the period-folded (B*p,C,H) buffers.
"""
from __future__ import annotations

import logging
import os

import torch
import torch.nn as nn

from .. import ops
from .._lib import ACT_LRELU
from ..layers.conv import NormedConv1d, NormedConv2d


def _unfold_view(fm, B, p):
    """(B*p, C, H) period-folded map -> the reference's (B, C, H, p) tensor (a view, no copy)."""
    _, Cc, H = fm.shape
    return fm.view(B, p, Cc, H).permute(0, 2, 3, 1)


class _PeriodDisc(nn.Module):
    def forward(self, x_btc):
        B = x_btc.shape[0]
        x = ops.mpd_fold(x_btc, self.period)
        fmaps = []
        prev = None                            # ops.ActFuse chain: layer L + 1's backward applies layer L's lrelu'
        for layer in self.layers:
            cur = ops.new_act_fuse()
            x = layer(x, fuse=(prev, cur))     # conv + leaky_relu(0.1) fused
            prev = cur
            fmaps.append(_unfold_view(x, B, self.period))
        fmaps.append(_unfold_view(self.output(x, fuse=(prev, None)), B, self.period))
        return fmaps


class DiscriminatorP(_PeriodDisc):
    """Full-size period discriminator -- discriminator.py:19-43."""

    def __init__(self, num_emg_channels: int, period, norm="weight_norm", name="DiscriminatorP"):
        super().__init__()
        self.name = name
        self.num_emg_channels = num_emg_channels
        self.layers = nn.ModuleList([
            NormedConv2d(num_emg_channels, 32, (5, 1), (3, 1), padding=(2, 0), norm=norm, out_act=ACT_LRELU),
            NormedConv2d(32, 128, (5, 1), (3, 1), padding=(2, 0), norm=norm, out_act=ACT_LRELU),
            NormedConv2d(128, 512, (5, 1), (3, 1), padding=(2, 0), norm=norm, out_act=ACT_LRELU),
            NormedConv2d(512, 1024, (5, 1), (3, 1), padding=(2, 0), norm=norm, out_act=ACT_LRELU),
            NormedConv2d(1024, 1024, (5, 1), 1, padding=(2, 0), norm=norm, out_act=ACT_LRELU)])
        self.output = NormedConv2d(1024, 1, kernel_size=(3, 1), padding=(1, 0))
        self.period = period


class DiscriminatorSmallerP(_PeriodDisc):
    """Small period discriminator -- discriminator.py:70-93."""

    def __init__(self, num_emg_channels: int, period, norm="weight_norm", name="DiscriminatorP"):
        super().__init__()
        self.name = name
        self.num_emg_channels = num_emg_channels
        self.layers = nn.ModuleList([
            NormedConv2d(num_emg_channels, 32, (3, 1), (1, 1), padding=(2, 0), norm=norm, out_act=ACT_LRELU),
            NormedConv2d(32, 256, (3, 1), (3, 1), padding=(2, 0), norm=norm, out_act=ACT_LRELU),
            NormedConv2d(256, 512, (3, 1), (3, 1), padding=(2, 0), norm=norm, out_act=ACT_LRELU)])
        self.output = NormedConv2d(512, 1, kernel_size=(3, 1), padding=(1, 0))
        self.period = period


class _ScaleDisc(nn.Module):
    def sn_prepare(self):
        """This forward's spectral-norm state of every layer (``ops.spectral_norm_multi``: the power iterations depend on
        the weights only, so they run ahead of the first conv, all layers in 4 launches; in training mode u / v advance,
        exactly once per call), or None for a weight-norm discriminator / the per-layer path."""
        if not (getattr(self, "norm", "weight_norm") == "spectral_norm" and ops.SN_BATCHED and len(self.layers) <= 8
                and self.layers[0].weight_orig.is_cuda):
            return None
        keep = torch.is_grad_enabled() and any(l.weight_orig.requires_grad for l in self.layers)
        return ops.spectral_norm_multi([(l.weight_orig, l.weight_u, l.weight_v) for l in self.layers], self.training, keep)

    def forward(self, x_bct, pre=None):
        fmaps = []
        x = x_bct
        if pre is None:
            pre = self.sn_prepare() or [None] * len(self.layers)
        # every feature map feeds exactly one conv (the feature-matching term's gradient is added inside the backward
        # kernels): the conv behind it applies the leaky-ReLU derivative in its input-gradient epilogue (ops.ActFuse)
        prev = None
        for layer, pr in zip(self.layers, pre):
            cur = ops.new_act_fuse()
            x = layer(x, fuse=(prev, cur)) if pr is None else layer(x, pr, fuse=(prev, cur))
            prev = cur
            fmaps.append(x)
        fmaps.append(self.output(x, fuse=(prev, None)))
        return fmaps


class DiscriminatorSmallerS(_ScaleDisc):
    """Small scale discriminator -- discriminator.py:47-67."""

    def __init__(self, num_emg_channels, norm="weight_norm", name="DiscriminatorS"):
        super().__init__()
        self.name = name
        self.norm = norm
        self.num_emg_channels = num_emg_channels
        self.layers = nn.ModuleList([
            NormedConv1d(num_emg_channels, 128, 15, 1, padding=7, norm=norm, out_act=ACT_LRELU),
            NormedConv1d(128, 256, 37, 2, groups=4, padding=18, norm=norm, out_act=ACT_LRELU),
            NormedConv1d(256, 512, 37, 2, groups=16, padding=18, norm=norm, out_act=ACT_LRELU),
            NormedConv1d(512, 1024, 5, 1, padding=2, norm=norm, out_act=ACT_LRELU)])
        self.output = NormedConv1d(1024, 1, 3, 1, padding=1)


class DiscriminatorS(_ScaleDisc):
    """Full-size scale discriminator -- discriminator.py:96-119."""

    def __init__(self, num_emg_channels, norm="weight_norm", name="DiscriminatorS"):
        super().__init__()
        self.name = name
        self.norm = norm
        self.num_emg_channels = num_emg_channels
        self.layers = nn.ModuleList([
            NormedConv1d(num_emg_channels, 128, 15, 1, padding=7, norm=norm, out_act=ACT_LRELU),
            NormedConv1d(128, 128, 41, 2, groups=4, padding=20, norm=norm, out_act=ACT_LRELU),
            NormedConv1d(128, 256, 41, 2, groups=16, padding=20, norm=norm, out_act=ACT_LRELU),
            NormedConv1d(256, 512, 41, 4, groups=16, padding=20, norm=norm, out_act=ACT_LRELU),
            NormedConv1d(512, 1024, 41, 4, groups=16, padding=20, norm=norm, out_act=ACT_LRELU),
            NormedConv1d(1024, 1024, 41, 1, groups=16, padding=20, norm=norm, out_act=ACT_LRELU),
            NormedConv1d(1024, 1024, 5, 1, padding=2, norm=norm, out_act=ACT_LRELU)])
        self.output = NormedConv1d(1024, 1, 3, 1, padding=1)


class _MultiDisc(nn.Module):
    PRIMES = [2, 3, 5, 7, 11]

    def _build(self, num_emg_channels, num_multi_pool, num_multi_scale, cls_p, cls_s):
        self.num_emg_channels = num_emg_channels
        self.multi_pooled_disc = nn.ModuleList([
            cls_p(num_emg_channels, self.PRIMES[i], name=f"DiscriminatorP-{self.PRIMES[i]}")
            for i in range(num_multi_pool)])
        self.multi_scale_disc = nn.ModuleList([
            cls_s(num_emg_channels=num_emg_channels, norm="spectral_norm" if i == 0 else "weight_norm",
                  name=f"DiscriminatorS-{i}")
            for i in range(num_multi_scale)])
        self.discriminator_names = [d.name for d in self.multi_pooled_disc] + [d.name for d in self.multi_scale_disc]

    def forward(self, x):
        """x: (B, T, C) -> [[fmaps..., logits] for 5 MPD + 3 MSD]  (discriminator.py:144-155 / 180-191).

        The period branch (one multi-problem launch per layer) and the scale branch run on two HIP streams, forked and
        joined here; autograd runs their backward passes on the same two streams.  (Round 1 measured EIGHT streams with
        the fp32 kernels and rejected them, 31.3 vs 29.0 ms/step: those kernels fill the matrix pipes on their own.  The
        bf16 kernels are short, latency-bound launches that leave most of the chip idle.)"""
        return self._forward(x, None, True)

    def _forward(self, x, sn_pre, fork: bool):
        self._prepare_weights()                # on the calling stream, BEFORE the fork: both branches read the copies
        x_scale, x = ops.fan_out(x, 2)         # two consumers (scale branch, period branch): their gradients meet in one sum
        xs = self._scale_inputs(ops.btc_to_bct(x_scale))           # before the fork: both streams read them
        main, side = self._fork() if fork else (torch.cuda.current_stream(), None)
        n_side = self._scales_on_side() if side is not None else 0
        scale = [None] * len(self.multi_scale_disc)
        run = lambda i: self.multi_scale_disc[i](xs[i], sn_pre[i] if sn_pre is not None else None)
        third = self._third_stream(main) if side is not None else None
        on_third = [len(scale) - n_side - 1] if (third is not None and len(scale) - n_side >= 2) else []
        if side is not None:
            self._handoff(side, [x] + xs[len(scale) - n_side:])
            with torch.cuda.stream(side):
                results = self._pooled_forward(x)
                for i in range(len(scale) - n_side, len(scale)):       # balance: the period branch is the shorter one
                    scale[i] = run(i)
        else:
            results = self._pooled_forward(x)
        if on_third:
            self._handoff(third, [xs[i] for i in on_third])
            with torch.cuda.stream(third):
                for i in on_third:
                    scale[i] = run(i)
        for i in range(len(scale) - n_side):
            if i not in on_third:
                scale[i] = run(i)
        if side is not None:
            main.wait_stream(side)
        if on_third:
            main.wait_stream(third)
        return results + scale

    @staticmethod
    def _handoff(side, tensors):
        """Tensors allocated on the calling stream and consumed on ``side`` (forward AND, as saved tensors, by the side
        branch's backward: the first conv's weight gradient reads xs[i] there).  Without this the caching allocator may
        hand such a block back to the calling stream's pool as soon as autograd drops the saved tensor, while the side
        stream's kernel is still queued -- inside a capture the reuse would be baked into the replayed graph."""
        for t in tensors:
            if t.is_cuda:
                t.record_stream(side)

    def _scales_on_side(self):
        """How many of the LAST scale discriminators run on the period branch's stream (STEGAN_D_SIDE_SCALES)."""
        return max(0, min(len(self.multi_scale_disc) - 1, int(os.environ.get("STEGAN_D_SIDE_SCALES", "1"))))

    def _scale_inputs(self, xs):
        """Inputs of the scale discriminators: x, pool(x), pool(pool(x)) (discriminator.py:150-154; the reference pools once
        more after the last scale and discards the result)."""
        out = [xs]
        for _ in range(len(self.multi_scale_disc) - 1):
            here, down = ops.fan_out(out[-1], 2)                   # this scale's discriminator | the next pooling
            out[-1] = here
            out.append(ops.avgpool421(down))
        return out

    def _prepare_weights(self):
        """Bring every conv's prepared weight copies up to date on the CURRENT stream (a no-op when they are): the
        copies are made lazily by the first conv that needs them, which after a fork would be on one branch's stream
        while the other branch reads them unordered."""
        self._prepare_sn_bf16()
        for m in self.modules():
            prep = getattr(m, "prepared", None)
            if prep is not None and m is not self:
                prep()

    def _prepare_sn_bf16(self):
        """The bf16 copies of all stale spectral-norm layers in one launch instead of one each (they are re-derived once per
        step, right after the discriminator's update: four 20 us launches on the main stream)."""
        from ..layers.conv import SNConv1d
        if not ops.bf16_enabled():
            return
        stale = [m for m in self.modules() if isinstance(m, SNConv1d) and m.weight_orig.is_cuda
                 and (m._prep is None or m._prep[0] != m._stamp())]
        if len(stale) < 2 or any(m._prep is None or getattr(m._prep[1][0], "h", None) is None for m in stale):
            return                             # first use: the per-layer path allocates the buffers
        ops.prep_bf16_many([(m.cfg, m.weight_orig.detach(), None, m._prep[1][0], m._prep[1][1]) for m in stale])
        for m in stale:
            m._prep = (m._stamp(), m._prep[1])

    _side = None
    _third = None

    def _third_stream(self, main):
        """A third stream for the middle scale discriminator (STEGAN_D_THIRD=1; experiment: with the round-2 kernels a third
        discriminator stream lost, 9.97 vs 9.24 ms per step)."""
        if os.environ.get("STEGAN_D_THIRD", "0") != "1":
            return None
        if _MultiDisc._third is None:
            _MultiDisc._third = torch.cuda.Stream()
        _MultiDisc._third.wait_stream(main)
        return _MultiDisc._third

    def _fork(self):
        """(current stream, side stream that waits for it), or (current, None) with STEGAN_D_STREAMS=0.  (In fp32 mode the
        fork lost at first, 21.9 vs 21.5 ms/step; together with the rebalanced branches and the generator's weight-gradient
        streams it wins, 17.1 vs 19.2.)  (A third stream for the
        weight-norm scale discriminators: 9.97 vs 9.24 ms/step -- with the encoder branch that makes four.)"""
        main = torch.cuda.current_stream()
        if not (ops.bf16_enabled() or ops.F32_STREAMS) or os.environ.get("STEGAN_D_STREAMS", "1") == "0":
            return main, None
        if _MultiDisc._side is None:
            _MultiDisc._side = torch.cuda.Stream()
        _MultiDisc._side.wait_stream(main)
        return main, _MultiDisc._side

    def _pooled_forward(self, x):
        """The period discriminators layer by layer: layer L of all of them is one multi-problem launch
        (``ops.wn_conv1d_multi``) -- same channel structure, own weights / batch (B * period) / length each."""
        discs = list(self.multi_pooled_disc)
        B = x.shape[0]
        hs = ops.mpd_fold_multi(x, [d.period for d in discs])
        fmaps = [[] for _ in discs]
        depth = len(discs[0].layers) + 1
        prev = None
        for L in range(depth):
            convs = [d.layers[L] if L < depth - 1 else d.output for d in discs]
            cur = ops.new_act_fuse() if L < depth - 1 else None
            hs = ops.wn_conv1d_multi(hs, convs, fuse=(prev, cur))
            prev = cur
            for f, h, d in zip(fmaps, hs, discs):
                f.append(_unfold_view(h, B, d.period))
        return fmaps

    def forward_pair(self, x_a, x_b):
        """``(self(x_a), self(x_b))`` -- the discriminator step's fake and real passes (train.py:190-191) -- with every
        stateless sub-discriminator run ONCE on the concatenated batch (same per-sample arithmetic, half the launches,
        fuller grids on the short late layers).  The spectral-norm scale discriminator stays two calls in the
        reference's order: in training mode each of its forwards advances the power iteration, so its two halves see
        different sigma estimates."""
        res_a, res_b = [], []
        for e in self._forward_pair(x_a, x_b, False):
            if e[0] == "both":
                n = e[2]
                res_a.append([f[:n] for f in e[1]]); res_b.append([f[n:] for f in e[1]])
            else:
                res_a.append(e[1]); res_b.append(e[2])
        return res_a, res_b

    def forward_pair_logits(self, x_a, x_b):
        """The logits of :meth:`forward_pair` without slicing the batch: one entry per sub-discriminator, ("both", logits
        of the concatenated batch (first ``n`` rows: x_a), n) or, for the spectral-norm discriminator, ("pair", logits_a,
        logits_b) -- the argument of ``losses.gan_losses.lsgan_pair_loss``."""
        out = []
        for e in self._forward_pair(x_a, x_b, True):
            out.append(("both", e[1][-1], e[2]) if e[0] == "both" else ("pair", e[1][-1], e[2][-1]))
        return out

    def _forward_pair(self, x_a, x_b, logits_only: bool):
        n = x_a.shape[0]
        if x_b.shape[0] != n:
            raise ValueError("forward_pair: the two batches must have the same size")
        both = torch.cat([x_a, x_b], 0)
        self._prepare_weights()
        xs = self._scale_inputs(ops.btc_to_bct(both))
        main, side = self._fork()
        n_side = self._scales_on_side() if side is not None else 0
        ns = len(self.multi_scale_disc)
        scale = [None] * ns

        def run(i):
            disc = self.multi_scale_disc[i]
            if disc.norm == "spectral_norm":
                return ("pair", disc(xs[i][:n]), disc(xs[i][n:]))          # batch-major: the halves are contiguous views
            return ("both", disc(xs[i]), n)

        third = self._third_stream(main) if side is not None else None
        on_third = [ns - n_side - 1] if (third is not None and ns - n_side >= 2) else []
        if side is not None:
            self._handoff(side, [both] + xs[ns - n_side:])
            with torch.cuda.stream(side):
                pooled = self._pooled_forward(both)
                for i in range(ns - n_side, ns):
                    scale[i] = run(i)
        else:
            pooled = self._pooled_forward(both)
        if on_third:
            self._handoff(third, [xs[i] for i in on_third])
            with torch.cuda.stream(third):
                for i in on_third:
                    scale[i] = run(i)
        for i in range(ns - n_side):
            if i not in on_third:
                scale[i] = run(i)
        if side is not None:
            main.wait_stream(side)
        if on_third:
            main.wait_stream(third)
        return [("both", fm, n) for fm in pooled] + scale


class DiscriminatorSmall(_MultiDisc):
    """discriminator.py:122-155."""

    def __init__(self, num_emg_channels: int, num_multi_pool=5, num_multi_scale=3):
        super().__init__()
        self._build(num_emg_channels, num_multi_pool, num_multi_scale, DiscriminatorSmallerP, DiscriminatorSmallerS)


class Discriminator(_MultiDisc):
    """discriminator.py:158-191."""

    def __init__(self, num_emg_channels: int, num_multi_pool=5, num_multi_scale=3):
        super().__init__()
        self._build(num_emg_channels, num_multi_pool, num_multi_scale, DiscriminatorP, DiscriminatorS)


def init_emg_discriminators(cfg):
    """Factory -- discriminator.py:194-203."""
    num_emg_channels = cfg.data.num_emg_channels
    if cfg.model.discriminator_small:
        logging.info(f"Initializing small discriminators with {num_emg_channels} channels")
        return DiscriminatorSmall(num_emg_channels)
    logging.info(f"Initializing FULL discriminators with {num_emg_channels} channels")
    return Discriminator(num_emg_channels)
