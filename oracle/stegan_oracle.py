"""CPU oracle for the STE-GAN hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A reference-free, functional restatement (plain torch-CPU fp32 ops over flat state dicts)
of the algorithm of scheck-k/ste-gan's train/inference hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module -- and there only as the checker / the timed CPU baseline, never as something the
product path falls back to.  ``ste_gan_amd`` never imports it.

Parity pin: every function here is checked against outputs of the reference's own
modules (imported from /root/reference in the build container by
``tests/golden/make_golden.py``; fixtures committed under ``tests/golden/*.npz``) in
``tests/test_oracle_vs_golden.py``.  The reference ships no tests / golden vectors of
its own, so those fixtures are the pin.

All state dicts use the reference's ``state_dict()`` key layout.
Each function cites the reference file:line (paths relative to /root/reference) it follows.
Arithmetic below the module layer follows PyTorch's documented op semantics
(``F.conv1d``, ``torch._weight_norm``, ``spectral_norm``, ``F.pad(reflect)``, ``AvgPool1d``,
``Tensor.unfold``, ``F.pairwise_distance``, ``F.cross_entropy``, ``LayerNorm``, ``BatchNorm1d``
in eval mode, ``AdamW``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

HOPSIZE = 16                    # ste_gan/constants.py:45
SILENCE_PHONEME_INDEX = 47      # ste_gan/constants.py:166-173 ('sil' is the last of 48 entries)
PERIODS = (2, 3, 5, 7, 11)      # ste_gan/models/discriminator.py:127,163


# --------------------------------------------------------------------------------------
# bf16-operand emulation (tests of the build's bf16 compute mode; off by default)
# --------------------------------------------------------------------------------------
# The build's BASELINE-config-3 mode rounds the two operands of every matrix-core contraction to bf16 and keeps everything
# else in fp32.  Inside ``with bf16_operands():`` this oracle rounds AT THE SAME POINTS, so that the
# bf16 train step can be compared with it to ~1e-3 instead of the 3 % / 8 % that a comparison with the fp32 reference
# allows.  The rounding points, per contraction y = W (*) x:
#   forward          bf16(x) (*) bf16(w_raw), then the per-channel factor (weight norm g/||v||, 1/sigma, folded BatchNorm
#                    scale) and the bias in fp32 -- w_raw is the direction tensor v / weight_orig / the frozen weight;
#   input gradient   bf16(dy) (*) bf16(w_raw * factor)  (per-channel factors are folded into the gradient copy of the
#                    weights BEFORE rounding; the spectral norm's single factor is applied after, in fp32);
#   weight gradient  bf16(x) (*) bf16(dy) -> the gradient w.r.t. the EFFECTIVE weight, from which the fp32 weight-norm /
#                    spectral-norm chain rule continues (as the build's wn_backward / spectral_norm_bwd kernels do).
# Contractions the build leaves on exact fp32 kernels stay exact here: the single-output-channel convs, and any direction
# whose group has fewer than 8 channels on either side (weight gradients: fewer than 16).  With the flag off (the default,
# and always for the golden fixtures) every function below is the plain fp32 restatement.
BF16_OPERANDS = False
OPERAND_DTYPE = torch.bfloat16


class bf16_operands:
    """``with bf16_operands():`` bf16 operands (the build's "bf16" mode); ``with bf16_operands(torch.float16):`` the "f16"
    mode (libstegan_hip_f16.so) -- same rounding points, the other 2-byte type.  fp16 training also needs the reference's
    loss scaling: ``train_step(..., scaler=...)``."""

    def __init__(self, dtype: torch.dtype = torch.bfloat16):
        self.dtype = dtype

    def __enter__(self):
        global BF16_OPERANDS, OPERAND_DTYPE
        self._old = (BF16_OPERANDS, OPERAND_DTYPE)
        BF16_OPERANDS, OPERAND_DTYPE = True, self.dtype
        return self

    def __exit__(self, *exc):
        global BF16_OPERANDS, OPERAND_DTYPE
        BF16_OPERANDS, OPERAND_DTYPE = self._old
        return False


def _r(t: Tensor, dtype: Optional[torch.dtype] = None) -> Tensor:
    return t.to(dtype or OPERAND_DTYPE).to(t.dtype)


def _emu_flags(cin: int, cout: int, groups: int) -> Tuple[bool, bool, bool]:
    """(forward, input gradient, weight gradient) run with bf16 operands in the build (ops.conv1d_* dispatch rules)."""
    cig, cog = cin // groups, cout // groups
    if cout == 1 and groups == 1:
        return False, False, False                     # conv_skinny.hip: exact fp32 in all three directions
    mm = cig >= 8 and cog >= 8
    return mm, mm, (cig >= 16 and cog >= 16)


class _ContractEmu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_eff, bias, w_raw, scale, conv, kw, uniform, flags):
        fwd_b = flags[0]
        ctx.dtype = OPERAND_DTYPE                  # the backward pass may run after the ``with bf16_operands(...)`` block
        y = conv(_r(x) if fwd_b else x, _r(w_raw) if fwd_b else w_raw, None, **kw)
        shape = (1, -1) + (1,) * (y.dim() - 2)
        if scale is not None:
            y = y * (scale if uniform else scale.view(shape))
        if bias is not None:
            y = y + bias.view(shape)
        ctx.conv, ctx.kw, ctx.uniform, ctx.flags = conv, kw, uniform, flags
        ctx.save_for_backward(x, w_raw, scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w_raw, scale = ctx.saved_tensors
        conv, kw, uniform, (_, bwd_b, wg_b) = ctx.conv, ctx.kw, ctx.uniform, ctx.flags
        _r = lambda t: globals()["_r"](t, ctx.dtype)
        dx = dw = db = None
        wshape = (-1,) + (1,) * (w_raw.dim() - 1)
        with torch.enable_grad():
            if ctx.needs_input_grad[0]:
                wb = w_raw if (scale is None or uniform) else w_raw * scale.view(wshape)
                xe = torch.zeros_like(x, requires_grad=True)
                ye = conv(xe, _r(wb) if bwd_b else wb, None, **kw)
                (dx,) = torch.autograd.grad(ye, xe, _r(dy) if bwd_b else dy)
                if scale is not None and uniform:
                    dx = dx * scale
            if ctx.needs_input_grad[1]:
                w0 = torch.zeros_like(w_raw, requires_grad=True)
                y0 = conv(_r(x) if wg_b else x, w0, None, **kw)
                (dw,) = torch.autograd.grad(y0, w0, _r(dy) if wg_b else dy)
        if ctx.needs_input_grad[2]:
            db = dy.sum([d for d in range(dy.dim()) if d != 1])
        return dx, dw, db, None, None, None, None, None, None


def _contract(x, w_eff, bias, w_raw, scale, conv, uniform=False, **kw):
    """One conv / linear contraction: plain fp32 ``conv(x, w_eff, bias)`` -- or, under bf16_operands(), the emulation.
    ``w_eff`` (the effective weight, with autograd history back to the parameters; None for frozen weights),
    ``w_raw`` + ``scale``: its factorisation as the build stores it."""
    if not BF16_OPERANDS:
        return conv(x, w_eff if w_eff is not None else (w_raw if scale is None else w_raw * scale.view((-1,) + (1,) * (w_raw.dim() - 1))),
                    bias, **kw)
    groups = kw.get("groups", 1)
    flags = _emu_flags(w_raw.shape[1] * groups, w_raw.shape[0], groups)
    return _ContractEmu.apply(x, w_eff, bias, w_raw.detach(), scale.detach() if scale is not None else None, conv, kw,
                              uniform, flags)


# --------------------------------------------------------------------------------------
# weight re-parametrisations
# --------------------------------------------------------------------------------------
def weight_norm_weight(v: Tensor, g: Tensor) -> Tensor:
    """w = g * v / ||v||, norm over all dims but 0, no eps.

    ste_gan/layers/conv.py:16-17,89-101 (old-style ``torch.nn.utils.weight_norm``, dim=0).
    """
    norm = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return v * (g / norm)


def spectral_norm_weight(w_orig: Tensor, u: Tensor, v: Tensor, training: bool = True,
                         eps: float = 1e-12) -> Tuple[Tensor, Tensor, Tensor]:
    """One power iteration (training mode) then w = w_orig / (u^T W v).

    ste_gan/layers/conv.py:89-101 with norm="spectral_norm" (old-style
    ``torch.nn.utils.spectral_norm``: n_power_iterations=1, eps=1e-12, dim=0).  Returns
    (w, u_new, v_new); u/v are treated as constants for autograd, as torch does.
    """
    w_mat = w_orig.flatten(1)
    if training:
        with torch.no_grad():
            v = F.normalize(torch.mv(w_mat.t(), u), dim=0, eps=eps)
            u = F.normalize(torch.mv(w_mat, v), dim=0, eps=eps)
    sigma = torch.dot(u, torch.mv(w_mat, v))
    return w_orig / sigma, u, v


def _wn_scale(v: Tensor, g: Tensor) -> Tensor:
    return (g.flatten() / v.flatten(1).norm(dim=1))


def _wn_conv1d(sd: SD, prefix: str, x: Tensor, **kw) -> Tensor:
    v, g = sd[prefix + ".weight_v"], sd[prefix + ".weight_g"]
    w = weight_norm_weight(v, g)
    if not BF16_OPERANDS:
        return F.conv1d(x, w, sd[prefix + ".bias"], **kw)
    return _contract(x, w, sd[prefix + ".bias"], v, _wn_scale(v, g), F.conv1d, **kw)


# --------------------------------------------------------------------------------------
# generator  (ste_gan/models/generator.py:78-162, ste_gan/layers/conv.py:29-84)
# --------------------------------------------------------------------------------------
def gblock_forward(sd: SD, prefix: str, x: Tensor, upsample: int) -> Tensor:
    """GBlock.forward -- ste_gan/layers/conv.py:29-84.

    conv1 = [ReLU, (Upsample), WNConv k3 d1, ReLU, WNConv k3 d3]; res1 = [(Upsample), WNConv k1];
    conv2 = [ReLU, WNConv k3 d9, ReLU, WNConv k3 d27]; out = y + conv2(y), y = conv1(x) + res1(x).
    nn.Sequential indices shift by one when the Upsample module is present.
    """
    o = 1 if upsample > 1 else 0
    h = F.relu(x)
    r = x
    if upsample > 1:
        h = F.interpolate(h, scale_factor=float(upsample), mode="nearest")
        r = F.interpolate(r, scale_factor=float(upsample), mode="nearest")
    h = _wn_conv1d(sd, f"{prefix}.conv1.{1 + o}", h, padding=1)
    h = _wn_conv1d(sd, f"{prefix}.conv1.{3 + o}", F.relu(h), dilation=3, padding=3)
    r = _wn_conv1d(sd, f"{prefix}.res1.{o}", r)
    y = h + r
    h = _wn_conv1d(sd, f"{prefix}.conv2.1", F.relu(y), dilation=9, padding=9)
    h = _wn_conv1d(sd, f"{prefix}.conv2.3", F.relu(h), dilation=27, padding=27)
    return y + h


def generator_upsamples(speech_feature_type: str) -> Tuple[int, ...]:
    """Per-GBlock upsample factors -- ste_gan/models/generator.py:116-131."""
    last = 2 if speech_feature_type == "SPEECH_UNITS" else 1
    return (1, 1, 2, 2, 2, last, 1, 1)


def generator_forward(sd: SD, speech_feats: Tensor, session_ids: Tensor,
                      speaking_mode_ids: Optional[Tensor] = None,
                      speech_feature_type: str = "SPEECH_UNITS") -> Tensor:
    """EMGGeneratorGanTTS.forward -- ste_gan/models/generator.py:140-162.

    (B,T_feat,F) float, (B,) int64 -> (B, T_emg, C_emg) in (-1,1).  Session embedding is
    gathered, broadcast over time and concatenated on the feature axis (143-147); the
    speaking-mode embedding is off by default (generator.py:89,101-105).
    """
    x = speech_feats
    T = x.shape[1]
    emb = sd["session_embeddings.weight"][session_ids]            # (B, 64)
    x = torch.cat((x, emb[:, None, :].expand(-1, T, -1)), dim=-1)
    if "speaking_mode_embeddings.weight" in sd:
        emb2 = sd["speaking_mode_embeddings.weight"][speaking_mode_ids]
        x = torch.cat((x, emb2[:, None, :].expand(-1, T, -1)), dim=-1)
    x = x.transpose(1, 2)                                          # b t c -> b c t
    x = _wn_conv1d(sd, "gblocks.0", x)
    for i, up in enumerate(generator_upsamples(speech_feature_type)):
        x = gblock_forward(sd, f"gblocks.{i + 1}", x, up)
    x = _wn_conv1d(sd, "last_conv.1", F.relu(x), padding=1)
    return torch.tanh(x.transpose(1, 2))


# --------------------------------------------------------------------------------------
# discriminators  (ste_gan/models/discriminator.py:19-203)
# --------------------------------------------------------------------------------------
MPD_SMALL = ((3, 1, 2), (3, 3, 2), (3, 3, 2))          # (k, stride, pad) discriminator.py:76-80
MPD_FULL = ((5, 3, 2),) * 4 + ((5, 1, 2),)             # discriminator.py:25-30
MSD_SMALL = ((1, 7, 1), (2, 18, 4), (2, 18, 16), (1, 2, 1))            # (stride, pad, groups) :54-58
MSD_FULL = ((1, 7, 1), (2, 20, 4), (2, 20, 16), (4, 20, 16), (4, 20, 16), (1, 20, 16), (1, 2, 1))  # :103-110


def mpd_forward(sd: SD, prefix: str, x: Tensor, period: int, small: bool) -> List[Tensor]:
    """DiscriminatorSmallerP / DiscriminatorP .forward -- discriminator.py:84-93 / 34-43.

    Right reflect-pad by ``period - T % period`` (a full period when divisible), fold to
    (B,C,T'/p,p), (k,1) convs + leaky_relu(0.1); weight-norm on every layer.
    """
    T = x.shape[-1]
    x = F.pad(x, (0, period - T % period), "reflect")
    x = x.view(x.shape[0], x.shape[1], x.shape[2] // period, period)
    fmaps = []
    for j, (k, s, p) in enumerate(MPD_SMALL if small else MPD_FULL):
        pre = f"{prefix}.layers.{j}"
        v, g = sd[pre + ".weight_v"], sd[pre + ".weight_g"]
        w = weight_norm_weight(v, g)
        x = F.leaky_relu(_contract(x, w, sd[pre + ".bias"], v, _wn_scale(v, g), F.conv2d, stride=(s, 1), padding=(p, 0)), 0.1)
        fmaps.append(x)
    pre = f"{prefix}.output"
    v, g = sd[pre + ".weight_v"], sd[pre + ".weight_g"]
    w = weight_norm_weight(v, g)
    fmaps.append(_contract(x, w, sd[pre + ".bias"], v, _wn_scale(v, g), F.conv2d, padding=(1, 0)))
    return fmaps


def msd_forward(sd: SD, prefix: str, x: Tensor, spectral: bool, small: bool,
                training: bool = True, sn_state: Optional[SD] = None) -> List[Tensor]:
    """DiscriminatorSmallerS / DiscriminatorS .forward -- discriminator.py:60-67 / 113-119.

    Body layers use spectral norm on scale 0 (discriminator.py:136,172), weight norm
    otherwise; the output conv always uses weight norm (:59,111).  ``sn_state`` receives the
    power-iterated u/v (the reference mutates its buffers in place on every training forward).
    """
    fmaps = []
    for j, (s, p, g) in enumerate(MSD_SMALL if small else MSD_FULL):
        pre = f"{prefix}.layers.{j}"
        if spectral:
            w, u, v = spectral_norm_weight(sd[pre + ".weight_orig"], sd[pre + ".weight_u"],
                                           sd[pre + ".weight_v"], training)
            if sn_state is not None:
                sn_state[pre + ".weight_u"], sn_state[pre + ".weight_v"] = u, v
            w_raw = sd[pre + ".weight_orig"]
            with torch.no_grad():
                scale = 1.0 / torch.dot(u, torch.mv(w_raw.flatten(1), v))           # 1 / sigma, one factor for the layer
            y = _contract(x, w, sd[pre + ".bias"], w_raw, scale, F.conv1d, uniform=True, stride=s, padding=p, groups=g)
        else:
            vv, gg = sd[pre + ".weight_v"], sd[pre + ".weight_g"]
            w = weight_norm_weight(vv, gg)
            y = _contract(x, w, sd[pre + ".bias"], vv, _wn_scale(vv, gg), F.conv1d, stride=s, padding=p, groups=g)
        x = F.leaky_relu(y, 0.1)
        fmaps.append(x)
    fmaps.append(_wn_conv1d(sd, f"{prefix}.output", x, padding=1))
    return fmaps


def discriminator_forward(sd: SD, x_btc: Tensor, small: bool = True, training: bool = True,
                          sn_state: Optional[SD] = None) -> List[List[Tensor]]:
    """DiscriminatorSmall / Discriminator .forward -- discriminator.py:144-155 / 180-191.

    (B,T,C) -> list over [MPD p=2,3,5,7,11, MSD 0,1,2] of fmap lists (last = logits).
    MSD scale i sees AvgPool1d(4,2,1)^i (x) (zero pad, count_include_pad).
    """
    x = x_btc.transpose(1, 2)
    results = []
    for i, p in enumerate(PERIODS):
        results.append(mpd_forward(sd, f"multi_pooled_disc.{i}", x, p, small))
    for i in range(3):
        results.append(msd_forward(sd, f"multi_scale_disc.{i}", x, i == 0, small, training, sn_state))
        x = F.avg_pool1d(x, kernel_size=4, stride=2, padding=1)
    return results


# --------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------
def lsgan_loss(d_out: List[List[Tensor]], target: float) -> Tensor:
    """sum_i mse(D_i[-1], target) -- ste_gan/train.py:192-196, 209-211."""
    loss = 0.0
    for scale in d_out:
        loss = loss + F.mse_loss(scale[-1], torch.full_like(scale[-1], target))
    return loss


def feature_matching_loss(d_fake: List[List[Tensor]], d_real: List[List[Tensor]]) -> Tensor:
    """sum_i sum_{j<last} l1(D_fake[i][j], D_real[i][j].detach()) -- ste_gan/train.py:257-262."""
    loss = 0.0
    for i in range(len(d_fake)):
        for j in range(len(d_fake[i]) - 1):
            loss = loss + F.l1_loss(d_fake[i][j], d_real[i][j].detach())
    return loss


def average_filter(x_bct: Tensor, window: int = 9) -> Tensor:
    """AverageFilter.forward -- ste_gan/layers/average_filter.py:22-28 (reflect pad k//2, AvgPool1d(k,1))."""
    return F.avg_pool1d(F.pad(x_bct, (window // 2, window // 2), mode="reflect"), window, 1)


def td_features(x_btc: Tensor, win: int, shift: int) -> Tensor:
    """TimeDomainFeatureLoss.calculate_time_domain_features -- losses/time_domain_loss.py:35-68.

    (B,T,C) -> (B, n_frames, C, 4): [mean(low), sum(low^2), sum(rect^2), mean(rect)] per frame,
    low = avg9(avg9(x)), rect = |x - low|, frames over a reflect-padded (win//2) time axis.
    """
    low = average_filter(average_filter(x_btc.transpose(1, 2))).transpose(1, 2)
    rect = (x_btc - low).abs()

    def frames(z):
        z = F.pad(z, (0, 0, win // 2, win // 2), mode="reflect")
        return z.unfold(1, win, shift)

    return torch.stack([frames(low).mean(-1), (frames(low) ** 2).sum(-1),
                        (frames(rect) ** 2).sum(-1), frames(rect).mean(-1)], dim=-1)


TD_SCALES = ((20, 8), (51, 13), (80, 16))      # losses/time_domain_loss.py:87-94


def multi_td_loss(x_real: Tensor, x_gen: Tensor) -> Tensor:
    """MultiTimeDomainFeatureLoss.forward -- losses/time_domain_loss.py:96-107."""
    loss = 0.0
    for win, shift in TD_SCALES:
        loss = loss + F.l1_loss(td_features(x_gen, win, shift), td_features(x_real, win, shift).detach())
    return loss


def speech_unit_loss(su_target: Tensor, su_pred: Tensor) -> Tensor:
    """mean_rows ||t - p + 1e-6||_2 -- losses/emg_encoder_loss.py:63-67 (F.pairwise_distance eps)."""
    return F.pairwise_distance(su_target.flatten(0, 1), su_pred.flatten(0, 1)).mean()


def phoneme_loss(ph_logits: Tensor, ph_target: Tensor) -> Tensor:
    """Cross entropy over 48 classes, mean over B*T -- losses/emg_encoder_loss.py:80-83."""
    return F.cross_entropy(ph_logits.transpose(1, 2), ph_target)


def phoneme_counters(ph_logits: Tensor, ph_target: Tensor) -> Dict[str, int]:
    """EMGEncoderLossOutput counter properties -- losses/emg_encoder_loss.py:27-53."""
    pred = ph_logits.argmax(-1).flatten()
    tgt = ph_target.flatten()
    correct = pred == tgt
    return {
        "num_phones": int(tgt.numel()),
        "num_silence_phones": int((tgt == SILENCE_PHONEME_INDEX).sum()),
        "num_correct_phones": int(correct.sum()),
        "num_correct_phones_no_silence": int((correct & (tgt != SILENCE_PHONEME_INDEX)).sum()),
    }


# --------------------------------------------------------------------------------------
# frozen EMG encoder (eval mode)  -- models/emg_encoder.py:36-88
# --------------------------------------------------------------------------------------
def _bn_eval(sd: SD, pre: str, x: Tensor, eps: float = 1e-5) -> Tensor:
    return F.batch_norm(x, sd[pre + ".running_mean"], sd[pre + ".running_var"],
                        sd[pre + ".weight"], sd[pre + ".bias"], False, 0.0, eps)


def _frozen_conv_bn(sd: SD, conv: str, bn: str, x: Tensor, eps: float = 1e-5, **kw) -> Tensor:
    """conv + eval-mode BatchNorm as the build runs it: the BatchNorm scale is a per-channel factor behind the contraction
    (folded into the weights' gradient copy), conv bias and shift are one folded bias."""
    s_ = sd[bn + ".weight"] / torch.sqrt(sd[bn + ".running_var"] + eps)
    b_ = (sd[conv + ".bias"] - sd[bn + ".running_mean"]) * s_ + sd[bn + ".bias"]
    return _contract(x, None, b_, sd[conv + ".weight"], s_, F.conv1d, **kw)


def _frozen_linear(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """F.linear over the last axis as the build's K = 1 conv over (B, C, T)."""
    if not BF16_OPERANDS:
        return F.linear(x, w, b)
    lead = x.shape[:-1]
    y = _contract(x.reshape(-1, x.shape[-1]).t().unsqueeze(0), None, b, w.unsqueeze(-1), None, F.conv1d)
    return y.squeeze(0).t().reshape(*lead, w.shape[0])


def encoder_resblock(sd: SD, pre: str, x: Tensor, stride: int = 2) -> Tensor:
    """ResBlock.forward (eval-mode BatchNorm) -- ste_gan/layers/conv.py:106-133."""
    if BF16_OPERANDS:
        h = F.relu(_frozen_conv_bn(sd, pre + ".conv1", pre + ".bn1", x, stride=stride, padding=1))
        h = _frozen_conv_bn(sd, pre + ".conv2", pre + ".bn2", h, padding=1)
        r = (_frozen_conv_bn(sd, pre + ".residual_path", pre + ".res_norm", x, stride=stride)
             if pre + ".residual_path.weight" in sd else x)
        return F.relu(h + r)
    h = F.relu(_bn_eval(sd, pre + ".bn1", F.conv1d(x, sd[pre + ".conv1.weight"], sd[pre + ".conv1.bias"],
                                                  stride=stride, padding=1)))
    h = _bn_eval(sd, pre + ".bn2", F.conv1d(h, sd[pre + ".conv2.weight"], sd[pre + ".conv2.bias"], padding=1))
    if pre + ".residual_path.weight" in sd:
        r = _bn_eval(sd, pre + ".res_norm", F.conv1d(x, sd[pre + ".residual_path.weight"],
                                                    sd[pre + ".residual_path.bias"], stride=stride))
    else:
        r = x
    return F.relu(h + r)


def relative_position_logits(q_bhtd: Tensor, emb: Tensor, max_rel: int = 100) -> Tensor:
    """LearnedRelativePositionalEmbedding.forward, unmasked, per-head embeddings.

    ste_gan/layers/transformer.py:168-306.  After the pad/reshape "skew" (271-306) the result is
    logits[b,h,i,j] = q[b,h,i,:] . emb[h, max_rel-1 + (j-i), :]  for |j-i| <= max_rel-1, and
    -1e8 beyond (zero-padded embeddings, 212-228, then the mask of 256-263).
    """
    B, H, L, D = q_bhtd.shape
    e = emb[..., 0]                                               # (H, 2*max-1, D)
    rel = torch.arange(L)[None, :] - torch.arange(L)[:, None]     # j - i
    inside = rel.abs() <= max_rel - 1
    idx = (rel + max_rel - 1).clamp(0, 2 * max_rel - 2)
    e_ij = e[:, idx]                                              # (H, L, L, D)
    logits = torch.einsum("bhid,hijd->bhij", q_bhtd, e_ij)
    return torch.where(inside[None, None], logits, torch.full_like(logits, -1e8))


def attention_forward(sd: SD, pre: str, x_tbf: Tensor, max_rel: int = 100) -> Tensor:
    """MultiHeadAttention.forward (eval: dropout off) -- ste_gan/layers/transformer.py:88-113."""
    wq, wk, wv, wo = (sd[f"{pre}.w_{n}"] for n in "qkvo")
    d = wq.shape[-1]
    if BF16_OPERANDS:
        # the build's projections are K = 1 convs over the packed (H A, F) weights (layers/transformer.py packed())
        H, Fd, A = wq.shape
        T_, B_ = x_tbf.shape[0], x_tbf.shape[1]
        proj = lambda w: _frozen_linear(x_tbf, w.permute(0, 2, 1).reshape(H * A, Fd), None).view(T_, B_, H, A).permute(1, 2, 0, 3)
        q, k, v = proj(wq), proj(wk), proj(wv)
        logits = torch.einsum("bhqa,bhka->bhqk", q, k) / (d ** 0.5)
        logits = logits + relative_position_logits(q, sd[f"{pre}.relative_positional.embeddings"], max_rel)
        probs = F.softmax(logits, dim=-1)
        o = torch.einsum("bhqk,bhka->bhqa", probs, v)                       # (B, H, T, A)
        o = o.permute(2, 0, 1, 3).reshape(T_, B_, H * A)
        return _frozen_linear(o, wo.reshape(H * A, Fd).t(), None)
    q = torch.einsum("tbf,hfa->bhta", x_tbf, wq)
    k = torch.einsum("tbf,hfa->bhta", x_tbf, wk)
    v = torch.einsum("tbf,hfa->bhta", x_tbf, wv)
    logits = torch.einsum("bhqa,bhka->bhqk", q, k) / (d ** 0.5)
    logits = logits + relative_position_logits(q, sd[f"{pre}.relative_positional.embeddings"], max_rel)
    probs = F.softmax(logits, dim=-1)
    o = torch.einsum("bhqk,bhka->bhqa", probs, v)
    return torch.einsum("bhta,haf->tbf", o, wo)


def transformer_layer(sd: SD, pre: str, src: Tensor) -> Tensor:
    """TransformerEncoderLayer.forward (post-LN, eval) -- ste_gan/layers/transformer.py:44-61."""
    D = src.shape[-1]
    src = src + attention_forward(sd, pre + ".self_attn", src)
    src = F.layer_norm(src, (D,), sd[pre + ".norm1.weight"], sd[pre + ".norm1.bias"], 1e-5)
    h = _frozen_linear(F.relu(_frozen_linear(src, sd[pre + ".linear1.weight"], sd[pre + ".linear1.bias"])),
                       sd[pre + ".linear2.weight"], sd[pre + ".linear2.bias"])
    src = src + h
    return F.layer_norm(src, (D,), sd[pre + ".norm2.weight"], sd[pre + ".norm2.bias"], 1e-5)


def encoder_forward(sd: SD, emg_btc: Tensor) -> Tuple[Tensor, Tensor]:
    """EMGEncoderTransformer.forward in eval mode -- models/emg_encoder.py:70-88.

    (B,T_emg,8) -> ((B,T_emg/16,256) speech units, (B,T_emg/16,48) phoneme logits).  The random
    left-shift (71-75) only runs in training mode; the GAN keeps the encoder in eval (train.py:59).
    """
    x = emg_btc.transpose(1, 2)
    i = 0
    while f"conv_blocks.{i}.conv1.weight" in sd:
        x = encoder_resblock(sd, f"conv_blocks.{i}", x)
        i += 1
    x = _frozen_linear(x.transpose(1, 2), sd["w_raw_in.weight"], sd["w_raw_in.bias"])
    x = x.transpose(0, 1)
    i = 0
    while f"transformer.layers.{i}.linear1.weight" in sd:
        x = transformer_layer(sd, f"transformer.layers.{i}", x)
        i += 1
    x = x.transpose(0, 1)
    return (_frozen_linear(x, sd["w_out.weight"], sd["w_out.bias"]),
            _frozen_linear(x, sd["w_aux.weight"], sd["w_aux.bias"]))


# --------------------------------------------------------------------------------------
# optimiser -- constants.py:57 (AdamW lr 2e-4 betas (.8,.99); torch defaults eps 1e-8, wd 1e-2)
# --------------------------------------------------------------------------------------
@dataclass
class AdamWState:
    step: int
    m: SD
    v: SD


def adamw_init(params: SD) -> AdamWState:
    return AdamWState(0, {k: torch.zeros_like(p) for k, p in params.items()},
                      {k: torch.zeros_like(p) for k, p in params.items()})


@torch.no_grad()
def adamw_step(params: SD, grads: SD, st: AdamWState, lr: float = 2e-4, b1: float = 0.8,
               b2: float = 0.99, eps: float = 1e-8, wd: float = 1e-2) -> None:
    """In-place decoupled-weight-decay Adam (torch.optim.AdamW single-tensor semantics)."""
    st.step += 1
    bc1 = 1.0 - b1 ** st.step
    bc2 = 1.0 - b2 ** st.step
    for k, p in params.items():
        g = grads.get(k)
        if g is None:
            continue
        p.mul_(1.0 - lr * wd)
        st.m[k].lerp_(g, 1.0 - b1)
        st.v[k].mul_(b2).addcmul_(g, g, value=1.0 - b2)
        denom = (st.v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(st.m[k], denom, value=-lr / bc1)


# --------------------------------------------------------------------------------------
# train step -- ste_gan/train.py:165-268 (fp32 path: on CPU autocast + GradScaler are no-ops)
# --------------------------------------------------------------------------------------
D_BUFFER_SUFFIXES = (".weight_u",)


def is_sn_buffer(sd: SD, key: str) -> bool:
    """spectral-norm u / v are buffers; note ``weight_v`` is a *parameter* under weight norm."""
    if key.endswith(".weight_u"):
        return True
    return key.endswith(".weight_v") and (key[:-len("weight_v")] + "weight_orig") in sd


@dataclass
class LossWeights:          # configs/ste_gan_base_gantts.yaml:17-41
    multi_td: float = 15.0
    feat_match: float = 7.0
    speech_unit: float = 1.0
    phoneme: float = 1.0


def train_step(g_sd: SD, d_sd: SD, enc_sd: Optional[SD], opt_g: AdamWState, opt_d: AdamWState,
               batch: Dict[str, Tensor], small_d: bool = True,
               speech_feature_type: str = "SPEECH_UNITS", weights: LossWeights = LossWeights(),
               lr: float = 2e-4, keep_grads: bool = False, grad_sync=None,
               scaler: Optional[Dict[str, float]] = None) -> Dict[str, object]:
    """One iteration of the hot loop, in place on g_sd / d_sd / optimiser state.

    Follows ste_gan/train.py:165-268 exactly, including the ordering facts of SURVEY.md 3.3:
    optD steps before the generator-side D forwards; D(x_real) is recomputed with the updated
    D; the spectral-norm power iteration runs on each of the 4 D forwards; D grads produced by
    loss_G.backward() are discarded.  ``batch`` keys: su (B,T,256) [SU targets], feats
    (generator input; == su for Soft-SU), x_real (B,16T,8), sess (B,), phonemes (B,T).
    ``grad_sync(dict of grads) -> dict`` (optional) stands in for the data-parallel gradient average between
    backward and optimiser step (the reference itself is single-process).
    ``scaler`` (optional): torch.cuda.amp.GradScaler as train.py:151,198-199,266-268 drives it, restated on a dict
    {"scale", "growth_tracker"} (+ optional "growth_factor" 2, "backoff_factor" 0.5, "growth_interval" 2000), updated in
    place: both losses are multiplied by the scale before backward, each optimiser step unscales its gradients and is
    SKIPPED when one of them is not finite, and one update per iteration halves the scale after a skip or doubles it after
    ``growth_interval`` clean iterations.  out["skipped"] = (D step skipped, G step skipped).
    """
    S = float(scaler["scale"]) if scaler is not None else 1.0
    skipped = [False, False]

    def unscale(grads, which):
        if scaler is None:
            return grads
        bad = any(not bool(torch.isfinite(g).all()) for g in grads.values())
        skipped[which] = bad
        return None if bad else {k: g / S for k, g in grads.items()}
    out: Dict[str, object] = {}
    g_params = {k: v.detach().requires_grad_(True) for k, v in g_sd.items()}
    d_keys = [k for k in d_sd if not is_sn_buffer(d_sd, k)]

    def d_leaf():
        p = {k: (d_sd[k].detach().requires_grad_(True) if k in d_keys else d_sd[k]) for k in d_sd}
        return p

    x_real = batch["x_real"]
    x_pred = generator_forward(g_params, batch["feats"], batch["sess"], None, speech_feature_type)   # :182

    # ---- discriminator step (train.py:189-199)
    d_params = d_leaf()
    sn = {}
    d_fake_det = discriminator_forward(d_params, x_pred.detach(), small_d, True, sn)
    d_params.update(sn)
    d_real = discriminator_forward(d_params, x_real, small_d, True, sn)
    d_params.update(sn)
    loss_d = lsgan_loss(d_fake_det, 0.0) + lsgan_loss(d_real, 1.0)
    d_grads = dict(zip(d_keys, torch.autograd.grad(loss_d * S, [d_params[k] for k in d_keys])))
    with torch.no_grad():
        for k in sn:
            d_sd[k] = sn[k].detach().clone()
    trainable = {k: d_sd[k] for k in d_keys}
    if grad_sync is not None:
        d_grads = grad_sync(d_grads)
    d_grads = unscale(d_grads, 0)
    if d_grads is not None:
        adamw_step(trainable, d_grads, opt_d, lr=lr)
    out["loss_d"] = float(loss_d)

    # ---- generator step (train.py:204-268)
    d_params = {k: d_sd[k] for k in d_sd}      # updated weights, no grad needed for D here
    sn = {}
    d_fake = discriminator_forward(d_params, x_pred, small_d, True, sn)
    d_params.update(sn)
    d_real = discriminator_forward(d_params, x_real, small_d, True, sn)
    with torch.no_grad():
        for k in sn:
            d_sd[k] = sn[k].detach().clone()
    loss_adv = lsgan_loss(d_fake, 1.0)
    td = multi_td_loss(x_real, x_pred)
    loss_g = loss_adv + weights.multi_td * td
    out["loss_adv_g"], out["loss_td"] = float(loss_adv), float(td)
    if enc_sd is not None:
        su_pred, ph_pred = encoder_forward(enc_sd, x_pred)
        su_l = speech_unit_loss(batch["su"], su_pred)
        ph_l = phoneme_loss(ph_pred, batch["phonemes"])
        loss_g = loss_g + weights.speech_unit * su_l + weights.phoneme * ph_l
        out["loss_su"], out["loss_ph"] = float(su_l), float(ph_l)
        out.update(phoneme_counters(ph_pred.detach(), batch["phonemes"]))
    fm = feature_matching_loss(d_fake, d_real)
    loss_g = loss_g + weights.feat_match * fm
    out["loss_fm"], out["loss_g"] = float(fm), float(loss_g)
    g_keys = list(g_params)
    g_grads = dict(zip(g_keys, torch.autograd.grad(loss_g * S, [g_params[k] for k in g_keys])))
    if grad_sync is not None:
        g_grads = grad_sync(g_grads)
    g_grads = unscale(g_grads, 1)
    if g_grads is not None:
        adamw_step(g_sd, g_grads, opt_g, lr=lr)
    if scaler is not None:                                     # scaler.update(), train.py:268
        if skipped[0] or skipped[1]:
            scaler["scale"] = S * float(scaler.get("backoff_factor", 0.5))
            scaler["growth_tracker"] = 0
        else:
            scaler["growth_tracker"] = int(scaler.get("growth_tracker", 0)) + 1
            if scaler["growth_tracker"] == int(scaler.get("growth_interval", 2000)):
                scaler["scale"] = S * float(scaler.get("growth_factor", 2.0))
                scaler["growth_tracker"] = 0
    out["skipped"] = tuple(skipped)
    if keep_grads:
        out["g_grads"], out["d_grads"] = g_grads, d_grads
        out["x_pred"] = x_pred.detach()
    return out
