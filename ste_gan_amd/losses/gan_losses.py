"""Adversarial and feature-matching losses of the train step.

The reference computes these inline (``ste_gan/train.py:192-196``, ``:209-211``, ``:257-262``); they live
here as functions so the train loop and the tests share them.  Each term is one fused
value+gradient kernel launch.
"""
from __future__ import annotations

from .. import ops


def lsgan_loss(d_out, target: float, weight: float = 1.0, unit_grad: bool = False):
    """weight * sum over the 8 sub-discriminators of mse(logits, target) -- train.py:192-196 (0 / 1), :209-211 (1).
    One autograd node for the whole sum (``ops.multi_pair_loss``); see there for ``unit_grad``."""
    return ops.multi_pair_loss([scale[-1] for scale in d_out], None, 1, weight, target=float(target), unit_grad=unit_grad)


def lsgan_pair_loss(entries, target_a: float = 0.0, target_b: float = 1.0, unit_grad: bool = False):
    """sum over the sub-discriminators of mse(logits_a, target_a) + mse(logits_b, target_b) -- the discriminator loss,
    train.py:192-196 -- from ``_MultiDisc.forward_pair_logits``: ("both", logits of the concatenated batch, n) entries
    are compared against target_a on the first half and target_b on the second in ONE term of weight 2 (two equal halves:
    the two means add up to twice the mean over the whole), ("pair", logits_a, logits_b) entries give two plain terms."""
    ts, ws, t1, t2, sp = [], [], [], [], []
    for e in entries:
        if e[0] == "both":
            lg, n = e[1], e[2]
            if lg.shape[0] != 2 * n:
                raise ValueError("lsgan_pair_loss: batched logits must hold two equal halves")
            ts.append(lg); ws.append(2.0); t1.append(target_a); t2.append(target_b); sp.append(lg.numel() // 2)
        else:
            ts += [e[1], e[2]]; ws += [1.0, 1.0]; t1 += [target_a, target_b]; t2 += [target_a, target_b]; sp += [0, 0]
    return ops.multi_pair_loss_const(ts, 1, ws, t1, t2, sp, unit_grad=unit_grad)


def feature_matching_loss(d_fake, d_real, weight: float = 1.0, unit_grad: bool = False, fold: bool = False):
    """weight * sum_i sum_{j<last} l1(D_fake[i][j], D_real[i][j].detach()) -- train.py:257-262.
    ``fold``: see ``ops.multi_pair_loss`` (the feature maps' producers add the gradients themselves)."""
    fake = [d_fake[i][j] for i in range(len(d_fake)) for j in range(len(d_fake[i]) - 1)]
    real = [d_real[i][j] for i in range(len(d_fake)) for j in range(len(d_fake[i]) - 1)]
    return ops.multi_pair_loss(fake, real, 0, weight, unit_grad=unit_grad, fold=fold)
