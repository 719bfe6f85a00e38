"""Adversarial and feature-matching losses of the train step.

The reference computes these inline (``ste_gan/train.py:192-196``, ``:209-211``, ``:257-262``); they live
here as functions so the train loop and the tests share them.  Each term is one fused
value+gradient kernel launch.
"""
from __future__ import annotations

from .. import ops


def lsgan_loss(d_out, target: float):
    """sum over the 8 sub-discriminators of mse(logits, target) -- train.py:192-196 (0 / 1), :209-211 (1)."""
    loss = 0
    for scale in d_out:
        loss = loss + ops.mse_to_const(scale[-1], target)
    return loss


def feature_matching_loss(d_fake, d_real):
    """sum_i sum_{j<last} l1(D_fake[i][j], D_real[i][j].detach()) -- train.py:257-262."""
    loss = 0
    for i in range(len(d_fake)):
        for j in range(len(d_fake[i]) - 1):
            loss = loss + ops.l1_loss(d_fake[i][j], d_real[i][j])
    return loss
