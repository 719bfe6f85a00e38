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


def feature_matching_loss(d_fake, d_real, weight: float = 1.0, unit_grad: bool = False, fold: bool = False):
    """weight * sum_i sum_{j<last} l1(D_fake[i][j], D_real[i][j].detach()) -- train.py:257-262.
    ``fold``: see ``ops.multi_pair_loss`` (the feature maps' producers add the gradients themselves)."""
    fake = [d_fake[i][j] for i in range(len(d_fake)) for j in range(len(d_fake[i]) - 1)]
    real = [d_real[i][j] for i in range(len(d_fake)) for j in range(len(d_fake[i]) - 1)]
    return ops.multi_pair_loss(fake, real, 0, weight, unit_grad=unit_grad, fold=fold)
