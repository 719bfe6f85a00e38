"""Time-domain feature losses, host side -- mirrors ``ste_gan/losses/time_domain_loss.py``.

``TimeDomainFeatureLoss`` (:13-73) and ``MultiTimeDomainFeatureLoss`` (:76-107) keep their
constructor arguments (every option: window / shift, ``apply_padding_windowing``, any odd
``average_filter_window_size``) and ``forward`` / ``time_domain_loss`` / ``calculate_time_domain_features`` results;
the whole loss chain (two reflect padded average filters, rectification, optionally reflect padded framing, the four
frame features for both signals, the L1 means of up to three scales and d loss / d x_generated) is ONE kernel launch
with every intermediate in LDS -- or, for rows too long for LDS (whole utterances in validation), in a global scratch
(ste_gan_amd/csrc/losses.hip).
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.nn as nn
from torch import Tensor

from .. import ops
from ..layers.average_filter import AverageFilter


class TimeDomainFeatureLoss(nn.Module):
    def __init__(self, num_channels, win_size_samples: int = 21, win_shift_samples: int = 8,
                 apply_padding_windowing: bool = True, average_filter_window_size: int = 9):
        super().__init__()
        if average_filter_window_size < 1 or average_filter_window_size % 2 == 0:
            raise ValueError("average_filter_window_size must be odd (average_filter.py:14: the filter keeps the length)")
        self.num_channels = num_channels
        self.average_filter = AverageFilter(num_channels, average_filter_window_size)
        self.win_size_samples = win_size_samples
        self.win_shift_samples = win_shift_samples
        self.apply_padding_windowing = apply_padding_windowing
        self.avg_filter_window_size = average_filter_window_size

    def double_average(self, x: Tensor) -> Tensor:
        """(B, T, C) -> (B, T, C): the average filter applied twice along time (time_domain_loss.py:52-56)."""
        y = self.average_filter(self.average_filter(ops.btc_to_bct(x)))
        return y.transpose(1, 2).contiguous()

    def calculate_time_domain_features(self, raw_x: Tensor) -> Tensor:
        """(B, T, C) -> (B, frames, C, 4): [mean(low), sum(low^2), sum(rect^2), mean(rect)] (time_domain_loss.py:58-69)."""
        return ops.td_features(raw_x, self.win_size_samples, self.win_shift_samples, self.avg_filter_window_size,
                               self.apply_padding_windowing)

    def time_domain_loss(self, x_real: Tensor, x_generated: Tensor):
        loss, _ = ops.multi_td_loss(x_real, x_generated, (self.win_size_samples,), (self.win_shift_samples,),
                                    self.avg_filter_window_size, self.apply_padding_windowing)
        return loss

    forward = time_domain_loss


class MultiTimeDomainFeatureLoss(nn.Module):
    """Sum of the TD losses at (20,8), (51,13), (80,16) -- time_domain_loss.py:87-107."""

    def __init__(self, num_channels: int):
        super().__init__()
        self.time_domain_losses = nn.ModuleList([
            TimeDomainFeatureLoss(num_channels, win_size_samples=20, win_shift_samples=8),
            TimeDomainFeatureLoss(num_channels, win_size_samples=51, win_shift_samples=13),
            TimeDomainFeatureLoss(num_channels, win_size_samples=80, win_shift_samples=16)])

    def time_domain_loss(self, x_real: Tensor, x_generated: Tensor, weight: float = 1.0,
                         unit_grad: bool = False) -> Tuple[Tensor, List[Tensor]]:
        wins = [m.win_size_samples for m in self.time_domain_losses]
        shifts = [m.win_shift_samples for m in self.time_domain_losses]
        loss, per = ops.multi_td_loss(x_real, x_generated, wins, shifts, weight=weight, unit_grad=unit_grad)
        return loss, [per[i] for i in range(len(wins))]

    def forward(self, x_real: Tensor, x_generated: Tensor, weight: float = 1.0, unit_grad: bool = False) -> Tensor:
        """``weight`` / ``unit_grad`` (the train step): the returned value AND its gradient carry the loss weight, and the
        caller promises an upstream gradient of 1 -- see ops._MultiTDFn.forward.  Defaults: the reference's module."""
        loss, _ = self.time_domain_loss(x_real, x_generated, weight, unit_grad)
        return loss
