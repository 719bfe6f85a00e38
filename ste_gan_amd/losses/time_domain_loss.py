"""Time-domain feature losses, host side -- mirrors ``ste_gan/losses/time_domain_loss.py``.

``TimeDomainFeatureLoss`` (:13-73) and ``MultiTimeDomainFeatureLoss`` (:76-107) keep their
constructor arguments and ``forward`` / ``time_domain_loss`` results; the whole chain (two reflect
padded 9-tap average filters, rectification, reflect padded framing, the four frame features for
both signals, the L1 means of up to three scales and d loss / d x_generated) is ONE kernel launch
with every intermediate in LDS (ste_gan_amd/csrc/losses.hip).
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.nn as nn
from torch import Tensor

from .. import ops


class TimeDomainFeatureLoss(nn.Module):
    def __init__(self, num_channels, win_size_samples: int = 21, win_shift_samples: int = 8,
                 apply_padding_windowing: bool = True, average_filter_window_size: int = 9):
        super().__init__()
        if not apply_padding_windowing or average_filter_window_size != 9:
            raise ValueError("ste_gan_amd.TimeDomainFeatureLoss supports the configuration the reference uses: "
                             "padded windowing and a 9-tap average filter")
        self.num_channels = num_channels
        self.win_size_samples = win_size_samples
        self.win_shift_samples = win_shift_samples
        self.apply_padding_windowing = apply_padding_windowing
        self.avg_filter_window_size = average_filter_window_size

    def time_domain_loss(self, x_real: Tensor, x_generated: Tensor):
        loss, _ = ops.multi_td_loss(x_real, x_generated, (self.win_size_samples,), (self.win_shift_samples,))
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

    def time_domain_loss(self, x_real: Tensor, x_generated: Tensor) -> Tuple[Tensor, List[Tensor]]:
        wins = [m.win_size_samples for m in self.time_domain_losses]
        shifts = [m.win_shift_samples for m in self.time_domain_losses]
        loss, per = ops.multi_td_loss(x_real, x_generated, wins, shifts)
        return loss, [per[i] for i in range(len(wins))]

    def forward(self, x_real: Tensor, x_generated: Tensor) -> Tensor:
        loss, _ = self.time_domain_loss(x_real, x_generated)
        return loss
