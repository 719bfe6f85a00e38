"""Average filter, host side -- mirrors ``ste_gan/layers/average_filter.py:10-28``.

``AverageFilter(in_channels, window_size=9, pad_signal=True)``: reflect padding by ``window_size // 2`` followed by
``AvgPool1d(window_size, stride=1)`` over the last axis of a (B, C, T) signal, as one HIP kernel with its adjoint as
backward (``stegan_average_filter_fwd / _bwd``, csrc/losses.hip).  The time-domain loss does not call this module in
its hot path -- its fused kernel applies the same filter twice inside LDS -- but the module is part of the reference's
layer surface and pins that arithmetic on its own (tests/test_gpu_losses.py).
"""
from __future__ import annotations

import torch.nn as nn
from torch import Tensor

from .. import ops


class AverageFilter(nn.Module):
    def __init__(self, in_channels: int, window_size: int = 9, pad_signal: bool = True):
        super().__init__()
        if window_size < 1:
            raise ValueError("window_size must be positive")
        self.in_channels = in_channels
        self.window_size = window_size
        self.padding = window_size // 2
        self.pad_signal = pad_signal

    def forward(self, x: Tensor) -> Tensor:
        return ops.average_filter(x, self.window_size, self.pad_signal)
