"""Frozen EMG encoder (EMG -> soft speech units + phoneme logits), host side.

Mirrors ``ste_gan/models/emg_encoder.py``: ``EMGEncoder`` (:24-32), ``EMGEncoderTransformer`` (:36-88),
``init_emg_encoder`` (:91-115), ``load_emg_encoder`` (:117-127) with identical constructor arguments and
state-dict keys (``conv_blocks.{i}.conv1|bn1|conv2|bn2|residual_path|res_norm``, ``w_raw_in``,
``transformer.layers.{i}...``, ``w_out``, ``w_aux``).  Only what the GAN needs is implemented: the eval-mode
forward (BatchNorm running statistics folded into the conv epilogue's scale / bias, no dropout, no random
shift) and the gradient w.r.t. the input EMG.  Weight gradients are never computed -- the reference's
autograd computes and discards them.  Training the encoder itself
(ste_gan/emg_encoder/train.py) is outside this path: ``forward`` in training mode raises.
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Tuple

import torch
import torch.nn as nn
from torch import Tensor

from .. import constants as K
from .. import ops
from .._lib import ACT_NONE, ACT_RELU
from ..layers.transformer import TransformerEncoderLayer


class EMGEncoder(nn.Module):
    """Base class for EMG encoders."""

    def forward(self, x: Tensor) -> Tuple[Tensor, Tensor]:
        raise NotImplementedError("Must be implemented by subclasses.")


class ResBlock(nn.Module):
    """conv-BN-ReLU-conv-BN + strided 1x1 residual -- ste_gan/layers/conv.py:106-133 (eval-mode BatchNorm)."""

    def __init__(self, num_ins, num_outs, stride=1):
        super().__init__()
        self.conv1 = nn.Conv1d(num_ins, num_outs, 3, padding=1, stride=stride)
        self.bn1 = nn.BatchNorm1d(num_outs)
        self.conv2 = nn.Conv1d(num_outs, num_outs, 3, padding=1)
        self.bn2 = nn.BatchNorm1d(num_outs)
        if stride != 1 or num_ins != num_outs:
            self.residual_path = nn.Conv1d(num_ins, num_outs, 1, stride=stride)
            self.res_norm = nn.BatchNorm1d(num_outs)
        else:
            self.residual_path = None
        self.stride, self.num_ins, self.num_outs = stride, num_ins, num_outs
        self._folded = None

    @staticmethod
    def _fold(conv, bn):
        s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        return s.contiguous(), ((conv.bias - bn.running_mean) * s + bn.bias).contiguous()

    def folded(self):
        mods = [self.conv1, self.bn1, self.conv2, self.bn2] + ([self.residual_path, self.res_norm]
                                                              if self.residual_path is not None else [])
        key = tuple((t.data_ptr(), t._version) for m in mods for t in list(m.parameters()) + list(m.buffers()))
        if self._folded is None or self._folded[0] != key:
            with torch.no_grad():
                f = [self._fold(self.conv1, self.bn1), self._fold(self.conv2, self.bn2)]
                if self.residual_path is not None:
                    f.append(self._fold(self.residual_path, self.res_norm))
            self._folded = (key, f)
        return self._folded[1]

    def forward(self, x):
        f = self.folded()
        ci, co, s = self.num_ins, self.num_outs, self.stride
        # the first ReLU rides on conv2's input staging (its mask in conv2's input-gradient epilogue: no activation-backward
        # launch); x feeds conv1 and the residual branch: the branch's backward runs first and parks its gradient for x,
        # conv1's input-gradient kernel adds it (ops.GradHandover) instead of an autograd add
        hx = ops.GradHandover() if (ops.HANDOVER and torch.is_grad_enabled() and x.requires_grad) else None
        h = ops.frozen_conv1d(x, self.conv1.weight, f[0][0], f[0][1], None, ops.ConvCfg(ci, co, 3, stride=s, pad=1),
                              (ops.TAKE, hx) if hx is not None else None)
        cfg2 = ops.ConvCfg(co, co, 3, pad=1, in_act=ACT_RELU, out_act=ACT_RELU)
        if self.residual_path is not None:
            r = ops.frozen_conv1d(x, self.residual_path.weight, f[2][0], f[2][1], None, ops.ConvCfg(ci, co, 1, stride=s),
                                  (ops.PARK_DX, hx) if hx is not None else None)
            return ops.frozen_conv1d(h, self.conv2.weight, f[1][0], f[1][1], r, cfg2)
        return ops.frozen_conv1d(h, self.conv2.weight, f[1][0], f[1][1], x, cfg2,
                                 (ops.PARK_RES, hx) if hx is not None else None)


class _LayerStack(nn.Module):
    """Keeps the ``transformer.layers.{i}`` key prefix of ``nn.TransformerEncoder``."""

    def __init__(self, make_layer, n):
        super().__init__()
        self.layers = nn.ModuleList([make_layer() for _ in range(n)])

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        return x


class EMGEncoderTransformer(EMGEncoder):
    """Conv + relative-position Transformer encoder -- models/emg_encoder.py:36-88."""

    def __init__(self, num_ins, num_outs, num_aux_outs, model_size: int = 768, num_extra_res_blocks: int = 3,
                 dropout: float = 0.2, num_transformer_layers: int = 6, dim_feedforward: int = 3072):
        super().__init__()
        blocks = [ResBlock(num_ins, model_size, 2)] + [ResBlock(model_size, model_size, 2)
                                                       for _ in range(num_extra_res_blocks)]
        self.conv_blocks = nn.Sequential(*blocks)
        self.w_raw_in = nn.Linear(model_size, model_size)
        self.transformer = _LayerStack(
            lambda: TransformerEncoderLayer(d_model=model_size, nhead=8, relative_positional=True,
                                            relative_positional_distance=100, dim_feedforward=dim_feedforward,
                                            dropout=dropout), num_transformer_layers)
        self.w_out = nn.Linear(model_size, num_outs)
        self.w_aux = nn.Linear(model_size, num_aux_outs)
        self.model_size, self.num_outs, self.num_aux_outs = model_size, num_outs, num_aux_outs

    def forward(self, x_raw: Tensor) -> Tuple[Tensor, Tensor]:
        """(B, T_emg, C_emg) -> ((B, T_emg/16, num_outs), (B, T_emg/16, num_aux_outs))."""
        if self.training:
            raise NotImplementedError("ste_gan_amd runs the EMG encoder frozen in eval mode (ste_gan/train.py:58-60); "
                                      "call .eval() -- encoder pre-training is outside the accelerated path")
        M = self.model_size
        x = ops.btc_to_bct(x_raw)
        x = self.conv_blocks(x)
        x = ops.frozen_conv1d(x, self.w_raw_in.weight.unsqueeze(-1), None, self.w_raw_in.bias, None, ops.ConvCfg(M, M, 1))
        x = self.transformer(x)
        x, x_aux = ops.fan_out(x, 2)                  # both heads read the last layer: one sum of their input gradients
        su = ops.frozen_conv1d(x, self.w_out.weight.unsqueeze(-1), None, self.w_out.bias, None,
                               ops.ConvCfg(M, self.num_outs, 1, out_btc=True))
        ph = ops.frozen_conv1d(x_aux, self.w_aux.weight.unsqueeze(-1), None, self.w_aux.bias, None,
                               ops.ConvCfg(M, self.num_aux_outs, 1, out_btc=True))
        return su, ph


def init_emg_encoder(cfg, device: torch.device = None) -> EMGEncoder:
    """models/emg_encoder.py:91-115."""
    emg_encoder_config = cfg.emg_encoder
    num_ins = cfg.data.num_emg_channels
    num_outs = K.SPEECH_UNITS_FEAT_SIZE
    num_aux_outs = len(K.PHONEME_INVENTORY)
    emg_encoder_type = emg_encoder_config["type"]
    emg_encoder_params = emg_encoder_config["params"]
    logging.info(f"Initializing EMG encoder with type {emg_encoder_type}")
    if emg_encoder_type == "EMGEncoderTransformer":
        emg_encoder = EMGEncoderTransformer(num_ins=num_ins, num_outs=num_outs, num_aux_outs=num_aux_outs,
                                            **emg_encoder_params)
    else:
        raise ValueError(f"Unknown EMG encoder type: {emg_encoder_type}")
    if device:
        emg_encoder = emg_encoder.to(device)
    return emg_encoder


def load_emg_encoder(cfg, device: torch.device, emg_encoder_checkpoint_path: Path) -> EMGEncoder:
    """models/emg_encoder.py:117-127."""
    emg_encoder = init_emg_encoder(cfg, device)
    logging.info(f"Loading checkpoint path: {emg_encoder_checkpoint_path}")
    state_dict = torch.load(emg_encoder_checkpoint_path, map_location=device)
    emg_encoder.load_state_dict(state_dict)
    emg_encoder.eval()
    return emg_encoder
