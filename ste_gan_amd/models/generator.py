"""EMG generator, host side -- mirrors ``ste_gan/models/generator.py``.

``EMGGenerator`` (:19-75), ``EMGGeneratorGanTTS`` (:78-162) and ``init_emg_generator``
(:165-195) keep the reference's constructor arguments, ``forward`` / ``generate`` /
``generate_from_data_dict`` signatures, error behaviour and state-dict keys
(``session_embeddings.weight``, ``gblocks.{0..8}...``, ``last_conv.1...``); the arithmetic is
42 fused conv launches of libstegan_hip.so.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn as nn
from torch import Tensor

from .. import constants as K
from .. import ops
from .._lib import ACT_NONE, ACT_RELU, ACT_TANH
from ..constants import DataType
from ..layers.conv import GBlock, WNConv1d, _Slot


class EMGGenerator(nn.Module):
    """Base class -- ste_gan/models/generator.py:19-75."""

    def __init__(self, speech_feature_type: str, speech_input_dim: int, num_sessions: int,
                 num_output_channels: int) -> None:
        super().__init__()
        self.speech_feature_type = speech_feature_type
        self.speech_input_dim = speech_input_dim
        self.num_output_channels = num_output_channels
        self.num_sessions = num_sessions

    def forward(self, speech_unit_sequence: Tensor, session_ids: Tensor, speaking_mode_ids: Tensor, ar=None):
        raise NotImplementedError("Must implement the EMGGenerator interface.")

    @torch.inference_mode()
    def generate(self, speech_unit_sequence: Tensor, session_ids: Tensor, speaking_mode_ids: Tensor) -> Tensor:
        return self(speech_unit_sequence, session_ids, speaking_mode_ids)

    @torch.inference_mode()
    def generate_from_data_dict(self, data_dict: Dict, device: torch.device) -> Tensor:
        """One utterance from an EMGDataset item -> (T_emg, C) on the CPU (generator.py:52-75)."""
        s_t = data_dict[self.speech_feature_type].to(device)
        sess_idx = data_dict[DataType.SESSION_INDEX].to(device)
        spk_mode_idx = data_dict[DataType.SPEAKING_MODE_INDEX].to(device)
        if len(s_t.shape) == 2:
            s_t, sess_idx, spk_mode_idx = s_t.unsqueeze(0), sess_idx.unsqueeze(0), spk_mode_idx.unsqueeze(0)
        return self.generate(s_t, sess_idx, spk_mode_idx).squeeze(0).detach().cpu()


class EMGGeneratorGanTTS(EMGGenerator):
    """GAN-TTS / CAR-GAN style generator of the paper -- generator.py:78-162."""

    def __init__(self, speech_feature_type: str, speech_input_dim: int, num_sessions: int, num_emg_channels: int,
                 use_speaking_mode_embedding: bool = False, use_session_embeddings: bool = True,
                 num_speaking_modes: int = 3, embedding_dim: int = 64, channels: int = 768):
        super().__init__(speech_feature_type=speech_feature_type, speech_input_dim=speech_input_dim,
                         num_sessions=num_sessions, num_output_channels=num_emg_channels)
        self.session_embeddings = nn.Embedding(num_sessions, embedding_dim) if use_session_embeddings else None
        self.use_session_embeddings = use_session_embeddings
        self.speaking_mode_embeddings = (nn.Embedding(num_speaking_modes, embedding_dim)
                                         if use_speaking_mode_embedding else None)
        self.use_speaking_mode_embedding = use_speaking_mode_embedding
        self.grad_sink = False      # set by optim.FlatParams: the tables' .grad are persistent, zeroed arena views
        self.input_size = self.speech_input_dim + embedding_dim * (int(use_session_embeddings)
                                                                   + int(use_speaking_mode_embedding))
        upsample_last = 2 if self.speech_feature_type == DataType.SPEECH_UNITS else 1
        c = channels
        self.gblocks = nn.Sequential(
            WNConv1d(self.input_size, c, kernel_size=1),
            GBlock(c, c), GBlock(c, c),
            GBlock(c, c // 2, upsample=2), GBlock(c // 2, c // 2, upsample=2), GBlock(c // 2, c // 2, upsample=2),
            GBlock(c // 2, c // 4, upsample=upsample_last),
            GBlock(c // 4, c // 4), GBlock(c // 4, c // 4),
        )
        # ReLU -> conv(k3) -> 'b c t -> b t c' -> tanh, one launch (generator.py:133-137,156-160)
        self.last_conv = nn.Sequential(
            _Slot("ReLU"),
            WNConv1d(c // 4, num_emg_channels, kernel_size=3, padding=1, in_act=ACT_RELU, out_act=ACT_TANH, out_btc=True))

    def forward(self, speech_unit_sequence: Tensor, session_ids: Tensor, speaking_mode_ids: Tensor):
        x = speech_unit_sequence
        T = x.shape[1]
        if self.use_session_embeddings and x.is_cuda:
            # nn.Embedding raises on an id outside its table; the gather kernel would read a zero row instead.  Checked on the
            # host wherever a read-back is possible (eager calls); a captured / replayed step cannot, and takes ids that the
            # data path has validated (data.loader: session / mode indices come from the corpus tables)
            if not torch.cuda.is_current_stream_capturing():
                _check_ids(session_ids, self.session_embeddings.num_embeddings, "session_ids")
                if self.use_speaking_mode_embedding:
                    _check_ids(speaking_mode_ids, self.speaking_mode_embeddings.num_embeddings, "speaking_mode_ids")
            # gather + broadcast over time + concat + 'b t c -> b c t' (generator.py:143-154) in one launch
            w1 = self.session_embeddings.weight
            w2 = self.speaking_mode_embeddings.weight if self.use_speaking_mode_embedding else None
            sink = self.grad_sink and torch.is_grad_enabled() and w1.requires_grad and w1.grad is not None and w1.grad.is_contiguous()
            x = ops.embed_concat(x, session_ids, w1, speaking_mode_ids if w2 is not None else None, w2,
                                 w1.grad if sink else None,
                                 w2.grad if (sink and w2 is not None and w2.grad is not None) else None)
        else:
            if self.use_session_embeddings:
                e = self.session_embeddings(session_ids)
                x = torch.cat((x, e[:, None, :].expand(-1, T, -1)), dim=-1)
            if self.use_speaking_mode_embedding:
                e = self.speaking_mode_embeddings(speaking_mode_ids)
                x = torch.cat((x, e[:, None, :].expand(-1, T, -1)), dim=-1)
            x = x.transpose(1, 2).contiguous()           # b t c -> b c t
        x = self.gblocks(x)
        return self.last_conv[1](x)                  # (B, T_emg, C_emg) in (-1, 1)


def _check_ids(ids: Tensor, n: int, what: str) -> None:
    """IndexError for an embedding id outside [0, n), as torch.nn.functional.embedding raises (one read-back of two words)."""
    if ids.numel() == 0:
        return
    lo, hi = torch.aminmax(ids)
    lo, hi = int(lo), int(hi)
    if lo < 0 or hi >= n:
        raise IndexError(f"index out of range in self ({what}: values in [{lo}, {hi}], embedding table has {n} rows)")


def init_emg_generator(cfg, emg_generator_type: str = "") -> EMGGenerator:
    """Factory from the merged YAML config -- ste_gan/models/generator.py:165-195."""
    speech_feature_type = cfg.model.speech_feature_type
    if speech_feature_type == DataType.SPEECH_UNITS:
        speech_input_dim = K.SPEECH_UNITS_FEAT_SIZE
    elif speech_feature_type == DataType.MFCCS:
        speech_input_dim = K.NUM_MFCCS
    else:
        raise ValueError(f"Unrecognized speech feature type: {speech_feature_type}")
    if not emg_generator_type:
        emg_generator_type = cfg.model.type
    params = dict(num_emg_channels=cfg.data.num_emg_channels, num_sessions=cfg.data.num_emg_sessions,
                  speech_feature_type=speech_feature_type, speech_input_dim=speech_input_dim)
    extra_params = cfg.model.params if "params" in cfg.model else {}
    if emg_generator_type == "EMGGeneratorGanTTS":
        return EMGGeneratorGanTTS(**params, **extra_params)
    raise ValueError(f"Unrecognized EMG generator type: {emg_generator_type}")
