"""Speech-unit and phoneme losses through the frozen EMG encoder -- mirrors ``ste_gan/losses/emg_encoder_loss.py``.

``EMGEncoderLossOutput`` keeps the five fields and four counter properties of the reference (:19-53);
the counters come out of the cross-entropy kernel as exact integers (one D2H read of 4 ints instead
of four ``.item()`` reductions).  ``EMGEncoderLoss`` (:56-84) keeps its constructor / forward signature.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor

from .. import ops
from ..constants import SILENCE_PHONEME_INDEX
from ..models.emg_encoder import EMGEncoder


@dataclass
class EMGEncoderLossOutput:
    speech_unit_pred: Tensor
    phoneme_pred: Tensor
    speech_unit_loss: Tensor
    phoneme_loss: Tensor
    phoneme_targets: Tensor
    _counters: Optional[Tensor] = field(default=None, repr=False)
    _argmax: Optional[Tensor] = field(default=None, repr=False)
    _host: Optional[list] = field(default=None, repr=False)

    def _c(self, i: int) -> int:
        if self._host is None:
            self._host = self._counters.tolist()        # single 16-byte readback
        return int(self._host[i])

    @property
    def num_phones(self) -> int:
        return int(self.phoneme_targets.numel())

    @property
    def num_silence_phones(self) -> int:
        return self._c(1)

    @property
    def num_correct_phones(self) -> int:
        return self._c(2)

    @property
    def num_correct_phones_no_silence(self) -> int:
        return self._c(3)


class EMGEncoderLoss(nn.Module):
    def __init__(self, emg_encoder: EMGEncoder) -> None:
        super().__init__()
        self.emg_encoder = emg_encoder
        self.emg_encoder.eval()

    def speech_unit_loss(self, speech_unit_target: Tensor, speech_unit_pred: Tensor) -> Tensor:
        return ops.speech_unit_loss(speech_unit_target, speech_unit_pred)

    def forward(self, emg_signal: Tensor, target_speech_units: Tensor, target_phoneme_sequence: Tensor,
                su_weight: float = 1.0, ph_weight: float = 1.0, unit_grad: bool = False) -> EMGEncoderLossOutput:
        """``su_weight`` / ``ph_weight`` / ``unit_grad`` (the train step only): the two returned losses AND their gradients
        carry the weights, upstream gradient 1 promised (ops._MultiTDFn.forward).  Defaults: the reference's module."""
        speech_unit_pred, phoneme_pred = self.emg_encoder(emg_signal)
        su_loss = ops.speech_unit_loss(target_speech_units, speech_unit_pred, su_weight, unit_grad)
        ph_loss, counters, argmax = ops.phoneme_ce(phoneme_pred, target_phoneme_sequence, SILENCE_PHONEME_INDEX, ph_weight,
                                                   unit_grad)
        return EMGEncoderLossOutput(speech_unit_pred, phoneme_pred, su_loss, ph_loss, target_phoneme_sequence,
                                    counters, argmax)
