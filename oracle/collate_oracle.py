"""CPU restatement of the reference's batch assembly -- TEST INFRASTRUCTURE, not a product path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module (see the header of
stegan_oracle.py).  It restates ``ste_gan/data/collate.py`` in plain torch-CPU indexing:

* ``draw_start_frames``  -- collate.py:116-121: one ``torch.randint(0, 1 + max(0, F - length), (1,))`` per utterance,
  in batch order, from the global torch RNG (so a fixed ``torch.manual_seed`` reproduces the reference's crops);
* ``collate``            -- collate.py:22-61 with prepare_ste_gan_features (:92-157) and prepare_emg (:67-89):
  train = random window of ``length = emg_train_length // hopsize`` feature frames, the matching ``2*length`` MFCC
  rows from ``2*start`` and ``emg_train_length`` EMG samples from ``start*hopsize``; valid = the same windows at
  start 0; test = the single whole utterance with a leading batch axis.

Pinned by tests/golden/collate_*.npz, which tests/golden/make_golden.py produced by calling the reference's own
``ste_gan_collate`` on the synthetic utterances of tests/golden/cases.py under a fixed seed.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

REAL_EMG, SPEECH_UNITS, PHONEMES = "REAL_EMG", "SPEECH_UNITS", "PHONEMES"
SESSION_INDEX, SPEAKING_MODE_INDEX, MFCCS = "SESSION_INDEX", "SPEAKING_MODE_IDX", "MFCCS"
VALID_PARTITIONS = ("train", "valid", "test")


def draw_start_frames(frame_counts: Sequence[int], length: int) -> List[int]:
    """collate.py:116-121 (same call shape, so the global RNG advances exactly as in the reference)."""
    return [int(torch.randint(0, 1 + max(0, int(f) - length), (1,)).item()) for f in frame_counts]


def collate(batch: List[Dict[str, torch.Tensor]], partition: str = "train", emg_train_length: int = 2048,
            hopsize: int = 16, start_frames: Optional[Sequence[int]] = None) -> Dict[str, torch.Tensor]:
    if partition not in VALID_PARTITIONS:                                   # collate.py:15-19
        raise ValueError('Partition must be one of ["train", "valid", "test"]')
    length = emg_train_length // hopsize                                    # collate.py:44
    sess = torch.stack([s[SESSION_INDEX] for s in batch], 0)                # collate.py:101-102
    spk = torch.stack([s[SPEAKING_MODE_INDEX] for s in batch], 0)
    if partition == "test":                                                 # collate.py:148-155, 85-88
        assert len(batch) == 1
        s = batch[0]
        return {REAL_EMG: s[REAL_EMG].unsqueeze(0), SPEECH_UNITS: s[SPEECH_UNITS].unsqueeze(0),
                PHONEMES: s[PHONEMES].unsqueeze(0), SESSION_INDEX: sess, SPEAKING_MODE_INDEX: spk,
                MFCCS: s[MFCCS].unsqueeze(0)}
    if partition == "train":
        if start_frames is None:
            start_frames = draw_start_frames([len(s[SPEECH_UNITS]) for s in batch], length)
    else:                                                                   # valid: collate.py:135-145
        start_frames = [0] * len(batch)
    su = torch.stack([s[SPEECH_UNITS][a:a + length] for s, a in zip(batch, start_frames)], 0)
    ph = torch.stack([s[PHONEMES][a:a + length] for s, a in zip(batch, start_frames)], 0)
    mf = torch.stack([s[MFCCS][2 * a:2 * a + 2 * length] for s, a in zip(batch, start_frames)], 0)
    emg = torch.stack([s[REAL_EMG][a * hopsize:a * hopsize + emg_train_length]        # collate.py:77-83
                       for s, a in zip(batch, start_frames)], 0)
    return {REAL_EMG: emg, SPEECH_UNITS: su, PHONEMES: ph, SESSION_INDEX: sess, SPEAKING_MODE_INDEX: spk, MFCCS: mf}
