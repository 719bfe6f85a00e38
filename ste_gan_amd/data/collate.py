"""Batch assembly on the device -- the role of ``ste_gan/data/collate.py`` with the corpus resident in HBM.

The reference slices each utterance on a DataLoader worker, ``torch.stack``s the crops and ships the batch through a
pickle hop to the trainer, which then copies it to the GPU (collate.py:22-61, loader.py:66-73, train.py:167-178).  Here
the utterances are uploaded ONCE (``UtteranceStore``), concatenated per field with row-offset tables, and a batch is one
launch of ``stegan_assemble_batch`` (include/stegan_hip.h) driven by two small int32 vectors: which utterances, and at
which feature frame each window starts.  Window starts are drawn on the host exactly as the reference draws them
(collate.py:118-121), so a fixed ``torch.manual_seed`` gives the reference's crops.

``ste_gan_collate`` keeps the reference's signature and output dict for code that still hands over a list of
per-utterance sample dicts.
"""
from __future__ import annotations

import ctypes as C
import warnings
from typing import Dict, List, Optional, Sequence

import torch

from .. import _lib
from .. import constants as K
from .._lib import Corpus, check
from ..constants import DataType

VALID_PARTITIONS = {"train", "valid", "test"}


def check_partition(partition: str) -> None:
    """collate.py:15-19."""
    if partition not in VALID_PARTITIONS:
        raise ValueError('Partition must be one of ["train", "valid", "test"]')


def draw_start_frames(frame_counts: Sequence[int], length: int) -> List[int]:
    """One ``torch.randint(0, 1 + max(0, F - length), (1,))`` per utterance in batch order (collate.py:118-121): the
    global torch RNG advances exactly as in the reference."""
    return [int(torch.randint(0, 1 + max(0, int(f) - length), (1,)).item()) for f in frame_counts]


def _cat(ts: List[torch.Tensor], dtype) -> torch.Tensor:
    return torch.cat([t.reshape(t.shape[0], -1) if t.dim() > 1 else t for t in ts], 0).to(dtype).contiguous()


def pack_samples(samples: Sequence[Dict]) -> Dict:
    """Per-utterance sample dicts -> the packed host form of a partition: every field concatenated along time plus the
    row counts (``emg_len`` / ``frames`` / ``mfcc_len``).  This is what a shard file holds (``save_packed``)."""
    if len(samples) == 0:
        raise ValueError("UtteranceStore needs at least one utterance")
    emg = [s[DataType.REAL_EMG] for s in samples]
    su = [s[DataType.SPEECH_UNITS] for s in samples]
    ph = [s[DataType.PHONEMES] for s in samples]
    mf = [s.get(DataType.MFCCS) for s in samples]
    if any(m is None for m in mf) and not all(m is None for m in mf):
        raise ValueError("either every utterance carries MFCCs or none does")
    for i, (e, u, p) in enumerate(zip(emg, su, ph)):
        if e.dim() != 2 or u.dim() != 2 or p.dim() != 1:
            raise ValueError(f"utterance {i}: expected EMG (T,C), units (F,D), phonemes (F,); got "
                             f"{tuple(e.shape)}, {tuple(u.shape)}, {tuple(p.shape)}")
        if len(u) != len(p):
            raise ValueError(f"utterance {i}: {len(u)} unit frames but {len(p)} phoneme labels")
    if any(e.shape[1] != emg[0].shape[1] for e in emg) or any(u.shape[1] != su[0].shape[1] for u in su):
        raise ValueError("utterances disagree on the EMG channel count or the unit dimension")
    has_mfcc = mf[0] is not None
    scalar = lambda key: torch.stack([torch.as_tensor(s[key]).reshape(()) for s in samples]).to(torch.int64)
    return {"emg": _cat(emg, torch.float32), "units": _cat(su, torch.float32), "phonemes": _cat(ph, torch.int64),
            "mfcc": _cat(mf, torch.float32) if has_mfcc else None,
            "session_idx": scalar(DataType.SESSION_INDEX), "mode_idx": scalar(DataType.SPEAKING_MODE_INDEX),
            "emg_len": [len(e) for e in emg], "frames": [len(u) for u in su],
            "mfcc_len": [len(m) for m in mf] if has_mfcc else [0] * len(samples),
            "utt_ids": [s.get(DataType.UTT_ID, "") for s in samples]}


_SHARD_MAGIC = b"STEGANSHARD1\n"
_SHARD_FIELDS = (("emg", "<f4"), ("units", "<f4"), ("phonemes", "<i8"), ("mfcc", "<f4"), ("session_idx", "<i8"), ("mode_idx", "<i8"))


def save_packed(packed: Dict, path) -> None:
    """One shard file per partition: magic, a JSON header (shapes, row counts, utterance ids, byte offsets), then the raw
    little-endian arrays, each 64-byte aligned -- so a later run maps it instead of unpickling one ``.pt`` per utterance
    and field (the reference's layout: 4 files x thousands of utterances, data/emg_dataset.py:229-251)."""
    import json
    import numpy as np
    head = {"emg_len": packed["emg_len"], "frames": packed["frames"], "mfcc_len": packed["mfcc_len"],
            "utt_ids": packed["utt_ids"], "arrays": {}}
    blobs, off = [], 0
    for name, dt in _SHARD_FIELDS:
        t = packed[name]
        if t is None:
            continue
        a = np.ascontiguousarray(t.numpy()).astype(dt, copy=False)
        head["arrays"][name] = {"dtype": dt, "shape": list(a.shape), "offset": off}
        blobs.append((off, a))
        off += (a.nbytes + 63) // 64 * 64
    hj = json.dumps(head).encode()
    pre = len(_SHARD_MAGIC) + 8 + len(hj)
    pad = (-pre) % 64
    with open(path, "wb") as f:
        f.write(_SHARD_MAGIC)
        f.write((len(hj) + pad).to_bytes(8, "little"))
        f.write(hj + b" " * pad)
        base = f.tell()
        for o, a in blobs:
            f.seek(base + o)
            f.write(a.tobytes())
        f.truncate(base + off)


def load_packed(path) -> Dict:
    """Inverse of ``save_packed``; the arrays are memory-mapped (read-only), so the upload streams from the page cache."""
    import json
    import numpy as np
    with open(path, "rb") as f:
        if f.read(len(_SHARD_MAGIC)) != _SHARD_MAGIC:
            raise ValueError(f"{path}: not a ste-gan shard file")
        hlen = int.from_bytes(f.read(8), "little")
        head = json.loads(f.read(hlen).decode())
        base = f.tell()
    out = {k: head[k] for k in ("emg_len", "frames", "mfcc_len", "utt_ids")}
    for name, _ in _SHARD_FIELDS:
        meta = head["arrays"].get(name)
        if meta is None:
            out[name] = None
            continue
        a = np.memmap(path, dtype=meta["dtype"], mode="r", offset=base + meta["offset"], shape=tuple(meta["shape"]))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                  # read-only mapping: the tensor is only ever a copy source
            out[name] = torch.from_numpy(np.asarray(a))
    return out


class UtteranceStore:
    """A partition's utterances resident on the device.

    ``samples`` are dicts shaped like ``EMGDataset.__getitem__`` output (data/emg_dataset.py:277-294): ``REAL_EMG``
    (T, C) float, ``SPEECH_UNITS`` (F, D) float, ``PHONEMES`` (F,) int, ``SESSION_INDEX`` / ``SPEAKING_MODE_INDEX``
    0-d int, optionally ``MFCCS`` (M, n_mfcc) float (``None`` for all utterances = no MFCC arena) -- or the packed form
    of ``pack_samples`` / ``load_packed``.  Row counts stay on the host too: window validation never touches the device.
    """

    def __init__(self, samples, device="cuda"):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(f"UtteranceStore lives in HBM; got device '{device}' (there is no CPU fallback)")
        self.device = device
        pk = samples if isinstance(samples, dict) else pack_samples(samples)
        self.emg_len, self.frames, self.mfcc_len = list(pk["emg_len"]), list(pk["frames"]), list(pk["mfcc_len"])
        self.utt_ids = list(pk.get("utt_ids", []))
        self.n = len(self.frames)
        if self.n == 0:
            raise ValueError("UtteranceStore needs at least one utterance")
        self.has_mfcc = pk["mfcc"] is not None
        self.emg_channels, self.unit_dim = pk["emg"].shape[1], pk["units"].shape[1]
        self.mfcc_dim = pk["mfcc"].shape[1] if self.has_mfcc else 0

        def offsets(lens):
            return torch.tensor([0] + list(torch.tensor(lens, dtype=torch.int64).cumsum(0)), dtype=torch.int64, device=device)

        self.emg = pk["emg"].to(device)
        self.units = pk["units"].to(device)
        self.phonemes = pk["phonemes"].to(device)
        self.mfcc = pk["mfcc"].to(device) if self.has_mfcc else None
        self.emg_off, self.frame_off = offsets(self.emg_len), offsets(self.frames)
        self.mfcc_off = offsets(self.mfcc_len) if self.has_mfcc else None
        self.session_idx = pk["session_idx"].to(device)
        self.mode_idx = pk["mode_idx"].to(device)
        P = lambda t: t.data_ptr() if t is not None else None
        self._desc = Corpus(P(self.emg), P(self.emg_off), P(self.units), P(self.phonemes), P(self.frame_off), P(self.mfcc),
                            P(self.mfcc_off), P(self.session_idx), P(self.mode_idx), self.n, self.emg_channels,
                            self.unit_dim, self.mfcc_dim)

    def __len__(self):
        return self.n

    @property
    def nbytes(self) -> int:
        ts = [self.emg, self.units, self.phonemes] + ([self.mfcc] if self.has_mfcc else [])
        return sum(t.numel() * t.element_size() for t in ts)

    # ------------------------------------------------------------------------------------------
    def check_windows(self, utts: Sequence[int], starts: Sequence[int], n_frames: int, n_emg: int, hopsize: int) -> None:
        """The reference's ``torch.stack`` raises when a crop comes out short (collate.py:81-83,132-135); same error here,
        decided on the host from the stored lengths before anything is launched."""
        for b, (u, a) in enumerate(zip(utts, starts)):
            if not 0 <= u < self.n:
                raise IndexError(f"utterance index {u} out of range for a store of {self.n}")
            got = (min(n_frames, max(0, self.frames[u] - a)), min(n_emg, max(0, self.emg_len[u] - a * hopsize)),
                   min(2 * n_frames, max(0, self.mfcc_len[u] - 2 * a)) if self.has_mfcc else 2 * n_frames)
            if a < 0 or got != (n_frames, n_emg, 2 * n_frames):
                raise RuntimeError(f"stack expects each tensor to be equal size, but utterance {u} (batch entry {b}) yields "
                                   f"{got[0]} feature frames / {got[1]} EMG samples / {got[2]} MFCC rows from frame {a}; "
                                   f"expected {n_frames} / {n_emg} / {2 * n_frames}")

    def assemble(self, utts, starts, n_frames: int, n_emg: int, hopsize: int = K.HOPSIZE, out: Optional[Dict] = None,
                 validate: bool = True) -> Dict[str, torch.Tensor]:
        """One gather launch -> the reference's batch dict (collate.py:54-61), tensors on the device.

        ``utts`` / ``starts``: host int sequences (validated, then uploaded) or device int32 tensors prepared by the
        caller (``validate=False``: e.g. slices of an epoch plan that was checked as a whole)."""
        dev = self.device
        if torch.is_tensor(utts) and utts.is_cuda:
            if validate:
                raise ValueError("device-side plans are validated by their producer; pass validate=False")
            u_dev, a_dev, B = utts, starts, int(utts.numel())
            if u_dev.dtype != torch.int32 or a_dev.dtype != torch.int32 or a_dev.numel() != B:
                raise TypeError("device plans must be int32 vectors of equal length")
        else:
            utts, starts = [int(u) for u in utts], [int(a) for a in starts]
            if len(utts) != len(starts) or not utts:
                raise ValueError("utts and starts must be non-empty and of equal length")
            if validate:
                self.check_windows(utts, starts, n_frames, n_emg, hopsize)
            plan = torch.tensor([utts, starts], dtype=torch.int32).to(dev)
            u_dev, a_dev, B = plan[0], plan[1], len(utts)
        if out is None:
            out = {DataType.REAL_EMG: torch.empty((B, n_emg, self.emg_channels), device=dev, dtype=torch.float32),
                   DataType.SPEECH_UNITS: torch.empty((B, n_frames, self.unit_dim), device=dev, dtype=torch.float32),
                   DataType.PHONEMES: torch.empty((B, n_frames), device=dev, dtype=torch.int64),
                   DataType.SESSION_INDEX: torch.empty((B,), device=dev, dtype=torch.int64),
                   DataType.SPEAKING_MODE_INDEX: torch.empty((B,), device=dev, dtype=torch.int64),
                   DataType.MFCCS: (torch.empty((B, 2 * n_frames, self.mfcc_dim), device=dev, dtype=torch.float32)
                                    if self.has_mfcc else None)}
        P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        check(_lib.lib().stegan_assemble_batch(
            C.byref(self._desc), P(u_dev), P(a_dev), B, n_frames, n_emg, hopsize, P(out[DataType.REAL_EMG]),
            P(out[DataType.SPEECH_UNITS]), P(out[DataType.PHONEMES]), P(out[DataType.MFCCS]),
            P(out[DataType.SESSION_INDEX]), P(out[DataType.SPEAKING_MODE_INDEX]),
            C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "stegan_assemble_batch")
        return out

    def whole_utterance(self, u: int) -> Dict[str, torch.Tensor]:
        """The test partition's batch of one (collate.py:85-88,148-155): views of the resident arenas, no copy."""
        if not 0 <= u < self.n:
            raise IndexError(f"utterance index {u} out of range for a store of {self.n}")
        e0, f0 = sum(self.emg_len[:u]), sum(self.frames[:u])
        m0 = sum(self.mfcc_len[:u])
        return {DataType.REAL_EMG: self.emg[e0:e0 + self.emg_len[u]].unsqueeze(0),
                DataType.SPEECH_UNITS: self.units[f0:f0 + self.frames[u]].unsqueeze(0),
                DataType.PHONEMES: self.phonemes[f0:f0 + self.frames[u]].unsqueeze(0),
                DataType.SESSION_INDEX: self.session_idx[u:u + 1],
                DataType.SPEAKING_MODE_INDEX: self.mode_idx[u:u + 1],
                DataType.MFCCS: self.mfcc[m0:m0 + self.mfcc_len[u]].unsqueeze(0) if self.has_mfcc else None}


def ste_gan_collate(batch, partition="train", emg_train_length: int = K.TRAIN_EMG_LENGTH, chunksize: int = K.CHUNK_SIZE,
                    hopsize: int = K.HOPSIZE, device="cuda") -> Dict[str, torch.Tensor]:
    """Same arguments, RNG use and output dict as the reference's ``ste_gan_collate`` (collate.py:22-61); the crops are
    cut by the gather kernel and the result lives on the device.  (For steady-state training use ``DeviceLoader``, which
    uploads the corpus once instead of once per call.)"""
    check_partition(partition)
    store = UtteranceStore(batch, device)
    if partition == "test":
        assert len(batch) == 1                                             # collate.py:87,149-151
        return store.whole_utterance(0)
    length = emg_train_length // hopsize                                    # collate.py:44
    starts = draw_start_frames(store.frames, length) if partition == "train" else [0] * len(batch)
    return store.assemble(range(len(batch)), starts, length, emg_train_length, hopsize)
