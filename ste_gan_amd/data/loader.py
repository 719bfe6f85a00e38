"""Loaders over a device-resident corpus -- the role of ``ste_gan/data/loader.py`` without DataLoader workers.

``loaders(...)`` / ``loaders_via_config(cfg)`` keep the reference's signatures (loader.py:19-29,92-109) and return three
iterables whose batches are the reference's ``DataType``-keyed dicts (collate.py:54-61), already on the GPU:

* train: a fresh random permutation per epoch (``RandomSampler``), random window per utterance (collate.py:118-121);
* valid: sequential, windows at frame 0 (collate.py:135-145);
* test : sequential, whole utterances at batch size 1 (collate.py:148-155), served as views of the resident arenas.

An epoch's (utterance, start frame) plan is drawn on the host in one go, validated against the stored lengths and
uploaded once; each batch is then a single ``stegan_assemble_batch`` launch reading its slice of the plan, so the
steady state has no per-batch host->device copy and nothing to pickle.
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Dict, Iterator, List, Optional, Sequence

import torch

from .. import constants as K
from ..constants import DataDir, DataType, SpeakingMode
from .collate import UtteranceStore, check_partition, draw_start_frames


class DeviceLoader:
    """Iterable over one partition of a resident corpus (``len()`` = batches per epoch, like a DataLoader)."""

    def __init__(self, store: UtteranceStore, partition: str, batch_size: int, emg_train_length: int,
                 hopsize: int = K.HOPSIZE, hop_aligned_emg: bool = False, rank: int = 0, world_size: int = 1):
        check_partition(partition)
        if not (0 <= rank < world_size):
            raise ValueError(f"rank {rank} outside world of {world_size}")
        self.rank, self.world_size = rank, world_size          # data-parallel shard of the TRAIN partition (see plan_epoch)
        self.session_id_to_idx: Optional[Dict] = None           # set by loaders(): the id maps the reference dumps as JSON
        self.speaking_mode_id_to_idx: Optional[Dict] = None
        if batch_size < 1:
            raise ValueError(f"batch_size should be a positive integer value, but got batch_size={batch_size}")
        self.store, self.partition, self.hopsize = store, partition, hopsize
        self.batch_size = 1 if partition == "test" else batch_size
        self.n_frames = emg_train_length // hopsize                        # collate.py:44
        # the reference crops emg_train_length samples even when that is not a multiple of the hop (chunk 200 -> 12
        # frames but 200 samples); hop_aligned_emg crops n_frames * hopsize instead, which is what the generator emits
        self.n_emg = self.n_frames * hopsize if hop_aligned_emg else emg_train_length

    def __len__(self) -> int:
        n, gb = len(self.store), self.batch_size * self._shards()
        full, rem = divmod(n, gb)
        return full + (1 if rem >= self._shards() else 0)                   # world 1: DataLoader default drop_last=False

    def _shards(self) -> int:
        return self.world_size if self.partition == "train" else 1

    def epoch_plan(self):
        """Host lists (utterance order, start frames) of THIS rank for one epoch (see :func:`plan_epoch`)."""
        return plan_epoch(self.store.frames, self.partition, self.batch_size, self.n_frames, self.rank, self._shards())

    def save_session_and_speaking_mode_mapping_json(self, save_dir) -> None:
        """``session_idx_to_id.json`` / ``speaking_mode_idx_to_id.json`` as the reference's train data set writes them
        (emg_dataset.py:202-217; train.py:121-122): the reference's eval / generate scripts read them back."""
        import json
        if self.session_id_to_idx is None or self.speaking_mode_id_to_idx is None:
            raise RuntimeError("this loader was not built from a corpus directory: no session / speaking-mode id maps")
        for name, m in (("session_idx_to_id.json", self.session_id_to_idx),
                        ("speaking_mode_idx_to_id.json", self.speaking_mode_id_to_idx)):
            path = Path(save_dir) / name
            logging.info(f"Saving {name[:-5]} to {path}")
            with open(path, "w") as fp:
                json.dump({idx: key for key, idx in m.items()}, fp)

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        st = self.store
        if self.partition == "test":
            for u in range(len(st)):
                yield st.whole_utterance(u)
            return
        order, starts = self.epoch_plan()
        st.check_windows(order, starts, self.n_frames, self.n_emg, self.hopsize)
        plan = torch.tensor([order, starts], dtype=torch.int32).to(st.device)
        for i in range(0, len(order), self.batch_size):
            yield st.assemble(plan[0, i:i + self.batch_size], plan[1, i:i + self.batch_size], self.n_frames, self.n_emg,
                              self.hopsize, validate=False)


def plan_epoch(frames: Sequence[int], partition: str, batch_size: int, n_frames: int, rank: int = 0, world_size: int = 1):
    """(utterance order, start frames) one rank walks in one epoch; host-only, uses the global torch RNG.

    world_size 1: the reference's RNG order -- ``RandomSampler``'s permutation (loader.py:50), then one ``torch.randint``
    per utterance in batch order (collate.py:118-121).  world_size N (train partition only): EVERY rank draws that same
    full plan for a global batch of N * batch_size (the ranks seed their RNG identically and consume it identically, so
    they stay in lockstep by construction) and keeps rows [rank * batch_size, (rank + 1) * batch_size) of every global
    batch: together the ranks see exactly the batches a single process with batch N * batch_size would.  A trailing
    partial global batch is split evenly (rows beyond a multiple of N are dropped, and the whole remainder when it has
    fewer than N rows) so that all ranks take the same number of steps -- each step carries a collective."""
    n = len(frames)
    if partition != "train":
        return list(range(n)), [0] * n
    order = torch.randperm(n).tolist()
    gb = batch_size * world_size
    starts: List[int] = []
    for i in range(0, n, gb):                                              # one collate call per (global) batch
        starts += draw_start_frames([frames[u] for u in order[i:i + gb]], n_frames)
    if world_size == 1:
        return order, starts
    my_order, my_starts = [], []
    for i in range(0, n, gb):
        rows = min(gb, n - i)
        per = batch_size if rows == gb else rows // world_size
        lo = i + rank * per
        my_order += order[lo:lo + per]
        my_starts += starts[lo:lo + per]
    return my_order, my_starts


# ---------------------------------------------------------------------------------------------
# corpus directory -> per-utterance sample dicts (host I/O, once at start-up)
# ---------------------------------------------------------------------------------------------
def _utt_session(utt_id: str) -> str:
    return utt_id.split("__")[0]                                           # emg_dataset.py:187-189


def _utt_mode(utt_id: str) -> str:
    return utt_id.split("__")[-1]                                          # emg_dataset.py:191-193


def _file_id(emg_path: Path, emg_synth_prefix: str = "emg_synth__") -> str:
    """emg_dataset.py:195-200: utterances of a SYNTHETIC-EMG directory (one whose name carries the prefix) share the
    units / phonemes / transcription files of the recording they were generated from, named by the 2nd id field."""
    if emg_synth_prefix in emg_path.parent.name:
        return emg_path.stem.split("__")[1]
    return emg_path.stem


def read_partition(root_dir: Path, partition: str, session_id_to_idx: Optional[Dict] = None,
                   speaking_mode_id_to_idx: Optional[Dict] = None, only_include_voiced: bool = True,
                   only_include_silent: bool = False, filter_by_length: bool = True, strict: bool = False,
                   train_emg_length: int = K.TRAIN_EMG_LENGTH):
    """Read ``<root>/<partition>/{emg,units,phonemes,mfccs}/*.pt`` the way ``EMGDataset`` selects and labels utterances
    (emg_dataset.py:50-151,229-294): sorted EMG paths; short utterances dropped unless testing; speaking-mode filter;
    session / speaking-mode ids from the ``{session}__...__{mode}`` utterance id, indexed in sorted order unless the
    train partition's mappings are handed in.  Returns (samples, session_id_to_idx, speaking_mode_id_to_idx, utt_ids).
    """
    check_partition(partition)
    assert not (only_include_silent and only_include_voiced), \
        "Either only_include_silent or only_include_voiced can be True."
    split = Path(root_dir) / partition
    is_test = partition == "test"
    samples, utt_ids = [], []
    n_filtered = 0
    for emg_path in sorted((split / DataDir.EMG).glob("*.pt")):
        emg = torch.load(emg_path)
        utt_id = emg_path.stem
        silent = _utt_mode(utt_id) != SpeakingMode.NORMAL
        if (filter_by_length and len(emg) < train_emg_length and not is_test) or (only_include_voiced and silent) \
                or (only_include_silent and not silent):
            n_filtered += 1
            continue
        fid = _file_id(emg_path)
        units = torch.load(split / DataDir.SPEECH_UNITS / f"{fid}.pt")
        phonemes = torch.load(split / DataDir.PHONEMES / f"{fid}.pt")
        mfcc_path = split / DataDir.MFCCS / f"{utt_id}.pt"
        mfccs = torch.load(mfcc_path) if mfcc_path.exists() else None
        if strict:                                                         # emg_dataset.py:254-261
            assert len(units) == len(phonemes)
            assert len(emg) == K.HOPSIZE * len(units)
        samples.append({DataType.UTT_ID: utt_id, DataType.REAL_EMG: emg, DataType.SPEECH_UNITS: units,
                        DataType.PHONEMES: phonemes, DataType.MFCCS: mfccs,
                        DataType.SESSION_ID: _utt_session(utt_id), DataType.SPEAKING_MODE_ID: _utt_mode(utt_id)})
        utt_ids.append(utt_id)
    logging.info(f"Filtered {n_filtered} EMG paths for partition {partition}; kept {len(samples)}.")
    if not session_id_to_idx:
        session_id_to_idx = {s: i for i, s in enumerate(sorted({x[DataType.SESSION_ID] for x in samples}))}
    if not speaking_mode_id_to_idx:
        speaking_mode_id_to_idx = {m: i for i, m in enumerate(sorted({x[DataType.SPEAKING_MODE_ID] for x in samples}))}
    for x in samples:
        x[DataType.SESSION_INDEX] = torch.tensor(session_id_to_idx[x[DataType.SESSION_ID]])
        x[DataType.SPEAKING_MODE_INDEX] = torch.tensor(speaking_mode_id_to_idx[x[DataType.SPEAKING_MODE_ID]])
    return samples, session_id_to_idx, speaking_mode_id_to_idx, utt_ids


def check_no_data_overlap(utt_id_lists: Sequence[Sequence[str]]) -> None:
    """emg_dataset.py:220-226."""
    seen = set()
    for ids in utt_id_lists:
        inter = seen & set(ids)
        if inter:
            raise ValueError(f"Intersecting utterance ids: {','.join(sorted(inter))}")
        seen |= set(ids)


def loaders_from_samples(train_samples, valid_samples, test_samples, batch_size: int = K.BATCH_SIZE,
                         train_emg_length: int = K.TRAIN_EMG_LENGTH, hopsize: int = K.HOPSIZE, device="cuda",
                         hop_aligned_emg: bool = False, rank: int = 0, world_size: int = 1):
    """(train, valid, test) DeviceLoaders over three lists of sample dicts (each uploaded once).  ``rank`` /
    ``world_size`` shard the train partition's epoch plan (:func:`plan_epoch`); valid / test stay whole."""
    mk = lambda samples, part: DeviceLoader(UtteranceStore(samples, device), part, batch_size, train_emg_length, hopsize,
                                            hop_aligned_emg, rank, world_size)
    return mk(train_samples, "train"), mk(valid_samples, "valid"), mk(test_samples, "test")


def _packed_partition(data_root, partition, shard_dir, s2i, m2i, **kw):
    """(packed partition, utterance ids, session map, mode map): from ``<shard_dir>/<partition>.shard`` when it exists,
    else from the per-utterance ``.pt`` files (and the shard is written for next time)."""
    from .collate import load_packed, pack_samples, save_packed
    shard = Path(shard_dir) / f"{partition}.shard" if shard_dir else None
    if shard is not None and shard.exists():
        pk = load_packed(shard)
        ids = pk["utt_ids"]
        if not s2i:                                       # same rule as read_partition: sorted ids of THIS partition
            s2i = {s: i for i, s in enumerate(sorted({_utt_session(u) for u in ids}))}
            m2i = {m: i for i, m in enumerate(sorted({_utt_mode(u) for u in ids}))}
        return pk, ids, s2i, m2i
    samples, s2i, m2i, ids = read_partition(data_root, partition, s2i, m2i, **kw)
    pk = pack_samples(samples)
    if shard is not None:
        shard.parent.mkdir(parents=True, exist_ok=True)
        save_packed(pk, shard)
    return pk, ids, s2i, m2i


def loaders(data_root: Path, strict=False, chunksize: int = K.CHUNK_SIZE, hopsize: int = K.HOPSIZE,
            train_emg_length: int = K.TRAIN_EMG_LENGTH, batch_size: int = K.BATCH_SIZE, device="cuda",
            hop_aligned_emg: bool = False, shard_dir=None, rank: int = 0, world_size: int = 1):
    """loader.py:19-90 over resident stores.  ``shard_dir``: cache each partition as one packed, memory-mappable shard
    file (``collate.save_packed``) and start from it when present (the selection parameters are baked into a shard:
    use a fresh directory per ``train_emg_length`` / filter setting)."""
    tr, tr_ids, s2i, m2i = _packed_partition(data_root, "train", shard_dir, None, None, strict=strict,
                                             train_emg_length=train_emg_length)
    va, va_ids, _, _ = _packed_partition(data_root, "valid", shard_dir, s2i, m2i, strict=strict,
                                         train_emg_length=train_emg_length)
    te, te_ids, _, _ = _packed_partition(data_root, "test", shard_dir, s2i, m2i, filter_by_length=False, strict=strict,
                                         train_emg_length=train_emg_length)
    check_no_data_overlap([tr_ids, va_ids, te_ids])
    out = loaders_from_samples(tr, va, te, batch_size, train_emg_length, hopsize, device, hop_aligned_emg, rank, world_size)
    for ld in out:
        ld.session_id_to_idx, ld.speaking_mode_id_to_idx = s2i, m2i
    return out


def loaders_via_config(cfg, device="cuda", hop_aligned_emg: bool = True, rank: int = 0, world_size: int = 1):
    """loader.py:92-109 (``train_emg_length = chunk_size``: no autoregressive extra input).

    ``hop_aligned_emg`` (default on for the trainer's loaders): a chunk size that is not a multiple of the hop (the
    shipped 200) makes the reference crop 12 feature frames but 200 EMG samples -- its ``torch.stack`` then fails
    whenever the last possible window of an exact-length utterance is drawn, and its losses fail on the 200-vs-192
    mismatch anyway.  The trainer crops real EMG to ``(chunk // 16) * 16`` samples, so its loaders
    cut exactly that; pass ``False`` for the reference's literal crop."""
    chunksize = cfg.train.chunk_size
    return loaders(data_root=Path(cfg.data.dataset_root), strict=cfg.data.strict, chunksize=chunksize, hopsize=K.HOPSIZE,
                   train_emg_length=chunksize, batch_size=cfg.train.batch_size, device=device,
                   hop_aligned_emg=hop_aligned_emg, rank=rank, world_size=world_size)
