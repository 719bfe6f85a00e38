"""CPU: ``data.loader.read_partition`` / ``check_no_data_overlap`` against a fixture recorded from the reference's own
``EMGDataset`` (data/emg_dataset.py:50-151,220-262) over the synthetic on-disk corpus of ``cases.write_corpus``
(tests/golden/make_golden.py::dataset_cases, tests/golden/dataset_corpus.json): which utterances every partition keeps and
in what order, how many it filters, the session / speaking-mode id -> index maps (train's maps handed to valid / test),
per-item indices and lengths, the strict length check and the overlap check."""
import json
import os

import pytest

import cases


@pytest.fixture(scope="module")
def ref():
    return json.load(open(os.path.join(cases.GOLDEN_DIR, "dataset_corpus.json")))


def _check(got, want, n_files):
    samples, s2i, m2i, ids = got
    assert ids == want["utt_ids"]
    assert n_files - len(ids) == want["num_filtered"]
    assert s2i == want["session_id_to_idx"] and m2i == want["speaking_mode_id_to_idx"]
    assert [int(x["SESSION_INDEX"]) for x in samples] == want["session_index"]
    assert [int(x["SPEAKING_MODE_IDX"]) for x in samples] == want["speaking_mode_index"]
    assert [len(x["REAL_EMG"]) for x in samples] == want["emg_lengths"]
    assert [int(x["SPEECH_UNITS"].shape[0]) for x in samples] == want["unit_frames"]
    assert [int(x["MFCCS"].shape[0]) for x in samples] == want["mfcc_frames"]
    assert [float(x["REAL_EMG"].double().sum()) for x in samples] == pytest.approx(want["emg_checksum"], rel=1e-12)
    assert len(s2i) == want["num_sessions"] and len(m2i) == want["num_speaking_modes"]
    assert all(x["REAL_EMG"].shape[-1] == want["num_emg_channels"] for x in samples)


def test_partitions_match_the_reference_dataset(tmp_path, ref):
    from ste_gan_amd.data.loader import check_no_data_overlap, read_partition
    cases.write_corpus(tmp_path, strict_ok=True)
    L = cases.CORPUS_EMG_LENGTH
    n = {p: len(v) for p, v in cases.CORPUS.items()}
    tr = read_partition(tmp_path, "train", strict=True, train_emg_length=L)            # flags of data/loader.py:28-48
    _check(tr, ref["train"], n["train"])
    va = read_partition(tmp_path, "valid", tr[1], tr[2], strict=True, train_emg_length=L)
    _check(va, ref["valid"], n["valid"])
    te = read_partition(tmp_path, "test", tr[1], tr[2], filter_by_length=False, strict=True, train_emg_length=L)
    _check(te, ref["test"], n["test"])
    check_no_data_overlap([tr[3], va[3], te[3]])
    _check(read_partition(tmp_path, "train", only_include_voiced=False, filter_by_length=False, train_emg_length=L),
           ref["train_all"], n["train"])
    _check(read_partition(tmp_path, "train", only_include_voiced=False, only_include_silent=True, train_emg_length=L),
           ref["train_silent"], n["train"])
    assert ref["overlap_raises"]
    with pytest.raises(ValueError, match="Intersecting"):
        check_no_data_overlap([tr[3], tr[3]])
    with pytest.raises(AssertionError):
        read_partition(tmp_path, "train", only_include_voiced=True, only_include_silent=True)


def test_strict_length_check(tmp_path, ref):
    """emg_dataset.py:254-261: with ``strict`` an utterance whose EMG is not exactly 16 x frames long fails an assertion
    (the reference on first access, here while the partition is read); without it the corpus loads."""
    from ste_gan_amd.data.loader import read_partition
    cases.write_corpus(tmp_path, strict_ok=False)
    L = cases.CORPUS_EMG_LENGTH
    assert ref["strict_violations"] == ["s2__u05__normal"]
    with pytest.raises(AssertionError):
        read_partition(tmp_path, "train", strict=True, train_emg_length=L)
    samples, _, _, ids = read_partition(tmp_path, "train", strict=False, train_emg_length=L)
    assert ids == ref["train"]["utt_ids"]
    assert len(samples[ids.index("s2__u05__normal")]["REAL_EMG"]) == 16 * 20 + 8
