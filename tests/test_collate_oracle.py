"""CPU: the collate restatement (oracle/collate_oracle.py) against the fixtures the reference's own ste_gan_collate
produced (tests/golden/collate_*.npz), plus the host-side planning logic of the device loaders."""
import numpy as np
import pytest
import torch

from oracle import collate_oracle as CO
from tests.golden import cases


def _check(out, g, part):
    for k, v in out.items():
        key = f"{part}.{k}"
        if key in g:
            assert np.array_equal(v.numpy(), g[key]), key
        else:
            np.testing.assert_allclose(cases.poscheck(v), g[key + ".poscheck"], rtol=1e-12, err_msg=key)


@pytest.mark.parametrize("name", list(cases.COLLATE_CASES))
def test_collate_oracle_matches_reference_fixture(golden, name):
    g = golden(name)
    frames, extra, emg_len, n_mfcc, seed = cases.COLLATE_CASES[name]
    samples = cases.collate_samples(name)
    torch.manual_seed(seed)
    _check(CO.collate(samples, "train", emg_len, 16), g, "train")
    _check(CO.collate(samples, "valid", emg_len, 16), g, "valid")


def test_collate_oracle_test_partition_and_errors():
    samples = cases.collate_samples("collate_c200")
    out = CO.collate(samples[:1], "test", 200, 16)
    assert out["REAL_EMG"].shape == (1, 16 * 40 + 8, 8) and out["MFCCS"].shape == (1, 80, 25)
    with pytest.raises(ValueError):
        CO.collate(samples, "dev", 200, 16)
    with pytest.raises(AssertionError):
        CO.collate(samples, "test", 200, 16)
    with pytest.raises(RuntimeError):                      # a window past the end: torch.stack refuses ragged crops
        CO.collate(samples, "train", 200, 16, start_frames=[0, 5, 0, 0, 0, 0, 0])


def test_start_frames_follow_the_global_rng():
    torch.manual_seed(3)
    a = CO.draw_start_frames([40, 12, 13, 100], 12)
    torch.manual_seed(3)
    b = [int(torch.randint(0, 1 + max(0, f - 12), (1,)).item()) for f in [40, 12, 13, 100]]
    assert a == b and a[1] == 0 and 0 <= a[2] <= 1


def test_host_draw_matches_oracle_draw():
    from ste_gan_amd.data.collate import check_partition, draw_start_frames
    torch.manual_seed(11)
    a = draw_start_frames([30, 50, 12], 12)
    torch.manual_seed(11)
    assert a == CO.draw_start_frames([30, 50, 12], 12)
    with pytest.raises(ValueError):
        check_partition("dev")


def test_store_refuses_cpu():
    from ste_gan_amd.data.collate import UtteranceStore
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        UtteranceStore(cases.collate_samples("collate_c200"), device="cpu")


def test_read_partition_layout(tmp_path):
    """Directory layout, filtering and id -> index mapping of EMGDataset (emg_dataset.py:50-151) on a tiny corpus."""
    from ste_gan_amd.data.loader import check_no_data_overlap, read_partition
    utts = {"train": [("s1__a__normal", 20), ("s2__b__normal", 10), ("s1__c__silent", 30)],
            "valid": [("s2__d__normal", 15)]}
    for part, lst in utts.items():
        for sub in ("emg", "units", "phonemes", "mfccs"):
            (tmp_path / part / sub).mkdir(parents=True)
        for uid, f in lst:
            torch.save(torch.randn(16 * f, 8), tmp_path / part / "emg" / f"{uid}.pt")
            torch.save(torch.randn(f, 256), tmp_path / part / "units" / f"{uid}.pt")
            torch.save(torch.randint(0, 48, (f,)), tmp_path / part / "phonemes" / f"{uid}.pt")
            torch.save(torch.randn(2 * f, 25), tmp_path / part / "mfccs" / f"{uid}.pt")
    tr, s2i, m2i, ids = read_partition(tmp_path, "train", train_emg_length=200)
    assert ids == ["s1__a__normal"]                        # 10-frame utterance too short, silent one filtered
    assert s2i == {"s1": 0} and m2i == {"normal": 0}
    tr2, s2i2, _, ids2 = read_partition(tmp_path, "train", train_emg_length=160, only_include_voiced=False)
    assert ids2 == ["s1__a__normal", "s1__c__silent", "s2__b__normal"] and s2i2 == {"s1": 0, "s2": 1}
    assert [int(x["SESSION_INDEX"]) for x in tr2] == [0, 0, 1] and tr2[0]["MFCCS"].shape == (40, 25)
    va, _, _, va_ids = read_partition(tmp_path, "valid", s2i2, {"normal": 0, "silent": 1}, train_emg_length=160)
    assert int(va[0]["SESSION_INDEX"]) == 1
    check_no_data_overlap([ids2, va_ids])
    with pytest.raises(ValueError, match="Intersecting"):
        check_no_data_overlap([ids2, ids])


def test_shard_roundtrip(tmp_path):
    """pack -> shard file -> memory-mapped load gives the same arrays, row counts and ids; bad files are refused."""
    from ste_gan_amd.data import load_packed, pack_samples, save_packed
    samples = cases.collate_samples("collate_c200")
    for i, s in enumerate(samples):
        s["UTT_ID"] = f"s{i % 2}__u{i}__normal"
    pk = pack_samples(samples)
    assert pk["emg"].shape == (sum(16 * f + 8 for f in cases.COLLATE_CASES["collate_c200"][0]), 8)
    save_packed(pk, tmp_path / "train.shard")
    q = load_packed(tmp_path / "train.shard")
    for k, v in pk.items():
        assert (torch.equal(v, q[k]) if torch.is_tensor(v) else v == q[k]), k
    for s in samples:
        s["MFCCS"] = None
    pk2 = pack_samples(samples)
    save_packed(pk2, tmp_path / "nomfcc.shard")
    assert load_packed(tmp_path / "nomfcc.shard")["mfcc"] is None
    (tmp_path / "junk.shard").write_bytes(b"not a shard")
    with pytest.raises(ValueError, match="not a ste-gan shard"):
        load_packed(tmp_path / "junk.shard")
