"""GPU: the batch-assembly kernel (stegan_assemble_batch via UtteranceStore / ste_gan_collate / DeviceLoader) against
the reference fixtures and the CPU oracle.  Integer / copy work: bit-exact."""
import numpy as np
import pytest
import torch

from oracle import collate_oracle as CO
from tests.golden import cases

pytestmark = pytest.mark.gpu


def _same(out, ref):
    for k, v in ref.items():
        got = out[k]
        assert got.is_cuda and got.dtype == v.dtype and tuple(got.shape) == tuple(v.shape), k
        assert torch.equal(got.cpu(), v), k


@pytest.mark.parametrize("name", list(cases.COLLATE_CASES))
def test_collate_matches_reference_fixture(golden, name):
    from ste_gan_amd.data import ste_gan_collate
    g = golden(name)
    frames, extra, emg_len, n_mfcc, seed = cases.COLLATE_CASES[name]
    samples = cases.collate_samples(name)
    for part in ("train", "valid"):
        torch.manual_seed(seed)
        out = ste_gan_collate(samples, partition=part, emg_train_length=emg_len, chunksize=emg_len, hopsize=16)
        for k, v in out.items():
            key = f"{part}.{k}"
            if key in g:
                assert np.array_equal(v.cpu().numpy(), g[key]), key
            else:
                np.testing.assert_allclose(cases.poscheck(v.cpu()), g[key + ".poscheck"], rtol=1e-12, err_msg=key)


def test_collate_test_partition_and_errors():
    from ste_gan_amd.data import ste_gan_collate
    samples = cases.collate_samples("collate_c200")
    _same(ste_gan_collate(samples[2:3], partition="test", emg_train_length=200), CO.collate(samples[2:3], "test", 200, 16))
    with pytest.raises(ValueError):
        ste_gan_collate(samples, partition="dev")
    with pytest.raises(AssertionError):
        ste_gan_collate(samples, partition="test")


def test_assemble_random_windows_vs_oracle():
    """Ragged corpus (odd MFCC width, EMG longer than 16*frames, repeated utterances in one batch, first and last
    possible windows) against the oracle's slicing."""
    from ste_gan_amd.data import UtteranceStore
    g = torch.Generator().manual_seed(5)
    frames = [12, 13, 50, 31, 200, 12, 77]
    samples = [{"REAL_EMG": torch.randn(16 * f + (i % 3) * 5, 8, generator=g), "SPEECH_UNITS": torch.randn(f, 256, generator=g),
                "PHONEMES": torch.randint(0, 48, (f,), generator=g), "SESSION_INDEX": torch.tensor(i % 5),
                "SPEAKING_MODE_IDX": torch.tensor(i % 2), "MFCCS": torch.randn(2 * f + 1, 25, generator=g)}
               for i, f in enumerate(frames)]
    store = UtteranceStore(samples)
    utts = [4, 0, 2, 4, 6, 1, 5, 3, 2]
    starts = [188, 0, 38, 0, 65, 1, 0, 19, 7]
    out = store.assemble(utts, starts, 12, 192, 16)
    _same(out, CO.collate([samples[u] for u in utts], "train", 192, 16, start_frames=starts))
    # windows that run off an utterance are refused on the host, like the reference's torch.stack
    with pytest.raises(RuntimeError, match="equal size"):
        store.assemble([1], [2], 12, 192, 16)
    with pytest.raises(RuntimeError, match="equal size"):
        store.assemble([0], [0], 12, 200, 16)              # 12-frame utterance, 192 samples, asked for 200
    with pytest.raises(IndexError):
        store.assemble([7], [0], 12, 192, 16)
    # an unvalidated bad window never reads outside its utterance: the missing tail comes back as zeros
    bad = store.assemble(torch.tensor([1], dtype=torch.int32, device="cuda"), torch.tensor([2], dtype=torch.int32, device="cuda"),
                         12, 192, 16, validate=False)
    assert torch.equal(bad["SPEECH_UNITS"][0, :11].cpu(), samples[1]["SPEECH_UNITS"][2:13])
    assert float(bad["SPEECH_UNITS"][0, 11:].abs().sum()) == 0.0 and int(bad["PHONEMES"][0, 11]) == 0


def test_store_without_mfccs():
    from ste_gan_amd.data import UtteranceStore
    samples = cases.collate_samples("collate_c200")
    for s in samples:
        s["MFCCS"] = None
    out = UtteranceStore(samples).assemble([0, 2], [3, 50], 12, 192, 16)
    assert out["MFCCS"] is None
    assert torch.equal(out["REAL_EMG"][1].cpu(), samples[2]["REAL_EMG"][800:992])


def test_device_loader_epochs():
    """train: every utterance once per epoch in RandomSampler order with collate's draws; valid: sequential from frame
    0; test: whole utterances; the last batch is short (drop_last=False)."""
    from ste_gan_amd.data import loaders_from_samples
    samples = cases.collate_samples("collate_c200")
    tr, va, te = loaders_from_samples(samples, samples[:3], samples[:2], batch_size=3, train_emg_length=200, hop_aligned_emg=True)
    assert (len(tr), len(va), len(te)) == (3, 1, 2)
    torch.manual_seed(21)
    got = list(tr)
    torch.manual_seed(21)
    order = torch.randperm(len(samples)).tolist()
    assert [b["REAL_EMG"].shape[0] for b in got] == [3, 3, 1]
    for i, b in enumerate(got):
        sub = [samples[u] for u in order[3 * i:3 * i + 3]]
        ref = CO.collate(sub, "train", 192, 16)            # hop-aligned: 12 frames <-> 192 samples
        _same(b, ref)
    _same(next(iter(va)), CO.collate(samples[:3], "valid", 192, 16))
    tb = list(te)
    assert len(tb) == 2 and tb[1]["REAL_EMG"].shape == (1, 16 * 13 + 8, 8)
    _same(tb[1], CO.collate(samples[1:2], "test", 200, 16))
