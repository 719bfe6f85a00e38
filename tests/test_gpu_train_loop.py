"""GPU, end to end: corpus directory -> resident stores -> device loaders -> train() (steps, validation, best-model and
numbered checkpoints) -> resume from the checkpoint directory.  Small widths, a synthetic corpus on disk in the
reference's layout (data/emg_dataset.py:50-151)."""
import os

import pytest
import torch
import yaml

from oracle import shapes as oshapes
from oracle.detfill import det_input, det_ints, fill_state_dict

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _corpus(root):
    parts = {"train": [("s1__u%02d__normal" % i, 30 + 7 * i) for i in range(6)] + [("s2__v%02d__normal" % i, 25 + i) for i in range(3)]
             + [("s1__short__normal", 8), ("s2__sil__silent", 40)],
             "valid": [("s1__w%02d__normal" % i, 28 + i) for i in range(3)],
             "test": [("s2__x00__normal", 20), ("s1__x01__normal", 5)]}
    for part, utts in parts.items():
        for sub in ("emg", "units", "phonemes", "mfccs"):
            (root / part / sub).mkdir(parents=True)
        for uid, f in utts:
            torch.save(det_input(uid + ".emg", (16 * f, 8), "tanh"), root / part / "emg" / f"{uid}.pt")
            torch.save(det_input(uid + ".su", (f, 256)), root / part / "units" / f"{uid}.pt")
            torch.save(det_ints(uid + ".ph", (f,), 48), root / part / "phonemes" / f"{uid}.pt")
            torch.save(det_input(uid + ".mf", (2 * f, 25)), root / part / "mfccs" / f"{uid}.pt")


def _cfg(tmp_path, max_steps):
    from ste_gan_amd.train_utils import DictConfig
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs/ste_gan_base_gantts.yaml")))
    cfg["data"] = yaml.safe_load(open(os.path.join(ROOT, "configs/data/gaddy_and_klein_corpus.yaml")))
    cfg["data"]["dataset_root"] = str(tmp_path / "corpus")
    cfg["data"]["strict"] = True
    cfg["model"]["params"] = {"channels": 32}
    cfg["emg_encoder"] = {"type": "EMGEncoderTransformer",
                          "params": {"model_size": 64, "num_transformer_layers": 2, "dim_feedforward": 128}}
    cfg["train"].update(batch_size=4, chunk_size=200, max_steps=max_steps, interval_log=1, interval_valid=2, interval_save=2)
    return DictConfig(cfg)


def test_train_loop_end_to_end_and_resume(tmp_path):
    from ste_gan_amd.data import loaders_via_config
    from ste_gan_amd.train import train
    _corpus(tmp_path / "corpus")
    enc_ckpt = tmp_path / "enc.pt"
    torch.save(fill_state_dict(oshapes.encoder_shapes(64, 3, 2, dim_feedforward=128)), enc_ckpt)
    cfg = _cfg(tmp_path, max_steps=3)

    tr, va, te = loaders_via_config(cfg)
    assert len(tr.store) == 9 and len(va.store) == 3 and len(te.store) == 2       # short + silent utterances filtered
    assert (len(tr), len(va), len(te)) == (3, 1, 2)
    b = next(iter(tr))
    assert b["REAL_EMG"].shape == (4, 192, 8) and b["SPEECH_UNITS"].shape == (4, 12, 256) and b["REAL_EMG"].is_cuda

    out = tmp_path / "run"
    out.mkdir()
    train(cfg, out, None, "cuda", debug=False, emg_enc_ckpt=enc_ckpt)
    names = {p.name for p in out.iterdir()}
    assert {".done", "netG-final.pt", "netD-final.pt", "checkpoint-final.pt", "netG-00000002.pt", "checkpoint-00000002.pt",
            "best_netG.pt", "best_netD.pt"} <= names
    ck = torch.load(out / "checkpoint-00000002.pt")
    assert ck["steps"] == 2 and set(ck) == {"epoch", "steps", "optG", "optD"}
    sdG = torch.load(out / "netG-00000002.pt")
    assert list(sdG) == list(oshapes.generator_shapes("SPEECH_UNITS", 32))          # reference key layout
    assert all(torch.isfinite(v).all() for v in sdG.values())

    # resume: picks the highest NUMBERED checkpoint (step 2), continues to the new max_steps
    (out / ".done").unlink()
    cfg2 = _cfg(tmp_path, max_steps=5)
    train(cfg2, out, out, "cuda", debug=False, emg_enc_ckpt=enc_ckpt)
    ck4 = torch.load(out / "checkpoint-00000004.pt")
    assert ck4["steps"] == 4
    st = next(iter(ck4["optG"]["state"].values()))
    # checkpoint 2 holds 3 optimiser steps (loop counter 0, 1, 2); the resumed loop starts again AT counter 2
    # (train.py:84-91,165: `steps` is restored, then the next batch runs) and runs counters 2, 3, 4
    assert float(st["step"]) == 6.0


def test_loaders_from_shard_cache_match_loaders_from_pt_files(tmp_path):
    """First call reads the per-utterance .pt files and writes one shard per partition; the second call starts from the
    memory-mapped shards: same stores, same batches under the same seed."""
    from ste_gan_amd.data import loaders
    _corpus(tmp_path / "corpus")
    kw = dict(strict=True, train_emg_length=192, batch_size=4)
    a = loaders(tmp_path / "corpus", shard_dir=tmp_path / "shards", **kw)
    assert sorted(p.name for p in (tmp_path / "shards").iterdir()) == ["test.shard", "train.shard", "valid.shard"]
    b = loaders(tmp_path / "corpus", shard_dir=tmp_path / "shards", **kw)
    c = loaders(tmp_path / "corpus", **kw)
    for la, lb, lc in zip(a, b, c):
        assert la.store.frames == lb.store.frames == lc.store.frames and lb.store.utt_ids == lc.store.utt_ids
        assert torch.equal(la.store.session_idx, lb.store.session_idx) and torch.equal(lb.store.session_idx, lc.store.session_idx)
        outs = []
        for ld in (la, lb, lc):
            torch.manual_seed(99)
            outs.append(list(ld))
        for x, y, z in zip(*outs):
            for k in x:
                if x[k] is not None:
                    assert torch.equal(x[k], y[k]) and torch.equal(y[k], z[k]), k


def test_graph_replay_in_the_training_loop_matches_eager_launches(tmp_path, monkeypatch):
    """train() replays a captured hipGraph for every batch shape it has seen before (the first batch of a shape runs eagerly
    and is that batch's step): same weights after 6 steps as with STEGAN_TRAIN_GRAPH=0, to fp32-atomics noise, and the eager
    validation passes in between see current weights (best-model file written in both runs)."""
    from ste_gan_amd.train import train
    _corpus(tmp_path / "corpus")
    enc_ckpt = tmp_path / "enc.pt"
    torch.save(fill_state_dict(oshapes.encoder_shapes(64, 3, 2, dim_feedforward=128)), enc_ckpt)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("STEGAN_TRAIN_GRAPH", mode)
        torch.manual_seed(1234)
        import random
        import numpy as np
        random.seed(1234); np.random.seed(1234)
        out = tmp_path / f"run{mode}"
        out.mkdir()
        train(_cfg(tmp_path, max_steps=5), out, None, "cuda", debug=False, emg_enc_ckpt=enc_ckpt)
        res[mode] = (torch.load(out / "netG-final.pt"), torch.load(out / "netD-final.pt"))
        assert (out / "best_netG.pt").exists()
    for a, b in zip(res["1"], res["0"]):
        assert list(a) == list(b)
        for k in a:
            ta, tb = a[k].float(), b[k].float()
            assert torch.isfinite(ta).all()
            # AdamW moves a weight by ~lr = 2e-4 per step whatever its gradient's size, so atomics noise in a near-zero
            # gradient can flip single updates (and the spectral-norm vectors follow the weights): compare tensor norms
            rel = float((ta - tb).norm()) / max(float(tb.norm()), 1e-12)
            assert rel <= 1e-2, (k, rel)


def test_train_loop_in_fp16_mode(tmp_path):
    """cfg.train.mixed_precision_dtype = "f16": the trainer runs on the fp16 build of the kernels with the device-side
    GradScaler (BASELINE config 5's arithmetic; eager first batch, graph replays after), leaves finite weights and the
    reference's files behind, and restores the compute mode and the default library on the way out."""
    from ste_gan_amd import _lib, ops
    from ste_gan_amd.train import train
    _corpus(tmp_path / "corpus")
    enc_ckpt = tmp_path / "enc.pt"
    torch.save(fill_state_dict(oshapes.encoder_shapes(64, 3, 2, dim_feedforward=128)), enc_ckpt)
    cfg = _cfg(tmp_path, max_steps=5)
    cfg.train.mixed_precision_dtype = "f16"
    out = tmp_path / "run16"
    out.mkdir()
    train(cfg, out, None, "cuda", debug=False, emg_enc_ckpt=enc_ckpt)
    assert ops.COMPUTE_DTYPE == "f32" and _lib.lib().stegan_operand_type() == b"bf16"
    assert {".done", "netG-final.pt", "netD-final.pt", "checkpoint-final.pt"} <= {p.name for p in out.iterdir()}
    for f in ("netG-final.pt", "netD-final.pt"):
        assert all(torch.isfinite(v.float()).all() for v in torch.load(out / f).values())
    cfg.train.mixed_precision_dtype = "fp8"
    with pytest.raises(ValueError, match="mixed_precision_dtype"):
        train(cfg, out, None, "cuda", debug=False, emg_enc_ckpt=enc_ckpt)


def test_train_loop_two_ranks_with_graph_replay(tmp_path):
    """train() under torch.distributed (2 ranks on the box's one card, gloo): rank-sharded loaders, the three-graph replay of
    the step around the two gradient all-reduces, validation / checkpoints on rank 0 behind barriers."""
    import subprocess
    import sys
    _corpus(tmp_path / "corpus")
    torch.save(fill_state_dict(oshapes.encoder_shapes(64, 3, 2, dim_feedforward=128)), tmp_path / "enc.pt")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29761", os.path.join(ROOT, "tests", "ddp_train_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert "RANK 0 DONE" in r.stdout and "RANK 1 DONE" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    out = tmp_path / "run_ddp"
    names = {p.name for p in out.iterdir()}
    assert {".done", "netG-final.pt", "netD-final.pt", "checkpoint-final.pt"} <= names
    sd = torch.load(out / "netG-final.pt")
    assert all(torch.isfinite(v).all() for v in sd.values())


def test_validation_pass_vs_reference_fixture(golden):
    """``train.validate`` (ste_gan/train.py:292-350) over a list of batches standing in for valid_loader, against the
    fixture make_golden.py::valid_cases recorded from the reference's modules: the four mean errors within 1e-4, the
    phoneme counters -- hence both accuracies -- exact.  The third batch is a 4 800-sample utterance (multi-TD row
    buffers in the global scratch)."""
    import numpy as np
    import cases
    from oracle import shapes as oshapes
    from oracle.detfill import fill_state_dict
    from ste_gan_amd.constants import DataType
    from ste_gan_amd.losses.emg_encoder_loss import EMGEncoderLoss
    from ste_gan_amd.losses.time_domain_loss import MultiTimeDomainFeatureLoss
    from ste_gan_amd.models.emg_encoder import EMGEncoderTransformer
    from ste_gan_amd.models.generator import init_emg_generator
    from ste_gan_amd.train import validate
    from test_gpu_train_step import make_cfg
    name = "valid_c48"
    g = golden(name)
    ch, (ms, nl, ff), _ = cases.VALID_CASES[name]
    netG = init_emg_generator(make_cfg(ch)); netG.load_state_dict(fill_state_dict(oshapes.generator_shapes("SPEECH_UNITS", ch)))
    enc = EMGEncoderTransformer(8, 256, 48, model_size=ms, num_transformer_layers=nl, dim_feedforward=ff)
    enc.load_state_dict(fill_state_dict(oshapes.encoder_shapes(ms, 3, nl, dim_feedforward=ff)))
    netG, enc = netG.cuda(), enc.cuda().eval()
    loader = [{DataType.REAL_EMG: b["x_real"], DataType.SPEECH_UNITS: b["su"], DataType.SESSION_INDEX: b["sess"],
               DataType.SPEAKING_MODE_INDEX: torch.zeros_like(b["sess"]), DataType.PHONEMES: b["phonemes"]}
              for b in cases.valid_batches(name)]
    val = validate(netG, MultiTimeDomainFeatureLoss(8), EMGEncoderLoss(enc), loader, DataType.SPEECH_UNITS, torch.device("cuda"))
    for k in ("speech_unit", "multi_td", "phoneme", "waveform"):
        assert abs(val[k] - float(g[k])) <= 1e-4 * abs(float(g[k])), (k, val[k], float(g[k]))
    assert val["phoneme_accuracy_avg"] == float(g["phoneme_accuracy_avg"])
    assert val["phoneme_accuracy_avg_no_sil"] == float(g["phoneme_accuracy_avg_no_sil"])
