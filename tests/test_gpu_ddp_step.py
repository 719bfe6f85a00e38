"""-m gpu: the data-parallel train step on the GPU kernels (VERDICT r02 "multi-rank correctness").

Two ranks (gloo, both on the box's one card) each run ``GANTrainStep`` on HALF of a detfill batch -- eagerly, eagerly with the
overlapped schedule of the single-graph capture (the side branch runs on across the D-gradient exchange), and as the three
replayed hipGraphs (the side branch split over the first two) around the two gradient exchanges -- for two optimiser steps.
(The single graph with CAPTURED collectives needs RCCL, one rank per GPU: its replay is checked against the golden
fixture with a 1-rank group, tests/test_gpu_train_step.py.)  Checked: the replicas' parameters
are bit-identical, and equal the single-process step on the WHOLE batch (every loss is a batch mean; the exchange
averages the gradients) to 2e-5 in every tensor's norm and 2e-4 element-wise.  The ranks end through
``parallel.shutdown`` and must exit with code 0 (no os._exit escape)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import shapes as oshapes
from oracle.detfill import det_input, det_ints, fill_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 2
CH, ENC = 48, (64, 2, 128)
B, T = 4, 6


def full_batch(s):
    tag = f"ddpstep.s{s}"
    su = det_input(tag + ".su", (B, T, 256))
    return {"su": su, "x_real": det_input(tag + ".x", (B, 16 * T, 8), "tanh"), "sess": det_ints(tag + ".sess", (B,), 17),
            "phonemes": det_ints(tag + ".ph", (B, T), 48)}


def to_device_batch(b):
    from ste_gan_amd.constants import DataType
    return {DataType.REAL_EMG: b["x_real"].cuda(), DataType.SPEECH_UNITS: b["su"].cuda(),
            DataType.SESSION_INDEX: b["sess"].cuda(), DataType.SPEAKING_MODE_INDEX: torch.zeros_like(b["sess"]).cuda(),
            DataType.PHONEMES: b["phonemes"].cuda()}


def build_nets():
    from test_gpu_train_step import make_cfg
    from ste_gan_amd.losses.emg_encoder_loss import EMGEncoderLoss
    from ste_gan_amd.losses.time_domain_loss import MultiTimeDomainFeatureLoss
    from ste_gan_amd.models.discriminator import init_emg_discriminators
    from ste_gan_amd.models.emg_encoder import EMGEncoderTransformer
    from ste_gan_amd.models.generator import init_emg_generator
    from ste_gan_amd.train import GANTrainStep, make_optimizers
    cfg = make_cfg(CH)
    netG = init_emg_generator(cfg); netG.load_state_dict(fill_state_dict(oshapes.generator_shapes("SPEECH_UNITS", CH)))
    netD = init_emg_discriminators(cfg); netD.load_state_dict(fill_state_dict(oshapes.discriminator_shapes(True)))
    ms, nl, ff = ENC
    enc = EMGEncoderTransformer(8, 256, 48, model_size=ms, num_transformer_layers=nl, dim_feedforward=ff)
    enc.load_state_dict(fill_state_dict(oshapes.encoder_shapes(ms, 3, nl, dim_feedforward=ff)))
    netG, netD, enc = netG.cuda(), netD.cuda(), enc.cuda().eval()
    optG, optD = make_optimizers(netG, netD)
    step = GANTrainStep(cfg, netG, netD, EMGEncoderLoss(enc), MultiTimeDomainFeatureLoss(8), optG, optD)
    return netG, netD, optG, optD, step


def _run_ranks(tmp_path, mode, port, **extra_env):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "ddp_step_worker.py"), str(tmp_path), mode]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("DONE") == 2, r.stdout[-2000:]
    return [torch.load(tmp_path / f"{mode}_rank{k}.pt") for k in (0, 1)]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["eager", "eager_overlap", "graph"])
def test_two_gpu_step_replicas_equal_the_full_batch_step(tmp_path, mode):
    r0, r1 = _run_ranks(tmp_path, mode, {"eager": 29771, "graph": 29772, "eager_overlap": 29773}[mode])
    netG, netD, optG, optD, step = build_nets()
    assert step.comm is None
    for s in range(STEPS):
        step(to_device_batch(full_batch(s)))
    torch.cuda.synchronize()
    ref = {"g": {k: v.cpu() for k, v in netG.state_dict().items()}, "d": {k: v.cpu() for k, v in netD.state_dict().items()}}
    for net in ("g", "d"):
        for k, v in ref[net].items():
            assert torch.equal(r0[net][k], r1[net][k]), f"replicas diverged on {net}.{k}"
            a, b = float(r0[net][k].double().norm()), float(v.double().norm())
            assert abs(a - b) <= 2e-5 * max(b, 1e-30), f"{net}.{k}: norm {a} vs {b}"
            if v.is_floating_point():
                torch.testing.assert_close(r0[net][k], v, rtol=2e-4, atol=2e-6, msg=lambda m: f"{net}.{k}: {m}")
    assert np.isfinite(np.array(r0["losses"])).all()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_grad_scaler_skips_on_every_rank_when_one_rank_overflows(tmp_path, mode):
    """fp16 mode under data parallelism: rank 1's first batch overflows the fp16 operands; the SUM all-reduce carries its
    non-finite gradients to rank 0, whose own were finite -- both GradScalers (checked BEHIND the exchange) skip the same
    optimiser steps of iteration 0, halve the scale once, apply iteration 1, and the replicas stay bit-identical."""
    r0, r1 = _run_ranks(tmp_path, mode, {"eager": 29775, "graph": 29776}[mode], DDP_STEP_DTYPE="f16", DDP_STEP_POISON="1")
    assert r0["scaler"] == r1["scaler"], (r0["scaler"], r1["scaler"])
    sc = r0["scaler"]
    assert sc["skipped"] >= 1 and sc["scale"] == 32768.0, sc             # one back-off from torch's 65536
    assert sc["d_steps"] + sc["g_steps"] + sc["skipped"] == 2 * STEPS, sc  # every optimiser step either applied or skipped
    for net in ("g", "d"):
        for k in r0[net]:
            assert torch.equal(r0[net][k], r1[net][k]), f"replicas diverged on {net}.{k}"
            assert torch.isfinite(r0[net][k].float()).all(), f"{net}.{k} is not finite: a poisoned step was applied"
