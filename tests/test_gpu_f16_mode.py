"""-m gpu: the fp16 operand mode end to end (``ops.set_compute_dtype("f16")`` = libstegan_hip_f16.so + the device-side
GradScaler) -- BASELINE config 5 as the reference states it: ``torch.autocast(dtype=torch.float16)`` with
``torch.cuda.amp.GradScaler()`` (ste_gan/train.py:151,181,198-199,204,266-268).

What differs from the reference's autocast is stated in DESIGN.md: activations, accumulators, masters and optimiser state
stay fp32 here -- only the two operands of each matrix-core contraction are rounded to fp16 -- so the comparison is with the
oracle run on fp16-rounded operands (``oracle.bf16_operands(torch.float16)``) and the same GradScaler restated on the CPU.

* layer by layer against that oracle (the 335 tensors of test_gpu_bf16_emulation.py, 1e-4);
* the loss scaler: a step whose scaled gradients overflow fp16 is SKIPPED on the device -- parameters, moments and Adam step
  counts bit-identical to before, the scale halved --, clean steps grow the scale every ``growth_interval`` iterations, and the
  decisions match the oracle's GradScaler step for step;
* two steps of the small fixture and one step of the config-5 geometry (MFCC input, T_emg = 400, full widths, B = 8) against
  the oracle; B = 128 through size-independent properties; eager and replayed as a hipGraph (bit-identical, scale included).
"""
import numpy as np
import pytest
import torch

import cases
import test_gpu_bf16_emulation as E
import test_gpu_conv_bf16 as T
from oracle import shapes as oshapes
from oracle import stegan_oracle as O
from oracle.detfill import fill_state_dict

pytestmark = pytest.mark.gpu
F16 = ("f16", torch.float16)


@pytest.fixture(autouse=True)
def f16_everywhere(monkeypatch):
    from ste_gan_amd import ops
    monkeypatch.setattr(E, "OPERAND", F16)
    monkeypatch.setattr(T, "OPERAND", F16)
    yield
    ops.set_compute_dtype("f32")


def test_every_layer_matches_the_f16_operand_oracle():
    from ste_gan_amd import ops
    ops.set_compute_dtype("f16")
    E.test_every_layer_matches_the_bf16_operand_oracle(ops)


def test_full_discriminator_step_vs_f16_operand_oracle():
    from ste_gan_amd import ops
    ops.set_compute_dtype("f16")
    E.test_full_discriminator_step_vs_operand_oracle(ops)


def _small(scaler_kw=None):
    from ste_gan_amd import ops
    from ste_gan_amd.optim import GradScaler
    from test_gpu_train_step import build
    ops.set_compute_dtype("f16")
    name = next(iter(cases.STEP_CASES))
    cfg, netG, netD, optG, optD, step = build(name)
    assert step.scaler is not None, "the f16 mode must come with a GradScaler"
    if scaler_kw is not None:
        step.scaler = GradScaler(optG.flat.flat.device, **scaler_kw)
    return name, netG, netD, optG, optD, step


def _oracle_state(name):
    ch, (ms, nl, ff), B, T, steps = cases.STEP_CASES[name]
    g = fill_state_dict(oshapes.generator_shapes("SPEECH_UNITS", ch)); d = fill_state_dict(oshapes.discriminator_shapes(True))
    e = fill_state_dict(oshapes.encoder_shapes(ms, 3, nl, dim_feedforward=ff))
    dt = {k: v for k, v in d.items() if not O.is_sn_buffer(d, k)}
    return g, d, e, dt, O.adamw_init(g), O.adamw_init(dt)


def test_overflowing_step_is_skipped_on_the_device():
    """A scale of 2^40 drives the scaled gradients far beyond fp16's 65504: inf operands, non-finite gradients, and BOTH
    optimiser steps are skipped -- nothing but the scale (halved) and the skip counter changes."""
    from test_gpu_train_step import to_batch
    name, netG, netD, optG, optD, step = _small(dict(init_scale=2.0 ** 40))
    before = {k: v.clone() for net in (netG, netD) for k, v in net.state_dict().items()}
    sn = {k for k in netD.state_dict() if k.endswith("weight_u") or (k.endswith("weight_v") and k[:-1] + "orig" in netD.state_dict())}
    out = step(to_batch(cases.step_batch(name, 0)))
    st = step.read_stats(out)
    torch.cuda.synchronize()
    assert step.scaler.get_scale() == 2.0 ** 39 and step.scaler.skipped_steps() == 2
    assert optD.device_step_count() == 0 and optG.device_step_count() == 0
    assert optG.step_count == 1 and optG.state_dict()["state"] == {}          # checkpoints carry the APPLIED step count
    assert float(optG.exp_avg.abs().max()) == 0.0 and float(optD.exp_avg_sq.abs().max()) == 0.0
    for net in (netG, netD):
        for k, v in net.state_dict().items():
            if k in sn:
                continue                                   # the spectral-norm power iteration advances in every forward
            assert torch.equal(v, before[k]), f"{k} changed in a skipped step"
    assert all(np.isfinite(st[k]) for k in ("discriminator", "multi_td", "speech_unit", "phoneme")), st   # values are unscaled
    assert float(optG.flat.grad.abs().max()) == 0.0 or not optG.flat.zero_in_step


def test_scale_grows_after_clean_iterations_and_matches_the_oracle_scaler():
    """growth_interval = 2, a scale that fits: the sequence of (scale, skipped) decisions equals the oracle's GradScaler, and
    the losses of every step agree with the oracle run on fp16-rounded operands."""
    from test_gpu_train_step import to_batch
    kw = dict(init_scale=2.0 ** 10, growth_interval=2)
    name, netG, netD, optG, optD, step = _small(kw)
    g, d, e, dt, og, od = _oracle_state(name)
    sc = {"scale": kw["init_scale"], "growth_tracker": 0, "growth_interval": 2}
    for s in range(4):
        b = cases.step_batch(name, s % cases.STEP_CASES[name][4])
        with O.bf16_operands(torch.float16):
            o = O.train_step(g, d, e, og, od, b, scaler=sc)
        out = step(to_batch(b)); st = step.read_stats(out)
        assert o["skipped"] == (False, False)
        assert st["loss_scale"] == sc["scale"], (s, st["loss_scale"], sc)
        got = np.array([st["discriminator"], st["generator"], st["multi_td"], st["speech_unit"], st["phoneme"], st["feature_matching"]])
        want = np.array([o["loss_d"], o["loss_g"], o["loss_td"], o["loss_su"], o["loss_ph"], o["loss_fm"]])
        assert float((np.abs(got - want) / np.abs(want)).max()) <= (1e-4 if s == 0 else 3e-3), (s, got, want)
    assert sc["scale"] == 2.0 ** 12 and step.scaler.skipped_steps() == 0
    assert optD.device_step_count() == 4 and optG.device_step_count() == 4


def test_small_step_vs_f16_operand_oracle_with_the_default_scaler():
    """torch's default scaler (65536): whatever it decides -- skip or step -- the build and the oracle decide alike, and a
    step both take agrees like the bf16 one does (test_gpu_bf16_emulation.py)."""
    from test_gpu_train_step import to_batch
    name, netG, netD, optG, optD, step = _small()
    g, d, e, dt, og, od = _oracle_state(name)
    sc = {"scale": 65536.0, "growth_tracker": 0}
    for s in range(cases.STEP_CASES[name][4]):
        b = cases.step_batch(name, s)
        with O.bf16_operands(torch.float16):
            o = O.train_step(g, d, e, og, od, b, keep_grads=True, scaler=sc)
        skipped_before = step.scaler.skipped_steps()
        out = step(to_batch(b)); st = step.read_stats(out)
        print(f"\\nstep {s}: oracle skipped {o['skipped']}, build skipped {step.scaler.skipped_steps() - skipped_before}, "
              f"scale {st['loss_scale']}")
        assert step.scaler.skipped_steps() - skipped_before == sum(o["skipped"])
        assert st["loss_scale"] == sc["scale"]
        got = np.array([st["discriminator"], st["generator"], st["multi_td"], st["speech_unit"], st["phoneme"], st["feature_matching"]])
        want = np.array([o["loss_d"], o["loss_g"], o["loss_td"], o["loss_su"], o["loss_ph"], o["loss_fm"]])
        assert float((np.abs(got - want) / np.abs(want)).max()) <= (1e-4 if s == 0 else 3e-3), (s, got, want)
        assert float((out["x_pred"].cpu() - o["x_pred"]).abs().max()) <= (1e-3 if s == 0 else 3e-3)
    from test_gpu_train_step import norms
    np.testing.assert_allclose(norms(netG.state_dict().values()), norms(g.values()), rtol=1e-3)
    np.testing.assert_allclose(norms(netD.state_dict().values()), norms(d.values()), rtol=1e-3)


def test_f16_graph_replay_equals_eager_bitwise_scale_included():
    """The captured step carries the scaler's decisions on the device: starting from a scale that overflows, replays skip and
    halve until the gradients fit, exactly as the eager steps do."""
    from ste_gan_amd.train import GraphedTrainStep
    from test_gpu_train_step import to_batch
    kw = dict(init_scale=2.0 ** 22, growth_interval=3)
    res = []
    for graphed in (False, True):
        name, netG, netD, optG, optD, step = _small(kw)
        batches = [to_batch(cases.step_batch(name, s % 2)) for s in range(6)]
        if graphed:
            sdG = {k: v.clone() for k, v in netG.state_dict().items()}
            sdD = {k: v.clone() for k, v in netD.state_dict().items()}
            gs = GraphedTrainStep(step, batches[0], warmup=2)
            netG.load_state_dict(sdG); netD.load_state_dict(sdD)
            for opt in (optG, optD):
                opt.reset_state()
            step.scaler.load_state_dict(dict(scale=kw["init_scale"], growth_factor=2.0, backoff_factor=0.5, growth_interval=3,
                                             _growth_tracker=0))
            step.scaler.state[3:].zero_()
            run = gs
        else:
            run = step
        scales = []
        for b in batches:
            out = run(b)
            scales.append(float(out["loss_scale"].item()))
        torch.cuda.synchronize()
        res.append((scales, step.scaler.skipped_steps(), optG.device_step_count(),
                    {k: v.clone() for net in (netG, netD) for k, v in net.state_dict().items()}, out["x_pred"].clone()))
        if graphed:
            gs.close()
    print("\\nscales:", res[0][0], "skipped:", res[0][1], "G steps applied:", res[0][2])
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1] and res[0][2] == res[1][2]
    assert res[0][1] > 0 and res[0][2] > 0, "the sequence should contain skipped AND applied steps"
    assert torch.equal(res[0][4], res[1][4])
    for k in res[0][3]:
        assert torch.equal(res[0][3][k], res[1][3][k]), k


def test_config5_geometry_f16_vs_oracle():
    """BASELINE config 5 geometry: MFCC input at 100 Hz, T_su = 25 -> T_emg = 400, full widths, B = 8, fp16 operands, scale 2^12."""
    from ste_gan_amd import ops
    from ste_gan_amd.optim import GradScaler
    from test_gpu_full_size import _batch, _build, _inputs, _state_dicts
    ops.set_compute_dtype("f16")
    inp = _inputs("c5", 8, 25, True)
    g, d, e = _state_dicts("MFCCS")
    netG, netD, optG, optD, step = _build("MFCCS", g, d, e)
    step.scaler = GradScaler(optG.flat.flat.device, init_scale=2.0 ** 12)
    out = step(_batch(inp, True)); st = step.read_stats(out)
    assert out["x_pred"].shape == (8, 400, 8)
    g, d, e = _state_dicts("MFCCS")
    dt = {k: v for k, v in d.items() if not O.is_sn_buffer(d, k)}
    sc = {"scale": 2.0 ** 12, "growth_tracker": 0}
    with O.bf16_operands(torch.float16):
        o = O.train_step(g, d, e, O.adamw_init(g), O.adamw_init(dt), inp, speech_feature_type="MFCCS", keep_grads=True, scaler=sc)
    assert o["skipped"] == (False, False) and step.scaler.skipped_steps() == 0
    # the build consumed its gradients (optimiser step): compare what the step reports and leaves behind
    got = np.array([st["discriminator"], st["generator"], st["multi_td"], st["speech_unit"], st["phoneme"], st["feature_matching"]])
    want = np.array([o["loss_d"], o["loss_g"], o["loss_td"], o["loss_su"], o["loss_ph"], o["loss_fm"]])
    err = float((np.abs(got - want) / np.abs(want)).max())
    xerr = float((out["x_pred"].cpu() - o["x_pred"]).abs().max())
    print("\\nconfig 5 geometry, f16: losses", f"{err:.2e}", "x_pred", f"{xerr:.2e}")
    assert err <= 1e-3 and xerr <= 2e-3
    from test_gpu_train_step import norms
    # (one AdamW step moves every weight by lr x sign(gradient): elements whose tiny gradient changes sign between two
    # roundings land 2 lr apart -- the bound of the bf16 step tests)
    np.testing.assert_allclose(norms(netG.state_dict().values()), norms(g.values()), rtol=1e-3)
    np.testing.assert_allclose(norms(netD.state_dict().values()), norms(d.values()), rtol=1e-3)


def test_config5_batch128_f16_properties():
    """B = 128, chunk 400, MFCC input, fp16 operands, default scaler: steps stay finite whatever the scaler decides, shapes are
    the reference's, the generator is batch independent inside the fp16 bound, and the scale only moves by the rules."""
    from ste_gan_amd import ops
    from ste_gan_amd.constants import DataType
    from test_gpu_full_size import _batch, _build, _inputs, _losses, _state_dicts
    ops.set_compute_dtype("f16")
    big = _inputs("c5big", 128, 25, True)
    g_sd, d_sd, e_sd = _state_dicts("MFCCS")
    netG, netD, optG, optD, step = _build("MFCCS", g_sd, d_sd, e_sd)
    bb = _batch(big, True)
    with torch.no_grad():
        y128 = netG(bb[DataType.MFCCS], bb[DataType.SESSION_INDEX], bb[DataType.SPEAKING_MODE_INDEX])
        y8 = netG(bb[DataType.MFCCS][:8].contiguous(), bb[DataType.SESSION_INDEX][:8].contiguous(),
                  bb[DataType.SPEAKING_MODE_INDEX][:8].contiguous())
    assert y128.shape == (128, 400, 8) and float((y128[:8] - y8).abs().max()) <= 1e-3
    scale = 65536.0
    for it in range(3):
        out = step(bb)
        st = step.read_stats(out)
        assert all(np.isfinite(v) for v in _losses(st)), st
        assert st["loss_scale"] in (scale, scale / 2), (it, st["loss_scale"], scale)
        scale = st["loss_scale"]
    assert out["x_pred"].shape == (128, 400, 8) and st["num_phones"] == 128 * 25
    assert all(torch.isfinite(p).all() for p in netG.parameters()) and all(torch.isfinite(p).all() for p in netD.parameters())
    print("\\nB = 128: scale after 3 iterations", scale, "skipped optimiser steps", step.scaler.skipped_steps())
