"""-m gpu: the train step is bit-reproducible (round 4).

Rounds 1-3 let the batch slices of the weight / bias gradients meet in fp32 atomics, whose order changes from launch to
launch.  The discriminator's update precedes the generator's backward pass (train.py:189-199 before 206-268), so last-bit
noise in D's weights reached every generator gradient -- amplified in bf16 mode, where a last-bit difference of an
activation can flip the rounding of a conv operand (2^-9 relative) -- and two launches of the same step differed by more
than the bound of the side-stream test (GPUTEST_r03).  Since round 4 every sum of the gradient path has a fixed order
(slabs + ordered reduce, one block per bias channel, single-writer read-modify-writes); only the reported loss WORDS still
meet in an atomic, and no gradient depends on them.

Done = two launches of the same step from the same state give bit-identical generator and discriminator parameters,
gradients and generated EMG: f32 and bf16, eager and replayed as a hipGraph, weight gradients inline or on side streams,
small fixture and the full BASELINE config-3 geometry."""
import pytest
import torch

import cases
from test_gpu_train_step import build, to_batch

pytestmark = pytest.mark.gpu


def _snapshot(netG, netD, out, grads=True):
    snap = {"x_pred": out["x_pred"].clone()}
    for tag, net in (("G", netG), ("D", netD)):
        for k, v in net.state_dict().items():                       # parameters AND spectral-norm u / v buffers
            snap[f"{tag}.{k}"] = v.detach().clone()
        if grads:
            for i, p in enumerate(net.parameters()):
                snap[f"{tag}.grad{i}"] = p.grad.clone()
    return snap


def _assert_identical(a, b, what):
    assert a.keys() == b.keys()
    bad = [k for k in a if not torch.equal(a[k], b[k])]
    if bad:
        k = bad[0]
        d = (a[k].double() - b[k].double()).abs()
        raise AssertionError(f"{what}: {len(bad)} of {len(a)} tensors differ between two launches of the same step; first: {k} "
                             f"(max |delta| {float(d.max()):.3e}, max |value| {float(b[k].abs().max()):.3e}, "
                             f"{int((d > 0).sum())} of {d.numel()} elements)")


def _run_eager(name, steps, wgrad_streams=None):
    from ste_gan_amd import ops
    old = ops.WGRAD_STREAMS
    if wgrad_streams is not None:
        ops.WGRAD_STREAMS = wgrad_streams
    try:
        cfg, netG, netD, optG, optD, step = build(name)
        for s in range(steps):
            out = step(to_batch(cases.step_batch(name, s)))
        torch.cuda.synchronize()
        return _snapshot(netG, netD, out)
    finally:
        ops.WGRAD_STREAMS = old


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_two_launches_of_the_eager_step_are_bit_identical(dtype):
    from ste_gan_amd import ops
    name = next(iter(cases.STEP_CASES))
    ops.set_compute_dtype(dtype)
    try:
        a = _run_eager(name, 2)
        b = _run_eager(name, 2)
        _assert_identical(a, b, f"eager {dtype}")
        # weight gradients on side streams (the default inside the generator's backward pass) or inline: the same kernels
        # with the same plans on other streams -- bit-identical too (the red test of GPUTEST_r03, tightened to equality)
        c = _run_eager(name, 2, wgrad_streams=0)
        _assert_identical(a, c, f"eager {dtype}, weight gradients inline vs on side streams")
    finally:
        ops.set_compute_dtype("f32")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_graph_replays_are_bit_identical_and_equal_the_eager_step(dtype):
    """The captured step replays the same kernels with the same arguments: restoring the initial state and replaying gives
    the same bits as the first replay -- and as the eager step (capture changes neither plans nor summation orders)."""
    from ste_gan_amd import ops
    from ste_gan_amd.train import GraphedTrainStep
    name = next(iter(cases.STEP_CASES))
    ops.set_compute_dtype(dtype)
    try:
        eager = _run_eager(name, 2)
        cfg, netG, netD, optG, optD, step = build(name)
        sdG = {k: v.clone() for k, v in netG.state_dict().items()}
        sdD = {k: v.clone() for k, v in netD.state_dict().items()}
        gs = GraphedTrainStep(step, to_batch(cases.step_batch(name, 0)), warmup=2)
        snaps = []
        for _ in range(2):
            netG.load_state_dict(sdG); netD.load_state_dict(sdD)
            for opt in (optG, optD):
                opt.reset_state()
            for s in range(2):
                out = gs(to_batch(cases.step_batch(name, s)))
            torch.cuda.synchronize()
            snaps.append(_snapshot(netG, netD, out, grads=False))        # (the replayed optimiser consumes the gradients)
        _assert_identical(snaps[0], snaps[1], f"graph replay {dtype}")
        _assert_identical({k: v for k, v in eager.items() if ".grad" not in k}, snaps[0], f"graph replay vs eager {dtype}")
        gs.close()
    finally:
        ops.set_compute_dtype("f32")


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_config3_full_size_step_is_bit_reproducible(dtype):
    """BASELINE config 3 geometry (B = 64, 768 generator channels, 768 x 6 encoder, T_emg = 192): all the sliced weight
    gradients, split-K sums and multi-stream placements of the benchmarked step, two launches, equal bits."""
    from ste_gan_amd import ops
    from test_gpu_full_size import _batch, _build, _inputs, _state_dicts
    ops.set_compute_dtype(dtype)
    try:
        snaps = []
        g_sd, d_sd, e_sd = _state_dicts("SPEECH_UNITS")
        for _ in range(2):
            netG, netD, optG, optD, step = _build("SPEECH_UNITS", g_sd, d_sd, e_sd)
            out = step(_batch(_inputs("c3", 64, 12, False), False))
            torch.cuda.synchronize()
            snaps.append(_snapshot(netG, netD, out))
            del netG, netD, optG, optD, step, out
            torch.cuda.empty_cache()
        _assert_identical(snaps[0], snaps[1], f"config 3 {dtype}")
    finally:
        ops.set_compute_dtype("f32")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_fused_activation_derivative_equals_the_separate_launch(dtype):
    """ops.ActFuse (round 4): inside the train step the conv BEHIND a discriminator feature map applies the leaky-ReLU
    derivative of the layer before it -- and adds the feature-matching gradient in front of it -- in its input-gradient
    epilogue, instead of one act_bwd launch per layer.  Same fp32 operations in the same order: the step's results do not
    change by a bit (bf16 mode: the fused epilogues; fp32 mode never links its layers, the parity path is untouched)."""
    from ste_gan_amd import ops
    name = next(iter(cases.STEP_CASES))
    ops.set_compute_dtype(dtype)
    old = ops.ACT_FUSE
    try:
        snaps = []
        for fuse in (True, False):
            ops.ACT_FUSE = fuse
            snaps.append(_run_eager(name, 2))
        _assert_identical(snaps[0], snaps[1], f"ActFuse on vs off, {dtype}")
    finally:
        ops.ACT_FUSE = old
        ops.set_compute_dtype("f32")
