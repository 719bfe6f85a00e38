"""CPU: the oracle's bf16-operand mode (``bf16_operands()``, test infrastructure for tests/test_gpu_bf16_emulation.py).

With the flag off nothing changes (the golden fixtures pin that: test_oracle_vs_golden.py).  With it on: a contraction whose
shape the build keeps on exact fp32 kernels is computed exactly, gradients included; a contraction the build runs on bf16
operands equals the plain one evaluated on hand-rounded operands; and the flag restores itself."""
import torch
import torch.nn.functional as F

from oracle import stegan_oracle as O


def _wn(cout, cin_g, k, seed):
    g = torch.Generator().manual_seed(seed)
    return {"c.weight_v": torch.randn(cout, cin_g, k, generator=g), "c.weight_g": torch.rand(cout, 1, 1, generator=g) + 0.5,
            "c.bias": torch.randn(cout, generator=g)}


def _grads(sd, x, dy, **kw):
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xx = x.clone().requires_grad_(True)
    y = O._wn_conv1d(leaves, "c", xx, **kw)
    return (y,) + torch.autograd.grad(y, [xx] + list(leaves.values()), dy)


def test_flag_is_scoped():
    assert not O.BF16_OPERANDS
    with O.bf16_operands():
        assert O.BF16_OPERANDS
        with O.bf16_operands():
            pass
        assert O.BF16_OPERANDS
    assert not O.BF16_OPERANDS


def test_shapes_the_build_keeps_exact_stay_exact():
    """Cout = 1 (the discriminators' output convs) and groups narrower than 8 channels: same numbers as the fp32 path, through
    the emulation's own backward (which re-derives dx / dW / db by hand)."""
    for cin, cout, groups in ((64, 1, 1), (4, 32, 1), (32, 32, 8)):
        sd = _wn(cout, cin // groups, 3, 1)
        x = torch.randn(2, cin, 20, generator=torch.Generator().manual_seed(2))
        dy = torch.randn(2, cout, 20, generator=torch.Generator().manual_seed(3))
        assert O._emu_flags(cin, cout, groups) == (False, False, cin // groups >= 16 and cout // groups >= 16)
        plain = _grads(sd, x, dy, padding=1, groups=groups)
        with O.bf16_operands():
            emu = _grads(sd, x, dy, padding=1, groups=groups)
        for a, b in zip(plain, emu):
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)


def test_rounding_points_of_a_bf16_contraction():
    cin, cout = 32, 48
    sd = _wn(cout, cin, 3, 4)
    x = torch.randn(2, cin, 24, generator=torch.Generator().manual_seed(5))
    dy = torch.randn(2, cout, 24, generator=torch.Generator().manual_seed(6))
    with O.bf16_operands():
        y, dx, dv, dg, db = _grads(sd, x, dy, padding=1)
    r = O._r
    v, g, b = sd["c.weight_v"], sd["c.weight_g"], sd["c.bias"]
    scale = (g.flatten() / v.flatten(1).norm(dim=1))
    y_ref = F.conv1d(r(x), r(v), None, padding=1) * scale.view(1, -1, 1) + b.view(1, -1, 1)
    torch.testing.assert_close(y, y_ref, rtol=1e-6, atol=1e-6)
    dx_ref = F.conv_transpose1d(r(dy), r(v * scale.view(-1, 1, 1)), None, padding=1)
    torch.testing.assert_close(dx, dx_ref, rtol=1e-5, atol=1e-6)
    # weight gradient: dW_eff from rounded operands, then the fp32 weight-norm chain rule
    w0 = torch.zeros_like(v, requires_grad=True)
    (dw,) = torch.autograd.grad(F.conv1d(r(x), w0, None, padding=1), w0, r(dy))
    vl, gl = v.clone().requires_grad_(True), g.clone().requires_grad_(True)
    dv_ref, dg_ref = torch.autograd.grad(O.weight_norm_weight(vl, gl), [vl, gl], dw)
    torch.testing.assert_close(dv, dv_ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dg, dg_ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(db, dy.sum((0, 2)), rtol=1e-5, atol=1e-5)
    # and it is not the fp32 result
    assert float((y - O._wn_conv1d(sd, "c", x, padding=1)).abs().max()) > 1e-4


def _fixture_step(scaler=None, dtype=None):
    import copy
    import cases
    from oracle import shapes as oshapes
    from oracle.detfill import fill_state_dict
    name = next(iter(cases.STEP_CASES))
    ch, (ms, nl, ff), B, T, steps = cases.STEP_CASES[name]
    g = fill_state_dict(oshapes.generator_shapes("SPEECH_UNITS", ch)); d = fill_state_dict(oshapes.discriminator_shapes(True))
    e = fill_state_dict(oshapes.encoder_shapes(ms, 3, nl, dim_feedforward=ff))
    dt = {k: v for k, v in d.items() if not O.is_sn_buffer(d, k)}
    g0, d0 = copy.deepcopy(g), copy.deepcopy(d)
    og, od = O.adamw_init(g), O.adamw_init(dt)
    b = cases.step_batch(name, 0)
    if dtype is None:
        o = O.train_step(g, d, e, og, od, b, scaler=scaler)
    else:
        with O.bf16_operands(dtype):
            o = O.train_step(g, d, e, og, od, b, scaler=scaler)
    return o, (g0, d0), (g, d), (og, od)


def test_grad_scaler_restatement():
    """train_step(scaler=...) = torch.cuda.amp.GradScaler as ste_gan/train.py:151,198-199,266-268 drives it."""
    # fp32 arithmetic: a power-of-two scale changes nothing but the growth tracker
    plain, _, (g1, d1), _ = _fixture_step()
    sc = {"scale": 1024.0, "growth_tracker": 0, "growth_interval": 1}
    scaled, _, (g2, d2), _ = _fixture_step(sc)
    assert scaled["skipped"] == (False, False) and sc["scale"] == 2048.0 and sc["growth_tracker"] == 0
    assert scaled["loss_g"] == plain["loss_g"] and scaled["loss_d"] == plain["loss_d"]        # the values are not scaled
    for k in g1:
        torch.testing.assert_close(g1[k], g2[k], rtol=1e-6, atol=1e-7)
    # fp16 operands with a scale that overflows them: both optimiser steps skipped, nothing moves, the scale is halved
    sc = {"scale": 2.0 ** 40, "growth_tracker": 7}
    o, (g0, d0), (g3, d3), (og, od) = _fixture_step(sc, torch.float16)
    assert o["skipped"] == (True, True) and sc["scale"] == 2.0 ** 39 and sc["growth_tracker"] == 0
    assert og.step == 0 and od.step == 0
    assert all(torch.equal(g0[k], g3[k]) for k in g0)
    assert all(torch.equal(d0[k], d3[k]) for k in d0 if not O.is_sn_buffer(d0, k))
