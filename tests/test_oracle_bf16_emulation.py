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
