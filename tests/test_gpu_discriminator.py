"""-m gpu: discriminators (small + full) forward / backward through libstegan_hip.so vs the golden
fixtures of the reference modules and the CPU oracle.  fp32, tolerances relative to tensor max-abs."""
import numpy as np
import pytest
import torch

import cases
from oracle import shapes as oshapes
from oracle import stegan_oracle as O
from oracle.detfill import fill_state_dict

pytestmark = pytest.mark.gpu


def close(a, b, rtol, what):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, f"{what}: {a.shape} vs {b.shape}"
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert err <= rtol, f"{what}: max err {err:.3e}"


def stats(t):
    t = t.detach().double().flatten()
    return np.array([float(t.sum()), float(t.abs().sum()), float((t * t).sum())])


@pytest.mark.parametrize("name", list(cases.DISC_CASES))
def test_discriminator_vs_golden(name, golden):
    from ste_gan_amd.models.discriminator import Discriminator, DiscriminatorSmall
    small, B, T = cases.DISC_CASES[name]
    g = golden(name)
    d = (DiscriminatorSmall if small else Discriminator)(8)
    d.load_state_dict(fill_state_dict(oshapes.discriminator_shapes(small)))
    d = d.cuda()
    x = cases.disc_input(name).cuda().requires_grad_(True)
    out1 = d(x)
    close(torch.cat([sc[-1].flatten() for sc in out1]).detach().cpu().numpy(), g["logits1"], 5e-5, "logits1")
    fms = [fm for sc in out1 for fm in sc]
    assert [list(fm.shape) + [0] * (4 - fm.dim()) for fm in fms] == g["fmap_shapes"].tolist()
    close(np.stack([stats(fm) for fm in fms]), g["fmap_stats"], 5e-5, "fmap_stats")
    head = np.stack([np.pad(fm.detach().flatten()[:8].double().cpu().numpy(), (0, max(0, 8 - fm.numel()))) for fm in fms])
    close(head, g["fmap_head"], 5e-5, "fmap_head")
    loss = 0
    for sc in out1:
        loss = loss + torch.nn.functional.mse_loss(sc[-1], torch.ones_like(sc[-1]))
        for fm in sc[:-1]:
            loss = loss + fm.abs().mean()
    close(float(loss), g["loss"], 5e-5, "loss")
    loss.backward()
    # leaky-ReLU has a kink at 0: of the ~10^6 pre-activations a few lie within fp32 rounding of it, and a different
    # (equally valid) summation order can flip their slope 0.1 <-> 1, which moves single dx entries by up to ~1e-3 of
    # max|dx|.  The norm of the difference stays tiny, which is what is asserted tightly.
    close(x.grad.cpu().numpy(), g["dx"], 2e-3, "dx (max-abs)")
    dxe = x.grad.cpu().numpy().astype(np.float64) - g["dx"]
    assert np.linalg.norm(dxe) <= 1e-3 * np.linalg.norm(g["dx"]), "dx (l2)"
    params = list(d.parameters())
    gn = np.array([float(p.grad.double().norm()) for p in params])
    close(gn, g["grad_norm"], 2e-4, "grad_norm")
    np.testing.assert_allclose(gn, g["grad_norm"], rtol=1e-3, atol=1e-5 * g["grad_norm"].max())
    gh = np.stack([np.pad(p.grad.flatten()[:4].double().cpu().numpy(), (0, max(0, 4 - p.numel()))) for p in params])
    close(gh, g["grad_head"], 5e-4, "grad_head")
    out2 = d(x.detach())
    close(torch.cat([sc[-1].flatten() for sc in out2]).detach().cpu().numpy(), g["logits2"], 5e-5, "logits2")
    sd = d.state_dict()
    for k in g.files:
        if k.startswith("sn."):
            close(sd[k[3:]].cpu().numpy(), g[k], 5e-5, k)


def test_eval_mode_does_not_power_iterate():
    from ste_gan_amd.models.discriminator import DiscriminatorSmall
    d = DiscriminatorSmall(8)
    d.load_state_dict(fill_state_dict(oshapes.discriminator_shapes(True)))
    d = d.cuda().eval()
    u0 = d.multi_scale_disc[0].layers[0].weight_u.clone()
    x = cases.disc_input("disc_small_t64").cuda()
    a = d(x); b = d(x)
    assert torch.equal(u0, d.multi_scale_disc[0].layers[0].weight_u)
    assert torch.equal(a[5][-1], b[5][-1])
    sd = {k: v.cpu() for k, v in d.state_dict().items()}
    ref = O.discriminator_forward(sd, x.cpu(), True, training=False)
    close(a[5][-1].detach().cpu().numpy(), ref[5][-1].detach().numpy(), 5e-5, "eval logits")


def test_batched_spectral_norm_equals_per_layer():
    """All spectral-norm layers in one call (ops.spectral_norm_multi) == each layer running its own power iteration:
    same bits in the feature maps / logits and the u / v buffers over two training forwards; the gradients (whose
    batch-slice partial sums meet in fp32 atomics, so their last bits vary from run to run) to 1e-6 of their scale."""
    from ste_gan_amd import ops
    from ste_gan_amd.models.discriminator import Discriminator
    res = []
    for batched in (True, False):
        ops.SN_BATCHED = batched
        try:
            d = Discriminator(8)
            d.load_state_dict(fill_state_dict(oshapes.discriminator_shapes(False)))
            d = d.cuda()
            sn = d.multi_scale_disc[0]
            xs = ops.btc_to_bct(cases.disc_input("disc_full_t192").cuda())
            f1 = sn(xs)
            f2 = sn(xs * 0.5)
            (f1[-1].square().mean() + f2[-1].mean() + f1[2].abs().mean()).backward()
            res.append(([f.detach().clone() for f in f1 + f2], [p.grad.clone() for p in sn.parameters()],
                        [b.clone() for b in sn.buffers()]))
        finally:
            ops.SN_BATCHED = True
    for what, a, b in zip(("outputs", "gradients", "buffers"), res[0], res[1]):
        assert len(a) == len(b)
        for i, (ta, tb) in enumerate(zip(a, b)):
            if what == "gradients":
                assert float((ta - tb).abs().max()) <= 1e-6 * float(tb.abs().max()) + 1e-30, f"{what}[{i}]"
            else:
                assert torch.equal(ta, tb), f"{what}[{i}] {tuple(ta.shape)}: max diff {float((ta - tb).abs().max()):.3e}"


def test_reflect_pad_too_large_raises():
    from ste_gan_amd.models.discriminator import DiscriminatorSmall
    d = DiscriminatorSmall(8).cuda()
    with pytest.raises(RuntimeError):          # T=4: period 11 needs pad 7 >= T (ATen raises too)
        d(torch.zeros(1, 4, 8, device="cuda"))


@pytest.mark.parametrize("B,T,Cc,periods", [(3, 192, 8, [2, 3, 5, 7, 11]), (2, 50, 4, [2, 5]), (1, 23, 1, [3, 4, 6, 7, 8, 9, 10, 11])])
def test_mpd_fold_multi_equals_the_single_folds(B, T, Cc, periods):
    """stegan_mpd_fold_fwd_multi / _bwd_multi: the folds of all period discriminators in one launch, the backward also
    summing their gradients -- against one stegan_mpd_fold_fwd / _bwd per period and autograd's accumulation."""
    from ste_gan_amd import ops
    g = torch.Generator().manual_seed(T)
    x0 = torch.randn(B, T, Cc, generator=g).cuda()
    xa, xb = x0.clone().requires_grad_(True), x0.clone().requires_grad_(True)
    multi = ops.mpd_fold_multi(xa, periods)
    single = [ops.mpd_fold(xb, p) for p in periods]
    douts = [torch.randn(s.shape, generator=g).cuda() for s in single]
    for m, s in zip(multi, single):
        assert torch.equal(m, s)
    torch.autograd.backward(multi, douts)
    torch.autograd.backward(single, douts)
    assert float((xa.grad - xb.grad).abs().max()) <= 1e-6 * float(xb.grad.abs().max())
