"""-m gpu: frozen EMG encoder + speech-unit / phoneme losses (forward, counters, d/d emg) vs the golden
fixtures of the reference modules, incl. the T_enc > 100 relative-position mask branch."""
import numpy as np
import pytest
import torch

import cases
from oracle import shapes as oshapes
from oracle.detfill import fill_state_dict

pytestmark = pytest.mark.gpu


def close(a, b, rtol, what):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, f"{what}: {a.shape} vs {b.shape}"
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert err <= rtol, f"{what}: max err {err:.3e}"


@pytest.mark.parametrize("name", list(cases.ENC_CASES))
def test_encoder_loss_vs_golden(name, golden):
    from ste_gan_amd.losses.emg_encoder_loss import EMGEncoderLoss
    from ste_gan_amd.models.emg_encoder import EMGEncoderTransformer
    ms, nl, ff, B, T = cases.ENC_CASES[name]
    g = golden(name)
    enc = EMGEncoderTransformer(8, 256, 48, model_size=ms, num_transformer_layers=nl, dim_feedforward=ff)
    enc.load_state_dict(fill_state_dict(oshapes.encoder_shapes(ms, 3, nl, dim_feedforward=ff)))
    enc = enc.cuda().eval()
    emg, su, ph = cases.enc_inputs(name)
    emg_d = emg.cuda().requires_grad_(True)
    out = EMGEncoderLoss(enc)(emg_d, su.cuda(), ph.cuda())
    close(out.speech_unit_pred.detach().cpu().numpy(), g["su_pred"], 5e-5, "su_pred")
    close(out.phoneme_pred.detach().cpu().numpy(), g["ph_pred"], 5e-5, "ph_pred")
    close(float(out.speech_unit_loss), g["su_loss"], 2e-5, "su_loss")
    close(float(out.phoneme_loss), g["ph_loss"], 2e-5, "ph_loss")
    assert [out.num_phones, out.num_silence_phones, out.num_correct_phones,
            out.num_correct_phones_no_silence] == g["counters"].tolist()
    assert np.array_equal(out._argmax.cpu().numpy(), g["argmax"])          # bit-exact phoneme argmax
    (out.speech_unit_loss + out.phoneme_loss).backward()
    close(emg_d.grad.cpu().numpy(), g["demg"], 3e-4, "demg")
    assert all(p.grad is None for p in enc.parameters())                    # frozen: no weight grads computed


def test_training_mode_raises():
    from ste_gan_amd.models.emg_encoder import EMGEncoderTransformer
    enc = EMGEncoderTransformer(8, 256, 48, model_size=64, num_transformer_layers=1, dim_feedforward=64).cuda().train()
    with pytest.raises(NotImplementedError):
        enc(torch.zeros(1, 64, 8, device="cuda"))


@pytest.mark.parametrize("B,C,T", [(64, 768, 12), (3, 100, 5), (4, 64, 40), (2, 48, 300), (2, 3000, 12), (5, 7, 1),
                                   (16, 768, 12), (40, 256, 8), (130, 64, 6), (7, 768, 9)])
def test_add_layernorm_against_torch(B, C, T):
    """stegan_add_layernorm_fwd / _bwd on (B, C, T): the slab kernels (short rows; a block owns a group of T / G columns of its
    batch element, G chosen from B: one column per block at B = 16, three at B = 64, the whole slab at B = 130) and the
    per-column kernels (T > 256, or a slab too large for the per-thread registers) against torch.layer_norm in fp64."""
    from ste_gan_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + C + T)
    x, r, dy = (torch.randn(B, C, T, generator=g) for _ in range(3))
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    xd, rd = x.double().requires_grad_(True), r.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm((xd + rd).transpose(1, 2), (C,), gamma.double(), beta.double(), 1e-5).transpose(1, 2)
    gx, gr = torch.autograd.grad(ref, (xd, rd), dy.double())
    xg, rg = x.cuda().requires_grad_(True), r.cuda().requires_grad_(True)
    y = ops.add_layernorm(xg, rg, gamma.cuda(), beta.cuda(), 1e-5)
    hx, hr = torch.autograd.grad(y, (xg, rg), dy.cuda())
    assert float((y.detach().cpu().double() - ref.detach()).abs().max()) <= 2e-5 * float(ref.abs().max())
    for got, want in ((hx, gx), (hr, gr)):
        assert float((got.cpu().double() - want).abs().max()) <= 5e-5 * float(want.abs().max()) + 1e-7


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_transformer_layer_gradient_handover(dtype, monkeypatch):
    """Input gradient of a TransformerEncoderLayer with the residual sums formed inside the convs' input-gradient kernels
    (ops.GradHandover) against autograd's own accumulation."""
    from ste_gan_amd import ops
    from ste_gan_amd.layers.transformer import TransformerEncoderLayer
    torch.manual_seed(11)
    layer = TransformerEncoderLayer(64, 4, dim_feedforward=128, relative_positional_distance=20).cuda().eval()
    for p in layer.parameters():
        p.requires_grad_(False)
    x0 = torch.randn(6, 64, 12, device="cuda")
    dy = torch.randn(6, 64, 12, device="cuda")
    ops.set_compute_dtype(dtype)
    try:
        res = {}
        for on in (True, False):
            monkeypatch.setattr(ops, "HANDOVER", on)
            x = x0.clone().requires_grad_(True)
            y = layer(x)
            (gx,) = torch.autograd.grad(y, x, dy)
            res[on] = (y.detach().clone(), gx.clone())
    finally:
        ops.set_compute_dtype("f32")
    for a, b in zip(res[True], res[False]):
        assert torch.isfinite(a).all()
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-9
