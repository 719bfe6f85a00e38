"""Pins the CPU oracle (oracle/stegan_oracle.py) to outputs of the reference's own modules.

Fixtures: tests/golden/*.npz, produced by tests/golden/make_golden.py from /root/reference.
Tolerances: fp32 forward values 1e-5 relative to the tensor's max-abs (same ATen ops, different
association order only); gradient norms 1e-4 relative; integer counters / argmax bit-exact.
"""
import numpy as np
import pytest
import torch

import cases
from oracle import shapes as oshapes
from oracle import stegan_oracle as O
from oracle.detfill import det_input, fill_state_dict


def close(a, b, rtol=1e-5, what=""):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, f"{what}: {a.shape} vs {b.shape}"
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max() / scale
    assert err <= rtol, f"{what}: max err {err:.3e} (rel to max-abs {scale:.3e})"


def grads_summary(grads):
    return (np.array([float(g.double().norm()) for g in grads]),
            np.stack([np.pad(g.flatten()[:4].double().numpy(), (0, max(0, 4 - g.numel()))) for g in grads]))


def stats(t):
    t = t.detach().double().flatten()
    return np.array([t.sum(), t.abs().sum(), (t * t).sum()])


@pytest.mark.parametrize("name", list(cases.GEN_CASES))
def test_generator(name, golden):
    ft, ch, B, T = cases.GEN_CASES[name]
    g = golden(name)
    sd = {k: v.requires_grad_(True) for k, v in fill_state_dict(oshapes.generator_shapes(ft, ch)).items()}
    feats, sess, spk = cases.gen_inputs(name)
    out = O.generator_forward(sd, feats, sess, spk, ft)
    close(out.detach().numpy(), g["out"], 1e-5, "out")
    assert out.shape[1] == T * (16 if ft == "SPEECH_UNITS" else 8)
    probe = det_input(name + ".probe", out.shape)
    grads = torch.autograd.grad((out * probe).sum(), list(sd.values()))
    gn, gh = grads_summary(grads)
    close(gn, g["grad_norm"], 1e-4, "grad_norm")
    close(gh, g["grad_head"], 1e-4, "grad_head")


@pytest.mark.parametrize("name", list(cases.DISC_CASES))
def test_discriminator(name, golden):
    small, B, T = cases.DISC_CASES[name]
    g = golden(name)
    sd = fill_state_dict(oshapes.discriminator_shapes(small))
    keys = [k for k in sd if not O.is_sn_buffer(sd, k)]
    for k in keys:
        sd[k].requires_grad_(True)
    x = cases.disc_input(name).requires_grad_(True)
    sn = {}
    out1 = O.discriminator_forward(sd, x, small, True, sn)
    close(np.concatenate([sc[-1].detach().flatten().numpy() for sc in out1]), g["logits1"], 1e-5, "logits1")
    fms = [fm for sc in out1 for fm in sc]
    assert [list(fm.shape) + [0] * (4 - fm.dim()) for fm in fms] == g["fmap_shapes"].tolist()
    close(np.stack([stats(fm) for fm in fms]), g["fmap_stats"], 1e-5, "fmap_stats")
    loss = O.lsgan_loss(out1, 1.0)
    for sc in out1:
        for fm in sc[:-1]:
            loss = loss + fm.abs().mean()
    close(float(loss), g["loss"], 1e-5, "loss")
    grads = torch.autograd.grad(loss, [x] + [sd[k] for k in keys])
    close(grads[0].numpy(), g["dx"], 1e-4, "dx")
    gn, gh = grads_summary(grads[1:])
    close(gn, g["grad_norm"], 1e-4, "grad_norm")
    close(gh, g["grad_head"], 1e-4, "grad_head")
    sd2 = {k: v.detach() for k, v in sd.items()}
    sd2.update(sn)
    sn2 = {}
    out2 = O.discriminator_forward(sd2, x.detach(), small, True, sn2)
    close(np.concatenate([sc[-1].flatten().numpy() for sc in out2]), g["logits2"], 1e-5, "logits2")
    for k, v in sn2.items():
        close(v.numpy(), g["sn." + k], 1e-5, k)


@pytest.mark.parametrize("name", list(cases.TD_CASES))
def test_multi_td(name, golden):
    g = golden(name)
    xr, xg = cases.td_inputs(name)
    xg.requires_grad_(True)
    loss = O.multi_td_loss(xr, xg)
    close(float(loss), g["loss"], 1e-5, "loss")
    close(torch.autograd.grad(loss, xg)[0].numpy(), g["dx"], 1e-4, "dx")
    close(O.td_features(xr, 20, 8).numpy(), g["feats0"], 1e-5, "feats0")


@pytest.mark.parametrize("name", list(cases.ENC_CASES))
def test_encoder_losses(name, golden):
    ms, nl, ff, B, T = cases.ENC_CASES[name]
    g = golden(name)
    sd = fill_state_dict(oshapes.encoder_shapes(ms, 3, nl, dim_feedforward=ff))
    emg, su, ph = cases.enc_inputs(name)
    emg.requires_grad_(True)
    su_pred, ph_pred = O.encoder_forward(sd, emg)
    close(su_pred.detach().numpy(), g["su_pred"], 2e-5, "su_pred")
    close(ph_pred.detach().numpy(), g["ph_pred"], 2e-5, "ph_pred")
    su_l, ph_l = O.speech_unit_loss(su, su_pred), O.phoneme_loss(ph_pred, ph)
    close(float(su_l), g["su_loss"], 1e-5, "su_loss")
    close(float(ph_l), g["ph_loss"], 1e-5, "ph_loss")
    c = O.phoneme_counters(ph_pred.detach(), ph)
    assert [c["num_phones"], c["num_silence_phones"], c["num_correct_phones"],
            c["num_correct_phones_no_silence"]] == g["counters"].tolist()
    assert np.array_equal(ph_pred.argmax(-1).numpy(), g["argmax"])          # bit-exact argmax
    close(torch.autograd.grad(su_l + ph_l, emg)[0].numpy(), g["demg"], 1e-4, "demg")


@pytest.mark.parametrize("name", list(cases.STEP_CASES))
def test_train_step(name, golden):
    ch, (ms, nl, ff), B, T, steps = cases.STEP_CASES[name]
    g = golden(name)
    g_sd = fill_state_dict(oshapes.generator_shapes("SPEECH_UNITS", ch))
    d_sd = fill_state_dict(oshapes.discriminator_shapes(True))
    e_sd = fill_state_dict(oshapes.encoder_shapes(ms, 3, nl, dim_feedforward=ff))
    d_train = {k: v for k, v in d_sd.items() if not O.is_sn_buffer(d_sd, k)}
    opt_g, opt_d = O.adamw_init(g_sd), O.adamw_init(d_train)
    for s in range(steps):
        out = O.train_step(g_sd, d_sd, e_sd, opt_g, opt_d, cases.step_batch(name, s), keep_grads=True)
        t = f"s{s}."
        got = [out[k] for k in ("loss_d", "loss_g", "loss_adv_g", "loss_td", "loss_su", "loss_ph", "loss_fm")]
        close(got, g[t + "losses"], 2e-5, t + "losses")
        assert [out["num_phones"], out["num_silence_phones"], out["num_correct_phones"],
                out["num_correct_phones_no_silence"]] == g[t + "counters"].tolist()
        close(out["x_pred"].numpy(), g[t + "x_pred"], 2e-5, t + "x_pred")
        gn, gh = grads_summary(list(out["g_grads"].values()))
        close(gn, g[t + "g_grad_norm"], 2e-4, t + "g_grad_norm")
        dn, dh = grads_summary(list(out["d_grads"].values()))
        close(dn, g[t + "d_grad_norm"], 2e-4, t + "d_grad_norm")
        pn, ph_ = grads_summary(list(g_sd.values()))
        close(pn, g[t + "g_param_norm"], 1e-5, t + "g_param_norm")
        close(ph_, g[t + "g_param_head"], 1e-4, t + "g_param_head")
        pn, ph_ = grads_summary(list(d_sd.values()))
        close(pn, g[t + "d_param_norm"], 1e-5, t + "d_param_norm")
