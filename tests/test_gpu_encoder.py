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
