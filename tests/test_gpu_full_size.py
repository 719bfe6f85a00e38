"""-m gpu: the train step at the sizes bench.py times ("full-size parity"):

(i)   BASELINE config 3 -- generator 768 channels, small discriminators, the full 768-d x 6-layer encoder, B=64, T_su=12
      (-> T_emg=192), detfill weights -- one step against ``oracle.train_step`` (ste_gan/train.py:165-268 restated on the
      CPU): fp32 within north_star's 1e-4, bf16 inside the bf16 bounds of test_gpu_train_step_bf16.py, and the same step
      replayed through ``GraphedTrainStep``.  At this size the kernels take the 128 x 128 tiles, the split-K plans and the
      multi-problem launches that the reduced-width fixtures never reach.
(ii)  BASELINE config 5 geometry (MFCC input at 100 Hz, T_su=25 -> 50 MFCC frames -> T_emg=400), full widths, B=8, bf16,
      against the oracle (models/generator.py:116-131: upsample_last=1).
(iii) the same at B=128 through size-independent properties: finite losses, output shape, and batch independence -- the
      first 8 samples of the B=128 generator output equal the B=8 run's (the networks have no cross-sample coupling).

The oracle step costs a few seconds at these sizes on the GPU box's host; its results are computed once per module.
"""
import numpy as np
import pytest
import torch

from oracle import shapes as oshapes
from oracle.detfill import det_input, det_ints, fill_state_dict
from test_gpu_train_step import close, make_cfg, norms

pytestmark = pytest.mark.gpu

CH = 768


def _state_dicts(feature):
    return (fill_state_dict(oshapes.generator_shapes(feature, CH)), fill_state_dict(oshapes.discriminator_shapes(True)),
            fill_state_dict(oshapes.encoder_shapes()))


def _build(feature, g_sd, d_sd, e_sd):
    from ste_gan_amd.losses.emg_encoder_loss import EMGEncoderLoss
    from ste_gan_amd.losses.time_domain_loss import MultiTimeDomainFeatureLoss
    from ste_gan_amd.models.discriminator import init_emg_discriminators
    from ste_gan_amd.models.emg_encoder import EMGEncoderTransformer
    from ste_gan_amd.models.generator import init_emg_generator
    from ste_gan_amd.train import GANTrainStep, make_optimizers
    cfg = make_cfg(CH)
    cfg.model.speech_feature_type = feature
    netG = init_emg_generator(cfg); netG.load_state_dict(g_sd)
    netD = init_emg_discriminators(cfg); netD.load_state_dict(d_sd)
    enc = EMGEncoderTransformer(8, 256, 48); enc.load_state_dict(e_sd)
    netG, netD, enc = netG.cuda(), netD.cuda(), enc.cuda().eval()
    optG, optD = make_optimizers(netG, netD)
    step = GANTrainStep(cfg, netG, netD, EMGEncoderLoss(enc), MultiTimeDomainFeatureLoss(8), optG, optD)
    return netG, netD, optG, optD, step


def _inputs(tag, B, T, mfcc):
    d = {"su": det_input(tag + ".su", (B, T, 256)), "x_real": det_input(tag + ".x", (B, 16 * T, 8), "tanh"),
         "sess": det_ints(tag + ".s", (B,), 17), "phonemes": det_ints(tag + ".ph", (B, T), 48)}
    d["feats"] = det_input(tag + ".mf", (B, 2 * T, 25)) if mfcc else d["su"]
    return d


def _batch(inp, mfcc):
    from ste_gan_amd.constants import DataType
    b = {DataType.REAL_EMG: inp["x_real"].cuda(), DataType.SPEECH_UNITS: inp["su"].cuda(),
         DataType.SESSION_INDEX: inp["sess"].cuda(), DataType.SPEAKING_MODE_INDEX: torch.zeros_like(inp["sess"]).cuda(),
         DataType.PHONEMES: inp["phonemes"].cuda()}
    if mfcc:
        b[DataType.MFCCS] = inp["feats"].cuda()
    return b


def _oracle(feature, inp):
    from oracle import stegan_oracle as O
    g_sd, d_sd, e_sd = _state_dicts(feature)
    d_train = {k: v for k, v in d_sd.items() if not O.is_sn_buffer(d_sd, k)}
    o = O.train_step(g_sd, d_sd, e_sd, O.adamw_init(g_sd), O.adamw_init(d_train), inp, speech_feature_type=feature,
                     keep_grads=True)
    return dict(losses=[o["loss_d"], o["loss_g"], o["loss_td"], o["loss_su"], o["loss_ph"], o["loss_fm"]],
                counters=[o["num_phones"], o["num_silence_phones"], o["num_correct_phones"],
                          o["num_correct_phones_no_silence"]],
                x_pred=o["x_pred"].numpy(), g_norm=norms(g_sd.values()), d_norm=norms(d_sd.values()))


@pytest.fixture(scope="module")
def ref_c3():
    return _oracle("SPEECH_UNITS", _inputs("c3", 64, 12, False))


@pytest.fixture(scope="module")
def ref_c5():
    return _oracle("MFCCS", _inputs("c5", 8, 25, True))


def _losses(st):
    return [st["discriminator"], st["generator"], st["multi_td"], st["speech_unit"], st["phoneme"], st["feature_matching"]]


def _counters(st):
    return [st["num_phones"], st["num_silence_phones"], st["num_correct_phones"], st["num_correct_phones_no_silence"]]


def _check_f32(st, x_pred, netG, netD, ref):
    close(_losses(st), ref["losses"], 1e-4, "losses")
    assert _counters(st) == ref["counters"]                       # phoneme argmax: bit-exact
    err = np.abs(x_pred - ref["x_pred"])
    assert (err <= 1e-4 * np.abs(ref["x_pred"]) + 1e-5).all(), f"x_pred: max err {err.max():.3e}"
    close(norms(netG.state_dict().values()), ref["g_norm"], 2e-5, "g params after the step")
    close(norms(netD.state_dict().values()), ref["d_norm"], 2e-5, "d params after the step")


def _check_bf16(st, x_pred, netG, netD, ref):
    np.testing.assert_allclose(np.array(_losses(st)), np.array(ref["losses"]), rtol=3e-2, err_msg="losses")
    assert np.abs(x_pred - ref["x_pred"]).max() <= 3e-2, "x_pred"
    got, want = np.array(_counters(st)), np.array(ref["counters"])
    assert got[0] == want[0] and abs(int(got[2]) - int(want[2])) <= max(2, want[0] // 50)     # a near-tie may flip in bf16
    np.testing.assert_allclose(norms(netG.state_dict().values()), ref["g_norm"], rtol=1e-3)
    np.testing.assert_allclose(norms(netD.state_dict().values()), ref["d_norm"], rtol=1e-3)


@pytest.mark.parametrize("dtype,graphed", [("f32", False), ("bf16", False), ("bf16", True), ("f32", True)])
def test_config3_full_size_step_vs_oracle(dtype, graphed, ref_c3):
    from ste_gan_amd import ops
    from ste_gan_amd.train import GraphedTrainStep
    inp = _inputs("c3", 64, 12, False)
    ops.set_compute_dtype(dtype)
    try:
        g_sd, d_sd, e_sd = _state_dicts("SPEECH_UNITS")
        netG, netD, optG, optD, step = _build("SPEECH_UNITS", g_sd, d_sd, e_sd)
        batch = _batch(inp, False)
        if graphed:
            sdG = {k: v.clone() for k, v in netG.state_dict().items()}
            sdD = {k: v.clone() for k, v in netD.state_dict().items()}
            gs = GraphedTrainStep(step, batch, warmup=2)
            netG.load_state_dict(sdG); netD.load_state_dict(sdD)       # the warm-up steps trained the nets: start over
            for opt in (optG, optD):
                opt.reset_state()
            out = gs(batch)
            st = gs.read_stats(out)
        else:
            out = step(batch)
            st = step.read_stats(out)
        x_pred = out["x_pred"].cpu().numpy()
        assert x_pred.shape == (64, 192, 8)
        (_check_f32 if dtype == "f32" else _check_bf16)(st, x_pred, netG, netD, ref_c3)
    finally:
        ops.set_compute_dtype("f32")


def test_config5_mfcc_geometry_bf16_vs_oracle(ref_c5):
    """chunk 400, MFCC input, full widths, B=8, bf16 operands."""
    from ste_gan_amd import ops
    inp = _inputs("c5", 8, 25, True)
    ops.set_compute_dtype("bf16")
    try:
        g_sd, d_sd, e_sd = _state_dicts("MFCCS")
        netG, netD, optG, optD, step = _build("MFCCS", g_sd, d_sd, e_sd)
        out = step(_batch(inp, True))
        st = step.read_stats(out)
        x_pred = out["x_pred"].cpu().numpy()
        assert x_pred.shape == (8, 400, 8)
        _check_bf16(st, x_pred, netG, netD, ref_c5)
    finally:
        ops.set_compute_dtype("f32")


def test_config5_batch128_properties():
    """B=128, chunk 400, MFCC input, bf16: a whole step stays finite, shapes are the reference's, and the generator is
    batch independent -- rows 0..7 of the B=128 forward equal the B=8 forward of the same rows.  Tile / split-K plans differ
    between the two batch sizes, so a layer's fp32 sums differ in their last bit, which can move the bf16 rounding of the
    next layer's operand by one unit (2^-9 relative): equality holds inside the bf16 bound of the step tests, not to 1e-6.
    (In fp32 mode the same comparison holds to 2e-5.)"""
    from ste_gan_amd import ops
    from ste_gan_amd.constants import DataType
    big = _inputs("c5big", 128, 25, True)
    ops.set_compute_dtype("bf16")
    try:
        g_sd, d_sd, e_sd = _state_dicts("MFCCS")
        netG, netD, optG, optD, step = _build("MFCCS", g_sd, d_sd, e_sd)
        bb = _batch(big, True)
        with torch.no_grad():
            y128 = netG(bb[DataType.MFCCS], bb[DataType.SESSION_INDEX], bb[DataType.SPEAKING_MODE_INDEX])
            y8 = netG(bb[DataType.MFCCS][:8].contiguous(), bb[DataType.SESSION_INDEX][:8].contiguous(),
                      bb[DataType.SPEAKING_MODE_INDEX][:8].contiguous())
        assert y128.shape == (128, 400, 8) and y8.shape == (8, 400, 8)
        assert float((y128[:8] - y8).abs().max()) <= 3e-3
        ops.set_compute_dtype("f32")
        with torch.no_grad():
            z128 = netG(bb[DataType.MFCCS], bb[DataType.SESSION_INDEX], bb[DataType.SPEAKING_MODE_INDEX])
            z8 = netG(bb[DataType.MFCCS][:8].contiguous(), bb[DataType.SESSION_INDEX][:8].contiguous(),
                      bb[DataType.SPEAKING_MODE_INDEX][:8].contiguous())
        assert float((z128[:8] - z8).abs().max()) <= 2e-5
        ops.set_compute_dtype("bf16")
        out = step(bb)
        st = step.read_stats(out)
        assert all(np.isfinite(v) for v in _losses(st)), st
        assert out["x_pred"].shape == (128, 400, 8) and st["num_phones"] == 128 * 25
        assert all(torch.isfinite(p).all() for p in netG.parameters())
        assert all(torch.isfinite(p).all() for p in netD.parameters())
    finally:
        ops.set_compute_dtype("f32")
