"""CPU: host-side mirror of the reference interface (config surface, state-dict layouts, error behaviour)."""
import argparse
import os

import pytest
import torch

from oracle import shapes as oshapes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_state_dict_layouts_match_reference_layouts():
    from ste_gan_amd.models.discriminator import Discriminator, DiscriminatorSmall
    from ste_gan_amd.models.emg_encoder import EMGEncoderTransformer
    from ste_gan_amd.models.generator import EMGGeneratorGanTTS
    def layout(m):
        return [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert layout(EMGGeneratorGanTTS("SPEECH_UNITS", 256, 17, 8, channels=48)) == list(oshapes.generator_shapes("SPEECH_UNITS", 48).items())
    assert layout(EMGGeneratorGanTTS("MFCCS", 25, 17, 8, channels=48)) == list(oshapes.generator_shapes("MFCCS", 48).items())
    assert layout(DiscriminatorSmall(8)) == list(oshapes.discriminator_shapes(True).items())
    assert layout(Discriminator(8)) == list(oshapes.discriminator_shapes(False).items())
    assert layout(EMGEncoderTransformer(8, 256, 48, model_size=64, num_transformer_layers=2, dim_feedforward=128)) == \
        list(oshapes.encoder_shapes(64, 3, 2, dim_feedforward=128).items())


def test_no_cpu_fallback():
    from ste_gan_amd.models.generator import EMGGeneratorGanTTS
    g = EMGGeneratorGanTTS("SPEECH_UNITS", 256, 17, 8, channels=16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        g(torch.zeros(1, 2, 256), torch.zeros(1, dtype=torch.long), torch.zeros(1, dtype=torch.long))


def test_factories_and_errors_follow_the_reference():
    from ste_gan_amd.models.discriminator import Discriminator, DiscriminatorSmall, init_emg_discriminators
    from ste_gan_amd.models.generator import init_emg_generator
    from ste_gan_amd.train_utils import DictConfig
    cfg = DictConfig({"model": {"type": "EMGGeneratorGanTTS", "speech_feature_type": "MFCCS", "discriminator_small": False,
                                "params": {"channels": 32}},
                      "data": {"num_emg_channels": 8, "num_emg_sessions": 17}})
    g = init_emg_generator(cfg)
    assert g.speech_input_dim == 25 and g.gblocks[0].cfg.cin == 25 + 64 and g.gblocks[6].res1[0].cfg.upsample == 1
    assert isinstance(init_emg_discriminators(cfg), Discriminator)
    cfg.model.discriminator_small = True
    d = init_emg_discriminators(cfg)
    assert isinstance(d, DiscriminatorSmall) and d.discriminator_names[0] == "DiscriminatorP-2" and len(d.discriminator_names) == 8
    cfg.model.speech_feature_type = "WAVEFORM"
    with pytest.raises(ValueError, match="Unrecognized speech feature type"):
        init_emg_generator(cfg)
    cfg.model.speech_feature_type = "SPEECH_UNITS"
    with pytest.raises(ValueError, match="Unrecognized EMG generator type"):
        init_emg_generator(cfg, "SomethingElse")


def test_config_surface_and_cli_overrides():
    from ste_gan_amd.train import crop_to_hop, parse_args
    from ste_gan_amd.train_utils import create_ste_gan_model_name, fix_state_dict, load_config
    os.chdir(ROOT)
    args = parse_args(["--batch_size", "64", "--chunk_size", "200", "--weight_su", "0", "--weight_td", "3.5"])
    cfg = load_config(args)
    assert cfg.train.batch_size == 64 and cfg.train.chunk_size == 200 and cfg.train.loss_multi_td_weight == 3.5
    assert cfg.train.loss_speech_unit_error is False and cfg.train.loss_phoneme_error is True   # weight < 1e-3 switches a loss off
    assert cfg.data.num_emg_sessions == 17 and cfg.emg_encoder.params.model_size == 768
    assert crop_to_hop(200) == 192 and crop_to_hop(2048) == 2048
    name = create_ste_gan_model_name(cfg, add_timestamp=False)
    assert name == "gaddy_voiced_EMGGeneratorGanTTS_SPEECH_UNITS_small_dis_chunk_200_with_adv_loss_fmw_7.0_tdw_3.5_suw_0.0_phw_1.0_wv_0.0"
    sd = fix_state_dict({"_orig_mod.gblocks.0.bias": 1, "last_conv.1.bias": 2})
    assert list(sd) == ["gblocks.0.bias", "last_conv.1.bias"]


def test_constants_namespace():
    import ste_gan_amd as K
    assert K.HOPSIZE == 16 and K.SILENCE_PHONEME_INDEX == 47 and len(K.PHONEME_INVENTORY) == 48
    assert K.DataType.SPEAKING_MODE_INDEX == "SPEAKING_MODE_IDX" and K.DataType.REAL_EMG == "REAL_EMG"
    opt = K.OPTIMIZER([torch.nn.Parameter(torch.zeros(1))])
    assert opt.defaults["lr"] == 2e-4 and opt.defaults["betas"] == (0.8, 0.99) and opt.defaults["weight_decay"] == 1e-2


def test_conv_cfg_output_length_and_error():
    from ste_gan_amd.ops import ConvCfg
    assert ConvCfg(8, 8, 3, dilation=27, pad=27).tout(12) == 12
    assert ConvCfg(8, 8, 37, stride=2, pad=18, groups=4).tout(192) == 96
    assert ConvCfg(8, 8, 3, stride=3, pad=2).tout(99) == 34
    assert ConvCfg(8, 8, 3, pad=1, upsample=2).tout(12) == 24
    with pytest.raises(RuntimeError, match="Kernel size can't be greater"):
        ConvCfg(8, 8, 41).tout(8)


def test_fold_window_bookkeeping_on_the_host():
    """ops.fm_fold_begin / fm_fold_end: a parked feature-matching gradient that no producer collected makes the end of the
    window raise (a graph shape the shortcut does not cover must fail loudly, not drop a loss term); the window state is
    reset either way."""
    from ste_gan_amd import ops
    ops.fm_fold_begin()
    assert ops._FM_WINDOW and not ops._FM_PENDING and not ops._FM_FOLDABLE
    ops.fm_fold_end()                                    # nothing parked: fine
    assert not ops._FM_WINDOW
    ops.fm_fold_begin()
    ops._FM_PENDING[(4096, 16)] = torch.zeros(16)
    with pytest.raises(RuntimeError, match="never collected"):
        ops.fm_fold_end()
    assert not ops._FM_WINDOW and not ops._FM_PENDING
    ops.fm_fold_begin()
    ops._FM_FOLDABLE.add((1, 2))
    ops.fm_fold_abort()
    assert not ops._FM_WINDOW and not ops._FM_FOLDABLE


def test_discriminator_pair_loss_rejects_unequal_halves():
    """lsgan_pair_loss prices the concatenated fake | real logits as one term of weight 2: only valid for equal halves."""
    from ste_gan_amd.losses.gan_losses import lsgan_pair_loss
    with pytest.raises(ValueError, match="two equal halves"):
        lsgan_pair_loss([("both", torch.zeros(5, 1, 7), 2)])
    from ste_gan_amd import ops
    with pytest.raises(ValueError, match="fold requires unit_grad"):
        ops.multi_pair_loss([torch.zeros(3)], [torch.zeros(3)], 0, fold=True)


def test_train_step_phases_exist_for_segmented_capture():
    """GraphedTrainStep captures GANTrainStep.phase_d / phase_g / phase_opt as three graphs under torch.distributed: the
    eager step must be exactly their composition with the two exchanges in between."""
    import inspect
    from ste_gan_amd.train import GANTrainStep
    src = inspect.getsource(GANTrainStep.__call__)
    order = [src.index(s) for s in ("self.phase_d(", "self.exchange_d(", "self.phase_g(", "self.exchange_g(", "self.phase_opt(")]
    assert order == sorted(order)


def test_graph_capture_error_mode_is_thread_local():
    """Round 2's teardown abort: a capture in the default "global" error mode is invalidated by the RCCL watchdog thread's
    hipEventQuery.  The fix is pinned here (CPU) instead of by re-running the abort on a GPU box."""
    import inspect
    from ste_gan_amd.train import GraphedTrainStep
    assert GraphedTrainStep.CAPTURE_ERROR_MODE == "thread_local"
    src = inspect.getsource(GraphedTrainStep.__init__)
    assert "capture_error_mode=self.CAPTURE_ERROR_MODE" in src and "environ.get(\"STEGAN_CAPTURE" not in src


def test_zero_pool_invariants_on_cpu(monkeypatch):
    """ops.pooled_zeros(): zeros whatever happened before -- a step that died (abort), a later step that needs more than the
    one before it, two alternating buffers so that a step's returned scalars survive the next step's fill."""
    from ste_gan_amd import ops
    monkeypatch.setattr(ops, "ZERO_POOL_BYTES", 1 << 16)
    monkeypatch.setattr(ops, "_ZPOOL", {"bufs": [None, None], "dirty": [0, 0], "floor": [0, 0], "cur": 0, "off": 0,
                                        "active": False, "filled": 0})
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    dev = torch.device("cpu")
    # outside a step: plain torch.zeros, the pool is not touched
    a = ops.pooled_zeros(8, dev)
    assert a.sum() == 0 and ops._ZPOOL["bufs"] == [None, None]
    # step 1 (small), its scalar survives step 2's begin (other buffer)
    ops.zero_pool_begin(dev)
    s1 = ops.pooled_zeros((), dev); s1.fill_(7.0)
    ops.zero_pool_end()
    ops.zero_pool_begin(dev)
    assert float(s1) == 7.0
    t = ops.pooled_zeros(1000, dev); t.fill_(3.0)
    # ... dies half way: the pool is closed and wholly dirty
    ops.zero_pool_abort()
    assert not ops._ZPOOL["active"]
    assert ops.pooled_zeros(4, dev).data_ptr() not in range(ops._ZPOOL["bufs"][ops._ZPOOL["cur"]].data_ptr(),
                                                            ops._ZPOOL["bufs"][ops._ZPOOL["cur"]].data_ptr() + (1 << 16))
    # steps 3 and 4 reuse both buffers and need MORE than any step before: still zeros
    for _ in range(2):
        ops.zero_pool_begin(dev)
        big = ops.pooled_zeros(8000, dev)
        assert float(big.abs().sum()) == 0.0
        big.fill_(1.0)
        ops.zero_pool_end()
    for _ in range(2):
        ops.zero_pool_begin(dev)
        bigger = ops.pooled_zeros(12000, dev)
        assert float(bigger.abs().sum()) == 0.0
        bigger.fill_(2.0)
        ops.zero_pool_end()
    # a request that does not fit falls back to a fresh fill
    ops.zero_pool_begin(dev)
    assert ops.pooled_zeros(1 << 20, dev).numel() == 1 << 20
    ops.zero_pool_end()


def test_zero_pool_capture_must_be_covered_by_its_fill(monkeypatch):
    from ste_gan_amd import ops
    monkeypatch.setattr(ops, "ZERO_POOL_BYTES", 1 << 16)
    monkeypatch.setattr(ops, "_ZPOOL", {"bufs": [None, None], "dirty": [0, 0], "floor": [0, 0], "cur": 0, "off": 0,
                                        "active": False, "filled": 0})
    capturing = [False]
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: capturing[0])
    dev = torch.device("cpu")
    ops.zero_pool_begin(dev); ops.pooled_zeros(100, dev); ops.zero_pool_end()          # eager warm-up: 400 bytes
    ops.zero_pool_begin(dev); ops.pooled_zeros(100, dev); ops.zero_pool_end()
    capturing[0] = True
    cur = ops._ZPOOL["cur"]
    ops.zero_pool_begin(dev)
    assert ops._ZPOOL["cur"] == cur                        # a capture keeps the last eager step's buffer
    ops.pooled_zeros(100, dev)
    ops.zero_pool_end()                                    # covered
    assert ops._ZPOOL["floor"][cur] >= 400
    ops.zero_pool_begin(dev)
    ops.pooled_zeros(5000, dev)
    with pytest.raises(RuntimeError, match="run the step eagerly once"):
        ops.zero_pool_end()
    capturing[0] = False
    # an eager step on the captured step's buffer clears at least what the replays dirty
    ops._ZPOOL["cur"] = cur ^ 1
    ops.zero_pool_begin(dev)
    assert ops._ZPOOL["cur"] == cur and ops._ZPOOL["filled"] >= ops._ZPOOL["floor"][cur]
    ops.zero_pool_end()


def test_bf16_image_is_dropped_after_an_in_place_write(monkeypatch):
    """ops._image_of: autograd's gradient buffers (and users) may add_ into a tensor that carries a bf16 image; the image
    then shows the old values, so the in-place version recorded when it was attached must still match."""
    from ste_gan_amd import ops
    monkeypatch.setattr(ops, "COMPUTE_DTYPE", "bf16")
    monkeypatch.setattr(ops, "HALF", True)
    monkeypatch.setattr(ops, "HALF_DTYPE", torch.bfloat16)
    monkeypatch.setattr(ops, "IMAGES", True)
    t = torch.zeros(2, 16, 12)
    h = torch.zeros(2, 2, 12, 8, dtype=torch.bfloat16)
    ops._attach_image(t, h)
    assert ops._image_of(t, 16, 12) is h
    assert ops._image_of(t, 16, 11) is None and ops._image_of(t, 24, 12) is None
    t.add_(1.0)
    assert ops._image_of(t, 16, 12) is None
    u = torch.zeros(2, 16, 12)
    u.h = h                                               # attached by hand, no version: not trusted
    assert ops._image_of(u, 16, 12) is None
    # an image of the OTHER 2-byte type (left over from before a mode switch) is not an image of this mode
    v = torch.zeros(2, 16, 12)
    ops._attach_image(v, h)
    monkeypatch.setattr(ops, "HALF_DTYPE", torch.float16)
    assert ops._image_of(v, 16, 12) is None


def test_zero_on_consume_flags():
    """FlatParams / ConvPack skip their fill in zero_grad() only while the arena is known to be clean (the optimiser launch
    consumed it); anything else -- a fresh arena, a step that died after backward -- gets a real fill."""
    from ste_gan_amd.optim import FlatParams

    class _Flat(FlatParams):
        def __init__(self):
            self.grad = torch.ones(8)
            self.params, self.offsets = [], []
    f = _Flat()
    f.zero_in_step = True
    f.zero_grad()                                          # not known clean: filled
    assert float(f.grad.abs().sum()) == 0 and f.clean is False
    f.grad.fill_(1.0); f.clean = True                      # (as after a consuming AdamW launch, whose kernel zeroes it)
    f.zero_grad()
    assert float(f.grad.sum()) == 8.0 and f.clean is False           # skipped: trusts the kernel
    f.zero_grad()                                          # a second call without a consuming launch in between: filled
    assert float(f.grad.abs().sum()) == 0
