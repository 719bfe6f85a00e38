"""-m gpu: the full GAN train step (GANTrainStep = ste_gan/train.py:165-268 on MI355X) for two consecutive
steps against the golden fixture recorded from the reference modules + torch.optim.AdamW on CPU:
loss scalars, phoneme counters (bit-exact), generated EMG, per-tensor gradient norms of G and D, and the
per-tensor parameter norms after each optimiser step (incl. spectral-norm u / v buffers)."""
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


def make_cfg(channels):
    from ste_gan_amd.train_utils import DictConfig
    import yaml, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "configs/ste_gan_base_gantts.yaml")))
    cfg["data"] = yaml.safe_load(open(os.path.join(root, "configs/data/gaddy_and_klein_corpus.yaml")))
    cfg["model"]["params"] = {"channels": channels}
    return DictConfig(cfg)


def build(name):
    from ste_gan_amd.losses.emg_encoder_loss import EMGEncoderLoss
    from ste_gan_amd.losses.time_domain_loss import MultiTimeDomainFeatureLoss
    from ste_gan_amd.models.discriminator import init_emg_discriminators
    from ste_gan_amd.models.emg_encoder import EMGEncoderTransformer
    from ste_gan_amd.models.generator import init_emg_generator
    from ste_gan_amd.optim import FlatParams, FusedAdamW
    from ste_gan_amd.train import GANTrainStep
    ch, (ms, nl, ff), B, T, steps = cases.STEP_CASES[name]
    cfg = make_cfg(ch)
    netG = init_emg_generator(cfg)
    netG.load_state_dict(fill_state_dict(oshapes.generator_shapes("SPEECH_UNITS", ch)))
    netD = init_emg_discriminators(cfg)
    netD.load_state_dict(fill_state_dict(oshapes.discriminator_shapes(True)))
    enc = EMGEncoderTransformer(8, 256, 48, model_size=ms, num_transformer_layers=nl, dim_feedforward=ff)
    enc.load_state_dict(fill_state_dict(oshapes.encoder_shapes(ms, 3, nl, dim_feedforward=ff)))
    netG, netD, enc = netG.cuda(), netD.cuda(), enc.cuda().eval()
    from ste_gan_amd.train import make_optimizers
    optG, optD = make_optimizers(netG, netD)
    step = GANTrainStep(cfg, netG, netD, EMGEncoderLoss(enc), MultiTimeDomainFeatureLoss(8), optG, optD)
    return cfg, netG, netD, optG, optD, step


def to_batch(b):
    from ste_gan_amd.constants import DataType
    return {DataType.REAL_EMG: b["x_real"].cuda(), DataType.SPEECH_UNITS: b["su"].cuda(),
            DataType.SESSION_INDEX: b["sess"].cuda(), DataType.SPEAKING_MODE_INDEX: torch.zeros_like(b["sess"]).cuda(),
            DataType.PHONEMES: b["phonemes"].cuda()}


def norms(tensors):
    return np.array([float(t.double().norm()) for t in tensors])


@pytest.mark.parametrize("name", list(cases.STEP_CASES))
def test_train_step_vs_golden(name, golden):
    g = golden(name)
    cfg, netG, netD, optG, optD, step = build(name)
    steps = cases.STEP_CASES[name][4]
    for s in range(steps):
        out = step(to_batch(cases.step_batch(name, s)))
        st = step.read_stats(out)
        t = f"s{s}."
        ref = g[t + "losses"]        # [loss_D, loss_G, adv, td, su, ph, fm]
        close([st["discriminator"], st["generator"], st["multi_td"], st["speech_unit"], st["phoneme"],
               st["feature_matching"]], ref[[0, 1, 3, 4, 5, 6]], 1e-4, t + "losses")
        assert [st["num_phones"], st["num_silence_phones"], st["num_correct_phones"],
                st["num_correct_phones_no_silence"]] == g[t + "counters"].tolist()
        torch.testing.assert_close(out["x_pred"].cpu(), torch.from_numpy(g[t + "x_pred"]), rtol=1e-4, atol=1e-5)
        # Bounds from the measured distances (scripts/step_margins.py, round 5: G gradient norms 1.6e-6, D 1.0e-6 at step 0 and
        # 1.3e-4 at step 1 -- one leaky-ReLU unit of the fixture's 84 near-zero D pre-activations on the other side of zero after
        # D's first update, make_golden.py records the kink margins --, parameter norms 1.8e-7): an order of magnitude of head
        # room for G and the parameters, the fixture's own kink for D (round 4 held all of them to 2e-3)
        gG = norms([p.grad for p in netG.parameters()])
        np.testing.assert_allclose(gG, g[t + "g_grad_norm"], rtol=2e-5, atol=2e-6 * g[t + "g_grad_norm"].max())
        # after the G step, netD's .grad still holds the gradients of loss_D (this path never adds loss_G's D grads)
        gD = norms([p.grad for p in netD.parameters()])
        np.testing.assert_allclose(gD, g[t + "d_grad_norm"], rtol=2e-4, atol=2e-6 * g[t + "d_grad_norm"].max())
        close(norms(netG.state_dict().values()), g[t + "g_param_norm"], 2e-6, t + "g_param_norm")
        close(norms(netD.state_dict().values()), g[t + "d_param_norm"], 2e-6, t + "d_param_norm")

        def heads_of(tensors):
            return np.stack([np.pad(v.flatten()[:4].double().cpu().numpy(), (0, max(0, 4 - v.numel()))) for v in tensors])

        # element values, not only norms: the first four entries of every parameter of BOTH networks after the update, and of
        # the generator's gradients (absolute bounds relative to the largest entry: 5.7e-6 / 5.7e-7 measured)
        close(heads_of(netG.state_dict().values()), g[t + "g_param_head"], 2e-6, t + "g_param_head")
        close(heads_of(netD.state_dict().values()), g[t + "d_param_head"], 2e-6, t + "d_param_head")
        close(heads_of([p.grad for p in netG.parameters()]), g[t + "g_grad_head"], 2e-5, t + "g_grad_head")


def test_optimizer_state_dict_roundtrip():
    """FusedAdamW speaks torch.optim.AdamW's checkpoint format (checkpoint-*.pt interchange)."""
    from ste_gan_amd.optim import FlatParams, FusedAdamW
    torch.manual_seed(0)
    lin = torch.nn.Linear(7, 5).cuda()
    ref = torch.nn.Linear(7, 5).cuda()
    ref.load_state_dict(lin.state_dict())
    opt = FusedAdamW(FlatParams(lin))
    topt = torch.optim.AdamW(ref.parameters(), lr=2e-4, betas=(.8, .99))
    x = torch.randn(3, 7, device="cuda")
    for _ in range(3):
        opt.zero_grad(); topt.zero_grad()
        lin(x).square().sum().backward(); ref(x).square().sum().backward()
        opt.step(); topt.step()
    for a, b in zip(lin.parameters(), ref.parameters()):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-7)
    sd = opt.state_dict()
    topt2 = torch.optim.AdamW(ref.parameters(), lr=2e-4, betas=(.8, .99))
    topt2.load_state_dict(sd)                       # torch accepts our format
    opt2 = FusedAdamW(FlatParams(torch.nn.Linear(7, 5).cuda()))
    opt2.load_state_dict(topt.state_dict())         # and we accept torch's
    assert opt2.step_count == 3
    torch.testing.assert_close(opt2.exp_avg, opt.exp_avg, rtol=2e-5, atol=1e-7)


def test_train_step_with_rccl_process_group(golden):
    """The data-parallel code path on the GPU (RCCL all-reduce of the flat gradient arenas on a side stream, 1/world folded
    into AdamW), exercised with a 1-rank "nccl" group: must reproduce the golden two-step fixture exactly like the
    single-process path.  (Multi-rank equivalence is covered on CPU/gloo in tests/test_ddp_gloo.py.)"""
    import os
    import torch.distributed as dist
    name = "step_c48_b3"
    g = golden(name)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cfg, netG, netD, optG, optD, step = build(name)
        assert step.comm is not None and step.comm.world_size == 1
        for s in range(2):
            out = step(to_batch(cases.step_batch(name, s)))
            st = step.read_stats(out)
            ref = g[f"s{s}.losses"]
            close([st["discriminator"], st["generator"]], ref[[0, 1]], 1e-4, f"s{s}.losses")
            close(norms(netG.state_dict().values()), g[f"s{s}.g_param_norm"], 2e-5, f"s{s}.g_param_norm")
            close(norms(netD.state_dict().values()), g[f"s{s}.d_param_norm"], 2e-5, f"s{s}.d_param_norm")
    finally:
        dist.destroy_process_group()


def test_mfcc_variant_step_vs_oracle():
    """BASELINE config 5 geometry (MFCC input at 100 Hz, chunk 400 -> 50 MFCC frames / 25 SU frames / 400 EMG samples),
    reduced batch and widths: one full train step against the CPU oracle (no golden fixture for this variant: the oracle
    itself is pinned by the generator MFCC fixture and the SU-feature train-step fixture)."""
    from oracle import stegan_oracle as O
    from oracle.detfill import det_input, det_ints
    from ste_gan_amd.constants import DataType
    from ste_gan_amd.losses.emg_encoder_loss import EMGEncoderLoss
    from ste_gan_amd.losses.time_domain_loss import MultiTimeDomainFeatureLoss
    from ste_gan_amd.models.discriminator import init_emg_discriminators
    from ste_gan_amd.models.emg_encoder import EMGEncoderTransformer
    from ste_gan_amd.models.generator import init_emg_generator
    from ste_gan_amd.train import GANTrainStep, make_optimizers
    B, T = 2, 25
    cfg = make_cfg(32)
    cfg.model.speech_feature_type = "MFCCS"
    g_sd = fill_state_dict(oshapes.generator_shapes("MFCCS", 32))
    d_sd = fill_state_dict(oshapes.discriminator_shapes(True))
    e_sd = fill_state_dict(oshapes.encoder_shapes(64, 3, 2, dim_feedforward=128))
    netG = init_emg_generator(cfg); netG.load_state_dict(g_sd)
    netD = init_emg_discriminators(cfg); netD.load_state_dict(d_sd)
    enc = EMGEncoderTransformer(8, 256, 48, model_size=64, num_transformer_layers=2, dim_feedforward=128); enc.load_state_dict(e_sd)
    netG, netD, enc = netG.cuda(), netD.cuda(), enc.cuda().eval()
    optG, optD = make_optimizers(netG, netD)
    step = GANTrainStep(cfg, netG, netD, EMGEncoderLoss(enc), MultiTimeDomainFeatureLoss(8), optG, optD)
    mfcc = det_input("mf.feats", (B, 2 * T, 25)); su = det_input("mf.su", (B, T, 256))
    x_real = det_input("mf.x", (B, 16 * T, 8), "tanh"); sess = det_ints("mf.s", (B,), 17); ph = det_ints("mf.ph", (B, T), 48)
    batch = {DataType.REAL_EMG: x_real.cuda(), DataType.SPEECH_UNITS: su.cuda(), DataType.MFCCS: mfcc.cuda(),
             DataType.SESSION_INDEX: sess.cuda(), DataType.SPEAKING_MODE_INDEX: torch.zeros_like(sess).cuda(),
             DataType.PHONEMES: ph.cuda()}
    st = step.read_stats(step(batch))
    d_train = {k: v for k, v in d_sd.items() if not O.is_sn_buffer(d_sd, k)}
    o = O.train_step(g_sd, d_sd, e_sd, O.adamw_init(g_sd), O.adamw_init(d_train),
                     {"su": su, "feats": mfcc, "x_real": x_real, "sess": sess, "phonemes": ph}, speech_feature_type="MFCCS")
    assert step.last["x_pred"].shape == (B, 400, 8)
    close([st["discriminator"], st["generator"], st["multi_td"], st["speech_unit"], st["phoneme"], st["feature_matching"]],
          [o["loss_d"], o["loss_g"], o["loss_td"], o["loss_su"], o["loss_ph"], o["loss_fm"]], 1e-4, "losses")
    assert st["num_correct_phones"] == o["num_correct_phones"] and st["num_silence_phones"] == o["num_silence_phones"]
    close(norms(netG.state_dict().values()), norms(g_sd.values()), 2e-5, "g params after step")
    close(norms(netD.state_dict().values()), norms(d_sd.values()), 2e-5, "d params after step")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_full_discriminator_step_vs_oracle(dtype):
    """`model.discriminator_small: false` (discriminator.py:158-191: k5 s3 period stacks to 1024 channels, k41 grouped scale
    stacks, SEVEN spectral-norm layers) through the whole train step -- batched discriminator pass with split targets, the
    rebalanced streams, one-launch spectral norm -- against the CPU oracle's train step: fp32 within 1e-4, bf16 within the
    tolerances of the bf16 mode (test_gpu_train_step_bf16.py)."""
    from oracle import stegan_oracle as O
    from oracle.detfill import det_input, det_ints
    from ste_gan_amd import ops
    from ste_gan_amd.constants import DataType
    from ste_gan_amd.losses.emg_encoder_loss import EMGEncoderLoss
    from ste_gan_amd.losses.time_domain_loss import MultiTimeDomainFeatureLoss
    from ste_gan_amd.models.discriminator import init_emg_discriminators
    from ste_gan_amd.models.emg_encoder import EMGEncoderTransformer
    from ste_gan_amd.models.generator import init_emg_generator
    from ste_gan_amd.train import GANTrainStep, make_optimizers
    B, T = 2, 24
    cfg = make_cfg(32)
    cfg.model.discriminator_small = False
    g_sd = fill_state_dict(oshapes.generator_shapes("SPEECH_UNITS", 32))
    d_sd = fill_state_dict(oshapes.discriminator_shapes(False))
    e_sd = fill_state_dict(oshapes.encoder_shapes(64, 3, 2, dim_feedforward=128))
    ops.set_compute_dtype(dtype)
    try:
        netG = init_emg_generator(cfg); netG.load_state_dict(g_sd)
        netD = init_emg_discriminators(cfg); netD.load_state_dict(d_sd)
        enc = EMGEncoderTransformer(8, 256, 48, model_size=64, num_transformer_layers=2, dim_feedforward=128); enc.load_state_dict(e_sd)
        netG, netD, enc = netG.cuda(), netD.cuda(), enc.cuda().eval()
        optG, optD = make_optimizers(netG, netD)
        step = GANTrainStep(cfg, netG, netD, EMGEncoderLoss(enc), MultiTimeDomainFeatureLoss(8), optG, optD)
        su = det_input("fd.su", (B, T, 256)); x_real = det_input("fd.x", (B, 16 * T, 8), "tanh")
        sess = det_ints("fd.s", (B,), 17); ph = det_ints("fd.ph", (B, T), 48)
        batch = {DataType.REAL_EMG: x_real.cuda(), DataType.SPEECH_UNITS: su.cuda(), DataType.SESSION_INDEX: sess.cuda(),
                 DataType.SPEAKING_MODE_INDEX: torch.zeros_like(sess).cuda(), DataType.PHONEMES: ph.cuda()}
        st = step.read_stats(step(batch))
    finally:
        ops.set_compute_dtype("f32")
    d_train = {k: v for k, v in d_sd.items() if not O.is_sn_buffer(d_sd, k)}
    o = O.train_step(g_sd, d_sd, e_sd, O.adamw_init(g_sd), O.adamw_init(d_train),
                     {"su": su, "feats": su, "x_real": x_real, "sess": sess, "phonemes": ph}, small_d=False)
    got = [st["discriminator"], st["generator"], st["multi_td"], st["speech_unit"], st["phoneme"], st["feature_matching"]]
    ref = [o["loss_d"], o["loss_g"], o["loss_td"], o["loss_su"], o["loss_ph"], o["loss_fm"]]
    if dtype == "f32":
        close(got, ref, 1e-4, "losses")
        assert st["num_correct_phones"] == o["num_correct_phones"]
        close(norms(netG.state_dict().values()), norms(g_sd.values()), 2e-5, "g params after step")
        close(norms(netD.state_dict().values()), norms(d_sd.values()), 2e-5, "d params after step")
    else:
        np.testing.assert_allclose(np.array(got), np.array(ref, dtype=np.float64), rtol=3e-2)
        np.testing.assert_allclose(norms(netG.state_dict().values()), norms(g_sd.values()), rtol=1e-3)
        np.testing.assert_allclose(norms(netD.state_dict().values()), norms(d_sd.values()), rtol=1e-3)


def test_steps_queued_without_host_sync_match_golden(golden):
    """The training loop does not synchronise per step, so the host may be several steps ahead of the stream: nothing a
    queued step reads (AdamW step count / bias corrections, prepared weights) may depend on host state at execution time."""
    name = "step_c48_b3"
    g = golden(name)
    cfg, netG, netD, optG, optD, step = build(name)
    outs = [step(to_batch(cases.step_batch(name, s))) for s in range(2)]          # no .item() / sync in between
    torch.cuda.synchronize()
    close(norms(netG.state_dict().values()), g["s1.g_param_norm"], 2e-5, "s1.g_param_norm")
    close(norms(netD.state_dict().values()), g["s1.d_param_norm"], 2e-5, "s1.d_param_norm")
    st = step.read_stats(outs[1])
    close([st["discriminator"], st["generator"]], g["s1.losses"][[0, 1]], 1e-4, "s1.losses")
    assert optG.step_count == 2 and int(optG._hyper.view(torch.int32)[3]) == 2


def test_graph_replay_matches_golden(golden):
    """hipGraph capture of the whole step: replaying it for the two fixture batches must give the eager results."""
    from ste_gan_amd.train import GraphedTrainStep
    name = "step_c48_b3"
    g = golden(name)
    cfg, netG, netD, optG, optD, step = build(name)
    sdG = {k: v.clone() for k, v in netG.state_dict().items()}
    sdD = {k: v.clone() for k, v in netD.state_dict().items()}
    gs = GraphedTrainStep(step, to_batch(cases.step_batch(name, 0)), warmup=2)
    # warm-up steps trained the nets: restore the fixture's initial state (in place: captured pointers stay valid)
    netG.load_state_dict(sdG); netD.load_state_dict(sdD)
    for opt in (optG, optD):
        opt.reset_state()
    for s in range(2):
        out = gs(to_batch(cases.step_batch(name, s)))
        st = gs.read_stats(out)
        ref = g[f"s{s}.losses"]
        close([st["discriminator"], st["generator"], st["multi_td"], st["speech_unit"], st["phoneme"], st["feature_matching"]],
              ref[[0, 1, 3, 4, 5, 6]], 1e-4, f"s{s}.losses")
        assert [st["num_phones"], st["num_silence_phones"], st["num_correct_phones"],
                st["num_correct_phones_no_silence"]] == g[f"s{s}.counters"].tolist()
        close(norms(netG.state_dict().values()), g[f"s{s}.g_param_norm"], 2e-5, f"s{s}.g_param_norm")
        close(norms(netD.state_dict().values()), g[f"s{s}.d_param_norm"], 2e-5, f"s{s}.d_param_norm")


@pytest.mark.parametrize("split_side", ["1", "0"])
def test_graph_replay_with_process_group_matches_golden(split_side, monkeypatch):
    """Under torch.distributed GraphedTrainStep captures the step as three hipGraphs (one memory pool) with the all-reduces
    launched eagerly between them (the side branch -- multi-TD + frozen encoder -- cut in two, forward in the first graph,
    backward in the second; STEGAN_SPLIT_SIDE=0: whole in the first, the round-2 arrangement).  With a 1-rank RCCL group every
    arrangement must reproduce the golden two-step fixture like the single-process graph and the eager path -- IN THIS
    PROCESS, ending in the ordered teardown (parallel.shutdown: drain, drop graphs + pool + comm stream,
    destroy_process_group).  Round 2 ran this in a child that left through os._exit: the abort it dodged was the
    capture-mode race fixed since (GraphedTrainStep.CAPTURE_ERROR_MODE, pinned in tests/test_host_logic.py)."""
    import torch.distributed as dist
    import ddp_graph_worker
    monkeypatch.setenv("STEGAN_DDP_GRAPH", "three")
    monkeypatch.setenv("STEGAN_SPLIT_SIDE", split_side)
    assert not dist.is_initialized()
    ddp_graph_worker.main()
    assert not dist.is_initialized()


def test_single_graph_with_captured_collectives_matches_golden():
    """STEGAN_DDP_GRAPH=single (opt-in): ONE hipGraph per step with the two RCCL all-reduces captured inside it.  Runs in a
    child process that prints the Python and C++ stacks of every thread on a fatal signal (tests/ddp_graph_worker.py).  Round 4
    saw this arrangement die with a silent SIGABRT during the capture in about one of six runs; round 5 took the host-side
    work out of the captured launch (parallel.FlatGradAllReduce: no event / stream construction, no record_stream while
    capturing).  ONLY that known signal is reported as xfail -- with the stacks the child printed --; a timeout, any other exit
    code or wrong numbers FAIL."""
    import os
    import signal
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, STEGAN_DDP_GRAPH="single", STEGAN_SPLIT_SIDE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(here, "ddp_graph_worker.py")], env=env, capture_output=True, text=True,
                       timeout=600)                      # (TimeoutExpired propagates: a hang is a failure)
    if r.returncode == -signal.SIGABRT:
        pytest.xfail("the opt-in single-graph capture with RCCL collectives died with SIGABRT; stacks:\n" + r.stderr[-6000:])
    assert r.returncode == 0, f"exit {r.returncode}\n" + r.stdout[-2000:] + r.stderr[-6000:]
    assert "RESULT OK" in r.stdout, r.stdout[-2000:]


def test_failed_single_graph_capture_falls_back_to_three_graphs_with_clean_state(monkeypatch):
    """The fallback the round-4 advisor found untested: the single-graph capture RAISES half way (here: inside the first
    captured gradient exchange), GraphedTrainStep cleans up (GANTrainStep.abort on a capturing stream: nothing recorded,
    arenas marked dirty) and captures the three-graph arrangement instead -- whose replays must reproduce the golden
    fixture, i.e. no state of the failed capture (gradient scale, parked gradients, zero pool, arena contents) survived."""
    import torch.distributed as dist
    import ddp_graph_worker
    from ste_gan_amd.train import GANTrainStep
    monkeypatch.setenv("STEGAN_DDP_GRAPH", "single")
    monkeypatch.setenv("STEGAN_SPLIT_SIDE", "1")
    real = GANTrainStep.exchange_d
    fired = []

    def failing(self):
        if torch.cuda.is_current_stream_capturing() and not fired:
            fired.append(1)
            raise RuntimeError("injected: the captured exchange fails")
        return real(self)

    monkeypatch.setattr(GANTrainStep, "exchange_d", failing)
    assert not dist.is_initialized()
    ddp_graph_worker.main(expect="three")
    assert fired and not dist.is_initialized()


def test_a_loss_term_with_weight_zero_is_logged_with_its_value_and_trains_nothing():
    """The reference logs every enabled loss term itself (train.py:213-262), whatever its weight.  With the multi-TD and the
    feature-matching weights set to 0 the step reports the same term values as the default step does on the same weights and
    batch, the generator loss loses exactly those two summands, and the generated EMG (step 0: before any update) is equal."""
    name = "step_c48_b3"
    batch = to_batch(cases.step_batch(name, 0))
    cfg, netG, netD, optG, optD, step = build(name)
    ref = step.read_stats(step(batch))
    cfg2, netG2, netD2, optG2, optD2, step2 = build(name)
    w_td, w_fm = float(cfg2.train.loss_multi_td_weight), float(cfg2.train.loss_feat_match_weight)
    cfg2.train.loss_multi_td_weight = 0.0
    cfg2.train.loss_feat_match_weight = 0.0
    got = step2.read_stats(step2(batch))
    assert abs(got["multi_td"] - ref["multi_td"]) <= 1e-6 * abs(ref["multi_td"]) and ref["multi_td"] != 0.0
    assert abs(got["feature_matching"] - ref["feature_matching"]) <= 1e-6 * abs(ref["feature_matching"])
    want = ref["generator"] - w_td * ref["multi_td"] - w_fm * ref["feature_matching"]
    assert abs(got["generator"] - want) <= 1e-5 * abs(ref["generator"]), (got["generator"], want)
    assert got["discriminator"] == ref["discriminator"]
