"""-m gpu: the bf16 compute mode against the oracle run on bf16-ROUNDED OPERANDS (``oracle.stegan_oracle.bf16_operands``).

The fp32 reference cannot bound the bf16 step tighter than a few per cent (test_gpu_train_step_bf16.py: 3 % on losses, 8 % on
gradient norms) -- that is the size of bf16 rounding itself, not of a mistake.  Inside ``bf16_operands()`` the oracle rounds
the two operands of every contraction exactly where the HIP kernels do (forward: x and the raw weights, the per-channel
factor applied in fp32 behind the sum; input gradient: dy and the factor-folded weights; weight gradient: x and dy; the
single-output-channel convs and groups narrower than 8 / 16 channels exact), so what remains between the two is fp32
summation order -- and its one amplifier: where two fp32 values differ in their last bit, a bf16 rounding behind them can
fall to the other side (2^-9 relative, one element), and a leaky-ReLU mask or a ReLU can flip on a unit that sits at zero.

(1) LAYER BY LAYER, both sides fed the oracle's intermediate: forward, input gradient and every parameter gradient of every
    conv layer of the eight discriminators, every generator block and every encoder stage -- 335 tensors -- agree to
    <= 1e-4 relative L2 (measured <= 2.9e-5: a handful of flipped roundings per tensor).  This pins the rounding points.
(2) THE WHOLE STEP against ``oracle.train_step`` under the same flag (measured on MI355X in brackets).  Config-3 size,
    one step: losses <= 1e-3 [7.5e-5], per-tensor gradient norms <= 1.2e-3 for G [3.8e-4] and <= 1e-4 for D [1.9e-5],
    generated EMG <= 2e-3 absolute [5.1e-4].  Reduced-width fixture, step 0: losses <= 1e-4 [1.1e-6], generated EMG
    <= 1e-5 [3e-8], D gradients <= 1e-4 per tensor [1.3e-5]; the generator's gradients come through the frozen
    discriminators and the frozen encoder, where single flips weigh more at these sizes (a 64-channel x 5-frame encoder
    map: one flipped rounding in block 2 is 4e-4 of the map two stages later; one flipped leaky mask in a 3 x 512 x 5
    feature map moves 160 input-gradient elements by 6e-4): the whole gradient vector <= 1e-3 [3.0e-4], per-tensor norms
    <= 5e-3 [1.7e-3].  Step 1 sits behind an AdamW update, whose first step moves every weight by lr x sign(gradient) --
    the least continuous function in the loop -- and is bounded on the losses (1e-3 [1.7e-4]) and the whole-vector errors.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cases
from oracle import shapes as oshapes
from oracle import stegan_oracle as O
from oracle.detfill import det_input, fill_state_dict

pytestmark = pytest.mark.gpu

LAYER_TOL = 1e-4
# the operand type under test: tests/test_gpu_f16_mode.py re-runs the layer test with ("f16", torch.float16)
OPERAND = ("bf16", torch.bfloat16)


@pytest.fixture
def bf16_mode():
    from ste_gan_amd import ops
    ops.set_compute_dtype(OPERAND[0])
    yield ops
    ops.set_compute_dtype("f32")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _leaves(sd, prefix):
    return {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith(prefix + ".") and v.is_floating_point()}


def _compare(tag, module, prefix, sd, run_oracle, run_build, x_in, worst, with_params=True, to_build=lambda t: t.cuda()):
    """forward + input gradient + parameter gradients of one stage, both sides fed ``x_in``."""
    leaves = _leaves(sd, prefix) if with_params else {}
    sd_l = dict(sd); sd_l.update(leaves)
    xo = x_in.clone().requires_grad_(True)
    xb = to_build(x_in).detach().requires_grad_(True)
    _with_image(xb)
    with O.bf16_operands(OPERAND[1]):
        yo = run_oracle(sd_l, xo)
    yb = run_build(xb)
    assert yb.shape == yo.shape, (tag, yb.shape, yo.shape)
    dy = det_input("emu.dy." + tag, tuple(yo.shape))
    pb = dict(module.named_parameters()) if with_params else {}
    names = [k for k in leaves if k[len(prefix) + 1:] in pb]             # (spectral-norm u / v are buffers, not parameters)
    go = torch.autograd.grad(yo, [xo] + [leaves[k] for k in names], dy, allow_unused=True)
    gb = torch.autograd.grad(yb, [xb] + [pb[k[len(prefix) + 1:]] for k in names], dy.cuda(), allow_unused=True)
    errs = {"fwd": rel(yb, yo), "dx": rel(gb[0] if gb[0].shape == x_in.shape else _from_build(gb[0], x_in), go[0])}
    ref = dict(zip(names, go[1:]))
    for k, a, b in zip(names, gb[1:], go[1:]):
        if b is None:
            continue
        short = k[len(prefix) + 1:]
        if short.endswith("weight_g"):
            # dL/dg = <dW, v / |v|> per output channel: a projection of the weight gradient that may cancel to a small
            # fraction of it, so its error is measured against |dW|, recovered from the orthogonal pair of gradients:
            # |dW|^2 = (dL/dg)^2 + (|v| / g)^2 |dL/dv|^2
            vk = k[:-1] + "v"
            vn = sd[vk].flatten(1).norm(dim=1).double(); gg = sd[k].flatten().double()
            dwn = (b.double().flatten().pow(2) + (vn / gg).pow(2) * ref[vk].double().flatten(1).pow(2).sum(1)).sum().sqrt()
            errs[short] = float((a.double().cpu() - b.double()).norm() / dwn)
        else:
            errs[short] = rel(a, b)
    for k, v in errs.items():
        worst.append((v, tag, k))


def _with_image(t):
    """Inside the step every activation of >= 16 channels arrives with the bf16 image its producer wrote (ops._new_image);
    the test's hand-made inputs get the same.  (It matters for ONE decision: the bf16 weight-gradient kernel stages a
    stride-3 source without an image through registers, runs out of its per-thread item budget on the period
    discriminators' layers and leaves them to the exact fp32 kernel -- more exact, but not what the step runs.)"""
    from ste_gan_amd import ops
    from test_gpu_conv_bf16 import make_image
    if t.dim() == 3 and t.shape[1] % 8 == 0 and t.shape[1] >= 16:
        ops._attach_image(t, make_image(t.detach()))
    return t


def _from_build(g, like):
    return g.reshape(like.shape[0], like.shape[3], like.shape[1], like.shape[2]).permute(0, 2, 3, 1) if like.dim() == 4 else g


def _nets(name):
    """The fixture's networks WITHOUT optimisers: parameter gradients come back through autograd (no gradient arena)."""
    from test_gpu_train_step import make_cfg
    from ste_gan_amd.models.discriminator import init_emg_discriminators
    from ste_gan_amd.models.emg_encoder import EMGEncoderTransformer
    from ste_gan_amd.models.generator import init_emg_generator
    ch, (ms, nl, ff), B, T, steps = cases.STEP_CASES[name]
    cfg = make_cfg(ch)
    g_sd = fill_state_dict(oshapes.generator_shapes("SPEECH_UNITS", ch))
    d_sd = fill_state_dict(oshapes.discriminator_shapes(True))
    e_sd = fill_state_dict(oshapes.encoder_shapes(ms, 3, nl, dim_feedforward=ff))
    netG = init_emg_generator(cfg); netG.load_state_dict(g_sd)
    netD = init_emg_discriminators(cfg); netD.load_state_dict(d_sd)
    enc = EMGEncoderTransformer(8, 256, 48, model_size=ms, num_transformer_layers=nl, dim_feedforward=ff)
    enc.load_state_dict(e_sd)
    return (g_sd, d_sd, e_sd), (netG.cuda(), netD.cuda().eval(), enc.cuda().eval()), (ch, B, T)


def test_every_layer_matches_the_bf16_operand_oracle(bf16_mode):
    ops = bf16_mode
    name = next(iter(cases.STEP_CASES))
    (g_sd, d_sd, e_sd), (netG, netD, enc), (ch, B, T) = _nets(name)
    worst = []
    netD._prepare_weights()
    x_emg = det_input("emu.x", (B, 16 * T, 8), "tanh")
    x_bct = x_emg.transpose(1, 2).contiguous()

    # ---- period discriminators: layer j of D_p on the oracle's (B, C, H, p) map; the build folds it to (B p, C, H)
    with torch.no_grad(), O.bf16_operands(OPERAND[1]):
        ref_d = O.discriminator_forward(dict(d_sd), x_emg, True, False, {})
    fold = lambda t: t.permute(0, 3, 1, 2).reshape(t.shape[0] * t.shape[3], t.shape[1], t.shape[2]).contiguous().cuda()
    for i, p in enumerate(O.PERIODS):
        disc = netD.multi_pooled_disc[i]
        Tp = x_bct.shape[-1]
        x4 = F.pad(x_bct, (0, p - Tp % p), "reflect").view(B, 8, -1, p)
        for j, (k, s, pad) in enumerate(O.MPD_SMALL + ((3, 1, 1),)):
            last = j == len(O.MPD_SMALL)
            pre = f"multi_pooled_disc.{i}." + ("output" if last else f"layers.{j}")
            layer = disc.output if last else disc.layers[j]

            def run_o(sd, x, pre=pre, s=s, pad=pad, last=last):
                v, g = sd[pre + ".weight_v"], sd[pre + ".weight_g"]
                y = O._contract(x, O.weight_norm_weight(v, g), sd[pre + ".bias"], v, O._wn_scale(v, g), F.conv2d,
                                stride=(s, 1), padding=(pad, 0))
                return y if last else F.leaky_relu(y, 0.1)

            def run_b(x, layer=layer, p=p):
                y = layer(x)
                return y.view(B, p, y.shape[1], y.shape[2]).permute(0, 2, 3, 1)

            _compare(f"P{p}.{j}", layer, pre, d_sd, run_o, run_b, x4 if j == 0 else ref_d[i][j - 1], worst, to_build=fold)

    # ---- scale discriminators (scale 0: spectral norm, eval mode -- no power iteration on either side)
    xs = [x_bct]
    for _ in range(2):
        xs.append(F.avg_pool1d(xs[-1], 4, 2, 1))
    for i in range(3):
        disc = netD.multi_scale_disc[i]
        h = xs[i]
        for j, (s, pad, g) in enumerate(O.MSD_SMALL + ((1, 1, 1),)):
            last = j == len(O.MSD_SMALL)
            pre = f"multi_scale_disc.{i}." + ("output" if last else f"layers.{j}")
            layer = disc.output if last else disc.layers[j]

            def run_o(sd, x, pre=pre, s=s, pad=pad, g=g, last=last, spectral=(i == 0 and not last)):   # (the output conv is weight-normed)
                if spectral:
                    w, u, v = O.spectral_norm_weight(sd[pre + ".weight_orig"], sd[pre + ".weight_u"], sd[pre + ".weight_v"], False)
                    with torch.no_grad():
                        sc = 1.0 / torch.dot(u, torch.mv(sd[pre + ".weight_orig"].flatten(1), v))
                    y = O._contract(x, w, sd[pre + ".bias"], sd[pre + ".weight_orig"], sc, F.conv1d, uniform=True,
                                    stride=s, padding=pad, groups=g)
                else:
                    y = O._wn_conv1d(sd, pre, x, stride=s, padding=pad, groups=g)
                return y if last else F.leaky_relu(y, 0.1)

            # eval-mode spectral norm divides by u^T W v of the fixture's UNCONVERGED u / v: the feature maps of scale 0 grow by
            # orders of magnitude per layer and leave fp16's range (65504) by layer 3.  Each layer is fed the oracle's map
            # scaled back by a power of two (exact in every format): the layer under test is linear in its input up to the bias
            m = float(h.abs().max())
            if m > 256.0:
                h = h * 2.0 ** -int(np.ceil(np.log2(m / 64.0)))
            _compare(f"S{i}.{j}", layer, pre, d_sd, run_o, lambda x, layer=layer: layer(x), h, worst)
            if not last:
                with torch.no_grad(), O.bf16_operands(OPERAND[1]):
                    h = run_o(d_sd, h)

    # ---- generator: first conv, every block, output conv
    su = det_input("emu.su", (B, T, 256))
    sess = torch.arange(B) % 3
    emb = g_sd["session_embeddings.weight"][sess]
    h = torch.cat((su, emb[:, None, :].expand(-1, T, -1)), -1).transpose(1, 2).contiguous()
    _compare("G.first", netG.gblocks[0], "gblocks.0", g_sd, lambda sd, x: O._wn_conv1d(sd, "gblocks.0", x),
             lambda x: netG.gblocks[0](x), h, worst)
    with torch.no_grad(), O.bf16_operands(OPERAND[1]):
        h = O._wn_conv1d(g_sd, "gblocks.0", h)
    for bi, up in enumerate(O.generator_upsamples("SPEECH_UNITS")):
        pre = f"gblocks.{bi + 1}"
        _compare(f"G.block{bi + 1}", netG.gblocks[bi + 1], pre, g_sd, lambda sd, x, pre=pre, up=up: O.gblock_forward(sd, pre, x, up),
                 lambda x, m=netG.gblocks[bi + 1]: m(x), h, worst)
        with torch.no_grad(), O.bf16_operands(OPERAND[1]):
            h = O.gblock_forward(g_sd, pre, h, up)
    _compare("G.last", netG.last_conv[1], "last_conv.1", g_sd,
             lambda sd, x: torch.tanh(O._wn_conv1d(sd, "last_conv.1", F.relu(x), padding=1).transpose(1, 2)),
             lambda x: netG.last_conv[1](x), h, worst)

    # ---- frozen encoder: blocks, input projection, transformer layers, the two heads
    h = x_bct
    for bi, blk in enumerate(enc.conv_blocks):
        pre = f"conv_blocks.{bi}"
        _compare(f"E.block{bi}", blk, pre, e_sd, lambda sd, x, pre=pre: O.encoder_resblock(sd, pre, x), lambda x, m=blk: m(x),
                 h, worst, with_params=False)
        with torch.no_grad(), O.bf16_operands(OPERAND[1]):
            h = O.encoder_resblock(e_sd, pre, h)
    M = enc.model_size
    lin = lambda w, b, cout, **kw: (lambda x: ops.frozen_conv1d(x, w.unsqueeze(-1), None, b, None, ops.ConvCfg(M, cout, 1, **kw)))
    _compare("E.w_raw_in", None, "", e_sd,
             lambda sd, x: O._frozen_linear(x.transpose(1, 2), sd["w_raw_in.weight"], sd["w_raw_in.bias"]).transpose(1, 2),
             lin(enc.w_raw_in.weight, enc.w_raw_in.bias, M), h, worst, with_params=False)
    with torch.no_grad(), O.bf16_operands(OPERAND[1]):
        h = O._frozen_linear(h.transpose(1, 2), e_sd["w_raw_in.weight"], e_sd["w_raw_in.bias"]).transpose(1, 2).contiguous()
    for li, layer in enumerate(enc.transformer.layers):
        pre = f"transformer.layers.{li}"
        run_o = lambda sd, x, pre=pre: O.transformer_layer(sd, pre, x.permute(2, 0, 1)).permute(1, 2, 0)
        _compare(f"E.layer{li}", layer, pre, e_sd, run_o, lambda x, m=layer: m(x), h, worst, with_params=False)
        with torch.no_grad(), O.bf16_operands(OPERAND[1]):
            h = run_o(e_sd, h).contiguous()
    for head, cout in (("w_out", 256), ("w_aux", 48)):
        _compare("E." + head, None, "", e_sd,
                 lambda sd, x, head=head: O._frozen_linear(x.transpose(1, 2), sd[head + ".weight"], sd[head + ".bias"]),
                 lin(getattr(enc, head).weight, getattr(enc, head).bias, cout, out_btc=True), h, worst, with_params=False)
    worst.sort(reverse=True)
    print("\nlargest layer-wise differences (relative L2):", [(f"{v:.2e}", t, k) for v, t, k in worst[:12]], "of", len(worst))
    bad = [(f"{v:.2e}", t, k) for v, t, k in worst if not v <= LAYER_TOL]
    assert not bad, f"stages differing from the bf16-operand oracle by more than {LAYER_TOL} (relative L2): {bad}"


def _step_errors(st, out, netG, netD, o):
    from test_gpu_full_size import _losses
    got = np.array(_losses(st))
    want = np.array([o["loss_d"], o["loss_g"], o["loss_td"], o["loss_su"], o["loss_ph"], o["loss_fm"]])
    res = {"losses": float((np.abs(got - want) / np.abs(want)).max()),
           "x_pred": float((out["x_pred"].cpu() - o["x_pred"]).abs().max())}
    for tag, net, grads in (("g", netG, o["g_grads"]), ("d", netD, o["d_grads"])):
        ps = dict(net.named_parameters())
        num = sum(float((ps[k].grad.double().cpu() - g.double()).pow(2).sum()) for k, g in grads.items())
        den = sum(float(g.double().pow(2).sum()) for g in grads.values())
        res[tag + "_vector"] = (num / den) ** 0.5
        nb = np.array([float(ps[k].grad.double().norm()) for k in grads])
        no = np.array([float(g.double().norm()) for g in grads.values()])
        res[tag + "_norms"] = float((np.abs(nb - no) / np.maximum(no, 1e-3 * no.max())).max())
    return res


def test_small_step_vs_bf16_operand_oracle(bf16_mode):
    from test_gpu_train_step import build, to_batch
    name = next(iter(cases.STEP_CASES))
    ch, (ms, nl, ff), B, T, steps = cases.STEP_CASES[name]
    g = fill_state_dict(oshapes.generator_shapes("SPEECH_UNITS", ch)); d = fill_state_dict(oshapes.discriminator_shapes(True))
    e = fill_state_dict(oshapes.encoder_shapes(ms, 3, nl, dim_feedforward=ff))
    dt = {k: v for k, v in d.items() if not O.is_sn_buffer(d, k)}
    og, od = O.adamw_init(g), O.adamw_init(dt)
    cfg, netG, netD, optG, optD, step = build(name)
    for s in range(steps):
        b = cases.step_batch(name, s)
        with O.bf16_operands(OPERAND[1]):
            o = O.train_step(g, d, e, og, od, b, keep_grads=True)
        out = step(to_batch(b))
        err = _step_errors(step.read_stats(out), out, netG, netD, o)
        print(f"\nstep {s}:", {k: f"{v:.2e}" for k, v in err.items()})
        assert err["losses"] <= (1e-4 if s == 0 else 1e-3), (s, err)
        assert err["x_pred"] <= (1e-5 if s == 0 else 3e-3), (s, err)
        assert err["g_vector"] <= (1e-3 if s == 0 else 6e-3) and err["d_vector"] <= (1e-4 if s == 0 else 4e-3), (s, err)
        if s == 0:
            assert err["g_norms"] <= 5e-3 and err["d_norms"] <= 1e-4, (s, err)


def test_config3_step_vs_bf16_operand_oracle(bf16_mode):
    """BASELINE config 3 at full size (B = 64, 768 channels, the 768 x 6 encoder), one step."""
    from test_gpu_full_size import _batch, _build, _inputs, _state_dicts
    inp = _inputs("c3", 64, 12, False)
    g, d, e = _state_dicts("SPEECH_UNITS")
    netG, netD, optG, optD, step = _build("SPEECH_UNITS", g, d, e)
    out = step(_batch(inp, False)); st = step.read_stats(out)
    g, d, e = _state_dicts("SPEECH_UNITS")
    dt = {k: v for k, v in d.items() if not O.is_sn_buffer(d, k)}
    with O.bf16_operands(OPERAND[1]):
        o = O.train_step(g, d, e, O.adamw_init(g), O.adamw_init(dt), inp, keep_grads=True)
    err = _step_errors(st, out, netG, netD, o)
    print("\nconfig 3:", {k: f"{v:.2e}" for k, v in err.items()})
    assert err["losses"] <= 1e-3 and err["x_pred"] <= 2e-3, err
    assert err["g_norms"] <= 1.2e-3 and err["d_norms"] <= 1e-4, err
    assert err["g_vector"] <= 3.5e-3 and err["d_vector"] <= 1e-4, err


def test_full_discriminator_step_vs_operand_oracle(bf16_mode):
    """`model.discriminator_small: false` (discriminator.py:158-191: k5 s3 period stacks to 1024 channels, k41 grouped scale
    stacks with strides up to 4, seven spectral-norm layers) through one train step against the oracle with rounded operands:
    which kernel takes a layer (2-byte matrix-core, exact fp32 fallback) must be what the oracle's shape rule says, or the
    losses drift apart at the 2^-9 level."""
    from oracle.detfill import det_ints
    from test_gpu_train_step import make_cfg
    from ste_gan_amd.constants import DataType
    from ste_gan_amd.losses.emg_encoder_loss import EMGEncoderLoss
    from ste_gan_amd.losses.time_domain_loss import MultiTimeDomainFeatureLoss
    from ste_gan_amd.models.discriminator import init_emg_discriminators
    from ste_gan_amd.models.emg_encoder import EMGEncoderTransformer
    from ste_gan_amd.models.generator import init_emg_generator
    from ste_gan_amd.optim import GradScaler
    from ste_gan_amd.train import GANTrainStep, make_optimizers
    B, T = 2, 24
    cfg = make_cfg(32)
    cfg.model.discriminator_small = False
    g_sd = fill_state_dict(oshapes.generator_shapes("SPEECH_UNITS", 32))
    d_sd = fill_state_dict(oshapes.discriminator_shapes(False))
    e_sd = fill_state_dict(oshapes.encoder_shapes(64, 3, 2, dim_feedforward=128))
    netG = init_emg_generator(cfg); netG.load_state_dict(g_sd)
    netD = init_emg_discriminators(cfg); netD.load_state_dict(d_sd)
    enc = EMGEncoderTransformer(8, 256, 48, model_size=64, num_transformer_layers=2, dim_feedforward=128); enc.load_state_dict(e_sd)
    netG, netD, enc = netG.cuda(), netD.cuda(), enc.cuda().eval()
    optG, optD = make_optimizers(netG, netD)
    step = GANTrainStep(cfg, netG, netD, EMGEncoderLoss(enc), MultiTimeDomainFeatureLoss(8), optG, optD)
    scaler = None
    if step.scaler is not None:                                  # fp16 operands: a scale that fits, on both sides
        step.scaler = GradScaler(optG.flat.flat.device, init_scale=2.0 ** 10)
        scaler = {"scale": 2.0 ** 10, "growth_tracker": 0}
    su = det_input("fd.su", (B, T, 256)); x_real = det_input("fd.x", (B, 16 * T, 8), "tanh")
    sess = det_ints("fd.s", (B,), 17); ph = det_ints("fd.ph", (B, T), 48)
    batch = {DataType.REAL_EMG: x_real.cuda(), DataType.SPEECH_UNITS: su.cuda(), DataType.SESSION_INDEX: sess.cuda(),
             DataType.SPEAKING_MODE_INDEX: torch.zeros_like(sess).cuda(), DataType.PHONEMES: ph.cuda()}
    out = step(batch); st = step.read_stats(out)
    d_train = {k: v for k, v in d_sd.items() if not O.is_sn_buffer(d_sd, k)}
    with O.bf16_operands(OPERAND[1]):
        o = O.train_step(g_sd, d_sd, e_sd, O.adamw_init(g_sd), O.adamw_init(d_train),
                         {"su": su, "feats": su, "x_real": x_real, "sess": sess, "phonemes": ph}, small_d=False, keep_grads=True,
                         scaler=scaler)
    err = _step_errors(st, out, netG, netD, o) if scaler is None else None
    got = np.array([st["discriminator"], st["generator"], st["multi_td"], st["speech_unit"], st["phoneme"], st["feature_matching"]])
    want = np.array([o["loss_d"], o["loss_g"], o["loss_td"], o["loss_su"], o["loss_ph"], o["loss_fm"]])
    rel_l = float((np.abs(got - want) / np.abs(want)).max())
    print("\nfull discriminator:", f"losses {rel_l:.2e}", {k: f"{v:.2e}" for k, v in (err or {}).items()})
    assert rel_l <= 1e-3, (got, want)
    if err is not None:                 # (the build consumed nothing here: eager step, gradients still in the arenas)
        assert err["d_vector"] <= 2e-3 and err["g_vector"] <= 2e-2, err
