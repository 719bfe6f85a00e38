"""-m gpu: generator forward + backward on MI355X (through libstegan_hip.so) against the CPU oracle
on the same deterministic weights / inputs, and against the golden fixtures produced by the
reference's own modules.  Tolerance: north_star's 1e-4 relative (fp32) per EMG sample, realised
as |a-b| <= 1e-4*|b| + 1e-5 (EMG samples live in (-1,1))."""
import numpy as np
import pytest
import torch

import cases
from oracle import shapes as oshapes
from oracle import stegan_oracle as O
from oracle.detfill import det_input, fill_state_dict

pytestmark = pytest.mark.gpu


def build(ft, ch):
    from ste_gan_amd.models.generator import EMGGeneratorGanTTS
    g = EMGGeneratorGanTTS(ft, 256 if ft == "SPEECH_UNITS" else 25, 17, 8, channels=ch)
    sd = fill_state_dict(oshapes.generator_shapes(ft, ch))
    g.load_state_dict(sd)
    return g.cuda(), sd


@pytest.mark.parametrize("name", list(cases.GEN_CASES))
def test_generator_vs_golden_and_oracle(name, golden):
    ft, ch, B, T = cases.GEN_CASES[name]
    g, sd = build(ft, ch)
    feats, sess, spk = cases.gen_inputs(name)
    out = g(feats.cuda(), sess.cuda(), spk.cuda())
    gold = golden(name)
    torch.testing.assert_close(out.detach().cpu(), torch.from_numpy(gold["out"]), rtol=1e-4, atol=1e-5)
    ref = O.generator_forward(sd, feats, sess, spk, ft)
    torch.testing.assert_close(out.detach().cpu(), ref, rtol=1e-4, atol=1e-5)
    # backward: gradient norms / leading entries of all 127 parameter tensors
    probe = det_input(name + ".probe", out.shape).cuda()
    (out * probe).sum().backward()
    gn = np.array([float(p.grad.double().norm()) for p in g.parameters()])
    np.testing.assert_allclose(gn, gold["grad_norm"], rtol=2e-4, atol=1e-6 * gold["grad_norm"].max())
    gh = np.stack([np.pad(p.grad.flatten()[:4].double().cpu().numpy(), (0, max(0, 4 - p.numel()))) for p in g.parameters()])
    np.testing.assert_allclose(gh, gold["grad_head"], rtol=1e-3, atol=2e-4 * np.abs(gold["grad_head"]).max())


def test_generate_b64_matches_oracle():
    """BASELINE config 2 shape: B=64, T_su=12 -> (64,192,8); full-width generator."""
    g, sd = build("SPEECH_UNITS", 768)
    feats = det_input("b64.feats", (64, 12, 256))
    sess = torch.arange(64) % 17
    out = g.generate(feats.cuda(), sess.cuda(), torch.zeros(64, dtype=torch.long).cuda())
    assert out.shape == (64, 192, 8)
    ref = O.generator_forward(sd, feats[:4], sess[:4], None)
    torch.testing.assert_close(out[:4].cpu(), ref, rtol=1e-4, atol=1e-5)
    # batch independence: sample 37 alone equals sample 37 in the batch
    solo = g.generate(feats[37:38].cuda(), sess[37:38].cuda(), torch.zeros(1, dtype=torch.long).cuda())
    torch.testing.assert_close(solo[0], out[37], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("T_su,ch", [(1, 768), (7, 768), (333, 96)])
def test_whole_utterance_generation_at_batch_one(T_su, ch):
    """Validation / sampling path (train.py:380-414, generator.py:48-75): one whole utterance of arbitrary length at
    batch 1 -- a single frame, an odd length, and a long one (5 328 samples) -- through ``generate`` and
    ``generate_from_data_dict`` (per-utterance dict with un-batched tensors, result (T_emg, C) on the CPU)."""
    g, sd = build("SPEECH_UNITS", ch)
    feats = det_input(f"utt{T_su}.feats", (1, T_su, 256))
    sess = torch.tensor([5])
    out = g.generate(feats.cuda(), sess.cuda(), torch.zeros(1, dtype=torch.long).cuda())
    assert out.shape == (1, 16 * T_su, 8) and not out.requires_grad
    ref = O.generator_forward(sd, feats, sess, None)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=1e-5)
    item = {"SPEECH_UNITS": feats[0], "SESSION_INDEX": torch.tensor(5), "SPEAKING_MODE_IDX": torch.tensor(0)}
    solo = g.generate_from_data_dict(item, torch.device("cuda"))
    assert solo.device.type == "cpu" and solo.shape == (16 * T_su, 8)
    torch.testing.assert_close(solo, out[0].cpu(), rtol=0, atol=0)


@pytest.mark.parametrize("mode", ["bf16", "f16"])
@pytest.mark.parametrize("T_su,ch", [(1, 768), (7, 768), (333, 96)])
def test_whole_utterance_generation_at_batch_one_in_the_two_byte_modes(T_su, ch, mode):
    """The reference validates and samples under fp16 autocast (train.py:326,387-395): the same whole-utterance path with
    2-byte matrix-core operands, against the oracle run on operands rounded at the same points (``O.bf16_operands``: the
    kernels' rounding, fp32 everywhere else).  Every launch of these shapes -- T = 1 ... 5 328 at batch one -- takes the
    kernels' ragged / single-sample plans (or declines to the fp32 staging path), which the B = 64 tests never reach."""
    from ste_gan_amd import ops
    dt = torch.bfloat16 if mode == "bf16" else torch.float16
    g, sd = build("SPEECH_UNITS", ch)
    feats = det_input(f"utt{T_su}.feats", (1, T_su, 256))
    sess = torch.tensor([5])
    ops.set_compute_dtype(mode)
    try:
        out = g.generate(feats.cuda(), sess.cuda(), torch.zeros(1, dtype=torch.long).cuda())
        item = {"SPEECH_UNITS": feats[0], "SESSION_INDEX": torch.tensor(5), "SPEAKING_MODE_IDX": torch.tensor(0)}
        solo = g.generate_from_data_dict(item, torch.device("cuda"))
    finally:
        ops.set_compute_dtype("f32")
    assert out.shape == (1, 16 * T_su, 8) and not out.requires_grad
    with torch.no_grad(), O.bf16_operands(dt):
        ref = O.generator_forward(sd, feats, sess, None)
    # the emulation bound of tests/test_gpu_bf16_emulation.py for a whole network: 1e-3 of the output's scale (tanh: <= 1)
    assert float((out.cpu() - ref).abs().max()) <= 1e-3, float((out.cpu() - ref).abs().max())
    torch.testing.assert_close(solo, out[0].cpu(), rtol=0, atol=0)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("up", [1, 2])
def test_gblock_gradient_handover_equals_autograd_accumulation(up, dtype, monkeypatch):
    """GBlock backward with the two fan-out sums formed inside the input-gradient kernels (ops.GradHandover) against the
    same block with autograd adding the two gradients of x and of y: same operands, same fp32 additions."""
    from ste_gan_amd import ops
    from ste_gan_amd.layers.conv import GBlock
    torch.manual_seed(3)
    blk = GBlock(48, 32 if up == 2 else 48, upsample=up).cuda()
    x0 = torch.randn(5, 48, 24, device="cuda")
    dy = torch.randn(5, 32 if up == 2 else 48, 24 * up, device="cuda")
    ops.set_compute_dtype(dtype)
    try:
        res = {}
        for on in (True, False):
            monkeypatch.setattr(ops, "HANDOVER", on)
            x = x0.clone().requires_grad_(True)
            blk.zero_grad(set_to_none=True)
            y = blk(x)
            y.backward(dy)
            res[on] = [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in blk.parameters()]
    finally:
        ops.set_compute_dtype("f32")
    for a, b in zip(res[True], res[False]):
        assert torch.isfinite(a).all()
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-9
    # second backward through a retained graph: the hand-over objects are spent, the parkers return their gradients
    monkeypatch.setattr(ops, "HANDOVER", True)
    x = x0.clone().requires_grad_(True)
    y = blk(x)
    (g1,) = torch.autograd.grad(y, x, dy, retain_graph=True)
    (g2,) = torch.autograd.grad(y, x, dy)
    assert float((g1 - g2).abs().max()) <= 2e-6 * float(g1.abs().max())


@pytest.mark.parametrize("two", [False, True])
def test_embed_concat_kernel_vs_torch(two):
    """ops.embed_concat (generator.py:143-154: embedding gather, broadcast over time, concat, 'b t c -> b c t') and its
    backward against the torch ops it replaces, ragged sizes, repeated and out-of-batch session ids."""
    from ste_gan_amd import ops
    torch.manual_seed(5)
    B, T, F_, E, S = 7, 37, 25, 64, 17
    x = torch.randn(B, T, F_, device="cuda")
    w1 = torch.randn(S, E, device="cuda", requires_grad=True)
    w2 = torch.randn(3, E, device="cuda", requires_grad=True) if two else None
    id1 = torch.tensor([3, 3, 0, 16, 3, 9, 0], device="cuda")
    id2 = torch.tensor([0, 2, 2, 1, 0, 0, 2], device="cuda") if two else None
    out = ops.embed_concat(x, id1, w1, id2, w2)
    ref = torch.cat([x, w1[id1][:, None, :].expand(-1, T, -1)] + ([w2[id2][:, None, :].expand(-1, T, -1)] if two else []), -1)
    ref = ref.transpose(1, 2).contiguous()
    assert torch.equal(out, ref)
    g = torch.randn_like(out)
    out.backward(g)
    gw = [w1.grad.clone()] + ([w2.grad.clone()] if two else [])
    w1.grad = None
    if two:
        w2.grad = None
    ref.backward(g)
    torch.testing.assert_close(gw[0], w1.grad, rtol=1e-5, atol=1e-5)
    if two:
        torch.testing.assert_close(gw[1], w2.grad, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("n,shape", [(2, (4, 8, 96)), (3, (2, 192, 8)), (4, (3, 16, 13)), (6, (2, 5, 7))], ids=str)
def test_fan_out_sums_consumer_gradients_in_one_launch(n, shape):
    """ops.fan_out: n aliases of a tensor, the consumers' gradients summed left to right by stegan_sum_n (one launch per
    four consumers) instead of the autograd engine's ATen add per extra consumer: same bits as the explicit sum, aliases
    share the producer's bytes (and its bf16 image), and nothing is made when no gradient flows."""
    from ste_gan_amd import ops
    torch.manual_seed(n)
    x = torch.randn(*shape, device="cuda", requires_grad=True)
    outs = ops.fan_out(x, n)
    assert len(outs) == n and all(o.data_ptr() == x.data_ptr() and o is not x for o in outs)
    gs = [torch.randn(*shape, device="cuda") for _ in range(n)]
    torch.autograd.backward(list(outs), gs)
    want = gs[0].clone()
    for g in gs[1:]:
        want += g
    assert torch.equal(x.grad, want)
    # a consumer that never produced a gradient is skipped; a single live one is passed through
    x.grad = None
    outs = ops.fan_out(x, 3)
    outs[1].mul(2.0).sum().backward()
    assert torch.equal(x.grad, torch.full_like(x, 2.0))
    with torch.no_grad():
        assert ops.fan_out(x, 2) == (x, x)
    assert ops.fan_out(x.detach(), 2)[0] is not None and ops.fan_out(x.detach(), 2)[0].requires_grad is False


def test_out_of_range_embedding_ids_raise_like_nn_embedding():
    """nn.Embedding raises IndexError for an id outside its table (generator.py:143-151 call it); the fused gather kernel would
    map such an id to a zero row, so the eager entry checks the ids on the host."""
    g, _ = build("SPEECH_UNITS", 96)
    feats = det_input("ids.feats", (2, 3, 256)).cuda()
    zero = torch.zeros(2, dtype=torch.long).cuda()
    n = g.session_embeddings.num_embeddings
    g(feats, torch.tensor([0, n - 1]).cuda(), zero)                       # the last row is fine
    with pytest.raises(IndexError):
        g(feats, torch.tensor([0, n]).cuda(), zero)
    with pytest.raises(IndexError):
        g(feats, torch.tensor([-1, 0]).cuda(), zero)
