"""-m gpu: kernels must compute the same thing whether or not another stream is busy.

Round 2 found the bf16 conv kernel returning wrong tiles in 2-10 % of its launches whenever a second HIP stream had work
in flight (never alone, so every single-stream parity test passed): hipcc lowers ``__syncthreads()`` to
``s_waitcnt vmcnt(0); s_barrier`` without waiting for the wave's outstanding LDS stores, and with another kernel's waves
on the CU a consumer wave's first ``ds_read`` after the barrier could overtake the producer wave's last ``ds_write``.  All
kernels now use ``block_sync()`` (csrc/common.h: explicit ``s_waitcnt lgkmcnt(0)``).  This test replays a captured graph with
one stream repeating a conv launch (forward, input gradient, weight gradient) into distinct outputs while a second stream
runs a different conv, and requires every result to equal the one computed in isolation, bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# (B, Cin, Cout, K, T): encoder linear layer, generator layer (1.125 rounds of blocks), discriminator layer
SHAPES = {"enc": (64, 768, 768, 1, 12), "gen": (64, 768, 768, 3, 96), "disc": (128, 512, 1024, 5, 48)}


def _make(ops, B, Cin, Cout, K, T):
    c = ops.ConvCfg(Cin, Cout, K, 1, 1, (K - 1) // 2)
    g = torch.Generator().manual_seed(B + Cin + K + T)
    x = torch.randn(B, Cin, T, generator=g).cuda()
    w = (torch.randn(Cout, Cin, K, generator=g) * 0.05).cuda()
    sc = (torch.rand(Cout, generator=g) + 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    dy = torch.randn(B, Cout, T, generator=g).cuda()
    wf, wb = ops.prep_weights(w, c, None, sc)
    return dict(c=c, x=x, w=w, sc=sc, b=b, dy=dy, wf=wf, wb=wb, T=T)


def _run(ops, m, what):
    if what == "fwd":
        return ops.conv1d_fwd(m["x"], m["w"], m["sc"], m["b"], None, m["c"], m["wf"])
    if what == "bwd_data":
        return ops.conv1d_bwd_data(m["dy"], m["w"], m["sc"], None, None, m["c"], m["T"], m["wb"])
    dw = torch.zeros_like(m["w"])
    ops.conv1d_bwd_weight(m["x"], m["dy"], m["c"], dw, None)
    return dw


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("what", ["fwd", "bwd_data", "bwd_weight"])
@pytest.mark.parametrize("a,b", [("gen", "disc"), ("enc", "gen")])
def test_conv_results_do_not_depend_on_a_busy_second_stream(a, b, what, dtype):
    from ste_gan_amd import ops
    ops.set_compute_dtype(dtype)
    try:
        ma, mb = _make(ops, *SHAPES[a]), _make(ops, *SHAPES[b])
        reps, replays = (12, 10) if dtype == "bf16" else (4, 4)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        for _ in range(2):                                   # warm-up on the streams: workspaces, lazy attributes
            with torch.cuda.stream(s1):
                _run(ops, ma, what)
            with torch.cuda.stream(s2):
                _run(ops, mb, "fwd")
        torch.cuda.synchronize()
        ref = _run(ops, ma, what).clone()
        torch.cuda.synchronize()
        outs = []
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(cap):
            with torch.cuda.graph(graph, stream=cap):
                s1.wait_stream(cap); s2.wait_stream(cap)
                with torch.cuda.stream(s1):
                    for _ in range(reps):
                        outs.append(_run(ops, ma, what))
                with torch.cuda.stream(s2):
                    for _ in range(reps):
                        _run(ops, mb, "fwd")
                cap.wait_stream(s1); cap.wait_stream(s2)
        torch.cuda.current_stream().wait_stream(cap)
        bad = 0
        for _ in range(replays):
            graph.replay()
            torch.cuda.synchronize()
            for o in outs:
                if what == "bwd_weight" and dtype == "f32":      # fp32 slices meet in atomics: last-bit noise, not equality
                    bad += int(float((o - ref).abs().max()) > 1e-4 * float(ref.abs().max()))
                else:
                    bad += int(not torch.equal(o, ref))
        assert bad == 0, f"{bad} of {reps * replays} concurrent {what} results differ from the isolated one"
    finally:
        ops.set_compute_dtype("f32")


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_encoder_and_generator_forward_do_not_depend_on_a_busy_second_stream(dtype):
    """Module level: the frozen EMG encoder (convs, LayerNorm, relative-position attention) and the generator, replayed on one
    stream while a discriminator-sized conv keeps a second stream busy, must return the bits they return alone."""
    from oracle import shapes as oshapes
    from oracle.detfill import det_input, det_ints, fill_state_dict
    from ste_gan_amd import ops
    from ste_gan_amd.models.emg_encoder import EMGEncoderTransformer
    from ste_gan_amd.models.generator import EMGGeneratorGanTTS
    ops.set_compute_dtype(dtype)
    try:
        enc = EMGEncoderTransformer(8, 256, 48, model_size=64, num_transformer_layers=2, dim_feedforward=128)
        enc.load_state_dict(fill_state_dict(oshapes.encoder_shapes(64, 3, 2, dim_feedforward=128)))
        enc = enc.cuda().eval()
        gen = EMGGeneratorGanTTS("SPEECH_UNITS", 256, 17, 8, channels=48)
        gen.load_state_dict(fill_state_dict(oshapes.generator_shapes("SPEECH_UNITS", 48)))
        gen = gen.cuda().eval()
        x = det_input("cc.x", (16, 192, 8), "tanh").cuda()
        su = det_input("cc.su", (16, 12, 256)).cuda()
        sess = det_ints("cc.s", (16,), 17).cuda()
        mb = _make(ops, *SHAPES["disc"])

        def work():
            with torch.no_grad():
                return list(enc(x)) + [gen(su, sess, torch.zeros_like(sess))]

        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        for _ in range(2):
            with torch.cuda.stream(s1):
                work()
            with torch.cuda.stream(s2):
                _run(ops, mb, "fwd")
        torch.cuda.synchronize()
        ref = [t.clone() for t in work()]
        torch.cuda.synchronize()
        outs = []
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(cap):
            with torch.cuda.graph(graph, stream=cap):
                s1.wait_stream(cap); s2.wait_stream(cap)
                with torch.cuda.stream(s1):
                    for _ in range(4):
                        outs.append(work())
                with torch.cuda.stream(s2):
                    for _ in range(40):
                        _run(ops, mb, "fwd")
                cap.wait_stream(s1); cap.wait_stream(s2)
        torch.cuda.current_stream().wait_stream(cap)
        bad = 0
        for _ in range(8):
            graph.replay()
            torch.cuda.synchronize()
            for o in outs:
                bad += sum(int(not torch.equal(a, b)) for a, b in zip(o, ref))
        assert bad == 0, f"{bad} module outputs differ from the isolated ones"
    finally:
        ops.set_compute_dtype("f32")
