"""What would two resident blocks per CU buy?  Two independent conv chains (a generator block's forward convs on stream A, another
block's input-gradient convs on stream B), alone and together, with the shipped library (one 768-thread block per CU) and with
the diagnostic build of scripts/build_tile_pair.sh (STEGAN_HIP_LIB=scripts/micro/libstegan_tile_pair.so):
    python scripts/tile_pair_probe.py ;  STEGAN_HIP_LIB=scripts/micro/libstegan_tile_pair.so python scripts/tile_pair_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ste_gan_amd import ops
from tile_microbench import image

ops.set_compute_dtype("bf16")
B = 64
# (Cin, Cout, K, dil, up, relu, Tin)
CH_A = [(384, 384, 3, 1, 1, 1, 96), (384, 384, 3, 3, 1, 1, 96), (384, 384, 3, 9, 1, 1, 96), (384, 384, 3, 27, 1, 1, 96)]
CH_B = [(192, 192, 3, 1, 1, 1, 192), (192, 192, 3, 3, 1, 1, 192), (192, 192, 3, 9, 1, 1, 192), (192, 192, 3, 27, 1, 1, 192)]
CH_C = [(768, 768, 3, 1, 1, 1, 12), (768, 768, 3, 3, 1, 1, 12), (768, 768, 3, 9, 1, 1, 12), (768, 768, 1, 1, 1, 0, 12)]


def make(chain, mode):
    items = []
    for (Cin, Cout, K, d, up, relu, Tin) in chain:
        c = ops.ConvCfg(Cin, Cout, K, 1, d, d * (K - 1) // 2, 1, up, relu)
        x = torch.randn(B, Cin, Tin, device="cuda"); w = torch.randn(Cout, Cin, K, device="cuda") * 0.05
        sc = torch.rand(Cout, device="cuda") + 0.5; bias = torch.randn(Cout, device="cuda")
        dy = torch.randn(B, Cout, c.tout(Tin), device="cuda")
        ops._attach_image(x, image(x)); ops._attach_image(dy, image(dy))
        wf, wb = ops.prep_weights(w, c, None, sc)
        if mode == "fwd":
            items.append(lambda x=x, w=w, sc=sc, bias=bias, c=c, wf=wf: ops.conv1d_fwd(x, w, sc, bias, None, c, wf))
        else:
            items.append(lambda dy=dy, w=w, sc=sc, x=x, c=c, Tin=Tin, wb=wb, relu=relu: ops.conv1d_bwd_data(dy, w, sc, x if relu else None, None, c, Tin, wb))
    return items


def time_graph(fn, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)


sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def both(a_items, b_items):
    def fn():
        cur = torch.cuda.current_stream()
        sa.wait_stream(cur); sb.wait_stream(cur)
        with torch.cuda.stream(sa):
            for f in a_items:
                f()
        with torch.cuda.stream(sb):
            for f in b_items:
                f()
        cur.wait_stream(sa); cur.wait_stream(sb)
    return fn


print("library:", os.environ.get("STEGAN_HIP_LIB", "shipped"))
for na, ca, ma, nb, cb, mb in (("384ch T96 fwd", CH_A, "fwd", "192ch T192 bwd", CH_B, "bwd"), ("768ch T12 fwd", CH_C, "fwd", "384ch T96 bwd", CH_A, "bwd"),
                               ("768ch T12 fwd", CH_C, "fwd", "768ch T12 bwd", CH_C, "bwd")):
    A, Bc = make(ca, ma), make(cb, mb)
    ta = time_graph(lambda: [f() for f in A]); tb = time_graph(lambda: [f() for f in Bc]); tab = time_graph(both(A, Bc))
    print(f"{na:16s} alone {ta:7.1f} us | {nb:16s} alone {tb:7.1f} us | together {tab:7.1f} us | sum {ta + tb:7.1f} | kernel {ops.LAST_CONV_KERNEL}", flush=True)
