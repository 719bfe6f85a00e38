"""Does a bf16 conv launch compute the same thing with other kernels in flight on a second stream?  Captures a graph with
stream A repeating one conv into distinct outputs and stream B running a different workload, replays it, and compares every
A output with the result of the same conv run alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ste_gan_amd import ops

ops.set_compute_dtype("bf16")
torch.manual_seed(0)
dev = "cuda"
R = int(os.environ.get("STRESS_REP", "40"))


def mk(B, Cin, Cout, K, T, s=1, d=1, g=1, relu=False):
    c = ops.ConvCfg(Cin, Cout, K, s, d, (K - 1) // 2 * d, g, 1, 1 if relu else 0)
    x = torch.randn(B, Cin, T, device=dev)
    w = torch.randn(Cout, Cin // g, K, device=dev) * 0.05
    sc = torch.rand(Cout, device=dev) + 0.5
    b = torch.randn(Cout, device=dev)
    wf, wb = ops.prep_weights(w, c, None, sc)
    return c, x, w, sc, b, wf, wb


A = {"enc768": (64, 768, 768, 1, 12), "enc3072": (64, 768, 3072, 1, 12), "g96": (64, 768, 768, 3, 96)}[os.environ.get("STRESS_A", "enc768")]
Bk = {"d1024": (128, 512, 1024, 5, 48), "g96": (64, 768, 768, 3, 96), "enc768": (64, 768, 768, 1, 12)}[os.environ.get("STRESS_B", "d1024")]
ca, xa, wa, sa, ba, wfa, wba = mk(*A)
cb, xb, wb_, sb, bb, wfb, wbb = mk(*Bk)
ref = ops.conv1d_fwd(xa, wa, sa, ba, None, ca, wfa).clone()
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for _ in range(2):                                    # warm-up on the streams (workspaces)
    with torch.cuda.stream(s1):
        ops.conv1d_fwd(xa, wa, sa, ba, None, ca, wfa)
    with torch.cuda.stream(s2):
        ops.conv1d_fwd(xb, wb_, sb, bb, None, cb, wfb)
torch.cuda.synchronize()
outs = []
cap = torch.cuda.Stream()
cap.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(cap):
    with torch.cuda.graph(g, stream=cap):
        s1.wait_stream(cap); s2.wait_stream(cap)
        with torch.cuda.stream(s1):
            for _ in range(R):
                outs.append(ops.conv1d_fwd(xa, wa, sa, ba, None, ca, wfa))
        with torch.cuda.stream(s2):
            for _ in range(R):
                y = ops.conv1d_fwd(xb, wb_, sb, bb, None, cb, wfb)
        cap.wait_stream(s1); cap.wait_stream(s2)
torch.cuda.current_stream().wait_stream(cap)
bad = 0
for rep in range(int(os.environ.get("STRESS_REPLAYS", "50"))):
    g.replay()
    torch.cuda.synchronize()
    for o in outs:
        if not torch.equal(o, ref):
            bad += 1
            if bad <= 2:
                d = (o - ref)
                idx = d.nonzero()
                bs, cs, ts = idx[:, 0].unique(), idx[:, 1].unique(), idx[:, 2].unique()
                print(f"  mismatch: {idx.shape[0]} elements, max |d| {float(d.abs().max()):.3e} (|ref| max {float(ref.abs().max()):.2f}), "
                      f"nan {int(torch.isnan(o).sum())}; b {bs.min().item()}..{bs.max().item()} ({bs.numel()}), "
                      f"c {cs.min().item()}..{cs.max().item()} ({cs.numel()}), t {ts.min().item()}..{ts.max().item()} ({ts.numel()})")
print(f"A={A} B={Bk}: {bad} of {len(outs) * int(os.environ.get('STRESS_REPLAYS', '50'))} concurrent results differ from the isolated one"
      f" (kernel {ops.LAST_CONV_KERNEL})")
