"""Run one direct-conv layer a few times (profiling target).  usage: one_conv.py Cin Cout K stride dil pad groups Tin B [mode]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ste_gan_amd import ops
Cin, Cout, K, s, d, p, g, Tin, B = [int(a) for a in sys.argv[1:10]]
mode = sys.argv[10] if len(sys.argv) > 10 else "fwd"
c = ops.ConvCfg(Cin, Cout, K, s, d, p, g, 1, 0)
x = torch.randn(B, Cin, Tin, device="cuda"); w = torch.randn(Cout, Cin // g, K, device="cuda") * 0.05
sc = torch.rand(Cout, device="cuda") + 0.5; bias = torch.randn(Cout, device="cuda")
dy = torch.randn(B, Cout, c.tout(Tin), device="cuda")
wf, wb = ops.prep_weights(w, c, None, sc)
for _ in range(5):
    if mode == "fwd": ops.conv1d_fwd(x, w, sc, bias, None, c, wf)
    elif mode == "wgrad": ops.conv1d_bwd_weight(x, dy, c, torch.zeros_like(w), None)
    else: ops.conv1d_bwd_data(dy, w, sc, None, None, c, Tin, wb)
torch.cuda.synchronize()
