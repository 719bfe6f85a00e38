"""conv_tile_wgrad_kernel against the fp64 contraction of rounded operands, and its time against conv_wgrad_bf16.hip:
    python scripts/tile_wgrad_check.py B Cin Cout K dil up relu Tin [time]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from ste_gan_amd import ops
from tile_microbench import image
from bf16_microbench import timeit

B, Cin, Cout, K, d, up, relu, Tin = [int(a) for a in sys.argv[1:9]]
pad = d * (K - 1) // 2
torch.manual_seed(0)
x = torch.randn(B, Cin, Tin); dy = torch.randn(B, Cout, Tin * up)
r = lambda t: t.to(torch.bfloat16).double()
hx = F.relu(x) if relu else x
if up == 2:
    hx = F.interpolate(hx, scale_factor=2.0, mode="nearest")
w0 = torch.zeros(Cout, Cin, K, dtype=torch.float64, requires_grad=True)
y = F.conv1d(r(hx), w0, None, dilation=d, padding=pad)
(dw_ref,) = torch.autograd.grad(y, w0, r(dy))
db_ref = dy.double().sum(dim=(0, 2))
ops.set_compute_dtype("bf16")
c = ops.ConvCfg(Cin, Cout, K, 1, d, pad, 1, up, 1 if relu else 0)
xg, dyg = x.cuda(), dy.cuda()
ops._attach_image(xg, image(xg)); ops._attach_image(dyg, image(dyg))
init = torch.randn(Cout, Cin, K)
dw = init.clone().cuda(); db = torch.ones(Cout, device="cuda")
ops.conv1d_bwd_weight(xg, dyg, c, dw, db)
got = dw.double().cpu() - init.double()
err = (got - dw_ref).abs()
sc = float(dw_ref.abs().max())
bad = err > 1e-3 * sc
print("db got", (db.cpu() - 1.0)[:3].tolist(), "ref", db_ref[:3].tolist())
print("kernel", ops.LAST_CONV_KERNEL, "dw max err", float(err.max()), "scale", sc, "bad", int(bad.sum()), "of", bad.numel(),
      "| db err", float((db.double().cpu() - 1.0 - db_ref).abs().max()))
if bad.any():
    idx = bad.nonzero()
    print("  bad co:", sorted(set(idx[:, 0].tolist()))[:48]); print("  bad ci:", sorted(set(idx[:, 1].tolist()))[:48])
    print("  bad k:", sorted(set(idx[:, 2].tolist())))
    for i in idx[:6].tolist():
        print("  ", i, float(got[tuple(i)]), float(dw_ref[tuple(i)]))
if len(sys.argv) > 9:
    t = timeit(lambda: ops.conv1d_bwd_weight(xg, dyg, c, dw, db))
    gf = 2.0 * B * Tin * up * Cout * Cin * K / 1e9
    print(f"  time {t:.1f} us  {gf / t * 1e3:.0f} TF")
