"""Where does conv_tile_kernel differ from the fp64 contraction of rounded operands?  (debugging aid)
    python scripts/tile_check.py B Cin Cout K dil up relu Tin"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from ste_gan_amd import ops
from tile_microbench import image

B, Cin, Cout, K, d, up, relu, Tin = [int(a) for a in sys.argv[1:9]]
pad = d * (K - 1) // 2
torch.manual_seed(0)
x = torch.randn(B, Cin, Tin); w = torch.randn(Cout, Cin, K) / (Cin * K) ** 0.5
scale = 0.5 + torch.rand(Cout); bias = torch.randn(Cout)
Tout = Tin * up
res = torch.randn(B, Cout, Tout); dy = torch.randn(B, Cout, Tout); add = torch.randn(B, Cin, Tin * up)
r = lambda t: t.to(torch.bfloat16).double()
hx = F.relu(x) if relu else x
if up == 2:
    hx = F.interpolate(hx, scale_factor=2.0, mode="nearest")
y_ref = F.conv1d(r(hx), r(w), None, dilation=d, padding=pad) * scale.double().view(1, -1, 1) + bias.double().view(1, -1, 1) + res.double()
ws_ = w * scale.view(-1, 1, 1)
xe = torch.zeros(B, Cin, Tin * up, dtype=torch.float64, requires_grad=True)
ye = F.conv1d(xe, r(ws_), None, dilation=d, padding=pad)
(gx,) = torch.autograd.grad(ye, xe, r(dy))
mask = (x > 0).double()
if up == 2:
    mask = F.interpolate(mask, scale_factor=2.0, mode="nearest")
dx_ref = gx * (mask if relu else 1.0) + add.double()

ops.set_compute_dtype("bf16")
c = ops.ConvCfg(Cin, Cout, K, 1, d, pad, 1, up, 1 if relu else 0)
xg, wg, sg, bg, rg, dyg, addg = (t.cuda() for t in (x, w, scale, bias, res, dy, add))
ops._attach_image(xg, image(xg)); ops._attach_image(dyg, image(dyg))
wf, wb = ops.prep_weights(wg, c, None, sg)
y = ops.conv1d_fwd(xg, wg, sg, bg, rg, c, wf)
kf = ops.LAST_CONV_KERNEL
dxe = ops.conv1d_bwd_data(dyg, wg, sg, xg if relu else None, addg, c, Tin, wb)


def report(name, got, ref, img):
    got = got.double().cpu()
    err = (got - ref).abs()
    bad = err > 1e-3 * float(ref.abs().max())
    print(name, "max err", float(err.max()), "bad", int(bad.sum()), "of", bad.numel(), end=" | ")
    if img is not None:
        print("image ok:", bool(torch.equal(img.view(torch.int16), image(got.float().cuda()).view(torch.int16))))
    else:
        print("no image")
    if bad.any():
        idx = bad.nonzero()
        print("  bad b:", sorted(set(idx[:, 0].tolist()))[:20])
        print("  bad ch:", sorted(set(idx[:, 1].tolist()))[:64])
        print("  bad t:", sorted(set(idx[:, 2].tolist()))[:64])
        for i in idx[:6].tolist():
            print("  ", i, float(got[tuple(i)]), float(ref[tuple(i)]))


report("fwd", y, y_ref, getattr(y, "h", None))
report("bwd", dxe, dx_ref, getattr(dxe, "h", None))
