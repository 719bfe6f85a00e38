"""Block timeline of the bf16 weight-gradient kernel (image path) from in-kernel stamps; see scripts/stamps2.py.
    bash scripts/build_stamps.sh && STEGAN_HIP_LIB=scripts/micro/libstegan_stamps.so python scripts/stamps_wgrad.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np, torch
buf = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
os.environ["STEGAN_STAMP_PTR"] = str(buf.data_ptr())
from ste_gan_amd import ops
from layer_table_util import make_image, timeit
ops.set_compute_dtype("bf16")
LAYERS = [("D 512->1024 k5 T48", 64, 512, 1024, 5, 1, 1, 2, 1, 1, False, 48),
          ("G 192->192 k3 d9 T192", 64, 192, 192, 3, 1, 9, 9, 1, 1, True, 192),
          ("G 768->768 k3 d1 T12", 64, 768, 768, 3, 1, 1, 1, 1, 1, True, 12),
          ("G 384->384 k3 d3 T96", 64, 384, 384, 3, 1, 3, 3, 1, 1, True, 96),
          ("D 256->512 k37 g16 T96", 64, 256, 512, 37, 2, 1, 18, 16, 1, False, 96)]
for (name, B, Cin, Cout, K, s, d, p, g, up, relu, Tin) in LAYERS:
    c = ops.ConvCfg(Cin, Cout, K, s, d, p, g, up, 1 if relu else 0)
    Tout = c.tout(Tin)
    x = torch.randn(B, Cin, Tin, device="cuda"); dy = torch.randn(B, Cout, Tout, device="cuda")
    ops._attach_image(x, make_image(x)); ops._attach_image(dy, make_image(dy))
    dw = torch.zeros(Cout, Cin // g, K, device="cuda"); db = torch.zeros(Cout, device="cuda")
    run = lambda: ops.conv1d_bwd_weight(x, dy, c, dw, db)
    t = timeit(run)
    torch.cuda.synchronize(); buf.zero_(); run(); torch.cuda.synchronize()
    raw = buf.cpu().numpy()
    st = raw[:1 << 19].reshape(-1, 8); pr = raw[1 << 19:].reshape(-1, 8)
    keep = (st[:, 0] != 0) & (st[:, 5] != 0)
    st, pr = st[keep].astype(np.float64), pr[keep].astype(np.float64)
    seg = lambda a, b: np.median(st[:, b] - st[:, a])
    print(f"{name:24s}: launch+reduce {t:5.1f} us | {len(st):4d} tile blocks | cycles: setup {seg(0, 1):5.0f}  to iter1 {seg(1, 2):5.0f}  "
          f"iter {seg(2, 3) / 4:5.0f}  loop {seg(1, 4):6.0f}  store {seg(4, 5):5.0f}  block {seg(0, 5):6.0f}")
    ok = pr[:, 4] != 0
    if ok.any():
        pm = lambda a, b: np.median((pr[ok][:, b] - (st[ok][:, 0] if a < 0 else pr[ok][:, a])))
        print(f"{'':26s}producer: setup {pm(-1, 0):5.0f}  first issue {pm(0, 1):5.0f} | iteration 4: DMA issue {pm(2, 3):5.0f}  chunk 5 landed after {pm(3, 4):5.0f}")
