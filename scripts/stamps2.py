"""Block timeline of the bf16 conv kernel (image path) from in-kernel s_memtime stamps of consumer wave 0, beside the
graph-timed duration of a launch: where the microseconds of a small-N launch go.
    bash scripts/build_stamps.sh && STEGAN_HIP_LIB=scripts/micro/libstegan_stamps.so python scripts/stamps2.py [substring]
Stamps: 0 entry, 2 consumer index setup done, 3 top of iteration 1, 4 top of iteration 5, 5 loop done, 6 epilogue issued,
7 stores drained."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np, torch
buf = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
os.environ["STEGAN_STAMP_PTR"] = str(buf.data_ptr())
from ste_gan_amd import ops
from plan_sweep import LAYERS, B
from layer_table_util import make_image, timeit
only = sys.argv[1] if len(sys.argv) > 1 else ""
ops.set_compute_dtype("bf16")
for (name, Cin, Cout, K, s, d, p, g, up, relu, Tin) in LAYERS:
    if only not in name: continue
    c = ops.ConvCfg(Cin, Cout, K, s, d, p, g, up, 1 if relu else 0)
    Tout = c.tout(Tin)
    x = torch.randn(B, Cin, Tin, device="cuda"); w = torch.randn(Cout, Cin // g, K, device="cuda") * 0.05
    sc = torch.rand(Cout, device="cuda") + 0.5; bias = torch.randn(Cout, device="cuda"); dy = torch.randn(B, Cout, Tout, device="cuda")
    ops._attach_image(x, make_image(x)); ops._attach_image(dy, make_image(dy))
    wf, wb = ops.prep_weights(w, c, None, sc)
    for mode in ("fwd", "bwd"):
        run = (lambda: ops.conv1d_fwd(x, w, sc, bias, None, c, wf)) if mode == "fwd" else \
              (lambda: ops.conv1d_bwd_data(dy, w, sc, x if relu else None, None, c, Tin, wb))
        t_launch = timeit(run)
        torch.cuda.synchronize(); buf.zero_()
        run(); torch.cuda.synchronize()
        raw = buf.cpu().numpy()
        st = raw[:1 << 19].reshape(-1, 8); pr = raw[1 << 19:(1 << 19) + (1 << 18)].reshape(-1, 8)
        qr = raw[(1 << 19) + (1 << 18):].reshape(-1, 8)
        keep = st[:, 0] != 0
        nq = len(qr); st, pr, qr = st[:nq][keep[:nq]].astype(np.float64), pr[:nq][keep[:nq]].astype(np.float64), qr[keep[:nq]].astype(np.float64)
        seg = lambda a, b: np.median((st[:, b] - st[:, a])[(st[:, a] != 0) & (st[:, b] != 0)]) if ((st[:, a] != 0) & (st[:, b] != 0)).any() else float("nan")
        print(f"{name:24s} {mode}: launch {t_launch:5.1f} us | {len(st):4d} blocks | cycles: setup {seg(0, 2):5.0f}  first chunk+iter0 {seg(2, 3):5.0f}  "
              f"iter {seg(3, 4) / 4:5.0f}  loop {seg(2, 5):6.0f}  epilogue {seg(5, 6):5.0f}  drain {seg(6, 7):5.0f}  block {seg(0, 7):6.0f}", flush=True)
        pm = lambda a, b: np.median((pr[:, b] - (st[:, 0] if a < 0 else pr[:, a]))[(pr[:, b] != 0)]) if (pr[:, b] != 0).any() else float("nan")
        qd = lambda a, b: np.median((b - a)[(a != 0) & (b != 0)]) if ((a != 0) & (b != 0)).any() else float("nan")
        print(f"{'':30s} producer wave 0, entry to first DMA: kernel arguments {qd(pr[:, 6], qr[:, 0]):5.0f}  tile decode {qd(qr[:, 0], qr[:, 1]):5.0f}  "
              f"phase table + common setup {qd(qr[:, 1], qr[:, 2]):5.0f}  W offsets + descriptor {qd(qr[:, 2], qr[:, 3]):5.0f}  W DMA issue {qd(qr[:, 3], pr[:, 7]):5.0f}")
        print(f"{'':30s} producer wave 0: enters +{pm(-1, 6):5.0f}  decode + W offsets + first W DMA {pm(6, 7):5.0f}  X offsets {pm(7, 0):5.0f} | setup {pm(-1, 0):5.0f}  prologue issue {pm(0, 1):5.0f}  chunk 0 lands +{pm(1, 2):5.0f}  | iteration 4: "
              f"DMA issue {pm(3, 4):5.0f}  then chunk 5 landed after {pm(4, 5):5.0f}", flush=True)
