"""Block timeline of the bf16 direct conv kernel from in-kernel s_memtime stamps.

Needs a diagnostic build of the library with -DSTEGAN_STAMPS (scripts/build_stamps.sh writes scripts/micro/libstegan_stamps.so):
    STEGAN_HIP_LIB=scripts/micro/libstegan_stamps.so python scripts/bf16_stamps.py [layer substring]
Stamps per block: 0 entry, 1 index setup done, 2 first chunk staged, 3 top of iteration 1, 4 top of iteration 5,
5 main loop done, 6 epilogue stores issued, 7 stores drained."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
buf = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
os.environ["STEGAN_STAMP_PTR"] = str(buf.data_ptr())
from ste_gan_amd import ops
from bf16_microbench import LAYERS
only = sys.argv[1] if len(sys.argv) > 1 else "G 768->768 k3 d1 T96"
mode = sys.argv[2] if len(sys.argv) > 2 else "fwd"
ops.set_compute_dtype("bf16")
for (name, Cin, Cout, K, s, d, p, g, up, relu, Tin, Be) in LAYERS:
    if only not in name: continue
    c = ops.ConvCfg(Cin, Cout, K, s, d, p, g, up, 1 if relu else 0)
    Tout = c.tout(Tin)
    x = torch.randn(Be, Cin, Tin, device="cuda"); w = torch.randn(Cout, Cin // g, K, device="cuda") * 0.05
    sc = torch.rand(Cout, device="cuda") + 0.5; bias = torch.randn(Cout, device="cuda"); dy = torch.randn(Be, Cout, Tout, device="cuda")
    wf, wb = ops.prep_weights(w, c, None, sc)
    run = (lambda: ops.conv1d_fwd(x, w, sc, bias, None, c, wf)) if mode == "fwd" else \
          (lambda: ops.conv1d_bwd_data(dy, w, sc, x if relu else None, None, c, Tin, wb))
    for _ in range(3): run()
    torch.cuda.synchronize(); buf.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record()
    torch.cuda.synchronize()
    st = buf.cpu().numpy().reshape(-1, 8); st = st[st[:, 0] != 0].astype(np.float64)
    t0 = st[:, 0].min()
    tick = 1.0             # s_memtime ticks are shader cycles (MI355X_MICROARCH.md); counters of different XCDs are not aligned
    print(f"{name} [{mode}]: {len(st)} blocks, event time {e0.elapsed_time(e1) * 1e3:.1f} us (incl. launch)")
    if os.environ.get("STAMPS_ITER"):          # build with STAMP_FLAGS=-DSTEGAN_STAMPS_ITER: tops of iterations 1, 5, ..., 21
        m = lambda a, b: np.median(st[:, b] - st[:, a]) / 4
        print("  cycles per iteration (median over blocks), iterations 1-5, 5-9, 9-13, 13-17, 17-21: " +
              "  ".join(f"{m(i, i + 1):.0f}" for i in range(1, 6)))
        continue
    seg = lambda a, b: np.median(st[:, b] - st[:, a]) * tick
    p90 = lambda a, b: np.percentile(st[:, b] - st[:, a], 90) * tick
    print(f"  per block, cycles (median | p90): setup {seg(0, 1):.0f} | {p90(0, 1):.0f}   first chunk {seg(1, 2):.0f} | {p90(1, 2):.0f}   "
          f"iter0 {seg(2, 3):.0f}   iters1-4 {seg(3, 4) / 4:.0f} each   loop total {seg(2, 5):.0f} | {p90(2, 5):.0f}   "
          f"epilogue issue {seg(5, 6):.0f} | {p90(5, 6):.0f}   drain {seg(6, 7):.0f}   block {seg(0, 7):.0f} | {p90(0, 7):.0f}")
