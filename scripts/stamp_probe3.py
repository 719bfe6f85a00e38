"""Block timeline of the direct conv kernel from in-kernel s_memtime stamps (needs a -DSTEGAN_STAMPS build, see
scripts/micro/): whole-kernel view -- when blocks start, how long prologue / main loop / epilogue take."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, numpy as np
buf = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
os.environ["STEGAN_STAMP_PTR"] = str(buf.data_ptr())
from ste_gan_amd import ops
B = int(os.environ.get("MB_B", "64"))
c = ops.ConvCfg(192, 192, 3, 1, 1, 1, 1, 1, 1)
x = torch.randn(B, 192, 192, device="cuda"); w = torch.randn(192, 192, 3, device="cuda") * 0.05
wf, wb = ops.prep_weights(w, c)
for _ in range(3): ops.conv1d_fwd(x, w, None, None, None, c, wf)
torch.cuda.synchronize(); buf.zero_()
ops.conv1d_fwd(x, w, None, None, None, c, wf)
torch.cuda.synchronize()
st = buf.cpu().numpy().reshape(-1, 8); st = st[st[:, 0] != 0].astype(np.float64)
t0 = st[:, 0].min()
# stamps (non-PROLOGUE build): 0 kernel entry, 1 before first DMA, 2 after first DMA issue, 3/4/6/7 around chunk 3, 5 end
print("blocks", len(st), "clock ticks are 100 MHz s_memtime? -> report raw")
for nm, col in [("entry", 0), ("prologue done", 1), ("first DMA issued", 2), ("x maps done", 4),
                ("loop end", 5), ("block end (stores drained)", 6)]:
    v = st[:, col] - t0
    print(f"{nm:22s} min {v.min():9.0f} median {np.median(v):9.0f} p90 {np.percentile(v, 90):9.0f} max {v.max():9.0f}")
d = st[:, 6] - st[:, 0]
print("block duration median", np.median(d), "max", d.max())
print("prologue median", np.median(st[:, 1] - st[:, 0]), " main loop (2->last stamp before epilogue?)")
