import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, numpy as np
buf = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
os.environ["STEGAN_STAMP_PTR"] = str(buf.data_ptr())
from ste_gan_amd import ops
c = ops.ConvCfg(192, 192, 3, 1, 1, 1, 1, 1, 1)
x = torch.randn(64, 192, 192, device="cuda"); w = torch.randn(192, 192, 3, device="cuda") * 0.05
wf, wb = ops.prep_weights(w, c)
for _ in range(3): ops.conv1d_fwd(x, w, None, None, None, c, wf)
torch.cuda.synchronize(); buf.zero_()
ops.conv1d_fwd(x, w, None, None, None, c, wf)
torch.cuda.synchronize()
st = buf.cpu().numpy().reshape(-1, 8); st = st[st[:, 0] != 0].astype(np.float64)
seg = np.diff(st[:, [0, 3, 4, 5, 6, 1, 2]], axis=1)
for i, nm in enumerate(["start -> taps (kernarg wait, scalar setup)", "x staging setup", "w staging setup", "colbase + acc init", "-> buf setup", "first DMA issue"]):
    print(f"{nm:45s} median {np.median(seg[:, i]):7.0f} p90 {np.percentile(seg[:, i], 90):7.0f}")
