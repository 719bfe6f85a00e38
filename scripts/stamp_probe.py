"""Diagnostic (needs a build with -DSTEGAN_STAMPS): per-block s_memtime stamps of the direct conv kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, numpy as np
buf = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
os.environ["STEGAN_STAMP_PTR"] = str(buf.data_ptr())
from ste_gan_amd import ops
def probe(name, Cin, Cout, K, d, T, B=64):
    c = ops.ConvCfg(Cin, Cout, K, 1, d, d * (K - 1) // 2, 1, 1, 1)
    x = torch.randn(B, Cin, T, device="cuda"); w = torch.randn(Cout, Cin, K, device="cuda") * 0.05
    wf, wb = ops.prep_weights(w, c)
    for _ in range(3): ops.conv1d_fwd(x, w, None, None, None, c, wf)
    torch.cuda.synchronize(); buf.zero_()
    ops.conv1d_fwd(x, w, None, None, None, c, wf)
    torch.cuda.synchronize()
    st = buf.cpu().numpy().reshape(-1, 8)
    st = st[st[:, 0] != 0].astype(np.float64)
    t0 = st[:, 0].min()
    seg = np.diff(st[:, :8], axis=1)
    print(f"{name}: blocks {len(st)}")
    for i, nm in enumerate(["prologue", "first load/store/load", "chunks 0-2", "chunk 3: wait at barrier", "chunk 3: store (vmcnt wait + ds_write)", "chunk 3: issue loads", "chunk 3: MFMA loop"]):
        print(f"   {nm:40s} median {np.median(seg[:, i]):8.0f}  p10 {np.percentile(seg[:, i], 10):8.0f}  p90 {np.percentile(seg[:, i], 90):8.0f}")
probe("G 192->192 k3 d1 T192", 192, 192, 3, 1, 192)
probe("G 768->768 k3 d1 T12", 768, 768, 3, 1, 12)
probe("D 512->1024 k5 T48", 512, 1024, 5, 1, 48)
