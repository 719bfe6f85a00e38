"""First-layer (Cin = 8) weight gradients: conv_wgrad_small.hip against the generic kernels (STEGAN_WGRAD_SMALL=0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ste_gan_amd import ops
from bf16_microbench import timeit
ops.set_compute_dtype(os.environ.get("MB_DTYPE", "bf16"))
# (name, B', Cout, K, pad, Tin)
for name, Bq, Cout, K, pad, Tin in [("MSD s1 8->128 k15", 128, 128, 15, 7, 192), ("MSD s2", 128, 128, 15, 7, 96), ("MSD s3", 128, 128, 15, 7, 48),
                                    ("MPD p2 8->32 k3", 256, 32, 3, 2, 96), ("MPD p3", 384, 32, 3, 2, 64), ("MPD p5", 640, 32, 3, 2, 39),
                                    ("MPD p7", 896, 32, 3, 2, 28), ("MPD p11", 1408, 32, 3, 2, 18)]:
    c = ops.ConvCfg(8, Cout, K, 1, 1, pad, 1, 1, 0)
    x = torch.randn(Bq, 8, Tin, device="cuda"); dy = torch.randn(Bq, Cout, c.tout(Tin), device="cuda")
    dw = torch.zeros(Cout, 8, K, device="cuda"); db = torch.zeros(Cout, device="cuda")
    t = timeit(lambda: ops.conv1d_bwd_weight(x, dy, c, dw, db))
    print(f"{name:22s} {t:7.1f} us  kernel={ops.LAST_CONV_KERNEL}")
