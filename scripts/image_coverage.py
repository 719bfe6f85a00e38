"""Which conv launches of one train step get their source as a bf16 activation image (LDS-DMA staging) and which still
convert an fp32 tensor -- i.e. which producers do not write an image yet.  python scripts/image_coverage.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ste_gan_amd import ops

ops.set_compute_dtype("bf16")
dev = torch.device("cuda", 0)
step = bench.build_step(bench.make_cfg("train_step"), dev)
batch = bench.synthetic_batch(dev, seed=1000)
step(batch); step(batch)
torch.cuda.synchronize()
ops.IMG_STATS = {}
step(batch)
torch.cuda.synchronize()
for (mode, had), shapes in sorted(ops.IMG_STATS.items()):
    print(f"mode {mode} image {had}: {len(shapes)} launches")
    if not had:
        for sh, n in collections.Counter(shapes).most_common(40):
            print("     ", n, "x  (B, Cin, Cout, K, Tin) =", sh)
