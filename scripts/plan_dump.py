"""Plans (tile, chunk, slices / splits) of every bf16 conv / weight-gradient launch of one eager train step at the bench
configuration: STEGAN_PLAN_DEBUG=1 python scripts/plan_dump.py 2> plans.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ste_gan_amd import ops
ops.set_compute_dtype("bf16")
dev = torch.device("cuda", 0)
step = bench.build_step(bench.make_cfg("train_step"), dev)
batch = bench.synthetic_batch(dev, 1000)
step(batch)
torch.cuda.synchronize()
print("=== step ===", file=sys.stderr, flush=True)
step(batch)
torch.cuda.synchronize()
