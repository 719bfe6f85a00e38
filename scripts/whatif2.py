"""What-if timings, round 4: one variant per process (the switches are read at import / first launch).
    STEGAN_WHATIF=no_dreal_g python scripts/whatif2.py      STEGAN_WHATIF_C=3 python scripts/whatif2.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ste_gan_amd import ops
from ste_gan_amd.train import GraphedTrainStep

ops.set_compute_dtype("bf16")
dev = torch.device("cuda", 0)
cfg = bench.make_cfg("train_step")
step = GraphedTrainStep(bench.build_step(cfg, dev), bench.synthetic_batch(dev, 1000), warmup=2)
batch = bench.synthetic_batch(dev, 1000)
for _ in range(5):
    step(batch)
torch.cuda.synchronize()
ts = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(100):
        step(batch)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 100 * 1e3)
print(f"WHATIF={os.environ.get('STEGAN_WHATIF', '')} WHATIF_C={os.environ.get('STEGAN_WHATIF_C', '')} "
      f"{min(ts):.3f} ms/step (3 x 100 replays: {' '.join(f'{t:.3f}' for t in ts)})", flush=True)
