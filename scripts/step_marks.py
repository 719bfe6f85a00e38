"""Time marks inside the captured train step: one-lane kernels storing the chip's 100 MHz real-time counter at the phase
boundaries of GANTrainStep (ops.mark), captured with the step and read back after replays -- where a replay's wall time goes
per stream, without a profiler in the way.    python scripts/step_marks.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from ste_gan_amd import ops
from ste_gan_amd.train import GraphedTrainStep

ops.set_compute_dtype(os.environ.get("PHASE_DTYPE", "bf16"))
dev = torch.device("cuda", 0)
eager = bench.build_step(bench.make_cfg("train_step"), dev)
batch = bench.synthetic_batch(dev, seed=1000)
for _ in range(2):
    eager(batch)
torch.cuda.synchronize()
ops.MARKS = {"buf": torch.zeros(256, dtype=torch.int64, device=dev), "names": []}
step = GraphedTrainStep(eager, batch, warmup=0)          # the capture records the marks
names = list(ops.MARKS["names"])
n = len(names)
acc = np.zeros(n)
reps = 30
for _ in range(5):
    step(batch)
for _ in range(reps):
    step(batch)
    torch.cuda.synchronize()
    v = ops.MARKS["buf"][:n].cpu().numpy().astype(np.float64)
    acc += (v - v[0]) / 100.0          # us since "step start"
acc /= reps
for nm, t in sorted(zip(names, acc), key=lambda kv: kv[1]):
    print(f"{t:9.1f} us  {nm}")

# host side: how long does hipGraphLaunch take to SUBMIT a replay (the GPU runs behind it)?
import time
torch.cuda.synchronize()
hs = []
for _ in range(20):
    t0 = time.perf_counter()
    step.graph.replay()
    hs.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
print(f"host time of graph.replay(): median {1e3 * sorted(hs)[len(hs) // 2]:.3f} ms, min {1e3 * min(hs):.3f} ms "
      f"(the GPU time of a step is the last mark above)")
