"""Where do the torch-side copy / fill / add launches of one train step come from?  (eager step under the torch profiler)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

from bench import build_step, synthetic_batch
cfgobj = bench.make_cfg("train_step")
dev = torch.device("cuda", 0)
step = build_step(cfgobj, dev)
batch = synthetic_batch(dev, seed=1)
for _ in range(2): step(batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step(batch)
torch.cuda.synchronize()
agg = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::zeros", "aten::zero_", "aten::fill_", "aten::add", "aten::add_", "aten::zeros_like", "aten::cat", "aten::mul", "aten::div"):
        st = [s for s in ev.stack if "ste_gan_amd" in s or "bench.py" in s]
        agg[(ev.name, st[0].split("/")[-1] if st else "(autograd engine / other)", str(ev.input_shapes)[:60])] += 1
for k, n in agg.most_common(60):
    print(n, k)
