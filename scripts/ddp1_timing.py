"""Step time of the three-graph (torch.distributed) replay with a 1-rank RCCL group against the single-graph replay, and
what of the difference is the segmentation itself (NO_EXCHANGE=1: the three graphs back to back, no all-reduce calls)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
from ste_gan_amd import ops
from ste_gan_amd.train import GraphedTrainStep

ops.set_compute_dtype("bf16")
dev = torch.device("cuda", 0)
if os.environ.get("WITH_GROUP", "1") == "1":
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29555", HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
cfg = bench.make_cfg("train_step")
eager = bench.build_step(cfg, dev)
batch = bench.synthetic_batch(dev, seed=1000)
step = GraphedTrainStep(eager, batch, warmup=2)
if os.environ.get("NO_EXCHANGE") == "1" and step.graphs:
    eager.exchange_d = lambda: setattr(eager, "_d_handle", None)
    eager.exchange_g = lambda: setattr(eager, "_g_handle", None)
for _ in range(3):
    step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 30
for _ in range(n):
    step(batch)
torch.cuda.synchronize()
print(f"graphs={'3' if step.graphs else '1'}  {1e3 * (time.perf_counter() - t0) / n:.3f} ms/step", flush=True)
os._exit(0)
