"""ms per step of the data-parallel launch arrangements on ONE GPU with a 1-rank RCCL group (the all-reduces are issued and
awaited like at N ranks, they just move nothing): three graphs with the side branch split over the first two ("3"), the
round-2 three graphs with the whole branch in the first ("3w"), ONE graph with the all-reduces captured inside it ("1c", round 4)
and -- without a process group -- the single graph ("1").
What it shows is the cost of the graph boundaries and of the lost overlap, not the exchange time.
    python scripts/ddp_graph_timing.py [3|3w|1c|1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
from ste_gan_amd import ops
from ste_gan_amd.train import GraphedTrainStep
from ste_gan_amd.parallel import shutdown

mode = sys.argv[1] if len(sys.argv) > 1 else "3"
ops.set_compute_dtype("bf16")
dev = torch.device("cuda", 0)
if mode != "1":
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29400 + os.getpid() % 200), STEGAN_SPLIT_SIDE=("0" if mode == "3w" else "1"),
                      STEGAN_DDP_GRAPH=("single" if mode == "1c" else "three"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
step = bench.build_step(bench.make_cfg("train_step"), dev)
batch = bench.synthetic_batch(dev, 1000)
gs = GraphedTrainStep(step, batch, warmup=2)
print("ddp_mode", gs.ddp_mode, flush=True)
for _ in range(5):
    gs(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 200
for _ in range(N):
    gs(batch)
torch.cuda.synchronize()
print(f"graphs={mode}: {(time.perf_counter() - t0) / N * 1e3:.3f} ms/step", flush=True)
# host side alone: enqueue time of one replay with the device idle at its start (nothing to wait for, nothing queued)
hs = []
for _ in range(10):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    gs(batch)
    hs.append(time.perf_counter() - t1)
torch.cuda.synchronize()
print(f"graphs={mode}: host enqueue {sorted(hs)[len(hs) // 2] * 1e3:.3f} ms/step (median of 10, device idle at the start)", flush=True)
if mode != "1":
    shutdown(gs, step)
