"""Can an RCCL all-reduce be CAPTURED into a hipGraph with this torch / ROCm (1-rank group on one GPU)?
    timeout -k 10 120 python scripts/rccl_capture_probe.py"""
import os, sys
import torch
import torch.distributed as dist

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", HSA_ENABLE_IPC_MODE_LEGACY="0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.ones(1 << 20, device="cuda")
y = torch.zeros_like(x)
dist.all_reduce(x)                      # warm-up: communicator creation outside the capture
torch.cuda.synchronize()
comm = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        x.mul_(2.0)
        comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(comm):
            dist.all_reduce(x)
        y.add_(1.0)                     # independent work beside the exchange
        torch.cuda.current_stream().wait_stream(comm)
        y.add_(x)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("CAPTURE OK: x[0] =", float(x[0]), "y[0] =", float(y[0]), flush=True)
except Exception as e:                  # noqa: BLE001
    print("CAPTURE FAILED:", type(e).__name__, str(e)[:300], flush=True)
finally:
    torch.cuda.synchronize()
    del g
    dist.destroy_process_group()
print("done", flush=True)
