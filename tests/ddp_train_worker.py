"""Child process (one per rank) of test_gpu_train_loop.py::test_train_loop_two_ranks_with_graph_replay: two ranks share the one
card of a GPU box and exchange gradients through the host (gloo), so the process group is created here, before train()."""
import os
import sys
from pathlib import Path

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))


def main():
    tmp = Path(sys.argv[1])
    from test_gpu_train_loop import _cfg
    from ste_gan_amd.train import train
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    out = tmp / "run_ddp"
    if rank == 0:
        out.mkdir(exist_ok=True)
    dist.barrier()
    train(_cfg(tmp, max_steps=4), out, None, "cuda:0", debug=False, emg_enc_ckpt=tmp / "enc.pt")
    dist.barrier()
    from ste_gan_amd.parallel import shutdown
    shutdown()                         # train() released its graphs on the way out
    print(f"RANK {rank} DONE", flush=True)


if __name__ == "__main__":
    main()
