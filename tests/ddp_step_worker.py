"""Child process (one per rank) of test_gpu_ddp_step.py: two ranks share the one card of a GPU box and exchange gradients
through the host (gloo).  Each rank runs GANTrainStep -- eagerly or as three replayed hipGraphs (argv[2]) -- on ITS shard
of a detfill batch for two optimiser steps and writes its parameters; the parent compares the replicas with each other
and with the single-process step on the whole batch.  Leaves through ``parallel.shutdown`` (ordered teardown), not os._exit."""
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))


def main():
    out_dir, mode = sys.argv[1], sys.argv[2]
    from test_gpu_ddp_step import STEPS, build_nets, full_batch, to_device_batch
    from ste_gan_amd import ops
    from ste_gan_amd.parallel import shard_batch, shutdown
    from ste_gan_amd.train import GraphedTrainStep
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ops.set_compute_dtype(os.environ.get("DDP_STEP_DTYPE", "f32"))
    netG, netD, optG, optD, step = build_nets()
    assert step.comm is not None and step.comm.world_size == world
    shards = [to_device_batch(shard_batch(full_batch(s), rank, world)) for s in range(STEPS)]
    if os.environ.get("DDP_STEP_POISON") == "1" and rank == 1:
        # ONE rank's first batch carries a value that overflows the fp16 operands: its gradients turn non-finite, the SUM
        # all-reduce spreads that to every rank, and every rank's GradScaler must skip the same optimiser steps
        from ste_gan_amd.constants import DataType
        shards[0][DataType.REAL_EMG] = shards[0][DataType.REAL_EMG] * 1e30
    runner = step
    if mode == "eager_overlap":
        # the schedule of the single-graph capture, eagerly: the side branch is NOT joined before the D-gradient exchange
        step.join_side_after_phase_d = False
    if mode == "graph":
        sdG = {k: v.clone() for k, v in netG.state_dict().items()}
        sdD = {k: v.clone() for k, v in netD.state_dict().items()}
        runner = GraphedTrainStep(step, shards[0], warmup=2)
        assert runner.graphs is not None and len(runner.graphs) == 3
        netG.load_state_dict(sdG); netD.load_state_dict(sdD)
        for opt in (optG, optD):
            opt.reset_state()
        if step.scaler is not None:                      # the warm-up steps moved the scale too: start over
            step.scaler.load_state_dict(dict(scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000,
                                             _growth_tracker=0))
            step.scaler.state[3:].zero_()
    losses = []
    for s in range(STEPS):
        st = runner.read_stats(runner(shards[s]))
        losses.append([st["discriminator"], st["generator"]])
    torch.cuda.synchronize()
    scaler = None
    if step.scaler is not None:
        scaler = dict(scale=step.scaler.get_scale(), skipped=step.scaler.skipped_steps(), d_steps=optD.device_step_count(),
                      g_steps=optG.device_step_count())
    torch.save({"g": {k: v.cpu() for k, v in netG.state_dict().items()}, "d": {k: v.cpu() for k, v in netD.state_dict().items()},
                "losses": losses, "scaler": scaler}, os.path.join(out_dir, f"{mode}_rank{rank}.pt"))
    dist.barrier()
    shutdown(runner, step)
    print(f"RANK {rank} DONE", flush=True)


if __name__ == "__main__":
    main()
