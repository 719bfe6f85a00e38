"""Diagnostic for the SIGABRT inside dist.destroy_process_group() after a three-hipGraph replay.

Each variant runs in its own child process (an abort ends it) with the C++ stack traces and RCCL warnings switched on;
the parent prints exit code + the tail of stderr.  Finding (round 3): the abort belongs to graph captures running in the default "global" capture-error mode while the RCCL
watchdog thread polls its work events (hipEventQuery from another thread invalidates the capture; the watchdog then dies
with the process): `cycles_global` reproduces it, every other mode -- captures in "thread_local" mode, the setting
GraphedTrainStep has used since 73d2040 -- survived 20 init / replay / destroy cycles with the graphs still alive.
Variants (PROBE_MODE):
  cycles / cycles_keep / cycles_close / cycles_global : four process groups in ONE process (eager, then 3 x three-graph
           replay), graphs dropped / kept alive / closed explicitly / captured in global error mode
  plain    : replay, then destroy_process_group() with every graph / stream / arena still alive  (the r02 failure)
  sync     : torch.cuda.synchronize() first
  ordered  : sync, drop the graphs + pool, drop the comm stream, sync, destroy
  nograph  : eager steps only (no capture), destroy
Usage: python scripts/teardown_probe.py            (parent; children are started with PROBE_MODE set)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def child(mode: str) -> None:
    import gc
    import torch
    import torch.distributed as dist
    import cases
    from test_gpu_train_step import build, to_batch
    from ste_gan_amd.train import GraphedTrainStep

    name = "step_c48_b3"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29300 + os.getpid() % 300), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if mode == "cycles_global":        # the r02 state: captures in the default global error mode beside a live RCCL watchdog
        os.environ["STEGAN_CAPTURE_ERROR_MODE"] = "global"
    if mode.startswith("cycles"):
        # what the r02 pytest session did in ONE process: a process group with eager steps, destroyed; then further
        # groups with three-graph replays, each destroyed with its graphs still alive
        keep = []
        for cyc in range(4):
            os.environ["MASTER_PORT"] = str(29300 + (os.getpid() + cyc) % 300)
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
            cfg, netG, netD, optG, optD, step = build(name)
            if cyc == 0:
                for s in range(2):
                    step(to_batch(cases.step_batch(name, s)))
            else:
                gs = GraphedTrainStep(step, to_batch(cases.step_batch(name, 0)), warmup=2)
                for s in range(2):
                    out = gs(to_batch(cases.step_batch(name, s)))
                gs.read_stats(out)
                if mode == "cycles_keep":
                    keep.append(gs)
                elif mode == "cycles_close":
                    gs.close()
            print(f"cycle {cyc}: destroying", flush=True)
            dist.destroy_process_group()
            print(f"cycle {cyc}: destroyed", flush=True)
        print("RESULT OK", flush=True)
        return
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    cfg, netG, netD, optG, optD, step = build(name)
    if mode == "nograph":
        for s in range(3):
            step(to_batch(cases.step_batch(name, s % 2)))
        torch.cuda.synchronize()
        print("steps done", flush=True)
        dist.destroy_process_group()
        print("RESULT OK", flush=True)
        return
    gs = GraphedTrainStep(step, to_batch(cases.step_batch(name, 0)), warmup=2)
    for s in range(3):
        out = gs(to_batch(cases.step_batch(name, s % 2)))
    print("replays queued", flush=True)
    if mode in ("sync", "ordered"):
        torch.cuda.synchronize()
    if mode == "ordered":
        gs.close() if hasattr(gs, "close") else None
        del gs, out
        gc.collect()
        torch.cuda.synchronize()
    dist.destroy_process_group()
    print("RESULT OK", flush=True)


def main():
    mode = os.environ.get("PROBE_MODE")
    if mode:
        child(mode)
        return
    modes = sys.argv[1:] or ["cycles_global", "cycles_global", "cycles_global", "cycles"]
    for m in modes:
        env = dict(os.environ, PROBE_MODE=m, TORCH_SHOW_CPP_STACKTRACES="1", NCCL_DEBUG="WARN", TORCH_CPP_LOG_LEVEL="INFO",
                   TORCH_NCCL_ASYNC_ERROR_HANDLING="1", PYTHONFAULTHANDLER="1")
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=600)
        print(f"===== mode {m}: exit {r.returncode}")
        print("--- stdout tail\n" + "\n".join(l for l in r.stdout.splitlines() if "alt_rsmi" not in l and l.strip())[-1500:])
        print("--- stderr tail\n" + "\n".join(l[:400] for l in r.stderr.splitlines() if "alt_rsmi" not in l and l.strip())[-6000:], flush=True)


if __name__ == "__main__":
    main()
