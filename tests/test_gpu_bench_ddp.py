"""bench.py under torch.distributed, 2 ranks (round-1 advisor finding: the instrumented pass ran on rank 0 only and its
all-reduces never completed).  One GPU box has ONE card, so both ranks share cuda:0 and exchange gradients through the
host (gloo; STEGAN_BENCH_ONE_GPU / STEGAN_BENCH_BACKEND) -- the control flow of every rank (timed steps, symmetric
instrumented steps, barrier, MAX over ranks, rank-0 JSON line, teardown) is the one the driver's RCCL runs take."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_with_profile_pass_completes():
    env = dict(os.environ, STEGAN_BENCH_BACKEND="gloo", STEGAN_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--profile-steps", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                    # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 128 and line["scaling"] == "weak"
    assert line["roofline"] is not None and line["roofline"]["frac"] > 0
    assert line["value"] > 0


@pytest.mark.gpu
def test_three_graph_step_matches_eager_step_under_ddp():
    """Under torch.distributed GraphedTrainStep replays the step as three hipGraphs around the two gradient all-reduces
    (train.GANTrainStep.phase_d / phase_g / phase_opt).  Two ranks, same seeds, the same batch every step, FIVE optimiser
    steps in both runs (the graph run spends two of them as GraphedTrainStep's own warm-up, so the eager run gets two more
    warm-up steps): the losses of the last step are EQUAL -- the step is bit-reproducible since round 4 (no float atomics
    on the gradient path, fixed-order reductions: parameters and gradients are equal, tests/test_gpu_reproducible.py), the
    host-staged sum of two ranks is commutative, a replayed graph launches the kernels the eager step launches, and since
    round 5 the REPORTED loss words are summed in a fixed order too (stegan_pair_loss_multi_det: the last float atomic of the
    library; 246.57199 against 246.57196 was seen with it).  (A mid-round-4 run once showed 3.596 against 4.166 here: the two modes had run a
    different NUMBER of optimiser steps while GraphedTrainStep's warm-up count was being changed -- the step counts above
    are what this test pins.)"""
    env = dict(os.environ, STEGAN_BENCH_BACKEND="gloo", STEGAN_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = {}
    for mode, extra, port, warm in (("graph", [], "29741", "1"), ("eager", ["--no-graph"], "29742", "3")):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup",
               warm, "--profile-steps", "0", "--no-cpu-baseline", "--no-f32"] + extra
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        res[mode] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert "3 hipGraphs" in res["graph"]["config"]["launch"] and res["eager"]["config"]["launch"] == "eager"
    lg, le = res["graph"]["last_step_losses"], res["eager"]["last_step_losses"]
    assert set(lg) == set(le)
    for k in lg:
        assert lg[k] == lg[k] and le[k] == le[k], f"{k}: NaN"           # finite in both
        assert lg[k] == le[k], f"{k}: graph {lg[k]} vs eager {le[k]}"


@pytest.mark.gpu
def test_bench_gpus_flag_starts_the_ranks_itself():
    """``python bench.py --gpus 2`` WITHOUT a launcher (the driver's command form): the parent starts two ranks under
    torch.distributed.run -- here both on the box's one card, gradients through the host -- and relays rank 0's line."""
    env = dict(os.environ, STEGAN_BENCH_BACKEND="gloo", STEGAN_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--profile-steps", "0",
           "--no-cpu-baseline", "--no-f32"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 128 and line["config"]["parallelism"] == "dp2"
