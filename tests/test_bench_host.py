"""CPU: the host-side pieces of bench.py that need no GPU -- the CPU-baseline thread count / model string (VERDICT r02:
"physical core count + CPU model"), the refusal of ``--gpus N`` on a node with fewer GPUs, the PMC-derived roofline fields,
and the algorithmic-bytes formula of the conv profiler."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_host_cpu_reports_usable_threads_and_a_model_string():
    import bench
    cores, model = bench.host_cpu()
    assert 1 <= cores <= (os.cpu_count() or 1)
    assert cores <= len(os.sched_getaffinity(0))
    assert isinstance(model, str) and model


def test_gpus_n_without_enough_gpus_is_refused_before_any_rank_starts():
    """On this CPU-only container ``python bench.py --gpus 2`` must fail loudly in the parent (no launcher, no ranks)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "STEGAN_BENCH_ONE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2" in r.stderr and "GPU(s)" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_step_rooflines_use_the_committed_counter_summary():
    import bench
    r = bench.step_rooflines(6.75e-3, "bf16")
    assert r["algorithmic_gflop"] == bench.STEP_GFLOP and abs(r["mfma_frac"] - 1016.9 / 6.75 / 2500.0) < 1e-3
    summary = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_summary.json")))
    moved = summary["hbm_bytes_per_step"]
    assert r["moved_gbytes_per_step_pmc"] == round(moved / 1e9, 3)
    assert abs(r["hbm_frac"] - moved / 1e9 / 6.75e-3 / 8000.0) < 1e-3          # on the bytes the kernels moved
    assert r["hbm_frac_layer_boundary_model"] < r["hbm_frac"]
    # every family the summary names carries the fields bench.py reads
    for fam in summary["families"].values():
        assert {"hbm_bytes_per_launch", "hbm_bytes_per_step", "mfma_busy_frac", "launches"} <= set(fam)
    assert 0.0 < summary["families"]["conv_bf16_kernel"]["mfma_busy_frac"] < 1.0


def test_conv_profiler_algorithmic_bytes():
    from ste_gan_amd import ops
    old = ops.COMPUTE_DTYPE
    try:
        shape = (0, 64, 768, 768, 3, 1, 1, 1, 1, 12, 12, 1, 0, 0)          # forward 768 -> 768, k3, T = 12
        x = y = 64 * 768 * 12
        w = 768 * 768 * 3
        ops.COMPUTE_DTYPE = "bf16"
        assert ops.ConvProfiler.alg_bytes(shape) == 2 * (x + w) + 6 * y    # bf16 operands, fp32 + image result
        assert ops.ConvProfiler.alg_bytes((2,) + shape[1:]) == 2 * (x + y) + 4 * w
        ops.COMPUTE_DTYPE = "f32"
        assert ops.ConvProfiler.alg_bytes((1,) + shape[1:]) == 4 * (y + w) + 4 * x
    finally:
        ops.COMPUTE_DTYPE = old
