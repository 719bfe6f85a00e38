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


def test_step_rooflines_are_algorithmic_and_counters_carry_their_provenance(monkeypatch, tmp_path):
    """``hbm_frac`` is algorithmic bytes / time / peak (SURVEY's models, never bytes moved); the bytes the
    kernels moved are separate ``moved_*`` fields that name the profile they came from and are refused when that profile was
    measured on other kernel sources."""
    import bench
    from ste_gan_amd.build import source_hash
    t, fps = 6.0e-3, 12288 / 6.0e-3
    monkeypatch.setattr(bench, "PMC", {})
    monkeypatch.setattr(bench, "PMC_STALE", "stale: test")
    r = bench.step_rooflines(t, "bf16", fps)
    assert r["algorithmic_gflop"] == bench.STEP_GFLOP and abs(r["mfma_frac"] - 1016.9 / 6.0 / 2500.0) < 1e-3
    assert abs(r["hbm_frac"] - 3.79 / t / 8000.0) < 1e-4 and abs(r["hbm_frac_fp32_model"] - 6.58 / t / 8000.0) < 1e-4
    assert abs(r["frames_frac_of_hbm_bound"] - fps / 26.0e6) < 1e-4 and abs(r["frames_frac_of_mfma_bound"] - fps / 30.4e6) < 1e-4
    assert abs(r["frac_of_target"] - fps / 13.0e6) < 1e-4
    assert r["moved_gbytes_per_step_pmc"] is None and r["moved_profile"] == "stale: test"
    # a profile measured on THESE sources is used, one measured on others is not
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (prof / "r09_pmc_summary.json").write_text(json.dumps({"csrc_sha": "0" * 16, "hbm_bytes_per_step": 1e10, "families": {}}))
    pmc, why = bench.load_pmc()
    assert pmc == {} and "measured on kernel sources" in why
    (prof / "r10_pmc_summary.json").write_text(json.dumps({"csrc_sha": source_hash(), "measured_utc": "2026-10-05T00:00Z",
                                                           "hbm_bytes_per_step": 1.2e10, "families": {}}))
    pmc, why = bench.load_pmc()
    assert why is None and pmc["_file"].endswith("r10_pmc_summary.json")
    monkeypatch.setattr(bench, "PMC", pmc)
    r = bench.step_rooflines(t, "bf16", fps)
    assert r["moved_gbytes_per_step_pmc"] == 12.0 and abs(r["moved_over_algorithmic"] - 12.0 / 3.79) < 0.01
    assert r["moved_profile"]["csrc_sha"] == source_hash()
    assert abs(r["hbm_frac"] - 3.79 / t / 8000.0) < 1e-4             # unchanged by the counters


def test_committed_counter_summaries_carry_the_fields_bench_reads():
    import glob
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")):
        summary = json.load(open(f))
        for fam in summary["families"].values():
            assert {"hbm_bytes_per_launch", "hbm_bytes_per_step", "mfma_busy_frac", "launches"} <= set(fam)
        assert 0.0 < summary["families"]["conv_bf16_kernel"]["mfma_busy_frac"] < 1.0


def test_conv_profiler_algorithmic_bytes():
    from ste_gan_amd import ops
    old = (ops.COMPUTE_DTYPE, ops.HALF)
    try:
        shape = (0, 64, 768, 768, 3, 1, 1, 1, 1, 12, 12, 1, 0, 0)          # forward 768 -> 768, k3, T = 12
        x = y = 64 * 768 * 12
        w = 768 * 768 * 3
        ops.COMPUTE_DTYPE, ops.HALF = "bf16", True
        assert ops.ConvProfiler.alg_bytes(shape) == 2 * (x + w) + 6 * y    # bf16 operands, fp32 + image result
        assert ops.ConvProfiler.alg_bytes((2,) + shape[1:]) == 2 * (x + y) + 4 * w
        ops.COMPUTE_DTYPE, ops.HALF = "f32", False
        assert ops.ConvProfiler.alg_bytes((1,) + shape[1:]) == 4 * (y + w) + 4 * x
    finally:
        ops.COMPUTE_DTYPE, ops.HALF = old


def test_bench_refuses_result_changing_switches_and_records_the_other_knobs():
    """with a diagnostic switch set (skipped entry points, what-if passes, ablated kernels, another library
    in place of the built one) ``bench.py`` exits non-zero BEFORE touching the GPU; every other STEGAN_* knob is printed with
    the result (``env``), this file's own rehearsal switches excepted."""
    import bench
    refused, recorded = bench.check_env({"STEGAN_SKIP": "stegan_act_bwd", "STEGAN_WHATIF_C": "3", "STEGAN_HIP_LIB_F16": "/x.so",
                                         "STEGAN_TILE_NS": "3", "STEGAN_BENCH_DETAIL": "1", "HOME": "/root"})
    assert refused == ["STEGAN_HIP_LIB_F16", "STEGAN_SKIP", "STEGAN_WHATIF_C"]
    assert recorded == {"STEGAN_TILE_NS": "3"}
    for name in ("STEGAN_SKIP", "STEGAN_WHATIF", "STEGAN_WHATIF_C", "STEGAN_BF16_DBG", "STEGAN_HIP_LIB"):
        env = dict(os.environ, **{name: "1"})
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and name in r.stderr and "refusing" in r.stderr, (name, r.stderr[-300:])
        assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_the_product_package_has_no_result_changing_switches():
    """The round-4 diagnostics (STEGAN_SKIP in the ctypes binding, STEGAN_WHATIF in ops / train) are gone from ste_gan_amd/; the
    library reads its wrong-result knobs in the diagnostic builds only (#ifdef STEGAN_BF16_ABLATE / STEGAN_STAMPS)."""
    import re
    pkg = os.path.join(ROOT, "ste_gan_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert "STEGAN_SKIP" not in text and "STEGAN_WHATIF" not in text, f
    for f in os.listdir(os.path.join(pkg, "csrc")):
        if not f.endswith((".hip", ".h")):
            continue
        lines = open(os.path.join(pkg, "csrc", f)).read().split("\n")
        depth_diag = 0
        for ln in lines:
            s = ln.strip()
            if s.startswith("#ifdef STEGAN_BF16_ABLATE") or s.startswith("#ifdef STEGAN_STAMPS") or s.startswith("#if defined(STEGAN_STAMPS)"):
                depth_diag += 1
            elif s.startswith("#endif") and depth_diag:
                depth_diag -= 1
            elif depth_diag == 0 and re.search(r'getenv\("STEGAN_(WHATIF_C|BF16_DBG|STAMP_PTR)"', s):
                raise AssertionError(f"{f}: {s}")
