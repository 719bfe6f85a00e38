#!/usr/bin/env python3
"""Benchmark of the STE-GAN hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One "step" = one full GAN train step (ste_gan/train.py:165-268: G fwd, 4 D fwd, both backward passes,
all losses incl. the frozen EMG encoder, 2 fused AdamW steps) on one synthetic batch per GPU:
B=64, chunk_size=200 -> T_su=12 Soft-SU frames / T_emg=192 EMG samples, random-init
full-size networks (G 768 ch, small D, 768-d 6-layer encoder), inputs resident in HBM before timing.
Metric: EMG frames/s = n_gpus * B * 192 * steps / wall time (weak scaling).  dtype = arithmetic type of the convolution
operands: bf16 (BASELINE config 3, the default: operands rounded to bf16 while they are staged, fp32 accumulation, fp32
activations / master weights / optimiser state) or f32 (the parity path, --dtype f32; at N=1 its step time is also
measured in the same run and reported as "fp32_parity_path").

Extra objects on the JSON line (task contract): "roofline" for the dominant kernel (per-launch HIP-event
timing of every conv launch in a separate instrumented pass; algorithmic FLOPs = nominal dense MACs x 2,
SURVEY.md section 8d) and "cpu_baseline" (the oracle's train step -- a torch-CPU restatement of the reference
-- timed on this box's host cores, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

B, T_SU, T_EMG = 64, 12, 192
MFMA_PEAK_TFLOPS = {"f32": 157.3,   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
                    "bf16": 2500.0, "f16": 2500.0}  # ... v_mfma_f32_32x32x16_bf16, dense (~2.5 PF; never the 2:1-sparsity figure)
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s measured with a float4 copy)
STEP_GFLOP = 1016.9                 # required nominal work of one step at B=64, T_emg=192 (508.4 GMAC)
STEP_GBYTES_F32 = 6.58              # ... layer-boundary traffic, every activation written and read once as fp32 (SURVEY's fp32 model)
STEP_GBYTES_BF16 = 3.79             # ... with 2-byte activations / weights and fp32 optimiser traffic: the model behind SURVEY's
                                    # 26.0 M frames/s HBM bound for config 3, the figure BASELINE.json's target is phrased in
FRAMES_HBM_BOUND = {"bf16": 26.0e6, "f16": 26.0e6, "f32": 14.9e6}     # SURVEY.md section 8d, frames/s on one GPU
FRAMES_MFMA_BOUND = {"bf16": 30.4e6, "f16": 30.4e6, "f32": 1.90e6}


def load_pmc():
    """Counters of the committed rocprofv3 --pmc passes over this command (scripts/pmc_summary.py): the newest
    profiles/r*_pmc_summary.json whose recorded kernel-source hash equals this tree's.  A profile measured on other kernels is
    NOT used (the fields it would feed are printed as null with the reason)."""
    import glob
    from ste_gan_amd.build import source_hash
    have = source_hash()
    stale = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("csrc_sha") == have:
            d["_file"] = os.path.relpath(f, ROOT)
            return d, None
        stale = stale or f"{os.path.relpath(f, ROOT)} was measured on kernel sources {d.get('csrc_sha', '(unrecorded)')}, the tree is {have}"
    return {}, stale or "no profiles/r*_pmc_summary.json"


PMC, PMC_STALE = {}, "not loaded"


def make_cfg(workload):
    import yaml
    from ste_gan_amd.train_utils import DictConfig
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "ste_gan_base_gantts.yaml")))
    cfg["data"] = yaml.safe_load(open(os.path.join(ROOT, "configs", "data", "gaddy_and_klein_corpus.yaml")))
    cfg["emg_encoder"] = yaml.safe_load(open(os.path.join(ROOT, "configs", "emg_encoder", "conv_transformer.yaml")))
    cfg["train"]["batch_size"], cfg["train"]["chunk_size"] = B, 200
    return DictConfig(cfg)


def synthetic_batch(device, seed):
    from ste_gan_amd.constants import DataType
    g = torch.Generator().manual_seed(seed)
    return {DataType.SPEECH_UNITS: torch.randn(B, T_SU, 256, generator=g).to(device),
            DataType.REAL_EMG: torch.tanh(torch.randn(B, T_EMG, 8, generator=g)).to(device),
            DataType.SESSION_INDEX: torch.randint(0, 17, (B,), generator=g).to(device),
            DataType.SPEAKING_MODE_INDEX: torch.zeros(B, dtype=torch.long).to(device),
            DataType.PHONEMES: torch.randint(0, 48, (B, T_SU), generator=g).to(device)}


def build_step(cfg, device):
    from ste_gan_amd.losses.emg_encoder_loss import EMGEncoderLoss
    from ste_gan_amd.losses.time_domain_loss import MultiTimeDomainFeatureLoss
    from ste_gan_amd.models.discriminator import init_emg_discriminators
    from ste_gan_amd.models.emg_encoder import init_emg_encoder
    from ste_gan_amd.models.generator import init_emg_generator
    from ste_gan_amd.optim import FlatParams, FusedAdamW
    from ste_gan_amd.train import GANTrainStep
    torch.manual_seed(0)                                  # identical replicas on every rank
    netG = init_emg_generator(cfg).to(device)
    netD = init_emg_discriminators(cfg).to(device)
    enc = init_emg_encoder(cfg, device).eval()
    from ste_gan_amd.train import make_optimizers
    optG, optD = make_optimizers(netG, netD)
    step = GANTrainStep(cfg, netG, netD, EMGEncoderLoss(enc), MultiTimeDomainFeatureLoss(8), optG, optD)
    return step


def step_rooflines(sec_per_step, dtype, frames_per_s):
    """Whole-step view against both roofs (SURVEY.md section 8d asks for both fractions and which one binds).  Every
    fraction here is ALGORITHMIC work / time / peak: bytes a step must move under SURVEY's layer-boundary model (fp32 and
    2-byte-storage variants), never bytes the kernels happened to move.  What they DID move (PMC: 2 x FETCH_SIZE + WRITE_SIZE
    over every kernel of a step -- bf16 images beside the fp32 tensors, split-K and weight-gradient slabs, re-reads the
    XCD L2s did not absorb) is reported separately, as ``moved_*``, with the profile it was read from."""
    gb = STEP_GBYTES_BF16 if dtype in ("bf16", "f16") else STEP_GBYTES_F32
    r = {"algorithmic_gflop": STEP_GFLOP,
         "algorithmic_gbytes_fp32_model": STEP_GBYTES_F32, "algorithmic_gbytes_bf16_model": STEP_GBYTES_BF16,
         "per_gpu_tflops": round(STEP_GFLOP / (1e3 * sec_per_step), 2),
         "mfma_frac": round(STEP_GFLOP / (1e3 * sec_per_step) / MFMA_PEAK_TFLOPS[dtype], 4),
         "hbm_frac": round(gb / sec_per_step / HBM_PEAK_GBS, 4),
         "hbm_frac_model": "2-byte operand storage (3.79 GB/step)" if dtype in ("bf16", "f16") else "fp32 storage (6.58 GB/step)",
         "hbm_frac_fp32_model": round(STEP_GBYTES_F32 / sec_per_step / HBM_PEAK_GBS, 4),
         "hbm_frac_bf16_model": round(STEP_GBYTES_BF16 / sec_per_step / HBM_PEAK_GBS, 4),
         "frames_frac_of_hbm_bound": round(frames_per_s / FRAMES_HBM_BOUND[dtype], 4),
         "frames_frac_of_mfma_bound": round(frames_per_s / FRAMES_MFMA_BOUND[dtype], 4),
         "target_frames_per_s": 0.5 * FRAMES_HBM_BOUND[dtype],
         "frac_of_target": round(frames_per_s / (0.5 * FRAMES_HBM_BOUND[dtype]), 4)}
    moved = PMC.get("hbm_bytes_per_step") if dtype == "bf16" else None
    if moved:
        r["moved_gbytes_per_step_pmc"] = round(moved / 1e9, 3)
        r["moved_over_algorithmic"] = round(moved / 1e9 / gb, 2)
        r["moved_gbytes_per_s"] = round(moved / 1e9 / sec_per_step, 1)
        r["moved_profile"] = {"file": PMC.get("_file"), "csrc_sha": PMC.get("csrc_sha"), "measured_utc": PMC.get("measured_utc")}
    else:
        r["moved_gbytes_per_step_pmc"] = None
        r["moved_profile"] = PMC_STALE if dtype == "bf16" else "counters exist for the bf16 step only"
    r["binding"] = ("neither roof: the step is ~520 launches of 6-25 us, each bound by its blocks' fixed costs and by the "
                    "CU's fetch path into LDS, and two streams cannot share a CU (DESIGN.md section 6.0)")
    return r


def host_cpu():
    """(physical cores, model name) of the host from /proc/cpuinfo (torch.get_num_threads() counts SMT siblings on some
    hosts and a cgroup share on others: the physical count is what the threads were pinned to below)."""
    cores, model, phys, core = set(), "unknown", None, None
    try:
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None and core is not None:
                cores.add((phys, core)); phys = core = None
        if phys is not None and core is not None:
            cores.add((phys, core))
    except OSError:
        pass
    n = len(cores) or (os.cpu_count() or 1)
    try:
        n = min(n, len(os.sched_getaffinity(0)))          # never more threads than this process may run on
    except AttributeError:
        pass
    try:                                                  # ... nor than its cgroup's CPU quota pays for
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n), model


def cpu_baseline(steps=5, warm=2):
    """Oracle (CPU restatement of the reference step, torch ATen/oneDNN fp32) on a bounded sample: B=64, warm + steps steps
    (2 warm-up + >= 5 timed), one thread per physical core."""
    from oracle import shapes as oshapes
    from oracle import stegan_oracle as O
    torch.manual_seed(0)
    cores, model = host_cpu()
    torch.set_num_threads(cores)

    def rnd(shapes):
        sd = {}
        for k, s in shapes.items():
            if k.endswith("num_batches_tracked"):
                sd[k] = torch.zeros((), dtype=torch.long)
            elif k.endswith("running_var") or k.endswith("weight_g"):
                sd[k] = torch.rand(s) + 0.5
            else:
                fan = max(1, int(torch.tensor(s[1:]).prod())) if len(s) > 1 else 1
                sd[k] = torch.randn(s) / fan ** 0.5
        return sd

    g_sd, d_sd, e_sd = rnd(oshapes.generator_shapes()), rnd(oshapes.discriminator_shapes(True)), rnd(oshapes.encoder_shapes())
    for k in d_sd:
        if O.is_sn_buffer(d_sd, k):
            d_sd[k] = torch.nn.functional.normalize(d_sd[k], dim=0)
    d_train = {k: v for k, v in d_sd.items() if not O.is_sn_buffer(d_sd, k)}
    og, od = O.adamw_init(g_sd), O.adamw_init(d_train)
    gen = torch.Generator().manual_seed(0)
    batch = {"su": torch.randn(B, T_SU, 256, generator=gen), "x_real": torch.tanh(torch.randn(B, T_EMG, 8, generator=gen)),
             "sess": torch.randint(0, 17, (B,), generator=gen), "phonemes": torch.randint(0, 48, (B, T_SU), generator=gen)}
    batch["feats"] = batch["su"]
    for _ in range(warm):
        O.train_step(g_sd, d_sd, e_sd, og, od, batch)
    times = []
    for _ in range(steps):
        t0 = time.time()
        O.train_step(g_sd, d_sd, e_sd, og, od, batch)
        times.append(time.time() - t0)
    dt = sum(times) / steps
    return {"value": B * T_EMG / dt, "unit": "EMG frames/s", "cores": cores, "kind": "port", "cpu_model": model,
            "sample": f"{steps} full train steps at B=64, T_emg=192 after {warm} warm-up steps (oracle/stegan_oracle.py "
                      f"train_step, torch-CPU fp32, {cores} threads = min(physical cores, affinity, cgroup CPU quota); mean {dt:.2f} s/step, "
                      f"min {min(times):.2f}, max {max(times):.2f})"}


def gen_infer(args, cfg, device, quiet=False):
    """BASELINE config 2: Generator.generate on (64, 12, 256) Soft-SU -> (64, 192, 8) in the current compute mode (one replayed
    hipGraph per forward unless --no-graph; not the headline metric)."""
    from ste_gan_amd import ops as _ops
    mode = "bf16 conv operands, fp32 accumulation / activations" if _ops.bf16_enabled() else "fp32"
    from ste_gan_amd.constants import DataType
    from ste_gan_amd.models.generator import init_emg_generator
    torch.manual_seed(0)
    netG = init_emg_generator(cfg).to(device).eval()
    b = synthetic_batch(device, 1000)
    eager = lambda: netG.generate(b[DataType.SPEECH_UNITS], b[DataType.SESSION_INDEX], b[DataType.SPEAKING_MODE_INDEX])
    with torch.no_grad():
        for _ in range(max(2, args.warmup)):
            eager()
        torch.cuda.synchronize()
        run, launch = eager, "eager launches"
        if not args.no_graph:                  # fixed shapes: the ~40 launches of a forward replay as one hipGraph
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    y_static = eager()
            torch.cuda.current_stream().wait_stream(side)
            run, launch = graph.replay, "hipGraph replay"
        for _ in range(args.warmup):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    line = {"metric": "EMG frames/sec, generator-only inference at B=64 chunk=200 (BASELINE config 2)",
            "value": round(B * T_EMG * args.steps / dt, 1), "unit": "EMG frames/s", "n_gpus": 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "dtype": _ops.COMPUTE_DTYPE, "data": "synthetic",
            "config": {"workload": f"Generator.generate, B=64, T_su=12 -> T_emg=192, {mode}, {launch}"},
            "algorithmic_tflops": round(2 * 53.02e9 * args.steps / dt / 1e12, 2)}
    if not quiet:
        print(json.dumps(line), flush=True)
    return line


def launch_ranks(n: int) -> int:
    """``python bench.py --gpus N`` outside a launcher: start N ranks (one per GPU) as FRESH child processes under
    ``torch.distributed.run`` with the same arguments and relay rank 0's JSON line.  This process never touches the GPU
    (nothing before this point initialises HIP; ``torch.cuda.device_count()`` does not on this image), and the ranks are
    children, not an exec of this process."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and not os.environ.get("STEGAN_BENCH_ONE_GPU"):
        raise SystemExit(f"bench.py: --gpus {n} but this node shows {have} GPU(s) "
                         "(STEGAN_BENCH_ONE_GPU=1 + STEGAN_BENCH_BACKEND=gloo rehearses N ranks on one card)")
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    for l in r.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if r.returncode != 0 or len(lines) != 1:
        raise SystemExit(f"bench.py: the {n}-rank run failed (exit {r.returncode}, {len(lines)} result lines)")
    print(lines[0], flush=True)
    return 0



# ---- the bench line is tamper-evident.  Switches that change RESULTS (round-4 diagnostics: skipped entry points, what-if
# passes, ablated kernels, another library in place of the built one) make the run refuse; every other STEGAN_* knob it saw
# (planner experiments: tile shapes, slice counts, stream counts -- same results, different speed) is printed in the line.
FORBIDDEN_ENV = ("STEGAN_SKIP", "STEGAN_WHATIF", "STEGAN_WHATIF_C", "STEGAN_BF16_DBG", "STEGAN_STAMP_PTR")
FORBIDDEN_ENV_PREFIXES = ("STEGAN_HIP_LIB",)
HARNESS_ENV = ("STEGAN_BENCH_",)               # this file's own rehearsal / detail switches: not kernel knobs


def check_env(environ=None):
    """(refused, recorded): names that make the run refuse, and the other STEGAN_* settings to print with the result."""
    environ = os.environ if environ is None else environ
    refused = sorted(k for k in environ if k in FORBIDDEN_ENV or k.startswith(FORBIDDEN_ENV_PREFIXES))
    recorded = {k: environ[k] for k in sorted(environ) if k.startswith("STEGAN_") and k not in refused
                and not k.startswith(HARNESS_ENV)}
    return refused, recorded


def guard_env():
    refused, recorded = check_env()
    if refused:
        raise SystemExit("bench.py: refusing to measure with " + ", ".join(refused) + " set: these switches change the step's "
                         "RESULTS (diagnostics) or swap the library under test; unset them")
    return recorded


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--profile-steps", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f32", action="store_true", help="skip the fp32 parity-path timing beside a bf16 run")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of replaying the captured hipGraph")
    ap.add_argument("--dtype", choices=["f32", "bf16", "f16"], default="bf16",
                    help="arithmetic type of the convolution operands (accumulation, activations, weights and optimiser state "
                         "stay fp32): f32 = parity path, bf16 = BASELINE config 3 (the metric's), f16 = the same step on the fp16 "
                         "build of the kernels with the device-side GradScaler (BASELINE config 5's arithmetic)")
    ap.add_argument("--workload", choices=["train_step", "gen_infer"], default="train_step",
                    help="train_step = the metric's workload (default); gen_infer = BASELINE config 2 (generator-only inference)")
    args = ap.parse_args()
    env_seen = guard_env()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # STEGAN_BENCH_BACKEND=gloo + STEGAN_BENCH_ONE_GPU=1: the multi-rank rehearsal of tests/test_gpu_bench_ddp.py
        # (several ranks on ONE card, host-staged all-reduce); the driver's runs use RCCL, one rank per GPU
        if os.environ.get("STEGAN_BENCH_ONE_GPU"):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        backend = os.environ.get("STEGAN_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    device = torch.device("cuda", local_rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")

    from ste_gan_amd import ops
    ops.set_compute_dtype(args.dtype)
    global PMC, PMC_STALE
    PMC, PMC_STALE = load_pmc()
    cfg = make_cfg("train_step")
    if args.workload == "gen_infer":
        return gen_infer(args, cfg, device)
    step = build_step(cfg, device)
    batch = synthetic_batch(device, seed=1000 + rank)      # rank-offset seeds: different data per replica
    eager_step = step
    graphed = not args.no_graph
    if graphed:        # one captured hipGraph per step; under torch.distributed three (the phases around the two all-reduces)
        from ste_gan_amd.train import GraphedTrainStep
        if world == 1:
            step = GraphedTrainStep(eager_step, batch, warmup=2)
        else:
            # every rank must take the same path (the eager step and the graphs issue the same collectives, but a rank
            # that failed to capture must not leave the others replaying alone): agree on the outcome
            try:
                step = GraphedTrainStep(eager_step, batch, warmup=2)
                ok = 1
            except Exception as e:                         # noqa: BLE001 -- reported, then the run continues eagerly
                print(f"[bench] rank {rank}: graph capture failed ({type(e).__name__}: {e}); eager launches", file=sys.stderr)
                ok = 0
            flag = torch.tensor([ok], device=device if dist.get_backend() == "nccl" else "cpu", dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                step, graphed = eager_step, False

    def barrier():
        if world > 1:
            dist.barrier()

    # inputs resident in HBM before the timed region: a replayed graph reads its static input buffers, so the synthetic
    # batch is placed there once (a training loop's batch-assembly kernel writes into them directly)
    run_batch = batch
    if graphed:
        step.load_batch(batch)
        run_batch = None
    for _ in range(args.warmup):
        step(run_batch)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(run_batch)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    stats = step.read_stats(step.last)

    # ---- roofline of the dominant kernel: instrumented pass, HIP events around every conv launch
    # EVERY rank runs the instrumented steps: under torch.distributed each of them carries the two gradient
    # all-reduces, and a collective that only rank 0 enters never completes.  Only rank 0 reports.
    roof = None
    if args.profile_steps > 0:
        ops.PROFILER = ops.ConvProfiler()
        # one stream for the instrumented pass: an event bracket around a launch on one stream would otherwise also time
        # whatever the other streams run beside it (and the choice of the "dominant" kernel would vary from run to run)
        saved = (eager_step.overlap_encoder, os.environ.get("STEGAN_D_STREAMS"))
        eager_step.overlap_encoder = False
        os.environ["STEGAN_D_STREAMS"] = "0"
        try:
            for _ in range(args.profile_steps):
                eager_step(batch)                          # instrumented pass: eager launches, HIP events per conv
        finally:
            eager_step.overlap_encoder = saved[0]
            if saved[1] is None:
                os.environ.pop("STEGAN_D_STREAMS", None)
            else:
                os.environ["STEGAN_D_STREAMS"] = saved[1]
        agg = ops.PROFILER.summary()
        # what a HIP event pair adds to the launch it brackets: the same pair around a one-block fill kernel whose own
        # run time (~1.5 us) is negligible -- marker packets, the dispatch gap behind a marker.  The kernels of this path
        # run 10-30 us, so the bracket is not small against them; the roofline uses durations net of it (and the
        # rocprofv3 kernel trace, which has no brackets, is the cross-check: profiles/).
        prof, ops.PROFILER = ops.PROFILER, None
        tiny = torch.zeros(64, device=device)
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
        for e0, e1 in pairs:
            e0.record(); tiny.fill_(1.0); e1.record()
        torch.cuda.synchronize()
        ov = sorted(e0.elapsed_time(e1) for e0, e1 in pairs)
        event_overhead_ms = max(0.0, ov[len(ov) // 2] - 0.0015)
        barrier()
    if rank == 0 and args.profile_steps > 0:
        if os.environ.get("STEGAN_BENCH_DETAIL"):
            by = ops.profiler_by_shape(prof)
            print("mode B Cin Cout K s d g up Tin Tout | launches/step ms/step TFLOP/s", file=sys.stderr)
            for shape, (n, ms, macs) in sorted(by.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("STEGAN_BENCH_DETAIL_N", "40"))]:
                print(shape, "|", n // args.profile_steps, round(ms / args.profile_steps, 3),
                      round(2 * macs / (ms * 1e-3) / 1e12, 1), file=sys.stderr)
        for v in agg.values():
            v["raw_ms"] = v["ms"]
            if v["kernel_timed"] == v["launches"]:          # the dispatches' own begin / end timestamps: no bracket inside
                v["ms"], v["timing"] = v["kernel_ms"], "kernel timestamps (hipExtLaunchKernelGGL start / stop events)"
            else:                                          # net of the event bracket (see above)
                v["ms"], v["timing"] = max(v["ms"] - event_overhead_ms * v["launches"], 0.25 * v["ms"]), "event bracket, net"
        # the dominant family = the one with the most kernel time per step.  (Chosen on the corrected durations: on the raw
        # event brackets -- ~6 us of bracket per launch -- a family of many short launches could outrank the one that
        # actually holds the GPU longest, and the choice flipped between runs.)
        name, a = max(agg.items(), key=lambda kv: kv[1]["ms"])
        raw_avg_ms = a["raw_ms"] / a["launches"]
        avg_ms = a["ms"] / a["launches"]
        flops_per_launch = 2.0 * a["nominal_macs"] / a["launches"]
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
        all_ms = sum(v["ms"] for v in agg.values())
        all_fl = 2.0 * sum(v["nominal_macs"] for v in agg.values())
        # HBM-side bytes and matrix-pipe busy fraction from the committed PMC passes over this same command (bench.py cannot
        # run rocprofv3 on itself): the newest profiles/r*_pmc_summary.json measured on THESE kernel sources (load_pmc; null
        # when the tree has moved on), written by scripts/pmc_summary.py from the raw per-dispatch rows kept beside it.  traffic = the kernel family's average per launch in that run; traffic_over_algorithmic compares
        # per-STEP totals (the family's PMC bytes per step / the algorithmic bytes of the same launches counted live here), so
        # it does not depend on how many problems a multi-problem launch carried.
        traffic = mfma_busy = over = None
        fam_key = "conv_wgrad_bf16" if "wgrad_bf16" in name else name.split("<")[0]
        fam_live = [v for k, v in agg.items() if ("wgrad_bf16" in k) == ("wgrad_bf16" in name) and k.split("<")[0] == name.split("<")[0]]
        try:
            fam = PMC["families"][fam_key]
            traffic = fam["hbm_bytes_per_launch"]
            mfma_busy = fam["mfma_busy_frac"]
            alg_step = sum(v["alg_bytes"] for v in fam_live) / args.profile_steps
            over = round(fam["hbm_bytes_per_step"] / alg_step, 2) if alg_step else None
        except (KeyError, TypeError):
            pass
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        roof = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 3), "peak": peak,
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                "traffic_over_algorithmic": over, "mfma_busy_frac": mfma_busy,
                "algorithmic_bytes_per_launch": int(a["alg_bytes"] / a["launches"]),
                "counters": ({"file": PMC.get("_file"), "csrc_sha": PMC.get("csrc_sha"), "measured_utc": PMC.get("measured_utc"),
                              "how": "rocprofv3 --pmc passes over this command (scripts/pmc_summary.py); raw rows beside the file"}
                             if PMC else PMC_STALE),
                "launches_per_step": a["launches"] // args.profile_steps, "avg_launch_us": round(avg_ms * 1e3, 2),
                "timing": a["timing"],
                "avg_launch_us_with_event_bracket": round(raw_avg_ms * 1e3, 2),
                "event_bracket_us": round(event_overhead_ms * 1e3, 2),
                "algorithmic_gflop_per_launch": round(flops_per_launch / 1e9, 4),
                "live_tap_frac": round(a["live_macs"] / a["nominal_macs"], 4),
                "all_conv_kernels": {"achieved": round(all_fl / (all_ms * 1e-3) / 1e12 / 1.0, 3),
                                     "ms_per_step": round(all_ms / args.profile_steps, 3),
                                     "per_kernel_ms_per_step": {k: round(v["ms"] / args.profile_steps, 3) for k, v in agg.items()}}}

    # the same step in the fp16 operand mode (the other build of the kernels + the device-side GradScaler: BASELINE config 5's
    # arithmetic on the metric's workload), with the headline's protocol (batch resident in the graph's static buffers).  A step
    # of its own: the scaler belongs to a step constructed in that mode.  (Measured BEFORE the fp32 pass: 1.7 s of dense fp32
    # MFMA work in front of it cost the fp16 number 6 % -- 6.47 against 6.10 ms -- on the same box.)
    f16_path = None
    if world == 1 and args.dtype == "bf16" and not args.no_f32:
        ops.set_compute_dtype("f16")
        try:
            from ste_gan_amd.train import GraphedTrainStep
            e16 = build_step(cfg, device)
            st16 = GraphedTrainStep(e16, batch, warmup=2) if graphed else e16
            if graphed:
                st16.load_batch(batch)
            arg16 = None if graphed else batch
            for _ in range(4):
                st16(arg16)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n16 = max(5, args.steps // 2)
            for _ in range(n16):
                out16 = st16(arg16)
            torch.cuda.synchronize()
            d16 = (time.perf_counter() - t1) / n16
            f16_path = {"dtype": "f16", "ms_per_step": round(1e3 * d16, 3), "value": round(B * T_EMG / d16, 1),
                        "unit": "EMG frames/s", "steps": n16, "loss_scale": float(out16["loss_scale"].item()),
                        "skipped_optimizer_steps": e16.scaler.skipped_steps(),
                        "note": "libstegan_hip_f16.so (same sources, fp16 operands) + optim.GradScaler on the device"}
            if graphed:
                st16.close()
            del st16, e16
        finally:
            ops.set_compute_dtype(args.dtype)
    # ... and on the fp32 parity path (exact fp32 MFMA), measured in the same process
    f32_path = None
    if world == 1 and args.dtype == "bf16" and not args.no_f32:
        ops.set_compute_dtype("f32")
        from ste_gan_amd.train import GraphedTrainStep
        st32 = GraphedTrainStep(eager_step, batch, warmup=2) if graphed else eager_step
        for _ in range(2):
            st32(batch)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n32 = max(5, args.steps // 2)
        for _ in range(n32):
            st32(batch)
        torch.cuda.synchronize()
        d32 = (time.perf_counter() - t1) / n32
        f32_path = {"dtype": "f32", "ms_per_step": round(1e3 * d32, 3), "value": round(B * T_EMG / d32, 1), "unit": "EMG frames/s",
                    "steps": n32, "note": "exact fp32 MFMA path, the one the parity tests pin to the reference within 1e-4"}
        ops.set_compute_dtype(args.dtype)
    gi = gi32 = None
    if rank == 0 and world == 1:             # BASELINE config 2 beside the headline number (same process, after the timed region;
        g = gen_infer(args, cfg, device, quiet=True)     # BEFORE the CPU baseline: its 128 worker threads keep spinning for a
        gi = {"value": g["value"], "unit": g["unit"], "ms_per_batch": g["ms_per_step"],      # while and slow this host-bound
              "workload": g["config"]["workload"]}                                           # eager loop 4x)
        if args.dtype != "f32":              # ... and in config 2's OWN dtype (fp32: the exact parity path)
            ops.set_compute_dtype("f32")
            try:
                g = gen_infer(args, cfg, device, quiet=True)
                gi32 = {"value": g["value"], "unit": g["unit"], "ms_per_batch": g["ms_per_step"], "dtype": "f32",
                        "workload": g["config"]["workload"],
                        "frac_of_fp32_mfma_bound": round(g["value"] / 18.2e6, 4)}        # 18.2 M frames/s
            finally:
                ops.set_compute_dtype(args.dtype)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        frames = world * B * T_EMG * args.steps
        line = {"metric": "EMG frames/sec (gen+disc fwd+bwd) at B=64 chunk=200", "value": round(frames / dt, 1),
                "unit": "EMG frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                "config": {"workload": "full GAN train step (G + small MPD/MSD D + multi-TD/SU/phoneme/FM losses + 2x AdamW), "
                                       "B=64/GPU, chunk=200 -> T_su=12, T_emg=192, " +
                                       ("bf16 conv operands, fp32 accumulation / activations / weights (BASELINE config 3)"
                                        if args.dtype == "bf16" else
                                        "fp16 conv operands + device-side GradScaler, fp32 accumulation / activations / weights"
                                        if args.dtype == "f16" else "fp32 parity path"),
                           "global_batch": world * B, "parallelism": f"dp{world}",
                           "launch": ("eager" if not graphed else "hipGraph replay" if world == 1 else
                                      "1 hipGraph per step, the 2 RCCL all-reduces captured inside it" if getattr(step, "ddp_mode", None) == "single"
                                      else "3 hipGraphs per step around 2 eager RCCL all-reduces")},
                "roofline": roof,
                # whole-step view against both roofs (SURVEY.md section 8d asks for both fractions and which one binds)
                "step_rooflines": step_rooflines(dt / args.steps, args.dtype, frames / dt),
                "fp32_parity_path": f32_path, "fp16_path": f16_path, "cpu_baseline": cpu, "generator_inference": gi,
                "generator_inference_f32": gi32, "env": env_seen,
                "last_step_losses": {k: round(float(v), 5) for k, v in stats.items() if isinstance(v, float)}}
        print(json.dumps(line), flush=True)
    if world > 1:
        # ordered teardown (ste_gan_amd.parallel.shutdown): every rank is done, drain the device, release the captured graphs
        # and the communication stream, then destroy the process group.  (Round 2 left through os._exit here: the abort it
        # dodged came from graph captures in the default "global" error mode racing the RCCL watchdog's event queries --
        # fixed by capture_error_mode="thread_local" in GraphedTrainStep, pinned by tests/test_host_logic.py.)
        from ste_gan_amd.parallel import shutdown
        dist.barrier()
        shutdown(step, eager_step)


if __name__ == "__main__":
    main()
