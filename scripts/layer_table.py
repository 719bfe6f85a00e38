"""Cost table of the train step, layer by layer: every distinct conv launch of one step (mode, shape; collected by the
instrumented eager pass of ops.ConvProfiler) timed IN ISOLATION -- 20 back-to-back launches captured into one hipGraph, so
neither the host nor an event bracket is in the number -- and listed with its launches per step, sorted by launches x time.
    python scripts/layer_table.py [bf16|f32] > table.txt
What the list is for: ranking where the serialised kernel time of a step goes (which T / channel groups, which mode)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ste_gan_amd import ops

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
ops.set_compute_dtype(dtype)
dev = torch.device("cuda", 0)
cfg = bench.make_cfg("train_step")
step = bench.build_step(cfg, dev)
batch = bench.synthetic_batch(dev, seed=1000)
step(batch); step(batch)
torch.cuda.synchronize()
ops.PROFILER = ops.ConvProfiler()
step.overlap_encoder = False
os.environ["STEGAN_D_STREAMS"] = "0"
step(batch)
torch.cuda.synchronize()
recs, ops.PROFILER = ops.PROFILER.records, None
shapes = {}
for name, nominal, live, e0, e1, shape in recs:
    a = shapes.setdefault(shape, [0, nominal, name])
    a[0] += 1


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n): fn()
    torch.cuda.current_stream().wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * n) * 1e3


def make_image(t):
    B, C, T = t.shape
    CB = (C + 7) // 8
    full = torch.zeros(B, CB * 8, T, device=t.device, dtype=t.dtype)
    full[:, :C] = t
    return full.view(B, CB, 8, T).permute(0, 1, 3, 2).contiguous().to(torch.bfloat16)


rows = []
only_mode = os.environ.get("LAYER_MODE")
with_images = os.environ.get("LAYER_IMAGES", "1") != "0"
for shape, (count, nominal, name) in shapes.items():
    mode, B, Cin, Cout, K, s, d, g, up, Tin, Tout, pad, in_act, out_act = shape
    if only_mode is not None and int(only_mode) != mode:
        continue
    c = ops.ConvCfg(Cin, Cout, K, s, d, pad, g, up, in_act, out_act)
    x = torch.randn(B, Cin, Tin, device=dev); w = torch.randn(Cout, Cin // g, K, device=dev) * 0.05
    sc = torch.rand(Cout, device=dev) + 0.5; bias = torch.randn(Cout, device=dev)
    dy = torch.randn(B, Cout, Tout, device=dev)
    if with_images and Cin >= 16:
        x.h = make_image(x)
    if with_images and Cout >= 16:
        dy.h = make_image(dy)
    wf, wb = ops.prep_weights(w, c, None, sc)
    try:
        if mode == 0:
            t = timeit(lambda: ops.conv1d_fwd(x, w, sc, bias, None, c, wf))
        elif mode == 1:
            t = timeit(lambda: ops.conv1d_bwd_data(dy, w, sc, x if in_act == 1 else None, None, c, Tin, wb))
        else:
            dw = torch.zeros_like(w); db = torch.zeros(Cout, device=dev)
            t = timeit(lambda: ops.conv1d_bwd_weight(x, dy, c, dw, db))
    except Exception as e:          # noqa: BLE001
        print("skip", shape, e, file=sys.stderr)
        continue
    rows.append((count * t, count, t, 2 * nominal / 1e9, shape, ops.LAST_CONV_KERNEL))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"# {dtype}: {len(rows)} distinct conv launches, {sum(r[1] for r in rows)} launches per step, {tot / 1e3:.3f} ms of isolated kernel time per step")
print("# us*n   n   us    GFLOP  TF/s | mode B Cin Cout K s d g up Tin Tout pad in_act out_act | kernel")
for tt, n, t, gf, shape, kern in rows:
    print(f"{tt:8.1f} {n:3d} {t:7.1f} {gf:7.2f} {gf / t * 1e3:6.0f} | {' '.join(str(v) for v in shape)} | {kern}")
by_mode = {}
for tt, n, t, gf, shape, kern in rows:
    k = (shape[0], shape[10] if shape[0] != 1 else shape[9] * shape[8])
    a = by_mode.setdefault(k, [0.0, 0]); a[0] += tt; a[1] += n
print("# by (mode, output length): us per step, launches")
for k, (tt, n) in sorted(by_mode.items(), key=lambda kv: -kv[1][0]):
    print(f"#   mode {k[0]} T {k[1]:4d}: {tt:8.1f} us  {n:3d} launches")
