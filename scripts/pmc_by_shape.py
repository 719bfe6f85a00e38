"""HBM-side traffic of every conv launch of one train step, by layer shape (traffic against algorithmic bytes).

Stage 1, on the GPU under rocprofv3 (one pass per counter, --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE --output-format csv):
    python3 scripts/pmc_by_shape.py run <records.json>
  two eager warm-up steps, then ONE step with ops.PROFILER on (per-conv launches, one stream, as bench.py's instrumented
  pass); the conv records (kernel family + shape) of that step are written in launch order.
Stage 2, offline:  python3 scripts/pmc_by_shape.py join <records.json> <fetch csv(.gz)|dir> <write csv(.gz)|dir> [out.json]
  the last len(records) dispatches of each kernel family are that step's launches, in the same order.
Algorithmic bytes of a launch: operands once (bf16 image 2 B / element, weights 2 B) + outputs once (fp32 4 B + bf16 image 2 B).
"""
import collections, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(out):
    import torch
    import bench
    from ste_gan_amd import ops
    ops.set_compute_dtype("bf16")
    dev = torch.device("cuda", 0)
    step = bench.build_step(bench.make_cfg("train_step"), dev)
    batch = bench.synthetic_batch(dev, 1000)
    step.overlap_encoder = False
    os.environ["STEGAN_D_STREAMS"] = "0"
    for _ in range(2):
        step(batch)
    torch.cuda.synchronize()
    ops.PROFILER = ops.ConvProfiler()
    step(batch)
    torch.cuda.synchronize()
    recs = [(r[0], list(r[5])) for r in ops.PROFILER.records]
    ops.PROFILER = None
    json.dump(recs, open(out, "w"))
    print(len(recs), "conv records")


def alg_bytes(name, s):
    mode, B, Cin, Cout, K, stride, dil, groups, up, Tin, Tout = s[:11]
    w = Cout * (Cin // groups) * K
    x, y = B * Cin * Tin, B * Cout * Tout
    if mode == 0:
        return 2 * x + 2 * w, 6 * y
    if mode == 1:                                  # (+ the fp32 mask source of a fused activation derivative)
        return 2 * y + 2 * w + (4 * x if len(s) > 12 and s[12] else 0), 6 * x
    return 2 * x + 2 * y, 4 * w


def join(recs_path, fdir, wdir, out=None):
    from pmc_summary import load
    recs = json.load(open(recs_path))
    fam_of = lambda n: ("conv_tile_wgrad_kernel" if n.startswith("conv_tile_wgrad") else "conv_tile_kernel" if n.startswith("conv_tile_kernel")
                        else "conv_wgrad_bf16" if "wgrad_bf16" in n else "conv_bf16_kernel" if n.startswith("conv_bf16") else None)
    want = collections.defaultdict(list)
    for name, shape in recs:
        if fam_of(name):
            want[fam_of(name)].append((name, shape))
    table = {}
    for fam, rl in want.items():
        keys = {"conv_tile_kernel": ("conv_tile_kernel(",), "conv_tile_wgrad_kernel": ("conv_tile_wgrad_kernel(",),
                "conv_bf16_kernel": ("conv_bf16_kernel<",), "conv_wgrad_bf16": ("conv_wgrad_bf16_img_kernel", "conv_wgrad_bf16_one_kernel",
                                                                               "conv_wgrad_bf16_kernel")}[fam]
        cols = []
        for d in (fdir, wdir):
            rows = [r for r in load(d)[1] if any(k in r["Kernel_Name"] for k in keys)]
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
            cols.append(rows[-len(rl):])
        agg = collections.OrderedDict()
        for (name, shape), rf, rw in zip(rl, *cols):
            a = agg.setdefault((name, tuple(shape[:14])), [0, 0.0, 0.0, rf["Kernel_Name"].split("(")[0][-40:], rf["Grid_Size"]])
            a[0] += 1
            a[1] += 2.0 * float(rf["Counter_Value"]) * 1024
            a[2] += float(rw["Counter_Value"]) * 1024
        table[fam] = agg
    res = []
    for fam, agg in table.items():
        print(f"== {fam}: mode B Cin Cout K s d g up Tin Tout | n | alg read / write MB | measured read / write MB | ratio  | kernel, grid")
        tr = tw = ar = aw = 0.0
        for (name, s), (n, f, w, kn, grid) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
            r_alg, w_alg = alg_bytes(name, s)
            tr += f; tw += w; ar += r_alg * n; aw += w_alg * n
            print(f"{s} | {n:3d} | {r_alg/1e6:7.2f} {w_alg/1e6:7.2f} | {f/n/1e6:7.2f} {w/n/1e6:7.2f} | {(f+w)/n/(r_alg+w_alg):5.2f} | {kn} {grid}")
            res.append({"family": fam, "shape": list(s), "launches": n, "alg_read": r_alg, "alg_write": w_alg,
                        "read": f / n, "write": w / n, "kernel": kn})
        print(f"   total per step: read {tr/1e9:.3f} GB (alg {ar/1e9:.3f}), write {tw/1e9:.3f} GB (alg {aw/1e9:.3f}), ratio {(tr+tw)/(ar+aw):.2f}")
    if out:
        json.dump(res, open(out, "w"), indent=0)


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        join(*sys.argv[2:])
