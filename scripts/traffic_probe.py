"""Fabric-side read traffic and run time of single bf16 conv launches against the block -> XCD placement (STEGAN_BF16_XLGM).
    python3 scripts/traffic_probe.py time                     (plain timing, us per launch)
    rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d DIR -- python3 scripts/traffic_probe.py run order.json
    python3 scripts/traffic_probe.py join order.json DIR [DIR ...]
"""
import collections, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

B = 64
LAYERS = [
    ("G 768->768 k3 d1 T12", 768, 768, 3, 1, 1, 1, 1, 1, True, 12),
    ("E 768->768 k1 T12", 768, 768, 1, 1, 1, 0, 1, 1, False, 12),
    ("E 768->3072 k1 T12", 768, 3072, 1, 1, 1, 0, 1, 1, False, 12),
    ("E 3072->768 k1 T12", 3072, 768, 1, 1, 1, 0, 1, 1, True, 12),
    ("G 768->768 k3 d1 T96", 768, 768, 3, 1, 1, 1, 1, 1, True, 96),
    ("G 384->384 k3 d3 T96", 384, 384, 3, 1, 3, 3, 1, 1, True, 96),
    ("G 192->192 k3 d3 T192", 192, 192, 3, 1, 3, 3, 1, 1, True, 192),
    ("D 512->1024 k5 T48", 512, 1024, 5, 1, 1, 2, 1, 1, False, 48),
]
MODES = ("model", "-1", "0", "1", "2", "3")


def setup():
    import torch
    from ste_gan_amd import ops
    from layer_table_util import make_image
    ops.set_compute_dtype("bf16")
    out = []
    for (name, Cin, Cout, K, s, d, p, g, up, relu, Tin) in LAYERS:
        c = ops.ConvCfg(Cin, Cout, K, s, d, p, g, up, 1 if relu else 0)
        x = torch.randn(B, Cin, Tin, device="cuda"); w = torch.randn(Cout, Cin // g, K, device="cuda") * 0.05
        sc = torch.rand(Cout, device="cuda") + 0.5; bias = torch.randn(Cout, device="cuda")
        ops._attach_image(x, make_image(x))
        wf, wb = ops.prep_weights(w, c, None, sc)
        out.append((name, lambda x=x, w=w, sc=sc, bias=bias, c=c, wf=wf: ops.conv1d_fwd(x, w, sc, bias, None, c, wf),
                    (2 * x.numel() + 2 * w.numel(), 6 * B * Cout * c.tout(Tin))))
    return out


def main():
    what = sys.argv[1]
    if what == "join":
        from pmc_summary import load
        order = json.load(open(sys.argv[2]))
        res = collections.defaultdict(lambda: collections.defaultdict(float))
        for d in sys.argv[3:]:
            rows = [r for r in load(d)[1] if "conv_bf16_kernel<" in r["Kernel_Name"]]
            ids = sorted({int(r["Dispatch_Id"]) for r in rows})
            assert len(ids) == len(order), (len(ids), len(order))
            pos = {i: k for k, i in enumerate(ids)}
            for r in rows:
                name, mode, alg = order[pos[int(r["Dispatch_Id"])]]
                res[(name, mode)][r["Counter_Name"]] += float(r["Counter_Value"]) / 3.0
                res[(name, mode)]["alg_r"], res[(name, mode)]["alg_w"] = alg
                res[(name, mode)]["grid"] = int(r["Grid_Size"]) // 512
        for (name, mode), c in res.items():
            extra = "  ".join(f"{k} {v:12.0f}" for k, v in c.items() if k not in ("alg_r", "alg_w", "grid", "FETCH_SIZE"))
            print(f"{name:24s} xlgm {mode:>5s} blocks {c['grid']:4.0f} alg read {c['alg_r']/1e6:6.2f} MB | FETCH x2 "
                  f"{2*c.get('FETCH_SIZE', 0)*1024/1e6:7.2f} MB | {extra}")
        return
    import torch
    from layer_table_util import timeit
    layers = setup()
    order = []
    for name, fn, alg in layers:
        for m in MODES:
            os.environ.pop("STEGAN_BF16_XLGM", None)
            if m != "model":
                os.environ["STEGAN_BF16_XLGM"] = m
            if what == "time":
                print(f"{name:24s} xlgm {m:>5s} {timeit(fn):7.2f} us", flush=True)
            else:
                for _ in range(3):
                    fn()
                    order.append((name, m, alg))
                torch.cuda.synchronize()
    if what == "run":
        json.dump(order, open(sys.argv[2], "w"))


if __name__ == "__main__":
    main()
