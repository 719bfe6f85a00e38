"""Tile / chunk / split-K sweep of the bf16 conv kernels (forward and input gradient) on the small-N layers, sources handed
over as bf16 images: which plan is fastest per layer, and what the cost model of conv_bf16.hip picks (variant "model").
    python scripts/plan_sweep.py [substring] > sweep.txt"""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ste_gan_amd import ops
from layer_table_util import make_image, timeit

B = 64
# (name, Cin, Cout, K, stride, dil, pad, groups, up, relu, Tin)
LAYERS = [
    ("E 768->3072 k1 T12", 768, 3072, 1, 1, 1, 0, 1, 1, False, 12),
    ("E 3072->768 k1 T12", 3072, 768, 1, 1, 1, 0, 1, 1, True, 12),
    ("E 768->768 k1 T12", 768, 768, 1, 1, 1, 0, 1, 1, False, 12),
    ("E 768->2304 k1 T12", 768, 2304, 1, 1, 1, 0, 1, 1, False, 12),
    ("G 768->768 k3 d1 T12", 768, 768, 3, 1, 1, 1, 1, 1, True, 12),
    ("G 768->768 k3 d9 T12", 768, 768, 3, 1, 9, 9, 1, 1, True, 12),
    ("G 768->768 k3 up T12", 768, 768, 3, 1, 1, 1, 1, 2, True, 12),
    ("G 768->768 k3 d3 T24", 768, 768, 3, 1, 3, 3, 1, 1, True, 24),
    ("G 384->384 k3 d3 T48", 384, 384, 3, 1, 3, 3, 1, 1, True, 48),
    ("D 512->1024 k5 T48", 512, 1024, 5, 1, 1, 2, 1, 1, False, 48),
    ("G 192->192 k3 d9 T192", 192, 192, 3, 1, 9, 9, 1, 1, True, 192),
]
ENV = {"tile": "STEGAN_BF16_TILE", "cch": "STEGAN_BF16_CCH_MAX", "splits": "STEGAN_BF16_SPLITS"}


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None
    ops.set_compute_dtype("bf16")
    variants = [dict()] + [dict(tile=t, cch=c, splits=s) for t in ("11", "12", "21", "22") for c in (32, 64)
                           for s in (1, 2, 3, 4, 6, 8, 12, 16)]
    for (name, Cin, Cout, K, s, d, p, g, up, relu, Tin) in LAYERS:
        if only and only not in name:
            continue
        c = ops.ConvCfg(Cin, Cout, K, s, d, p, g, up, 1 if relu else 0)
        Tout = c.tout(Tin)
        x = torch.randn(B, Cin, Tin, device="cuda"); w = torch.randn(Cout, Cin // g, K, device="cuda") * 0.05
        sc = torch.rand(Cout, device="cuda") + 0.5; bias = torch.randn(Cout, device="cuda")
        dy = torch.randn(B, Cout, Tout, device="cuda")
        ops._attach_image(x, make_image(x)); ops._attach_image(dy, make_image(dy))
        wf, wb = ops.prep_weights(w, c, None, sc)
        gf = 2.0 * B * Tout * Cout * (Cin // g) * K / 1e9
        res = {0: [], 1: []}
        for v in variants:
            for k_ in ENV.values():
                os.environ.pop(k_, None)
            for k_, val in v.items():
                os.environ[ENV[k_]] = str(val)
            try:
                tf = timeit(lambda: ops.conv1d_fwd(x, w, sc, bias, None, c, wf))
                tb = timeit(lambda: ops.conv1d_bwd_data(dy, w, sc, x if relu else None, None, c, Tin, wb))
            except Exception as e:      # noqa: BLE001
                continue
            tag = "model" if not v else f"t{v['tile']} c{v['cch']} s{v['splits']}"
            res[0].append((tf, tag)); res[1].append((tb, tag))
        for mode in (0, 1):
            model = [t for t, tag in res[mode] if tag == "model"][0]
            best = sorted(res[mode])[:4]
            print(f"{name:24s} mode {mode} {gf:6.2f} GF | model {model:6.1f} us | best " +
                  "  ".join(f"{t:5.1f} ({tag})" for t, tag in best), flush=True)
    for k_ in ENV.values():
        os.environ.pop(k_, None)


if __name__ == "__main__":
    main()
