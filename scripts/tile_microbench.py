"""Per-layer timing of the image-path forward / input-gradient convs (stride 1, ungrouped): conv_tile.hip against
conv_bf16.hip.  The choice is a process-wide option, so run it twice:

    STEGAN_TILE=0 python scripts/tile_microbench.py > old.txt ; STEGAN_TILE=1 python scripts/tile_microbench.py > new.txt
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ste_gan_amd import ops
from bf16_microbench import timeit

B = int(os.environ.get("MB_B", "64"))
# (name, Cin, Cout, K, dil, up, relu, Tin)
LAYERS = [
    ("G in 320->768 k1 T12", 320, 768, 1, 1, 1, False, 12),
    ("G 768->768 k3 d1 T12", 768, 768, 3, 1, 1, True, 12),
    ("G 768->768 k3 d3 T12", 768, 768, 3, 3, 1, True, 12),
    ("G 768->768 k3 d9 T12", 768, 768, 3, 9, 1, True, 12),
    ("G 768->768 k3 d27 T12", 768, 768, 3, 27, 1, True, 12),
    ("G 768->768 k1 T12", 768, 768, 1, 1, 1, False, 12),
    ("G 768->384 k3 up T12->24", 768, 384, 3, 1, 2, True, 12),
    ("G 768->384 k1 up T12->24", 768, 384, 1, 1, 2, False, 12),
    ("G 384->384 k3 d3 T24", 384, 384, 3, 3, 1, True, 24),
    ("G 384->384 k3 up T24->48", 384, 384, 3, 1, 2, True, 24),
    ("G 384->384 k3 d9 T48", 384, 384, 3, 9, 1, True, 48),
    ("G 384->384 k3 up T48->96", 384, 384, 3, 1, 2, True, 48),
    ("G 384->384 k3 d27 T96", 384, 384, 3, 27, 1, True, 96),
    ("G 384->192 k3 up T96->192", 384, 192, 3, 1, 2, True, 96),
    ("G 384->192 k1 up T96->192", 384, 192, 1, 1, 2, False, 96),
    ("G 192->192 k3 d1 T192", 192, 192, 3, 1, 1, True, 192),
    ("G 192->192 k3 d27 T192", 192, 192, 3, 27, 1, True, 192),
    ("G 192->192 k1 T192", 192, 192, 1, 1, 1, False, 192),
    ("E 768->768 k3 T96", 768, 768, 3, 1, 1, False, 96),
    ("E 768->768 k3 T48", 768, 768, 3, 1, 1, False, 48),
    ("E 768->768 k3 T24", 768, 768, 3, 1, 1, False, 24),
    ("E 768->2304 k1 T12", 768, 2304, 1, 1, 1, False, 12),
    ("E 768->3072 k1 T12", 768, 3072, 1, 1, 1, False, 12),
    ("E 3072->768 k1 T12", 3072, 768, 1, 1, 1, False, 12),
]


def image(t):
    Bq, C, T = t.shape
    return t.view(Bq, C // 8, 8, T).permute(0, 1, 3, 2).contiguous().to(ops.HALF_DTYPE)


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None
    ops.set_compute_dtype(os.environ.get("MB_DTYPE", "bf16"))
    print(f"# STEGAN_TILE={os.environ.get('STEGAN_TILE', '1')} B={B}")
    print(f"{'layer':28s} {'GFLOP':>7s} | {'fwd us':>8s} {'TF':>6s} {'kern':>6s} | {'bwd us':>8s} {'TF':>6s}")
    tot_f = tot_b = 0.0
    for (name, Cin, Cout, K, d, up, relu, Tin) in LAYERS:
        if only and only not in name:
            continue
        pad = d * (K - 1) // 2
        c = ops.ConvCfg(Cin, Cout, K, 1, d, pad, 1, up, 1 if relu else 0)
        Tout = c.tout(Tin)
        x = torch.randn(B, Cin, Tin, device="cuda"); w = torch.randn(Cout, Cin, K, device="cuda") * 0.05
        sc = torch.rand(Cout, device="cuda") + 0.5; bias = torch.randn(Cout, device="cuda")
        dy = torch.randn(B, Cout, Tout, device="cuda")
        ops._attach_image(x, image(x)); ops._attach_image(dy, image(dy))
        gf = 2.0 * B * Tout * Cout * Cin * K / 1e9
        wf, wb = ops.prep_weights(w, c, None, sc)
        t_f = timeit(lambda: ops.conv1d_fwd(x, w, sc, bias, None, c, wf))
        kf = ops.LAST_CONV_KERNEL
        t_b = timeit(lambda: ops.conv1d_bwd_data(dy, w, sc, x if relu else None, None, c, Tin, wb))
        tot_f += t_f; tot_b += t_b
        print(f"{name:28s} {gf:7.2f} | {t_f:8.1f} {gf / t_f * 1e3:6.0f} {kf:>6s} | {t_b:8.1f} {gf / t_b * 1e3:6.0f}", flush=True)
    print(f"# sum fwd {tot_f:.1f} us, bwd {tot_b:.1f} us")


if __name__ == "__main__":
    main()
