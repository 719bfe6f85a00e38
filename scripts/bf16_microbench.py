"""Per-layer timing of the bf16-operand conv kernels on the hot-path shapes, under the experiment overrides of
csrc/conv_bf16.hip (STEGAN_BF16_TILE / _CCH_MAX / _SPLITS / _NO_XV), against the fp32 direct kernels.

    python scripts/bf16_microbench.py [substring of layer name] [--variants="tile=22,cch=32;tile=21;..."]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ste_gan_amd import ops
B = int(os.environ.get("MB_B", "64"))
# (name, Cin, Cout, K, stride, dil, pad, groups, up, relu, Tin, Beff)
LAYERS = [
    ("G 768->768 k3 d1 T12", 768, 768, 3, 1, 1, 1, 1, 1, True, 12, B),
    ("G 768->768 k3 d27 T12", 768, 768, 3, 1, 27, 27, 1, 1, True, 12, B),
    ("G 768->768 k3 up T12->24", 768, 768, 3, 1, 1, 1, 1, 2, True, 12, B),
    ("G 768->768 k3 d3 T48", 768, 768, 3, 1, 3, 3, 1, 1, True, 48, B),
    ("G 768->768 k3 d1 T96", 768, 768, 3, 1, 1, 1, 1, 1, True, 96, B),
    ("G 384->384 k3 d9 T96", 384, 384, 3, 1, 9, 9, 1, 1, True, 96, B),
    ("G 192->192 k3 d1 T192", 192, 192, 3, 1, 1, 1, 1, 1, True, 192, B),
    ("G 192->192 k3 d27 T192", 192, 192, 3, 1, 27, 27, 1, 1, True, 192, B),
    ("G 384->192 k1 up T96", 384, 192, 1, 1, 1, 0, 1, 2, False, 96, B),
    ("D msd 512->1024 k5 T48", 512, 1024, 5, 1, 1, 2, 1, 1, False, 48, B),
    ("D msd 128->256 k37 s2 g4 T192", 128, 256, 37, 2, 1, 18, 4, 1, False, 192, B),
    ("D msd 256->512 k37 s2 g16 T96", 256, 512, 37, 2, 1, 18, 16, 1, False, 96, B),
    ("D msd 8->128 k15 T192", 8, 128, 15, 1, 1, 7, 1, 1, False, 192, B),
    ("D mpd2 32->256 k3 s3 H99", 32, 256, 3, 3, 1, 2, 1, 1, False, 99, B * 2),
    ("D mpd2 256->512 k3 s3 H34", 256, 512, 3, 3, 1, 2, 1, 1, False, 34, B * 2),
    ("D mpd11 256->512 k3 s3 H8", 256, 512, 3, 3, 1, 2, 1, 1, False, 8, B * 11),
    ("E enc 768->3072 k1 T12", 768, 3072, 1, 1, 1, 0, 1, 1, False, 12, B),
    ("E enc 3072->768 k1 T12", 3072, 768, 1, 1, 1, 0, 1, 1, False, 12, B),
    ("E enc 768->768 k1 T12", 768, 768, 1, 1, 1, 0, 1, 1, False, 12, B),
    ("E enc 768->768 k3 s2 T96", 768, 768, 3, 2, 1, 1, 1, 1, False, 96, B),
]
ENV = {"wr": "STEGAN_BF16_WGRAD_R", "wsl": "STEGAN_BF16_WGRAD_SLICES", "tile": "STEGAN_BF16_TILE", "cch": "STEGAN_BF16_CCH_MAX", "splits": "STEGAN_BF16_SPLITS", "noxv": "STEGAN_BF16_NO_XV", "dbg": "STEGAN_BF16_DBG"}


def timeit(fn, n=20):
    """us per call; the calls are captured into one hipGraph so that host overhead (ctypes, torch.empty: ~15 us per
    call) does not hide short kernels."""
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


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    only = args[0] if args else None
    variants = [""]
    for a in sys.argv[1:]:
        if a.startswith("--variants="):
            variants = a.split("=", 1)[1].split(";")
    print(f"{'layer':32s} {'GFLOP':>7s} | {'variant':18s} | {'f32 fwd':>8s} {'bf16 fwd':>8s} {'TF':>6s} | {'f32 bwd':>8s} {'bf16 bwd':>8s} {'TF':>6s}"
          f" | {'f32 wgr':>8s} {'bf16 wgr':>8s} {'TF':>6s}")
    for (name, Cin, Cout, K, s, d, p, g, up, relu, Tin, Be) in LAYERS:
        if only and only not in name: continue
        c = ops.ConvCfg(Cin, Cout, K, s, d, p, g, up, 1 if relu else 0)
        Tout = c.tout(Tin)
        x = torch.randn(Be, Cin, Tin, device="cuda"); w = torch.randn(Cout, Cin // g, K, device="cuda") * 0.05
        sc = torch.rand(Cout, device="cuda") + 0.5; bias = torch.randn(Cout, device="cuda")
        dy = torch.randn(Be, Cout, Tout, device="cuda")
        gf = 2.0 * Be * Tout * Cout * (Cin // g) * K / 1e9
        ops.set_compute_dtype("f32")
        wf, wb = ops.prep_weights(w, c, None, sc)
        t_f = timeit(lambda: ops.conv1d_fwd(x, w, sc, bias, None, c, wf))
        t_b = timeit(lambda: ops.conv1d_bwd_data(dy, w, sc, x if relu else None, None, c, Tin, wb))
        dw = torch.zeros_like(w); db = torch.zeros(Cout, device="cuda")
        t_w = timeit(lambda: ops.conv1d_bwd_weight(x, dy, c, dw, db))
        ops.set_compute_dtype("bf16")
        wf, wb = ops.prep_weights(w, c, None, sc)
        for v in variants:
            for k_ in ENV.values(): os.environ.pop(k_, None)
            for kv in filter(None, v.split(",")):
                k_, val = kv.split("=")
                os.environ[ENV[k_]] = val
            t_fh = timeit(lambda: ops.conv1d_fwd(x, w, sc, bias, None, c, wf))
            kf = ops.LAST_CONV_KERNEL
            t_bh = timeit(lambda: ops.conv1d_bwd_data(dy, w, sc, x if relu else None, None, c, Tin, wb))
            kb = ops.LAST_CONV_KERNEL
            t_wh = timeit(lambda: ops.conv1d_bwd_weight(x, dy, c, dw, db))
            kw = ops.LAST_CONV_KERNEL
            print(f"{name:32s} {gf:7.2f} | {v:18s} | {t_f:8.1f} {t_fh:8.1f} {gf/t_fh*1e3:6.0f}{'' if kf == 'bf16' else '*'} | "
                  f"{t_b:8.1f} {t_bh:8.1f} {gf/t_bh*1e3:6.0f}{'' if kb == 'bf16' else '*'} | "
                  f"{t_w:8.1f} {t_wh:8.1f} {gf/t_wh*1e3:6.0f}{'' if kw == 'bf16' else '*'}", flush=True)
        ops.set_compute_dtype("f32")


if __name__ == "__main__":
    main()
