"""Per-layer timing of the conv kernels (generic implicit GEMM vs direct) on the hot-path shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ste_gan_amd import ops
B = int(os.environ.get("MB_B", "64"))
# (name, Cin, Cout, K, stride, dil, pad, groups, up, relu, Tin, Beff)
LAYERS = [
    ("G 768->768 k3 d1 T12", 768, 768, 3, 1, 1, 1, 1, 1, True, 12, B),
    ("G 768->768 k3 d27 T12", 768, 768, 3, 1, 27, 27, 1, 1, True, 12, B),
    ("G 768->384 k3 up T12->24", 768, 384, 3, 1, 1, 1, 1, 2, True, 12, B),
    ("G 384->384 k3 d3 T48", 384, 384, 3, 1, 3, 3, 1, 1, True, 48, B),
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
    ("E enc 768->768 k3 s2 T96", 768, 768, 3, 2, 1, 1, 1, 1, False, 96, B),
]
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
only = sys.argv[1] if len(sys.argv) > 1 else None
print(f"{'layer':34s} {'GFLOP':>7s} | {'fwd gen':>8s} {'fwd dir':>8s} {'TF dir':>7s} | {'bwd gen':>8s} {'bwd dir':>8s} {'TF dir':>7s} | {'bww':>8s} {'TF':>6s}")
for (name, Cin, Cout, K, s, d, p, g, up, relu, Tin, Be) in LAYERS:
    if only and only not in name: continue
    c = ops.ConvCfg(Cin, Cout, K, s, d, p, g, up, 1 if relu else 0)
    Tout = c.tout(Tin)
    x = torch.randn(Be, Cin, Tin, device="cuda"); w = torch.randn(Cout, Cin // g, K, device="cuda") * 0.05
    sc = torch.rand(Cout, device="cuda") + 0.5; bias = torch.randn(Cout, device="cuda")
    dy = torch.randn(Be, Cout, Tout, device="cuda")
    wf, wb = ops.prep_weights(w, c, None, sc)
    dw = torch.zeros_like(w); db = torch.zeros(Cout, device="cuda")
    gf = 2.0 * Be * Tout * Cout * (Cin // g) * K / 1e9
    t_fg = timeit(lambda: ops.conv1d_fwd(x, w, sc, bias, None, c))
    t_fd = timeit(lambda: ops.conv1d_fwd(x, w, sc, bias, None, c, wf))
    t_bg = timeit(lambda: ops.conv1d_bwd_data(dy, w, sc, x if relu else None, None, c, Tin))
    t_bd = timeit(lambda: ops.conv1d_bwd_data(dy, w, sc, x if relu else None, None, c, Tin, wb))
    t_w = timeit(lambda: ops.conv1d_bwd_weight(x, dy, c, dw, None))
    print(f"{name:34s} {gf:7.2f} | {t_fg:8.1f} {t_fd:8.1f} {gf/t_fd*1e3:7.1f} | {t_bg:8.1f} {t_bd:8.1f} {gf/t_bd*1e3:7.1f} | {t_w:8.1f} {gf/t_w*1e3:6.1f}")
