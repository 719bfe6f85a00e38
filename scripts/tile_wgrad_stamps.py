"""In-kernel time stamps of conv_tile_wgrad_kernel (diagnostic build: scripts/build_tile_stamps.sh):

    STEGAN_HIP_LIB=scripts/micro/libstegan_tile_stamps.so python scripts/tile_wgrad_stamps.py [layer substring]

Consumer wave 0: 0 entry, 1 first chunk visible, 2 chunk loop done, 3 LDS reduction done, 4 bias done, 5 stores drained.
Producer wave 0: 6 set-up done, 7 last issue.  Medians over the tile blocks, in cycles; span from the 100 MHz real-time counter."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ste_gan_amd import ops, _lib
from tile_microbench import LAYERS, image, B
from bf16_microbench import timeit


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None
    ops.set_compute_dtype("bf16")
    L = _lib.lib()
    L.stegan_tile_wgrad_debug_stamps.argtypes = [C.c_void_p]
    buf = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda")
    cols = ["land0", "loop", "reduce", "bias", "store", "total", "p.setup", "p.loop"]
    print(f"{'layer':28s} | " + " ".join(f"{n:>8s}" for n in cols) + " | spread us | span us | blocks | op us (with reduce)")
    for (name, Cin, Cout, K, d, up, relu, Tin) in LAYERS:
        if only and only not in name:
            continue
        pad = d * (K - 1) // 2
        c = ops.ConvCfg(Cin, Cout, K, 1, d, pad, 1, up, 1 if relu else 0)
        Tout = c.tout(Tin)
        x = torch.randn(B, Cin, Tin, device="cuda"); dy = torch.randn(B, Cout, Tout, device="cuda")
        ops._attach_image(x, image(x)); ops._attach_image(dy, image(dy))
        dw = torch.zeros(Cout, Cin, K, device="cuda"); db = torch.zeros(Cout, device="cuda")
        for _ in range(3):
            ops.conv1d_bwd_weight(x, dy, c, dw, db)
        torch.cuda.synchronize()
        buf.zero_()
        L.stegan_tile_wgrad_debug_stamps(C.c_void_p(buf.data_ptr()))
        for _ in range(2):
            ops.conv1d_bwd_weight(x, dy, c, dw, db)
        torch.cuda.synchronize()
        L.stegan_tile_wgrad_debug_stamps(C.c_void_p(0))
        t = timeit(lambda: ops.conv1d_bwd_weight(x, dy, c, dw, db))
        s = buf.view(-1, 16).cpu()
        s = s[(s[:, 0] != 0) & (s[:, 5] != 0)]
        if len(s) == 0:
            print(f"{name:28s} | (not taken by the tile kernel)   op {t:.1f} us")
            continue
        m = lambda a, b: int((s[:, a] - s[:, b]).double().median())
        vals = [m(1, 0), m(2, 1), m(3, 2), m(4, 3), m(5, 4), m(5, 0), m(6, 0), m(7, 6)]
        rt0, rt1 = s[:, 8], s[:, 9]
        print(f"{name:28s} | " + " ".join(f"{v:8d}" for v in vals) + f" | {float(rt0.max() - rt0.min()) / 100.0:9.2f} | "
              f"{float(rt1.max() - rt0.min()) / 100.0:7.2f} | {len(s)} | {t:.1f}")


if __name__ == "__main__":
    main()
