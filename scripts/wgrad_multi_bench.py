"""A generator block's five weight gradients: chained per-layer launches (ops.WGRAD_CHAIN) against ONE multi-layer launch of the
tile kernel (ops._wgrad_multi, stegan_conv1d_bwd_weight_multi_hxc).  STEGAN_PLAN_DEBUG=1 prints the plans."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ste_gan_amd import ops
from tile_microbench import image
from bf16_microbench import timeit
ops.set_compute_dtype("bf16")
B = 64
BLOCKS = {"384 T96": [(384, 384, 3, 1, 2, 1, 48), (384, 384, 3, 3, 1, 1, 96), (384, 384, 3, 9, 1, 1, 96), (384, 384, 3, 27, 1, 1, 96), (384, 384, 1, 1, 2, 0, 48)],
          "192 T192": [(192, 192, 3, 1, 1, 1, 192), (192, 192, 3, 3, 1, 1, 192), (192, 192, 3, 9, 1, 1, 192), (192, 192, 3, 27, 1, 1, 192), (192, 192, 1, 1, 1, 0, 192)],
          "768 T12": [(768, 768, 3, 1, 1, 1, 12), (768, 768, 3, 3, 1, 1, 12), (768, 768, 3, 9, 1, 1, 12), (768, 768, 3, 27, 1, 1, 12), (768, 768, 1, 1, 1, 0, 12)]}
for name, layers in BLOCKS.items():
    data = []
    for (Cin, Cout, K, d, up, relu, Tin) in layers:
        c = ops.ConvCfg(Cin, Cout, K, 1, d, d * (K - 1) // 2, 1, up, relu)
        x = torch.randn(B, Cin, Tin, device="cuda"); dy = torch.randn(B, Cout, c.tout(Tin), device="cuda")
        ops._attach_image(x, image(x)); ops._attach_image(dy, image(dy))
        data.append((x, dy, c, torch.zeros(Cout, Cin, K, device="cuda"), torch.zeros(Cout, device="cuda")))
    ok = ops._wgrad_multi(data); ops.wgrad_join()
    def chained():
        chain = [None]
        for x, dy, c, dw, db in data:
            ops.conv1d_bwd_weight(x, dy, c, dw, db, defer=True, chain=chain)
        ops.wgrad_chain_end(chain); ops.wgrad_join()
    def multi():
        ops._wgrad_multi(data); ops.wgrad_join()
    print(f"{name:10s} multi taken {ok}   chained {timeit(chained):7.1f} us   multi {timeit(multi):7.1f} us", flush=True)
