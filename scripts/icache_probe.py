"""Does a conv launch pay for a cold instruction cache?  Two layers that run DIFFERENT template variants of conv_bf16_kernel
(64 x 64 tiles / 64 x 128 tiles), timed alone (the same kernel 20 times in a row: warm code) and alternating (A, B, A, B, ...:
every launch follows a launch of other code).  If alternating costs more than the mean of the two, the difference is code
fetch.    python scripts/icache_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from ste_gan_amd import ops
from layer_table_util import make_image, timeit
ops.set_compute_dtype("bf16")
B = 64
def layer(Cin, Cout, K, d, p, relu, Tin):
    c = ops.ConvCfg(Cin, Cout, K, 1, d, p, 1, 1, 1 if relu else 0)
    x = torch.randn(B, Cin, Tin, device="cuda"); w = torch.randn(Cout, Cin, K, device="cuda") * 0.05
    sc = torch.rand(Cout, device="cuda") + 0.5; bias = torch.randn(Cout, device="cuda")
    ops._attach_image(x, make_image(x))
    wf, wb = ops.prep_weights(w, c, None, sc)
    return lambda: ops.conv1d_fwd(x, w, sc, bias, None, c, wf)
a = layer(768, 768, 1, 1, 0, False, 12)          # <1, 1, 3>
b = layer(192, 192, 3, 9, 9, True, 192)          # <1, 2, 2>
c = layer(768, 3072, 1, 1, 0, False, 12)         # <2, 2, 3>
d = layer(512, 1024, 5, 1, 2, False, 48)         # another tile / chunk variant
e = layer(384, 384, 3, 3, 3, True, 48)
f = layer(768, 768, 3, 1, 1, True, 12)
ta, tb, tc = timeit(a), timeit(b), timeit(c)
td, te, tf = timeit(d), timeit(e), timeit(f)
def six():
    a(); b(); c(); d(); e(); f()
t6 = timeit(six, n=5) / 6
print(f"six layers alone: {ta:.2f} {tb:.2f} {tc:.2f} {td:.2f} {te:.2f} {tf:.2f} us (mean {(ta + tb + tc + td + te + tf) / 6:.2f}); in turn: {t6:.2f} us per launch")
print("kernels:", [ops.LAST_CONV_KERNEL])
def abc():
    a(); b(); c()
def ab():
    a(); b()
tab, tabc = timeit(ab) / 2, timeit(abc) / 3
print(f"alone: A {ta:.2f} us, B {tb:.2f} us, C {tc:.2f} us; alternating A,B: {tab:.2f} us per launch (mean alone {(ta + tb) / 2:.2f}); "
      f"A,B,C: {tabc:.2f} us per launch (mean alone {(ta + tb + tc) / 3:.2f})")
