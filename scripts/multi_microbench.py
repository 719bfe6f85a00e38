"""Five period-discriminator problems of one layer: five launches vs one multi-problem launch."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ste_gan_amd import ops, _lib
B = 64
def problems(cin, cout, K, s, pad, Hs):
    out = []
    for p_, H in Hs:
        c = ops.ConvCfg(cin, cout, K, s, 1, pad, 1, 1, 0, 2)
        x = torch.randn(B * p_, cin, H, device="cuda"); w = torch.randn(cout, cin, K, device="cuda") * 0.05
        sc = torch.rand(cout, device="cuda") + 0.5; bias = torch.randn(cout, device="cuda")
        wf, wb = ops.prep_weights(w, c, None, sc)
        out.append((c, x, w, sc, bias, wf))
    return out
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
L = _lib.lib()
for name, args in [("32->256 k3 s3", (32, 256, 3, 3, 2, [(2, 96), (3, 64), (5, 39), (7, 28), (11, 18)])),
                   ("256->512 k3 s3", (256, 512, 3, 3, 2, [(2, 33), (3, 22), (5, 14), (7, 10), (11, 7)])),
                   ("512->1024?? n/a", None)]:
    if args is None: continue
    ps = problems(*args)
    ys = [torch.empty((x.shape[0], c.cout, c.tout(x.shape[2])), device="cuda") for c, x, *_ in ps]
    def separate():
        return [ops.conv1d_fwd(x, w, sc, b, None, c, wf) for c, x, w, sc, b, wf in ps]
    descs = [ops._desc(c, x.shape[0], x.shape[2], c.tout(x.shape[2])) for c, x, *_ in ps]
    n = len(ps)
    arr = lambda items: (C.c_void_p * n)(*[C.c_void_p(t.data_ptr()) if t is not None else None for t in items])
    dptr = (C.c_void_p * n)(*[C.cast(C.pointer(d), C.c_void_p) for d in descs])
    xa, wfa, sca, ba, ya = arr([p[1] for p in ps]), arr([p[5] for p in ps]), arr([p[3] for p in ps]), arr([p[4] for p in ps]), arr(ys)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def multi():
        rc = L.stegan_conv1d_fwd_multi(n, dptr, xa, wfa, sca, ba, None, ya, st)
        assert rc == 0, rc
    ref = separate(); multi(); torch.cuda.synchronize()
    err = max(float((a - b).abs().max()) for a, b in zip(ref, ys))
    print(f"{name}: separate {timeit(separate):7.1f} us   multi {timeit(multi):7.1f} us   max |diff| {err:.2e}")
