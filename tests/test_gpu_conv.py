"""-m gpu: the implicit-GEMM conv kernels (fwd / bwd-data / bwd-weight) through the C ABI
against a plain torch-CPU fp32 reference of the same op (F.conv1d + autograd)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (B, Cin, Cout, K, stride, dil, pad, groups, upsample, in_relu, Tin)  -- shapes of the hot path + edge cases
CASES = [
    (4, 320, 768, 1, 1, 1, 0, 1, 1, False, 12),      # generator input 1x1
    (3, 96, 96, 3, 1, 1, 1, 1, 1, True, 12),         # GBlock conv1a
    (3, 96, 96, 3, 1, 27, 27, 1, 1, True, 12),       # d=27 at T=12: outer taps live entirely in padding
    (2, 96, 48, 3, 1, 1, 1, 1, 2, True, 24),         # upsample x2 + relu
    (2, 96, 48, 1, 1, 1, 0, 1, 2, False, 24),        # res1 after upsample
    (2, 48, 48, 3, 1, 9, 9, 1, 1, True, 192),
    (2, 192, 8, 3, 1, 1, 1, 1, 1, True, 192),        # last conv (Cout=8)
    (6, 8, 32, 3, 1, 1, 2, 1, 1, False, 97),         # MPD layer 0 (pad 2, k 3)
    (6, 32, 256, 3, 3, 1, 2, 1, 1, False, 99),       # MPD stride 3
    (2, 8, 128, 15, 1, 1, 7, 1, 1, False, 192),      # MSD k15
    (2, 128, 256, 37, 2, 1, 18, 4, 1, False, 192),   # MSD grouped g4 stride 2
    (2, 256, 512, 37, 2, 1, 18, 16, 1, False, 96),   # MSD grouped g16
    (2, 512, 1024, 5, 1, 1, 2, 1, 1, False, 48),
    (2, 1024, 1, 3, 1, 1, 1, 1, 1, False, 48),       # output conv Cout=1
    (2, 128, 128, 41, 2, 1, 20, 4, 1, False, 64),    # full-D k41
    (2, 256, 512, 41, 4, 1, 20, 16, 1, False, 48),   # full-D stride 4
    (1, 8, 64, 3, 2, 1, 1, 1, 1, False, 64),         # encoder ResBlock conv1 stride 2
    (1, 64, 64, 1, 2, 1, 0, 1, 1, False, 33),        # encoder residual 1x1 stride 2, odd T
    (1, 5, 7, 3, 1, 2, 2, 1, 1, False, 1),           # T=1, ragged channel counts
    (5, 70, 130, 3, 1, 1, 1, 1, 1, False, 50),       # ragged M/N/R tails
    (64, 48, 48, 3, 1, 1, 1, 1, 1, True, 192),       # N = 12288: BN=128 tiles of the direct kernel
    (64, 48, 24, 3, 1, 1, 1, 1, 2, True, 96),        # upsample with BN=128
    (3, 16, 16, 3, 2, 1, 1, 1, 1, False, 7),         # stride 2, odd lengths: uneven bwd phases
    (2, 8, 8, 1, 2, 1, 0, 1, 1, False, 9),           # k1 stride 2: one bwd phase has no taps (must write zeros)
    (2, 64, 64, 41, 1, 1, 20, 16, 1, False, 24),     # full-D k41 grouped stride 1
    (22, 256, 512, 3, 3, 1, 2, 1, 1, False, 11),     # MPD p=11 tail: bwd phases with 4,4,3 columns per row
    (22, 512, 1, 3, 1, 1, 1, 1, 1, False, 4),        # MPD output conv at H=4
    (4, 32, 32, 5, 4, 1, 2, 1, 1, False, 10),        # stride 4: phases with 3,3,2,2 columns
    (3, 128, 256, 37, 2, 1, 18, 4, 1, False, 48),    # tap groups, short rows: most of the 37 taps reach into the padding
    (3, 128, 256, 37, 2, 1, 18, 4, 1, False, 24),    # ... T = 24: live taps are a sub-range (k_lo > 0)
    (3, 256, 512, 37, 2, 1, 18, 16, 1, False, 12),   # ... T = 12 -> 6, narrow groups
    (3, 256, 512, 37, 2, 1, 18, 16, 1, False, 24),
    (3, 8, 128, 15, 1, 1, 7, 1, 1, False, 24),       # k15 tap groups at T = 24 and 96
    (3, 8, 128, 15, 1, 1, 7, 1, 1, False, 96),
    (300, 8, 32, 3, 1, 1, 2, 1, 1, False, 18),       # MPD p = 11 layer 0: more chunks than blocks, 20 of a chunk's 64 columns live
    (5, 8, 128, 15, 1, 1, 7, 1, 1, False, 130),      # k15 first layer: three chunks a row, the last one 2 columns
]


def ref(x, w, scale, bias, res, cfg):
    B, Cin, Cout, K, s, d, p, g, up, relu, Tin = cfg
    h = F.relu(x) if relu else x
    if up == 2:
        h = F.interpolate(h, scale_factor=2.0, mode="nearest")
    y = F.conv1d(h, w * scale.view(-1, 1, 1), bias, stride=s, dilation=d, padding=p, groups=g)
    return y + res if res is not None else y


@pytest.mark.parametrize("cfg", CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_fwd_bwd(cfg):
    from ste_gan_amd import ops
    B, Cin, Cout, K, s, d, p, g, up, relu, Tin = cfg
    gen = torch.Generator().manual_seed(hash(cfg) % (2 ** 31))
    x = torch.randn(B, Cin, Tin, generator=gen, requires_grad=True)
    w = (torch.randn(Cout, Cin // g, K, generator=gen) / (Cin // g * K) ** 0.5).requires_grad_(True)
    scale = (0.5 + torch.rand(Cout, generator=gen))
    bias = torch.randn(Cout, generator=gen)
    c = ops.ConvCfg(Cin, Cout, K, s, d, p, g, up, 1 if relu else 0)
    Tout = c.tout(Tin)
    res = torch.randn(B, Cout, Tout, generator=gen)
    y_ref = ref(x, w, scale, bias, res, cfg)
    dy = torch.randn(B, Cout, Tout, generator=gen)
    dx_ref, dw_ref = torch.autograd.grad(y_ref, [x, w], dy)
    dw_ref = dw_ref / scale.view(-1, 1, 1)        # kernel returns the gradient of the RAW (unscaled) contraction
    dev = "cuda"
    xg, wg, sg, bg, rg, dyg = (t.detach().to(dev) for t in (x, w, scale, bias, res, dy))
    y = ops.conv1d_fwd(xg, wg, sg, bg, rg, c)
    tol = dict(rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(y.cpu(), y_ref.detach(), **tol)
    # bwd-data: relu mask + fold for the upsample
    dxe = ops.conv1d_bwd_data(dyg, wg, sg, xg if relu else None, None, c, Tin)
    dx = ops.fold_upsample2(dxe) if up == 2 else dxe
    torch.testing.assert_close(dx.cpu(), dx_ref, rtol=2e-5, atol=2e-5 * float(dx_ref.abs().max()))
    # fast path: direct-conv kernel on the K-major transposed weight copies (strided bwd runs as dense phases)
    wf, wb = ops.prep_weights(wg, c, None, sg)
    y2 = ops.conv1d_fwd(xg, wg, sg, bg, rg, c, wf)
    torch.testing.assert_close(y2.cpu(), y_ref.detach(), **tol)
    add = torch.randn(B, Cin, Tin * up, generator=gen)
    dxe2 = ops.conv1d_bwd_data(dyg, wg, sg, xg if relu else None, add.cuda(), c, Tin, wb)
    dxe_ref = (dxe + add.cuda()).cpu()
    torch.testing.assert_close(dxe2.cpu(), dxe_ref, rtol=2e-5, atol=2e-5 * float(dxe_ref.abs().max()))
    # bwd-weight (+ bias grad), accumulate semantics: start from a non-zero buffer
    dw = torch.full_like(wg, 0.25)
    db = torch.full((Cout,), -1.0, device=dev)
    ops.conv1d_bwd_weight(xg, dyg, c, dw, db)
    torch.testing.assert_close(dw.cpu() - 0.25, dw_ref, rtol=1e-4, atol=1e-4 * float(dw_ref.abs().max()))
    torch.testing.assert_close(db.cpu() + 1.0, dy.sum((0, 2)), rtol=1e-4, atol=1e-4 * float(dy.sum((0, 2)).abs().max()))


def test_bad_shape_raises():
    from ste_gan_amd import ops
    c = ops.ConvCfg(8, 8, 3, pad=1)
    with pytest.raises(RuntimeError):
        ops.conv1d_fwd(torch.randn(2, 7, 16, device="cuda"), torch.randn(8, 8, 3, device="cuda"), None, None, None, c)
    with pytest.raises(RuntimeError):       # no silent CPU fallback
        ops.conv1d_fwd(torch.randn(2, 8, 16), torch.randn(8, 8, 3), None, None, None, c)


# (B, Cin, K, stride, dil, pad, Tin, out_act)
SKINNY_CASES = [
    (64, 1024, 3, 1, 1, 1, 48, 0),      # MSD output conv of the hot path
    (22, 512, 3, 1, 1, 1, 4, 0),        # MPD output conv, H = 4
    (10, 512, 3, 1, 1, 1, 5, 0),        # odd H
    (3, 70, 3, 1, 1, 1, 50, 2),         # ragged channels, leaky-ReLU epilogue
    (2, 33, 5, 2, 1, 2, 71, 0),         # stride 2, k5, rows longer than one 64-step tile
    (2, 16, 3, 1, 3, 3, 130, 3),        # dilation 3, tanh, three tiles
    (1, 8, 8, 1, 1, 0, 9, 0),           # K = 8 (the kernel's tap limit), no padding
    (2, 300, 1, 1, 1, 0, 7, 0),         # 1x1
]


@pytest.mark.parametrize("cfg", SKINNY_CASES, ids=lambda c: "x".join(map(str, c)))
def test_single_output_channel_kernels(cfg):
    """conv_skinny.hip (Cout == 1: forward, input gradient, weight gradient) against F.conv1d + autograd in fp64, and the
    kernel selection: these shapes must take the streaming kernels, everything they decline the general ones."""
    from ste_gan_amd import ops
    B, Cin, K, s, d, p, Tin, act = cfg
    gen = torch.Generator().manual_seed(hash(cfg) % (2 ** 31))
    x = torch.randn(B, Cin, Tin, generator=gen, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(1, Cin, K, generator=gen, dtype=torch.float64) / (Cin * K) ** 0.5).requires_grad_(True)
    scale = 0.5 + torch.rand(1, generator=gen, dtype=torch.float64)
    bias = torch.randn(1, generator=gen, dtype=torch.float64)
    c = ops.ConvCfg(Cin, 1, K, s, d, p, 1, 1, 0, act)
    Tout = c.tout(Tin)
    pre = F.conv1d(x, w * scale.view(-1, 1, 1), bias, stride=s, dilation=d, padding=p)
    y_ref = {0: lambda v: v, 2: lambda v: F.leaky_relu(v, 0.1), 3: torch.tanh}[act](pre)
    dy = torch.randn(B, 1, Tout, generator=gen, dtype=torch.float64)
    dx_ref, dw_ref = torch.autograd.grad(pre, [x, w], dy)
    dw_ref = dw_ref / scale.view(-1, 1, 1)
    xg, wg, sg, bg, dyg = (t.detach().float().cuda() for t in (x, w, scale, bias, dy))
    y = ops.conv1d_fwd(xg, wg, sg, bg, None, c)
    assert ops.LAST_CONV_KERNEL == "skinny"
    torch.testing.assert_close(y.double().cpu(), y_ref.detach(), rtol=1e-5, atol=1e-5)
    dx = ops.conv1d_bwd_data(dyg, wg, sg, None, None, c, Tin)
    assert ops.LAST_CONV_KERNEL == "skinny"
    torch.testing.assert_close(dx.double().cpu(), dx_ref, rtol=1e-5, atol=1e-5 * float(dx_ref.abs().max()))
    dw = torch.full_like(wg, 0.25)
    db = torch.full((1,), -1.0, device="cuda")
    ops.conv1d_bwd_weight(xg, dyg, c, dw, db)
    assert ops.LAST_CONV_KERNEL == "skinny"
    torch.testing.assert_close(dw.double().cpu() - 0.25, dw_ref, rtol=1e-4, atol=2e-5 * float(dw_ref.abs().max()))
    torch.testing.assert_close(db.double().cpu() + 1.0, dy.sum((0, 2)), rtol=1e-4, atol=1e-4 * float(dy.abs().sum()))
    # declined: residual (forward), mask / add (input gradient), more than one output channel
    res = torch.randn(B, 1, Tout, device="cuda")
    y2 = ops.conv1d_fwd(xg, wg, sg, bg, res, c)
    assert ops.LAST_CONV_KERNEL != "skinny"
    torch.testing.assert_close(y2.double().cpu(), {0: lambda v: v, 2: lambda v: F.leaky_relu(v, 0.1), 3: torch.tanh}[act](
        pre.detach() + res.double().cpu()), rtol=2e-5, atol=2e-5)


def test_single_output_channel_multi_launch_equals_single_launches():
    """stegan_conv1d_fwd_skinny_multi / _bwd_data_skinny_multi (the period discriminators' output convs in one launch)
    against one launch per problem: same kernels' bodies, so bit equality."""
    from ste_gan_amd import ops
    g = torch.Generator().manual_seed(9)
    shapes = [(128, 1024, 24), (192, 1024, 16), (320, 1024, 10), (448, 1024, 7), (704, 1024, 5)]       # (B * period, Cin, H)
    cfgs, xs, ws, scs, bs, dys = [], [], [], [], [], []
    for B, Cin, T in shapes:
        cfgs.append(ops.ConvCfg(Cin, 1, 3, 1, 1, 1))
        xs.append(torch.randn(B, Cin, T, generator=g).cuda())
        ws.append((torch.randn(1, Cin, 3, generator=g) * 0.05).cuda())
        scs.append((torch.rand(1, generator=g) + 0.5).cuda())
        bs.append(torch.randn(1, generator=g).cuda())
        dys.append(torch.randn(B, 1, T, generator=g).cuda())
    ys = ops.conv1d_fwd_multi(xs, cfgs, [None] * 5, scs, bs, ws)
    assert ys is not None
    dxs = ops.conv1d_bwd_data_multi(dys, cfgs, [None] * 5, [s[2] for s in shapes], ws, scs)
    assert dxs is not None
    for x, w, sc, b, dy, c, y, dx in zip(xs, ws, scs, bs, dys, cfgs, ys, dxs):
        y1 = ops.conv1d_fwd(x, w, sc, b, None, c)
        assert ops.LAST_CONV_KERNEL == "skinny"
        assert torch.equal(y, y1)
        dx1 = ops.conv1d_bwd_data(dy, w, sc, None, None, c, x.shape[2])
        assert torch.equal(dx, dx1)
    # weight + bias gradients of all five in one launch: slabs summed in a fixed order -- bit-equal to the single launches
    from ste_gan_amd import _lib
    import ctypes as C
    n = len(shapes)
    dws = [torch.full((1, s[1], 3), 0.5, device="cuda") for s in shapes]
    dbs = [torch.full((1,), -2.0, device="cuda") for _ in shapes]
    descs = [ops._desc(c, x.shape[0], x.shape[2], dy.shape[2]) for c, x, dy in zip(cfgs, xs, dys)]
    dptr = (C.c_void_p * n)(*[C.cast(C.pointer(d), C.c_void_p) for d in descs])
    wsf = sum(int(_lib.lib().stegan_conv1d_bwd_weight_skinny_floats(C.byref(d))) for d in descs)
    ws = torch.empty(wsf, device="cuda")
    assert _lib.lib().stegan_conv1d_bwd_weight_skinny_multi(n, dptr, ops._ptr_array(xs), ops._ptr_array(dys), ops._ptr_array(dws),
                                                            ops._ptr_array(dbs), ops._p(ws), wsf - 4, None) == 1   # too small
    rc = _lib.lib().stegan_conv1d_bwd_weight_skinny_multi(n, dptr, ops._ptr_array(xs), ops._ptr_array(dys), ops._ptr_array(dws),
                                                          ops._ptr_array(dbs), ops._p(ws), wsf, None)
    assert rc == 0
    torch.cuda.synchronize()
    for x, dy, c, dw, db in zip(xs, dys, cfgs, dws, dbs):
        dw1 = torch.full_like(dw, 0.5)
        db1 = torch.full_like(db, -2.0)
        ops.conv1d_bwd_weight(x, dy, c, dw1, db1)
        assert torch.equal(dw, dw1) and torch.equal(db, db1)
