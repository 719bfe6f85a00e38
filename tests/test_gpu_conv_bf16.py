"""-m gpu: the bf16-operand direct conv kernels (stegan_conv1d_fwd_h / _bwd_data_h, csrc/conv_bf16.hip) through the C ABI.

Reference: the SAME contraction on the CPU in fp64 with both operands first rounded to bf16 exactly as the kernel rounds
them (activations after ReLU / upsample; the forward weights raw, the input-gradient weights after folding the
per-channel scale) -- so the comparison isolates the kernel's indexing / staging / accumulation from the bf16 rounding
itself: tolerance 2e-5 relative to the result's scale (fp32 accumulation order).  A second check bounds the distance to
the un-rounded fp32 result by the bf16 rounding model (2^-8 relative per operand, random signs: ~1e-2 of the scale)."""
import pytest
import torch
import torch.nn.functional as F

from test_gpu_conv import CASES

pytestmark = pytest.mark.gpu

# the 2-byte operand type under test: tests/test_gpu_conv_f16.py re-runs this module's kernel tests with ("f16", torch.float16)
# (libstegan_hip_f16.so: the same sources built with fp16 operands)
OPERAND = ("bf16", torch.bfloat16)
# tolerance of the comparison with the fp64 contraction of rounded operands, relative to the result's scale: what the fp32
# accumulation inside the MFMA leaves.  (fp16 products carry 22 significant bits against bf16's 16, so the MFMA's fp32 sums
# round more of them away: the f16 module runs with 2x this.)
TOL = 2e-5
# the 2-byte-operand kernel families: the round 2-4 kernels ("bf16") and the one-tile-per-CU kernels of round 5 ("tile")
HALF_KERNELS = ("bf16", "tile")

EXTRA = [
    (64, 96, 96, 3, 1, 27, 27, 1, 1, True, 192),     # 128 x 128 tiles, d = 27 halo, N = 12288
    (64, 192, 96, 3, 1, 1, 1, 1, 2, True, 96),       # upsample, 128-column tiles
    (8, 384, 256, 1, 1, 1, 0, 1, 1, False, 96),      # K = 1, 64-channel chunks
    (4, 136, 200, 5, 1, 1, 2, 1, 1, False, 40),      # ragged channels: 136 = 2 chunks of 64 + 8, 200 rows
    (16, 64, 64, 3, 1, 1, 1, 1, 1, False, 12),       # short rows: a tile spans many batch segments
    (16, 64, 64, 3, 1, 1, 1, 1, 2, True, 12),        # upsample with 16-byte loads (8 slots per item), many batch segments
    (6, 72, 48, 5, 1, 3, 6, 1, 2, False, 20),        # upsample, dilated 5 taps, ragged channels, Tq = 40
    (6, 32, 48, 3, 1, 1, 1, 1, 2, True, 18),         # upsample whose rows are not float4-aligned
    (10, 32, 128, 5, 3, 1, 2, 1, 1, False, 2731),    # period discriminator rows
    (6, 128, 256, 3, 1, 1, 1, 1, 1, True, 61),       # ragged rows, a tile spans several batch elements
    (7, 64, 64, 3, 1, 3, 3, 1, 1, False, 13),        # ... short ragged rows, dilated
    (16, 64, 64, 3, 1, 9, 9, 1, 1, True, 12),        # short rows, wide halos: the wgrad's half-chunk plan (one-item variant)
    (10, 128, 128, 3, 1, 3, 3, 1, 1, True, 12),      # ... d = 3, 128-row co tile
]
RAGGED = [c for c in CASES + EXTRA if c[10] % 4 != 0 and c[1] // c[7] >= 8 and c[2] // c[7] >= 8]


def bf16r(t):
    return t.to(OPERAND[1]).to(torch.float64)


def make_image(t):
    """The bf16 activation image of a (B, C, T) tensor: (B, ceil(C/8), T, 8), channel-block-major (ops._image_of)."""
    B, C, T = t.shape
    CB = (C + 7) // 8
    full = torch.zeros(B, CB * 8, T, device=t.device, dtype=t.dtype)
    full[:, :C] = t
    return full.view(B, CB, 8, T).permute(0, 1, 3, 2).contiguous().to(OPERAND[1])


@pytest.mark.parametrize("cfg", CASES + EXTRA, ids=lambda c: "x".join(map(str, c)))
def test_conv_bf16_with_activation_images(cfg):
    """The same parity bar with the source handed over as a bf16 image (staged by LDS-DMA, ReLU applied at the operand
    read) and the output image requested: the image the kernel writes is exactly bf16(fp32 output), rearranged."""
    test_conv_bf16_fwd_bwd_data(cfg, images=True)


@pytest.mark.parametrize("cfg", CASES + EXTRA, ids=lambda c: "x".join(map(str, c)))
def test_conv_bf16_fwd_bwd_data(cfg, images=False):
    from ste_gan_amd import ops
    B, Cin, Cout, K, s, d, p, g, up, relu, Tin = cfg
    gen = torch.Generator().manual_seed(hash(cfg) % (2 ** 31))
    x = torch.randn(B, Cin, Tin, generator=gen)
    w = torch.randn(Cout, Cin // g, K, generator=gen) / (Cin // g * K) ** 0.5
    scale = 0.5 + torch.rand(Cout, generator=gen)
    bias = torch.randn(Cout, generator=gen)
    c = ops.ConvCfg(Cin, Cout, K, s, d, p, g, up, 1 if relu else 0)
    Tout = c.tout(Tin)
    res = torch.randn(B, Cout, Tout, generator=gen)
    dy = torch.randn(B, Cout, Tout, generator=gen)
    add = torch.randn(B, Cin, Tin * up, generator=gen)

    hx = F.relu(x) if relu else x
    if up == 2:
        hx = F.interpolate(hx, scale_factor=2.0, mode="nearest")
    y_ref = F.conv1d(bf16r(hx), bf16r(w), None, stride=s, dilation=d, padding=p, groups=g) * scale.double().view(1, -1, 1) \
        + bias.double().view(1, -1, 1) + res.double()
    y_f32 = F.conv1d(hx.double(), w.double(), None, stride=s, dilation=d, padding=p, groups=g) * scale.double().view(1, -1, 1) \
        + bias.double().view(1, -1, 1) + res.double()
    # input gradient of the conv w.r.t. its (effective-resolution) input, operands rounded as the kernel rounds them
    ws_ = w * scale.view(-1, 1, 1)

    def dx_of(dy_, w_):
        xe = torch.zeros(B, Cin, Tin * up, dtype=torch.float64, requires_grad=True)
        ye = F.conv1d(xe, w_, None, stride=s, dilation=d, padding=p, groups=g)
        (gx,) = torch.autograd.grad(ye, xe, dy_)
        return gx

    mask = (x > 0).double()
    if up == 2:
        mask = F.interpolate(mask, scale_factor=2.0, mode="nearest")
    dx_ref = dx_of(bf16r(dy), bf16r(ws_)) * (mask if relu else 1.0) + add.double()
    dx_f32 = dx_of(dy.double(), ws_.double()) * (mask if relu else 1.0) + add.double()

    dev = "cuda"
    xg, wg, sg, bg, rg, dyg, addg = (t.to(dev) for t in (x, w, scale, bias, res, dy, add))
    ops.set_compute_dtype(OPERAND[0])
    try:
        wf, wb = ops.prep_weights(wg, c, None, sg)
        assert wf.h.dtype == OPERAND[1] and wb.h is not None
        if images:
            ops._attach_image(xg, make_image(xg)); ops._attach_image(dyg, make_image(dyg))
            assert ops._image_of(xg, Cin, Tin) is not None
        y = ops.conv1d_fwd(xg, wg, sg, bg, rg, c, wf)
        k_fwd = ops.LAST_CONV_KERNEL
        dxe = ops.conv1d_bwd_data(dyg, wg, sg, xg if relu else None, addg, c, Tin, wb)
        k_bwd = ops.LAST_CONV_KERNEL
        if k_fwd in HALF_KERNELS and Cout % 8 == 0 and (Cout // g) % 8 == 0 and Cout >= 16:
            assert torch.equal(y.h.view(torch.int16), make_image(y).view(torch.int16)), "output image != bf16(y)"
        if k_bwd in HALF_KERNELS and up == 1 and Cin % 8 == 0 and (Cin // g) % 8 == 0 and Cin >= 16:
            assert torch.equal(dxe.h.view(torch.int16), make_image(dxe).view(torch.int16)), "input-gradient image != bf16(dx)"
    finally:
        ops.set_compute_dtype("f32")
    cin_g, cout_g = Cin // g, Cout // g
    # shapes the bf16 kernel declines run on the implicit-GEMM kernel from the raw tensor (bf16 mode keeps no fp32 copies)
    assert (k_fwd in HALF_KERNELS) if (cin_g >= 8 and cout_g >= 8) else (k_fwd == "igemm"), k_fwd
    assert (k_bwd in HALF_KERNELS) if (cout_g >= 8 and cin_g >= 8) else (k_bwd == "igemm"), k_bwd
    for got, ref, f32, kern in ((y, y_ref, y_f32, k_fwd), (dxe, dx_ref, dx_f32, k_bwd)):
        got = got.double().cpu()
        scale_ = float(f32.abs().max())
        if kern in HALF_KERNELS:
            assert float((got - ref).abs().max()) <= TOL * scale_, (kern, float((got - ref).abs().max()), scale_)
            assert float((got - f32).abs().max()) <= 3e-2 * scale_
        else:                                             # shape left to the exact fp32 kernel
            assert float((got - f32).abs().max()) <= 2e-5 * scale_


@pytest.mark.parametrize("cfg", RAGGED, ids=lambda c: "x".join(map(str, c)))
def test_conv_bf16_ragged_rows_with_16_byte_loads(cfg, monkeypatch):
    """The opt-in STEGAN_BF16_XVRAG path (dwordx4 loads at dword alignment, the row's tail zeroed): same parity bar."""
    monkeypatch.setenv("STEGAN_BF16_XVRAG", "1")
    test_conv_bf16_fwd_bwd_data(cfg)


@pytest.mark.parametrize("cfg", CASES + EXTRA, ids=lambda c: "x".join(map(str, c)))
def test_conv_bf16_bwd_weight_with_activation_images(cfg):
    """Both operands handed over as bf16 images (LDS-DMA staging, ReLU at the operand read): the same parity bar."""
    test_conv_bf16_bwd_weight(cfg, images=True)


@pytest.mark.parametrize("cfg", CASES + EXTRA, ids=lambda c: "x".join(map(str, c)))
def test_conv_bf16_bwd_weight(cfg, images=False):
    """dW (accumulated into a non-zero buffer) and db of stegan_conv1d_bwd_weight_h against fp64 on bf16-rounded operands."""
    from ste_gan_amd import ops
    B, Cin, Cout, K, s, d, p, g, up, relu, Tin = cfg
    gen = torch.Generator().manual_seed(hash(cfg) % (2 ** 31) + 1)
    x = torch.randn(B, Cin, Tin, generator=gen)
    c = ops.ConvCfg(Cin, Cout, K, s, d, p, g, up, 1 if relu else 0)
    Tout = c.tout(Tin)
    dy = torch.randn(B, Cout, Tout, generator=gen)
    hx = F.relu(x) if relu else x
    if up == 2:
        hx = F.interpolate(hx, scale_factor=2.0, mode="nearest")

    def dw_of(x_, dy_):
        w0 = torch.zeros(Cout, Cin // g, K, dtype=torch.float64, requires_grad=True)
        y0 = F.conv1d(x_, w0, None, stride=s, dilation=d, padding=p, groups=g)
        (gw,) = torch.autograd.grad(y0, w0, dy_)
        return gw

    dw_ref, dw_f32 = dw_of(bf16r(hx), bf16r(dy)), dw_of(hx.double(), dy.double())
    db_ref = dy.double().sum((0, 2))
    dw = torch.full((Cout, Cin // g, K), 0.25, device="cuda")
    db = torch.full((Cout,), -1.0, device="cuda")
    ops.set_compute_dtype(OPERAND[0])
    try:
        xg, dyg = x.cuda(), dy.cuda()
        if images:
            ops._attach_image(xg, make_image(xg)); ops._attach_image(dyg, make_image(dyg))
        ops.conv1d_bwd_weight(xg, dyg, c, dw, db)
        kern = ops.LAST_CONV_KERNEL
    finally:
        ops.set_compute_dtype("f32")
    if Cout == 1 and g == 1:
        assert kern == "skinny", kern                      # single output channel: streaming fp32 kernels
    elif Cin // g < 16 or Cout // g < 16:
        assert kern == "igemm", kern                       # narrow groups stay on the exact fp32 kernels
    elif s == 1:
        assert kern in HALF_KERNELS, kern                        # (strided layers with long halo'd rows may not fit: fp32 then)
    got = dw.double().cpu() - 0.25
    scale_ = float(dw_f32.abs().max())
    if kern in HALF_KERNELS:
        assert float((got - dw_ref).abs().max()) <= 1.5 * TOL * scale_ + 1e-6, (float((got - dw_ref).abs().max()), scale_)
        assert float((got - dw_f32).abs().max()) <= 3e-2 * scale_
    else:
        assert float((got - dw_f32).abs().max()) <= 1e-4 * scale_
    assert float((db.double().cpu() + 1.0 - db_ref).abs().max()) <= 1e-4 * float(db_ref.abs().max()) + 1e-5


def test_prep_bf16_layout():
    """wfh[g][k][ci/8][co][ci%8] / wbh[g][k][co/8][ci][co%8] x scale, zero padding and the all-zero plane k = K."""
    from ste_gan_amd import ops
    Cin, Cout, K, g = 24, 40, 3, 2
    c = ops.ConvCfg(Cin, Cout, K, 1, 1, 1, g)
    gen = torch.Generator().manual_seed(5)
    w = torch.randn(Cout, Cin // g, K, generator=gen)
    sc = 0.5 + torch.rand(Cout, generator=gen)
    ops.set_compute_dtype(OPERAND[0])
    try:
        wf, wb = ops.prep_weights(w.cuda(), c, None, sc.cuda())
        wfh, wbh = wf.h.cpu(), wb.h.cpu()
    finally:
        ops.set_compute_dtype("f32")
    cog, cig = Cout // g, Cin // g
    # device layout [g][k][cb][m][8]  ->  [g][k][m][c] for the comparison
    wfh = wfh.view(g, K + 1, 8, 128, 8).permute(0, 1, 3, 2, 4).reshape(g, K + 1, 128, 64)
    wbh = wbh.view(g, K + 1, 8, 128, 8).permute(0, 1, 3, 2, 4).reshape(g, K + 1, 128, 64)
    want_f = torch.zeros(g, K + 1, 128, 64)
    want_b = torch.zeros(g, K + 1, 128, 64)
    for gi in range(g):
        blk = w[gi * cog:(gi + 1) * cog]                                  # (cog, cig, K)
        want_f[gi, :K, :cog, :cig] = blk.permute(2, 0, 1)
        want_b[gi, :K, :cig, :cog] = (blk * sc[gi * cog:(gi + 1) * cog].view(-1, 1, 1)).permute(2, 1, 0)
    assert torch.equal(wfh, want_f.to(OPERAND[1]))
    assert torch.equal(wbh, want_b.to(OPERAND[1]))


def test_gradients_summed_by_autograd_do_not_use_a_stale_image():
    """Two convs read the same activation x WITHOUT the gradient hand-over: each backward returns an input gradient that
    carries a bf16 image, and autograd's gradient buffer adds the second into the first IN PLACE -- the tensor object (and
    its image of the first summand) survives the sum.  ops._image_of must see the in-place version move and fall back to
    staging from the fp32 sum; the producer's gradients then equal a run with images switched off (both round the same
    fp32 sum to bf16)."""
    from ste_gan_amd import ops
    from ste_gan_amd.layers.conv import WNConv1d
    torch.manual_seed(11)
    ops.set_compute_dtype(OPERAND[0])
    try:
        conv0 = WNConv1d(64, 64, 3, padding=1).cuda()
        conv1 = WNConv1d(64, 96, 3, padding=1).cuda()
        conv2 = WNConv1d(64, 32, 1).cuda()
        inp = torch.randn(4, 64, 48, device="cuda")
        g1, g2 = torch.randn(4, 96, 48, device="cuda"), torch.randn(4, 32, 48, device="cuda")
        res = []
        for images in (True, False):
            ops.IMAGES = images
            for c in (conv0, conv1, conv2):
                for p in c.parameters():
                    p.grad = None
            a = inp.clone().requires_grad_(True)
            x = conv0(a)
            assert (ops._image_of(x, 64, 48) is not None) == images
            torch.autograd.backward([conv1(x), conv2(x)], [g1, g2])
            res.append([a.grad.clone()] + [p.grad.clone() for p in conv0.parameters()])
        for got, want in zip(*res):
            assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max()) + 1e-7
    finally:
        ops.IMAGES = True
        ops.set_compute_dtype("f32")


@pytest.mark.parametrize("shape", [(6, 32, 13), (4, 256, 34), (10, 24, 7), (3, 16, 48)], ids=str)
def test_activation_backward_writes_the_image_for_any_row_length(shape):
    """stegan_act_bwd_add_img / _multi_img: (dy + add) * lrelu'(y) and its bf16 image, rows of any length (the period
    discriminators fold the signal into rows of 97 / p, 34, 12 ... steps: not multiples of 4)."""
    from ste_gan_amd import ops
    from ste_gan_amd._lib import ACT_LRELU
    torch.manual_seed(sum(shape))
    ops.set_compute_dtype(OPERAND[0])
    try:
        dy, add, y = (torch.randn(*shape, device="cuda") for _ in range(3))
        want = (dy + add) * torch.where(y > 0, 1.0, 0.1)
        out = ops.act_bwd(dy, y, ACT_LRELU, add)
        torch.testing.assert_close(out, want, rtol=1e-6, atol=1e-7)
        h = ops._image_of(out, shape[1], shape[2])
        assert h is not None and torch.equal(h.view(torch.int16), make_image(out).view(torch.int16))
    finally:
        ops.set_compute_dtype("f32")


@pytest.mark.parametrize("slices", ["1", "2"])
def test_weight_gradient_accumulates_onto_a_non_constant_dw(slices, monkeypatch):
    """dW += ...: two accumulating calls onto a RANDOM dW, sliced (slabs + ordered sum) and unsliced (read-modify-write in the
    kernel's epilogue).  Every other test starts from a constant dW, which hid a wrong wide load in the unsliced path: a
    buffer_load_dwordx2 at a 4-but-not-8-byte aligned offset returned its first dword twice (tap 1 == tap 0 for k = 3)."""
    from ste_gan_amd import ops
    monkeypatch.setenv("STEGAN_BF16_WGRAD_SLICES", slices)
    B, Cin, Cout, K, Tin = 2, 96, 48, 3, 24
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(B, Cin, Tin, generator=gen)
    c = ops.ConvCfg(Cin, Cout, K, 1, 1, 1, 1, 2, 1)                 # k3, nearest x2, input ReLU
    dy = torch.randn(B, Cout, c.tout(Tin), generator=gen)
    hx = F.interpolate(F.relu(x), scale_factor=2.0, mode="nearest")
    w0 = torch.zeros(Cout, Cin, K, dtype=torch.float64, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv1d(bf16r(hx), w0, None, padding=1), w0, bf16r(dy))
    ops.set_compute_dtype(OPERAND[0])
    try:
        for images in (False, True):
            init = torch.randn(Cout, Cin, K, generator=gen)
            dw, db = init.cuda().clone(), torch.zeros(Cout, device="cuda")
            xg, dyg = x.cuda(), dy.cuda()
            if images:
                ops._attach_image(xg, make_image(xg)); ops._attach_image(dyg, make_image(dyg))
            ops.conv1d_bwd_weight(xg, dyg, c, dw, db)
            ops.conv1d_bwd_weight(xg, dyg, c, dw, db)
            assert ops.LAST_CONV_KERNEL in ("bf16", "tile")
            err = (dw.double().cpu() - init.double() - 2 * ref).abs()
            assert float(err.max()) <= 3e-5 * float(ref.abs().max()), err.amax((0, 1)).tolist()
    finally:
        ops.set_compute_dtype("f32")


def test_chained_weight_gradients_equal_the_unchained_ones_bit_for_bit():
    """ops.conv1d_bwd_weight(..., chain=...): the slab sum of a sliced layer is run by the NEXT layer's launch
    (stegan_conv1d_bwd_weight_hxc) and only the chain's tail pays a reduce launch.  Same adds in the same order: dW of
    every layer equals the per-layer path's bits (db to summation-order noise) -- for a chain of sliced tile layers, an unsliced one in the middle (nothing
    pending behind it) and a shape the tile kernel declines (its predecessor's sum then runs as a launch of its own)."""
    from ste_gan_amd import ops
    gen = torch.Generator().manual_seed(5)
    # (B, Cin, Cout, K, dil, up, relu, Tin): sliced x3, one slice (T = 12), declined by the tile kernel (Tout % 4), sliced
    layers = [(16, 96, 96, 3, 1, 1, 1, 48), (16, 96, 48, 3, 3, 2, 1, 24), (16, 48, 96, 1, 1, 1, 0, 96), (16, 768, 96, 3, 1, 1, 1, 12),
              (8, 64, 64, 3, 1, 1, 0, 30), (16, 96, 96, 3, 9, 1, 1, 96)]
    ops.set_compute_dtype(OPERAND[0])
    try:
        data = []
        for (B, Cin, Cout, K, d, up, relu, Tin) in layers:
            c = ops.ConvCfg(Cin, Cout, K, 1, d, d * (K - 1) // 2, 1, up, relu)
            x = torch.randn(B, Cin, Tin, generator=gen).cuda(); dy = torch.randn(B, Cout, c.tout(Tin), generator=gen).cuda()
            ops._attach_image(x, make_image(x)); ops._attach_image(dy, make_image(dy))
            init = torch.randn(Cout, Cin, K, generator=gen).cuda()
            data.append((c, x, dy, init))
        plain = []
        for c, x, dy, init in data:
            dw, db = init.clone(), torch.ones(c.cout, device="cuda")
            ops.conv1d_bwd_weight(x, dy, c, dw, db)
            plain.append((dw, db))
        chain, got, kinds = [None], [], []
        for c, x, dy, init in data:
            dw, db = init.clone(), torch.ones(c.cout, device="cuda")
            ops.conv1d_bwd_weight(x, dy, c, dw, db, defer=True, chain=chain)
            kinds.append((ops.LAST_CONV_KERNEL, chain[0] is not None))
            got.append((dw, db))
        assert chain[0] is not None                          # the last layer is sliced: its sum is still pending
        ops.wgrad_chain_end(chain)
        ops.wgrad_join()
        torch.cuda.synchronize()
        assert [k for k, _ in kinds].count("tile") >= 4 and sum(p for _, p in kinds) >= 3, kinds
        for i, ((dw, db), (dw0, db0)) in enumerate(zip(got, plain)):
            assert torch.equal(dw, dw0), (i, kinds[i], float((dw - dw0).abs().max()))
            # (the bias gradient is summed by a wave of the tile blocks here and by a block of the reduce launch there: two
            # fixed orders, the same value to fp32 summation noise)
            assert float((db - db0).abs().max()) <= 2e-6 * float(db0.abs().max()) + 1e-6, (i, kinds[i])
    finally:
        ops.set_compute_dtype("f32")


def test_multi_layer_weight_gradient_launch_equals_the_per_layer_launches():
    """ops._wgrad_multi: the weight gradients of a batch of layers (a generator block's five convs) as ONE launch of the tile
    kernel + one slab-sum launch (stegan_conv1d_bwd_weight_multi_hxc).  Same operands, same fp32 accumulation; the batch is cut
    into fewer slices than in the per-layer launches, so dW agrees to fp32 summation-order noise (1e-6 of its scale), db exactly
    where the per-layer path also sums it in the tile blocks.  A batch with a layer the tile kernel does not take is refused
    as a whole (nothing enqueued)."""
    from ste_gan_amd import ops
    gen = torch.Generator().manual_seed(9)
    # (B, Cin, Cout, K, dil, up, relu, Tin): a GBlock with an upsampling first conv and its 1 x 1 residual
    layers = [(16, 192, 192, 3, 1, 2, 1, 24), (16, 192, 192, 3, 3, 1, 1, 48), (16, 192, 192, 3, 9, 1, 1, 48),
              (16, 192, 192, 3, 27, 1, 1, 48), (16, 192, 192, 1, 1, 2, 0, 24)]
    ops.set_compute_dtype(OPERAND[0])
    try:
        batch, plain = [], []
        for (B, Cin, Cout, K, d, up, relu, Tin) in layers:
            c = ops.ConvCfg(Cin, Cout, K, 1, d, d * (K - 1) // 2, 1, up, relu)
            x = torch.randn(B, Cin, Tin, generator=gen).cuda(); dy = torch.randn(B, Cout, c.tout(Tin), generator=gen).cuda()
            ops._attach_image(x, make_image(x)); ops._attach_image(dy, make_image(dy))
            init = torch.randn(Cout, Cin, K, generator=gen).cuda()
            dw0, db0 = init.clone(), torch.ones(Cout, device="cuda")
            ops.conv1d_bwd_weight(x, dy, c, dw0, db0)
            plain.append((dw0, db0))
            batch.append((x, dy, c, init.clone(), torch.ones(Cout, device="cuda")))
        assert ops._wgrad_multi(batch) is True and ops.LAST_CONV_KERNEL == "tile"
        ops.wgrad_join()
        torch.cuda.synchronize()
        for i, ((x, dy, c, dw, db), (dw0, db0)) in enumerate(zip(batch, plain)):
            scale = float((dw0 - 0).abs().max())
            assert float((dw - dw0).abs().max()) <= 2e-6 * scale + 1e-6, (i, float((dw - dw0).abs().max()), scale)
            assert float((db - db0).abs().max()) <= 2e-6 * float(db0.abs().max()) + 1e-6, i
        # a strided layer in the batch: refused as a whole, dW untouched
        cs = ops.ConvCfg(192, 192, 3, 2, 1, 1, 1, 1, 0)
        xs = torch.randn(16, 192, 48, generator=gen).cuda(); dys = torch.randn(16, 192, cs.tout(48), generator=gen).cuda()
        ops._attach_image(xs, make_image(xs)); ops._attach_image(dys, make_image(dys))
        keep = batch[0][3].clone()
        assert ops._wgrad_multi([batch[0], (xs, dys, cs, torch.zeros(192, 192, 3, device="cuda"), None)]) is False
        torch.cuda.synchronize()
        assert torch.equal(batch[0][3], keep)
    finally:
        ops.set_compute_dtype("f32")
