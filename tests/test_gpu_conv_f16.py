"""-m gpu: the matrix-core kernels built with fp16 operands (libstegan_hip_f16.so, ``ops.set_compute_dtype("f16")`` -- BASELINE
config 5, the reference's autocast dtype, ste_gan/train.py:181,204).

Same sources as the bf16 build (csrc/common.h: the operand type is a build parameter), so the same tests with the other
rounding: every kernel against an fp64 contraction of operands rounded to fp16 exactly where the kernel rounds them (4e-5 of
the result's scale: forward, input gradient, weight gradient; raw tensors and activation images), the image contents
bit-exact, the prepared weight layouts, ragged rows."""
import pytest
import torch

import test_gpu_conv_bf16 as T

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def f16_operands(monkeypatch):
    from ste_gan_amd import _lib, ops
    monkeypatch.setattr(T, "OPERAND", ("f16", torch.float16))
    monkeypatch.setattr(T, "TOL", 4e-5)
    yield
    ops.set_compute_dtype("f32")
    assert _lib.lib().stegan_operand_type() == b"bf16"                   # the default build serves the fp32 path


def test_the_f16_build_is_loaded():
    from ste_gan_amd import _lib, ops
    ops.set_compute_dtype("f16")
    assert _lib.lib().stegan_operand_type() == b"f16" and ops.HALF and ops.HALF_DTYPE == torch.float16


@pytest.mark.parametrize("cfg", T.CASES + T.EXTRA, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("images", [False, True], ids=["raw", "images"])
def test_conv_f16_fwd_bwd_data(cfg, images):
    T.test_conv_bf16_fwd_bwd_data(cfg, images=images)


@pytest.mark.parametrize("cfg", T.CASES + T.EXTRA, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("images", [False, True], ids=["raw", "images"])
def test_conv_f16_bwd_weight(cfg, images):
    T.test_conv_bf16_bwd_weight(cfg, images=images)


def test_prep_f16_layout():
    T.test_prep_bf16_layout()


def test_f16_stale_image_and_ragged_activation_backward():
    T.test_gradients_summed_by_autograd_do_not_use_a_stale_image()
    for shape in [(6, 32, 13), (4, 256, 34)]:
        T.test_activation_backward_writes_the_image_for_any_row_length(shape)


def test_f16_overflow_becomes_inf_not_garbage():
    """fp16 tops out at 65504: an operand beyond it is staged as inf (IEEE conversion, not saturation) and the fp32 sum says so
    -- what the loss scaler's skip-on-inf (train.GradScaler) relies on."""
    from ste_gan_amd import ops
    ops.set_compute_dtype("f16")
    c = ops.ConvCfg(32, 32, 3, pad=1)
    x = torch.ones(2, 32, 16, device="cuda"); x[0, 3, 5] = 1e6
    w = torch.full((32, 32, 3), 0.01, device="cuda")
    wf, wb = ops.prep_weights(w, c, None, None)
    y = ops.conv1d_fwd(x, w, None, None, None, c, wf)
    assert ops.LAST_CONV_KERNEL in ("bf16", "tile")
    assert torch.isinf(y[0, :, 4:7]).all() and torch.isfinite(y[1]).all()
