"""ctypes binding of libstegan_hip.so (the C ABI declared in include/stegan_hip.h).

There is deliberately NO fallback: if the shared library is missing or a kernel returns a
non-zero status, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STEGAN_HIP_LIB", os.path.join(_HERE, "libstegan_hip.so"))   # override: diagnostic builds only
# the same sources built with fp16 as the matrix-core operand type (csrc/common.h: STEGAN_OPERAND_F16) -- same C ABI
LIB_PATHS = {"bf16": LIB_PATH, "f16": os.environ.get("STEGAN_HIP_LIB_F16", os.path.join(_HERE, "libstegan_hip_f16.so"))}
ABI_VERSION = 1

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "B", "Cin", "Cout", "groups", "K", "stride", "dilation", "pad", "Tin", "Tout",
        "upsample", "in_act", "out_act", "out_btc")]


class Corpus(C.Structure):
    """Mirrors struct stegan_corpus."""
    _fields_ = [(n, C.c_void_p) for n in ("emg", "emg_row_off", "units", "phonemes", "frame_off", "mfcc", "mfcc_row_off",
                                          "session_idx", "mode_idx")] + \
               [(n, C.c_int32) for n in ("n_utts", "emg_channels", "unit_dim", "mfcc_dim")]


_P = C.c_void_p
_I32 = C.c_int32
_I64 = C.c_int64
_D = C.POINTER(ConvDesc)

# name -> argtypes; every symbol declared in include/stegan_hip.h must be listed here
SIGNATURES = {
    "stegan_abi_version": [],
    "stegan_operand_type": [],
    "stegan_selftest_host": [],
    "stegan_conv1d_fwd": [_D, _P, _P, _P, _P, _P, _P, _P],
    "stegan_conv1d_bwd_data": [_D, _P, _P, _P, _P, _P, _P, _P],
    "stegan_conv1d_bwd_weight": [_D, _P, _P, _P, _P, _P],
    "stegan_conv1d_bwd_weight_ws": [_D, _P, _P, _P, _P, _P, _I64, _P],
    "stegan_conv_prep": [_D, _P, _P, _P, _P, _P],
    "stegan_conv_prep_multi": [_I32, _P, _I32, _P],
    "stegan_conv1d_fwd_t": [_D, _P, _P, _P, _P, _P, _P, _P, _I64, _P],
    "stegan_conv1d_bwd_data_t": [_D, _P, _P, _P, _P, _P, _P, _P, _I64, _P],
    "stegan_conv_prep_bf16": [_D, _P, _P, _P, _P, _P],
    "stegan_conv_prep_bf16_multi": [_I32, _P, _I32, _P],
    "stegan_conv1d_fwd_h": [_D, _P, _P, _P, _P, _P, _P, _P, _I64, _P],
    "stegan_conv1d_bwd_data_h": [_D, _P, _P, _P, _P, _P, _P, _P, _I64, _P],
    "stegan_conv1d_bwd_weight_h": [_D, _P, _P, _P, _P, _P, _I64, _P],
    "stegan_conv1d_fwd_multi_h": [_I32, _P, _P, _P, _P, _P, _P, _P, _P],
    "stegan_conv1d_fwd_hx": [_D, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P],
    "stegan_conv1d_bwd_data_hx": [_D, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P],
    "stegan_timestamp": [_P, _P],
    "stegan_kernel_timing": [_I32],
    "stegan_kernel_timing_collect": [_P, _I64],
    "stegan_adamw_step_z": [_P, _P, _P, _P, _I64, _P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _I32, _P],
    "stegan_wn_backward_z": [_I32, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _P],
    "stegan_act_bwd_add_img": [_P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _P],
    "stegan_act_bwd_add_multi_img": [_I32, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _P],
    "stegan_fold_upsample2_img": [_P, _P, _P, _I64, _I32, _I32, _P],
    "stegan_relpos_attention_fwd_img": [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P],
    "stegan_relpos_attention_bwd_img": [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P],
    "stegan_add_layernorm_fwd_img": [_P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, C.c_float, _P],
    "stegan_add_layernorm_bwd_img": [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P],
    "stegan_conv1d_bwd_weight_hx": [_D, _P, _P, _P, _P, _P, _P, _P, _I64, _P],
    "stegan_conv1d_bwd_weight_hxd": [_D, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P],
    "stegan_conv1d_bwd_weight_hxc": [_D, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P],
    "stegan_conv1d_bwd_weight_multi_hxc": [_I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _P, _P, _P],
    "stegan_wgrad_reduce_multi": [_I32, _P, _P],
    "stegan_conv1d_fwd_multi_hx": [_I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "stegan_conv1d_bwd_data_multi_hx": [_I32, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "stegan_conv1d_bwd_data_multi_h": [_I32, _P, _P, _P, _P, _P, _P, _P],
    "stegan_bias_grad": [_P, _P, _I32, _I32, _I32, _P],
    "stegan_conv1d_plan": [_D, _I32, _P, _P, _P],
    "stegan_wn_scales": [_I32, _P, _P, _P, _P, _P, _I32, _P],
    "stegan_wn_backward": [_I32, _P, _P, _P, _P, _P, _P, _P, _I32, _P],
    "stegan_wn_scale1": [_P, _P, _P, _I32, _I32, _P],
    "stegan_wn_backward1": [_P, _P, _P, _P, _P, _I32, _I32, _P],
    "stegan_spectral_norm_fwd": [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P],
    "stegan_spectral_norm_fwd_multi": [_I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _P],
    "stegan_spectral_norm_bwd": [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _P],
    "stegan_spectral_norm_bwd_acc": [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P],
    "stegan_spectral_norm_bwd_acc_multi": [_I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _P],
    "stegan_mpd_fold_fwd": [_P, _P, _I32, _I32, _I32, _I32, _P],
    "stegan_mpd_fold_bwd": [_P, _P, _I32, _I32, _I32, _I32, _I32, _P],
    "stegan_mpd_fold_fwd_multi": [_P, _P, _I32, _I32, _I32, _P, _I32, _P],
    "stegan_mpd_fold_bwd_multi": [_P, _P, _I32, _I32, _I32, _P, _I32, _I32, _P],
    "stegan_btc_to_bct": [_P, _P, _I32, _I32, _I32, _P],
    "stegan_bct_to_btc": [_P, _P, _I32, _I32, _I32, _I32, _P],
    "stegan_avgpool421_fwd": [_P, _P, _I64, _I32, _P],
    "stegan_avgpool421_bwd": [_P, _P, _P, _I64, _I32, _P],
    "stegan_pair_loss_multi_ex": [_I32, _P, _P, _P, _P, _P, _P, _P, _I32, _P, _P, _P],
    "stegan_pair_loss_multi_det": [_I32, _P, _P, _P, _P, _P, _P, _P, _I32, _P, _P, _P, _I64, _P],
    "stegan_pair_loss_multi": [_I32, _P, _P, _P, C.c_float, _I32, C.c_float, _P, _P, _P],
    "stegan_pair_loss_fwd_bwd": [_P, _P, C.c_float, _I64, _I32, C.c_float, _P, _P, _P],
    "stegan_scale_by_scalar": [_P, _P, _P, _I64, _P],
    "stegan_multi_td_fwd_bwd": [_P, _P, _I32, _I32, _I32, _I32, _P, _P, C.c_float, _P, _P, _P, _P, _P],
    "stegan_multi_td_fwd_bwd_ex": [_P, _P, _I32, _I32, _I32, _I32, _P, _P, _I32, _I32, C.c_float, _P, _P, _P, _P, _P, _I64, _P],
    "stegan_td_features": [_P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _P],
    "stegan_average_filter_fwd": [_P, _P, _I64, _I32, _I32, _I32, _P],
    "stegan_average_filter_bwd": [_P, _P, _I64, _I32, _I32, _I32, _P],
    "stegan_su_loss_fwd_bwd": [_P, _P, _I32, _I32, C.c_float, _P, _P, _P, _P],
    "stegan_phoneme_ce_fwd_bwd": [_P, _P, _I32, _I32, _I32, C.c_float, _P, _P, _P, _P, _P, _P],
    "stegan_relpos_attention_fwd": [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P],
    "stegan_relpos_attention_bwd": [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P],
    "stegan_add_layernorm_fwd": [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, C.c_float, _P],
    "stegan_add_layernorm_bwd": [_P, _P, _P, _P, _P, _I32, _I32, _I32, _P],
    "stegan_adamw_step": [_P, _P, _P, _P, _I64, _P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _P],
    "stegan_conv1d_fwd_multi": [_I32, _P, _P, _P, _P, _P, _P, _P, _P],
    "stegan_conv1d_bwd_data_multi": [_I32, _P, _P, _P, _P, _P, _P, _P],
    "stegan_assemble_batch": [C.POINTER(Corpus), _P, _P, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _P],
    "stegan_adamw_advance": [_P, C.c_double, C.c_double, _P],
    "stegan_sum_n": [_I32, _P, _P, _I64, _P],
    "stegan_loss_grad_scale": [_P],
    "stegan_grad_scaler_check": [_P, _I64, _P, _P],
    "stegan_grad_scaler_update": [_P, C.c_float, C.c_float, _I32, _P],
    "stegan_adamw_advance_s": [_P, C.c_double, C.c_double, _P, _P],
    "stegan_adamw_step_zs": [_P, _P, _P, _P, _I64, _P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _I32, _P, _P],
    "stegan_conv1d_fwd_skinny": [_P, _P, _P, _P, _P, _P, _P, _P],
    "stegan_conv1d_bwd_data_skinny": [_D, _P, _P, _P, _P, _P, _P, _P, _P],
    "stegan_conv1d_bwd_weight_skinny": [_D, _P, _P, _P, _P, _P, _I64, _P],
    "stegan_conv1d_fwd_skinny_multi": [_I32, _P, _P, _P, _P, _P, _P, _P],
    "stegan_conv1d_bwd_data_skinny_multi": [_I32, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "stegan_conv1d_bwd_weight_skinny_multi": [_I32, _P, _P, _P, _P, _P, _P, _I64, _P],
    "stegan_act_bwd": [_P, _P, _P, _I64, _I32, _P],
    "stegan_act_bwd_add": [_P, _P, _P, _P, _I64, _I32, _P],
    "stegan_act_bwd_add_multi": [_I32, _P, _P, _P, _P, _P, _I32, _P],
    "stegan_fold_upsample2": [_P, _P, _I64, _I32, _P],
    "stegan_scalar_combine": [_I32, _P, _I32, _P, _P, _P],
    "stegan_embed_concat_fwd": [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P],
    "stegan_embed_concat_bwd": [_P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P],
}

# symbols whose return value is not a status code
INT64_RETURN = {"stegan_spectral_norm_workspace_floats": [_I32, _I32],
                "stegan_multi_td_workspace_floats": [_I32, _I32],
                "stegan_multi_td_scratch_floats": [_I32, _I32, _I32, _I32, _P, _P, _I32],
                "stegan_conv_prep_floats": [_D, _I32],
                "stegan_conv_prep_bf16_elems": [_D, _I32],
                "stegan_conv_prep_bf16_tiles": [_D],
                "stegan_conv_workspace_floats": [_D, _I32],
                "stegan_conv1d_bwd_weight_skinny_floats": [_D],
                "stegan_kernel_timing_count": []}

_lib = None
_libs = {}
_operand = "bf16"


class SteganHipError(RuntimeError):
    pass


def select_operand_type(kind: str) -> None:
    """Which build of the library ``lib()`` hands out: "bf16" (libstegan_hip.so; also serves the fp32 parity path, whose
    kernels do not depend on the operand type) or "f16" (libstegan_hip_f16.so).  ops.set_compute_dtype calls this."""
    global _lib, _operand
    if kind not in LIB_PATHS:
        raise ValueError(f"operand type must be one of {sorted(LIB_PATHS)}, got {kind!r}")
    _operand = kind
    _lib = _libs.get(kind)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = LIB_PATHS[_operand]
        if not os.path.exists(path):
            raise SteganHipError(
                f"{path} is missing: build it with `python -m ste_gan_amd.build` "
                "(ste_gan_amd has no CPU / eager fallback)")
        handle = C.CDLL(path)
        for name, args in SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError if the .so does not export it
            fn.argtypes = args
            fn.restype = C.c_int
        for name, args in INT64_RETURN.items():
            fn = getattr(handle, name)
            fn.argtypes = args
            fn.restype = C.c_int64
        if handle.stegan_abi_version() != ABI_VERSION:
            raise SteganHipError(f"{path}: ABI version mismatch: rebuild")
        handle.stegan_operand_type.restype = C.c_char_p
        if handle.stegan_operand_type().decode() != _operand:
            raise SteganHipError(f"{path} was built for {handle.stegan_operand_type().decode()} operands, not {_operand}: rebuild")
        _lib = _libs[_operand] = handle
    return _lib


def check(status: int, what: str) -> None:
    if status == 0:
        return
    if status == -1:
        # same class the reference surfaces for bad shapes (ATen raises RuntimeError)
        raise RuntimeError(f"{what}: invalid argument or unsupported shape")
    raise SteganHipError(f"{what}: HIP error {status}")
