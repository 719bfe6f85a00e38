"""Thin host wrappers + autograd glue over the C ABI of libstegan_hip.so.

PyTorch is used for device memory, streams and autograd bookkeeping only; all arithmetic of
the hot path runs in the hand-written gfx950 kernels.  Every wrapper fails loudly when a
tensor is not on the GPU or the library is missing -- there is no eager / CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, replace
from typing import Optional

import torch

from . import _lib
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH, ConvDesc, check

Tensor = torch.Tensor


@dataclass(frozen=True)
class ConvCfg:
    """Static configuration of one fused conv launch (mirrors stegan_conv_desc)."""
    cin: int
    cout: int
    k: int
    stride: int = 1
    dilation: int = 1
    pad: int = 0
    groups: int = 1
    upsample: int = 1
    in_act: int = ACT_NONE
    out_act: int = ACT_NONE
    out_btc: bool = False

    def tout(self, tin: int) -> int:
        span = tin * self.upsample + 2 * self.pad - self.dilation * (self.k - 1) - 1
        if span < 0:
            raise RuntimeError(
                f"Calculated padded input size per channel: ({tin * self.upsample + 2 * self.pad}). "
                f"Kernel size: ({self.dilation * (self.k - 1) + 1}). Kernel size can't be greater than actual input size")
        return span // self.stride + 1


def _p(t: Optional[Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _gpu_f32(t: Tensor, name: str) -> Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"ste_gan_amd: '{name}' lives on {t.device}; the HIP kernels need a cuda (MI355X) tensor "
                           "and there is no CPU fallback")
    if t.dtype != torch.float32:
        raise RuntimeError(f"ste_gan_amd: '{name}' must be float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _desc(cfg: ConvCfg, B: int, tin: int, tout: int) -> ConvDesc:
    return ConvDesc(B, cfg.cin, cfg.cout, cfg.groups, cfg.k, cfg.stride, cfg.dilation, cfg.pad, tin, tout,
                    cfg.upsample, cfg.in_act, cfg.out_act, int(cfg.out_btc))


# ---------------------------------------------------------------------------------------------
# optional per-launch timing (bench.py): HIP events on the stream the kernel is launched on
# ---------------------------------------------------------------------------------------------
class ConvProfiler:
    """Collects (kernel instantiation, nominal MACs, live MACs, start/end event) for every conv launch."""

    def __init__(self, kernel_timestamps: bool = True):
        self.records = []
        # durations from the dispatches' own begin / end timestamps (stegan_kernel_timing, bf16 conv and weight-gradient
        # kernels) beside the event brackets: record i owns the pairs [first, first + n) of the library's list
        self.kernel_timestamps = kernel_timestamps
        self.pairs = []
        if kernel_timestamps:
            check(_lib.lib().stegan_kernel_timing(1), "stegan_kernel_timing")

    def close(self):
        if self.kernel_timestamps:
            check(_lib.lib().stegan_kernel_timing(0), "stegan_kernel_timing")
            self.kernel_timestamps = False

    def plan(self, d: ConvDesc, mode: int):
        if d.in_act == ACT_LRELU:              # (input gradient that also applies the producer's lrelu': same work, plain desc)
            d = ConvDesc(*[getattr(d, f) for f, _ in ConvDesc._fields_])
            d.in_act = ACT_NONE
        tile, splits, live = C.c_int32(), C.c_int32(), C.c_int64()
        check(_lib.lib().stegan_conv1d_plan(C.byref(d), mode, C.byref(tile), C.byref(splits), C.byref(live)),
              "stegan_conv1d_plan")
        nominal = d.B * d.Tout * d.Cout * (d.Cin // d.groups) * d.K
        return f"conv_igemm_kernel<{mode}, {tile.value // 64}, {tile.value // 64}>", nominal, live.value

    def kernel_ms(self):
        """Per record: the summed kernel-timestamp durations of its timed launches (ms), or None where it had none."""
        n = _lib.lib().stegan_kernel_timing_count()
        buf = (C.c_float * max(1, n))()
        check(_lib.lib().stegan_kernel_timing_collect(buf, n), "stegan_kernel_timing_collect")
        return [sum(buf[f:f + c]) if c and f + c <= n else None for f, c in self.pairs]

    def summary(self):
        torch.cuda.synchronize()
        kms = self.kernel_ms() if self.pairs else []
        self.close()
        agg = {}
        for i, (name, nominal, live, e0, e1, *rest) in enumerate(self.records):
            a = agg.setdefault(name, dict(launches=0, ms=0.0, nominal_macs=0, live_macs=0, alg_bytes=0, kernel_ms=0.0,
                                          kernel_timed=0))
            a["launches"] += 1
            a["ms"] += e0.elapsed_time(e1)
            if i < len(kms) and kms[i] is not None:
                a["kernel_ms"] += kms[i]
                a["kernel_timed"] += 1
            a["nominal_macs"] += nominal
            a["live_macs"] += live
            if rest:
                a["alg_bytes"] += self.alg_bytes(rest[0])
        return agg

    @staticmethod
    def alg_bytes(shape) -> int:
        """Bytes a launch must move if every operand is read once and every result written once, as this path stores them:
        activations and their gradients read as bf16 images (2 B; 4 B in fp32 mode), weights 2 B (4 B), results written as
        fp32 plus -- in bf16 mode -- the bf16 image for the next layer (6 B); weight gradients fp32.  An input gradient with a
        fused activation derivative (in_act != 0) also reads the fp32 tensor the mask is taken from (4 B per element; VERDICT
        r04 item 6: that operand was missing here).  Parked feature-matching / hand-over gradients added in the epilogue are
        not visible in the descriptor and stay uncounted."""
        mode, B, Cin, Cout, K, stride, dil, groups, up, Tin, Tout = shape[:11]
        in_act = shape[12] if len(shape) > 12 else 0
        e = 2 if HALF else 4
        o = 6 if HALF else 4
        w, x, y = Cout * (Cin // groups) * K, B * Cin * Tin, B * Cout * Tout
        if mode == 0:
            return e * (x + w) + o * y
        if mode == 1:
            return e * (y + w) + o * x + (4 * x if in_act else 0)
        return e * (x + y) + 4 * w


PROFILER: Optional[ConvProfiler] = None

LAST_CONV_KERNEL = ""    # "bf16" | "direct" | "igemm": which kernel family took the most recent conv launch (tests)


OK_TILE = 2              # STEGAN_OK_TILE: success, the launch went to the one-tile-per-CU kernels (conv_tile*.hip)


def _timed(d: ConvDesc, mode: int, fn, kind: str = "igemm"):
    """Runs one conv launch (``fn`` returns the C status), records it for the profiler when one is on, and returns the status
    with the informational STEGAN_OK_TILE folded into 0 (LAST_CONV_KERNEL / the profiler's family name say "tile" then)."""
    global LAST_CONV_KERNEL
    if PROFILER is None:
        rc = fn()
        if rc != 1:
            LAST_CONV_KERNEL = "tile" if rc == OK_TILE else kind
        return 0 if rc == OK_TILE else rc
    name, nominal, live = PROFILER.plan(d, mode)
    if kind == "direct":
        name = "conv_direct_kernel" + ("<fwd>" if mode == 0 else "<bwd_data>")
    elif kind == "bf16":
        name = "conv_wgrad_bf16_kernel" if mode == 2 else "conv_bf16_kernel" + ("<fwd>" if mode == 0 else "<bwd_data>")
    elif kind == "skinny":
        name = ("conv_cout1_fwd_kernel", "conv_cout1_bwd_data_kernel", "conv_cout1_wgrad_kernel")[mode]
    elif mode == 2:
        name = "bwd_weight (conv_wgrad_kernel | conv_igemm_kernel<2>)"
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    first = _lib.lib().stegan_kernel_timing_count() if PROFILER.kernel_timestamps else 0
    e0.record()
    rc = fn()
    e1.record()
    if rc == OK_TILE:
        name = "conv_tile_wgrad_kernel" if mode == 2 else "conv_tile_kernel" + ("<fwd>" if mode == 0 else "<bwd_data>")
    if rc != 1:
        LAST_CONV_KERNEL = "tile" if rc == OK_TILE else kind
        PROFILER.pairs.append((first, (_lib.lib().stegan_kernel_timing_count() - first) if PROFILER.kernel_timestamps else 0))
        PROFILER.records.append((name, nominal, live, e0, e1,
                                 (mode, d.B, d.Cin, d.Cout, d.K, d.stride, d.dilation, d.groups, d.upsample, d.Tin, d.Tout,
                                  d.pad, d.in_act, d.out_act)))
    return 0 if rc == OK_TILE else rc

def profiler_by_shape(prof: "ConvProfiler"):
    """(mode, shape) -> [launches, total ms, nominal MACs] for diagnostics."""
    torch.cuda.synchronize()
    agg = {}
    for name, nominal, live, e0, e1, shape in prof.records:
        a = agg.setdefault(shape + (name.split("<")[0].split(" ")[0],), [0, 0.0, 0])
        a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += nominal
    return agg


# ---------------------------------------------------------------------------------------------
# raw launches
# ---------------------------------------------------------------------------------------------
_SPLITK_MAX = 16
_WORKSPACES = {}
_WORKSPACES_RETIRED = []      # outgrown workspaces stay allocated: captured graphs hold their addresses (see _workspace)


def _workspace(device, floats: int) -> Tensor:
    """Split-K scratch shared by all conv launches of a (device, stream): launches are stream ordered, and the
    epilogue kernel that consumes the partial sums is enqueued right behind the conv that produced them.

    A workspace that has to grow is RETIRED, never freed: a captured hipGraph bakes the address of the workspace it was
    recorded with into its kernel nodes, so freeing an outgrown buffer would leave every earlier capture (an earlier kernel
    of the same capture, or the graph of another batch shape in train()) writing its partial sums into memory the allocator
    hands to someone else -- seen as a GPU memory fault in train()'s second graph.  Growth is geometric and capped at
    512 MiB, so a stream retires at most a few buffers."""
    floats = min(floats, 1 << 27)                      # cap at 512 MiB; the kernel lowers the split to fit
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() < floats:
        if ws is not None:
            _WORKSPACES_RETIRED.append(ws)
        size = max(floats, 1 << 22, 2 * ws.numel() if ws is not None else 0)
        ws = torch.empty(min(size, 1 << 27), device=device, dtype=torch.float32)
        _WORKSPACES[key] = ws
    return ws


WEIGHT_EPOCH = 0      # bumped by every FusedAdamW launch: prepared (transposed) weight copies older than this are stale

# Arithmetic type of the convolution operands: "f32" (the parity path: exact fp32 MFMA), "bf16" (BASELINE configs 3 / 4) or
# "f16" (BASELINE config 5, the reference's own autocast dtype): both operands rounded to the 2-byte type while they are
# staged, fp32 accumulation, fp32 activations / master weights / optimiser state -- csrc/conv_bf16.hip, whose operand type is
# a build parameter: "f16" loads libstegan_hip_f16.so (same sources, same C ABI; _lib.select_operand_type).  The reference
# reaches its reduced-precision path through ``cfg.train.mixed_precision`` (torch.autocast, train.py:151,181,204); the
# trainer maps that flag (and ``cfg.train.mixed_precision_dtype``) onto this switch.  fp16's 5-bit exponent needs the
# reference's loss scaling too: train.GradScaler (GradScaler semantics on the device, train.py:151,198-199,266-268).
# Everything named "bf16" below serves both 2-byte types; HALF says one of them is on, HALF_DTYPE is its torch dtype.
COMPUTE_DTYPE = "f32"
HALF = False
HALF_DTYPE = torch.bfloat16


def set_compute_dtype(name: str) -> None:
    global COMPUTE_DTYPE, HALF, HALF_DTYPE
    if name not in ("f32", "bf16", "f16"):
        raise ValueError(f"compute dtype must be 'f32', 'bf16' or 'f16', got {name!r}")
    if name != COMPUTE_DTYPE:
        COMPUTE_DTYPE = name
        HALF = name != "f32"
        if HALF:
            HALF_DTYPE = torch.float16 if name == "f16" else torch.bfloat16
        _lib.select_operand_type(name if HALF else "bf16")     # (the fp32 path's kernels are the same in both builds)
        bump_weight_epoch()            # prepared weight copies carry (or lack) their 2-byte companions per mode


def bf16_enabled() -> bool:
    return HALF


def _bf16_of(w: Optional[Tensor]) -> Optional[Tensor]:
    """The bf16 companion of a prepared fp32 weight copy (attribute ``h``, attached by prep_weights / ConvPack), or None
    in fp32 mode."""
    return getattr(w, "h", None) if (w is not None and HALF) else None


# ---- bf16 activation images (round 3) -------------------------------------------------------------------------------
# In bf16 mode a conv launch also writes its output as a bf16 image in the channel-block-major layout the kernels stage by
# LDS-DMA -- (B, ceil(C / 8), T, 8), attribute ``.h`` of the fp32 tensor -- and takes its input from such an image when the
# tensor carries one: the consumer then moves half the bytes with no loads into registers, no conversion and no LDS stores
# (the producer waves' instruction stream was what bounded the bf16 kernels: DESIGN.md section 6).  The fp32 tensor stays
# the tensor of record (residuals, masks, losses, parity); an image is only ever read by a conv kernel.
IMAGES = os.environ.get("STEGAN_BF16_IMAGES", "1") != "0"
IMG_STATS: Optional[dict] = None      # diagnostics (scripts/image_coverage.py): (mode, had image) -> [launches, shapes]


def _image_of(t: Optional[Tensor], channels: int, length: int) -> Optional[Tensor]:
    """The bf16 image attached to ``t`` when it matches (B, ceil(C/8), T, 8) AND ``t`` has not been written in place since
    the image was attached (autograd's gradient buffers add_ into the tensor they hold when it is the last reference; a
    user's in-place op does the same): the image would then show the old values.  Else None."""
    h = getattr(t, "h", None) if (t is not None and IMAGES and HALF) else None
    if h is None or h.dtype != HALF_DTYPE or not h.is_contiguous():
        return None
    if tuple(h.shape) != (t.shape[0], (channels + 7) // 8, length, 8) or h.device != t.device:
        return None
    if getattr(t, "h_version", None) != _version_of(t):
        return None
    return h


def _version_of(t: Tensor) -> int:
    """In-place version of ``t``; inference tensors (EMGGenerator.generate runs under inference_mode) keep no counter --
    nothing of this package writes into an activation in place, so they count as never modified."""
    return -1 if t.is_inference() else t._version


def _attach_image(t: Tensor, h: Optional[Tensor]) -> None:
    """Hang the bf16 image ``h`` on ``t`` together with the in-place version of ``t`` it mirrors (see _image_of)."""
    if h is not None:
        t.h = h
        t.h_version = _version_of(t)


def _new_image(like: Tensor, channels: int, groups: int, length: int) -> Optional[Tensor]:
    """An empty image for a (B, channels, length) output, or None when the layer does not qualify (whole 8-channel blocks
    per group) or images are off."""
    if not (IMAGES and HALF) or channels % 8 or (channels // groups) % 8 or channels < 16:
        return None
    return torch.empty((like.shape[0], channels // 8, length, 8), device=like.device, dtype=HALF_DTYPE)


# ---- zeroed scratch for one train step ---------------------------------------------------------------------------------
# The step needs ~30 zero-initialised buffers (loss words, counters, the spectral-norm layers' raw weight gradients): one
# torch.zeros() -- one fill launch -- each.  Inside GANTrainStep they are carved from ONE pool that a single fill zeroes at
# the start of the step (zero_pool_begin); outside a step pooled_zeros() is torch.zeros().
#
# Invariants (round 4, after the advisor's review):
#   * everything beyond a buffer's high-water mark ``dirty`` (the most bytes any step has handed out of it since it was
#     last filled whole) is zero; zero_pool_begin() clears [0, dirty) -- so a step that needs MORE than the one before it
#     still gets zeros, whether the earlier step ran eagerly, was captured, or is being replayed;
#   * a captured step bakes the size of that fill into its graph: zero_pool_end() raises when the capture handed out more
#     than its own fill covered (a replay would then reuse bytes nobody clears);
#   * a step that dies leaves the pool through zero_pool_abort(): closed, and marked dirty as a whole;
#   * two buffers alternate between eager steps, so the scalars a step returns (views of its pool) stay valid while the
#     NEXT step runs; a replayed graph owns the buffer it was captured with (static outputs, as with any hipGraph).
#   * ``floor``: what captured steps take of a buffer -- their replays dirty it behind the host's back, any number of times.
_ZPOOL = {"bufs": [None, None], "dirty": [0, 0], "floor": [0, 0], "cur": 0, "off": 0, "active": False, "filled": 0}
ZERO_POOL_BYTES = int(os.environ.get("STEGAN_ZERO_POOL_MB", "48")) << 20


def zero_pool_begin(device) -> None:
    if ZERO_POOL_BYTES <= 0:
        return
    capturing = torch.cuda.is_current_stream_capturing()
    if not capturing:
        _ZPOOL["cur"] ^= 1                     # (a capture keeps the buffer of the eager step before it: its dirty mark
    i = _ZPOOL["cur"]                          #  is that step's usage, which is what the captured step will use)
    buf = _ZPOOL["bufs"][i]
    if buf is None or buf.device != torch.device(device):
        buf = _ZPOOL["bufs"][i] = torch.empty(ZERO_POOL_BYTES, device=device, dtype=torch.uint8)
        _ZPOOL["dirty"][i] = ZERO_POOL_BYTES
    n = min(ZERO_POOL_BYTES, (max(_ZPOOL["dirty"][i], _ZPOOL["floor"][i], 256) + 255) & ~255)
    buf[:n].zero_()
    if not capturing:
        _ZPOOL["dirty"][i] = 0                 # (a captured fill executes at replay time: the mark stays)
    _ZPOOL["off"], _ZPOOL["active"], _ZPOOL["filled"] = 0, True, n


def zero_pool_end() -> None:
    if not _ZPOOL["active"]:
        return
    _ZPOOL["active"] = False
    i, off = _ZPOOL["cur"], _ZPOOL["off"]
    _ZPOOL["dirty"][i] = max(_ZPOOL["dirty"][i], off)
    if torch.cuda.is_current_stream_capturing():
        _ZPOOL["floor"][i] = max(_ZPOOL["floor"][i], off)
    if torch.cuda.is_current_stream_capturing() and off > _ZPOOL["filled"]:
        raise RuntimeError(f"ste_gan_amd: the captured step takes {off} bytes of zeroed scratch but its fill covers "
                           f"{_ZPOOL['filled']}: run the step eagerly once before capturing it")


def zero_pool_abort() -> None:
    """Exception path of a step: close the pool and treat the whole buffer as dirty."""
    if _ZPOOL["active"]:
        _ZPOOL["active"] = False
        _ZPOOL["dirty"][_ZPOOL["cur"]] = ZERO_POOL_BYTES


def pooled_zeros(shape, device, dtype=torch.float32) -> Tensor:
    """torch.zeros(shape) from the step's pre-zeroed pool when one is open (and the request fits), else a fresh fill."""
    shape = tuple(shape) if not isinstance(shape, int) else (shape,)
    n = 1
    for d_ in shape:
        n *= int(d_)
    nbytes = n * torch.empty((), dtype=dtype).element_size()
    off = (_ZPOOL["off"] + 15) & ~15
    buf = _ZPOOL["bufs"][_ZPOOL["cur"]] if _ZPOOL["active"] else None
    if buf is None or buf.device != torch.device(device) or off + nbytes > ZERO_POOL_BYTES:
        return torch.zeros(shape, device=device, dtype=dtype)
    _ZPOOL["off"] = off + nbytes
    return buf[off:off + nbytes].view(dtype).view(shape)


# ---- diagnostics: time marks inside the (captured) step, scripts/step_marks.py -----------------------------------------
MARKS: Optional[dict] = None        # {"buf": int64 tensor, "names": [...]} when switched on


def mark(name: str) -> None:
    """Enqueue a time mark on the current stream (a one-lane kernel storing the 100 MHz real-time counter)."""
    if MARKS is None:
        return
    i = len(MARKS["names"])
    MARKS["names"].append(name)
    check(_lib.lib().stegan_timestamp(C.c_void_p(MARKS["buf"].data_ptr() + 8 * i), _stream()), "stegan_timestamp")


def bump_weight_epoch() -> None:
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1


def prep_weights(w3: Tensor, cfg: ConvCfg, bufs=None, scale: Optional[Tensor] = None):
    """K-major transposed copies (wf for the forward, wb for the input gradient) of a (Cout, Cin/g, K) tensor.
    A per-output-channel ``scale`` (weight-norm / BatchNorm factor) is folded into wb."""
    w3 = _gpu_f32(w3, "weight")
    d = _desc(cfg, 1, cfg.k * cfg.dilation + 1, 1)          # only channel / K / groups fields matter here
    d.Tin = max(1, cfg.dilation * (cfg.k - 1) + 1 - 2 * cfg.pad)
    d.Tout = (d.Tin * cfg.upsample + 2 * cfg.pad - cfg.dilation * (cfg.k - 1) - 1) // cfg.stride + 1
    L = _lib.lib()
    if bufs is None:
        bufs = (torch.zeros(int(L.stegan_conv_prep_floats(C.byref(d), 0)), device=w3.device, dtype=torch.float32),
                torch.zeros(int(L.stegan_conv_prep_floats(C.byref(d), 1)), device=w3.device, dtype=torch.float32))
    # bf16 mode keeps ONLY the bf16 copies current (the fp32 buffers then merely carry them as ``.h``): every conv runs on
    # the bf16 or the single-output-channel kernels, and a shape both decline goes to the implicit-GEMM kernel, which reads
    # the raw tensor -- conv1d_fwd / conv1d_bwd_data never hand the fp32 copies to a kernel in that mode
    if not HALF:
        check(L.stegan_conv_prep(C.byref(d), _p(w3), _p(scale), _p(bufs[0]), _p(bufs[1]), _stream()), "stegan_conv_prep")
    if HALF:
        for which, b in enumerate(bufs):
            if getattr(b, "h", None) is None:
                # zeroed ONCE: the prep kernel writes live elements only (padding rows / channels / the zero tap plane stay 0)
                b.h = torch.zeros(int(L.stegan_conv_prep_bf16_elems(C.byref(d), which)), device=w3.device, dtype=HALF_DTYPE)
        check(L.stegan_conv_prep_bf16(C.byref(d), _p(w3), _p(scale), _p(bufs[0].h), _p(bufs[1].h), _stream()),
              "stegan_conv_prep_bf16")
    return bufs


def bf16_prep_row(cfg: ConvCfg, v: Tensor, scale: Optional[Tensor], wf_h: Tensor, wb_h: Tensor) -> bytes:
    """One entry of the device table stegan_conv_prep_bf16_multi reads (struct PrepH of csrc/conv_bf16.hip)."""
    import struct
    L = _lib.lib()
    d = _desc(cfg, 1, 1, 1)
    nf, nb = int(L.stegan_conv_prep_bf16_elems(C.byref(d), 0)), int(L.stegan_conv_prep_bf16_elems(C.byref(d), 1))
    if wf_h.numel() != nf or wb_h.numel() != nb:
        raise RuntimeError("bf16_prep_row: companion buffers do not match the layer")
    G = cfg.groups
    cog, cig = cfg.cout // G, cfg.cin // G
    up = lambda a, b: (a + b - 1) // b * b
    return struct.pack("<4q8i2q", v.data_ptr(), scale.data_ptr() if scale is not None else 0, wf_h.data_ptr(), wb_h.data_ptr(),
                       cog, cig, cfg.k, G, up(cog, 128), up(cig, 64), up(cig, 128), up(cog, 64), nf, nb)


_PREP_TABLES: dict = {}


def prep_bf16_many(entries) -> None:
    """bf16 copies of several layers in ONE launch: ``entries`` = [(cfg, weight (Cout, Cin/g, K), scale or None, wf, wb)]
    with wf / wb the layers' prepared fp32 buffers carrying their bf16 companions (``.h``).  The device table is cached by
    the tensors' addresses (parameters live in a FlatParams arena: stable)."""
    L = _lib.lib()
    key = tuple((w.data_ptr(), wf.h.data_ptr(), wb.h.data_ptr()) for _, w, _, wf, wb in entries)
    ent = _PREP_TABLES.get(key)
    if ent is None:
        raw = b"".join(bf16_prep_row(cfg, w, sc, wf.h, wb.h) for cfg, w, sc, wf, wb in entries)
        table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(entries[0][1].device)
        tiles = max(int(L.stegan_conv_prep_bf16_tiles(C.byref(_desc(cfg, 1, 1, 1)))) for cfg, *_ in entries)
        ent = (table, max(1, min(4096, tiles)), [t for e in entries for t in (e[1], e[3], e[4])])      # keeps the tensors alive
        _PREP_TABLES[key] = ent
    check(L.stegan_conv_prep_bf16_multi(len(entries), _p(ent[0]), ent[1], _stream()), "stegan_conv_prep_bf16_multi")


def conv1d_fwd(x: Tensor, w: Tensor, scale: Optional[Tensor], bias: Optional[Tensor],
               residual: Optional[Tensor], cfg: ConvCfg, wf: Optional[Tensor] = None) -> Tensor:
    x = _gpu_f32(x, "x")
    w = _gpu_f32(w, "weight")
    if x.dim() != 3 or x.shape[1] != cfg.cin:
        raise RuntimeError(f"Given groups={cfg.groups}, weight of size {list(w.shape)}, expected input[B, {cfg.cin}, T] "
                           f"but got input of size {list(x.shape)}")
    B, _, tin = x.shape
    tout = cfg.tout(tin)
    y = torch.empty((B, tout, cfg.cout) if cfg.out_btc else (B, cfg.cout, tout), device=x.device, dtype=torch.float32)
    if residual is not None:
        residual = _gpu_f32(residual, "residual")
        if residual.shape != y.shape:
            raise RuntimeError(f"residual shape {list(residual.shape)} does not match conv output {list(y.shape)}")
    d = _desc(cfg, B, tin, tout)
    if cfg.cout == 1 and SKINNY:             # single-output-channel convs: streaming VALU kernels (conv_skinny.hip)
        rc = _timed(d, 0, lambda: _lib.lib().stegan_conv1d_fwd_skinny(C.byref(d), _p(x), _p(w), _p(scale), _p(bias),
                                                                      _p(residual), _p(y), _stream()), "skinny")
        if rc != 1:
            check(rc, "stegan_conv1d_fwd_skinny")
            return y
    wfh = _bf16_of(wf)
    if wfh is not None:
        ws = _workspace(x.device, y.numel() * _SPLITK_MAX)
        xh = _image_of(x, cfg.cin, tin)
        if IMG_STATS is not None:
            IMG_STATS.setdefault((0, xh is not None), []).append((B, cfg.cin, cfg.cout, cfg.k, tin))
        yh = None if cfg.out_btc else _new_image(y, cfg.cout, cfg.groups, tout)
        rc = _timed(d, 0, lambda: _lib.lib().stegan_conv1d_fwd_hx(C.byref(d), _p(x), _p(xh), _p(wfh), _p(scale), _p(bias),
                                                                  _p(residual), _p(y), _p(yh), _p(ws), ws.numel(), _stream()),
                    "bf16")
        if rc != 1:
            check(rc, "stegan_conv1d_fwd_hx")
            _attach_image(y, yh)
            return y
    if wf is not None and not HALF:       # (bf16 mode does not maintain the fp32 copies: prep_weights)
        ws = _workspace(x.device, y.numel() * _SPLITK_MAX)
        rc = _timed(d, 0, lambda: _lib.lib().stegan_conv1d_fwd_t(C.byref(d), _p(x), _p(wf), _p(scale), _p(bias),
                                                                 _p(residual), _p(y), _p(ws), ws.numel(), _stream()),
                    "direct")
        if rc != 1:
            check(rc, "stegan_conv1d_fwd_t")
            return y
    check(_timed(d, 0, lambda: _lib.lib().stegan_conv1d_fwd(C.byref(d), _p(x), _p(w), _p(scale), _p(bias), _p(residual),
                                                            _p(y), _stream())), "stegan_conv1d_fwd")
    return y


class ActFuse:
    """Link between a conv whose output goes through leaky_relu (the producer) and the conv that reads that output (the
    consumer): in the backward pass the CONSUMER's input-gradient launch applies the producer's activation derivative --
    and adds the feature-matching gradient of the tensor in front of it -- in its epilogue (conv1d_bwd_data(act_src=...)),
    sets ``done``, and the producer skips its own activation-backward launch.  Set up by the model that knows the
    topology (the discriminator stacks: every feature map has exactly one conv behind it), like GradHandover."""
    __slots__ = ("done",)

    def __init__(self):
        self.done = False


def conv1d_bwd_data(dy: Tensor, w: Tensor, scale: Optional[Tensor], mask_src: Optional[Tensor],
                    add_src: Optional[Tensor], cfg: ConvCfg, tin: int, wb: Optional[Tensor] = None,
                    uniform_scale: bool = False, act_src: Optional[Tensor] = None,
                    act_add: Optional[Tensor] = None) -> Tensor:
    """``wb`` (from prep_weights WITH the per-channel scale folded in) selects the direct kernel; ``uniform_scale``
    says that ``scale`` holds one value repeated (spectral norm) and was NOT folded into wb.
    ``act_src`` (round 4): the conv's input x, which is the leaky_relu(0.1) OUTPUT of the layer before: the result is then
    that layer's PRE-activation gradient, (conv_T(dy) + act_add) * lrelu'(act_src), formed in the kernel's epilogue
    (``act_add``: the feature-matching gradient of x or None); kernels without that epilogue are followed by act_bwd."""
    dy = _gpu_f32(dy, "dy")
    B, _, tout = dy.shape
    dx = torch.empty((B, cfg.cin, tin * cfg.upsample), device=dy.device, dtype=torch.float32)
    d = _desc(cfg, B, tin, tout)
    fuse = act_src is not None
    if fuse:
        if mask_src is not None or add_src is not None or cfg.upsample != 1 or cfg.in_act != ACT_NONE:
            raise RuntimeError("ste_gan_amd.conv1d_bwd_data: act_src excludes mask_src / add_src / upsample / in_act")
        if act_src.shape != dx.shape or (act_add is not None and act_add.numel() != dx.numel()):
            raise RuntimeError("ste_gan_amd.conv1d_bwd_data: act_src / act_add do not match the input gradient")
        dl = _desc(replace(cfg, in_act=ACT_LRELU), B, tin, tout)
        mk, ad = act_src, act_add
    else:
        dl, mk, ad = d, mask_src, add_src
    if cfg.cout == 1 and SKINNY:
        dxh = _new_image(dx, cfg.cin, cfg.groups, tin) if fuse else None
        rc = _timed(dl, 1, lambda: _lib.lib().stegan_conv1d_bwd_data_skinny(C.byref(dl), _p(dy), _p(_gpu_f32(w, "weight")),
                                                                            _p(scale), _p(mk), _p(ad), _p(dx), _p(dxh),
                                                                            _stream()), "skinny")
        if rc != 1:
            check(rc, "stegan_conv1d_bwd_data_skinny")
            _attach_image(dx, dxh)
            return dx
    wbh = _bf16_of(wb)
    if wbh is not None:
        ws = _workspace(dy.device, dx.numel() * _SPLITK_MAX)
        us = scale if uniform_scale else None
        dyh = _image_of(dy, cfg.cout, tout)
        if IMG_STATS is not None:
            IMG_STATS.setdefault((1, dyh is not None), []).append((B, cfg.cin, cfg.cout, cfg.k, tin))
        dxh = _new_image(dx, cfg.cin, cfg.groups, tin * cfg.upsample) if cfg.upsample == 1 else None
        rc = _timed(dl, 1, lambda: _lib.lib().stegan_conv1d_bwd_data_hx(C.byref(dl), _p(dy), _p(dyh), _p(wbh), _p(us),
                                                                        _p(mk), _p(ad), _p(dx), _p(dxh), _p(ws),
                                                                        ws.numel(), _stream()),
                    "bf16")
        if rc != 1:
            check(rc, "stegan_conv1d_bwd_data_hx")
            _attach_image(dx, dxh)
            return dx
    if wb is not None and not HALF:
        ws = _workspace(dy.device, dx.numel() * _SPLITK_MAX)
        us = scale if uniform_scale else None
        rc = _timed(d, 1, lambda: _lib.lib().stegan_conv1d_bwd_data_t(C.byref(d), _p(dy), _p(wb), _p(us), _p(mask_src),
                                                                      _p(add_src), _p(dx), _p(ws), ws.numel(), _stream()),
                    "direct")
        if rc != 1:
            check(rc, "stegan_conv1d_bwd_data_t")
            return act_bwd(dx, act_src, ACT_LRELU, act_add) if fuse else dx
    check(_timed(d, 1, lambda: _lib.lib().stegan_conv1d_bwd_data(C.byref(d), _p(dy), _p(w), _p(scale), _p(mask_src),
                                                                 _p(add_src), _p(dx), _stream())),
          "stegan_conv1d_bwd_data")
    return act_bwd(dx, act_src, ACT_LRELU, act_add) if fuse else dx


class WgradReduce(C.Structure):       # mirrors struct stegan_wgrad_reduce
    _fields_ = [("partial", C.c_void_p), ("dw", C.c_void_p), ("per_tap", C.c_int64), ("used_floats", C.c_int64),
                ("slices", C.c_int32), ("K", C.c_int32), ("k_lo", C.c_int32), ("K_eff", C.c_int32)]


# Deferred slab sums of the sliced weight gradients (sink mode: dW is a view of a gradient arena that nothing reads before
# ConvPack.finalize_grads -> wgrad_join): the slabs of a phase's layers live side by side in one arena and ONE multi-problem
# launch per phase adds them into the arenas (52 reduce launches a step otherwise).  The arena is a fixed buffer per device
# (captured graphs hold its address); a layer that does not fit any more sums its slabs at once, the round-2 way.
# MEASURED AND SWITCHED OFF: 6.97 ms per step against 6.87 with the per-layer reduce launches -- those are spread over the
# backward pass and fill gaps on busy streams, the batched sum sits alone between the last weight gradient and the optimiser.
# Kept as an option (STEGAN_WGRAD_DEFER=1; tests/test_gpu_train_step.py runs the golden step through it).
WGRAD_DEFER = os.environ.get("STEGAN_WGRAD_DEFER", "0") == "1"
# Round 5: inside a batch of weight gradients issued on one side stream (_wgrad_flush) the slab sum of a sliced layer is run by
# the NEXT layer's launch (its consumer waves are idle while its producers set up); only the batch's last layer pays a reduce
# launch.  The slabs live in the arena below (a region per layer, recycled at wgrad_join).
WGRAD_CHAIN = os.environ.get("STEGAN_WGRAD_CHAIN", "1") == "1"
_WG_ARENA_FLOATS = int(os.environ.get("STEGAN_WGRAD_ARENA_MB", "1024")) * (1 << 18)
_WG_ARENA: dict = {}           # device -> [arena, floats in use]
_WG_DEFERRED: list = []        # (WgradReduce, dw tensor kept alive)


def wgrad_chain_end(chain: list) -> None:
    """Sums the pending slabs a chain of conv1d_bwd_weight(..., chain=chain) calls left behind (current stream)."""
    if chain[0] is not None:
        item, _ = chain[0]
        check(_lib.lib().stegan_wgrad_reduce_multi(1, C.byref(item), _stream()), "stegan_wgrad_reduce_multi")
        chain[0] = None


def _wgrad_reduce_flush() -> None:
    """The pending slab sums as one launch on the CURRENT stream (which the caller has ordered behind every weight-gradient
    launch: wgrad_join), and the arena is free again for the launches ordered behind this one."""
    if _WG_DEFERRED:
        arr = (WgradReduce * len(_WG_DEFERRED))(*[it for it, _ in _WG_DEFERRED])
        check(_lib.lib().stegan_wgrad_reduce_multi(len(_WG_DEFERRED), arr, _stream()), "stegan_wgrad_reduce_multi")
        _WG_DEFERRED.clear()
    for st in _WG_ARENA.values():
        st[1] = 0


def conv1d_bwd_weight(x: Tensor, dy: Tensor, cfg: ConvCfg, dw: Tensor, db: Optional[Tensor], defer: bool = False,
                      chain: Optional[list] = None) -> None:
    """Accumulates into dw (and db).  ``defer``: the caller promises that dw is read only after ops.wgrad_join() (gradient
    sinks); the sum over the batch slices may then be postponed to that call.  ``chain``: a one-element list owned by a caller
    that issues several weight gradients on ONE stream (_wgrad_flush): it carries the pending slab sum of the launch before,
    which this launch runs at its head (stegan_conv1d_bwd_weight_hxc), and receives this launch's; the caller sums the tail."""
    global LAST_CONV_KERNEL
    B, _, tin = x.shape
    d = _desc(cfg, B, tin, dy.shape[2])
    if cfg.cout == 1 and SKINNY:
        ws = _workspace(x.device, int(_lib.lib().stegan_conv1d_bwd_weight_skinny_floats(C.byref(d))))
        rc = _timed(d, 2, lambda: _lib.lib().stegan_conv1d_bwd_weight_skinny(C.byref(d), _p(x), _p(dy), _p(dw), _p(db),
                                                                             _p(ws), ws.numel(), _stream()), "skinny")
        if rc != 1:
            check(rc, "stegan_conv1d_bwd_weight_skinny")
            return
    if HALF:
        xh, dyh = _image_of(x, cfg.cin, tin), _image_of(dy, cfg.cout, dy.shape[2])
        if IMG_STATS is not None:
            IMG_STATS.setdefault((2, xh is not None and dyh is not None), []).append((B, cfg.cin, cfg.cout, cfg.k, tin))
        if chain is not None and defer and PROFILER is None and x.is_cuda:
            st = _WG_ARENA.get(x.device)
            if st is None:
                st = _WG_ARENA[x.device] = [torch.empty(_WG_ARENA_FLOATS, device=x.device, dtype=torch.float32), 0]
            arena, used = st
            room = arena.numel() - used
            prev = chain[0]
            if room >= dw.numel() * 8:
                item = WgradReduce()
                ws = arena[used:]
                rc = _lib.lib().stegan_conv1d_bwd_weight_hxc(C.byref(d), _p(x), _p(xh), _p(dy), _p(dyh), _p(dw), _p(db), _p(ws),
                                                             min(room, dw.numel() * 64), C.byref(prev[0]) if prev else None,
                                                             C.byref(item), _stream())
                if rc != 1:
                    LAST_CONV_KERNEL = "tile" if rc == OK_TILE else "bf16"
                    check(0 if rc == OK_TILE else rc, "stegan_conv1d_bwd_weight_hxc")
                    chain[0] = None
                    if item.slices > 0:
                        st[1] = used + ((int(item.used_floats) + 63) & ~63)
                        chain[0] = (item, dw)
                    return
            wgrad_chain_end(chain)                     # declined (or no room): the pending sum runs on its own, then the usual path
        if defer and WGRAD_DEFER and PROFILER is None and x.is_cuda:
            st = _WG_ARENA.get(x.device)
            if st is None:
                st = _WG_ARENA[x.device] = [torch.empty(_WG_ARENA_FLOATS, device=x.device, dtype=torch.float32), 0]
            arena, used = st
            room = arena.numel() - used
            if room >= dw.numel() * 8:                    # room for at least 8 slices: worth a deferred launch
                item = WgradReduce()
                ws = arena[used:]
                rc = _lib.lib().stegan_conv1d_bwd_weight_hxd(C.byref(d), _p(x), _p(xh), _p(dy), _p(dyh), _p(dw), _p(db),
                                                             _p(ws), min(room, dw.numel() * 64), C.byref(item), _stream())
                if rc != 1:
                    LAST_CONV_KERNEL = "tile" if rc == OK_TILE else "bf16"
                    check(0 if rc == OK_TILE else rc, "stegan_conv1d_bwd_weight_hxd")
                    if item.slices > 0:
                        st[1] = used + ((int(item.used_floats) + 63) & ~63)
                        _WG_DEFERRED.append((item, dw))
                    return
        ws = _workspace(x.device, dw.numel() * 64)            # slabs for the batch slices of the reduction
        rc = _timed(d, 2, lambda: _lib.lib().stegan_conv1d_bwd_weight_hx(C.byref(d), _p(x), _p(xh), _p(dy), _p(dyh), _p(dw),
                                                                         _p(db), _p(ws), ws.numel(), _stream()), "bf16")
        if rc != 1:
            check(rc, "stegan_conv1d_bwd_weight_hx")
            return
    ws = _workspace(x.device, dw.numel() * 64)            # slabs for the batch slices of the reduction
    check(_timed(d, 2, lambda: _lib.lib().stegan_conv1d_bwd_weight_ws(C.byref(d), _p(x), _p(dy), _p(dw), _p(db), _p(ws),
                                                                      ws.numel(), _stream())),
          "stegan_conv1d_bwd_weight_ws")


# ---- weight gradients beside the input-gradient chain (generator backward only) -------------------------------------
# In a backward pass the input gradient of layer L feeds layer L - 1, the weight gradient of layer L feeds nothing until
# the optimiser.  With the gradients accumulated into a ConvPack arena ("sink" mode) the weight-gradient launches can go
# to a side stream while the chain of input gradients runs on.  Measured: a loss when other streams are busy (the
# discriminator step: 11.5-11.9 vs 11.2 ms/step), so it is switched on only around the generator's backward pass, where
# nothing else is in flight (wgrad_async(True / False), train.GANTrainStep).  The operands (x, dy) stay referenced until
# wgrad_join() -- called by ConvPack.finalize_grads -- has made the consuming stream wait for the side streams.
WGRAD_STREAMS = int(os.environ.get("STEGAN_WGRAD_STREAMS", "4"))     # side streams (round-robin); 0: never.  Measured 10.0 / 9.73 / 9.40 / 9.77 / 9.26 / 9.41 ms per step with 0 / 1 / 2 / 3 / 4 / 6
_WG_ON = [False]
_WG_SIDE: list = []
_WG_KEEP: list = []
_WG_NEXT = [0]


def wgrad_async(on: bool) -> None:
    _WG_ON[0] = bool(on) and WGRAD_STREAMS > 0 and (HALF or F32_STREAMS)


F32_STREAMS = os.environ.get("STEGAN_F32_STREAMS", "1") == "1"      # the discriminator fork and the generator's weight-gradient
                                                                     # streams in fp32 mode too: 19.2 -> 17.1 ms/step
WGRAD_BATCH = int(os.environ.get("STEGAN_WGRAD_BATCH", "5"))        # weight gradients launched per fork (8.14 ms/step; 8.30 with 1, 8.21 with 2, 8.17 with 3, 8.16 with 4, 8.20 with 8)
_WG_PENDING: list = []
_WG_USED: dict = {}            # side streams forked since the last wgrad_join() (only those are joined: waiting on a
                               # stream that took no part in the current capture would be a capture-isolation error)


# Round 5: the weight gradients of a batch (a generator block's five convs) as ONE launch of the tile kernel
# (stegan_conv1d_bwd_weight_multi_hxc) + ONE launch for their slab sums, when every layer of the batch is a shape for it.
WGRAD_MULTI = os.environ.get("STEGAN_WGRAD_MULTI", "1") == "1"
_WG_STREAM_PENDING: dict = {}       # stream -> [(WgradReduce, dw kept alive)]: slab sums waiting for the next launch on that stream


def _wgrad_multi(batch) -> bool:
    """True: the batch went out as one launch (and one reduce).  False: nothing was enqueued; go layer by layer."""
    global LAST_CONV_KERNEL
    n = len(batch)
    if not (HALF and 2 <= n <= 5):
        return False
    dev = batch[0][0].device
    st = _WG_ARENA.get(dev)
    if st is None:
        st = _WG_ARENA[dev] = [torch.empty(_WG_ARENA_FLOATS, device=dev, dtype=torch.float32), 0]
    arena, used = st
    descs, xs, xhs, dys, dyhs, dws, dbs, wss, wsf = [], [], [], [], [], [], [], [], []
    off = used
    for x, dy, cfg, dw, db in batch:
        if cfg.stride != 1 or cfg.groups != 1 or not x.is_cuda:
            return False
        B, _, tin = x.shape
        xh, dyh = _image_of(x, cfg.cin, tin), _image_of(dy, cfg.cout, dy.shape[2])
        if xh is None or dyh is None:
            return False
        need = dw.numel() * 4                                   # room for four slices (the plan aims at two)
        if off + need > arena.numel():
            return False
        descs.append(_desc(cfg, B, tin, dy.shape[2]))
        xs.append(x); xhs.append(xh); dys.append(dy); dyhs.append(dyh); dws.append(dw); dbs.append(db)
        wss.append(arena[off:off + need]); wsf.append(need)
        off += (need + 63) & ~63
    # the slab sums the previous launch ON THIS STREAM left pending ride in this launch's prologue; this launch's own wait for
    # the next one on the stream, or for wgrad_join (one launch for all that are left)
    sid = torch.cuda.current_stream().cuda_stream
    prev = _WG_STREAM_PENDING.pop(sid, [])
    parr = (WgradReduce * max(1, len(prev)))(*[it for it, _ in prev])
    nxt = (WgradReduce * n)()
    rc = _lib.lib().stegan_conv1d_bwd_weight_multi_hxc(
        n, (C.POINTER(ConvDesc) * n)(*[C.pointer(d) for d in descs]), _ptr_array(xs), _ptr_array(xhs), _ptr_array(dys),
        _ptr_array(dyhs), _ptr_array(dws), _ptr_array(dbs), _ptr_array(wss), (C.c_int64 * n)(*wsf), len(prev),
        parr if prev else None, nxt, _stream())
    if rc == 1:
        if prev:
            _WG_STREAM_PENDING[sid] = prev                      # untouched: still pending
        return False
    check(0 if rc == OK_TILE else rc, "stegan_conv1d_bwd_weight_multi_hxc")
    LAST_CONV_KERNEL = "tile"
    st[1] = off
    items = [(WgradReduce.from_buffer_copy(nxt[i]), dws[i]) for i in range(n) if nxt[i].slices > 0]
    if items:
        _WG_STREAM_PENDING[sid] = items
    return True


def _wgrad_flush() -> None:
    if not _WG_PENDING:
        return
    while len(_WG_SIDE) < WGRAD_STREAMS:
        _WG_SIDE.append(torch.cuda.Stream())
    side = _WG_SIDE[_WG_NEXT[0] % WGRAD_STREAMS]
    _WG_NEXT[0] += 1
    _WG_USED[side.cuda_stream] = side
    side.wait_stream(torch.cuda.current_stream())       # ONE cross-queue edge for the whole batch (11 us each inside a
    with torch.cuda.stream(side):                       # captured graph: scripts/graph_edge_cost.py)
        for i in range(0, len(_WG_PENDING), 5):          # (a multi-problem launch takes five layers)
            part = _WG_PENDING[i:i + 5]
            # layers the tile kernel cannot take (the generator's 8-channel output conv) must not cost the others their launch
            wide = [t for t in part if min(t[2].cin, t[2].cout) >= 32 and t[2].stride == 1 and t[2].groups == 1]
            if WGRAD_MULTI and len(wide) >= 2 and _wgrad_multi(wide):
                part = [t for t in part if not any(t is w for w in wide)]
            chain = [None] if WGRAD_CHAIN else None         # the slab sum of layer i rides in the launch of layer i + 1
            for x, dy, cfg, dw, db in part:
                conv1d_bwd_weight(x, dy, cfg, dw, db, defer=True, chain=chain)
            if chain is not None:
                wgrad_chain_end(chain)
    _WG_KEEP.extend((x, dy) for x, dy, _, _, _ in _WG_PENDING)
    _WG_PENDING.clear()


def _wgrad_sink(x: Tensor, dy: Tensor, cfg: ConvCfg, dw: Tensor, db: Optional[Tensor]) -> None:
    if not _WG_ON[0] or PROFILER is not None or not x.is_cuda:
        conv1d_bwd_weight(x, dy, cfg, dw, db, defer=True)
        return
    _WG_PENDING.append((x, dy, cfg, dw, db))
    if len(_WG_PENDING) >= WGRAD_BATCH:
        _wgrad_flush()


_SINK_STREAMS: dict = {}       # streams on which sink-mode gradients were written since the last wgrad_join()


def _sink_touch() -> None:
    """A backward node is about to write parameter gradients straight into an arena (sink mode) on the current stream.
    autograd knows nothing about those writes -- they do not pass through AccumulateGrad -- so when a sub-graph ran on a side
    stream (the period branch of the discriminators) nothing makes the stream that calls the optimiser wait for them: the
    consumer must (wgrad_join, called by ConvPack.finalize_grads).  Found as NaN losses once the side branch had become the
    longer one."""
    s_ = torch.cuda.current_stream()
    _SINK_STREAMS[s_.cuda_stream] = s_


def wgrad_join() -> None:
    """The calling stream waits for every gradient written in sink mode: the weight gradients launched aside
    (_wgrad_sink) and the backward nodes that ran on other streams (_sink_touch)."""
    _wgrad_flush()
    cur = torch.cuda.current_stream()
    for s_ in _WG_USED.values():
        cur.wait_stream(s_)
    _WG_USED.clear()
    _WG_KEEP.clear()
    for key, s_ in list(_SINK_STREAMS.items()):
        if key != cur.cuda_stream:
            cur.wait_stream(s_)
    _SINK_STREAMS.clear()
    for items in _WG_STREAM_PENDING.values():       # slab sums no later launch took along: with the deferred ones, one launch
        _WG_DEFERRED.extend(items)
    _WG_STREAM_PENDING.clear()
    _wgrad_reduce_flush()          # behind every weight-gradient launch of the phase: their slabs' sums, one launch
    sn_bwd_flush()                 # ... and the spectral-norm transforms of the raw gradients those launches produced


def wgrad_abort() -> None:
    """Exception path of a train step: forget the weight gradients queued for / running on side streams and the streams to
    join (the device is drained first, so dropping the operand references cannot free memory a queued kernel reads)."""
    if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
        torch.cuda.synchronize()
    _WG_ON[0] = False
    _WG_PENDING.clear(); _WG_KEEP.clear(); _WG_USED.clear(); _SINK_STREAMS.clear(); _WG_DEFERRED.clear()
    _WG_STREAM_PENDING.clear()
    _SN_PENDING.clear()
    for st in _WG_ARENA.values():
        st[1] = 0
    fm_fold_abort()


def wn_scale(v: Tensor, g: Tensor) -> Tensor:
    v = _gpu_f32(v, "weight_v")
    g = _gpu_f32(g, "weight_g")
    rows = v.shape[0]
    scale = torch.empty(rows, device=v.device, dtype=torch.float32)
    check(_lib.lib().stegan_wn_scale1(_p(v), _p(g), _p(scale), rows, v.numel() // rows, _stream()), "stegan_wn_scale1")
    return scale


def fold_upsample2(dxe: Tensor) -> Tensor:
    B, Cc, Te = dxe.shape
    dx = torch.empty((B, Cc, Te // 2), device=dxe.device, dtype=torch.float32)
    dxh = _new_image(dx, Cc, 1, Te // 2)
    check(_lib.lib().stegan_fold_upsample2_img(_p(dxe), _p(dx), _p(dxh), B * Cc, Cc, Te // 2, _stream()),
          "stegan_fold_upsample2_img")
    _attach_image(dx, dxh)
    return dx


# ---------------------------------------------------------------------------------------------
# autograd: weight-normalised fused conv
# ---------------------------------------------------------------------------------------------
# ---- gradient hand-over at a fan-out point inside a block ------------------------------------------------------------
# In a GBlock the block input feeds two convs (the main path and the 1x1 residual conv) and the middle activation feeds a
# conv and, as identity residual, the last conv's epilogue.  Autograd would receive two gradients for each of those tensors
# and add them with a launch of its own (16 a step, on the generator backward's critical path).  Instead the backward
# node that runs FIRST parks its contribution here and returns None, and the node that runs second passes the parked
# tensor to its input-gradient kernel as ``add`` (the kernels add it after the ReLU mask -- exactly the sum autograd would
# form).  Both orders are right: a parker that finds the taker already done returns its gradient the ordinary way.
HANDOVER = os.environ.get("STEGAN_GRAD_HANDOVER", "1") != "0"
PARK_DX, PARK_RES, TAKE = 1, 2, 3


class GradHandover:
    __slots__ = ("parked", "taken")

    def __init__(self):
        self.parked = None
        self.taken = False


class _WNConvFn(torch.autograd.Function):
    """y = out_act( (g/||v||)[co] * (v (*) in_act(up(x))) + bias + residual ) with hand-written backward.
    ``hand`` = (PARK_DX | PARK_RES | TAKE, GradHandover) or None: see the gradient hand-over above."""

    @staticmethod
    def forward(ctx, x, v, g, bias, residual, cfg: ConvCfg, prepared=None, sink=None, hand=None, fuse=None):
        """``fuse`` = (ActFuse of the layer that produced x or None, ActFuse of this layer's own output activation or None)"""
        ctx.sink = sink
        ctx.hand = hand
        ctx.fuse_in, ctx.fuse_out = fuse if fuse is not None else (None, None)
        w3 = v.reshape(v.shape[0], v.shape[1], -1)
        if prepared is None:
            scale, wf, wb = wn_scale(v, g), None, None
        else:
            scale, wf, wb = prepared
        y = conv1d_fwd(x, w3, scale, bias, residual, cfg, wf)
        ctx.cfg = cfg
        ctx.has_res = residual is not None
        ctx.has_bias = bias is not None
        ctx.wb = wb
        ctx.save_for_backward(x, v, g, scale, y if cfg.out_act != ACT_NONE else None)
        if not cfg.out_btc:
            _fm_announce(y, cfg.out_act, any(ctx.needs_input_grad))
        return y

    @staticmethod
    def backward(ctx, dy):
        cfg: ConvCfg = ctx.cfg
        x, v, g, scale, y = ctx.saved_tensors
        dy = _gpu_f32(dy, "grad_output")
        if cfg.out_btc:
            dy = dy.transpose(1, 2).contiguous()
            if y is not None:
                y = y.transpose(1, 2).contiguous()
        if cfg.out_act != ACT_NONE and not (ctx.fuse_out is not None and ctx.fuse_out.done):
            dy = act_bwd(dy, y, cfg.out_act, None if cfg.out_btc else _fm_take(y))
        # (fuse_out.done: the consumer of y applied this layer's activation derivative in its input-gradient epilogue)
        w3 = v.reshape(v.shape[0], v.shape[1], -1)
        dx = dv = dg = db = dres = None
        role, hand = ctx.hand if ctx.hand is not None else (0, None)
        if ctx.needs_input_grad[0]:
            add = None
            if role == TAKE:
                hand.taken = True
                add, hand.parked = hand.parked, None
                if add is not None and (add.shape[0] != x.shape[0] or add.shape[1] != cfg.cin
                                        or add.shape[2] != x.shape[2] * cfg.upsample or not add.is_contiguous()):
                    raise RuntimeError("ste_gan_amd: parked gradient does not match the taker's input gradient")
            if ctx.fuse_in is not None and ACT_FUSE and add is None and cfg.in_act == ACT_NONE and cfg.upsample == 1:
                dxe = conv1d_bwd_data(dy, w3, scale, None, None, cfg, x.shape[2], ctx.wb, act_src=x, act_add=_fm_take(x))
                ctx.fuse_in.done = True
            else:
                dxe = conv1d_bwd_data(dy, w3, scale, x if cfg.in_act == ACT_RELU else None, add, cfg, x.shape[2], ctx.wb)
            if role == PARK_DX and not hand.taken:
                hand.parked = dxe                    # at the upsampled resolution: the taker folds the sum once
            else:
                dx = fold_upsample2(dxe) if cfg.upsample == 2 else dxe
        if ctx.sink is not None and ctx.needs_input_grad[1]:
            # fused mode: accumulate the raw gradients into the network's arenas; (dv, dg) follow in ConvPack.finalize_grads.
            _sink_touch()
            _wgrad_sink(x, dy, cfg, ctx.sink[0], ctx.sink[1] if ctx.has_bias else None)
        elif ctx.needs_input_grad[1] or ctx.needs_input_grad[2] or (ctx.has_bias and ctx.needs_input_grad[3]):
            dw = torch.zeros_like(w3)
            db = torch.zeros(cfg.cout, device=dy.device, dtype=torch.float32) if ctx.has_bias else None
            conv1d_bwd_weight(x, dy, cfg, dw, db)
            dv = torch.empty_like(w3)
            dg = torch.empty(cfg.cout, device=dy.device, dtype=torch.float32)
            rows = cfg.cout
            check(_lib.lib().stegan_wn_backward1(_p(w3), _p(g), _p(dw), _p(dv), _p(dg), rows, w3.numel() // rows,
                                                 _stream()), "stegan_wn_backward1")
            dv = dv.view_as(v)
            dg = dg.view_as(g)
        if ctx.has_res and ctx.needs_input_grad[4]:
            if role == PARK_RES and not hand.taken and dy.is_contiguous():
                hand.parked = dy
            else:
                dres = dy
        return dx, dv, dg, db, dres, None, None, None, None, None


ACT_FUSE = os.environ.get("STEGAN_ACT_FUSE", "1") != "0"      # the consumer's input-gradient epilogue applies lrelu' (ActFuse)
_ACT_FUSE_SCOPE = [0]


class act_fuse_scope:
    """Inside this scope the discriminators link their layers with ActFuse objects.  ONLY valid when no gradient reaches a
    feature map through autograd except from the conv behind it -- the train step's discriminator passes: the D step uses
    the logits alone, the generator step's feature-matching gradients are parked for the backward kernels (fm_fold_begin),
    never returned to autograd.  A caller that differentiates a loss of the feature maps through autograd (the API allows
    it, tests/test_gpu_discriminator.py does) must NOT be in this scope: the producer would skip its activation derivative
    for that second gradient."""

    def __enter__(self):
        _ACT_FUSE_SCOPE[0] += 1
        return self

    def __exit__(self, *exc):
        _ACT_FUSE_SCOPE[0] -= 1
        return False


def new_act_fuse() -> Optional["ActFuse"]:
    """An ActFuse link inside an act_fuse_scope (and with gradients on), else None: the layers then apply their own
    activation derivatives, as autograd's general contract needs."""
    # (bf16 mode only: the fp32 kernels have no such epilogue -- they would run per problem and post-apply the derivative,
    # more launches than the layers' own act_bwd)
    return ActFuse() if (ACT_FUSE and _ACT_FUSE_SCOPE[0] > 0 and torch.is_grad_enabled() and HALF) else None


def _ptr_array(items):
    return (C.c_void_p * len(items))(*[C.c_void_p(t.data_ptr()) if t is not None else None for t in items])


def conv1d_fwd_multi(xs, cfgs, wfs, scales, biases, raws=None):
    """Up to five forward convs that share a kernel variant in ONE launch (stegan_conv1d_fwd_multi); None when the
    library says they cannot share one (the caller then launches them one by one).  ``raws``: the raw (Cout, Cin/g, K)
    weights, for the single-output-channel kernels."""
    n = len(xs)
    xs = [_gpu_f32(x, "x") for x in xs]
    ys, descs = [], []
    for x, cfg in zip(xs, cfgs):
        B, _, tin = x.shape
        tout = cfg.tout(tin)
        ys.append(torch.empty((B, cfg.cout, tout), device=x.device, dtype=torch.float32))
        descs.append(_desc(cfg, B, tin, tout))
    dptr = (C.c_void_p * n)(*[C.cast(C.pointer(d), C.c_void_p) for d in descs])
    if SKINNY and raws is not None and all(c.cout == 1 for c in cfgs):
        rc = _lib.lib().stegan_conv1d_fwd_skinny_multi(n, dptr, _ptr_array(xs), _ptr_array([_gpu_f32(w, "weight") for w in raws]),
                                                       _ptr_array(scales), _ptr_array(biases), _ptr_array(ys), _stream())
        if rc != 1:
            check(rc, "stegan_conv1d_fwd_skinny_multi")
            return ys
    whs = [_bf16_of(w) for w in wfs]
    if all(w is not None for w in whs):
        xhs = [_image_of(x, c.cin, x.shape[2]) for x, c in zip(xs, cfgs)]
        yhs = [_new_image(y, c.cout, c.groups, y.shape[2]) for y, c in zip(ys, cfgs)]
        rc = _lib.lib().stegan_conv1d_fwd_multi_hx(n, dptr, _ptr_array(xs), _ptr_array(xhs) if all(h is not None for h in xhs) else None,
                                                   _ptr_array(whs), _ptr_array(scales), _ptr_array(biases), None,
                                                   _ptr_array(ys), _ptr_array(yhs) if any(h is not None for h in yhs) else None,
                                                   _stream())
        if rc != 1:
            check(rc, "stegan_conv1d_fwd_multi_hx")
            for y, yh in zip(ys, yhs):
                _attach_image(y, yh)
            return ys
        return None                     # per-problem launches (each picks bf16 or fp32 on its own)
    if HALF:
        return None                     # (the fp32 copies are not maintained in bf16 mode)
    rc = _lib.lib().stegan_conv1d_fwd_multi(n, dptr, _ptr_array(xs), _ptr_array(wfs), _ptr_array(scales), _ptr_array(biases),
                                            None, _ptr_array(ys), _stream())
    if rc == 1:
        return None
    check(rc, "stegan_conv1d_fwd_multi")
    return ys


def conv1d_bwd_data_multi(dys, cfgs, wbs, tins, raws=None, scales=None, act_srcs=None, act_adds=None):
    """``act_srcs`` / ``act_adds``: see conv1d_bwd_data (every problem's input is the leaky-ReLU output of the layer before)."""
    n = len(dys)
    dys = [_gpu_f32(t, "dy") for t in dys]
    fuse = act_srcs is not None
    dxs, descs = [], []
    for dy, cfg, tin in zip(dys, cfgs, tins):
        B, _, tout = dy.shape
        dxs.append(torch.empty((B, cfg.cin, tin), device=dy.device, dtype=torch.float32))
        descs.append(_desc(replace(cfg, in_act=ACT_LRELU) if fuse else cfg, B, tin, tout))
    dptr = (C.c_void_p * n)(*[C.cast(C.pointer(d), C.c_void_p) for d in descs])
    masks = _ptr_array(act_srcs) if fuse else None
    adds = _ptr_array(act_adds) if (fuse and act_adds is not None and any(a is not None for a in act_adds)) else None
    if SKINNY and raws is not None and all(c.cout == 1 for c in cfgs):
        dxhs = [_new_image(t, c.cin, c.groups, t.shape[2]) for t, c in zip(dxs, cfgs)] if fuse else None
        rc = _lib.lib().stegan_conv1d_bwd_data_skinny_multi(n, dptr, _ptr_array(dys),
                                                            _ptr_array([_gpu_f32(w, "weight") for w in raws]),
                                                            _ptr_array(scales) if scales is not None else None, masks, adds,
                                                            _ptr_array(dxs),
                                                            _ptr_array(dxhs) if dxhs and any(h is not None for h in dxhs) else None,
                                                            _stream())
        if rc != 1:
            check(rc, "stegan_conv1d_bwd_data_skinny_multi")
            if dxhs:
                for t, th in zip(dxs, dxhs):
                    _attach_image(t, th)
            return dxs
    whs = [_bf16_of(w) for w in wbs]
    if all(w is not None for w in whs):
        dyhs = [_image_of(t, c.cout, t.shape[2]) for t, c in zip(dys, cfgs)]
        dxhs = [_new_image(t, c.cin, c.groups, t.shape[2]) for t, c in zip(dxs, cfgs)]
        rc = _lib.lib().stegan_conv1d_bwd_data_multi_hx(n, dptr, _ptr_array(dys),
                                                        _ptr_array(dyhs) if all(h is not None for h in dyhs) else None,
                                                        _ptr_array(whs), masks, adds, _ptr_array(dxs),
                                                        _ptr_array(dxhs) if any(h is not None for h in dxhs) else None, _stream())
        if rc != 1:
            check(rc, "stegan_conv1d_bwd_data_multi_hx")
            for t, th in zip(dxs, dxhs):
                _attach_image(t, th)
            return dxs
        return None
    if HALF or fuse:
        return None
    rc = _lib.lib().stegan_conv1d_bwd_data_multi(n, dptr, _ptr_array(dys), _ptr_array(wbs), None, None, _ptr_array(dxs),
                                                 _stream())
    if rc == 1:
        return None
    check(rc, "stegan_conv1d_bwd_data_multi")
    return dxs


class _WNConvMultiFn(torch.autograd.Function):
    """The same layer of several sub-discriminators (one channel structure; own weights, batch and length each) as ONE
    autograd node: forward and input gradient are single multi-problem launches when the library can share a kernel
    variant across them, per-problem launches otherwise.  ``layers`` = [(v, g, bias, cfg, (scale, wf, wb), sink)]."""

    @staticmethod
    def forward(ctx, layers, n, fuse, *ts):
        """``fuse`` = (ActFuse of the layer that produced the inputs | None, ActFuse of this layer's output activation | None):
        one object for all problems (they run as one node)."""
        ctx.fuse_in, ctx.fuse_out = fuse if fuse is not None else (None, None)
        xs = ts[:n]
        cfgs = [l[3] for l in layers]
        ys = None
        if PROFILER is None and all(c.upsample == 1 and not c.out_btc for c in cfgs):
            ys = conv1d_fwd_multi(xs, cfgs, [l[4][1] for l in layers], [l[4][0] for l in layers], [l[2] for l in layers],
                                  [l[0].reshape(l[0].shape[0], l[0].shape[1], -1) for l in layers])
        if ys is None:
            ys = [conv1d_fwd(x, l[0].reshape(l[0].shape[0], l[0].shape[1], -1), l[4][0], l[2], None, l[3], l[4][1])
                  for x, l in zip(xs, layers)]
        ctx.layers, ctx.n = layers, n
        ctx.wgrad = any(ctx.needs_input_grad[3 + n:])
        ctx.save_for_backward(*xs, *ys)
        for y, c in zip(ys, cfgs):
            if not c.out_btc:
                _fm_announce(y, c.out_act, any(ctx.needs_input_grad))
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        n, layers = ctx.n, ctx.layers
        saved = ctx.saved_tensors
        xs, ys = saved[:n], saved[n:]
        gs = [_gpu_f32(dy, "grad_output") for dy in dys]
        acts = {l[3].out_act for l in layers}
        if ctx.fuse_out is not None and ctx.fuse_out.done:
            pass         # the consumer layer applied this layer's activation derivative in its input-gradient epilogue
        elif ACT_MULTI and len(acts) == 1 and ACT_NONE not in acts and n <= 8 and not any(l[3].out_btc for l in layers) \
                and all(g.is_contiguous() for g in gs):
            # the activation backward of all problems (folded feature-matching gradients included) in one launch
            adds = [_fm_take(y) for y in ys]
            if any(a is not None and a.numel() != g.numel() for a, g in zip(adds, gs)):
                raise RuntimeError("ste_gan_amd.act_bwd: folded gradient does not match the activation gradient")
            outs = [torch.empty_like(g) for g in gs]
            numels = (C.c_int64 * n)(*[g.numel() for g in gs])
            ohs = [_new_image(o, o.shape[1], 1, o.shape[2]) if o.dim() == 3 else None for o in outs]
            check(_lib.lib().stegan_act_bwd_add_multi_img(
                n, _ptr_array(gs), _ptr_array(adds), _ptr_array(ys), _ptr_array(outs), _ptr_array(ohs), numels,
                (C.c_int32 * n)(*[o.shape[1] if o.dim() == 3 else 0 for o in outs]),
                (C.c_int32 * n)(*[o.shape[2] if o.dim() == 3 else 0 for o in outs]), acts.pop(), _stream()),
                "stegan_act_bwd_add_multi_img")
            for o, oh in zip(outs, ohs):
                _attach_image(o, oh)
            gs = outs
        else:
            gs = [act_bwd(g, y, l[3].out_act, None if l[3].out_btc else _fm_take(y)) if l[3].out_act != ACT_NONE else g
                  for g, y, l in zip(gs, ys, layers)]
        if ctx.wgrad:
            _sink_touch()
            done = False
            if SKINNY and PROFILER is None and n <= 5 and all(l[5] is not None and l[3].cout == 1 for l in layers):
                # the single-output-channel output convs: all problems' weight (and bias) gradients in one launch
                descs = [_desc(l[3], x.shape[0], x.shape[2], g.shape[2]) for x, g, l in zip(xs, gs, layers)]
                dptr = (C.c_void_p * n)(*[C.cast(C.pointer(d), C.c_void_p) for d in descs])
                ws = _workspace(xs[0].device, sum(int(_lib.lib().stegan_conv1d_bwd_weight_skinny_floats(C.byref(d_)))
                                                  for d_ in descs))
                rc = _lib.lib().stegan_conv1d_bwd_weight_skinny_multi(
                    n, dptr, _ptr_array(xs), _ptr_array(gs), _ptr_array([l[5][0] for l in layers]),
                    _ptr_array([l[5][1] if l[2] is not None else None for l in layers]), _p(ws), ws.numel(), _stream())
                if rc != 1:
                    check(rc, "stegan_conv1d_bwd_weight_skinny_multi")
                    done = True
            if not done:
                for x, g, l in zip(xs, gs, layers):
                    if l[5] is not None:
                        conv1d_bwd_weight(x, g, l[3], l[5][0], l[5][1] if l[2] is not None else None, defer=True)
        dxs = [None] * n
        need = [ctx.needs_input_grad[3 + i] for i in range(n)]
        if any(need):
            cfgs = [l[3] for l in layers]
            out = None
            fuse = ctx.fuse_in is not None and ACT_FUSE and all(need) and all(c.in_act == ACT_NONE and c.upsample == 1 for c in cfgs)
            srcs = list(xs) if fuse else None
            adds = [_fm_take(x) for x in xs] if fuse else None
            if PROFILER is None and all(need):
                out = conv1d_bwd_data_multi(gs, cfgs, [l[4][2] for l in layers], [x.shape[2] for x in xs],
                                            [l[0].reshape(l[0].shape[0], l[0].shape[1], -1) for l in layers],
                                            [l[4][0] for l in layers], srcs, adds)
            if out is None:
                out = [conv1d_bwd_data(g, l[0].reshape(l[0].shape[0], l[0].shape[1], -1), l[4][0], None, None, l[3],
                                       x.shape[2], l[4][2], act_src=(x if fuse else None), act_add=(adds[i] if fuse else None))
                       if nd else None for i, (g, x, l, nd) in enumerate(zip(gs, xs, layers, need))]
            if fuse:
                ctx.fuse_in.done = True
            dxs = out
        return (None, None, None, *dxs, *([None] * (len(ctx.needs_input_grad) - 3 - n)))


def wn_conv1d_multi(xs, convs, fuse=None):
    """convs: one WNConv1d per input (same layer of several sub-discriminators).  Falls back to per-module calls when a
    conv needs weight gradients but has no gradient sink.  ``fuse``: see _WNConvMultiFn.forward."""
    grad = torch.is_grad_enabled()
    needs_w = grad and any(c.weight_v.requires_grad for c in convs)
    # the multi node's backward passes no ReLU mask and no residual: only plain convs (in_act NONE) take this path
    if any(c.cfg.in_act != ACT_NONE for c in convs) or (needs_w and any(c._pack is None for c in convs)):
        return [c(x, fuse=fuse) for c, x in zip(convs, xs)]         # (one ActFuse object shared by the problems)
    layers, params = [], []
    for c in convs:
        sink = (c._pack[4], c.bias.grad) if (needs_w and c._pack is not None) else None
        layers.append((c.weight_v, c.weight_g, c.bias, c.cfg, c.prepared(), sink))
        params += [c.weight_v, c.weight_g, c.bias]
    return list(_WNConvMultiFn.apply(layers, len(xs), fuse, *xs, *params))


def wn_conv1d(x, v, g, bias, residual, cfg: ConvCfg, prepared=None, sink=None, hand=None, fuse=None) -> Tensor:
    """prepared = (scale, wf, wb) of the CURRENT weights (see layers.conv.WNConv1d.prepared) or None;
    sink = (raw dw arena slice, bias-gradient view) switches the backward to fused accumulation (optim.ConvPack);
    hand = (role, GradHandover): this conv's part in a gradient hand-over (GBlock);
    fuse = (ActFuse of x's producer | None, ActFuse of this conv's output activation | None)."""
    return _WNConvFn.apply(x, v, g, bias, residual, cfg, prepared, sink, hand, fuse)


def act_bwd(dy: Tensor, y: Tensor, act: int, add: Optional[Tensor] = None) -> Tensor:
    """(dy [+ add]) * act'(.) expressed through the stored OUTPUT y (lrelu: sign(y) = sign(pre); tanh: 1 - y^2)."""
    out = torch.empty_like(dy)
    if add is not None and (add.numel() != dy.numel() or not dy.is_contiguous()):
        raise RuntimeError("ste_gan_amd.act_bwd: folded gradient does not match the activation gradient")
    outh = None
    if dy.dim() == 3 and dy.is_contiguous():
        outh = _new_image(out, dy.shape[1], 1, dy.shape[2])
    Cc, T = (dy.shape[1], dy.shape[2]) if outh is not None else (0, 0)
    check(_lib.lib().stegan_act_bwd_add_img(_p(dy), _p(add), _p(y), _p(out), _p(outh), dy.numel(), Cc, T, act, _stream()),
          "stegan_act_bwd_add_img")
    _attach_image(out, outh)
    return out


# ---- folding the feature-matching gradient into the producer's activation backward -------------------------------
# A discriminator feature map y has two consumers in the generator step: the next conv and the feature-matching L1 term
# (train.py:236-239).  Its second gradient is known since the FORWARD pass (the loss kernel writes it next to the loss),
# so instead of returning it to autograd -- which sums the two with one elementwise kernel per feature map, 48 a step --
# multi_pair_loss(fold=True) parks it here and the conv that produced y adds it inside its activation backward.  Keys are
# (data_ptr, numel) of y's storage; a conv announces its activated output while a fold window is open
# (fm_fold_begin .. fm_fold_end, the train step's generator phase); fm_fold_end() raises when a parked gradient was never
# collected, so a graph shape this shortcut does not cover fails loudly instead of dropping a term.
_FM_WINDOW = False
_FM_FOLDABLE: set = set()
_FM_PENDING: dict = {}
FM_FOLD_STATS = {"parked": 0, "collected": 0}      # of the last closed window (tests)
_FM_COUNT = [0, 0]


def fm_fold_begin():
    global _FM_WINDOW
    _FM_FOLDABLE.clear(); _FM_PENDING.clear()
    _FM_COUNT[0] = _FM_COUNT[1] = 0
    _FM_WINDOW = True


def fm_fold_end():
    global _FM_WINDOW
    _FM_WINDOW = False
    left = len(_FM_PENDING)
    _FM_FOLDABLE.clear(); _FM_PENDING.clear()
    cur = torch.cuda.current_stream() if _FM_STREAMS else None
    for key, s_ in _FM_STREAMS.items():         # the consumers' kernels first, then the references go
        if key != cur.cuda_stream:
            cur.wait_stream(s_)
    _FM_STREAMS.clear(); _FM_TAKEN.clear()
    FM_FOLD_STATS["parked"], FM_FOLD_STATS["collected"] = _FM_COUNT
    if left:
        raise RuntimeError(f"ste_gan_amd: {left} folded feature-matching gradient(s) were never collected by their producers")


def fm_fold_abort():
    """Close the window without the check (exception paths)."""
    global _FM_WINDOW
    _FM_WINDOW = False
    _FM_FOLDABLE.clear(); _FM_PENDING.clear()
    _FM_STREAMS.clear(); _FM_TAKEN.clear()


def _fm_announce(y: Tensor, act: int, needs_grad: bool):
    if _FM_WINDOW and needs_grad and act != ACT_NONE and y.is_contiguous():
        _FM_FOLDABLE.add((y.data_ptr(), y.numel()))


_FM_TAKEN: list = []           # collected gradients, kept alive until fm_fold_end() has joined their consumers' streams
_FM_STREAMS: dict = {}


def _fm_take(y: Tensor):
    g = _FM_PENDING.pop((y.data_ptr(), y.numel()), None) if _FM_PENDING else None
    if g is not None:
        _FM_COUNT[1] += 1
        # the gradient was allocated on the stream of the loss kernel and is read here, possibly on another stream (the
        # period branch's backward): dropping the last reference now would hand its memory back to the allocating stream's
        # pool while this stream's kernel is still queued -- a reuse race seen as sporadic NaN losses
        _FM_TAKEN.append(g)
        s_ = torch.cuda.current_stream()
        _FM_STREAMS[s_.cuda_stream] = s_
    return g


# ---------------------------------------------------------------------------------------------
# autograd: spectral-normalised fused conv (MSD scale 0)
# ---------------------------------------------------------------------------------------------
class _SNConvFn(torch.autograd.Function):
    """y = out_act( (w_orig (*) x) / sigma + bias ), sigma = u^T W v after one in-place power iteration."""

    @staticmethod
    def forward(ctx, x, w_orig, u, v, bias, cfg: ConvCfg, training: bool, prepared=None, pre=None, sink=None, fuse=None):
        """``sink`` = (weight_orig.grad, bias.grad) arena views: the backward pass accumulates into them directly and
        returns no parameter gradients (as the ConvPack sink of the weight-norm convs does); ``fuse``: see _WNConvFn."""
        ctx.sink = sink
        ctx.fuse_in, ctx.fuse_out = fuse if fuse is not None else (None, None)
        wf, wb = prepared if prepared is not None else (None, None)
        need_w = ctx.needs_input_grad[1]
        if pre is not None:
            # (scale, sigma, u copy, v copy) of THIS forward, computed for all layers of the discriminator at once
            # (spectral_norm_multi): the power iteration has already advanced u / v in place
            scale, sigma, u_keep, v_keep = pre
            if need_w and (u_keep is None or v_keep is None):
                raise RuntimeError("ste_gan_amd.sn_conv1d: precomputed spectral-norm state lacks the u / v copies")
        else:
            rows = w_orig.shape[0]
            cols = w_orig.numel() // rows
            L = _lib.lib()
            ws = torch.empty(int(L.stegan_spectral_norm_workspace_floats(rows, cols)), device=x.device, dtype=torch.float32)
            scale = torch.empty(rows, device=x.device, dtype=torch.float32)
            sigma = torch.empty(1, device=x.device, dtype=torch.float32)
            check(L.stegan_spectral_norm_fwd(_p(w_orig), _p(u), _p(v), _p(scale), _p(sigma), _p(ws), rows, cols,
                                             int(training), _stream()), "stegan_spectral_norm_fwd")
            # u / v are cloned because the next forward power-iterates them in place (torch clones them too) -- but only
            # the weight gradient reads them, so frozen-weight passes (generator step, no_grad) skip the two copies
            u_keep = u.clone() if need_w else u
            v_keep = v.clone() if need_w else v
        y = conv1d_fwd(x, w_orig, scale, bias, None, cfg, wf)
        ctx.cfg = cfg
        ctx.wb = wb
        ctx.save_for_backward(x, w_orig, u_keep if u_keep is not None else u, v_keep if v_keep is not None else v, scale,
                              sigma, y if cfg.out_act != ACT_NONE else None)
        _fm_announce(y, cfg.out_act, any(ctx.needs_input_grad))
        return y

    @staticmethod
    def backward(ctx, dy):
        cfg: ConvCfg = ctx.cfg
        x, w_orig, u, v, scale, sigma, y = ctx.saved_tensors
        dy = _gpu_f32(dy, "grad_output")
        if cfg.out_act != ACT_NONE and not (ctx.fuse_out is not None and ctx.fuse_out.done):
            dy = act_bwd(dy, y, cfg.out_act, _fm_take(y))
        dx = dw_orig = db = None
        if ctx.needs_input_grad[0]:
            if ctx.fuse_in is not None and ACT_FUSE and cfg.upsample == 1:
                dx = conv1d_bwd_data(dy, w_orig, scale, None, None, cfg, x.shape[2], ctx.wb, uniform_scale=True,
                                     act_src=x, act_add=_fm_take(x))
                ctx.fuse_in.done = True
            else:
                dx = conv1d_bwd_data(dy, w_orig, scale, None, None, cfg, x.shape[2], ctx.wb, uniform_scale=True)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[4]:
            sink = ctx.sink
            if sink is not None:
                _sink_touch()
            dw = pooled_zeros(w_orig.shape, dy.device)
            db = torch.zeros(cfg.cout, device=dy.device, dtype=torch.float32) if sink is None else sink[1]
            conv1d_bwd_weight(x, dy, cfg, dw, db)
            rows = w_orig.shape[0]
            ws = torch.empty(rows, device=dy.device, dtype=torch.float32)
            dw_orig = torch.empty_like(w_orig) if sink is None else sink[0]
            if sink is not None and SN_BWD_DEFER:
                # sink mode: only the optimiser reads this gradient -- the transform waits for wgrad_join (finalize_grads),
                # where all of a step's run as one launch per stage (sn_bwd_flush)
                _SN_PENDING.append((w_orig, dw, u, v, sigma, ws, dw_orig, rows, w_orig.numel() // rows))
            else:
                check(_lib.lib().stegan_spectral_norm_bwd_acc(_p(w_orig), _p(dw), _p(u), _p(v), _p(sigma), _p(ws), _p(dw_orig),
                                                              rows, w_orig.numel() // rows, 0 if sink is None else 1, _stream()),
                      "stegan_spectral_norm_bwd_acc")
            if sink is not None:
                dw_orig = db = None              # already in the gradient arena (or on its way: sn_bwd_flush)
        return dx, dw_orig, None, None, db, None, None, None, None, None, None


SN_BWD_DEFER = os.environ.get("STEGAN_SN_BWD_DEFER", "1") == "1"
_SN_PENDING: list = []          # (w_orig, raw dw, u, v, sigma, workspace, dW arena view, rows, cols) in backward order


def sn_bwd_flush() -> None:
    """The spectral-norm backward transforms a backward pass left pending (sink mode), on the current stream, which the caller
    has ordered behind every weight-gradient launch (wgrad_join).  Two passes through one layer (fake, real) add into the SAME
    arena view with different u / v / sigma: they go into different launches, in the order the backward pass produced them, so
    the sums are formed exactly as by the per-layer launches (bit-identical)."""
    if not _SN_PENDING:
        return
    rounds, seen = [], {}
    for e in _SN_PENDING:
        r = seen.get(e[6].data_ptr(), 0)
        seen[e[6].data_ptr()] = r + 1
        while len(rounds) <= r:
            rounds.append([])
        rounds[r].append(e)
    i32 = lambda xs: (C.c_int32 * len(xs))(*xs)
    for grp in rounds:
        n = len(grp)
        col = lambda k: _ptr_array([e[k] for e in grp])
        check(_lib.lib().stegan_spectral_norm_bwd_acc_multi(n, col(0), col(1), col(2), col(3), col(4), col(5), col(6),
                                                            i32([e[7] for e in grp]), i32([e[8] for e in grp]), 1, _stream()),
              "stegan_spectral_norm_bwd_acc_multi")
    _SN_PENDING.clear()


def sn_conv1d(x, w_orig, u, v, bias, cfg: ConvCfg, training: bool, prepared=None, pre=None, sink=None, fuse=None) -> Tensor:
    return _SNConvFn.apply(x, w_orig, u, v, bias, cfg, training, prepared, pre, sink, fuse)


SKINNY = os.environ.get("STEGAN_SKINNY", "1") != "0"   # Cout == 1 convs on the streaming kernels of conv_skinny.hip
SN_BATCHED = os.environ.get("STEGAN_SN_BATCHED", "1") != "0"      # scale discriminator: all spectral-norm layers' power iterations in one call (0: per layer)


def spectral_norm_multi(layers, training: bool, keep_uv: bool):
    """One forward's spectral-norm state of up to 8 layers -- ``layers`` = [(weight_orig, u, v), ...] -- in four launches
    (stegan_spectral_norm_fwd_multi) instead of four per layer: the layers' power iterations are independent of each other
    and of the activations, so running them ahead of the first conv gives the reference's values (conv.py:92-94,
    discriminator.py:100-113).  In training mode u / v advance in place, exactly once.  Returns, per layer, the ``pre``
    argument of :func:`sn_conv1d`: (scale (rows,), sigma (1,), u copy, v copy); the copies are None unless ``keep_uv``
    (they are what the weight gradient reads after later forwards have moved u / v on)."""
    n = len(layers)
    L = _lib.lib()
    dev = layers[0][0].device
    rows = [int(w.shape[0]) for w, _, _ in layers]
    cols = [int(w.numel() // w.shape[0]) for w, _, _ in layers]
    nws = [int(L.stegan_spectral_norm_workspace_floats(r, c)) for r, c in zip(rows, cols)]
    # one allocation: [workspace | scale | sigma (padded to 4) | u copy | v copy] per layer, 16-byte aligned pieces
    sizes = []
    for r, c, w_ in zip(rows, cols, nws):
        sizes += [w_, r, 4, r if keep_uv else 0, c if keep_uv else 0]
    offs, tot = [], 0
    for z in sizes:
        offs.append(tot)
        tot += (z + 3) & ~3
    flat = torch.empty(tot, device=dev, dtype=torch.float32)
    views = [flat[o:o + z] for o, z in zip(offs, sizes)]
    ws, scale, sigma, uk, vk = (views[i::5] for i in range(5))
    sigma = [t[:1] for t in sigma]
    i32 = lambda xs: (C.c_int32 * n)(*xs)
    check(L.stegan_spectral_norm_fwd_multi(
        n, _ptr_array([w for w, _, _ in layers]), _ptr_array([u for _, u, _ in layers]), _ptr_array([v for _, _, v in layers]),
        _ptr_array(scale), _ptr_array(sigma), _ptr_array(uk) if keep_uv else None, _ptr_array(vk) if keep_uv else None,
        _ptr_array(ws), i32(rows), i32(cols), int(training), _stream()), "stegan_spectral_norm_fwd_multi")
    return [(scale[i], sigma[i], uk[i] if keep_uv else None, vk[i] if keep_uv else None) for i in range(n)]


# ---------------------------------------------------------------------------------------------
# autograd: generator input (embedding gather + concat + transpose)
# ---------------------------------------------------------------------------------------------
class _EmbedConcatFn(torch.autograd.Function):
    """(B, T, F) speech features + embedding rows of the batch's session (and speaking-mode) ids -> (B, F + E (+ E), T), the
    generator's first conv input (generator.py:143-154), in one launch; backward: the tables' gradients, one small launch
    each (the speech features need none: a leaf of the data path)."""

    @staticmethod
    def forward(ctx, x, id1, w1, id2, w2, sink1=None, sink2=None):
        """``sink1`` / ``sink2``: the tables' ``.grad`` views in a FlatParams arena -- the backward pass then accumulates into
        them directly and returns no gradient (as the convs' gradient sinks do)."""
        ctx.sinks = (sink1, sink2)
        x = _gpu_f32(x, "speech_unit_sequence")
        w1 = _gpu_f32(w1, "session_embeddings.weight")
        if not id1.is_cuda or id1.dtype != torch.int64:
            raise RuntimeError("ste_gan_amd: session ids must be a cuda int64 tensor")
        id1 = id1.contiguous()
        B, T, F = x.shape
        E = w1.shape[1]
        two = w2 is not None
        if two:
            w2 = _gpu_f32(w2, "speaking_mode_embeddings.weight")
            id2 = id2.contiguous()
            if w2.shape[1] != E or not id2.is_cuda or id2.dtype != torch.int64:
                raise RuntimeError("ste_gan_amd: the two embeddings must share their width; ids cuda int64")
        Ct = F + E * (2 if two else 1)
        out = torch.empty((B, Ct, T), device=x.device, dtype=torch.float32)
        outh = _new_image(out, Ct, 1, T)
        check(_lib.lib().stegan_embed_concat_fwd(_p(x), _p(id1), _p(w1), _p(id2) if two else None, _p(w2) if two else None,
                                                 _p(out), _p(outh), B, T, F, E, w1.shape[0], w2.shape[0] if two else 0,
                                                 _stream()), "stegan_embed_concat_fwd")
        _attach_image(out, outh)
        ctx.dims = (B, T, F, E, Ct, two)
        ctx.save_for_backward(id1, id2 if two else None)
        ctx.shapes = (w1.shape, w2.shape if two else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, T, F, E, Ct, two = ctx.dims
        id1, id2 = ctx.saved_tensors
        dout = _gpu_f32(dout, "grad_output")
        dw1 = dw2 = None
        s1, s2 = ctx.sinks
        if ctx.needs_input_grad[2]:
            if s1 is not None:
                _sink_touch()
            dw1 = s1 if s1 is not None else torch.zeros(ctx.shapes[0], device=dout.device, dtype=torch.float32)
            check(_lib.lib().stegan_embed_concat_bwd(_p(dout), _p(id1), _p(dw1), B, T, Ct, F, E, ctx.shapes[0][0], _stream()),
                  "stegan_embed_concat_bwd")
            if s1 is not None:
                dw1 = None                       # already in the gradient arena
        if two and ctx.needs_input_grad[4]:
            if s2 is not None:
                _sink_touch()
            dw2 = s2 if s2 is not None else torch.zeros(ctx.shapes[1], device=dout.device, dtype=torch.float32)
            check(_lib.lib().stegan_embed_concat_bwd(_p(dout), _p(id2), _p(dw2), B, T, Ct, F + E, E, ctx.shapes[1][0], _stream()),
                  "stegan_embed_concat_bwd")
            if s2 is not None:
                dw2 = None
        return None, None, dw1, None, dw2, None, None


def embed_concat(x, id1, w1, id2=None, w2=None, sink1=None, sink2=None) -> Tensor:
    return _EmbedConcatFn.apply(x, id1, w1, id2, w2, sink1, sink2)


# ---------------------------------------------------------------------------------------------
# autograd: discriminator input plumbing
# ---------------------------------------------------------------------------------------------
class _MPDFoldFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_btc, period: int):
        x_btc = _gpu_f32(x_btc, "x")
        B, T, Cc = x_btc.shape
        n_pad = period - T % period
        if n_pad >= T:
            raise RuntimeError(f"Argument #4: Padding size should be less than the corresponding input dimension, "
                               f"but got: padding (0, {n_pad}) at dimension 2 of input {[B, Cc, T]}")
        H = (T + n_pad) // period
        out = torch.empty((B * period, Cc, H), device=x_btc.device, dtype=torch.float32)
        check(_lib.lib().stegan_mpd_fold_fwd(_p(x_btc), _p(out), B, T, Cc, period, _stream()), "stegan_mpd_fold_fwd")
        ctx.dims = (B, T, Cc, period)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, T, Cc, period = ctx.dims
        dout = _gpu_f32(dout, "grad")
        dx = torch.empty((B, T, Cc), device=dout.device, dtype=torch.float32)
        check(_lib.lib().stegan_mpd_fold_bwd(_p(dout), _p(dx), B, T, Cc, period, 0, _stream()), "stegan_mpd_fold_bwd")
        return dx, None


class _BtcToBctFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _gpu_f32(x, "x")
        B, T, Cc = x.shape
        y = torch.empty((B, Cc, T), device=x.device, dtype=torch.float32)
        check(_lib.lib().stegan_btc_to_bct(_p(x), _p(y), B, T, Cc, _stream()), "stegan_btc_to_bct")
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _gpu_f32(dy, "grad")
        B, Cc, T = dy.shape
        dx = torch.empty((B, T, Cc), device=dy.device, dtype=torch.float32)
        check(_lib.lib().stegan_bct_to_btc(_p(dy), _p(dx), B, T, Cc, 0, _stream()), "stegan_bct_to_btc")
        return dx


class _AvgPool421Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _gpu_f32(x, "x")
        B, Cc, T = x.shape
        To = (T + 2 - 4) // 2 + 1
        y = torch.empty((B, Cc, To), device=x.device, dtype=torch.float32)
        check(_lib.lib().stegan_avgpool421_fwd(_p(x), _p(y), B * Cc, T, _stream()), "stegan_avgpool421_fwd")
        ctx.T = T
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _gpu_f32(dy, "grad")
        B, Cc, To = dy.shape
        dx = torch.empty((B, Cc, ctx.T), device=dy.device, dtype=torch.float32)
        check(_lib.lib().stegan_avgpool421_bwd(_p(dy), None, _p(dx), B * Cc, ctx.T, _stream()), "stegan_avgpool421_bwd")
        return dx


def mpd_fold(x_btc, period):
    return _MPDFoldFn.apply(x_btc, period)


class _MPDFoldMultiFn(torch.autograd.Function):
    """The folds of all period discriminators as ONE node: one launch forward, and one launch backward that also forms
    the sum of the sub-discriminators' gradients w.r.t. the signal (stegan_mpd_fold_fwd_multi / _bwd_multi)."""

    @staticmethod
    def forward(ctx, x_btc, periods):
        x_btc = _gpu_f32(x_btc, "x")
        B, T, Cc = x_btc.shape
        outs = []
        for period in periods:
            n_pad = period - T % period
            if n_pad >= T:
                raise RuntimeError(f"Argument #4: Padding size should be less than the corresponding input dimension, "
                                   f"but got: padding (0, {n_pad}) at dimension 2 of input {[B, Cc, T]}")
            outs.append(torch.empty((B * period, Cc, (T + n_pad) // period), device=x_btc.device, dtype=torch.float32))
        pa = (C.c_int32 * len(periods))(*periods)
        check(_lib.lib().stegan_mpd_fold_fwd_multi(_p(x_btc), _ptr_array(outs), B, T, Cc, pa, len(periods), _stream()),
              "stegan_mpd_fold_fwd_multi")
        ctx.dims = (B, T, Cc, tuple(periods), [o.shape for o in outs])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        B, T, Cc, periods, shapes = ctx.dims
        dev = next(d for d in douts if d is not None).device
        ds = [_gpu_f32(d, "grad") if d is not None else torch.zeros(sh, device=dev, dtype=torch.float32)
              for d, sh in zip(douts, shapes)]
        dx = torch.empty((B, T, Cc), device=dev, dtype=torch.float32)
        pa = (C.c_int32 * len(periods))(*periods)
        check(_lib.lib().stegan_mpd_fold_bwd_multi(_ptr_array(ds), _p(dx), B, T, Cc, pa, len(periods), 0, _stream()),
              "stegan_mpd_fold_bwd_multi")
        return dx, None


MPD_FOLD_MULTI = os.environ.get("STEGAN_MPD_FOLD_MULTI", "1") != "0"
ACT_MULTI = os.environ.get("STEGAN_ACT_MULTI", "1") != "0"          # one activation-backward launch per period-discriminator layer


def mpd_fold_multi(x_btc, periods):
    """[mpd_fold(x, p) for p in periods] in one launch (and one backward launch); at most 8 periods per launch."""
    periods = [int(p) for p in periods]
    if not MPD_FOLD_MULTI or len(periods) < 2 or len(periods) > 8:
        return [mpd_fold(x_btc, p) for p in periods]
    return list(_MPDFoldMultiFn.apply(x_btc, periods))


def btc_to_bct(x):
    return _BtcToBctFn.apply(x)


def avgpool421(x):
    return _AvgPool421Fn.apply(x)


# ---------------------------------------------------------------------------------------------
# autograd: losses.  Each kernel emits the scalar and the unit gradient in one pass; backward only
# rescales the stored gradient by the upstream 0-dim gradient (read on the device, no host sync).
# ---------------------------------------------------------------------------------------------
def _chain(unit_grad: Tensor, grad_out: Tensor) -> Tensor:
    out = torch.empty_like(unit_grad)
    g = _gpu_f32(grad_out.reshape(1), "grad_output")
    check(_lib.lib().stegan_scale_by_scalar(_p(unit_grad), _p(g), _p(out), unit_grad.numel(), _stream()),
          "stegan_scale_by_scalar")
    return out


def _memory_order_pair(a: Tensor, b: Optional[Tensor]):
    """Flat views of a (and b) in a's MEMORY order -- permuted dense views (the MPD feature maps) need no copy
    because the means below are permutation invariant.  Returns (a_flat, b_flat, restore(da_flat) -> like a)."""
    if a.is_contiguous() and (b is None or b.is_contiguous()):
        return a.reshape(-1), (b.reshape(-1) if b is not None else None), (lambda g: g.view(a.shape))
    order = sorted(range(a.dim()), key=lambda i: -a.stride(i))
    ap = a.permute(order)
    if ap.is_contiguous() and (b is None or (b.shape == a.shape and b.stride() == a.stride())):
        inv = [order.index(i) for i in range(a.dim())]
        bp = b.permute(order) if b is not None else None
        return ap.reshape(-1), (bp.reshape(-1) if bp is not None else None), (lambda g: g.view(ap.shape).permute(inv))
    a_c = a.contiguous()
    return a_c.reshape(-1), (b.contiguous().reshape(-1) if b is not None else None), (lambda g: g.view(a.shape))


class _PairLossFn(torch.autograd.Function):
    """mean |a-b| (mode 0) or mean (a-b)^2 (mode 1); b may be None (constant target)."""

    @staticmethod
    def forward(ctx, a, b, target: float, mode: int):
        a = a if a.is_cuda else _gpu_f32(a, "input")
        af, bf, restore = _memory_order_pair(a, b)
        af = _gpu_f32(af, "input")
        if bf is not None:
            bf = _gpu_f32(bf, "target")
            if bf.numel() != af.numel():
                raise RuntimeError(f"The size of tensor a ({list(a.shape)}) must match the size of tensor b ({list(b.shape)})")
        loss = pooled_zeros((), af.device)
        need = ctx.needs_input_grad[0]
        da = torch.empty_like(af) if need else None
        part = _workspace(af.device, 128)
        check(_lib.lib().stegan_pair_loss_multi_det(1, _ptr_array([af]), _ptr_array([bf]) if bf is not None else None,
                                                    (C.c_int64 * 1)(af.numel()), (C.c_float * 1)(1.0),
                                                    (C.c_float * 1)(float(target)), None, None, mode, _p(loss), _ptr_array([da]),
                                                    _p(part), part.numel(), _stream()), "stegan_pair_loss_multi_det")
        ctx.restore = restore
        ctx.save_for_backward(da)
        return loss

    @staticmethod
    def backward(ctx, g):
        (da,) = ctx.saved_tensors
        if da is None:
            return None, None, None, None
        return ctx.restore(_chain(da, g)), None, None, None


class _MultiPairLossFn(torch.autograd.Function):
    """weight * sum_i mean(|a_i - b_i|)  (mode 0)  or  weight * sum_i mean((a_i - b_i)^2)  (mode 1) in one autograd node:
    every pair's kernel adds into ONE loss word and writes its own gradient, so a 40-term feature-matching sum costs 40
    launches instead of 40 x (zero-fill + loss kernel + scalar add + gradient scale).

    ``unit_grad``: the caller promises that the returned (already weighted) value enters the scalar handed to
    ``backward()`` as a plain summand, i.e. its upstream gradient is exactly 1 -- the stored gradients are then returned
    as they are instead of being multiplied by a device scalar that is known to be one."""

    @staticmethod
    def forward(ctx, mode: int, weight, target, unit_grad: int, n: int, *ts):
        fold = unit_grad == 2          # unit upstream gradient AND gradients parked for the producers (see _fm_announce)
        unit_grad = bool(unit_grad)
        spec = None
        if isinstance(weight, tuple):  # per-pair (weights, targets, targets2, splits): multi_pair_loss_const
            spec, weight, target = weight, 0.0, 0.0
        a_list, b_list = ts[:n], (ts[n:] if target is None else (None,) * n)
        loss = pooled_zeros((), a_list[0].device)
        grads, restores = [], []
        L = _lib.lib()
        afs, bfs, das = [], [], []
        for i, (a, b) in enumerate(zip(a_list, b_list)):
            af, bf, restore = _memory_order_pair(a, b)
            af = _gpu_f32(af, "input")
            if bf is not None:
                bf = _gpu_f32(bf, "target")
                if bf.numel() != af.numel():
                    raise RuntimeError(f"The size of tensor a ({list(a.shape)}) must match the size of tensor b ({list(b.shape)})")
            da = torch.empty_like(af) if ctx.needs_input_grad[5 + i] else None
            afs.append(af); bfs.append(bf); das.append(da); restores.append(restore)
        # every pair in ONE launch (stegan_pair_loss_multi), all adding into the one loss word
        m = len(afs)
        numel = (C.c_int64 * m)(*[t.numel() for t in afs])
        # (stegan_pair_loss_multi_det: the blocks' shares of the value go through a scratch and a fixed-order sum -- no float
        # atomics, the reported loss words of two launches agree to the bit)
        fa = lambda xs: (C.c_float * m)(*[float(v) for v in xs])
        part = _workspace(afs[0].device, 128 * m)
        if spec is None:
            check(L.stegan_pair_loss_multi_det(m, _ptr_array(afs), _ptr_array(bfs), numel, fa([weight] * m),
                                               fa([float(target or 0.0)] * m), None, None, mode, _p(loss), _ptr_array(das),
                                               _p(part), part.numel(), _stream()), "stegan_pair_loss_multi_det")
        else:
            ws_, t1, t2, sp = spec
            check(L.stegan_pair_loss_multi_det(m, _ptr_array(afs), None, numel, fa(ws_), fa(t1), fa(t2),
                                               (C.c_int64 * m)(*[int(v) for v in sp]), mode, _p(loss), _ptr_array(das),
                                               _p(part), part.numel(), _stream()), "stegan_pair_loss_multi_det")
        for af, da in zip(afs, das):
            key = (af.data_ptr(), af.numel())
            if fold and da is not None and key in _FM_FOLDABLE and key not in _FM_PENDING:
                _FM_PENDING[key] = da          # collected by the conv that produced a_i, inside its activation backward
                _FM_COUNT[0] += 1
                da = None
            grads.append(da)
        ctx.restores, ctx.unit_grad, ctx.n_extra = restores, unit_grad, len(ts) - n
        ctx.save_for_backward(*[g for g in grads if g is not None])
        ctx.has = [g is not None for g in grads]
        return loss

    @staticmethod
    def backward(ctx, g):
        saved = list(ctx.saved_tensors)
        out = []
        for has, restore in zip(ctx.has, ctx.restores):
            if not has:
                out.append(None)
                continue
            da = saved.pop(0)
            out.append(restore(da if ctx.unit_grad else _chain(da, g)))
        return (None, None, None, None, None, *out, *([None] * ctx.n_extra))


def multi_pair_loss(a_list, b_list, mode: int, weight: float = 1.0, target=None, unit_grad: bool = False,
                    fold: bool = False):
    """``fold`` (needs ``unit_grad``, and an open fm_fold_begin() window): gradients of inputs that are the activated
    outputs of this package's convs are not returned to autograd but added by those convs' backward passes."""
    a_list = list(a_list)
    extra = [] if target is not None else [b.detach() for b in b_list]
    if fold and not unit_grad:
        raise ValueError("multi_pair_loss: fold requires unit_grad")
    return _MultiPairLossFn.apply(mode, float(weight), target, 2 if fold else int(bool(unit_grad)), len(a_list), *a_list,
                                  *extra)


def multi_pair_loss_const(a_list, mode: int, weights, targets, targets2=None, splits=None, unit_grad: bool = False):
    """sum_i weights[i] * mean over a_i of |a - t| (mode 0) or (a - t)^2 (mode 1) against CONSTANT targets in one launch:
    t = targets[i] for the first splits[i] elements of a_i in memory order and targets2[i] for the rest (splits[i] = 0:
    one target).  The discriminator step's batched fake | real logits meet zeros | ones this way without being sliced
    (slicing costs autograd two zero-fills, two copies and an add per tensor on the way back)."""
    a_list = list(a_list)
    n = len(a_list)
    spec = (tuple(weights), tuple(targets), tuple(targets2 if targets2 is not None else targets),
            tuple(splits if splits is not None else [0] * n))
    return _MultiPairLossFn.apply(mode, spec, 0.0, int(bool(unit_grad)), n, *a_list)


def l1_loss(a, b):
    return _PairLossFn.apply(a, b.detach(), 0.0, 0)


def mse_loss(a, b):
    return _PairLossFn.apply(a, b.detach(), 0.0, 1)


def mse_to_const(a, target: float):
    return _PairLossFn.apply(a, None, float(target), 1)


class _MultiTDFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_real, x_gen, wins, shifts, avg_window: int = 9, pad_windowing: bool = True, weight: float = 1.0,
                unit_grad: bool = False):
        """``weight`` scales the loss value AND the stored gradient; ``unit_grad``: the caller promises an upstream gradient
        of exactly 1 (the weighted value is a plain summand of the scalar it differentiates): backward then returns the
        stored gradient as it is, no scaling launch."""
        ctx.unit_grad = bool(unit_grad)
        x_real = _gpu_f32(x_real, "x_real")
        x_gen = _gpu_f32(x_gen, "x_generated")
        if x_real.shape != x_gen.shape:
            raise RuntimeError(f"The size of tensor a ({list(x_gen.shape)}) must match the size of tensor b "
                               f"({list(x_real.shape)})")
        B, T, Cc = x_gen.shape
        L = _lib.lib()
        n = len(wins)
        loss = pooled_zeros((), x_gen.device)
        per = pooled_zeros(3, x_gen.device)
        ws = torch.empty(int(L.stegan_multi_td_workspace_floats(B, Cc)), device=x_gen.device, dtype=torch.float32)
        dx = torch.empty_like(x_gen) if ctx.needs_input_grad[1] else None
        wa = (C.c_int32 * n)(*wins)
        sa = (C.c_int32 * n)(*shifts)
        # rows too long for LDS (whole utterances in validation) keep their row buffers in a global scratch
        ns = int(L.stegan_multi_td_scratch_floats(B, T, Cc, n, wa, sa, int(pad_windowing)))
        if ns < 0:
            raise RuntimeError("stegan_multi_td_scratch_floats: invalid argument or unsupported shape")
        scratch = torch.empty(ns, device=x_gen.device, dtype=torch.float32) if ns > 0 else None
        check(L.stegan_multi_td_fwd_bwd_ex(_p(x_real), _p(x_gen), B, T, Cc, n, wa, sa, int(avg_window), int(pad_windowing),
                                           float(weight), _p(loss), _p(per), _p(dx), _p(ws), _p(scratch), ns, _stream()),
              "stegan_multi_td_fwd_bwd_ex")
        ctx.save_for_backward(dx)
        ctx.mark_non_differentiable(per)
        return loss, per

    @staticmethod
    def backward(ctx, g, _gper):
        (dx,) = ctx.saved_tensors
        if dx is None:
            return None, None, None, None, None, None, None, None
        return None, (dx if ctx.unit_grad else _chain(dx, g)), None, None, None, None, None, None


def multi_td_loss(x_real, x_gen, wins, shifts, avg_window: int = 9, pad_windowing: bool = True, weight: float = 1.0,
                  unit_grad: bool = False):
    return _MultiTDFn.apply(x_real.detach(), x_gen, tuple(wins), tuple(shifts), int(avg_window), bool(pad_windowing),
                            float(weight), bool(unit_grad))


def td_features(x: Tensor, win: int, shift: int, avg_window: int = 9, pad_windowing: bool = True) -> Tensor:
    """TimeDomainFeatureLoss.calculate_time_domain_features (time_domain_loss.py:58-69): (B, T, C) -> (B, F, C, 4), no
    gradient (the loss kernel carries its own backward)."""
    x = _gpu_f32(x.detach(), "raw_x")
    B, T, Cc = x.shape
    P = win // 2 if pad_windowing else 0
    if win > T + 2 * P or P >= T:
        raise RuntimeError(f"maximum size for tensor at dimension 1 is {T + 2 * P} but size is {win}")
    F = (T + 2 * P - win) // shift + 1
    feats = torch.empty((B, F, Cc, 4), device=x.device, dtype=torch.float32)
    scratch = torch.empty(3 * B * Cc * T, device=x.device, dtype=torch.float32)
    check(_lib.lib().stegan_td_features(_p(x), B, T, Cc, int(win), int(shift), int(avg_window), int(pad_windowing), _p(feats),
                                        _p(scratch), _stream()), "stegan_td_features")
    return feats


class _AverageFilterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, window: int, pad_signal: bool):
        x = _gpu_f32(x, "x")
        T = x.shape[-1]
        rows = x.numel() // T
        To = T if pad_signal else T - window + 1
        if (pad_signal and window // 2 >= T) or To < 1:
            raise RuntimeError(f"Argument #4: Padding size should be less than the corresponding input dimension, but got: "
                               f"padding ({window // 2}, {window // 2}) at dimension {x.dim() - 1} of input {list(x.shape)}")
        y = torch.empty(x.shape[:-1] + (To,), device=x.device, dtype=torch.float32)
        check(_lib.lib().stegan_average_filter_fwd(_p(x), _p(y), rows, T, window, int(pad_signal), _stream()),
              "stegan_average_filter_fwd")
        ctx.dims = (rows, T, window, pad_signal, x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        rows, T, window, pad_signal, shape = ctx.dims
        dy = _gpu_f32(dy, "grad_output")
        dx = torch.empty(shape, device=dy.device, dtype=torch.float32)
        check(_lib.lib().stegan_average_filter_bwd(_p(dy), _p(dx), rows, T, window, int(pad_signal), _stream()),
              "stegan_average_filter_bwd")
        return dx, None, None


def average_filter(x: Tensor, window: int = 9, pad_signal: bool = True) -> Tensor:
    return _AverageFilterFn.apply(x, int(window), bool(pad_signal))


class _SULossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, target, pred, weight: float = 1.0, unit_grad: bool = False):
        ctx.unit_grad = bool(unit_grad)             # see _MultiTDFn.forward
        target = _gpu_f32(target, "speech_unit_target")
        pred = _gpu_f32(pred, "speech_unit_pred")
        D = pred.shape[-1]
        rows = pred.numel() // D
        loss = pooled_zeros((), pred.device)
        ws = torch.empty(rows, device=pred.device, dtype=torch.float32)
        dp = torch.empty_like(pred) if ctx.needs_input_grad[1] else None
        check(_lib.lib().stegan_su_loss_fwd_bwd(_p(target), _p(pred), rows, D, float(weight), _p(loss), _p(dp), _p(ws), _stream()),
              "stegan_su_loss_fwd_bwd")
        ctx.save_for_backward(dp)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dp,) = ctx.saved_tensors
        if dp is None:
            return None, None, None, None
        return None, (dp if ctx.unit_grad else _chain(dp, g)), None, None


def speech_unit_loss(target, pred, weight: float = 1.0, unit_grad: bool = False):
    return _SULossFn.apply(target.detach(), pred, float(weight), bool(unit_grad))


class _PhonemeCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, silence_index: int, weight: float = 1.0, unit_grad: bool = False):
        ctx.unit_grad = bool(unit_grad)             # see _MultiTDFn.forward
        logits = _gpu_f32(logits, "phoneme_pred")
        if not targets.is_cuda or targets.dtype != torch.int64:
            raise RuntimeError("ste_gan_amd: phoneme targets must be a cuda int64 tensor")
        targets = targets.contiguous()
        P = logits.shape[-1]
        rows = logits.numel() // P
        if targets.numel() != rows:
            raise RuntimeError(f"Expected target size {list(logits.shape[:-1])}, got {list(targets.shape)}")
        dev = logits.device
        loss = pooled_zeros((), dev)
        counters = pooled_zeros(4, dev, torch.int32)
        argmax = torch.empty(logits.shape[:-1], device=dev, dtype=torch.int64)
        ws = torch.empty(rows, device=dev, dtype=torch.float32)
        dl = torch.empty_like(logits) if ctx.needs_input_grad[0] else None
        check(_lib.lib().stegan_phoneme_ce_fwd_bwd(_p(logits), _p(targets), rows, P, silence_index, float(weight), _p(loss),
                                                   _p(counters), _p(dl), _p(argmax), _p(ws), _stream()),
              "stegan_phoneme_ce_fwd_bwd")
        ctx.save_for_backward(dl)
        ctx.mark_non_differentiable(counters, argmax)
        return loss, counters, argmax

    @staticmethod
    def backward(ctx, g, _c, _a):
        (dl,) = ctx.saved_tensors
        if dl is None:
            return None, None, None, None, None
        return (dl if ctx.unit_grad else _chain(dl, g)), None, None, None, None


def phoneme_ce(logits, targets, silence_index, weight: float = 1.0, unit_grad: bool = False):
    return _PhonemeCEFn.apply(logits, targets, silence_index, float(weight), bool(unit_grad))


def scalar_combine(ins, rows):
    """[sum_j rows[i][j] * ins[j] for i] as 0-dim device tensors, ONE launch (stegan_scalar_combine): the step's loss terms
    -> its reported scalars.  ``ins``: 0-dim cuda float tensors (detached), ``rows``: coefficient rows."""
    n_in, n_out = len(ins), len(rows)
    ins = [_gpu_f32(t.detach().reshape(()), "loss term") for t in ins]
    outs = torch.empty(n_out, device=ins[0].device, dtype=torch.float32)
    views = [outs[i] for i in range(n_out)]
    coeff = (C.c_float * (n_in * n_out))(*[float(c) for r in rows for c in r])
    check(_lib.lib().stegan_scalar_combine(n_in, _ptr_array(ins), n_out, _ptr_array(views), coeff, _stream()),
          "stegan_scalar_combine")
    return views


class _FanOutFn(torch.autograd.Function):
    """n aliases of x, one per consumer; backward = ONE fixed-order sum of the consumers' gradients (stegan_sum_n) instead of
    the autograd engine's ATen add per extra consumer."""

    @staticmethod
    def forward(ctx, x, n):
        outs = tuple(x.view_as(x) for _ in range(n))
        h = _image_of(x, x.shape[1], x.shape[2]) if x.dim() == 3 else None
        if h is not None:
            for o in outs:
                _attach_image(o, h)                 # the aliases are the same bytes: the producer's image serves them all
        return outs

    @staticmethod
    def backward(ctx, *gs):
        live = [_gpu_f32(g, "grad") for g in gs if g is not None]
        if not live:
            return None, None
        if len(live) == 1:
            return live[0], None
        out = torch.empty_like(live[0])
        acc = live
        while len(acc) > 1:                          # (more than four consumers: in groups, left to right)
            head, acc = acc[:4], acc[4:]
            check(_lib.lib().stegan_sum_n(len(head), _ptr_array(head), _p(out), out.numel(), _stream()), "stegan_sum_n")
            acc = [out] + acc
        return out, None


FAN_OUT = os.environ.get("STEGAN_FAN_OUT", "1") != "0"


def fan_out(x: Tensor, n: int):
    """``n`` aliases of ``x`` for its ``n`` autograd consumers (a tuple; ``x`` itself n times when no gradient flows, on the
    CPU, or with STEGAN_FAN_OUT=0 -- the engine's own accumulation then)."""
    if not (FAN_OUT and n > 1 and x.is_cuda and x.requires_grad and torch.is_grad_enabled()):
        return (x,) * n
    return _FanOutFn.apply(x, n)


_ONES: dict = {}


def one_like(t: Tensor) -> Tensor:
    """A persistent 0-dim 1.0 on ``t``'s device: the upstream gradient of a scalar root without autograd's ones_like fill."""
    o = _ONES.get(t.device)
    if o is None:
        o = _ONES[t.device] = torch.ones((), device=t.device, dtype=torch.float32)
    return o


# ---------------------------------------------------------------------------------------------
# autograd: frozen-weight ops of the EMG encoder (gradient w.r.t. the activation only)
# ---------------------------------------------------------------------------------------------
class _FrozenConvFn(torch.autograd.Function):
    """y = out_act(scale * (w (*) x) + bias + residual) with constant w / scale / bias: backward yields dx (+ dres)."""

    @staticmethod
    def forward(ctx, x, w, scale, bias, residual, cfg: ConvCfg, prepared=None, hand=None):
        wf, wb = prepared if prepared is not None else (None, None)
        y = conv1d_fwd(x, w, scale, bias, residual, cfg, wf)
        ctx.cfg = cfg
        ctx.wb = wb
        ctx.has_res = residual is not None
        ctx.tin = x.shape[2]
        ctx.hand = hand
        ctx.save_for_backward(w, scale, y if cfg.out_act != ACT_NONE else None, x if cfg.in_act == ACT_RELU else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        cfg: ConvCfg = ctx.cfg
        w, scale, y, x_mask = ctx.saved_tensors
        dy = _gpu_f32(dy, "grad_output")
        if cfg.out_btc:
            dy = dy.transpose(1, 2).contiguous()
            if y is not None:
                y = y.transpose(1, 2).contiguous()
        if cfg.out_act != ACT_NONE:
            dy = act_bwd(dy, y, cfg.out_act)
        dx = None
        if ctx.needs_input_grad[0]:
            add = None
            if ctx.hand is not None and ctx.hand[0] == TAKE:            # gradient hand-over (see GradHandover)
                hand = ctx.hand[1]
                hand.taken = True
                add, hand.parked = hand.parked, None
                if add is not None and (add.shape[0] != dy.shape[0] or add.shape[1] != cfg.cin
                                        or add.shape[2] != ctx.tin * cfg.upsample or not add.is_contiguous()):
                    raise RuntimeError("ste_gan_amd: parked gradient does not match the taker's input gradient")
            dxe = conv1d_bwd_data(dy, w, scale, x_mask, add, cfg, ctx.tin, ctx.wb)
            if ctx.hand is not None and ctx.hand[0] == PARK_DX and not ctx.hand[1].taken:
                ctx.hand[1].parked = dxe
            else:
                dx = fold_upsample2(dxe) if cfg.upsample == 2 else dxe
        dres = None
        if ctx.has_res and ctx.needs_input_grad[4]:
            if ctx.hand is not None and ctx.hand[0] == PARK_RES and not ctx.hand[1].taken and dy.is_contiguous():
                ctx.hand[1].parked = dy
            else:
                dres = dy
        return dx, None, None, None, dres, None, None, None


_FROZEN_PREP = {}


def frozen_prepared(w: Tensor, scale: Optional[Tensor], cfg: ConvCfg):
    """Transposed copies of a FROZEN weight (its constant per-channel scale folded into wb), cached by storage address
    + in-place version.  The entry keeps `w` and `scale` alive, so their addresses cannot be handed to another tensor
    (a freed-and-reused address with an equal version counter would otherwise serve a stale copy)."""
    key = (w.data_ptr(), tuple(w.shape), cfg.groups, scale.data_ptr() if scale is not None else 0)
    ent = _FROZEN_PREP.get(key)
    stamp = (w._version, scale._version if scale is not None else 0, COMPUTE_DTYPE)
    if ent is None or ent[0] != stamp:
        ent = (stamp, prep_weights(w.detach(), cfg, ent[1] if ent is not None else None, scale), w, scale)
        _FROZEN_PREP[key] = ent
    return ent[1]


def frozen_conv1d(x, w, scale, bias, residual, cfg: ConvCfg, hand=None) -> Tensor:
    return _FrozenConvFn.apply(x, w, scale, bias, residual, cfg, frozen_prepared(w, scale, cfg), hand)


class _RelPosAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, rel_emb, H: int, max_rel: int):
        qkv = _gpu_f32(qkv, "qkv")
        B, C3, T = qkv.shape
        Cm = C3 // 3
        A = Cm // H
        probs = torch.empty((B, H, T, T), device=qkv.device, dtype=torch.float32)
        out = torch.empty((B, Cm, T), device=qkv.device, dtype=torch.float32)
        outh = _new_image(out, Cm, 1, T)
        check(_lib.lib().stegan_relpos_attention_fwd_img(_p(qkv), _p(rel_emb), _p(probs), _p(out), _p(outh), B, H, A, T, max_rel,
                                                         _stream()), "stegan_relpos_attention_fwd_img")
        _attach_image(out, outh)
        ctx.dims = (B, H, A, T, max_rel)
        ctx.save_for_backward(qkv, rel_emb, probs)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, rel_emb, probs = ctx.saved_tensors
        B, H, A, T, max_rel = ctx.dims
        dout = _gpu_f32(dout, "grad_output")
        ws = torch.empty_like(probs)
        dqkv = torch.empty_like(qkv)
        dqkvh = _new_image(dqkv, 3 * H * A, 1, T)
        check(_lib.lib().stegan_relpos_attention_bwd_img(_p(qkv), _p(rel_emb), _p(probs), _p(dout), _p(ws), _p(dqkv), _p(dqkvh),
                                                         B, H, A, T, max_rel, _stream()), "stegan_relpos_attention_bwd_img")
        _attach_image(dqkv, dqkvh)
        return dqkv, None, None, None


def relpos_attention(qkv, rel_emb, H, max_rel):
    return _RelPosAttnFn.apply(qkv, rel_emb, H, max_rel)


class _AddLayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, gamma, beta, eps: float, hand=None):
        ctx.hand = hand
        x = _gpu_f32(x, "x")
        residual = _gpu_f32(residual, "residual")
        B, Cm, T = x.shape
        y = torch.empty_like(x)
        xhat = torch.empty_like(x)
        rstd = torch.empty(B * T, device=x.device, dtype=torch.float32)
        yh = _new_image(y, Cm, 1, T)
        check(_lib.lib().stegan_add_layernorm_fwd_img(_p(x), _p(residual), _p(gamma), _p(beta), _p(y), _p(yh), _p(xhat), _p(rstd),
                                                      B, Cm, T, eps, _stream()), "stegan_add_layernorm_fwd_img")
        _attach_image(y, yh)
        ctx.save_for_backward(xhat, rstd, gamma)
        return y

    @staticmethod
    def backward(ctx, dy):
        xhat, rstd, gamma = ctx.saved_tensors
        dy = _gpu_f32(dy, "grad_output")
        B, Cm, T = dy.shape
        dz = torch.empty_like(dy)
        dzh = _new_image(dz, Cm, 1, T)
        check(_lib.lib().stegan_add_layernorm_bwd_img(_p(dy), _p(xhat), _p(rstd), _p(gamma), _p(dz), _p(dzh), B, Cm, T, _stream()),
              "stegan_add_layernorm_bwd_img")
        _attach_image(dz, dzh)
        hand = ctx.hand[1] if ctx.hand is not None and ctx.hand[0] == PARK_DX else None
        if hand is not None and not hand.taken and ctx.needs_input_grad[0]:
            hand.parked = dz                       # the conv that also reads x adds it inside its input-gradient kernel
            return None, dz, None, None, None, None
        return dz, dz, None, None, None, None


def add_layernorm(x, residual, gamma, beta, eps=1e-5, hand=None):
    """y = LayerNorm_C(x + residual) * gamma + beta on (B, C, T).  hand = (PARK_DX, GradHandover): x also feeds a conv whose
    backward runs later and takes this node's gradient for x as ``add`` (transformer residuals)."""
    return _AddLayerNormFn.apply(x, residual, gamma, beta, eps, hand)
