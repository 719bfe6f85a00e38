"""Flat parameter arenas + fused AdamW.

All trainable tensors of a network are re-homed as views into ONE contiguous fp32 arena (and their
``.grad`` into a second one), so that (a) the optimiser is a single HBM-streaming kernel launch per
network instead of one per tensor, (b) ``zero_grad`` is one memset and (c) data-parallel training
all-reduces exactly two flat buffers per step.  ``FusedAdamW`` keeps the
hyper-parameters of ``ste_gan.OPTIMIZER`` (constants.py:57) and reads / writes ``torch.optim.AdamW``
style state dicts so ``checkpoint-*.pt`` files interchange with the reference (train.py:431-436).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Iterable, List

import torch
import torch.nn as nn

from . import _lib
from ._lib import check

_ALIGN = 4          # floats: every tensor starts 16-byte aligned inside the arena


class FlatParams:
    """Re-homes ``module``'s parameters (in ``parameters()`` order) into flat param / grad arenas."""

    def __init__(self, module: nn.Module):
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError("module has no trainable parameters")
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatParams: move the module to the GPU first (ste_gan_amd has no CPU path)")
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(total, device=dev, dtype=torch.float32)
        self.params, self.offsets = params, offs
        with torch.no_grad():
            for p, o in zip(params, offs):
                view = self.flat[o:o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grad[o:o + p.numel()].view_as(p)
        self.numel = total
        self.epoch = 0                         # bumped by every fused AdamW launch on this arena (layers.conv._owner_epoch)
        for m in module.modules():
            if hasattr(m, "grad_sink"):        # spectral-norm convs may accumulate straight into these views
                m.grad_sink = True
            if hasattr(m, "_prep"):            # layers that keep prepared weight copies of their own
                m._owner = self

    zero_in_step = False       # the optimiser kernels zero the arena as they consume it (FusedAdamW.zero_in_step)
    clean = False              # the arena is known to be all zero: set by a consuming AdamW launch (or a fill), cleared by
                               # zero_grad() -- which skips its fill only while this holds, so gradients a dead step left
                               # behind are never added into the next one

    def zero_grad(self):
        if not (self.zero_in_step and self.clean):
            self.grad.zero_()
        self.clean = False                                   # the backward pass about to run writes into the arena
        for p, o in zip(self.params, self.offsets):          # autograd may have swapped the tensor: re-attach the view
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view_as(p)

    def gather_grads(self):
        """Fold any gradient autograd stored outside the arena back into it (defensive; normally a no-op)."""
        for p, o in zip(self.params, self.offsets):
            if p.grad is not None and p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                self.grad[o:o + p.numel()].view_as(p).copy_(p.grad)
                p.grad = self.grad[o:o + p.numel()].view_as(p)


class ConvPack:
    """Network-wide (multi-tensor) handling of the weight-normalised convolutions of one module tree.

    * ``prepare()``: ONE launch computes every weight-norm factor g/||v||, one more writes every K-major transposed
      weight copy (instead of three small launches per conv and weight version);
    * gradient sink: during backward each conv accumulates its raw weight gradient into a slice of one zeroed arena
      and its bias gradient straight into ``bias.grad`` (a view of the FlatParams gradient arena); ``finalize_grads()``
      then converts all raw gradients to (dv, dg) in one launch, writing into the ``.grad`` views.  Valid because dv, dg
      are linear in the raw gradient for fixed (v, g), so contributions of several forward passes (D sees two per
      step) can be summed first.  Spectral-norm convs keep their per-call path (their factor changes every forward).
    """

    def __init__(self, module: nn.Module, flat: "FlatParams"):
        from . import ops
        from .layers.conv import WNConv1d
        self.convs = [m for m in module.modules() if isinstance(m, WNConv1d)]
        if not self.convs:
            raise ValueError("no weight-normalised convolutions in this module")
        dev = flat.flat.device
        L = _lib.lib()
        self.flat = flat
        sizes = [c.weight_v.numel() for c in self.convs]
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + (n + _ALIGN - 1) // _ALIGN * _ALIGN)
        self.dw = torch.zeros(offs[-1], device=dev, dtype=torch.float32)
        rows = [c.cfg.cout for c in self.convs]
        self.scale = torch.empty(sum(rows), device=dev, dtype=torch.float32)
        so = [0]
        for r in rows:
            so.append(so[-1] + r)
        self.max_rows = max(rows)
        n = len(self.convs)
        self.n = n

        def ptrs(vals):
            return torch.tensor(vals, dtype=torch.int64).to(dev)

        v_p, g_p, s_p, dw_p, dv_p, dg_p, cols = [], [], [], [], [], [], []
        prep_rows, self.max_tiles = [], 1
        self._keep = []
        for i, c in enumerate(self.convs):
            v, g = c.weight_v, c.weight_g
            if v.grad is None or g.grad is None or c.bias.grad is None:
                raise RuntimeError("ConvPack needs FlatParams gradient views on every parameter")
            d = ops._desc(c.cfg, 1, 1, 1)
            wf = torch.zeros(int(L.stegan_conv_prep_floats(C.byref(d), 0)), device=dev, dtype=torch.float32)
            wb = torch.zeros(int(L.stegan_conv_prep_floats(C.byref(d), 1)), device=dev, dtype=torch.float32)
            sc = self.scale[so[i]:so[i + 1]]
            dwi = self.dw[offs[i]:offs[i] + sizes[i]]
            c._pack = (self, sc, wf, wb, dwi)
            self._keep.append((wf, wb))
            v_p.append(v.data_ptr()); g_p.append(g.data_ptr()); s_p.append(sc.data_ptr()); dw_p.append(dwi.data_ptr())
            dv_p.append(v.grad.data_ptr()); dg_p.append(g.grad.data_ptr()); cols.append(v.numel() // v.shape[0])
            G = c.cfg.groups
            cog, cig, K = c.cfg.cout // G, c.cfg.cin // G, c.cfg.k
            te, tc = (cig * K + 31) // 32, (cog + 31) // 32
            self.max_tiles = max(self.max_tiles, G * te * tc)
            ldf, ldb = (cog + 63) // 64 * 64, (cig + 63) // 64 * 64
            prep_rows.append((v.data_ptr(), sc.data_ptr(), wf.data_ptr(), wb.data_ptr(), cog, cig, K, G, ldf, ldb, te, tc))
        self.t_v, self.t_g, self.t_s, self.t_dw, self.t_dv, self.t_dg = map(ptrs, (v_p, g_p, s_p, dw_p, dv_p, dg_p))
        self.t_rows = torch.tensor(rows, dtype=torch.int32).to(dev)
        self.t_cols = torch.tensor(cols, dtype=torch.int32).to(dev)
        import struct
        raw = b"".join(struct.pack("<4q8i", *r) for r in prep_rows)
        self.t_prep = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self._ptr_key = tuple(v_p + g_p + dv_p + dg_p)
        self.t_prep_h = None             # bf16 prep table, built when the bf16 compute mode is first used
        self.version = 0                 # bumped whenever this network's weights change (optimizer step, load_state_dict)
        self.prepared_version = None
        module.register_load_state_dict_post_hook(lambda mod, keys: self.invalidate())

    def invalidate(self):
        self.version += 1

    def _check(self):
        key = tuple([c.weight_v.data_ptr() for c in self.convs] + [c.weight_g.data_ptr() for c in self.convs] +
                    [c.weight_v.grad.data_ptr() for c in self.convs] + [c.weight_g.grad.data_ptr() for c in self.convs])
        if key != self._ptr_key:
            raise RuntimeError("parameter storage moved after ConvPack was built (rebuild FlatParams / FusedAdamW)")

    def _build_bf16_table(self):
        """bf16 companions (attribute ``h`` of the fp32 copies, ops._bf16_of) + the device table of
        stegan_conv_prep_bf16_multi."""
        import struct
        from . import ops
        L = _lib.lib()
        dev = self.flat.flat.device
        rows, most = [], 1
        for c in self.convs:
            _, sc, wf, wb, _ = c._pack
            d = ops._desc(c.cfg, 1, 1, 1)
            nf, nb = int(L.stegan_conv_prep_bf16_elems(C.byref(d), 0)), int(L.stegan_conv_prep_bf16_elems(C.byref(d), 1))
            if c.cfg.k >= 48:
                raise ValueError("ste_gan_amd: bf16 weight copies support kernel sizes < 48")
            wf.h = torch.zeros(nf, device=dev, dtype=ops.HALF_DTYPE)     # zero ONCE: the prep kernel writes live elements only
            wb.h = torch.zeros(nb, device=dev, dtype=ops.HALF_DTYPE)
            G = c.cfg.groups
            cog, cig, K = c.cfg.cout // G, c.cfg.cin // G, c.cfg.k
            up = lambda a, b: (a + b - 1) // b * b
            rows.append((c.weight_v.data_ptr(), sc.data_ptr(), wf.h.data_ptr(), wb.h.data_ptr(), cog, cig, K, G,
                         up(cog, 128), up(cig, 64), up(cig, 128), up(cog, 64), nf, nb))
            most = max(most, int(L.stegan_conv_prep_bf16_tiles(C.byref(d))))
        raw = b"".join(struct.pack("<4q8i2q", *r) for r in rows)
        self.t_prep_h = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self.prep_h_blocks = max(1, min(4096, most))              # tiles of the largest conv (grid-stride beyond that)
        self._h_dtype = ops.HALF_DTYPE

    def prepare(self):
        """Scales + transposed copies for the CURRENT weights (2 launches for the whole network; 3 in bf16 mode)."""
        from . import ops
        key = (self.version, ops.COMPUTE_DTYPE)
        if self.prepared_version == key:
            return
        self._check()
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        P = lambda t: C.c_void_p(t.data_ptr())
        L = _lib.lib()
        check(L.stegan_wn_scales(self.n, P(self.t_v), P(self.t_g), P(self.t_s), P(self.t_rows), P(self.t_cols),
                                 self.max_rows, s), "stegan_wn_scales")
        if not ops.bf16_enabled():          # bf16 mode maintains only the bf16 copies (ops.prep_weights)
            check(L.stegan_conv_prep_multi(self.n, P(self.t_prep), self.max_tiles, s), "stegan_conv_prep_multi")
        if ops.bf16_enabled():
            if self.t_prep_h is None or self._h_dtype != ops.HALF_DTYPE:
                self._build_bf16_table()
            check(L.stegan_conv_prep_bf16_multi(self.n, P(self.t_prep_h), self.prep_h_blocks, s), "stegan_conv_prep_bf16_multi")
        self.prepared_version = key

    zero_in_step = False
    clean = False              # as FlatParams.clean, for the raw dW arena (consumed by stegan_wn_backward_z)

    def zero(self):
        if not (self.zero_in_step and self.clean):
            self.dw.zero_()
        self.clean = False

    def finalize_grads(self):
        """raw dw -> (dv, dg) for every conv, written into the .grad views (1 launch)."""
        from . import ops
        ops.wgrad_join()                       # weight gradients launched on a side stream (ops._wgrad_sink)
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        P = lambda t: C.c_void_p(t.data_ptr())
        check(_lib.lib().stegan_wn_backward_z(self.n, P(self.t_v), P(self.t_g), P(self.t_dw), P(self.t_dv), P(self.t_dg),
                                              P(self.t_rows), P(self.t_cols), self.max_rows, int(self.zero_in_step), s),
              "stegan_wn_backward_z")
        self.clean = bool(self.zero_in_step)


class FusedAdamW:
    """AdamW(lr=2e-4, betas=(.8,.99), eps=1e-8, weight_decay=1e-2) over a FlatParams arena."""

    def __init__(self, flat: FlatParams, lr=2e-4, betas=(0.8, 0.99), eps=1e-8, weight_decay=1e-2, pack: "ConvPack" = None):
        self.flat = flat
        self.pack = pack            # optional multi-tensor conv handling (see ConvPack)
        self.lr, self.betas, self.eps, self.weight_decay = lr, tuple(betas), eps, weight_decay
        self.initial_lr = lr
        self.exp_avg = torch.zeros_like(flat.flat)
        self.exp_avg_sq = torch.zeros_like(flat.flat)
        self.step_count = 0
        # device {lr, 1-b1^t, 1-b2^t, t}: t and the bias corrections advance on the device (stegan_adamw_advance); the
        # host only fills lr when it changes (a by-value fill kernel, stream ordered) and mirrors t in step_count
        self._hyper = torch.zeros(4, device=flat.flat.device, dtype=torch.float32)
        self._lr_on_device = None
        self.param_groups = [dict(lr=lr, betas=self.betas, eps=eps, weight_decay=weight_decay, initial_lr=lr)]

    def zero_grad(self, set_to_none: bool = False):
        self.flat.zero_grad()
        if self.pack is not None:
            self.pack.zero()

    def set_zero_in_step(self, on: bool) -> None:
        """With ``on`` the AdamW kernel zeroes the gradient arena (and the weight-norm backward the raw dW arena) as it
        consumes it, and ``zero_grad()`` launches nothing: ``.grad`` then reads zero after ``step()`` -- unlike
        torch.optim, which keeps it until ``zero_grad()`` -- so this is for the replayed train step, where nobody looks
        (two fills of 47 + 94 MB sat at the start of every step).  Switching it on zeroes the arenas once, eagerly."""
        on = bool(on)
        if on and not self.flat.zero_in_step:
            self.discard_grads()
        self.flat.zero_in_step = on
        if self.pack is not None:
            self.pack.zero_in_step = on

    def discard_grads(self) -> None:
        """Zero both gradient arenas with fills, now (switching zero-on-consume on; the exception path of a train step)."""
        self.flat.grad.zero_()
        self.flat.clean = True
        if self.pack is not None:
            self.pack.dw.zero_()
            self.pack.clean = True

    def finalize_grads(self):
        """Call between backward() and step() / the gradient all-reduce: joins the streams on which backward nodes wrote
        gradients straight into the arena (ops.wgrad_join) and, with a ConvPack attached, turns the raw direction
        gradients into (dv, dg)."""
        from . import ops
        ops.wgrad_join()
        if self.pack is not None:
            self.pack.finalize_grads()

    def sync_lr(self):
        """Push the current learning rate to the device if it changed (never inside a graph capture: the captured step
        reads hyper[0], so a later change only needs this stream-ordered fill ahead of the next replay)."""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_on_device:
            self._hyper[0:1].fill_(lr)
            self._lr_on_device = lr

    def _set_device_step(self, t: int):
        self._hyper.view(torch.int32)[3:4].fill_(int(t))

    def reset_state(self):
        """Forget the moments and the step count (a freshly constructed optimiser over the same arenas)."""
        self.exp_avg.zero_(); self.exp_avg_sq.zero_()
        self.step_count = 0
        self._set_device_step(0)

    def stage_hyper(self, scaler: "GradScaler" = None):
        """Advance the step counter and the bias corrections on the device (one tiny launch, graph-capturable).  With a
        ``scaler`` the device decides whether this step counts (skip on non-finite gradients); ``step_count`` on the host
        then counts ATTEMPTED steps -- ``device_step_count()`` reads the applied ones."""
        self.step_count += 1
        if scaler is not None:
            self._may_skip = True              # the device may skip steps from here on: checkpoints read ITS count
        if not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        check(_lib.lib().stegan_adamw_advance_s(C.c_void_p(self._hyper.data_ptr()), float(self.betas[0]), float(self.betas[1]),
                                                C.c_void_p(scaler.state.data_ptr()) if scaler is not None else None,
                                                C.c_void_p(torch.cuda.current_stream().cuda_stream)), "stegan_adamw_advance_s")

    def device_step_count(self) -> int:
        """The step count the device holds (a host sync): differs from ``step_count`` by the steps a GradScaler skipped."""
        return int(self._hyper.view(torch.int32)[3].item())

    def launch(self, grad_scale: float = 1.0, scaler: "GradScaler" = None):
        f = self.flat
        check(_lib.lib().stegan_adamw_step_zs(
            C.c_void_p(f.flat.data_ptr()), C.c_void_p(f.grad.data_ptr()), C.c_void_p(self.exp_avg.data_ptr()),
            C.c_void_p(self.exp_avg_sq.data_ptr()), f.numel, C.c_void_p(self._hyper.data_ptr()),
            self.betas[0], self.betas[1], self.eps, self.weight_decay, grad_scale, int(f.zero_in_step),
            C.c_void_p(scaler.state.data_ptr()) if scaler is not None else None,
            C.c_void_p(torch.cuda.current_stream().cuda_stream)), "stegan_adamw_step_zs")
        self.flat.clean = bool(f.zero_in_step)      # (zero-on-consume: .grad reads 0 from here on, see set_zero_in_step)
        self.flat.epoch += 1           # parameters changed behind torch's back: this network's prepared copies are stale
        if self.pack is not None:
            self.pack.invalidate()

    def step(self, grad_scale: float = 1.0, finalized: bool = False, scaler: "GradScaler" = None):
        """``scaler`` (GradScaler below) = ``scaler.step(optimizer)`` of the reference (train.py:198, 266): the gradients are
        tested for inf / NaN and unscaled inside the AdamW kernel, or the whole step is skipped -- all on the device."""
        if not finalized:
            self.finalize_grads()
        self.flat.gather_grads()
        if scaler is not None:
            scaler.check(self.flat.grad)
        self.stage_hyper(scaler)
        self.launch(grad_scale, scaler)

    # ---- torch.optim.AdamW compatible checkpoint format -------------------------------------
    def state_dict(self):
        state = {}
        # (with a GradScaler the device skips steps the host cannot know of: the applied count lives there)
        steps = self.device_step_count() if getattr(self, "_may_skip", False) else self.step_count
        if steps > 0:
            for i, (p, o) in enumerate(zip(self.flat.params, self.flat.offsets)):
                state[i] = {"step": torch.tensor(float(steps)),
                            "exp_avg": self.exp_avg[o:o + p.numel()].view_as(p).clone(),
                            "exp_avg_sq": self.exp_avg_sq[o:o + p.numel()].view_as(p).clone()}
        g = self.param_groups[0]
        group = dict(lr=g["lr"], betas=self.betas, eps=self.eps, weight_decay=self.weight_decay, amsgrad=False,
                     maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                     initial_lr=g.get("initial_lr", self.initial_lr), params=list(range(len(self.flat.params))))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        self.param_groups[0]["lr"] = g["lr"]
        self.param_groups[0]["initial_lr"] = g.get("initial_lr", g["lr"])
        self.betas, self.eps, self.weight_decay = tuple(g["betas"]), g["eps"], g["weight_decay"]
        steps = set()
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(self.flat.params, self.flat.offsets)):
                st = sd["state"].get(i)
                if st is None:
                    continue
                self.exp_avg[o:o + p.numel()].view_as(p).copy_(st["exp_avg"])
                self.exp_avg_sq[o:o + p.numel()].view_as(p).copy_(st["exp_avg_sq"])
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("FusedAdamW needs one common step count across parameters")
        self.step_count = steps.pop() if steps else 0
        self._set_device_step(self.step_count)
        self._lr_on_device = None


class GradScaler:
    """torch.cuda.amp.GradScaler as ste_gan/train.py:151,198-199,266-268 drives it, with its state and every decision on the
    device (include/stegan_hip.h, "loss scaling on the device") so that a captured train step replays it:

        scaler.attach()                       # the loss kernels multiply the gradients they write by the scale
        ... loss_D.backward() ...; optD.step(scaler=scaler)        # = scaler.scale(loss_D).backward(); scaler.step(optD)
        ... loss_G.backward() ...; optG.step(scaler=scaler)
        scaler.update()                       # once per iteration: back off after a skipped step, grow every `interval`

    Needed by the fp16 operand mode (gradients below 6e-8 vanish when they are rounded to fp16 operands, above 65504 they
    become inf); in the fp32 and bf16 modes a power-of-two scale is exact and changes nothing."""

    def __init__(self, device, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5,
                 growth_interval: int = 2000):
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self.state = torch.zeros(8, device=device, dtype=torch.float32)
        self._set(float(init_scale), 0)
        self._attached_to = None

    def _set(self, scale: float, tracker: int):
        self.state[0:2] = torch.tensor([scale, 1.0 / scale], dtype=torch.float32)
        self.state.view(torch.int32)[2:3].fill_(int(tracker))

    def attach(self) -> None:
        lib = _lib.lib()
        check(lib.stegan_loss_grad_scale(C.c_void_p(self.state.data_ptr())), "stegan_loss_grad_scale")
        self._attached_to = lib

    def detach(self) -> None:
        if self._attached_to is not None:
            self._attached_to.stegan_loss_grad_scale(None)
            self._attached_to = None

    def check(self, grads: torch.Tensor) -> None:
        check(_lib.lib().stegan_grad_scaler_check(C.c_void_p(grads.data_ptr()), grads.numel(), C.c_void_p(self.state.data_ptr()),
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)), "stegan_grad_scaler_check")

    def update(self) -> None:
        check(_lib.lib().stegan_grad_scaler_update(C.c_void_p(self.state.data_ptr()), self.growth_factor, self.backoff_factor,
                                                   self.growth_interval, C.c_void_p(torch.cuda.current_stream().cuda_stream)),
              "stegan_grad_scaler_update")

    # ---- host views (each a device sync: logging, checkpoints, tests)
    def get_scale(self) -> float:
        return float(self.state[0].item())

    def skipped_steps(self) -> int:
        return int(self.state.view(torch.int32)[5].item())

    def state_dict(self):
        """Same keys as torch.cuda.amp.GradScaler.state_dict()."""
        return {"scale": self.get_scale(), "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": int(self.state.view(torch.int32)[2].item())}

    def load_state_dict(self, sd):
        self.growth_factor, self.backoff_factor = float(sd["growth_factor"]), float(sd["backoff_factor"])
        self.growth_interval = int(sd["growth_interval"])
        self._set(float(sd["scale"]), int(sd["_growth_tracker"]))


class ExponentialLR:
    """``torch.optim.lr_scheduler.ExponentialLR`` as the reference uses it (train.py:98-104,470-472): stepped per epoch,
    and CHAINED from the group's current lr like torch's -- ``step()`` multiplies ``param_groups[0]["lr"]`` by gamma,
    except for the very first call of a fresh schedule (``last_epoch`` -1 -> 0), which leaves the initial lr in place.
    On resume (``last_epoch = start_epoch``) the constructor's own step therefore yields loaded_lr * gamma, exactly what
    the reference's pinned torch 2.0.1 computes from a loaded optimiser state (``get_lr``: ``group["lr"]`` if
    ``last_epoch == 0`` else ``group["lr"] * gamma``), so ``checkpoint-*.pt`` files interchange lr-exactly.  (Newer
    torch releases skip the constructor's multiplication on resume; the pinned behaviour is the one mirrored.)"""

    def __init__(self, optimizer: FusedAdamW, gamma: float, last_epoch: int = -1):
        self.optimizer, self.gamma = optimizer, gamma
        g = optimizer.param_groups[0]
        if last_epoch == -1:
            g.setdefault("initial_lr", g["lr"])
        elif "initial_lr" not in g:
            raise KeyError("param 'initial_lr' is not specified in param_groups[0] when resuming an optimizer")
        self.last_epoch = last_epoch
        self.step()

    def step(self):
        self.last_epoch += 1
        g = self.optimizer.param_groups[0]
        if self.last_epoch > 0:
            g["lr"] = g["lr"] * self.gamma

    def get_last_lr(self):
        return [self.optimizer.param_groups[0]["lr"]]
