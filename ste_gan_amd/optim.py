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

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):          # autograd may have swapped the tensor: re-attach the view
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view_as(p)

    def gather_grads(self):
        """Fold any gradient autograd stored outside the arena back into it (defensive; normally a no-op)."""
        for p, o in zip(self.params, self.offsets):
            if p.grad is not None and p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                self.grad[o:o + p.numel()].view_as(p).copy_(p.grad)
                p.grad = self.grad[o:o + p.numel()].view_as(p)


class FusedAdamW:
    """AdamW(lr=2e-4, betas=(.8,.99), eps=1e-8, weight_decay=1e-2) over a FlatParams arena."""

    def __init__(self, flat: FlatParams, lr=2e-4, betas=(0.8, 0.99), eps=1e-8, weight_decay=1e-2):
        self.flat = flat
        self.lr, self.betas, self.eps, self.weight_decay = lr, tuple(betas), eps, weight_decay
        self.initial_lr = lr
        self.exp_avg = torch.zeros_like(flat.flat)
        self.exp_avg_sq = torch.zeros_like(flat.flat)
        self.step_count = 0
        self._hyper = torch.zeros(3, device=flat.flat.device, dtype=torch.float32)
        self._hyper_host = torch.zeros(3, dtype=torch.float32).pin_memory()
        self.param_groups = [dict(lr=lr, betas=self.betas, eps=eps, weight_decay=weight_decay, initial_lr=lr)]

    def zero_grad(self, set_to_none: bool = False):
        self.flat.zero_grad()

    def stage_hyper(self):
        """Advance the step counter and upload {lr, 1-b1^t, 1-b2^t} (async H2D from pinned memory)."""
        self.step_count += 1
        lr = self.param_groups[0]["lr"]
        self._hyper_host[0] = lr
        self._hyper_host[1] = 1.0 - self.betas[0] ** self.step_count
        self._hyper_host[2] = 1.0 - self.betas[1] ** self.step_count
        self._hyper.copy_(self._hyper_host, non_blocking=True)

    def launch(self, grad_scale: float = 1.0):
        f = self.flat
        check(_lib.lib().stegan_adamw_step(
            C.c_void_p(f.flat.data_ptr()), C.c_void_p(f.grad.data_ptr()), C.c_void_p(self.exp_avg.data_ptr()),
            C.c_void_p(self.exp_avg_sq.data_ptr()), f.numel, C.c_void_p(self._hyper.data_ptr()),
            self.betas[0], self.betas[1], self.eps, self.weight_decay, grad_scale,
            C.c_void_p(torch.cuda.current_stream().cuda_stream)), "stegan_adamw_step")
        from . import ops
        ops.bump_weight_epoch()        # parameters changed behind torch's back: transposed weight copies are stale

    def step(self, grad_scale: float = 1.0):
        self.flat.gather_grads()
        self.stage_hyper()
        self.launch(grad_scale)

    # ---- torch.optim.AdamW compatible checkpoint format -------------------------------------
    def state_dict(self):
        state = {}
        if self.step_count > 0:
            for i, (p, o) in enumerate(zip(self.flat.params, self.flat.offsets)):
                state[i] = {"step": torch.tensor(float(self.step_count)),
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


class ExponentialLR:
    """lr = initial_lr * gamma^epoch, stepped per epoch -- train.py:98-104,470-472."""

    def __init__(self, optimizer: FusedAdamW, gamma: float, last_epoch: int = -1):
        self.optimizer, self.gamma = optimizer, gamma
        self.last_epoch = last_epoch
        self.step()

    def step(self):
        self.last_epoch += 1
        g = self.optimizer.param_groups[0]
        g["lr"] = g["initial_lr"] * self.gamma ** self.last_epoch

    def get_last_lr(self):
        return [self.optimizer.param_groups[0]["lr"]]
