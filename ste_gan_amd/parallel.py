"""Data-parallel gradient exchange: one process per GPU, RCCL over xGMI (torch.distributed "nccl").

The batch dimension shards across ranks (independent samples; no BatchNorm in G / D, the encoder's is in
eval mode), so the only data-path exchange is the mean of the two flat gradient arenas, once per optimiser
step each.  The all-reduce is enqueued on a dedicated side stream so it overlaps
compute that does not depend on it; ``wait`` makes the compute stream depend on its completion
(stream-ordered, no host sync).  With the "gloo" backend (CPU tensors, used by the CPU tests) the same class
degrades to a blocking all-reduce.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


class FlatGradAllReduce:
    """Everything the exchange needs from the host side exists BEFORE a launch: the communication stream and the two
    events of every arena are created on first use OUTSIDE any capture and reused (``prime``).  A launch made while the
    stream is being captured (GraphedTrainStep's single-graph mode) then consists of exactly: event record -> stream wait ->
    the collective -> event record -- no event / stream construction, and no ``record_stream`` (caching-allocator
    bookkeeping on a capturing side stream; the arena outlives every step, so the call protects nothing there)."""

    def __init__(self, process_group=None):
        self.group = process_group
        self.world_size = dist.get_world_size(process_group)
        self._stream: Optional[torch.cuda.Stream] = None
        self._events = {}                       # arena address -> (ready, done)

    def prime(self, flat_grad: torch.Tensor) -> None:
        """Create the communication stream and this arena's events now (call outside a capture; ``launch`` does it
        itself when it runs eagerly first, as every train loop does: the warm-up steps precede the capture)."""
        if not flat_grad.is_cuda or dist.get_backend(self.group) != "nccl":
            return
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=flat_grad.device)
        if flat_grad.data_ptr() not in self._events:
            self._events[flat_grad.data_ptr()] = (torch.cuda.Event(), torch.cuda.Event())

    def launch(self, flat_grad: torch.Tensor):
        """SUM all-reduce of the arena (the 1/world_size factor is folded into the AdamW kernel's grad_scale)."""
        if not flat_grad.is_cuda:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            return None
        if dist.get_backend(self.group) != "nccl":
            # gloo over device memory (multi-rank rehearsals that share one card): blocking and staged through the host
            host = flat_grad.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            flat_grad.copy_(host)
            return None
        capturing = torch.cuda.is_current_stream_capturing()
        if self._stream is None or flat_grad.data_ptr() not in self._events:
            if capturing:
                raise RuntimeError("ste_gan_amd.parallel: the gradient exchange must run eagerly once (or prime()) before it is "
                                   "captured: its stream and events are not created inside a capture")
            self.prime(flat_grad)
        ready, done = self._events[flat_grad.data_ptr()]
        ready.record(torch.cuda.current_stream(flat_grad.device))
        self._stream.wait_event(ready)                      # gradients fully written before the exchange starts
        with torch.cuda.stream(self._stream):
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            done.record(self._stream)
        if not capturing:
            flat_grad.record_stream(self._stream)
        return done

    def wait(self, handle) -> None:
        if handle is not None:
            torch.cuda.current_stream().wait_event(handle)

    def close(self) -> None:
        """Drain and drop the communication stream (part of the ordered shutdown, :func:`shutdown`)."""
        if self._stream is not None:
            self._stream.synchronize()
            self._stream = None
        self._events.clear()


def shutdown(*steps) -> None:
    """Ordered end of a torch.distributed job: drain the device, release what the captured steps hold (graphs + their
    memory pool, static batches, the communication stream: ``close()`` of GraphedTrainStep / GANTrainStep), drain again,
    and only then destroy the process group.  destroy_process_group() tears down the RCCL communicator and its
    watchdog; nothing of this package may still be in flight on, or hold events of, the streams RCCL used."""
    import gc
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        torch.cuda.synchronize()
    for st in steps:
        if st is not None and hasattr(st, "close"):
            st.close()
    gc.collect()
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        torch.cuda.synchronize()
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def shard_batch(batch: dict, rank: int, world_size: int) -> dict:
    """Contiguous batch-dimension shard of a DataType-keyed batch dict for this rank."""
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v) and v.dim() >= 1:
            n = v.shape[0]
            if n % world_size:
                raise ValueError(f"batch size {n} is not divisible by world size {world_size}")
            per = n // world_size
            out[k] = v[rank * per:(rank + 1) * per]
        else:
            out[k] = v
    return out
