"""Data-parallel plumbing: one process per GPU, gradients averaged with ONE flat all-reduce.

The reference's training loop is single-process (train.py:340-400); data parallelism here is the standard
extension: every rank runs the same step on its own shard of structures (no data-path collective -- the graph
of one structure never crosses a rank) and the 1.66 M fp32 gradients (6.6 MB) are summed once per step over
RCCL (backend "nccl" on ROCm; "gloo" in the CPU tests).  One bucket: at 6.6 MB a ring all-reduce over xGMI is
latency-bound, so splitting it only adds launches.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None):
    """(rank, local_rank, world) from torchrun's environment; initialises the process group when world > 1."""
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous [lo, hi) slice of n_items for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class FlatGradients:
    """One flat fp32 buffer that holds every parameter's gradient, in parameter order.  After `reduce()` each
    `p.grad` IS a view of it (no copy back), so the optimizer that follows - the in-tree multi-tensor AdamW takes the
    views' pointers - steps on the buffer the collective left its result in.  The buffer is kept between steps (the
    6.6 MB are allocated once); `timings` receives one (start, end) HIP-event pair per collective when given."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None

    def _buffer(self, like):
        if self.flat is None or self.flat.device != like.device or self.flat.dtype != like.dtype:
            self.flat = torch.empty(self.numel, device=like.device, dtype=like.dtype)
        return self.flat

    def reduce(self, world: int, group=None, timings=None):
        if not self.params:
            return
        flat = self._buffer(self.params[0])
        views, off = [], 0
        for p in self.params:
            views.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        # gather: ONE multi-tensor copy of the gradients autograd produced (separate tensors) into the flat buffer;
        # a parameter whose gradient already is the view (nothing re-allocated it) costs nothing
        src, dst = [], []
        for p, v in zip(self.params, views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
        if dst:
            torch._foreach_copy_(dst, src)
        ev = None
        if timings is not None and flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if flat.is_cuda and dist.get_backend(group) == "gloo":  # CPU rehearsal backend: stage through host memory
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
        if ev is not None:
            ev[1].record()
            timings.append(ev)
        for p, v in zip(self.params, views):
            p.grad = v          # re-point: the optimizer reads the reduced values in place


_FLAT = {}


def allreduce_gradients(params, world: int, group=None, timings=None):
    """Average `.grad` over ranks with a single flat all-reduce (parameters without a grad count as zeros); afterwards
    every `.grad` is a view of the flat buffer (FlatGradients)."""
    params = [p for p in params if p.requires_grad]
    # MPNN_FORCE_ALLREDUCE=1: a one-rank process group still takes the collective path (rehearsal of the RCCL path on a 1-GPU box)
    forced = os.environ.get("MPNN_FORCE_ALLREDUCE") == "1" and dist.is_initialized()
    if (world <= 1 and not forced) or not params:
        return
    key = tuple(id(p) for p in params)
    fg = _FLAT.get(key)
    if fg is None:
        _FLAT.clear()          # one model at a time: do not pin the buffers of models that are gone
        fg = _FLAT[key] = FlatGradients(params)
    fg.reduce(world, group, timings)


def broadcast_parameters(params, src: int = 0, group=None):
    """Make every rank start from rank `src`'s weights."""
    params = list(params)
    if not dist.is_initialized() or not params:
        return
    flat = torch.cat([p.data.reshape(-1) for p in params])
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        host = flat.cpu()
        dist.broadcast(host, src=src, group=group)
        flat.copy_(host)
    else:
        dist.broadcast(flat, src=src, group=group)
    off = 0
    for p in params:
        n = p.numel()
        p.data.copy_(flat[off:off + n].view_as(p))
        off += n
