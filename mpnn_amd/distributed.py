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


def allreduce_gradients(params, world: int, group=None):
    """Average `.grad` over ranks with a single flat all-reduce (parameters without a grad count as zeros)."""
    params = [p for p in params if p.requires_grad]
    # MPNN_FORCE_ALLREDUCE=1: a one-rank process group still takes the collective path (rehearsal of the RCCL path on a 1-GPU box)
    forced = os.environ.get("MPNN_FORCE_ALLREDUCE") == "1" and dist.is_initialized()
    if (world <= 1 and not forced) or not params:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    if flat.is_cuda and dist.get_backend(group) == "gloo":  # CPU rehearsal backend: stage through host memory
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(world)
    off, dst, src = 0, [], []
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            dst.append(p.grad)
            src.append(g)
        off += n
    if dst:
        torch._foreach_copy_(dst, src)  # one multi-tensor launch instead of one copy per parameter


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
