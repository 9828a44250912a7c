"""AdamW and gradient-norm clipping of the reference's training loop on the HIP library.

`AdamW` is `torch.optim.AdamW(lr, betas, eps, weight_decay)` as the reference constructs it (train.py:217; no amsgrad,
not maximize): same update, same `state_dict` layout (`step`, `exp_avg`, `exp_avg_sq` per parameter), so the
`optimizer_state_dict` of a reference checkpoint (train.py:548-568) loads.  One launch covers 48 parameter tensors
(`mpnn_adamw_step`); with `max_grad_norm` the clip of train.py:389-390 is folded into the same pass (the norm is
reduced on the device by `mpnn_grad_sumsq`, nothing synchronises with the host).  `clip_grad_norm_` is the stand-alone
form with torch's signature.  Parameters and gradients must be fp32 tensors on a ROCm device (no CPU fallback).
"""
from __future__ import annotations

import ctypes

import torch

from ._lib import call, ptr, stream_ptr


def _tables(tensors):
    n = len(tensors)
    return (ctypes.c_void_p * n)(*[t.data_ptr() for t in tensors])


def _check(tensors, what):
    for t in tensors:
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError(f"mpnn_amd.optim: {what} must be contiguous fp32 tensors on a ROCm device "
                               "(the HIP path has no CPU fallback)")


def grad_sumsq(grads):
    """Device scalar: sum of squares over all gradient elements (ceil(n / 48) + 1 launches; per-workgroup partials are
    added in a fixed tree, so the value - and the clip coefficient derived from it - is the same on every run)."""
    _check(grads, "gradients")
    dev = grads[0].device
    out = torch.empty(1, device=dev, dtype=torch.float32)
    sizes = [g.numel() for g in grads]
    numel = (ctypes.c_int * len(grads))(*sizes)
    need = sum(((max(sizes[k:k + 48]) + 2047) // 2048) * len(sizes[k:k + 48]) for k in range(0, len(sizes), 48))
    ws = torch.empty(need, device=dev, dtype=torch.float32)
    call("mpnn_grad_sumsq", len(grads), _tables(grads), numel, ptr(out), ptr(ws), need, stream_ptr())
    return out


def clip_grad_norm_(parameters, max_norm):
    """torch.nn.utils.clip_grad_norm_(parameters, max_norm) with the L2 norm: scales every `.grad` in place by
    min(1, max_norm / (total_norm + 1e-6)) and returns total_norm (a device tensor; no host synchronisation)."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    grads = [p.grad for p in parameters if p.grad is not None and p.grad.numel()]
    if not grads:
        return torch.zeros(())
    with torch.cuda.device(grads[0].device):
        ss = grad_sumsq(grads)
        numel = (ctypes.c_int * len(grads))(*[g.numel() for g in grads])
        call("mpnn_grad_scale", len(grads), _tables(grads), numel, ptr(ss), float(max_norm), stream_ptr())
    return ss.sqrt().reshape(())


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("mpnn_amd.optim.AdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False,
                                      maximize=False, foreach=None, capturable=False, differentiable=False,
                                      fused=None, decoupled_weight_decay=True))
        self.max_grad_norm = float(max_grad_norm)
        self.last_grad_sumsq = None  # device scalar of the last step (its square root is the unclipped gradient norm)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        live = [[p for p in g["params"] if p.grad is not None and p.numel()] for g in self.param_groups]
        every = [p for ps in live for p in ps]
        if not every:
            return loss
        _check(every, "parameters")
        _check([p.grad for p in every], "gradients")
        with torch.cuda.device(every[0].device):
            ss = None
            if self.max_grad_norm > 0.0:  # the clip norm spans every group, as clip_grad_norm_(model.parameters()) does
                ss = self.last_grad_sumsq = grad_sumsq([p.grad for p in every])
            for group, ps in zip(self.param_groups, live):
                if not ps:
                    continue
                steps = set()
                for p in ps:
                    st = self.state[p]
                    if not st:
                        st["step"] = torch.tensor(0.0)
                        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["step"] += 1
                    steps.add(int(st["step"]))
                # parameters that joined later have their own bias correction: one call per distinct step count
                for k in sorted(steps):
                    sel = [p for p in ps if int(self.state[p]["step"]) == k]
                    m = [self.state[p]["exp_avg"] for p in sel]
                    v = [self.state[p]["exp_avg_sq"] for p in sel]
                    _check(m + v, "optimizer state")
                    numel = (ctypes.c_int * len(sel))(*[p.numel() for p in sel])
                    lr = group["lr"]
                    call("mpnn_adamw_step", len(sel), _tables(sel), _tables([p.grad for p in sel]), _tables(m), _tables(v),
                         numel, float(lr), float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]),
                         float(group["weight_decay"]), k, ptr(ss), self.max_grad_norm, stream_ptr())
            # the kernels write the parameters behind autograd's back: bump the version counters, which key the
            # inference kernels' packed-weight cache (engine.PackedWeights) and guard saved tensors
            try:
                torch.autograd.graph.increment_version(every)
            except (AttributeError, TypeError):
                for p in every:
                    torch.autograd.graph.increment_version(p) if hasattr(torch.autograd.graph, "increment_version") else None
        return loss
