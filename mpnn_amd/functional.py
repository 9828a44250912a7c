"""The reference's gather helpers (src/mpnn/models/model_utils.py:13-45) on the HIP library.

Same names, argument order and results as `gather_edges`, `gather_nodes`, `gather_nodes_t`, `cat_neighbors_nodes`;
every one is an indexed copy of whole feature rows (`mpnn_gather_indexed`, include/mpnn_hip.h).  Inside
`ProteinMPNN.forward` none of them runs - the gathers are fused into the edge kernels - they exist for callers that
use the free functions directly.  Inputs must be on a ROCm device (no CPU fallback); features are computed in fp32.
Differentiable in the gathered tensor (the gradient is a scatter-add over the same indices, an ATen `index_add_`).
"""
from __future__ import annotations

import torch

from ._lib import call, ptr, stream_ptr


def _check(src, idx, n_src, what):
    if not src.is_cuda or not idx.is_cuda:
        raise RuntimeError(f"mpnn_amd.{what}: tensors must be on a ROCm device; the HIP path has no CPU fallback")
    if idx.numel():
        lo, hi = torch.aminmax(idx)
        if int(lo) < 0 or int(hi) >= n_src:   # torch.gather raises for these too; off the hot path, so a sync is fine
            raise IndexError(f"mpnn_amd.{what}: neighbour index out of range [0, {n_src})")


def _launch(src, idx, G, n_src, n_idx, C, out, out_ld, col0):
    if G * n_idx == 0 or C == 0:
        return
    call("mpnn_gather_indexed", ptr(src), ptr(idx), G, n_src, n_idx, C, ptr(out), out_ld, col0, stream_ptr())


class _Gather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, idx, G, n_src, n_idx, head, head_cols):
        """src viewed as [G, n_src, C]; idx [G, n_idx] -> [G * n_idx, head_cols + C] with `head` in the leading columns."""
        C = src.shape[-1]
        pad = (-C) % 4
        s32 = src.detach().to(torch.float32)
        if pad:
            s32 = torch.nn.functional.pad(s32, (0, pad))
        s32 = s32.contiguous()
        Cp = C + pad
        i64 = idx.detach().to(torch.int64).contiguous()
        if head is None:
            out = torch.empty(G * n_idx, Cp, device=src.device, dtype=torch.float32)
            _launch(s32, i64, G, n_src, n_idx, Cp, out, Cp, 0)
            res = out[:, :C] if pad else out
        else:  # cat_neighbors_nodes: [h_neighbors | gathered]; both column blocks are multiples of 4 here
            h32 = head.detach().to(torch.float32).contiguous().view(G * n_idx, head_cols)
            out = torch.empty(G * n_idx, head_cols + Cp, device=src.device, dtype=torch.float32)
            _launch(h32, None, 1, G * n_idx, G * n_idx, head_cols, out, head_cols + Cp, 0)
            _launch(s32, i64, G, n_src, n_idx, Cp, out, head_cols + Cp, head_cols)
            res = out
        ctx.save_for_backward(i64)
        ctx.meta = (G, n_src, n_idx, C, head_cols if head is not None else 0, src.dtype)
        return res.to(src.dtype) if src.dtype != torch.float32 else res

    @staticmethod
    def backward(ctx, d_out):
        (i64,) = ctx.saved_tensors
        G, n_src, n_idx, C, hc, dtype = ctx.meta
        d_out = d_out.to(torch.float32)
        d_src = d_head = None
        if ctx.needs_input_grad[0]:
            flat = (i64 + torch.arange(G, device=i64.device).view(G, 1) * n_src).reshape(-1)
            d_src = torch.zeros(G * n_src, C, device=d_out.device, dtype=torch.float32)
            d_src.index_add_(0, flat, d_out[:, hc:hc + C].contiguous())
            d_src = d_src.to(dtype)
        if hc and ctx.needs_input_grad[5]:
            d_head = d_out[:, :hc].contiguous()
        return d_src, None, None, None, None, d_head, None


def gather_edges(edges, neighbor_idx):
    """Features [B,N,N,C] at neighbour indices [B,N,K] -> [B,N,K,C]   (reference :13-18)."""
    B, N, N2, C = edges.shape
    K = neighbor_idx.shape[-1]
    _check(edges, neighbor_idx, N2, "gather_edges")
    out = _Gather.apply(edges.reshape(B * N * N2, C), neighbor_idx.reshape(B * N, K), B * N, N2, K, None, 0)
    return out.reshape(B, N, K, C)


def gather_nodes(nodes, neighbor_idx):
    """Features [B,N,C] at neighbour indices [B,N,K] -> [B,N,K,C]   (reference :21-30)."""
    B, N, C = nodes.shape
    Bi, Ni, K = neighbor_idx.shape
    _check(nodes, neighbor_idx, N, "gather_nodes")
    out = _Gather.apply(nodes.reshape(B * N, C), neighbor_idx.reshape(B, Ni * K), B, N, Ni * K, None, 0)
    return out.reshape(Bi, Ni, K, C)


def gather_nodes_t(nodes, neighbor_idx):
    """Features [B,N,C] at neighbour index [B,K] -> [B,K,C]   (reference :33-38)."""
    B, N, C = nodes.shape
    K = neighbor_idx.shape[-1]
    _check(nodes, neighbor_idx, N, "gather_nodes_t")
    out = _Gather.apply(nodes.reshape(B * N, C), neighbor_idx.reshape(B, K), B, N, K, None, 0)
    return out.reshape(B, K, C)


def cat_neighbors_nodes(h_nodes, h_neighbors, E_idx):
    """[h_neighbors | h_nodes[E_idx]] along the feature axis -> [B,N,K,C_nb + C]   (reference :41-45)."""
    B, N, C = h_nodes.shape
    Bi, Ni, K = E_idx.shape
    Cn = h_neighbors.shape[-1]
    _check(h_nodes, E_idx, N, "cat_neighbors_nodes")
    if Cn % 4 or C % 4 or h_neighbors.dtype != h_nodes.dtype:
        return torch.cat([h_neighbors, gather_nodes(h_nodes, E_idx)], -1)
    out = _Gather.apply(h_nodes.reshape(B * N, C), E_idx.reshape(B, Ni * K), B, N, Ni * K,
                        h_neighbors.reshape(Bi * Ni * K, Cn), Cn)
    return out.reshape(Bi, Ni, K, Cn + C)
