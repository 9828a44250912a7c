"""HIP-graph replay of the inference forward for a fixed input shape.

`GraphedForward` captures one forward into a HIP graph (torch.cuda.CUDAGraph: the C-ABI launches go to the capturing
stream like any other kernel; the forward does no host synchronisation and allocates only through torch's graph
pool) and replays it with the inputs copied into static buffers - one host call per forward, for serving loops that
must not spend host time on ~30 launches.  Measured on MI355X (tools/bench_latency.py): the replay is bit-identical
and takes the same 0.70 ms as eager launches for one 100-500-residue protein - at that size the forward is bound by
the kernels' own latency (25-40 us each), not by the host, so the graph frees host time rather than shortening the
step.
"""
from __future__ import annotations

import torch


class GraphedForward:
    """model: mpnn_amd.ProteinMPNN in eval mode; example inputs fix B, L (and dtypes).  The decoding order must be
    passed explicitly (`fix_order`): a random order drawn inside the captured region would be replayed unchanged."""

    def __init__(self, model, X, S, mask, chain_M, residue_idx, chain_encoding_all, fix_order):
        if model.training:
            raise ValueError("GraphedForward captures the inference forward: call model.eval() first")
        self.model = model
        self.static = [t.clone() for t in (X, S, mask, chain_M, residue_idx, chain_encoding_all, fix_order)]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):  # warm-up outside the capture: weight packing, allocator pools
                model(*self.static[:6], fix_order=self.static[6])
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.out = model(*self.static[:6], fix_order=self.static[6])
        # the graph reads the packed weights of THIS moment; keep them alive and notice when the model moves on
        self._weights = model.__dict__.get("_packed")

    def __call__(self, X, S, mask, chain_M, residue_idx, chain_encoding_all, fix_order):
        from .engine import PackedWeights

        # stale when the packed set was replaced (eager forward after a change, refresh_weights, training step) OR when
        # any parameter changed since the capture without a re-pack yet (load_state_dict, in-place copy_, a non-fused
        # optimizer step): the key is (data_ptr, _version, device) per parameter, taken when the weights were packed
        if (self._weights is None or self.model.__dict__.get("_packed") is not self._weights
                or PackedWeights.make_key(self.model) != self._weights.key):
            raise RuntimeError("the model's weights changed or were re-packed since the capture (training step, "
                               "load_state_dict, in-place update, refresh_weights): capture a new GraphedForward")
        for dst, src in zip(self.static, (X, S, mask, chain_M, residue_idx, chain_encoding_all, fix_order)):
            if dst.shape != src.shape:
                raise ValueError(f"GraphedForward was captured for shape {tuple(dst.shape)}, got {tuple(src.shape)}")
            dst.copy_(src)
        self.graph.replay()
        return self.out
