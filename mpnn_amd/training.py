"""Training step on the gfx950 kernels (reference: src/mpnn/train.py:63-73 loss, :130-175 step;
model_utils.py:427-529 forward differentiated by autograd).

What runs where (round 1):
  * per-edge work (85 % of the FLOPs): message MLPs of the three encoder and three decoder layers and the
    encoder edge updates -- forward AND backward -- on the hand-written HIP kernels
    (mpnn_message_train / mpnn_message_bwd / mpnn_enc_edge_update_train / mpnn_enc_edge_update_bwd /
    mpnn_edge_wgrad / mpnn_gather_rows), wrapped as torch.autograd.Function so the optimizer sees ordinary
    `.grad`s on the reference-shaped parameters;
  * neighbour search, backbone atoms, decoding-order ranks: the forward HIP kernels (no gradient flows there);
  * the edge featurizer (RBF / positional embedding -> edge_embedding -> norm_edges -> W_e): forward and backward on
    HIP kernels too (mpnn_edge_features_train / mpnn_edge_features_bwd / mpnn_feature_wgrad + mpnn_edge_wgrad);
  * per-node work (W3, LayerNorms, FFN, the next kernels' projections): mpnn_node_update_train / mpnn_node_update_bwd,
    weight gradients on mpnn_node_wgrad; only the occupancy MLP, the sequence-embedding term, the 21-letter head and
    the loss are ATen ops under autograd.
There is no CPU path: every tensor must live on the ROCm device.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F
from torch.amp import custom_bwd, custom_fwd  # backward runs under the forward's autocast state (off: modules.forward)

from . import engine
from . import _lib
from ._lib import call, ptr, stream_ptr

H = 128


# ---------------------------------------------------------------------------
# graph bookkeeping shared by all layers of one step
# ---------------------------------------------------------------------------
class Graph:
    """Neighbour lists of one batch plus the inverse list (edges grouped by the node they gather from)."""

    def __init__(self, E_idx64, E_idx32, mask, edge_mask, K):
        B, L, _ = E_idx64.shape
        dev = E_idx64.device
        self.B, self.L, self.K = B, L, K
        self.N = B * L
        self.KT = (K + 15) // 16
        self.Kp = 16 * self.KT
        self.E_idx64, self.E_idx32, self.mask, self.edge_mask = E_idx64, E_idx32, mask, edge_mask
        self.max_chunks = (self.N * K) // self.GATHER_CHUNK + self.N
        self.rank = None
        self.sel = None
        # inverse lists (edges grouped by the node they gather from, in the order of a stable sort by target)
        n_edges = self.N * K
        self.inv_ptr = torch.empty(self.N + 1, device=dev, dtype=torch.int32)
        self.chunk_ptr = torch.empty(self.N + 1, device=dev, dtype=torch.int32)
        self.chunk_node = torch.empty(self.max_chunks, device=dev, dtype=torch.int32)
        self.inv_row = torch.empty(n_edges, device=dev, dtype=torch.int32)
        if K <= L and self.N * L <= self.SLOT_MAP_LIMIT:
            trusted = E_idx32 is not None   # mpnn_knn_topk's own lists: in range, a residue's neighbours distinct
            if E_idx32 is None:   # stand-alone use (tests, caller-supplied lists): the padded int32 form
                E_idx32 = torch.zeros(self.N, self.Kp, device=dev, dtype=torch.int32)
                E_idx32[:, :K] = E_idx64.reshape(self.N, K)
            slot_map = torch.empty(self.N * L, device=dev, dtype=torch.uint8)
            status = None if trusted else torch.zeros(1, device=dev, dtype=torch.int32)
            call("mpnn_inverse_lists", ptr(E_idx32), B, L, K, ptr(slot_map), ptr(self.inv_ptr), ptr(self.chunk_ptr),
                 ptr(self.chunk_node), ptr(self.inv_row), ptr(status), stream_ptr())
            if status is not None and int(status) != 0:
                # a neighbour listed twice (or out of range): the slot map holds one edge per (target, centre) pair
                if int(E_idx64.min()) < 0 or int(E_idx64.max()) >= L:
                    raise ValueError(f"E_idx: neighbour ids must lie in [0, {L})")
                self._lists_by_sort(E_idx64)
        else:
            self._lists_by_sort(E_idx64)

    @property
    def dest(self):
        """node each edge gathers from, [N K] in edge order"""
        base = (torch.arange(self.B, device=self.E_idx64.device) * self.L).view(self.B, 1, 1)
        return (self.E_idx64 + base).reshape(-1)

    @property
    def src(self):
        """node each edge belongs to, [N K] in edge order"""
        return torch.arange(self.N, device=self.E_idx64.device).view(-1, 1).expand(-1, self.K).reshape(-1)

    SLOT_MAP_LIMIT = 1 << 30   # bytes of scratch mpnn_inverse_lists may take (B L^2); above: sort-based construction
    GATHER_CHUNK = 64          # list entries per work unit of mpnn_gather_rows (mpnn_edge_bwd.hip)

    def _lists_by_sort(self, E_idx64):
        """The same lists from ATen ops (a stable sort of the edges by target): for batches whose B L^2-byte slot map
        would be too large, and the check of mpnn_inverse_lists in tests/test_gpu_train.py."""
        B, L, K, dev = self.B, self.L, self.K, E_idx64.device
        dest = self.dest
        src = torch.arange(self.N, device=dev).view(-1, 1)
        rows = (src * self.Kp + torch.arange(K, device=dev).view(1, -1)).reshape(-1)
        perm = torch.argsort(dest, stable=True)
        self.inv_row = rows[perm].to(torch.int32).contiguous()
        # (torch.bincount synchronises with the host to size its output; scatter_add does not)
        counts = torch.zeros(self.N, device=dev, dtype=torch.int64).scatter_add_(0, dest, torch.ones_like(dest))
        self.inv_ptr = torch.cat([counts.new_zeros(1), torch.cumsum(counts, 0)]).to(torch.int32).contiguous()
        # mpnn_gather_rows works on chunks of <= 64 list entries (hub residues!), at least one per node
        chunks = torch.clamp((counts + self.GATHER_CHUNK - 1) // self.GATHER_CHUNK, min=1)
        ends = torch.cumsum(chunks, 0)
        self.chunk_ptr = torch.cat([chunks.new_zeros(1), ends]).to(torch.int32).contiguous()
        # (searchsorted instead of repeat_interleave: no host read of the total; entries past the last chunk are unused)
        self.chunk_node = torch.searchsorted(ends, torch.arange(self.max_chunks, device=dev), right=True) \
            .clamp_(max=self.N - 1).to(torch.int32).contiguous()

    def set_order(self, rank):
        """rank [N] int32 decoding rank; sel = 1 where the edge read the encoder block (neighbour not yet decoded)."""
        self.rank = rank
        self.sel = torch.empty(self.N * self.K, device=rank.device, dtype=torch.uint8)
        call("mpnn_inverse_sel", ptr(rank), ptr(self.inv_ptr), ptr(self.inv_row), self.N, self.K, ptr(self.sel),
             stream_ptr())

    def frag_numel(self):
        return self.N * 8 * self.KT * 256

    def full_count(self):
        """[N] tensor holding K: the decoder's neighbour count (reference :143: every slot counts), one per step."""
        if getattr(self, "_full_count", None) is None:
            self._full_count = torch.full((self.N,), float(self.K), device=self.E_idx64.device, dtype=torch.float32)
        return self._full_count


def _pack_flat(blocks):
    """blocks: list of (view [128,128] with unit inner stride, transposed?) -> their packed forms (f16 hi/lo A
    fragments, 16384 floats each) back to back in one tensor, by one launch (mpnn_pack_blocks).
    Packing right before the kernel that reads the blocks is deliberate: one up-front launch per step for all ~190
    blocks (38 launches -> 1) was built and measured 0.4-0.6 ms per step SLOWER at 20 x 512 (10.87 -> 11.3 ms on one
    box, three repetitions; 11.65 with a persistent buffer) although it removes GPU work and host time - a block packed
    microseconds before its reader is still in the caches when all 256 workgroups ask for it at once, one packed
    milliseconds earlier has been flushed by the 250 MB edge tensors in between."""
    import ctypes
    n = len(blocks)
    views = [b[0].detach() for b in blocks]
    for v in views:
        assert v.shape == (H, H) and v.stride(1) == 1, "pack blocks must be 128x128 with unit inner stride"
    out = torch.empty(n * 16384, device=views[0].device, dtype=torch.float32)
    ptrs = (ctypes.c_void_p * n)(*[v.data_ptr() for v in views])
    lds = (ctypes.c_int * n)(*[v.stride(0) for v in views])
    trs = (ctypes.c_int * n)(*[1 if b[1] else 0 for b in blocks])
    call("mpnn_pack_blocks", ptrs, lds, trs, n, ptr(out), stream_ptr())
    return out


def _pack_blocks(blocks):
    """_pack_flat as a list of [16384] tensors, one per block."""
    out = _pack_flat(blocks)
    return [out[i * 16384:(i + 1) * 16384] for i in range(len(blocks))]


# mpnn_message_bwd_fused (K <= 48) instead of mpnn_message_bwd + 2 x mpnn_edge_wgrad
FUSED_MSG_BWD = os.environ.get("MPNN_FUSED_MSG_BWD", "1") != "0"
# mpnn_enc_edge_update_bwd_fused (K <= 48) instead of mpnn_enc_edge_update_bwd + 2 of its 3 mpnn_edge_wgrad passes
FUSED_UPD_BWD = os.environ.get("MPNN_FUSED_UPD_BWD", "1") != "0"


def _wgrad(A, a_rows, amax, Bfrag, b_gelu, g, want_bias):
    dW = _zeros(A.device, H, H)
    db = _zeros(A.device, H) if want_bias else None
    ws = _workspace(A.device)
    call("mpnn_edge_wgrad", ptr(A), int(a_rows), ptr(amax), ptr(Bfrag), int(b_gelu), g.N, g.K, ptr(dW), ptr(db),
         ptr(ws), ws.numel(), stream_ptr())
    return dW, db


def _gather(rows, g, sel):
    outA = torch.empty(g.N, H, device=rows.device, dtype=torch.float32)   # every row is written by the kernel
    outB = torch.empty_like(outA) if sel is not None else None
    part = torch.empty(g.max_chunks * 2 * H, device=rows.device, dtype=torch.float32)
    call("mpnn_gather_rows", ptr(rows), ptr(g.inv_ptr), ptr(g.inv_row), ptr(sel), ptr(g.chunk_ptr), ptr(g.chunk_node),
         g.N, g.max_chunks, ptr(outA), ptr(outB), ptr(part), stream_ptr())
    return outA, outB


_WS = {}


class _ZeroArena:
    """One zero-filled block per backward pass, handed out in slices: the ~190 small gradient accumulators of a step
    (dW, dbias, LayerNorm sums, |max| scalars) would otherwise cost one fill launch each."""

    def __init__(self):
        self.buf = None
        self.amax = None   # one float on the device: max |buf| as its writers left it (the fused kernels' dE_absmax)
        self.used = 0

    def reset(self, dev, n_floats=4 << 20):
        self.buf = torch.zeros(n_floats, device=dev, dtype=torch.float32)
        self.used = 0

    def take(self, *shape):
        n = 1
        for d in shape:
            n *= d
        n_al = (n + 63) & ~63  # 256-byte aligned slices
        if self.buf is None or self.used + n_al > self.buf.numel():
            return torch.zeros(shape, device=self.buf.device if self.buf is not None else None, dtype=torch.float32)
        out = self.buf[self.used:self.used + n].view(shape)
        self.used += n_al
        return out


_ZEROS = _ZeroArena()


def _zeros(dev, *shape):
    if _ZEROS.buf is None or _ZEROS.buf.device != dev:
        return torch.zeros(shape, device=dev, dtype=torch.float32)
    return _ZEROS.take(*shape)


def _workspace(dev, n_rows=0):
    """Per-device scratch for the partial sums of the weight-gradient, backward, head and loss kernels (calls on one
    stream reuse it in order).  n_rows: rows (residues) of the call, for the entry points whose need grows with them."""
    lib = _lib.load()
    need = max(lib.mpnn_workspace_bytes(b"mpnn_edge_wgrad", 1 << 30, 1),
               lib.mpnn_workspace_bytes(b"mpnn_message_bwd_fused", 0, 0),
               lib.mpnn_workspace_bytes(b"mpnn_enc_edge_update_bwd_fused", 0, 0),
               lib.mpnn_workspace_bytes(b"mpnn_enc_edge_update_bwd", 0, 0),
               lib.mpnn_workspace_bytes(b"mpnn_logits_log_softmax_bwd", int(n_rows), 0),
               lib.mpnn_workspace_bytes(b"mpnn_node_update_bwd", int(n_rows), 0)) // 4
    ws = _WS.get(dev)
    if ws is None or ws.numel() < need:
        ws = _WS[dev] = torch.empty(need, device=dev, dtype=torch.float32)
    return ws


def _pow2_scale(t):
    """Exact power of two that brings max |t| to [2^12, 2^13) (1 for an all-zero tensor), as a device scalar: the
    library's GEMMs split their operands into f16 hi + lo halves, which resolve only 3e-8 ABSOLUTE below 6e-5 -
    gradient-sized operands (loss / 2000) are scaled up first and the product scaled back (mpnn_bwd_common.h does the
    same inside the backward kernels).  No host synchronisation."""
    amax = t.detach().abs().max()
    e = torch.floor(torch.log2(torch.clamp(amax, min=1e-38)))
    return torch.where(amax > 0, torch.exp2(12.0 - e), torch.ones_like(amax))


def _linear_rows(x, Wp, bias):
    """y [rows, 128] = x [rows, 128] Wp^T (+ bias) on mpnn_linear_rows (Wp: one split-packed 128 x 128 block)."""
    x = x.contiguous()
    y = torch.empty(x.shape[0], H, device=x.device, dtype=torch.float32)
    if x.shape[0]:
        call("mpnn_linear_rows", ptr(x), H, H, ptr(Wp), ptr(bias), x.shape[0], ptr(y), stream_ptr())
    return y


class LinearFn(torch.autograd.Function):
    """nn.Linear(128 -> 128) on [rows, 128] entirely on the library: forward and the input gradient on
    mpnn_linear_rows (split-f16 MFMA, fp32-class), weight / bias gradients on mpnn_node_wgrad.  (Round 2 used
    F.linear / `dy @ W` here: hipBLASLt launches on the training path.)"""

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, x, W, b):
        assert W.shape == (H, H) and x.shape[-1] == H, "LinearFn serves the path's 128 x 128 blocks"
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        wp, = _pack_blocks([(W, False)])
        return _linear_rows(x.detach().float(), wp, b.detach().float().contiguous() if b is not None else None)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = dy.contiguous()
        x = x.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            wt, = _pack_blocks([(W, True)])
            sc = _pow2_scale(dy)
            dx = _linear_rows(dy * sc, wt, None) / sc
        dW = _zeros(dy.device, *W.shape)
        db = _zeros(dy.device, W.shape[0]) if ctx.has_bias else None
        ws = _workspace(dy.device)
        call("mpnn_node_wgrad", ptr(dy), dy.shape[1], dy.shape[1], ptr(x), x.shape[1], x.shape[1], dy.shape[0], ptr(dW),
             dW.shape[1], ptr(db), ptr(ws), ws.numel(), stream_ptr())
        return dx, dW, db


def linear(x, W, b=None):
    return LinearFn.apply(x, W, b)


class TableRowsFn(torch.autograd.Function):
    """rows[n] = table[idx[n]] (the decoder's per-letter sequence term, reference :478-479 h_S = W_s(S) followed by its
    W1 column block).  Backward: d table = one_hot(idx)^T d rows as a node-level weight gradient on mpnn_node_wgrad -
    a fixed summation order, where ATen's index_add / embedding backward meet rows by atomics."""

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, table, idx, onehot):
        ctx.save_for_backward(onehot)
        ctx.n_rows = table.shape[0]
        return table.index_select(0, idx)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, d_rows):
        (onehot,) = ctx.saved_tensors
        d_rows = d_rows.contiguous()
        dev = d_rows.device
        dWt = _zeros(dev, H, onehot.shape[1])      # [feature][letter]
        ws = _workspace(dev)
        call("mpnn_node_wgrad", ptr(d_rows), H, H, ptr(onehot), onehot.shape[1], ctx.n_rows, d_rows.shape[0], ptr(dWt),
             onehot.shape[1], None, ptr(ws), ws.numel(), stream_ptr())
        return dWt[:, :ctx.n_rows].t().contiguous(), None, None


class SeqTermFn(torch.autograd.Function):
    """A[n] = A0[n] + (W_s W1c^T)[S[n]]: the decoder's sequence term (reference :478-483, h_S = W_s(S) entering the
    third column block of W1) added to the rows the decoded neighbours gather.  Forward: the n_letters-row table
    (mpnn_seq_table) and one add-gather (mpnn_add_table_rows).  Backward: the table's gradient as a node-level weight
    gradient against one-hot rows of S (mpnn_node_wgrad: a fixed summation order), then dW_s and dW1c from it in one
    small launch (mpnn_seq_table_bwd).  All plain fp32: gradient-sized operands need no scaling here.  (Round 2 built
    this from LinearFn + TableRowsFn: ~20 small launches per layer and backward pass.)"""

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, A0, Ws, W1c, S, onehot):
        assert W1c.shape == (H, H) and W1c.stride(1) == 1 and Ws.shape[1] == H
        A0 = A0.contiguous()
        Wsc = Ws.detach().float().contiguous()
        n_letters = Wsc.shape[0]
        TS = torch.empty(n_letters, H, device=A0.device, dtype=torch.float32)
        st = stream_ptr()
        call("mpnn_seq_table", ptr(Wsc), W1c.data_ptr(), W1c.stride(0), n_letters, ptr(TS), st)
        A = torch.empty_like(A0)
        call("mpnn_add_table_rows", ptr(A0), ptr(TS), ptr(S), n_letters, A0.shape[0], ptr(A), st)
        ctx.save_for_backward(Wsc, W1c, onehot)
        return A

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, dA):
        Wsc, W1c, onehot = ctx.saved_tensors
        dA = dA.contiguous()
        dev = dA.device
        n_letters = Wsc.shape[0]
        dTSt = _zeros(dev, H, onehot.shape[1])      # [feature][letter]
        ws = _workspace(dev)
        call("mpnn_node_wgrad", ptr(dA), H, H, ptr(onehot), onehot.shape[1], n_letters, dA.shape[0], ptr(dTSt),
             onehot.shape[1], None, ptr(ws), ws.numel(), stream_ptr())
        dWs = torch.empty(n_letters, H, device=dev, dtype=torch.float32)
        dW1c = torch.empty(H, H, device=dev, dtype=torch.float32)
        call("mpnn_seq_table_bwd", ptr(dTSt), onehot.shape[1], ptr(Wsc), W1c.data_ptr(), W1c.stride(0), n_letters,
             ptr(dWs), ptr(dW1c), stream_ptr())
        return dA, dWs, dW1c, None, None


class GradSlot:
    """Shared gradient buffer of one edge tensor hE_l.  Its consumers' backward kernels accumulate in place; the first
    one hands the buffer to autograd, the others return None (autograd runs the producer only after all of them).
    The producer's backward (EdgeUpdateFn / FeaturesFn, `out_slot`) clears the slot again, so a second backward over
    the same graph (retain_graph=True, two losses on one forward) starts from a fresh buffer."""

    def __init__(self):
        self.buf = None
        self.amax = None   # one float on the device: max |buf| as its writers left it (the fused kernels' dE_absmax)


# ---------------------------------------------------------------------------
# autograd wrappers of the edge kernels
# ---------------------------------------------------------------------------
class MessageFn(torch.autograd.Function):
    """gsum_i = sum_k w_ik gelu(W2 gelu(P_i + s_ik (Row_ik + W1b e_ik)) + b2)   (mpnn_edge.hip k_message)."""

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, g, slot, dec, hE, P, RowA, RowB, W1b, W2, b2):
        # dec: False (encoder), True (decoder), "valid" (decoder whose neighbour sum runs over edge_mask = 1 only: the
        # edge-list model, mpnn_amd.sparse)
        w_valid = 1 if dec == "valid" else 0
        dec = bool(dec)
        dev = hE.device
        P, RowA = P.contiguous(), RowA.contiguous()
        RowB = RowB.contiguous() if RowB is not None else None
        gsum = torch.empty(g.N, H, device=dev, dtype=torch.float32)
        cnt = torch.empty(g.N, device=dev, dtype=torch.float32) if not dec else None
        a1 = torch.empty(g.frag_numel(), device=dev, dtype=torch.float32)
        a2 = torch.empty_like(a1)
        b2c = b2.detach().contiguous()
        w1p, w2p = _pack_blocks([(W1b, False), (W2, False)])  # named: a temporary's block could be reused before the launch
        call("mpnn_message_train", int(dec), ptr(hE), ptr(P), ptr(RowA), ptr(RowB), ptr(g.rank) if dec else None,
             ptr(g.E_idx32), ptr(g.mask), ptr(g.edge_mask), ptr(w1p), ptr(w2p), ptr(b2c),
             g.B, g.L, g.K, ptr(gsum), ptr(cnt), ptr(a1), ptr(a2), w_valid, stream_ptr())
        ctx.g, ctx.dec, ctx.slot, ctx.w_valid = g, dec, slot, w_valid
        ctx.save_for_backward(hE, a1, a2, W1b, W2)
        if dec:
            cnt = g.full_count()
        ctx.mark_non_differentiable(cnt)
        ctx.set_materialize_grads(False)  # (no zero tensor for cnt's "gradient": a fill launch per layer)
        return gsum, cnt

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, dgsum, _dcnt):
        g, dec = ctx.g, ctx.dec
        hE, a1, a2, W1b, W2 = ctx.saved_tensors
        dev = hE.device
        dgsum = dgsum.contiguous() if dgsum is not None else torch.zeros(g.N, H, device=dev, dtype=torch.float32)
        rows = torch.empty(g.N * g.Kp, H, device=dev, dtype=torch.float32)
        dP = torch.empty(g.N, H, device=dev, dtype=torch.float32)
        slot = ctx.slot
        first = slot.buf is None
        if first:
            slot.buf = torch.empty_like(a1)
            slot.amax = None
        dE = slot.buf
        w2t, w1t = _pack_blocks([(W2, True), (W1b, True)])
        if FUSED_MSG_BWD and g.K <= 48:
            # one kernel: chain + both weight-gradient contractions (da2 never exists in memory)
            dW2, db2, dW1b = _zeros(dev, H, H), _zeros(dev, H), _zeros(dev, H, H)
            ws = _workspace(dev)
            if first:
                slot.amax = _zeros(dev, 1)
            call("mpnn_message_bwd_fused", int(dec), ptr(dgsum), ptr(a1), ptr(a2), ptr(hE), ptr(g.E_idx32), ptr(g.mask),
                 ptr(g.edge_mask), ptr(w2t), ptr(w1t), g.B, g.L, g.K, ptr(rows), ptr(dP), ptr(dE), 0 if first else 1,
                 ctx.w_valid, ptr(dW2), ptr(db2), ptr(dW1b), ptr(slot.amax), ptr(ws), ws.numel(), stream_ptr())
            dRowA, dRowB = _gather(rows, g, g.sel if dec else None)
            return None, None, None, dE if first else None, dP, dRowA, dRowB, dW1b, dW2, db2
        da2 = torch.empty_like(a1)
        amax = _zeros(dev, 2)
        slot.amax = None  # this writer records no max |dE|
        call("mpnn_message_bwd", int(dec), ptr(dgsum), ptr(a1), ptr(a2), ptr(g.E_idx32), ptr(g.mask), ptr(g.edge_mask),
             ptr(w2t), ptr(w1t), g.B, g.L, g.K, ptr(da2), ptr(rows),
             ptr(dP), ptr(dE), 0 if first else 1, ptr(amax), ctx.w_valid, stream_ptr())
        dRowA, dRowB = _gather(rows, g, g.sel if dec else None)
        dW2, db2 = _wgrad(da2, False, amax[0:1], a1, True, g, True)
        dW1b, _ = _wgrad(rows, True, amax[1:2], hE, False, g, False)
        return None, None, None, dE if first else None, dP, dRowA, dRowB, dW1b, dW2, db2


class EdgeUpdateFn(torch.autograd.Function):
    """e' = LN3(e + W13 gelu(W12 gelu(P'_i + Q'_j + W11b e) + b12) + b13)   (mpnn_edge.hip k_enc_edge_update)."""

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, g, slot, out_slot, p_drop, seed, hE, P, Q, W11b, W12, b12, W13, b13, lnw, lnb):
        dev = hE.device
        P, Q = P.contiguous(), Q.contiguous()
        out = torch.empty_like(hE)
        a1, a2, z = torch.empty_like(hE), torch.empty_like(hE), torch.empty_like(hE)
        vec = [t.detach().contiguous() for t in (b12, b13, lnw, lnb)]
        w1p, w2p, w3p = _pack_blocks([(W11b, False), (W12, False), (W13, False)])
        rstd_max = _zeros(dev, 1)
        call("mpnn_enc_edge_update_train", ptr(hE), ptr(P), ptr(Q), ptr(g.E_idx32), ptr(w1p),
             ptr(w2p), ptr(vec[0]), ptr(w3p), ptr(vec[1]), ptr(vec[2]), ptr(vec[3]), g.B, g.L,
             g.K, ptr(out), ptr(a1), ptr(a2), ptr(z), float(p_drop), int(seed), ptr(rstd_max), stream_ptr())
        ctx.g, ctx.slot, ctx.out_slot, ctx.drop = g, slot, out_slot, (float(p_drop), int(seed))
        ctx.rstd_max = rstd_max
        ctx.save_for_backward(hE, a1, a2, z, W11b, W12, W13, lnw)
        return out

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        g = ctx.g
        hE, a1, a2, z, W11b, W12, W13, lnw = ctx.saved_tensors
        dev = hE.device
        dy = dy.contiguous()
        out_slot = ctx.out_slot
        # max |dy|: what the fused kernels that wrote it recorded, when dy IS their buffer; else two small ATen launches
        dy_amax = out_slot.amax if (out_slot.buf is not None and out_slot.amax is not None
                                    and out_slot.buf.data_ptr() == dy.data_ptr()) else None
        out_slot.buf, out_slot.amax = None, None  # every consumer of this kernel's output has run: its buffer is spent
        assert ctx.slot.buf is None, "the edge update is the first consumer of its input to run backward"
        dE = torch.empty_like(hE)
        ctx.slot.buf = dE
        rows = torch.empty(g.N * g.Kp, H, device=dev, dtype=torch.float32)
        dP = torch.empty(g.N, H, device=dev, dtype=torch.float32)
        dln = _zeros(dev, 2 * H)
        w3t, w2t, w1t = _pack_blocks([(W13, True), (W12, True), (W11b, True)])
        ws = _workspace(dev)
        if FUSED_UPD_BWD and g.K <= 48:
            # one kernel: LayerNorm', the chain and the W13 / W12 weight-gradient contractions (dz, da2 never exist in
            # memory); the W11b contraction reads the rows it wrote
            if dy_amax is None:
                dy_amax = dy.abs().max().reshape(1)
            dW13, db13, dW12, db12 = _zeros(dev, H, H), _zeros(dev, H), _zeros(dev, H, H), _zeros(dev, H)
            amax = _zeros(dev, 2)
            ctx.slot.amax = amax[1:2]
            call("mpnn_enc_edge_update_bwd_fused", ptr(dy), ptr(z), ptr(a2), ptr(a1), ptr(lnw.detach().contiguous()),
                 ptr(w3t), ptr(w2t), ptr(w1t), g.B, g.L, g.K, ptr(rows), ptr(amax[0:1]), ptr(dP), ptr(dE), ptr(dln),
                 ptr(dW13), ptr(db13), ptr(dW12), ptr(db12), ptr(dy_amax), ptr(ctx.rstd_max), ptr(amax[1:2]),
                 ctx.drop[0], ctx.drop[1], ptr(ws), ws.numel(), stream_ptr())
            dQ, _ = _gather(rows, g, None)
            dW11b, _ = _wgrad(rows, True, amax[0:1], hE, False, g, False)
            return None, None, None, None, None, dE, dP, dQ, dW11b, dW12, db12, dW13, db13, dln[:H], dln[H:]
        dz, da2 = torch.empty_like(hE), torch.empty_like(hE)
        amax = _zeros(dev, 3)
        ctx.slot.amax = None
        call("mpnn_enc_edge_update_bwd", ptr(dy), ptr(z), ptr(a2), ptr(a1), ptr(lnw.detach().contiguous()),
             ptr(w3t), ptr(w2t), ptr(w1t),
             g.B, g.L, g.K, ptr(dz), ptr(da2), ptr(rows), ptr(dP), ptr(dE), ptr(dln), ptr(amax), ctx.drop[0], ctx.drop[1],
             ptr(ws), ws.numel(), stream_ptr())
        dQ, _ = _gather(rows, g, None)
        dW13, db13 = _wgrad(dz, False, amax[0:1], a2, True, g, True)
        dW12, db12 = _wgrad(da2, False, amax[1:2], a1, True, g, True)
        dW11b, _ = _wgrad(rows, True, amax[2:3], hE, False, g, False)
        return None, None, None, None, None, dE, dP, dQ, dW11b, dW12, db12, dW13, db13, dln[:H], dln[H:]


class FromNaturalFn(torch.autograd.Function):
    """[N, K, 128] row-major edges -> fragment layout (and back for the gradient)."""

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, g, h_nat):
        ctx.g = g
        return engine.edges_from_natural(h_nat.contiguous(), g.N, g.K)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, d_frag):
        g = ctx.g
        return None, engine.edges_to_natural(d_frag.contiguous(), g.N, g.K)


# ---------------------------------------------------------------------------
# featurizer: RBF + positional embedding -> edge_embedding -> norm_edges -> W_e   (reference :334-358, :464)
# ---------------------------------------------------------------------------
class FeaturesFn(torch.autograd.Function):
    """hE0 (fragment layout) from backbone atoms; parameters: embeddings.linear, edge_embedding, norm_edges, W_e."""

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, g, out_slot, atoms, residue_idx, chain_labels, n_atoms, Wpos, bpos, Wemb, lnw, lnb, We, be,
                bemb=None):
        dev = atoms.device
        st = stream_ptr()
        table = torch.empty(66 * 16, device=dev, dtype=torch.float32)
        Wl, bl = Wpos.detach().contiguous(), bpos.detach().contiguous()
        call("mpnn_pos_table", ptr(Wl), ptr(bl), 66, 16, ptr(table), st)
        mu = torch.linspace(2.0, 22.0, 16, device=dev, dtype=torch.float32)
        Wc = Wemb.detach().contiguous()
        nkk = (Wc.shape[1] + 31) // 32
        Wemb_p = torch.empty(8 * nkk * 512, device=dev, dtype=torch.float32)
        call("mpnn_pack_linear_split", ptr(Wc), H, Wc.shape[1], Wc.shape[1], 0, 16, ptr(Wemb_p), st)
        We_p, = _pack_blocks([(We, False)])
        vec = [t.detach().contiguous() for t in (lnw, lnb, be)]
        hE = torch.empty(g.frag_numel(), device=dev, dtype=torch.float32)
        E0, E = torch.empty_like(hE), torch.empty_like(hE)
        bemb_c = bemb.detach().float().contiguous() if bemb is not None else None
        call("mpnn_edge_features_train", ptr(atoms), ptr(g.E_idx32), ptr(residue_idx), ptr(chain_labels), ptr(table),
             ptr(mu), ptr(Wemb_p), ptr(bemb_c), ptr(vec[0]), ptr(vec[1]), ptr(We_p), ptr(vec[2]), g.B, g.L, g.K, n_atoms,
             ptr(E0), ptr(E), ptr(hE), st)
        ctx.g, ctx.n_atoms, ctx.out_slot, ctx.has_bemb = g, n_atoms, out_slot, bemb is not None
        ctx.save_for_backward(atoms, residue_idx, chain_labels, table, mu, E0, E, Wemb, lnw, We)
        return hE

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, dhE):
        g, na = ctx.g, ctx.n_atoms
        ctx.out_slot.buf = None  # every consumer of hE0 has run
        atoms, residue_idx, chain_labels, table, mu, E0, E, Wemb, lnw, We = ctx.saved_tensors
        dev = dhE.device
        st = stream_ptr()
        dhE = dhE.contiguous()
        dE0 = torch.empty_like(E0)
        dln = _zeros(dev, 2 * H)
        amax = _zeros(dev, 2)
        wet, = _pack_blocks([(We, True)])
        ws = _workspace(dev)
        call("mpnn_edge_features_bwd", ptr(dhE), ptr(E0), ptr(lnw.detach().contiguous()), ptr(wet), g.B, g.L, g.K,
             ptr(dE0), ptr(dln), ptr(amax), ptr(ws), ws.numel(), st)
        dWe, dbe = _wgrad(dhE, False, amax[0:1], E, False, g, True)
        nrbf = na * na * 16
        nch = (na * na + 5 + 7) // 8
        # d / d(raw feature) [128, nch*128]: dE0 contracted with the raw features (RBFs, position one-hot), which the
        # kernel recomputes per residue instead of reading 1 GB of materialised chunks (mpnn_feature_chunks, round 2)
        dFc = _zeros(dev, nch, H, H)
        dbemb = _zeros(dev, H) if ctx.has_bemb else None                    # column sums of dE0 (None without the bias)
        call("mpnn_feature_wgrad", ptr(dE0), ptr(amax[1:2]), ptr(atoms), ptr(g.E_idx32), ptr(residue_idx),
             ptr(chain_labels), ptr(mu), g.B, g.L, g.K, na, ptr(dFc), ptr(dbemb), ptr(ws), ws.numel(), st)
        dF = dFc.permute(1, 0, 2).reshape(H, nch * H)
        G = dF[:, nrbf:nrbf + 66]                                           # sum of dE0 rows per position bucket
        W = Wemb.detach()
        npos = W.shape[1] - nrbf
        # two 128 x 66 x 16 contractions as broadcast products + ATen tree sums (a library GEMM launch each otherwise)
        dT = (G.unsqueeze(2) * W[:, :npos].unsqueeze(1)).sum(0)             # gradient of the 66 x 16 embedding table
        dWemb = torch.cat([(G.unsqueeze(2) * table.view(1, 66, 16)).sum(1), dF[:, :nrbf]], dim=1)
        return (None, None, None, None, None, None, dT.t().contiguous(), dT.sum(0), dWemb, dln[:H], dln[H:], dWe, dbe,
                dbemb)


def _node_wgrad(dy, x, want_bias):
    """dW [out, in] = dy^T x, db = column sums of dy (rows may be strided views with unit inner stride)."""
    dW = _zeros(dy.device, dy.shape[1], x.shape[1])
    db = _zeros(dy.device, dy.shape[1]) if want_bias else None
    ws = _workspace(dy.device)
    call("mpnn_node_wgrad", ptr(dy), dy.shape[1], dy.shape[1], ptr(x), x.shape[1], x.shape[1], dy.shape[0], ptr(dW),
         dW.shape[1], ptr(db), ptr(ws), ws.numel(), stream_ptr())
    return dW, db


def _node_wgrad_batch(jobs):
    """jobs: list of (dy [N,out], x [N,in], want_bias) sharing N -> list of (dW, db | None); one launch."""
    import ctypes
    n = len(jobs)
    dev = jobs[0][0].device
    outs = [(_zeros(dev, dy.shape[1], x.shape[1]), _zeros(dev, dy.shape[1]) if wb else None) for dy, x, wb in jobs]
    vp, ip = ctypes.c_void_p * n, ctypes.c_int * n
    ws = _workspace(dev)
    call("mpnn_node_wgrad_batch", n, vp(*[ptr(j[0]) for j in jobs]), ip(*[j[0].shape[1] for j in jobs]),
         ip(*[j[0].shape[1] for j in jobs]), vp(*[ptr(j[1]) for j in jobs]), ip(*[j[1].shape[1] for j in jobs]),
         ip(*[j[1].shape[1] for j in jobs]), vp(*[ptr(o[0]) for o in outs]), ip(*[o[0].shape[1] for o in outs]),
         vp(*[ptr(o[1]) for o in outs]), jobs[0][0].shape[0], ptr(ws), ws.numel(), stream_ptr())
    return outs


class NodeUpdateFn(torch.autograd.Function):
    """h' = mask LN2(x1 + drop(FFN(x1))), x1 = LN1(h + drop((W3 gsum + cnt b3) / scale)), and the next edge kernels'
    projections Wproj_p h' + b_p   (reference :81-88, :127-137, :151-155; mpnn_node_bwd.hip)."""

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, meta, hV_in, gsum, cnt, mask, W3, b3, ln1w, ln1b, Win, bin_, Wout, bout, ln2w, ln2b, *projs):
        dev = hV_in.device
        N = hV_in.shape[0]
        nproj = len(projs) // 2
        scale, p_drop, seed = meta
        hV_in, gsum = hV_in.contiguous(), gsum.contiguous()
        base = _pack_flat([(W3, False)] + [(Win[H * q:H * (q + 1)], False) for q in range(4)] +
                          [(Wout[:, H * q:H * (q + 1)], False) for q in range(4)] +
                          [(projs[2 * p], False) for p in range(nproj)])
        w3p, winp, woutp = base[:16384], base[16384:5 * 16384], base[5 * 16384:9 * 16384]
        vec = [v.detach().contiguous() for v in (b3, ln1w, ln1b, bin_, bout, ln2w, ln2b)]
        wprojp = bproj = PQ = None
        if nproj:
            wprojp = base[9 * 16384:]
            zero_h = _zeros(dev, H)
            bproj = torch.cat([projs[2 * p + 1].detach().float() if projs[2 * p + 1] is not None else zero_h
                               for p in range(nproj)])  # one launch (a fill and a copy per bias otherwise)
            PQ = torch.empty(nproj, N, H, device=dev, dtype=torch.float32)
        hV_out = torch.empty(N, H, device=dev, dtype=torch.float32)
        x1pre, x2pre = torch.empty_like(hV_out), torch.empty_like(hV_out)
        a_pre = torch.empty(N, 4 * H, device=dev, dtype=torch.float32)
        call("mpnn_node_update_train", ptr(hV_in), ptr(gsum), ptr(cnt), 0.0, ptr(mask), ptr(w3p), ptr(vec[0]), ptr(vec[1]),
             ptr(vec[2]), ptr(winp), ptr(vec[3]), ptr(woutp), ptr(vec[4]), ptr(vec[5]), ptr(vec[6]), float(scale),
             ptr(wprojp), ptr(bproj), nproj, N, ptr(hV_out), ptr(PQ), ptr(x1pre), ptr(a_pre), ptr(x2pre), float(p_drop),
             int(seed), stream_ptr())
        ctx.meta, ctx.nproj = meta, nproj
        ctx.has_bias = [projs[2 * p + 1] is not None for p in range(nproj)]
        ctx.save_for_backward(gsum, cnt, mask, W3, ln1w, ln1b, Win, Wout, ln2w, hV_out, x1pre, a_pre, x2pre,
                              *[projs[2 * p] for p in range(nproj)])
        ctx.set_materialize_grads(False)  # an output nobody differentiated through arrives as None, not as a zero tensor
        return (hV_out,) + tuple(PQ[p] for p in range(nproj))

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, dY, *dPQ):
        scale, p_drop, seed = ctx.meta
        nproj = ctx.nproj
        gsum, cnt, mask, W3, ln1w, ln1b, Win, Wout, ln2w, hV_out, x1pre, a_pre, x2pre = ctx.saved_tensors[:13]
        Wp = ctx.saved_tensors[13:]
        dev = gsum.device
        N = gsum.shape[0]
        dY = dY.contiguous() if dY is not None else None
        wprojT = None
        # the projections' gradients stay where their producers left them (stacked: a copy of nproj x N x 128 floats)
        dPQ = [d.contiguous() if d is not None else None for d in dPQ]
        import ctypes
        dPQ_tab = (ctypes.c_void_p * max(nproj, 1))(*[ptr(d) for d in dPQ]) if nproj else None
        base = _pack_flat([(W, True) for W in Wp] + [(Wout[:, H * q:H * (q + 1)], True) for q in range(4)] +
                          [(Win[H * q:H * (q + 1)], True) for q in range(4)] + [(W3, True)])
        o = nproj * 16384
        wprojT = base[:o] if nproj else None
        woutT, winT, w3T = base[o:o + 4 * 16384], base[o + 4 * 16384:o + 8 * 16384], base[o + 8 * 16384:o + 9 * 16384]
        d_hV_in, d_gsum = torch.empty(N, H, device=dev), torch.empty(N, H, device=dev)
        x1, d_o, d_dh = torch.empty(N, H, device=dev), torch.empty(N, H, device=dev), torch.empty(N, H, device=dev)
        hid, d_a = torch.empty(N, 4 * H, device=dev), torch.empty(N, 4 * H, device=dev)
        dvec = _zeros(dev, 5 * H)
        ws = _workspace(dev, N)
        call("mpnn_node_update_bwd_each", ptr(dY), dPQ_tab, ptr(x1pre), ptr(a_pre), ptr(x2pre), ptr(cnt), 0.0, ptr(mask),
             ptr(ln1w.detach().contiguous()), ptr(ln1b.detach().contiguous()), ptr(ln2w.detach().contiguous()),
             float(scale), ptr(wprojT), nproj, ptr(woutT), ptr(winT), ptr(w3T), N, ptr(d_hV_in), ptr(d_gsum), ptr(x1),
             ptr(hid), ptr(d_a), ptr(d_o), ptr(d_dh), ptr(dvec), float(p_drop), int(seed), ptr(ws), ws.numel(),
             stream_ptr())
        live = [p for p in range(nproj) if dPQ[p] is not None]
        res = _node_wgrad_batch([(d_dh, gsum, False), (d_a, x1, True), (d_o, hid, True)] +
                                [(dPQ[p], hV_out, ctx.has_bias[p]) for p in live])
        (dW3, _), (dWin, dbin), (dWout, dbout) = res[:3]
        grads = [None, d_hV_in, d_gsum, None, None, dW3, dvec[4 * H:], dvec[:H], dvec[H:2 * H], dWin, dbin, dWout, dbout,
                 dvec[2 * H:3 * H], dvec[3 * H:4 * H]]
        by_proj = dict(zip(live, res[3:]))
        for p in range(nproj):  # (a projection nobody differentiated through gets no gradient: autograd reads None as 0)
            grads += list(by_proj.get(p, (None, None)))
        return tuple(grads)


def _node_update(lyr, hV, gsum, cnt, mask, p_drop, projs):
    """projs: list of (weight view [128,128], bias | None) -> (h', [projection tensors])."""
    seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p_drop > 0 else 0  # host generator: no device sync
    flat = []
    for W, b in projs:
        flat += [W, b]
    out = NodeUpdateFn.apply((float(lyr.scale), float(p_drop), seed), hV, gsum, cnt, mask, lyr.W3.weight, lyr.W3.bias,
                             lyr.norm1.weight, lyr.norm1.bias, lyr.dense.W_in.weight, lyr.dense.W_in.bias,
                             lyr.dense.W_out.weight, lyr.dense.W_out.bias, lyr.norm2.weight, lyr.norm2.bias, *flat)
    return out[0], list(out[1:])


def forward_train(model, X, S, mask, chain_M, residue_idx, chain_encoding_all, fix_order=None,
                  fix_backbone_noise=None):
    """ProteinMPNN.forward with autograd (reference :427-529) -> log_probs [B,L,21]."""
    engine._require_device(X, "X")
    dev = X.device
    mask = engine._f32(mask)
    B, L = mask.shape
    N = B * L
    S = engine._i64(S)
    residue_idx, chain_labels = engine._i64(residue_idx), engine._i64(chain_encoding_all)
    _ZEROS.reset(dev)  # gradient accumulators of this step's backward pass
    fm = model.features
    training = model.training
    p_drop = float(model.dropout_p) if training else 0.0
    engine.invalidate_packed(model)  # an optimizer step follows: the inference kernels' packed weights go stale
    with torch.no_grad():
        atoms, _, E_idx64, E_idx32, _, edge_mask, K = engine._geometry(fm, X, mask, fix_backbone_noise, training)
        occ = engine._occupancy(fm, atoms, mask, B, L)
        g = Graph(E_idx64, E_idx32, mask.reshape(-1).contiguous(), edge_mask, K)
    if K > 64:
        raise NotImplementedError("mpnn_amd covers k_neighbors <= 64")
    slot = GradSlot()
    hE = FeaturesFn.apply(g, slot, atoms, residue_idx, chain_labels, 6 if fm.use_virtual_center else 5,
                          fm.embeddings.linear.weight, fm.embeddings.linear.bias, fm.edge_embedding.weight,
                          fm.norm_edges.weight, fm.norm_edges.bias, model.W_e.weight, model.W_e.bias)
    hV = torch.zeros(N, H, device=dev, dtype=torch.float32)
    has_occ = model.occupancy_mlp is not None and occ is not None
    if has_occ:
        hV = hV + model.occupancy_mlp(occ.view(N, 1))
    enc, dec = list(model.encoder_layers), list(model.decoder_layers)
    # every [128, 128 n] weight is cut into its column blocks ONCE (torch.split: one autograd node whose backward is a
    # single cat of the blocks' gradients; slicing block by block costs a zero fill, a copy and an add per use)
    W1s = {id(lyr): lyr.W1.weight.split(H, dim=1) for lyr in enc + dec}
    W11s = {id(lyr): lyr.W11.weight.split(H, dim=1) for lyr in enc}

    def msg_blocks(lyr):      # P = W1a h + b1, Q = W1c h
        return [(W1s[id(lyr)][0], lyr.W1.bias), (W1s[id(lyr)][2], None)]

    def dec_first_blocks():   # Pa_0, A_0 (without the sequence term), the encoder rows of every decoder layer
        if not dec:
            return []
        return ([(W1s[id(dec[0])][0], dec[0].W1.bias), (W1s[id(dec[0])][3], None)] +
                [(W1s[id(d_)][3], None) for d_ in dec])

    if enc and not has_occ:   # h_V = 0: P is the bias row, Q is zero (their weight blocks get zero gradients either way)
        proj = [enc[0].W1.bias.float().expand(N, H), hV]  # (hV is the zero tensor here; both are only read)
    else:
        first = msg_blocks(enc[0]) if enc else dec_first_blocks()
        proj = [linear(hV, W, b) for W, b in first]
    for l, lyr in enumerate(enc):
        W1, W11 = W1s[id(lyr)], W11s[id(lyr)]
        gsum, cnt = MessageFn.apply(g, slot, False, hE, proj[0], proj[1], None, W1[1], lyr.W2.weight, lyr.W2.bias)
        nxt = msg_blocks(enc[l + 1]) if l + 1 < len(enc) else dec_first_blocks()
        hV, proj = _node_update(lyr, hV, gsum, cnt, g.mask, p_drop, [(W11[0], lyr.W11.bias), (W11[2], None)] + nxt)
        drop_seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p_drop > 0 else 0
        out_slot = GradSlot()
        hE = EdgeUpdateFn.apply(g, slot, out_slot, p_drop, drop_seed, hE, proj[0], proj[1], W11[1],
                                lyr.W12.weight, lyr.W12.bias, lyr.W13.weight, lyr.W13.bias, lyr.norm3.weight,
                                lyr.norm3.bias)
        slot = out_slot
        proj = proj[2:]

    if dec:
        with torch.no_grad():
            cm = engine._f32(chain_M) * mask
            if fix_order is not None:
                order = engine._i64(fix_order)
            else:
                order = torch.argsort((cm + 0.0001) * torch.abs(torch.randn(cm.shape, device=dev)))
            rank = torch.zeros(N, device=dev, dtype=torch.int32)
            call("mpnn_order_rank", ptr(order), B, L, ptr(rank), stream_ptr())
            g.set_order(rank)
        # sequence term W1c W_s[S] of the decoded neighbours as a 21-row table per layer (the inference path's TS):
        # TS_l = W_s W1c_l^T, rows picked by S
        Sflat = S.reshape(-1)
        n_vocab = model.W_s.weight.shape[0]
        onehot = F.one_hot(Sflat, 32 * ((n_vocab + 31) // 32)).to(torch.float32)
        Pa, A0, Benc = proj[0], proj[1], proj[2:]
        for l, lyr in enumerate(dec):
            W1 = W1s[id(lyr)]
            A = SeqTermFn.apply(A0, model.W_s.weight, W1[2], Sflat, onehot)
            gsum, cnt = MessageFn.apply(g, slot, True, hE, Pa, A, Benc[l], W1[1], lyr.W2.weight, lyr.W2.bias)
            nxt = [(W1s[id(dec[l + 1])][0], dec[l + 1].W1.bias), (W1s[id(dec[l + 1])][3], None)] \
                if l + 1 < len(dec) else []
            hV, pr = _node_update(lyr, hV, gsum, cnt, g.mask, p_drop, nxt)
            if pr:
                Pa, A0 = pr
    return HeadFn.apply(hV, model.W_out.weight, model.W_out.bias).view(B, L, -1)


class HeadFn(torch.autograd.Function):
    """log_softmax(W_out h + b) (reference :527-528): mpnn_logits_log_softmax forward, mpnn_logits_log_softmax_bwd
    backward (d_logits, d_hV, dW, db in one launch)."""

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, hV, W, bias):
        N, n_letters = hV.shape[0], W.shape[0]
        W = W.detach().float().contiguous()
        nto, nti = (n_letters + 15) // 16, H // 16
        Wp = torch.empty(nto * nti * 256, device=hV.device, dtype=torch.float32)
        st = stream_ptr()
        call("mpnn_pack_linear", ptr(W), n_letters, H, H, 0, ptr(Wp), st)
        b = bias.detach().float().contiguous()
        hV = hV.contiguous()
        logp = torch.empty(N, n_letters, device=hV.device, dtype=torch.float32)
        call("mpnn_logits_log_softmax", ptr(hV), ptr(Wp), ptr(b), n_letters, N, ptr(logp), st)
        ctx.save_for_backward(hV, W, logp)
        return logp

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, d_logp):
        hV, W, logp = ctx.saved_tensors
        N, n_letters = logp.shape
        d_logp = d_logp.contiguous().float()
        d_hV = torch.empty_like(hV)
        dW = _zeros(hV.device, n_letters, H)
        db = _zeros(hV.device, n_letters)
        ws = _workspace(hV.device, N)
        call("mpnn_logits_log_softmax_bwd", ptr(d_logp), ptr(logp), ptr(hV), ptr(W), n_letters, N, ptr(d_hV), ptr(dW),
             ptr(db), ptr(ws), ws.numel(), stream_ptr())
        return d_hV, dW, db


class SmoothedLossFn(torch.autograd.Function):
    """loss_smoothed (reference train.py:63-73) in one launch: per-residue loss, the masked sum / 2000, and the gradient
    of that sum, kept for backward."""

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, log_probs, S, mask, weight):
        n_letters = log_probs.shape[-1]
        lp = log_probs.contiguous().float()
        N = lp.numel() // n_letters
        dev = lp.device
        loss = torch.empty(S.shape, device=dev, dtype=torch.float32)
        total = torch.empty((), device=dev, dtype=torch.float32)
        d_lp = torch.empty_like(lp)
        ws = torch.empty((N + 255) // 256, device=dev, dtype=torch.float32)
        call("mpnn_loss_smoothed", ptr(lp), ptr(S.contiguous()), ptr(mask.contiguous().float()), N, n_letters,
             float(weight), 1.0 / 2000.0, ptr(loss), ptr(total), ptr(d_lp), ptr(ws), ws.numel(), stream_ptr())
        ctx.save_for_backward(d_lp)
        ctx.mark_non_differentiable(loss)
        ctx.set_materialize_grads(False)
        return loss, total

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, _g_loss, g_total):
        (d_lp,) = ctx.saved_tensors
        if g_total is None:
            return None, None, None, None
        return d_lp * g_total, None, None, None


def loss_smoothed_fused(S, log_probs, mask, weight=0.1):
    """`loss_smoothed` on the HIP kernel (same returns; the per-residue loss is not differentiable here, the scalar is)."""
    return SmoothedLossFn.apply(log_probs, S, mask, weight)


def loss_smoothed(S, log_probs, mask, weight=0.1):
    """Label-smoothed NLL, summed and divided by 2000 (reference train.py:63-73) -> (per-residue loss, scalar)."""
    n = log_probs.shape[-1]
    S1 = F.one_hot(S, n).float()
    S1 = S1 + weight / float(n)
    S1 = S1 / S1.sum(-1, keepdim=True)
    loss = -(S1 * log_probs).sum(-1)
    return loss, torch.sum(loss * mask) / 2000.0


def loss_nll(S, log_probs, mask):
    """Per-residue NLL, its masked mean and the per-residue argmax hits (reference train.py:51-60; the metrics the
    reference's loop accumulates after every step, :397-399)."""
    loss = -log_probs.gather(-1, S.unsqueeze(-1)).squeeze(-1)
    true_false = torch.argmax(log_probs, -1) == S
    return loss, torch.sum(loss * mask) / torch.sum(mask), true_false


def make_optimizer(model, lr=1e-3, weight_decay=1e-2, total_steps=None, max_grad_norm=0.0):
    """AdamW and, when `total_steps` is given, the reference's schedule (train.py:217-222): linear warm-up from 1e-8 over
    the first 1 % of the steps, then cosine annealing to 0.  On a ROCm device the optimizer is the in-tree multi-tensor
    kernel (`mpnn_amd.optim.AdamW`: all 1.66 M parameters in three launches, gradient clipping of train.py:389-390
    folded in when `max_grad_norm` > 0); a model with any parameter off the device gets torch.optim.AdamW (train_step
    then clips with torch.nn.utils.clip_grad_norm_).  -> (optimizer, scheduler or None)."""
    from torch.optim.lr_scheduler import CosineAnnealingLR, LinearLR, SequentialLR
    if all(p.is_cuda for p in model.parameters()):
        from .optim import AdamW
        optimizer = AdamW(model.parameters(), lr=lr, weight_decay=weight_decay, max_grad_norm=max_grad_norm)
    else:
        optimizer = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=weight_decay)
    if total_steps is None:
        return optimizer, None
    warmup_steps = int(0.01 * total_steps)
    cosine = CosineAnnealingLR(optimizer, T_max=max(1, total_steps - warmup_steps), eta_min=0.0)
    if warmup_steps == 0:
        # fewer than 100 steps: the reference's SequentialLR would freeze the rate at lr * 1e-8 (a zero-length LinearLR
        # never leaves its start factor); cosine only
        return optimizer, cosine
    warmup = LinearLR(optimizer, start_factor=1e-8, total_iters=warmup_steps)
    return optimizer, SequentialLR(optimizer, schedulers=[warmup, cosine], milestones=[warmup_steps])


def train_step(model, optimizer, batch, world_size=1, fix_order=None, fix_backbone_noise=None, scheduler=None,
               max_grad_norm=None, allreduce_timings=None):
    """One optimisation step (reference train.py:376-395): forward, smoothed loss, backward, (gradient all-reduce),
    optional gradient-norm clipping (:389-390), optimizer and scheduler step.  max_grad_norm: None keeps what the
    in-tree optimizer was built with (no clipping for other optimizers); a number sets it for this and later steps,
    0 switching the clip off."""
    from . import distributed
    X, S, mask, chain_M, residue_idx, chain_enc = batch
    optimizer.zero_grad(set_to_none=True)
    log_probs = forward_train(model, X, S, mask, chain_M, residue_idx, chain_enc, fix_order, fix_backbone_noise)
    mask_for_loss = mask * chain_M
    _, loss = loss_smoothed_fused(S, log_probs, mask_for_loss)
    loss.backward()
    # no-op for a single rank; allreduce_timings (a list) receives one HIP-event pair around the collective
    distributed.allreduce_gradients(model.parameters(), world_size, timings=allreduce_timings)
    from . import optim
    if isinstance(optimizer, optim.AdamW):
        if max_grad_norm is not None:
            optimizer.max_grad_norm = float(max_grad_norm)      # clipped inside the update pass; 0 switches it off again
    elif max_grad_norm is not None and max_grad_norm > 0.0:
        params = [p for p in model.parameters() if p.grad is not None]
        if all(p.grad.is_cuda and p.grad.dtype == torch.float32 and p.grad.is_contiguous() for p in params):
            optim.clip_grad_norm_(params, max_grad_norm)
        else:   # gradients the HIP kernels do not take (CPU, non-contiguous, other dtypes): torch's own clip
            torch.nn.utils.clip_grad_norm_(params, max_grad_norm)
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    return loss.detach(), log_probs.detach()
