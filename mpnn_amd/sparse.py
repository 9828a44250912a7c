"""The reference's second model family on the same gfx950 kernels: the PyTorch-Geometric style ProteinMPNN of
src/mpnn/models/protein_mpnn.py (flat node tensors + a `batch` vector, k-NN / radius graphs with self loops, sum
aggregation over an edge list).  SURVEY.md section 8f, row N4.

Same constructor, same `forward` signature and the same state_dict keys as the reference class, so a checkpoint of the
reference loads with `load_state_dict`.  `forward` takes one extra, optional argument `decoding_order` because the
reference draws the order inside the call (protein_mpnn.py:30-31).  With grad mode on (training, or fine-tuning in eval
mode) the call builds an autograd graph over the dense model's backward kernels (`_forward_train`: backbone noise
:461-462, the five dropouts :136-144 / :198-200); under `torch.no_grad()` it runs the inference kernels.

How the edge list maps onto the dense kernels (nothing here is a second implementation of the arithmetic):
  * nodes are scattered into a padded [B, Lmax] batch; every residue owns a neighbour list of K slots
    (knn: K = min(num_neighbors, Lmax); radius graph: K = the largest in-radius degree, at most 64);
  * a slot is a VALID edge when centre and neighbour are both resolved residues of the same graph (and within the
    cutoff): the reference builds its graph on the valid residues only (:403-416), so exactly the valid slots are its
    edges, self loop included; invalid slots carry weight 0 in every neighbour sum (edge_mask of the kernels);
  * the neighbour search runs on coordinates in which unresolved / padding residues sit far outside the structure,
    so the K nearest of a valid residue are its K nearest VALID residues (the dense model's tie between masked
    columns and the farthest valid one, model_utils.py:273-275, has no counterpart here);
  * edge_mlp = Linear(+bias) -> LayerNorm -> Linear is mpnn_edge_features_biased; its input order [RBF | positional],
    the atom-pair order (atom a of the NEIGHBOUR against atom b of the CENTRE, :434-435) and the sign of the sequence
    offset (neighbour - centre, :480) are absorbed into the packed weight and the positional table;
  * Encoder = mpnn_enc_message + mpnn_node_update + mpnn_enc_edge_update with node mask 1 (the reference does not
    re-mask nodes, :136-138), the column blocks of out_v.0 / out_e.0 re-assigned ([x_i | x_j | e], and for out_e
    [x_row | x_col | e] = [neighbour | centre | e], :141-142);
  * Decoder = mpnn_dec_message_valid + mpnn_node_update: both branches of Decoder.message read the neighbour's
    CURRENT features and differ only in the label term (:212-214).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import _lib
from ._lib import call, ptr, stream_ptr
from .engine import (H, PackedWeights, _f32, _i64, _require_device, check_weight_range, numerics_guard)

__all__ = ["ProteinMPNN", "Encoder", "Decoder", "PositionWiseFeedForward", "PositionalEncoding", "to_dense_layout"]

MAX_K = 64  # neighbour slots per residue the edge kernels hold (four 16-edge column tiles)


class PositionWiseFeedForward(nn.Module):
    """protein_mpnn.py:54-68 (parameters: out.0, out.2)."""

    def __init__(self, in_channels: int, hidden_channels: int) -> None:
        super().__init__()
        self.out = nn.Sequential(nn.Linear(in_channels, hidden_channels), nn.GELU(),
                                 nn.Linear(hidden_channels, in_channels))


class PositionalEncoding(nn.Module):
    """protein_mpnn.py:71-85 (parameter: emb.weight [2 * 32 + 2, 16])."""

    def __init__(self, hidden_channels: int, max_relative_feature: int = 32) -> None:
        super().__init__()
        self.max_relative_feature = max_relative_feature
        self.emb = nn.Embedding(2 * max_relative_feature + 2, hidden_channels)


def _mlp3(in_channels, hidden_channels):
    return nn.Sequential(nn.Linear(in_channels, hidden_channels), nn.GELU(),
                         nn.Linear(hidden_channels, hidden_channels), nn.GELU(),
                         nn.Linear(hidden_channels, hidden_channels))


class Encoder(nn.Module):
    """Parameter container of protein_mpnn.py:88-153; the layer runs inside ProteinMPNN.forward."""

    def __init__(self, in_channels: int, hidden_channels: int, dropout: float = 0.1, scale: float = 30) -> None:
        super().__init__()
        self.out_v = _mlp3(in_channels, hidden_channels)
        self.out_e = _mlp3(in_channels, hidden_channels)
        self.dropout1, self.dropout2, self.dropout3 = (nn.Dropout(dropout) for _ in range(3))
        self.norm1, self.norm2, self.norm3 = (nn.LayerNorm(hidden_channels) for _ in range(3))
        self.scale = scale
        self.dense = PositionWiseFeedForward(hidden_channels, hidden_channels * 4)


class Decoder(nn.Module):
    """Parameter container of protein_mpnn.py:156-217."""

    def __init__(self, in_channels: int, hidden_channels: int, dropout: float = 0.1, scale: float = 30) -> None:
        super().__init__()
        self.out_v = _mlp3(in_channels, hidden_channels)
        self.dropout1, self.dropout2 = (nn.Dropout(dropout) for _ in range(2))
        self.norm1, self.norm2 = (nn.LayerNorm(hidden_channels) for _ in range(2))
        self.scale = scale
        self.dense = PositionWiseFeedForward(hidden_channels, hidden_channels * 4)


def to_dense_layout(batch: torch.Tensor):
    """`batch` [n] (sorted graph ids, as PyG's knn_graph requires) -> (B, Lmax, slot [n]): node v sits at row
    slot[v] = batch[v] * Lmax + (v - first node of its graph) of the padded [B * Lmax] layout.  One host read (Lmax)."""
    batch = _i64(batch)
    if batch.numel() == 0:
        raise ValueError("empty batch")
    counts = torch.bincount(batch)
    B = counts.numel()
    ordered = (batch[1:] >= batch[:-1]).all() if batch.numel() > 1 else torch.ones((), dtype=torch.bool, device=batch.device)
    Lmax, ok, lo = torch.stack([counts.max(), ordered.to(counts.dtype), batch.min()]).tolist()  # the one host read
    if not ok or lo < 0:
        raise ValueError("`batch` must hold non-negative graph ids in non-decreasing order (as knn_graph requires)")
    first = torch.cumsum(counts, 0) - counts
    slot = batch * Lmax + (torch.arange(batch.numel(), device=batch.device) - first[batch])
    return B, Lmax, slot


class _Packed:
    """Kernel-layout weights of one sparse model (cached on the module, rebuilt when a parameter changes)."""

    def __init__(self, model: "ProteinMPNN"):
        self.key = PackedWeights.make_key(model)
        dev = model.output.weight.device
        st = stream_ptr()
        self._keep = []

        def pack_split(W, col0=0, ncols=None, out=None):
            W = _f32(W)
            self._keep.append(W)
            out_f, ld = W.shape
            ncols = ld - col0 if ncols is None else ncols
            nto, nkk = (out_f + 15) // 16, (ncols + 31) // 32
            if out is None:
                out = torch.empty(nto * nkk * 2 * 64 * 4, device=dev, dtype=torch.float32)
            call("mpnn_pack_linear_split", ptr(W), out_f, ncols, ld, col0, 0, ptr(out), st)
            return out

        def proj_group(blocks):
            n = len(blocks)
            if n == 0:
                return None, None, 0
            Wp = torch.empty(n * 16384, device=dev, dtype=torch.float32)
            bp = torch.zeros(n * H, device=dev, dtype=torch.float32)
            for i, (W, col0, b) in enumerate(blocks):
                pack_split(W, col0, H, out=Wp[i * 16384:(i + 1) * 16384])
                if b is not None:
                    bp[i * H:(i + 1) * H] = _f32(b)
            return Wp, bp, n

        def ffn(lyr):
            w_in, w_out = lyr.dense.out[0], lyr.dense.out[2]
            return dict(Win=pack_split(w_in.weight), bin=_f32(w_in.bias),
                        Wout=torch.cat([pack_split(w_out.weight, H * q, H) for q in range(4)]), bout=_f32(w_out.bias))

        A = 6 if model.use_virtual_center else 5
        self.n_atoms = A
        # positional table: the kernel indexes it with clip(centre - neighbour + 32, 0, 64) (65 = other chain); the
        # reference embeds clip(neighbour - centre + 32, 0, 64) (:480-483, :82-85): rows 0..64 reversed
        emb = _f32(model.embedding.emb.weight)
        self.pos_table = torch.cat([emb[:65].flip(0), emb[65:66]]).contiguous()
        self.rbf_mu = torch.linspace(2.0, 22.0, 16, device=dev, dtype=torch.float32)
        # edge_mlp.0: input [RBF of (atom a at the neighbour, atom b at the centre), a-major | positional]; the kernel
        # generates tile p = (centre atom p // A, neighbour atom p % A), then the positional tile
        W0 = _f32(model.edge_mlp[0].weight)
        p = torch.arange(A * A, device=dev)
        src_block = (p % A) * A + (p // A)
        cols = (src_block[:, None] * 16 + torch.arange(16, device=dev)[None, :]).reshape(-1)
        cols = torch.cat([cols, A * A * 16 + torch.arange(16, device=dev)])
        self.Wemb = pack_split(W0[:, cols].contiguous())
        self.bemb = _f32(model.edge_mlp[0].bias)
        self.ln_e_w, self.ln_e_b = _f32(model.edge_mlp[1].weight), _f32(model.edge_mlp[1].bias)
        self.We, self.be = pack_split(model.edge_mlp[2].weight), _f32(model.edge_mlp[2].bias)

        enc, dec = list(model.encoder_layers), list(model.decoder_layers)
        self.enc, self.dec = [], []
        for lyr in enc:
            v, e = lyr.out_v, lyr.out_e
            d = dict(W1b=pack_split(v[0].weight, 2 * H, H), W2=pack_split(v[2].weight), b2=_f32(v[2].bias),
                     W3=pack_split(v[4].weight), b3=_f32(v[4].bias),
                     W11b=pack_split(e[0].weight, 2 * H, H), W12=pack_split(e[2].weight), b12=_f32(e[2].bias),
                     W13=pack_split(e[4].weight), b13=_f32(e[4].bias),
                     ln1w=_f32(lyr.norm1.weight), ln1b=_f32(lyr.norm1.bias), ln2w=_f32(lyr.norm2.weight),
                     ln2b=_f32(lyr.norm2.bias), ln3w=_f32(lyr.norm3.weight), ln3b=_f32(lyr.norm3.bias),
                     scale=float(lyr.scale), **ffn(lyr))
            self.enc.append(d)
        for lyr in dec:
            v = lyr.out_v
            d = dict(W1b=pack_split(v[0].weight, 2 * H, H), W2=pack_split(v[2].weight), b2=_f32(v[2].bias),
                     W3=pack_split(v[4].weight), b3=_f32(v[4].bias),
                     ln1w=_f32(lyr.norm1.weight), ln1b=_f32(lyr.norm1.bias), ln2w=_f32(lyr.norm2.weight),
                     ln2b=_f32(lyr.norm2.bias), scale=float(lyr.scale), **ffn(lyr))
            self.dec.append(d)

        def enc_msg_blocks(l):  # out_v.0 on [x_i | x_j | e] (:151): P = W[:, :H] x_i + b, Q = W[:, H:2H] x_j
            lin = enc[l].out_v[0]
            return [(lin.weight, 0, lin.bias), (lin.weight, H, None)]

        def enc_upd_blocks(l):  # out_e.0 on [x_row | x_col | e] = [neighbour | centre | e] (:140-142)
            lin = enc[l].out_e[0]
            return [(lin.weight, H, lin.bias), (lin.weight, 0, None)]

        def dec_blocks(l):  # out_v.0 on [x_i | x_j | e | label_j] (:212-216): Pa | A (+ label table) | B
            lin = dec[l].out_v[0]
            return [(lin.weight, 0, lin.bias), (lin.weight, H, None), (lin.weight, H, None)]

        first_dec = dec_blocks(0) if dec else []
        self.g_init = proj_group(enc_msg_blocks(0) if enc else first_dec)
        self.g_enc = [proj_group(enc_upd_blocks(l) + (enc_msg_blocks(l + 1) if l + 1 < len(enc) else first_dec))
                      for l in range(len(enc))]
        self.g_dec = [proj_group(dec_blocks(l + 1)) if l + 1 < len(dec) else (None, None, 0) for l in range(len(dec))]
        # label table: T[s, l*H:(l+1)*H] = out_v.0.weight[:, 3H:4H] of decoder l applied to label_embedding[s]
        self.n_vocab = model.label_embedding.weight.shape[0]
        self.TL = None
        if dec:
            lab = _f32(model.label_embedding.weight)
            g_t = proj_group([(d_.out_v[0].weight, 3 * H, None) for d_ in dec])
            self.TL = torch.empty(self.n_vocab, len(dec) * H, device=dev, dtype=torch.float32)
            call("mpnn_node_update", ptr(lab), None, None, 0.0, None, None, None, None, None, None, None, None, None,
                 None, None, 1.0, 0, ptr(g_t[0]), ptr(g_t[1]), g_t[2], None, 0, 0, None, 0, self.n_vocab, None,
                 ptr(self.TL), st)
            self._keep += [lab, g_t]
        Wo = _f32(model.output.weight)
        self._keep.append(Wo)
        self.n_letters = Wo.shape[0]
        self.Wlogit = torch.empty(((self.n_letters + 15) // 16) * (H // 16) * 256, device=dev, dtype=torch.float32)
        call("mpnn_pack_linear", ptr(Wo), self.n_letters, H, H, 0, ptr(self.Wlogit), st)
        self.blogit = _f32(model.output.bias)
        self.occ = None
        if model.occupancy_cutoff is not None:
            m = model.occupancy_mlp
            W2 = _f32(m[2].weight)
            W2p = torch.empty(8 * 8 * 256, device=dev, dtype=torch.float32)
            call("mpnn_pack_linear", ptr(W2), H, H, H, 0, ptr(W2p), st)
            self._keep.append(W2)
            self.occ = dict(w0=_f32(m[0].weight).reshape(-1), b0=_f32(m[0].bias), W2=W2p, b2=_f32(m[2].bias),
                            lnw=_f32(m[3].weight), lnb=_f32(m[3].bias))


def _packed(model) -> _Packed:
    pw = model.__dict__.get("_packed")
    if pw is None or pw.key != PackedWeights.make_key(model):
        check_weight_range(model)
        saved = _lib.set_precision("fp32")  # derived tables are always built with fp32-class products
        try:
            pw = _Packed(model)
        finally:
            _lib.set_precision(saved)
        model.__dict__["_packed"] = pw
    return pw


class ProteinMPNN(nn.Module):
    """Drop-in for the reference's `mpnn.models.protein_mpnn.ProteinMPNN` (:220-515) on MI355X.

    forward(x [n,4,3], chain_seq_label [n], mask [n], chain_mask_all [n], residue_idx [n], chain_encoding_all [n],
    batch [n]) -> log-probs [n, vocab_size] for every node, resolved or not, as the reference returns them."""

    def __init__(self, hidden_dim: int = 128, num_encoder_layers: int = 3, num_decoder_layers: int = 3,
                 num_neighbors: int = 30, edge_cutoff: Optional[float] = None, num_rbf: int = 16, dropout: float = 0.1,
                 augment_eps: float = 0.2, num_positional_embedding: int = 16, vocab_size: int = 21,
                 checkpoint_featurize: bool = False, use_virtual_center: bool = False,
                 occupancy_cutoff: Optional[float] = None) -> None:
        super().__init__()
        if hidden_dim != H or num_rbf != 16 or num_positional_embedding != 16:
            raise ValueError("mpnn_amd: the gfx950 kernels are built for hidden_dim = 128, num_rbf = 16 and "
                             "num_positional_embedding = 16 (the reference's defaults)")
        if not 1 <= vocab_size <= 32:
            raise ValueError("mpnn_amd: vocab_size must be in 1..32")
        if edge_cutoff is None and not 1 <= num_neighbors <= MAX_K:
            raise ValueError(f"mpnn_amd: num_neighbors must be in 1..{MAX_K}")
        self.augment_eps = augment_eps
        self.hidden_dim = hidden_dim
        self.num_neighbors = num_neighbors
        self.edge_cutoff = edge_cutoff
        self.num_rbf = num_rbf
        self.checkpoint_featurize = checkpoint_featurize
        self.use_virtual_center = use_virtual_center
        self.occupancy_cutoff = occupancy_cutoff
        self.embedding = PositionalEncoding(num_positional_embedding)
        n_pairs = (6 if use_virtual_center else 5) ** 2
        self.edge_mlp = nn.Sequential(nn.Linear(num_positional_embedding + num_rbf * n_pairs, hidden_dim),
                                      nn.LayerNorm(hidden_dim), nn.Linear(hidden_dim, hidden_dim))
        if occupancy_cutoff is not None:
            self.occupancy_mlp = nn.Sequential(nn.Linear(1, hidden_dim), nn.GELU(), nn.Linear(hidden_dim, hidden_dim),
                                               nn.LayerNorm(hidden_dim))
        self.label_embedding = nn.Embedding(vocab_size, hidden_dim)
        self.encoder_layers = nn.ModuleList([Encoder(hidden_dim * 3, hidden_dim, dropout)
                                             for _ in range(num_encoder_layers)])
        self.decoder_layers = nn.ModuleList([Decoder(hidden_dim * 4, hidden_dim, dropout)
                                             for _ in range(num_decoder_layers)])
        self.output = nn.Linear(hidden_dim, vocab_size)
        self.precision = "fp32"
        self.reset_parameters()

    def reset_parameters(self):
        """protein_mpnn.py:315-319."""
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    # -----------------------------------------------------------------------------------------------------------
    def _graph(self, Xd, maskd, B, L):
        """atoms, E_idx32, E_idx64, valid [B,L,K] (float), K for the padded batch."""
        dev = Xd.device
        st = stream_ptr()
        atoms = torch.empty(B * L * 24, device=dev, dtype=torch.float32)
        ca_soa = torch.empty(B, 4, L, device=dev, dtype=torch.float32)
        call("mpnn_backbone_atoms", ptr(Xd), None, ptr(maskd), B, L, ptr(atoms), ptr(ca_soa), st)
        # search coordinates: unresolved and padding residues far outside every structure, all columns unmasked
        m = maskd.view(B, 1, L) > 0
        far = 4.0 * (ca_soa[:, :3] * m).abs().amax() + 1000.0
        ca_soa[:, :3] = torch.where(m, ca_soa[:, :3], far)
        ca_soa[:, 3] = 1.0

        def search(K):
            Kp = 16 * ((K + 15) // 16)
            E64 = torch.empty(B, L, K, device=dev, dtype=torch.int64)
            E32 = torch.empty(B * L * Kp, device=dev, dtype=torch.int32)
            em = torch.empty(B, L, K, device=dev, dtype=torch.float32) if self.edge_cutoff is not None else None
            call("mpnn_knn_topk", ptr(ca_soa), B, L, K, float(self.edge_cutoff or 0.0), ptr(E64), ptr(E32), None,
                 ptr(em), st)
            valid = maskd.view(B, L, 1) * torch.gather(maskd.view(B, 1, L).expand(B, L, L), 2, E64)
            if em is not None:
                valid = valid * em
            return E64, E32, valid.contiguous()

        if self.edge_cutoff is None:
            K = min(int(self.num_neighbors), L)
            E64, E32, valid = search(K)
        else:
            # radius graph (:407-414): neighbour lists sized to the largest in-radius degree (one host read)
            K0 = min(MAX_K, L)
            E64, E32, valid = search(K0)
            deg = int(valid.sum(-1).max())
            if K0 < L and deg >= K0:
                raise RuntimeError(f"mpnn_amd: a residue has {MAX_K} or more neighbours within edge_cutoff = "
                                   f"{self.edge_cutoff}; the edge kernels hold at most {MAX_K} per residue")
            K = max(deg, 1)
            if K != K0:
                E64, E32, valid = search(K)
        return atoms, E32, E64, valid, K

    def forward(self, x: torch.Tensor, chain_seq_label: torch.Tensor, mask: torch.Tensor,
                chain_mask_all: torch.Tensor, residue_idx: torch.Tensor, chain_encoding_all: torch.Tensor,
                batch: torch.Tensor, decoding_order: Optional[torch.Tensor] = None) -> torch.Tensor:
        """protein_mpnn.py:450-515.  decoding_order (optional, [B, Lmax] permutations of the padded rows) replaces the
        order the reference draws at :30-31; without it the same distribution is drawn here."""
        _require_device(x, "x")
        # autograd semantics as for any nn.Module (and as mpnn_amd.ProteinMPNN): with grad mode on and trainable
        # parameters the differentiable path runs - in eval mode too (no dropout, no noise); inference callers use
        # torch.no_grad()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            with torch.autocast(device_type="cuda", enabled=False):
                return self._forward_train(x, chain_seq_label, mask, chain_mask_all, residue_idx, chain_encoding_all,
                                           batch, decoding_order)
        if self.training:
            raise RuntimeError("mpnn_amd.sparse.ProteinMPNN in training mode needs grad mode on (its dropout / backbone "
                               "noise live on the differentiable path); call .eval() for inference under no_grad")
        guard = numerics_guard(self, x.device)
        guard.poll()
        _lib.set_precision(self.precision)
        try:
            with torch.no_grad():
                out = self._forward(x, chain_seq_label, mask, chain_mask_all, residue_idx, chain_encoding_all, batch,
                                    decoding_order)
            guard.watch(out, "a forward call of the sparse model")
            return out
        finally:
            _lib.set_precision("fp32")

    def check_numerics(self):
        """As mpnn_amd.ProteinMPNN.check_numerics: wait for the calls issued so far and raise RuntimeError naming every
        call that produced inf / NaN log-probs; without it a bad call is only warned about when the model is called next."""
        g = self.__dict__.get("_numerics")
        if g is not None:
            g.check()
        return self

    def _dense_inputs(self, x, S, mask, chain_mask_all, residue_idx, chain_encoding_all, batch, noise_eps=0.0):
        """Flat node tensors -> the padded [B * Lmax] layout of the dense kernels.
        -> (B, L, slot, scatter, Xd, maskd, ridx, chain, Sd, maskf)"""
        dev = x.device
        n = x.shape[0]
        if x.dim() != 3 or x.shape[1] < 4 or x.shape[2] != 3:
            raise ValueError("x must be [n, 4, 3] (N, CA, C, O)")
        for name, t in (("chain_seq_label", S), ("mask", mask), ("chain_mask_all", chain_mask_all),
                        ("residue_idx", residue_idx), ("chain_encoding_all", chain_encoding_all), ("batch", batch)):
            if t.shape[0] != n or t.dim() != 1:
                raise ValueError(f"{name} must have shape [{n}]")
        B, L, slot = to_dense_layout(batch)
        N = B * L
        maskf = _f32(mask)

        def scatter(src, dtype):
            out = torch.zeros((N,) + tuple(src.shape[1:]), device=dev, dtype=dtype)
            out[slot] = src.to(dtype)
            return out

        xv = _f32(x[:, :4])
        if noise_eps > 0:   # :461-462
            xv = xv + noise_eps * torch.randn_like(xv)
        # unresolved residues keep whatever coordinates they came with in the reference, which never reads them
        xv = torch.where((maskf > 0).view(n, 1, 1), xv, torch.zeros((), device=dev))
        Xd = scatter(xv, torch.float32)
        maskd = scatter(maskf, torch.float32)
        ridx = scatter(_i64(residue_idx), torch.int64)
        chain = scatter(_i64(chain_encoding_all), torch.int64)
        Sd = scatter(_i64(S), torch.int64)
        return B, L, slot, scatter, Xd, maskd, ridx, chain, Sd, maskf

    def _decoding_order(self, decoding_order, scatter, chain_mask_all, maskf, B, L, dev):
        if decoding_order is not None:
            order = _i64(decoding_order)
            if tuple(order.shape) != (B, L):
                raise ValueError(f"decoding_order must be [{B}, {L}] (the padded batch)")
            if not bool((torch.sort(order, dim=-1).values == torch.arange(L, device=dev)).all()):
                raise ValueError("every row of decoding_order must be a permutation of 0..Lmax-1")
            return order
        # :27-31 (padding rows carry 0, as to_dense_batch fills them)
        key = scatter(_f32(chain_mask_all) * maskf, torch.float32).view(B, L)
        return torch.argsort((key + 0.0001) * torch.abs(torch.randn(B, L, device=dev)))

    def _forward(self, x, S, mask, chain_mask_all, residue_idx, chain_encoding_all, batch, decoding_order):
        dev = x.device
        st = stream_ptr()
        pw = _packed(self)
        B, L, slot, scatter, Xd, maskd, ridx, chain, Sd, maskf = self._dense_inputs(
            x, S, mask, chain_mask_all, residue_idx, chain_encoding_all, batch)
        N = B * L
        if int(pw.n_vocab) <= 0:
            raise ValueError("empty vocabulary")

        atoms, E32, E64, valid, K = self._graph(Xd, maskd, B, L)
        KT = (K + 15) // 16
        hE = torch.empty(N * 8 * KT * 256, device=dev, dtype=torch.float32)
        call("mpnn_edge_features_biased", ptr(atoms), ptr(E32), ptr(ridx), ptr(chain), ptr(pw.pos_table),
             ptr(pw.rbf_mu), ptr(pw.Wemb), ptr(pw.bemb), ptr(pw.ln_e_w), ptr(pw.ln_e_b), ptr(pw.We), ptr(pw.be), B, L, K,
             pw.n_atoms, None, ptr(hE), st)

        hV = torch.zeros(N, H, device=dev, dtype=torch.float32)
        if pw.occ is not None:
            occ = torch.empty(N, device=dev, dtype=torch.float32)
            call("mpnn_occupancy", ptr(atoms), ptr(maskd), B, L, float(self.occupancy_cutoff), ptr(occ), st)
            o = pw.occ
            call("mpnn_occupancy_mlp", ptr(occ), ptr(o["w0"]), ptr(o["b0"]), ptr(o["W2"]), ptr(o["b2"]), ptr(o["lnw"]),
                 ptr(o["lnb"]), N, ptr(hV), st)

        ones = torch.ones(N, device=dev, dtype=torch.float32)
        gsum = torch.empty(N, H, device=dev, dtype=torch.float32)
        cnt = torch.empty(N, device=dev, dtype=torch.float32)
        n_enc, n_dec = len(pw.enc), len(pw.dec)
        ldt = pw.TL.shape[1] if pw.TL is not None else 0

        def node_update(hV_in, lw, cnt_, group, table):
            Wp, bp, nproj = group
            PQ = torch.empty(N, nproj * H, device=dev, dtype=torch.float32) if nproj else None
            tb = (ptr(pw.TL), ldt, table[0], ptr(Sd), table[1]) if table is not None else (None, 0, 0, None, 0)
            if lw is None:
                call("mpnn_node_update", ptr(hV_in), None, None, 0.0, None, None, None, None, None, None, None, None,
                     None, None, None, 1.0, 0, ptr(Wp), ptr(bp), nproj, *tb, N, None, ptr(PQ), st)
                return hV_in, PQ
            hV_out = torch.empty_like(hV_in)
            call("mpnn_node_update", ptr(hV_in), ptr(gsum), ptr(cnt_), 0.0, ptr(ones), ptr(lw["W3"]), ptr(lw["b3"]),
                 ptr(lw["ln1w"]), ptr(lw["ln1b"]), ptr(lw["Win"]), ptr(lw["bin"]), ptr(lw["Wout"]), ptr(lw["bout"]),
                 ptr(lw["ln2w"]), ptr(lw["ln2b"]), lw["scale"], 1, ptr(Wp), ptr(bp), nproj, *tb, N, ptr(hV_out),
                 ptr(PQ), st)
            return hV_out, PQ

        def label_table(layer, block):  # adds T_layer[label] to projection block `block` (the decoder's A rows)
            return (H * layer, block)

        PQ = None
        if pw.g_init[2]:
            _, PQ = node_update(hV, None, None, pw.g_init, label_table(0, 1) if (n_enc == 0 and n_dec) else None)

        for l, lw in enumerate(pw.enc):
            offP, offQ = (0, H) if l == 0 else (2 * H, 3 * H)
            call("mpnn_enc_message", ptr(hE), ptr(PQ), PQ.shape[1], offP, offQ, ptr(E32), ptr(ones), ptr(valid),
                 ptr(lw["W1b"]), ptr(lw["W2"]), ptr(lw["b2"]), B, L, K, ptr(gsum), ptr(cnt), st)
            last = l + 1 == n_enc
            hV, PQ = node_update(hV, lw, cnt, pw.g_enc[l], label_table(0, 3) if (last and n_dec) else None)
            call("mpnn_enc_edge_update", ptr(hE), ptr(PQ), PQ.shape[1], 0, H, ptr(E32), ptr(lw["W11b"]),
                 ptr(lw["W12"]), ptr(lw["b12"]), ptr(lw["W13"]), ptr(lw["b13"]), ptr(lw["ln3w"]), ptr(lw["ln3b"]), B, L,
                 K, st)

        if n_dec:
            order = self._decoding_order(decoding_order, scatter, chain_mask_all, maskf, B, L, dev)
            rank = torch.zeros(N, device=dev, dtype=torch.int32)
            call("mpnn_order_rank", ptr(order), B, L, ptr(rank), st)
            degree = valid.sum(-1).reshape(N).contiguous()
            base = 2 * H if n_enc else 0
            for l, lw in enumerate(pw.dec):
                off = base if l == 0 else 0
                call("mpnn_dec_message_valid", ptr(hE), ptr(PQ), PQ.shape[1], off, off + H, off + 2 * H, ptr(rank),
                     ptr(E32), ptr(ones), ptr(valid), ptr(lw["W1b"]), ptr(lw["W2"]), ptr(lw["b2"]), B, L, K, ptr(gsum),
                     st)
                grp = pw.g_dec[l]
                hV, PQ = node_update(hV, lw, degree, grp, label_table(l + 1, 1) if grp[2] else None)

        lp = torch.empty(N, pw.n_letters, device=dev, dtype=torch.float32)
        call("mpnn_logits_log_softmax", ptr(hV), ptr(pw.Wlogit), ptr(pw.blogit), pw.n_letters, N, ptr(lp), st)
        return lp[slot]

    # -----------------------------------------------------------------------------------------------------------
    def _forward_train(self, x, S, mask, chain_mask_all, residue_idx, chain_encoding_all, batch, decoding_order):
        """The forward of protein_mpnn.py:450-515 under autograd, on the dense model's training kernels
        (mpnn_amd.training: saved pre-activations, hand-written backward, weight gradients, deterministic sums).  The
        edge list maps onto them exactly as in `_forward`: node mask 1, slot validity as the edge weight
        (mpnn_message_train with w_valid for the decoder), the column blocks of out_v.0 / out_e.0 taken as views of
        the parameters so that their gradients land on the reference-shaped tensors."""
        import types

        from . import training as T

        dev = x.device
        p_drop = float(self.encoder_layers[0].dropout1.p if len(self.encoder_layers) else
                       (self.decoder_layers[0].dropout1.p if len(self.decoder_layers) else 0.0)) if self.training else 0.0
        eps = float(self.augment_eps) if self.training else 0.0
        B, L, slot, scatter, Xd, maskd, ridx, chain, Sd, maskf = self._dense_inputs(
            x, S, mask, chain_mask_all, residue_idx, chain_encoding_all, batch, noise_eps=eps)
        N = B * L
        T._ZEROS.reset(dev)
        self.__dict__.pop("_packed", None)   # an optimizer step follows: the inference kernels' packed weights go stale
        check_weight_range(self)
        with torch.no_grad():
            atoms, E32, E64, valid, K = self._graph(Xd, maskd, B, L)
            ones = torch.ones(N, device=dev, dtype=torch.float32)
            g = T.Graph(E64, E32, ones, valid, K)
            degree = valid.sum(-1).reshape(N).contiguous()
        A = 6 if self.use_virtual_center else 5
        # parameters in the kernels' conventions, as differentiable views / permutations (see _Packed)
        emb = self.embedding.emb.weight
        Wpos = torch.cat([emb[:65].flip(0), emb[65:66]]).t()             # table[d][p] = Wpos[p][d]
        bpos = torch.zeros(emb.shape[1], device=dev, dtype=torch.float32)
        pidx = torch.arange(A * A, device=dev)
        src_block = (pidx % A) * A + (pidx // A)
        rbf_cols = (src_block[:, None] * 16 + torch.arange(16, device=dev)[None, :]).reshape(-1)
        cols = torch.cat([A * A * 16 + torch.arange(16, device=dev), rbf_cols])   # [positional | RBF] as FeaturesFn packs
        W0 = self.edge_mlp[0].weight[:, cols]
        slot_e = T.GradSlot()
        hE = T.FeaturesFn.apply(g, slot_e, atoms, ridx, chain, A, Wpos, bpos, W0, self.edge_mlp[1].weight,
                                self.edge_mlp[1].bias, self.edge_mlp[2].weight, self.edge_mlp[2].bias,
                                self.edge_mlp[0].bias)
        hV = torch.zeros(N, H, device=dev, dtype=torch.float32)
        has_occ = self.occupancy_cutoff is not None
        if has_occ:
            with torch.no_grad():
                occ = torch.empty(N, device=dev, dtype=torch.float32)
                call("mpnn_occupancy", ptr(atoms), ptr(maskd), B, L, float(self.occupancy_cutoff), ptr(occ), stream_ptr())
            hV = hV + self.occupancy_mlp(occ.view(N, 1))

        enc, dec = list(self.encoder_layers), list(self.decoder_layers)

        def view(lyr):   # the attribute names training._node_update reads (the dense model's)
            return types.SimpleNamespace(W3=lyr.out_v[4], norm1=lyr.norm1, norm2=lyr.norm2, scale=lyr.scale,
                                         dense=types.SimpleNamespace(W_in=lyr.dense.out[0], W_out=lyr.dense.out[2]))

        Wv = {id(l_): l_.out_v[0].weight.split(H, dim=1) for l_ in enc + dec}
        We = {id(l_): l_.out_e[0].weight.split(H, dim=1) for l_ in enc}

        def msg_blocks(lyr):        # out_v.0 on [x_i | x_j | e] (:151): P = W[:, :H] x_i + b, Q = W[:, H:2H] x_j
            return [(Wv[id(lyr)][0], lyr.out_v[0].bias), (Wv[id(lyr)][1], None)]

        def dec_blocks(lyr):        # out_v.0 on [x_i | x_j | e | label_j] (:212-216): Pa, and the neighbour term
            return [(Wv[id(lyr)][0], lyr.out_v[0].bias), (Wv[id(lyr)][1], None)]

        first = msg_blocks(enc[0]) if enc else (dec_blocks(dec[0]) if dec else [])
        if enc and not has_occ:     # h_V = 0: P is the bias row, Q is zero
            proj = [enc[0].out_v[0].bias.float().expand(N, H), torch.zeros(N, H, device=dev, dtype=torch.float32)]
        else:
            proj = [T.linear(hV, W, b) for W, b in first]

        slot_cur = slot_e
        for l, lyr in enumerate(enc):
            gsum, cnt = T.MessageFn.apply(g, slot_cur, False, hE, proj[0], proj[1], None, Wv[id(lyr)][2],
                                          lyr.out_v[2].weight, lyr.out_v[2].bias)
            nxt = msg_blocks(enc[l + 1]) if l + 1 < len(enc) else (dec_blocks(dec[0]) if dec else [])
            # out_e.0 on [x_row | x_col | e] = [neighbour | centre | e] (:140-142): centre term P', neighbour term Q'
            upd = [(We[id(lyr)][1], lyr.out_e[0].bias), (We[id(lyr)][0], None)]
            hV, proj = T._node_update(view(lyr), hV, gsum, cnt, ones, p_drop, upd + nxt)
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p_drop > 0 else 0
            out_slot = T.GradSlot()
            hE = T.EdgeUpdateFn.apply(g, slot_cur, out_slot, p_drop, seed, hE, proj[0], proj[1], We[id(lyr)][2],
                                      lyr.out_e[2].weight, lyr.out_e[2].bias, lyr.out_e[4].weight, lyr.out_e[4].bias,
                                      lyr.norm3.weight, lyr.norm3.bias)
            slot_cur = out_slot
            proj = proj[2:]

        if dec:
            with torch.no_grad():
                order = self._decoding_order(decoding_order, scatter, chain_mask_all, maskf, B, L, dev)
                rank = torch.zeros(N, device=dev, dtype=torch.int32)
                call("mpnn_order_rank", ptr(order), B, L, ptr(rank), stream_ptr())
                g.set_order(rank)
            n_vocab = self.label_embedding.weight.shape[0]
            onehot = torch.nn.functional.one_hot(Sd, 32 * ((n_vocab + 31) // 32)).to(torch.float32)
            for l, lyr in enumerate(dec):
                Pa, Bv = proj[0], proj[1]
                # both branches of Decoder.message read the neighbour's current features; the label term is added
                # where the neighbour is decoded earlier (:212-214)
                TL = T.linear(self.label_embedding.weight, Wv[id(lyr)][3])
                Arow = Bv + T.TableRowsFn.apply(TL, Sd, onehot)
                gsum, _ = T.MessageFn.apply(g, slot_cur, "valid", hE, Pa, Arow, Bv, Wv[id(lyr)][2], lyr.out_v[2].weight,
                                            lyr.out_v[2].bias)
                nxt = dec_blocks(dec[l + 1]) if l + 1 < len(dec) else []
                hV, proj = T._node_update(view(lyr), hV, gsum, degree, ones, p_drop, nxt)
        lp = T.HeadFn.apply(hV, self.output.weight, self.output.bias)
        return lp[slot]
