"""CPU oracle for the sparse (edge-list) ProteinMPNN  --  TEST INFRASTRUCTURE, NOT PRODUCT.

A plain-PyTorch (CPU, fp32) restatement of `src/mpnn/models/protein_mpnn.py` (all `:line` citations below are to
that file), written functionally over a flat `state_dict` with explicit edge lists and `index_add_` in place of
torch_geometric's MessagePassing, so that it runs where torch_geometric / torch_cluster are absent.

PARITY UNPINNED for the model as a whole: the reference class imports `torch_geometric` (knn_graph, radius_graph,
MessagePassing, to_dense_batch), which is not installed in the build container and cannot be fetched, and the
reference's test-suite holds no output tensors for it.  What IS pinned:
  * the autoregressive mask, against the construction in the reference's own regression test
    (`tests/_test_autoregressive_mask.py:11-36`, the einsum over one-hot permutation matrices) on that test's inputs;
  * everything this model shares with the dense one - RBF, virtual centre, occupancy, exact-erf GELU, LayerNorm -
    through `mpnn_oracle`, which is pinned against reference-generated golden vectors.
The graph builders restate the documented behaviour of torch_cluster 1.6 (knn: the k nearest INCLUDING the point
itself when loop=True, edge = (neighbour -> centre); radius: squared distance < r^2, at most max_num_neighbors).

Only `tests/` may import this module.
"""

from __future__ import annotations

import itertools
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass
class Config:
    """Constructor arguments that change the maths (:254-311)."""

    num_neighbors: int = 30
    num_encoder_layers: int = 3
    num_decoder_layers: int = 3
    hidden_dim: int = 128
    edge_cutoff: Optional[float] = None
    use_virtual_center: bool = False
    occupancy_cutoff: Optional[float] = None
    num_rbf: int = 16
    max_relative_feature: int = 32
    scale: float = 30.0


def _linear(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def _layer_norm(sd, name, x):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], 1e-5)


def _mlp3(sd, name, x):
    """Linear - GELU - Linear - GELU - Linear (:100-113, :168-174)."""
    x = F.gelu(_linear(sd, name + ".0", x))
    x = F.gelu(_linear(sd, name + ".2", x))
    return _linear(sd, name + ".4", x)


# --------------------------------------------------------------------------
# graphs
# --------------------------------------------------------------------------
def knn_edges(pos: Tensor, batch: Tensor, k: int) -> Tensor:
    """knn_graph(pos, k, batch, loop=True) (:416): for every point its min(k, graph size) nearest points of the same
    graph, itself included; returns [2, E] with row 0 = neighbour (source), row 1 = centre (target)."""
    rows, cols = [], []
    for g in torch.unique(batch).tolist():
        idx = torch.nonzero(batch == g).flatten()
        p = pos[idx].double()
        d2 = ((p[:, None, :] - p[None, :, :]) ** 2).sum(-1)
        kk = min(k, idx.numel())
        nb = torch.topk(d2, kk, dim=-1, largest=False).indices  # [n, kk]
        rows.append(idx[nb].reshape(-1))
        cols.append(idx[:, None].expand(-1, kk).reshape(-1))
    return torch.stack([torch.cat(rows), torch.cat(cols)])


def radius_edges(pos: Tensor, batch: Tensor, r: float, loop: bool, max_num_neighbors: int = 1000) -> Tensor:
    """radius_graph(pos, r, batch, loop, max_num_neighbors) (:353-359, :408-414): squared distance < r^2."""
    rows, cols = [], []
    for g in torch.unique(batch).tolist():
        idx = torch.nonzero(batch == g).flatten()
        p = pos[idx]
        d2 = ((p[:, None, :] - p[None, :, :]) ** 2).sum(-1)
        hit = d2 < r * r
        if not loop:
            hit &= ~torch.eye(idx.numel(), dtype=torch.bool)
        assert int(hit.sum(-1).max()) <= max_num_neighbors, "beyond max_num_neighbors: the reference would truncate"
        c, n = torch.nonzero(hit, as_tuple=True)  # centre, neighbour
        rows.append(idx[n])
        cols.append(idx[c])
    return torch.stack([torch.cat(rows), torch.cat(cols)])


# --------------------------------------------------------------------------
# decoding order and the autoregressive mask
# --------------------------------------------------------------------------
def dense_slots(batch: Tensor) -> Tuple[int, int, Tensor]:
    """to_dense_batch's layout (:27): node v -> (graph batch[v], position v - first node of the graph)."""
    counts = torch.bincount(batch)
    first = torch.cumsum(counts, 0) - counts
    return counts.numel(), int(counts.max()), torch.arange(batch.numel()) - first[batch]


def autoregressive_mask_einsum(order: Tensor, batch: Tensor, edge_index: Tensor) -> Tensor:
    """:32-49 with the decoding order given ([B, Lmax] over the padded rows): the O(L^3) einsum over one-hot
    permutation matrices, then the per-edge lookup order_mask_backward[graph, centre, neighbour]."""
    L = order.shape[1]
    perm = F.one_hot(order, num_classes=L).float()
    tri = 1 - torch.triu(torch.ones(L, L))
    omb = torch.einsum("ij, biq, bjp->bqp", tri, perm, perm)
    row, col = edge_index
    _, _, pos = dense_slots(batch)
    return omb[batch[row], pos[col], pos[row]].unsqueeze(-1)


def autoregressive_mask_rank(order: Tensor, batch: Tensor, edge_index: Tensor) -> Tensor:
    """The same mask as a rank comparison: the neighbour is visible iff it is decoded before the centre."""
    B, L = order.shape
    rank = torch.empty_like(order)
    rank.scatter_(1, order, torch.arange(L).expand(B, L))
    row, col = edge_index
    _, _, pos = dense_slots(batch)
    return (rank[batch[col], pos[col]] > rank[batch[row], pos[row]]).float().unsqueeze(-1)


def draw_decoding_order(chain_mask_all: Tensor, mask: Tensor, batch: Tensor, generator=None) -> Tensor:
    """:27-31: argsort of (chain_mask * mask + 1e-4) * |noise| over the padded batch (padding rows hold 0)."""
    B, L, pos = dense_slots(batch)
    key = torch.zeros(B, L)
    key[batch, pos] = (chain_mask_all * mask).float()
    return torch.argsort((key + 1e-4) * torch.abs(torch.randn(B, L, generator=generator)))


# --------------------------------------------------------------------------
# featurizer
# --------------------------------------------------------------------------
def _virtual_center(Ca, Cb, N):
    """:321-338."""
    u_bc = Cb - Ca
    u_bc = u_bc / (torch.norm(u_bc, dim=-1, keepdim=True) + 1e-6)
    u_nc = N - Ca
    u_nc = u_nc / (torch.norm(u_nc, dim=-1, keepdim=True) + 1e-6)
    n_plane = torch.cross(u_nc, u_bc, dim=-1)
    n_plane = n_plane / (torch.norm(n_plane, dim=-1, keepdim=True) + 1e-6)
    return Ca + 3.06 * (-torch.cross(n_plane, u_bc, dim=-1))


def _rbf(D: Tensor, num_rbf: int) -> Tensor:
    """:441-448."""
    mu = torch.linspace(2.0, 22.0, num_rbf).view(1, -1)
    sigma = (22.0 - 2.0) / num_rbf
    return torch.exp(-(((D.unsqueeze(-1) - mu) / sigma) ** 2))


def featurize(cfg: Config, x: Tensor, mask: Tensor, batch: Tensor):
    """:385-439 -> edge_index [2,E] (original node ids), RBF features [E, 16 A^2], occupancy [n,1] or None."""
    N, Ca, C, O = (x[:, i, :] for i in range(4))
    b, c = Ca - N, C - Ca
    a = torch.cross(b, c, dim=-1)
    Cb = -0.58273431 * a + 0.56802827 * b - 0.54067466 * c + Ca
    atoms = [N, Ca, C, O, Cb]
    V = None
    if cfg.use_virtual_center or cfg.occupancy_cutoff is not None:
        V = _virtual_center(Ca, Cb, N)
    if cfg.use_virtual_center:
        atoms.append(V)
    valid = mask.bool()
    ids = torch.nonzero(valid).flatten()
    if cfg.edge_cutoff is not None:
        e = radius_edges(Ca[valid], batch[valid], cfg.edge_cutoff, loop=True)
    else:
        e = knn_edges(Ca[valid], batch[valid], cfg.num_neighbors)
    edge_index = ids[e]
    row, col = edge_index

    occupancy = None
    if cfg.occupancy_cutoff is not None:  # :340-383
        oe = radius_edges(V[valid], batch[valid], cfg.occupancy_cutoff, loop=False)
        Vv = V[valid]
        d2 = ((Vv[oe[0]] - Vv[oe[1]]) ** 2).sum(-1)
        sigma = cfg.occupancy_cutoff / 3.0
        occ = torch.zeros(Vv.shape[0]).index_add_(0, oe[1], torch.exp(-d2 / (2 * sigma ** 2)))
        occupancy = torch.zeros(x.shape[0], 1)
        occupancy[valid] = torch.log1p(occ).unsqueeze(-1)

    feats = []
    for A, Bt in itertools.product(atoms, repeat=2):  # A at the neighbour (row), B at the centre (col), :434-437
        feats.append(_rbf(torch.sqrt(torch.sum((A[row] - Bt[col]) ** 2, 1) + 1e-6), cfg.num_rbf))
    return edge_index, torch.cat(feats, dim=-1), occupancy


# --------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------
def _ffn(sd, name, x):
    return _linear(sd, name + ".out.2", F.gelu(_linear(sd, name + ".out.0", x)))


def encoder_layer(sd: Dict[str, Tensor], name: str, cfg: Config, x: Tensor, edge_index: Tensor, e: Tensor):
    """:123-153 (dropout is the identity in eval mode)."""
    row, col = edge_index  # row = source j, col = target i
    msg = _mlp3(sd, name + ".out_v", torch.cat([x[col], x[row], e], dim=-1))
    agg = torch.zeros_like(x).index_add_(0, col, msg)
    x = _layer_norm(sd, name + ".norm1", x + agg / cfg.scale)
    x = _layer_norm(sd, name + ".norm2", x + _ffn(sd, name + ".dense", x))
    h = _mlp3(sd, name + ".out_e", torch.cat([x[row], x[col], e], dim=-1))
    e = _layer_norm(sd, name + ".norm3", e + h)
    return x, e


def decoder_layer(sd, name, cfg, x, edge_index, e, label, m_att):
    """:182-217."""
    row, col = edge_index
    h1 = torch.cat([x[row], e, label[row]], dim=-1)
    h0 = torch.cat([x[row], e, torch.zeros_like(label[row])], dim=-1)
    h = torch.cat([x[col], h1 * m_att + h0 * (1 - m_att)], dim=-1)
    agg = torch.zeros_like(x).index_add_(0, col, _mlp3(sd, name + ".out_v", h))
    x = _layer_norm(sd, name + ".norm1", x + agg / cfg.scale)
    return _layer_norm(sd, name + ".norm2", x + _ffn(sd, name + ".dense", x))


def forward(sd: Dict[str, Tensor], cfg: Config, x: Tensor, chain_seq_label: Tensor, mask: Tensor,
            chain_mask_all: Tensor, residue_idx: Tensor, chain_encoding_all: Tensor, batch: Tensor,
            decoding_order: Optional[Tensor] = None, return_graph: bool = False):
    """:450-515 in eval mode; decoding_order [B, Lmax] fixes what :30-31 would draw."""
    edge_index, rbf, occupancy = featurize(cfg, x, mask, batch)
    row, col = edge_index
    offset = residue_idx[row] - residue_idx[col]
    same = ((chain_encoding_all[row] - chain_encoding_all[col]) == 0).long()
    mrf = cfg.max_relative_feature
    d = torch.clip(offset + mrf, 0, 2 * mrf) * same + (1 - same) * (2 * mrf + 1)  # :82-84
    e_pos = sd["embedding.emb.weight"][d.long()]
    h = _linear(sd, "edge_mlp.0", torch.cat([rbf, e_pos], dim=-1))
    h_e = _linear(sd, "edge_mlp.2", _layer_norm(sd, "edge_mlp.1", h))

    h_v = torch.zeros(x.shape[0], cfg.hidden_dim)
    if occupancy is not None:
        o = F.gelu(_linear(sd, "occupancy_mlp.0", occupancy))
        h_v = h_v + _layer_norm(sd, "occupancy_mlp.3", _linear(sd, "occupancy_mlp.2", o))
    for l in range(cfg.num_encoder_layers):
        h_v, h_e = encoder_layer(sd, f"encoder_layers.{l}", cfg, h_v, edge_index, h_e)

    label = sd["label_embedding.weight"][chain_seq_label]
    if decoding_order is None:
        decoding_order = draw_decoding_order(chain_mask_all, mask, batch)
    m_att = autoregressive_mask_rank(decoding_order, batch, edge_index)
    for l in range(cfg.num_decoder_layers):
        h_v = decoder_layer(sd, f"decoder_layers.{l}", cfg, h_v, edge_index, h_e, label, m_att)
    lp = F.log_softmax(_linear(sd, "output", h_v), dim=-1)
    if return_graph:
        return lp, edge_index
    return lp
