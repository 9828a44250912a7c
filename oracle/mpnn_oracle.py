"""CPU oracle for the ProteinMPNN hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

A plain-PyTorch (CPU, fp32) restatement of the algorithm in the reference file
`src/mpnn/models/model_utils.py` (all `:line` citations below are to that
file).  It is written functionally over a flat `state_dict` so it shares no
code with either the reference modules or the product package `mpnn_amd`.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import this module, and only as the checker / the timed CPU baseline.  The
product path (`mpnn_amd`) never imports it and has no CPU fallback.

Pinning: the reference's own test-suite holds no golden tensors for the dense
model, so this oracle is pinned against outputs of the
reference itself, generated in the build container by
`tests/golden/make_golden.py` (imports the reference file by path) and
committed as `tests/golden/*.npz`; `tests/test_oracle_golden.py` replays them.
The virtual-centre geometry is additionally pinned by the reference's
known-answer test `tests/test_virtual_center.py:100-118`.

Two formulations are provided where the reference's is asymptotically
wasteful:
  * `formulation="reference"`: materialises the same [B,L,L] temporaries and
    the O(L^3) permutation einsum the reference does (:270-289, :301-307,
    :492-506).  This is what `cpu_baseline` times.
  * `formulation="edge"`: per-edge distances and rank comparison
    (O(L*K)); bit-identical results, used to keep big parity cases fast.
"""

from __future__ import annotations

import itertools
from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass
class Config:
    """Constructor arguments that change the maths (:366-381)."""

    k_neighbors: int = 48
    num_encoder_layers: int = 3
    num_decoder_layers: int = 3
    hidden_dim: int = 128
    edge_cutoff: Optional[float] = None
    use_virtual_center: bool = False
    occupancy_cutoff: Optional[float] = None
    num_rbf: int = 16
    max_relative_feature: int = 32
    scale: float = 30.0  # EncLayer/DecLayer scale (:51, :101)


# --------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------
def _linear(sd: Dict[str, Tensor], name: str, x: Tensor) -> Tensor:
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _layer_norm(sd: Dict[str, Tensor], name: str, x: Tensor) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], 1e-5)


def _gelu(x: Tensor) -> Tensor:
    return F.gelu(x)  # exact erf form: nn.GELU() default (:70, :115, :149)


def _sqrt(x: Tensor) -> Tensor:
    """IEEE-754 correctly rounded fp32 square root (computed in fp64, rounded once).

    torch's CPU sqrt kernel is NOT correctly rounded (measured in the build container, torch
    2.10 CPU, every ATEN_CPU_CAPABILITY: 0.67 % of fp32 inputs come out 1 ulp off), whereas
    the reference's production device path (CUDA sqrtf) and this repo's HIP kernels are.  The
    oracle follows IEEE so that "bit-exact neighbour indices" is platform independent; against
    the CPU-generated golden vectors distances therefore agree to 1 ulp and indices agree
    wherever two candidates are more than 2 ulp apart (tests/test_oracle_golden.py)."""
    return torch.sqrt(x.double()).to(x.dtype)


def take_neighbors(nodes: Tensor, E_idx: Tensor) -> Tensor:
    """nodes [B,L,C], E_idx [B,L,K] -> [B,L,K,C]   (gather_nodes, :21-30)."""
    B, L, K = E_idx.shape
    flat = E_idx.reshape(B, L * K, 1).expand(-1, -1, nodes.shape[-1])
    return torch.gather(nodes, 1, flat).reshape(B, L, K, nodes.shape[-1])


def take_pairs(pair: Tensor, E_idx: Tensor) -> Tensor:
    """pair [B,L,L], E_idx [B,L,K] -> [B,L,K]   (gather_edges, :13-18)."""
    return torch.gather(pair, 2, E_idx)


# --------------------------------------------------------------------------
# featurizer
# --------------------------------------------------------------------------
def backbone_atoms(X: Tensor, cfg: Config) -> list:
    """[N, CA, C, O, CB(, V)] each [B,L,3]   (:317-330)."""
    n, ca, c, o = X[:, :, 0], X[:, :, 1], X[:, :, 2], X[:, :, 3]
    b = ca - n
    cc = c - ca
    a = torch.cross(b, cc, dim=-1)
    cb = -0.58273431 * a + 0.56802827 * b - 0.54067466 * cc + ca
    atoms = [n, ca, c, o, cb]
    if cfg.use_virtual_center:
        atoms.append(virtual_center(ca, cb, n))
    return atoms


def virtual_center(ca: Tensor, cb: Tensor, n: Tensor) -> Tensor:
    """V = CA - 3.06 * ((unit(N-CA) x unit(CB-CA))_unit x unit(CB-CA))   (:215-230)."""

    def unit(v):
        return v / (torch.norm(v, dim=-1, keepdim=True) + 1e-6)

    u_bc = unit(cb - ca)
    u_nc = unit(n - ca)
    plane = unit(torch.cross(u_nc, u_bc, dim=-1))
    return ca + 3.06 * (-torch.cross(plane, u_bc, dim=-1))


def adjusted_distances(ca: Tensor, mask: Tensor, cfg: Config) -> Tensor:
    """D_adjust [B,L,L] exactly as `_dist` builds it   (:270-279)."""
    m2 = mask[:, None, :] * mask[:, :, None]
    dx = ca[:, None, :, :] - ca[:, :, None, :]
    D = m2 * _sqrt(torch.sum(dx**2, 3) + 1e-6)
    D_max, _ = torch.max(D, -1, keepdim=True)
    D_adj = D + (1.0 - m2) * D_max
    if cfg.edge_cutoff is not None:
        within = (cfg.edge_cutoff > D).float()
        D_adj = D_adj + (1.0 - within) * 1e6
    return D_adj


def knn(ca: Tensor, mask: Tensor, cfg: Config, tie_break: str = "index", E_idx_given: Optional[Tensor] = None):
    """K' = min(k, L) nearest by D_adjust, ascending   (:281-289).

    tie_break="index": equal distances are ordered by ascending column index
    (stable sort).  torch.topk's order among EQUAL values is implementation
    defined, so "index" is the documented convention of the
    product kernels; for untied entries it equals tie_break="topk", which calls
    torch.topk exactly like the reference.
    """
    D_adj = adjusted_distances(ca, mask, cfg)
    K = min(cfg.k_neighbors, ca.shape[1])
    if E_idx_given is not None:  # tests only: adopt a given (e.g. the reference's) resolution of ties
        E_idx = E_idx_given
        D_nb = torch.gather(D_adj, 2, E_idx)
    elif tie_break == "topk":
        D_nb, E_idx = torch.topk(D_adj, K, dim=-1, largest=False)
    else:
        D_sorted, order = torch.sort(D_adj, dim=-1, stable=True)
        D_nb, E_idx = D_sorted[..., :K].contiguous(), order[..., :K].contiguous()
    edge_mask = None
    if cfg.edge_cutoff is not None:
        edge_mask = (D_nb < cfg.edge_cutoff).float()
    return D_nb, E_idx, edge_mask


def occupancy(V: Tensor, mask: Tensor, cutoff: float) -> Tensor:
    """log1p(sum_{j != i, valid, d2 < r2} exp(-d2 / (2 (r/3)^2)))   (:232-268)."""
    diff = V.unsqueeze(2) - V.unsqueeze(1)
    d_sq = torch.sum(diff**2, dim=-1)
    valid = mask.unsqueeze(1) * mask.unsqueeze(2) * (1.0 - torch.eye(V.size(1)).unsqueeze(0))
    within = (d_sq < (cutoff**2)).float()
    sigma = cutoff / 3.0
    occ = torch.exp(-d_sq / (2 * sigma**2)) * valid * within
    return torch.log1p(torch.sum(occ, dim=2, keepdim=True))


def rbf_expand(D: Tensor, num_rbf: int = 16) -> Tensor:
    """16 Gaussians, centres linspace(2,22,16), sigma 1.25   (:291-299)."""
    mu = torch.linspace(2.0, 22.0, num_rbf).view(*([1] * D.dim()), -1)
    sigma = (22.0 - 2.0) / num_rbf
    return torch.exp(-(((D.unsqueeze(-1) - mu) / sigma) ** 2))


def pair_distance_at_edges(A: Tensor, Bt: Tensor, E_idx: Tensor, formulation: str) -> Tensor:
    """sqrt(|A_i - B_j|^2 + 1e-6) at j = E_idx[i,k]   (:301-305)."""
    if formulation == "reference":
        D = _sqrt(torch.sum((A[:, :, None, :] - Bt[:, None, :, :]) ** 2, -1) + 1e-6)
        return take_pairs(D, E_idx)
    Bj = take_neighbors(Bt, E_idx)
    return _sqrt(torch.sum((A[:, :, None, :] - Bj) ** 2, -1) + 1e-6)


def positional_codes(residue_idx: Tensor, chain_labels: Tensor, E_idx: Tensor, cfg: Config) -> Tensor:
    """Bucket index d in [0,65]: clip(ri - rj + 32, 0, 64) on the same chain, 65 otherwise
    (:349-356, :170-172)."""
    offset = take_pairs(residue_idx[:, :, None] - residue_idx[:, None, :], E_idx)
    same = take_pairs(((chain_labels[:, :, None] - chain_labels[:, None, :]) == 0).long(), E_idx)
    m = cfg.max_relative_feature
    return torch.clip(offset + m, 0, 2 * m) * same + (1 - same) * (2 * m + 1)


def features(sd, cfg: Config, X, mask, residue_idx, chain_labels, backbone_noise=None,
             formulation: str = "reference", tie_break: str = "index", E_idx_given=None):
    """ProteinFeatures.forward (:309-360).  Returns dict with E, E_idx, occupancy,
    edge_mask, D_neighbors."""
    if backbone_noise is not None:
        X = X + backbone_noise
    atoms = backbone_atoms(X, cfg)
    ca, cb, n = atoms[1], atoms[4], atoms[0]
    D_nb, E_idx, edge_mask = knn(ca, mask, cfg, tie_break, E_idx_given)
    occ = None
    if cfg.occupancy_cutoff is not None:
        V = atoms[-1] if cfg.use_virtual_center else virtual_center(ca, cb, n)
        occ = occupancy(V, mask, cfg.occupancy_cutoff)
    rbfs = [
        rbf_expand(pair_distance_at_edges(a1, a2, E_idx, formulation), cfg.num_rbf)
        for a1, a2 in itertools.product(atoms, repeat=2)
    ]
    d = positional_codes(residue_idx, chain_labels, E_idx, cfg)
    onehot = F.one_hot(d, 2 * cfg.max_relative_feature + 2).to(torch.float32)
    E_pos = _linear(sd, "features.embeddings.linear", onehot)
    E = torch.cat([E_pos] + rbfs, -1)
    E = _linear(sd, "features.edge_embedding", E)
    E = _layer_norm(sd, "features.norm_edges", E)
    return dict(E=E, E_idx=E_idx, occupancy=occ, edge_mask=edge_mask, D_neighbors=D_nb)


# --------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------
def _ffn(sd, prefix: str, x: Tensor) -> Tensor:
    """PositionWiseFeedForward (:151-155)."""
    return _linear(sd, prefix + ".W_out", _gelu(_linear(sd, prefix + ".W_in", x)))


def _mlp3(sd, prefix: str, names, x: Tensor) -> Tensor:
    a, b, c = names
    return _linear(sd, f"{prefix}.{c}", _gelu(_linear(sd, f"{prefix}.{b}", _gelu(_linear(sd, f"{prefix}.{a}", x)))))


def enc_layer(sd, prefix: str, cfg: Config, h_V, h_E, E_idx, mask_V=None, mask_attend=None):
    """EncLayer.forward in eval mode (dropout = identity)   (:73-95)."""
    K = E_idx.shape[-1]
    h_EV = torch.cat([h_V[:, :, None, :].expand(-1, -1, K, -1), h_E, take_neighbors(h_V, E_idx)], -1)
    msg = _mlp3(sd, prefix, ("W1", "W2", "W3"), h_EV)
    if mask_attend is not None:
        msg = mask_attend.unsqueeze(-1) * msg
    h_V = _layer_norm(sd, prefix + ".norm1", h_V + torch.sum(msg, -2) / cfg.scale)
    h_V = _layer_norm(sd, prefix + ".norm2", h_V + _ffn(sd, prefix + ".dense", h_V))
    if mask_V is not None:
        h_V = mask_V.unsqueeze(-1) * h_V
    h_EV = torch.cat([h_V[:, :, None, :].expand(-1, -1, K, -1), h_E, take_neighbors(h_V, E_idx)], -1)
    h_E = _layer_norm(sd, prefix + ".norm3", h_E + _mlp3(sd, prefix, ("W11", "W12", "W13"), h_EV))
    return h_V, h_E


def dec_layer(sd, prefix: str, cfg: Config, h_V, h_ESV, mask_V=None, mask_attend=None):
    """DecLayer.forward in eval mode   (:118-138)."""
    K = h_ESV.shape[-2]
    h_EV = torch.cat([h_V[:, :, None, :].expand(-1, -1, K, -1), h_ESV], -1)
    msg = _mlp3(sd, prefix, ("W1", "W2", "W3"), h_EV)
    if mask_attend is not None:
        msg = mask_attend.unsqueeze(-1) * msg
    h_V = _layer_norm(sd, prefix + ".norm1", h_V + torch.sum(msg, -2) / cfg.scale)
    h_V = _layer_norm(sd, prefix + ".norm2", h_V + _ffn(sd, prefix + ".dense", h_V))
    if mask_V is not None:
        h_V = mask_V.unsqueeze(-1) * h_V
    return h_V


# --------------------------------------------------------------------------
# causal decoding-order mask
# --------------------------------------------------------------------------
def decoding_order(chain_M: Tensor, mask: Tensor, noise: Tensor) -> Tensor:
    """argsort((chain_M*mask + 1e-4) * |noise|)   (:485-491); `noise` replaces randn."""
    return torch.argsort((chain_M * mask + 0.0001) * torch.abs(noise))


def order_mask_backward(order: Tensor, E_idx: Tensor, formulation: str = "reference") -> Tensor:
    """[B,L,K] float: 1 where neighbour E_idx[b,i,k] is decoded strictly before i   (:492-507)."""
    B, L = order.shape
    if formulation == "reference":
        P = F.one_hot(order, num_classes=L).to(torch.float32)
        tri = 1 - torch.triu(torch.ones(L, L))
        M = torch.einsum("ij, biq, bjp->bqp", tri, P, P)
        return torch.gather(M, 2, E_idx)
    rank = torch.empty_like(order)
    rank.scatter_(1, order, torch.arange(L).expand(B, -1))
    return (rank[:, :, None] > torch.gather(rank, 1, E_idx.reshape(B, -1)).reshape(E_idx.shape)).float()


# --------------------------------------------------------------------------
# full model
# --------------------------------------------------------------------------
def forward(sd, cfg: Config, X, S, mask, chain_M, residue_idx, chain_encoding_all,
            fix_order=None, fix_backbone_noise=None, order_noise=None,
            formulation: str = "reference", tie_break: str = "index", keep: bool = False,
            E_idx_given=None):
    """ProteinMPNN.forward in eval mode (:427-529).  Returns log_probs, or a dict of
    intermediates when keep=True.  `order_noise` stands in for the reference's internal
    torch.randn draw when fix_order is None."""
    out = {}
    f = features(sd, cfg, X, mask, residue_idx, chain_encoding_all, fix_backbone_noise,
                 formulation, tie_break, E_idx_given)
    E, E_idx, occ, edge_mask = f["E"], f["E_idx"], f["occupancy"], f["edge_mask"]
    h_V = torch.zeros(E.shape[0], E.shape[1], E.shape[-1])
    if cfg.occupancy_cutoff is not None and occ is not None:
        o = _linear(sd, "occupancy_mlp.0", occ)
        o = _linear(sd, "occupancy_mlp.2", _gelu(o))
        h_V = h_V + _layer_norm(sd, "occupancy_mlp.3", o)
    h_E = _linear(sd, "W_e", E)
    mask_attend = mask.unsqueeze(-1) * take_neighbors(mask.unsqueeze(-1), E_idx).squeeze(-1)
    if edge_mask is not None:
        mask_attend = mask_attend * edge_mask
    if keep:
        out.update(f, h_E0=h_E, mask_attend=mask_attend, h_V_enc=[], h_E_enc=[], h_V_dec=[])
    for l in range(cfg.num_encoder_layers):
        h_V, h_E = enc_layer(sd, f"encoder_layers.{l}", cfg, h_V, h_E, E_idx, mask, mask_attend)
        if keep:
            out["h_V_enc"].append(h_V)
            out["h_E_enc"].append(h_E)

    h_S = F.embedding(S, sd["W_s.weight"])
    h_ES = torch.cat([h_E, take_neighbors(h_S, E_idx)], -1)
    h_EXV_enc = torch.cat([h_E, torch.zeros_like(take_neighbors(h_S, E_idx)), take_neighbors(h_V, E_idx)], -1)

    chain_M = chain_M * mask
    if fix_order is not None:
        order = fix_order
    else:
        order = decoding_order(chain_M, torch.ones_like(mask), order_noise)
    bw = order_mask_backward(order, E_idx, formulation).unsqueeze(-1)
    m1 = mask.view(mask.size(0), mask.size(1), 1, 1)
    mask_bw = m1 * bw
    mask_fw = m1 * (1.0 - bw)
    if edge_mask is not None:
        mask_bw = mask_bw * edge_mask.unsqueeze(-1)
        mask_fw = mask_fw * edge_mask.unsqueeze(-1)
    h_EXV_enc_fw = mask_fw * h_EXV_enc
    for l in range(cfg.num_decoder_layers):
        h_ESV = torch.cat([h_ES, take_neighbors(h_V, E_idx)], -1)
        h_ESV = mask_bw * h_ESV + h_EXV_enc_fw
        h_V = dec_layer(sd, f"decoder_layers.{l}", cfg, h_V, h_ESV, mask)
        if keep:
            out["h_V_dec"].append(h_V)
    logits = _linear(sd, "W_out", h_V)
    log_probs = F.log_softmax(logits, dim=-1)
    if keep:
        out.update(mask_bw=mask_bw.squeeze(-1), mask_fw=mask_fw.squeeze(-1), order=order,
                   log_probs=log_probs)
        return out
    return log_probs


# --------------------------------------------------------------------------
# training semantics (reference: src/mpnn/train.py:63-73)
# --------------------------------------------------------------------------
def loss_smoothed(S: Tensor, log_probs: Tensor, mask: Tensor, weight: float = 0.1):
    """Label-smoothed NLL, summed and divided by the constant 2000 (train.py:63-73)."""
    S1 = F.one_hot(S, 21).float()
    S1 = S1 + weight / float(S1.size(-1))
    S1 = S1 / S1.sum(-1, keepdim=True)
    loss = -(S1 * log_probs).sum(-1)
    return loss, torch.sum(loss * mask) / 2000.0
