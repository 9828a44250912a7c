"""Deterministic synthetic inputs and weights for the ProteinMPNN hot path.

Everything here is integer-hash based (numpy uint64 arithmetic), so the same
arrays are regenerated bit-for-bit on any machine: golden fixtures only need to
store the recipe (seed, shapes) plus the expected outputs.

Input contract produced here is the one `featurize` hands to the model
(reference: src/mpnn/data/data_utils.py:170-183): X f32 [B,L,4,3] (N, CA, C, O),
S i64 in [0,20], mask / chain_M f32 in {0,1}, residue_idx i64 with +100 jumps
between chains and -100 on padding, chain_encoding_all i64 numbered from 1 with
0 on padding.
"""

from __future__ import annotations

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def hash_u64(n: int, seed: int) -> np.ndarray:
    """splitmix64-style hash of (index, seed) -> uint64[n]."""
    with np.errstate(over="ignore"):
        x = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        x = x + np.uint64(seed & 0xFFFFFFFF) * np.uint64(0xD1B54A32D192ED03) + np.uint64(0x2545F4914F6CDD1D)
        x ^= x >> np.uint64(30)
        x = x * np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x = x * np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


def hash_uniform(n: int, seed: int) -> np.ndarray:
    """float32[n] uniform in [-1, 1), exactly reproducible (24-bit mantissa)."""
    u = (hash_u64(n, seed) >> np.uint64(40)).astype(np.float64)
    return (u / float(1 << 24) * 2.0 - 1.0).astype(np.float32)


def hash_normal(n: int, seed: int) -> np.ndarray:
    """float32[n] ~ approx N(0,1): sum of 4 hashed uniforms, rescaled (exactly reproducible)."""
    acc = np.zeros(n, dtype=np.float64)
    for k in range(4):
        acc += hash_uniform(n, seed * 4 + k + 1000003).astype(np.float64)
    return (acc * (3.0 / 4.0) ** 0.5).astype(np.float32)


def _name_seed(name: str) -> int:
    h = 2166136261
    for ch in name.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h


def fill_parameters_(module: torch.nn.Module, seed: int = 0) -> None:
    """Overwrite every parameter of `module` with hash-derived values.

    Matrices get a xavier-uniform-sized amplitude (reference init:
    model_utils.py:423-425), LayerNorm weights 1 +- 0.2, biases +-0.1.
    Works on any module that exposes named_parameters (reference or ours).
    """
    with torch.no_grad():
        for name, p in module.named_parameters():
            n = p.numel()
            u = hash_uniform(n, _name_seed(name) ^ (seed * 7919))
            if p.dim() > 1:
                fan_out, fan_in = p.shape[0], p.shape[1]
                amp = (6.0 / (fan_in + fan_out)) ** 0.5
                if "W_s" in name:  # embedding table
                    amp = 1.0
                v = u * np.float32(amp)
            elif "norm" in name and name.endswith("weight"):
                v = np.float32(1.0) + np.float32(0.2) * u
            else:
                v = np.float32(0.1) * u
            p.copy_(torch.from_numpy(v.reshape(tuple(p.shape))).to(p.device, p.dtype))


def synthetic_backbone(B: int, L: int, seed: int = 0, bond_jitter: float = 0.03) -> np.ndarray:
    """Random-walk C-alpha trace (3.8 +- bond_jitter A steps, mild compaction) with N/C/O placed
    at ideal bond lengths (1.46 / 1.52 / 1.23 A) in a per-residue random frame.
    Returns float32 [B, L, 4, 3] in atom order N, CA, C, O.

    The jitter (real CA-CA virtual bonds vary by about as much) keeps residues i-1 and i+1 from being
    EXACTLY equidistant from i: with exact 3.8 A steps every row of the k-NN holds a near-tie whose order flips
    with 1-ulp differences between sqrt implementations, and element-wise comparisons with the reference's
    CPU-generated vectors had to skip most rows.
    """
    g = hash_normal(B * L * 3, seed * 11 + 1).reshape(B, L, 3).astype(np.float64)
    steps = 3.8 + bond_jitter * hash_uniform(B * L, seed * 11 + 5).reshape(B, L).astype(np.float64)
    # persistent direction so the chain is locally extended but globally compact
    step = np.zeros_like(g)
    cur = g[:, 0] / (np.linalg.norm(g[:, 0], axis=-1, keepdims=True) + 1e-9)
    pos = np.zeros((B, 3))
    ca = np.zeros((B, L, 3))
    for i in range(L):
        cur = 0.6 * cur + 0.8 * g[:, i] - 0.002 * pos
        cur = cur / (np.linalg.norm(cur, axis=-1, keepdims=True) + 1e-9)
        pos = pos + steps[:, i:i + 1] * cur
        ca[:, i] = pos
        step[:, i] = cur
    f = hash_normal(B * L * 9, seed * 11 + 2).reshape(B, L, 3, 3).astype(np.float64)
    u1 = f[:, :, 0] / (np.linalg.norm(f[:, :, 0], axis=-1, keepdims=True) + 1e-9)
    v = f[:, :, 1] - (f[:, :, 1] * u1).sum(-1, keepdims=True) * u1
    u2 = v / (np.linalg.norm(v, axis=-1, keepdims=True) + 1e-9)
    u3 = np.cross(u1, u2)
    n_at = ca - 1.46 * u1
    ang = np.deg2rad(111.0)
    c_dir = -np.cos(ang) * (-u1) + np.sin(ang) * u2  # N-CA-C angle 111 deg
    c_at = ca + 1.52 * c_dir
    o_dir = 0.5 * c_dir + 0.6 * u3 + 0.3 * u2
    o_dir = o_dir / (np.linalg.norm(o_dir, axis=-1, keepdims=True) + 1e-9)
    o_at = c_at + 1.23 * o_dir
    X = np.stack([n_at, ca, c_at, o_at], axis=2)
    return X.astype(np.float32)


def synthetic_batch(
    B: int,
    L: int,
    seed: int = 0,
    lengths: list[int] | None = None,
    n_chains: int = 1,
    design_fraction: float = 1.0,
    device: str | torch.device = "cpu",
) -> dict[str, torch.Tensor]:
    """A padded batch obeying the featurize() output contract.

    lengths: valid length per protein (default all L). n_chains: chains per
    protein (equal split, residue_idx jumps by +100 between chains, cf.
    data_utils.py:57,120). design_fraction: fraction of chains with chain_M=1.
    """
    X = synthetic_backbone(B, L, seed)
    lengths = list(lengths) if lengths is not None else [L] * B
    assert len(lengths) == B and max(lengths) <= L
    S = (hash_u64(B * L, seed * 11 + 3) % np.uint64(21)).astype(np.int64).reshape(B, L)
    mask = np.zeros((B, L), np.float32)
    chain_M = np.zeros((B, L), np.float32)
    residue_idx = np.full((B, L), -100, np.int64)
    chain_enc = np.zeros((B, L), np.int64)
    n_design = max(1, int(round(design_fraction * n_chains)))
    for b, n in enumerate(lengths):
        mask[b, :n] = 1.0
        bounds = [round(c * n / n_chains) for c in range(n_chains + 1)]
        for c in range(n_chains):
            lo, hi = bounds[c], bounds[c + 1]
            residue_idx[b, lo:hi] = np.arange(lo, hi) + 100 * c
            chain_enc[b, lo:hi] = c + 1
            if c < n_design:
                chain_M[b, lo:hi] = 1.0
        X[b, n:] = 0.0
        S[b, n:] = 0
    # decoding order: a fixed permutation per protein
    key = hash_uniform(B * L, seed * 11 + 4).reshape(B, L)
    order = np.argsort(key, axis=-1, kind="stable").astype(np.int64)
    out = dict(X=X, S=S, mask=mask, chain_M=chain_M, residue_idx=residue_idx,
               chain_encoding_all=chain_enc, fix_order=order)
    return {k: torch.from_numpy(v).to(device) for k, v in out.items()}


def synthetic_entries(n: int, seed: int = 0, min_len: int = 40, max_len: int = 400, max_chains: int = 4,
                      missing_fraction: float = 0.02) -> list[dict]:
    """Dict-of-chains entries in the format the reference's `featurize` consumes (data_utils.py:51-183; produced there
    by the PDB parsers): `seq`, `masked_list`, `visible_list`, `seq_chain_<c>`, `coords_chain_<c>` = {`N/CA/C/O_chain_<c>`:
    [len,3] arrays with NaN rows for unresolved atoms}.  Lengths ~ log-normal (median ~ 250, clipped), 1..max_chains
    chains; every third entry is a homo-oligomer (a visible chain repeating a designed chain's sequence)."""
    alphabet = "ACDEFGHIKLMNPQRSTVWYX"
    entries = []
    for e in range(n):
        s0 = seed * 7919 + e * 104729
        u = 0.5 * (hash_uniform(8, s0) + 1.0)  # [0, 1)
        total = int(np.clip(np.exp(np.log(250.0) + 0.6 * hash_normal(1, s0 + 1)[0]), min_len, max_len))
        n_chains = 1 + int(u[0] * max_chains) % max_chains
        cuts = np.sort((0.5 * (hash_uniform(n_chains - 1, s0 + 2) + 1.0) * (total - 8 * n_chains)).astype(np.int64)) if n_chains > 1 \
            else np.zeros(0, np.int64)
        bounds = np.concatenate([[0], cuts + 8 * np.arange(1, n_chains), [total]])
        chain_len = np.diff(bounds).astype(np.int64)
        homo = e % 3 == 2 and n_chains >= 2
        if homo:  # homo-oligomer: chain B is a copy of chain A's sequence (featurize compares whole chain sequences)
            chain_len[1] = chain_len[0]
        total = int(chain_len.sum())
        bounds = np.concatenate([[0], np.cumsum(chain_len)])
        X = synthetic_backbone(1, total, seed=s0 % 100003)[0].astype(np.float64)
        miss = 0.5 * (hash_uniform(total * 4, s0 + 3).reshape(total, 4) + 1.0) < missing_fraction
        X[miss] = np.nan
        letters = [chr(ord("A") + c) for c in range(n_chains)]
        entry = {"name": f"synthetic_{seed}_{e}"}
        seqs = []
        for c, letter in enumerate(letters):
            lo, hi = int(bounds[c]), int(bounds[c + 1])
            idx = (hash_u64(hi - lo, s0 + 10 + c) % np.uint64(21)).astype(np.int64)
            seq = seqs[0] if (homo and c == 1) else "".join(alphabet[i] for i in idx)
            seqs.append(seq)
            entry[f"seq_chain_{letter}"] = seq
            entry[f"coords_chain_{letter}"] = {f"{a}_chain_{letter}": X[lo:hi, k] for k, a in enumerate(("N", "CA", "C", "O"))}
        n_masked = max(1, int(round(u[1] * n_chains)))
        entry["masked_list"] = letters[:n_masked]
        entry["visible_list"] = letters[n_masked:]
        entry["seq"] = "".join(seqs)
        entries.append(entry)
    return entries


def synthetic_flat_batch(lengths: list[int], seed: int = 0, n_chains: int = 1, design_fraction: float = 1.0,
                         unresolved_every: int = 0, device: str | torch.device = "cpu") -> dict[str, torch.Tensor]:
    """The same synthetic proteins in the flat layout of the sparse model (mpnn_amd.sparse.ProteinMPNN; reference
    src/mpnn/models/protein_mpnn.py:450-459): x [n,4,3], chain_seq_label, mask, chain_mask_all, residue_idx,
    chain_encoding_all, batch [n], plus decoding_order [B, max(lengths)] over the padded rows.
    unresolved_every = m > 0 marks every m-th residue unresolved (mask 0, coordinates NaN: nothing may read them)."""
    B, L = len(lengths), max(lengths)
    d = synthetic_batch(B, L, seed=seed, lengths=lengths, n_chains=n_chains, design_fraction=design_fraction)
    keep = d["mask"] > 0
    batch = torch.arange(B)[:, None].expand(B, L)[keep]
    out = dict(x=d["X"][keep], chain_seq_label=d["S"][keep], mask=d["mask"][keep], chain_mask_all=d["chain_M"][keep],
               residue_idx=d["residue_idx"][keep], chain_encoding_all=d["chain_encoding_all"][keep], batch=batch,
               decoding_order=d["fix_order"])
    if unresolved_every > 0:
        gone = (torch.arange(batch.numel()) % unresolved_every) == unresolved_every - 1
        out["mask"] = torch.where(gone, torch.zeros(()), out["mask"])
        out["x"] = torch.where(gone[:, None, None], torch.full((), float("nan")), out["x"])
    return {k: v.to(device) for k, v in out.items()}
