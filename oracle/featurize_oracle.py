"""CPU oracle for the host batch assembly around the hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

numpy restatement of the reference's batch featurizer `src/mpnn/data/data_utils.py:51-183` (`featurize`) and of its
token-budget loader `src/mpnn/data/protein_mpnn_dataset.py:60-87` (`StructureLoader`).  All `:line` citations are to
those two files.  Only `tests/` and `bench.py`'s CPU-baseline timing may import this module.

PARITY UNPINNED: `data_utils.py` imports `Bio` and `torch_geometric` at module level (`:9,12`), neither of which is
installed here, so the reference function cannot be run to generate golden vectors and the reference's tests hold no
fixture for it.  This file is a line-by-line restatement only; the GPU path (`mpnn_amd.data`) is checked against it
bit for bit, which is as far as parity for this row can be claimed.

The one deliberate difference: the reference shuffles the chain order with the global `random` module (`:88`); here the
caller passes the generator (or `shuffle=False`), so that both sides of a test can use the same order.
"""
from __future__ import annotations

import random as _random

import numpy as np

AA_ALPHABET = "ACDEFGHIKLMNPQRSTVWYX"  # src/mpnn/common/constants.py:5


def chain_order(entry: dict, rng: _random.Random | None = None, shuffle: bool = True):
    """`:69-88`: chains whose sequence equals a masked chain's become masked too; then the order is shuffled.
    -> (all_chains in final order, set of masked chain letters).  Does not modify `entry`."""
    masked = list(entry["masked_list"])
    visible = list(entry["visible_list"])
    all_chains = masked + visible
    visible_seq, masked_seq = {}, {}
    for letter in all_chains:
        seq = entry[f"seq_chain_{letter}"]
        if letter in visible:
            visible_seq[letter] = seq
        elif letter in masked:
            masked_seq[letter] = seq
    for vm in masked_seq.values():
        for kv, vv in visible_seq.items():
            if vm == vv:
                if kv not in masked:
                    masked.append(kv)
                if kv in visible:
                    visible.remove(kv)
    all_chains = masked + visible
    if shuffle:
        (rng or _random).shuffle(all_chains)
    return all_chains, set(masked)


def featurize(batch, orders=None, rng: _random.Random | None = None, shuffle: bool = True):
    """`:51-183` on numpy arrays.  `orders`: optional per-entry (all_chains, masked set) from `chain_order`.
    -> X [B,L,4,3] f32, S i64, mask f32, lengths i32, chain_M f32, residue_idx i64, mask_self f32 [B,L,L],
       chain_encoding_all i64  (the dtypes after the reference's final conversion, `:175-181`)."""
    B = len(batch)
    lengths = np.array([len(b["seq"]) for b in batch], dtype=np.int32)                       # :54
    L_max = max(len(b["seq"]) for b in batch)                                                # :55
    X = np.zeros([B, L_max, 4, 3])                                                           # :56 (float64)
    residue_idx = -100 * np.ones([B, L_max], dtype=np.int32)                                 # :57
    chain_M = np.zeros([B, L_max], dtype=np.int32)                                           # :58
    mask_self = np.ones([B, L_max, L_max], dtype=np.int32)                                   # :61
    chain_encoding_all = np.zeros([B, L_max], dtype=np.int32)                                # :64
    S = np.zeros([B, L_max], dtype=np.int32)                                                 # :67
    for i, b in enumerate(batch):
        all_chains, masked = orders[i] if orders is not None else chain_order(b, rng, shuffle)
        xs, ms, seqs, encs = [], [], [], []
        c, l0, l1 = 1, 0, 0
        for letter in all_chains:                                                            # :96-148
            seq = b[f"seq_chain_{letter}"]
            n = len(seq)
            coords = b[f"coords_chain_{letter}"]
            x_chain = np.stack([np.asarray(coords[f"{a}_chain_{letter}"]) for a in ("N", "CA", "C", "O")], 1)
            xs.append(x_chain)
            ms.append(np.ones(n) if letter in masked else np.zeros(n))
            seqs.append(seq)
            encs.append(c * np.ones(n))
            l1 += n
            mask_self[i, l0:l1, l0:l1] = 0
            residue_idx[i, l0:l1] = 100 * (c - 1) + np.arange(l0, l1)
            l0 += n
            c += 1
        x = np.concatenate(xs, 0)
        all_sequence = "".join(seqs)
        l = len(all_sequence)
        X[i] = np.pad(x, [[0, L_max - l], [0, 0], [0, 0]], "constant", constant_values=(np.nan,))   # :155
        chain_M[i] = np.pad(np.concatenate(ms, 0), [[0, L_max - l]], "constant", constant_values=(0.0,))
        chain_encoding_all[i] = np.pad(np.concatenate(encs, 0), [[0, L_max - l]], "constant", constant_values=(0.0,))
        S[i, :l] = np.asarray([AA_ALPHABET.index(a) for a in all_sequence], dtype=np.int32)  # :167
    isnan = np.isnan(X)                                                                      # :170
    mask = np.isfinite(np.sum(X, (2, 3)))                                                    # :171
    X[isnan] = 0.0                                                                           # :172
    return (X.astype(np.float32), S.astype(np.int64), mask.astype(np.float32), lengths, chain_M.astype(np.float32),
            residue_idx.astype(np.int64), mask_self.astype(np.float32), chain_encoding_all.astype(np.int64))


def structure_loader_clusters(lengths, batch_size=100):
    """`protein_mpnn_dataset.py:60-76`: length-sorted greedy batches under a padded-token budget.  As in the
    reference, the entry that does not fit closes the batch and is itself DROPPED (`batch = []` at `:72-73`)."""
    sorted_ix = np.argsort(lengths)
    clusters, batch = [], []
    for ix in sorted_ix:
        size = lengths[ix]
        if size * (len(batch) + 1) <= batch_size:
            batch.append(int(ix))
        else:
            clusters.append(batch)
            batch = []
    if len(batch) > 0:
        clusters.append(batch)
    return clusters
