"""Batch assembly for the hot path (SURVEY section 8f "next" N1).

The reference builds every batch on the host: `featurize` (src/mpnn/data/data_utils.py:51-183) pads numpy arrays
out of dict-of-chains entries, then copies seven tensors to the device; `StructureLoader`
(src/mpnn/data/protein_mpnn_dataset.py:60-87) groups entries under a padded-token budget.  At ~1 M residues/s per GPU
that host work is the bottleneck of the loop.  Here the entries are flattened ONCE into a `StructureShard` (flat
per-residue tensors + offsets) that lives in HBM, and a batch is one kernel launch (`mpnn_assemble_batch`) that writes
the same eight outputs `featurize` returns, in the dtypes the model takes.

`featurize` reshuffles the chain order of every entry on EVERY batch (data_utils.py:88), so `residue_idx` offsets and
chain numbers change from draw to draw.  `assemble(ids, shuffle_chains=True, rng=...)` does the same on the device
(`mpnn_assemble_batch_ordered`: the host draws one permutation of a handful of chain segments per structure, the kernel
lays the residues out in that order); build the shard with `shuffle=False` and pass the `random.Random` the reference
would have used to get its exact permutations.  Without `shuffle_chains` the order stored in the shard is kept.

Host code in this file is plumbing (numpy over Python dicts, executed once per dataset); the per-batch path is the HIP
kernel and has no CPU fallback.
"""
from __future__ import annotations

import random as _random

import numpy as np
import torch

from ._lib import call, ptr, stream_ptr

AA_ALPHABET = "ACDEFGHIKLMNPQRSTVWYX"  # src/mpnn/common/constants.py:5
_AA_INDEX = {a: i for i, a in enumerate(AA_ALPHABET)}


def flatten_entry(entry: dict, rng: _random.Random | None = None, shuffle: bool = True):
    """One dict-of-chains entry (keys `seq`, `masked_list`, `visible_list`, `seq_chain_<c>`, `coords_chain_<c>` with
    `N/CA/C/O_chain_<c>` arrays) -> flat per-residue arrays in the chain order `featurize` would use.

    Follows data_utils.py:69-88: a visible chain whose sequence equals a masked chain's is designed too; the chain
    order is shuffled (with `rng`, or the global `random` like the reference when `rng` is None).  The entry is not
    modified.  -> dict(X [L,4,3] f32 (NaN where atoms are missing), S [L] u8, chain [L] i32 (1-based, in final order),
    designed [L] u8)."""
    masked = list(entry["masked_list"])
    visible = list(entry["visible_list"])
    seq_of = {c: entry[f"seq_chain_{c}"] for c in masked + visible}
    # promotion in the reference's iteration order (masked chains outer, visible chains inner): it fixes the list the
    # shuffle starts from.  A letter listed in both counts as visible there (:76-79).
    initial_visible = [c for c in masked + visible if c in visible]
    initial_masked = [c for c in masked + visible if c in masked and c not in visible]
    for m in dict.fromkeys(initial_masked):
        for v in dict.fromkeys(initial_visible):
            if seq_of[m] == seq_of[v]:
                if v not in masked:
                    masked.append(v)
                if v in visible:
                    visible.remove(v)
    order = masked + visible
    if shuffle:
        (rng or _random).shuffle(order)
    designed_set = set(masked)
    xs, ss, cs, ds = [], [], [], []
    for number, c in enumerate(order, start=1):
        seq = seq_of[c]
        coords = entry[f"coords_chain_{c}"]
        x = np.stack([np.asarray(coords[f"{a}_chain_{c}"], dtype=np.float64) for a in ("N", "CA", "C", "O")], 1)
        if x.shape != (len(seq), 4, 3):
            raise ValueError(f"chain {c}: coordinates {x.shape} do not match a sequence of {len(seq)}")
        xs.append(x.astype(np.float32))
        ss.append(np.fromiter((_AA_INDEX[a] for a in seq), dtype=np.uint8, count=len(seq)))
        cs.append(np.full(len(seq), number, np.int32))
        ds.append(np.full(len(seq), 1 if c in designed_set else 0, np.uint8))
    return dict(X=np.concatenate(xs, 0), S=np.concatenate(ss, 0), chain=np.concatenate(cs, 0),
                designed=np.concatenate(ds, 0))


class StructureShard:
    """Flat per-residue tensors of many structures + offsets; `.to("cuda")` once, then `assemble(ids)` per batch."""

    FIELDS = ("X", "S", "chain", "designed", "offsets")

    def __init__(self, X, S, chain, designed, offsets):
        self.X, self.S, self.chain, self.designed, self.offsets = X, S, chain, designed, offsets
        off = offsets.cpu().numpy()
        self.lengths = np.diff(off).astype(np.int64)   # host copy: batch shapes are host decisions
        # chain segments of every structure (host copies, like the lengths): starts where the stored chain number changes
        ch = chain.cpu().numpy()
        cut = np.flatnonzero(np.diff(ch) != 0) + 1
        starts = np.unique(np.concatenate([off[:-1], cut]))
        starts = starts[starts < ch.shape[0]] if ch.shape[0] else starts[:0]
        self._seg_start = starts.astype(np.int64)
        self._seg_len = np.diff(np.concatenate([starts, [ch.shape[0]]])).astype(np.int32)
        self._seg_of = np.searchsorted(starts, off[:-1], side="left").astype(np.int64)   # first segment of a structure
        self._seg_cnt = np.diff(np.concatenate([self._seg_of, [starts.shape[0]]])).astype(np.int64)

    @classmethod
    def from_entries(cls, entries, rng: _random.Random | None = None, shuffle: bool = True):
        flat = [flatten_entry(e, rng, shuffle) for e in entries]
        offsets = np.zeros(len(flat) + 1, np.int64)
        np.cumsum([f["S"].shape[0] for f in flat], out=offsets[1:])
        cat = {k: np.concatenate([f[k] for f in flat], 0) for k in ("X", "S", "chain", "designed")}
        return cls(torch.from_numpy(cat["X"]), torch.from_numpy(cat["S"]), torch.from_numpy(cat["chain"]),
                   torch.from_numpy(cat["designed"]), torch.from_numpy(offsets))

    def __len__(self):
        return len(self.lengths)

    def to(self, device):
        return StructureShard(*(getattr(self, k).to(device).contiguous() for k in self.FIELDS))

    def state(self):
        """Plain tensors, for `torch.save` / `torch.load(..., weights_only=True)`."""
        return {k: getattr(self, k).cpu() for k in self.FIELDS}

    @classmethod
    def from_state(cls, state):
        return cls(*(state[k] for k in cls.FIELDS))

    def chain_orders(self, ids, rng=None):
        """One permutation of its chain segments per structure in `ids`, drawn like data_utils.py:88
        (`random.shuffle(all_chains)`; `rng`: a `random.Random`, default the global module like the reference)."""
        out = []
        for i in np.asarray(ids, dtype=np.int64).reshape(-1):
            order = list(range(int(self._seg_cnt[i])))
            (rng or _random).shuffle(order)
            out.append(order)
        return out

    def assemble(self, ids, with_mask_self: bool = False, shuffle_chains: bool = False, rng=None, chain_orders=None):
        """The eight outputs of `featurize` (data_utils.py:183) for the structures `ids`, padded to the longest:
        X [B,L,4,3] f32, S i64, mask f32, lengths i32 (device tensor), chain_M f32, residue_idx i64, mask_self f32
        [B,L,L] (None unless asked: it is O(L^2) and only the interface loss reads it), chain_encoding_all i64.
        `shuffle_chains=True` (or explicit `chain_orders`, one permutation of range(n_chains) per id) reorders the
        chains of every structure for THIS batch as `featurize` does (data_utils.py:88)."""
        dev = self.X.device
        if dev.type != "cuda":
            raise RuntimeError("StructureShard.assemble runs on the GPU: move the shard with .to('cuda') first "
                               "(mpnn_amd has no CPU fallback)")
        ids_host = np.asarray(ids, dtype=np.int64).reshape(-1)
        if ids_host.size == 0 or ids_host.min() < 0 or ids_host.max() >= len(self):
            raise IndexError("structure ids out of range")
        B, L = int(ids_host.size), int(self.lengths[ids_host].max())
        ids_dev = torch.from_numpy(ids_host.astype(np.int32)).to(dev, non_blocking=True)
        X = torch.empty(B, L, 4, 3, device=dev, dtype=torch.float32)
        S = torch.empty(B, L, device=dev, dtype=torch.int64)
        mask = torch.empty(B, L, device=dev, dtype=torch.float32)
        chain_M = torch.empty(B, L, device=dev, dtype=torch.float32)
        residue_idx = torch.empty(B, L, device=dev, dtype=torch.int64)
        chain_enc = torch.empty(B, L, device=dev, dtype=torch.int64)
        lengths = torch.empty(B, device=dev, dtype=torch.int32)
        mask_self = None
        if shuffle_chains and chain_orders is None:
            chain_orders = self.chain_orders(ids_host, rng)
        with torch.cuda.device(dev):  # launches go to the current device's stream
            st = stream_ptr()
            if chain_orders is not None:
                if len(chain_orders) != B:
                    raise ValueError("chain_orders needs one permutation per id")
                seg_ptr = np.zeros(B + 1, np.int32)
                src, ln = [], []
                for r, (i, order) in enumerate(zip(ids_host, chain_orders)):
                    order = np.asarray(order, dtype=np.int64)
                    if sorted(order.tolist()) != list(range(int(self._seg_cnt[i]))):
                        raise ValueError(f"chain order of structure {int(i)} is not a permutation of its "
                                         f"{int(self._seg_cnt[i])} chains")
                    k = self._seg_of[i] + order
                    src.append(self._seg_start[k])
                    ln.append(self._seg_len[k])
                    seg_ptr[r + 1] = seg_ptr[r] + order.shape[0]
                # one small pinned upload per batch: [seg_src (i64) | seg_len (i32) | seg_ptr (i32)]
                seg_src = torch.from_numpy(np.concatenate(src)).to(dev, non_blocking=True)
                seg_len = torch.from_numpy(np.concatenate(ln)).to(dev, non_blocking=True)
                seg_ptr_d = torch.from_numpy(seg_ptr).to(dev, non_blocking=True)
                call("mpnn_assemble_batch_ordered", ptr(self.X), ptr(self.S), ptr(self.designed), ptr(seg_ptr_d),
                     ptr(seg_src), ptr(seg_len), B, L, ptr(X), ptr(S), ptr(mask), ptr(chain_M), ptr(residue_idx),
                     ptr(chain_enc), ptr(lengths), st)
            else:
                call("mpnn_assemble_batch", ptr(self.X), ptr(self.S), ptr(self.chain), ptr(self.designed),
                     ptr(self.offsets), ptr(ids_dev), B, L, ptr(X), ptr(S), ptr(mask), ptr(chain_M), ptr(residue_idx),
                     ptr(chain_enc), ptr(lengths), st)
            if with_mask_self:
                mask_self = torch.empty(B, L, L, device=dev, dtype=torch.float32)
                call("mpnn_mask_self", ptr(chain_enc), B, L, ptr(mask_self), st)
        return X, S, mask, lengths, chain_M, residue_idx, mask_self, chain_enc


class TokenBudgetBatcher:
    """Batches of structure ids under a padded-token budget, as the reference's `StructureLoader`
    (protein_mpnn_dataset.py:60-87): ids sorted by length, a batch grows while `length * (count + 1) <= batch_size`,
    batches are visited in random order each epoch.

    `drop_overflow=True` reproduces the reference exactly, including that the entry which does not fit closes the
    batch and is itself left out (`batch = []`, :72-73); `False` starts the next batch with it instead."""

    def __init__(self, lengths, batch_size: int = 10000, shuffle: bool = True, drop_overflow: bool = True,
                 rng: np.random.Generator | None = None):
        lengths = np.asarray(lengths)
        self.shuffle, self.rng = shuffle, rng
        clusters, batch = [], []
        for ix in np.argsort(lengths):
            if lengths[ix] * (len(batch) + 1) <= batch_size:
                batch.append(int(ix))
            else:
                clusters.append(batch)
                batch = [] if drop_overflow or lengths[ix] > batch_size else [int(ix)]
        if batch:
            clusters.append(batch)
        self.clusters = [c for c in clusters if c] if not drop_overflow else clusters

    def __len__(self):
        return len(self.clusters)

    def __iter__(self):
        order = np.arange(len(self.clusters))
        if self.shuffle:
            (self.rng or np.random).shuffle(order)
        for i in order:
            yield self.clusters[i]
