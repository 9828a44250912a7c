"""Host side of the batch assembly (mpnn_amd.data) against the numpy restatement of the reference's featurize /
StructureLoader (oracle/featurize_oracle.py; parity unpinned, see its header)."""
import io
import random

import numpy as np
import torch

from mpnn_amd.data import StructureShard, TokenBudgetBatcher, flatten_entry
from mpnn_amd.synthetic import synthetic_entries
from oracle import featurize_oracle as fo


def test_flatten_entry_matches_featurize_chain_logic():
    entries = synthetic_entries(18, seed=3)
    assert any(len(e["masked_list"]) + len(e["visible_list"]) > 1 for e in entries)
    promoted = 0
    for i, e in enumerate(entries):
        before = (list(e["masked_list"]), list(e["visible_list"]))
        flat = flatten_entry(e, random.Random(100 + i))
        order, masked = fo.chain_order(e, random.Random(100 + i))
        assert (list(e["masked_list"]), list(e["visible_list"])) == before, "entry must not be modified"
        promoted += len(masked) - len(before[0])
        X, S, mask, lengths, chain_M, ridx, mask_self, chain_enc = fo.featurize([e], orders=[(order, masked)])
        L = int(lengths[0])
        assert flat["S"].shape[0] == L
        assert np.array_equal(flat["S"].astype(np.int64), S[0, :L])
        assert np.array_equal(flat["chain"].astype(np.int64), chain_enc[0, :L])
        assert np.array_equal(flat["designed"].astype(np.float32), chain_M[0, :L])
        assert np.array_equal(np.nan_to_num(flat["X"], nan=0.0), X[0, :L])
        assert np.array_equal(np.isfinite(flat["X"].sum((1, 2))).astype(np.float32), mask[0, :L])
    assert promoted > 0, "the synthetic set must contain a homo-oligomer whose visible copy becomes designed"


def test_token_budget_batcher_is_the_reference_loader():
    lengths = np.array([len(e["seq"]) for e in synthetic_entries(60, seed=5)])
    for budget in (300, 1000, 4000, 100000):
        want = fo.structure_loader_clusters(lengths, budget)
        got = TokenBudgetBatcher(lengths, budget, shuffle=False).clusters
        assert got == want
        for c in got:
            assert len(c) * max(lengths[c], default=0) <= budget
    # the reference drops the entry that closes a batch; drop_overflow=False keeps every entry that fits the budget
    kept = TokenBudgetBatcher(lengths, 1000, shuffle=False, drop_overflow=False).clusters
    assert sorted(i for c in kept for i in c) == sorted(np.nonzero(lengths <= 1000)[0].tolist())
    assert all(len(c) * lengths[c].max() <= 1000 for c in kept)
    dropped = TokenBudgetBatcher(lengths, 1000, shuffle=False).clusters
    assert sum(map(len, dropped)) < sum(map(len, kept))
    # shuffling permutes batches, not their contents
    shuffled = list(TokenBudgetBatcher(lengths, 1000, shuffle=True, rng=np.random.default_rng(0)))
    assert sorted(map(tuple, shuffled)) == sorted(map(tuple, dropped))


def test_shard_offsets_and_state_round_trip():
    entries = synthetic_entries(7, seed=9)
    shard = StructureShard.from_entries(entries, shuffle=False)
    assert len(shard) == 7
    assert shard.lengths.tolist() == [len(e["seq"]) for e in entries]
    assert shard.X.shape == (shard.lengths.sum(), 4, 3) and shard.X.dtype == torch.float32
    assert shard.S.dtype == torch.uint8 and int(shard.S.max()) <= 20
    buf = io.BytesIO()
    torch.save(shard.state(), buf)
    buf.seek(0)
    again = StructureShard.from_state(torch.load(buf, weights_only=True))
    for k in StructureShard.FIELDS:
        a, b = getattr(shard, k), getattr(again, k)
        assert torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float()))
    try:
        shard.assemble([0, 1])
    except RuntimeError as err:
        assert "no CPU fallback" in str(err)
    else:
        raise AssertionError("assemble on a CPU shard must refuse")
