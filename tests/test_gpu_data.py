"""GPU batch assembly (mpnn_assemble_batch through the C ABI) against the numpy restatement of the reference's
featurize (oracle/featurize_oracle.py).  Bar: every output bit-exact (integer / byte work and copies)."""
import random

import numpy as np
import pytest
import torch

from mpnn_amd.data import StructureShard, TokenBudgetBatcher
from mpnn_amd.synthetic import synthetic_entries
from oracle import featurize_oracle as fo

pytestmark = pytest.mark.gpu

NAMES = ("X", "S", "mask", "lengths", "chain_M", "residue_idx", "mask_self", "chain_encoding_all")


def build(n, seed, with_inf=False):
    entries = synthetic_entries(n, seed=seed)
    if with_inf:
        entries[1]["coords_chain_A"]["CA_chain_A"][3, 1] = np.inf   # masked out but NOT zeroed (:170-172)
    rngs = [random.Random(1000 + i) for i in range(n)]
    orders = [fo.chain_order(e, random.Random(1000 + i)) for i, e in enumerate(entries)]
    flat_rng = iter(rngs)

    class _Seq:  # flatten_entry draws its shuffle from the same per-entry generator the oracle used
        def shuffle(self, x):
            next(flat_rng).shuffle(x)

    shard = StructureShard.from_entries(entries, rng=_Seq()).to("cuda")
    return entries, orders, shard


@pytest.mark.parametrize("ids", [[0], [1, 5, 2], list(range(12)), [7, 7, 3], [11, 0]])
def test_assemble_equals_featurize(ids):
    entries, orders, shard = build(12, seed=21, with_inf=True)
    want = fo.featurize([entries[i] for i in ids], orders=[orders[i] for i in ids])
    got = shard.assemble(ids, with_mask_self=True)
    torch.cuda.synchronize()
    for name, w, g in zip(NAMES, want, got):
        g = g.cpu().numpy()
        assert g.dtype == w.dtype, (name, g.dtype, w.dtype)
        assert g.shape == w.shape, (name, g.shape, w.shape)
        assert np.array_equal(g, w, equal_nan=True), f"{name}: {(g != w).sum()} entries differ"
    assert shard.assemble(ids)[6] is None


def test_token_budget_epoch_feeds_the_model():
    import mpnn_amd
    from mpnn_amd.synthetic import fill_parameters_

    entries, orders, shard = build(40, seed=33)
    model = mpnn_amd.ProteinMPNN(k_neighbors=32, augment_eps=0.0, dropout=0.0).eval()
    fill_parameters_(model, seed=0)
    model = model.cuda()
    batches = list(TokenBudgetBatcher(shard.lengths, batch_size=1500, shuffle=True, rng=np.random.default_rng(1)))
    assert len(batches) >= 3
    for ids in batches[:3]:
        X, S, mask, lengths, chain_M, ridx, _, chain_enc = shard.assemble(ids)
        ref = fo.featurize([entries[i] for i in ids], orders=[orders[i] for i in ids])
        order = torch.argsort(torch.rand(S.shape, generator=torch.Generator().manual_seed(len(ids))), dim=-1).cuda()
        with torch.no_grad():
            a = model(X, S, mask, chain_M, ridx, chain_enc, fix_order=order)
            b = model(*(torch.from_numpy(ref[k]).cuda() for k in (0, 1, 2, 4, 5, 7)), fix_order=order)
        assert torch.equal(a, b)
        assert int(lengths.max()) == X.shape[1]


def test_bad_ids_raise():
    _, _, shard = build(4, seed=2)
    with pytest.raises(IndexError):
        shard.assemble([0, 4])
    with pytest.raises(IndexError):
        shard.assemble([])


def test_training_over_assembled_batches_of_changing_shape():
    """The loop the pieces are for: shard -> token-budget batches (different B and L every step) -> train_step with
    the reference's recipe.  Finite loss that falls from the first to the third epoch, finite weights."""
    import mpnn_amd
    from mpnn_amd import training
    from mpnn_amd.synthetic import fill_parameters_

    torch.manual_seed(0)  # dropout seeds, backbone noise and decoding orders are drawn from torch's generators
    _, _, shard = build(40, seed=44)
    model = mpnn_amd.ProteinMPNN(k_neighbors=32, augment_eps=0.1, dropout=0.1).train()
    fill_parameters_(model, seed=0)
    model = model.cuda()
    batches = [b for b in TokenBudgetBatcher(shard.lengths, batch_size=1200, shuffle=False, drop_overflow=False).clusters]
    assert len({(len(b), int(shard.lengths[b].max())) for b in batches}) >= 3
    opt, sched = training.make_optimizer(model, lr=2e-3, total_steps=3 * len(batches))

    def held_out_loss():  # smoothed loss per designed residue of one fixed batch, inference kernels, eval mode
        X, S, mask, _, chain_M, ridx, _, chain_enc = shard.assemble(batches[0])
        order = torch.argsort(torch.rand(S.shape, generator=torch.Generator().manual_seed(5)), dim=-1).cuda()
        model.eval()
        with torch.no_grad():
            lp = model(X, S, mask, chain_M, ridx, chain_enc, fix_order=order)
        model.train()
        _, total = training.loss_smoothed(S, lp, mask * chain_M)
        return float(total) * 2000.0 / float((mask * chain_M).sum())

    before = held_out_loss()
    for epoch in range(3):
        for ids in batches:
            X, S, mask, _, chain_M, ridx, _, chain_enc = shard.assemble(ids)
            loss, lp = training.train_step(model, opt, (X, S, mask, chain_M, ridx, chain_enc), scheduler=sched,
                                           max_grad_norm=1.0)
            assert np.isfinite(float(loss)) and lp.shape == (len(ids), X.shape[1], 21)
    after = held_out_loss()
    assert all(torch.isfinite(p).all() for p in model.parameters())
    assert np.isfinite(after) and after < before, (before, after)


@pytest.mark.parametrize("ids", [[0], [1, 5, 2], list(range(12)), [7, 7, 3]])
def test_per_batch_chain_shuffle_equals_featurize(ids):
    """featurize shuffles the chain order of an entry EVERY time it is drawn (data_utils.py:88): a shard built in the
    unshuffled order + assemble(shuffle_chains=True, rng) lays the chains out as the reference would with the same
    generator - chain numbers, the +100 residue-index jumps, designed flags and coordinates bit for bit."""
    entries = synthetic_entries(12, seed=21)
    shard = StructureShard.from_entries(entries, shuffle=False).to("cuda")
    for draw in range(3):                               # the same structures come out in a new order on every draw
        want = fo.featurize([entries[i] for i in ids], rng=random.Random(500 + draw), shuffle=True)
        got = shard.assemble(ids, with_mask_self=True, shuffle_chains=True, rng=random.Random(500 + draw))
        torch.cuda.synchronize()
        for name, w, g in zip(NAMES, want, got):
            g = g.cpu().numpy()
            assert g.dtype == w.dtype and g.shape == w.shape, name
            assert np.array_equal(g, w, equal_nan=True), f"draw {draw}, {name}: {(g != w).sum()} entries differ"
    # explicit orders: the identity permutation is the stored order
    same = shard.assemble(ids, chain_orders=[list(range(int(shard._seg_cnt[i]))) for i in ids])
    plain = shard.assemble(ids)
    for a, b in zip(same, plain):
        assert a is None and b is None or torch.equal(a, b)
    with pytest.raises(ValueError):
        shard.assemble(ids, chain_orders=[[0, 0]] * len(ids))
