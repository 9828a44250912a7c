"""CPU tests of the sparse (edge-list) model's host side and of its oracle.  The reference class
(src/mpnn/models/protein_mpnn.py) needs torch_geometric, which is absent here: the model as a whole is PARITY UNPINNED
(oracle/sparse_oracle.py header); what the reference's own files pin is checked below."""
import pytest
import torch

from oracle import sparse_oracle as so


# inputs of the reference's regression test tests/_test_autoregressive_mask.py:41-52 (data, not code)
REF_EDGE_INDEX = torch.tensor([[0, 0, 1, 1, 2, 2, 0, 2, 1, 3, 3, 4, 4],
                               [0, 1, 0, 1, 2, 0, 2, 1, 2, 3, 4, 3, 4]])
REF_BATCH = torch.tensor([0, 0, 0, 1, 1])


def test_rank_mask_equals_reference_einsum_on_reference_inputs():
    """tests/_test_autoregressive_mask.py:39-71 compares two constructions of the mask on these inputs; the oracle's
    einsum form restates protein_mpnn.py:32-49, and the rank comparison the kernels use must agree with it for every
    decoding order of the padded [2, 3] batch."""
    import itertools

    perms = [torch.tensor(p) for p in itertools.permutations(range(3))]
    for p0 in perms:
        for p1 in perms:
            order = torch.stack([p0, p1])
            a = so.autoregressive_mask_einsum(order, REF_BATCH, REF_EDGE_INDEX)
            b = so.autoregressive_mask_rank(order, REF_BATCH, REF_EDGE_INDEX)
            assert a.shape == (13, 1) and torch.equal(a, b)
    # self loops never attend to their own label
    loops = REF_EDGE_INDEX[0] == REF_EDGE_INDEX[1]
    assert float(b[loops].abs().max()) == 0.0


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_rank_mask_random_multi_graph(seed):
    """The reference's randomised case (:125-163): graphs of 3, 4 and 5 nodes, fully connected with self edges."""
    sizes = [3, 4, 5]
    batch = torch.cat([torch.full((n,), i) for i, n in enumerate(sizes)])
    rows, cols, off = [], [], 0
    for n in sizes:
        idx = torch.arange(n) + off
        rows.append(idx[:, None].expand(n, n).reshape(-1))
        cols.append(idx[None, :].expand(n, n).reshape(-1))
        off += n
    ei = torch.stack([torch.cat(rows), torch.cat(cols)])
    g = torch.Generator().manual_seed(seed)
    n = batch.numel()
    cm = torch.randint(0, 2, (n,), generator=g).float()
    order = so.draw_decoding_order(cm, torch.ones(n), batch, generator=g)
    assert torch.equal(so.autoregressive_mask_einsum(order, batch, ei), so.autoregressive_mask_rank(order, batch, ei))


def expected_keys(n_enc, n_dec, A, occ):
    """state_dict layout of the reference class, read off its constructor (protein_mpnn.py:254-311)."""
    H = 128
    k = {"embedding.emb.weight": (66, 16), "edge_mlp.0.weight": (H, 16 + 16 * A * A), "edge_mlp.0.bias": (H,),
         "edge_mlp.1.weight": (H,), "edge_mlp.1.bias": (H,), "edge_mlp.2.weight": (H, H), "edge_mlp.2.bias": (H,),
         "label_embedding.weight": (21, H), "output.weight": (21, H), "output.bias": (21,)}
    if occ:
        k.update({"occupancy_mlp.0.weight": (H, 1), "occupancy_mlp.0.bias": (H,), "occupancy_mlp.2.weight": (H, H),
                  "occupancy_mlp.2.bias": (H,), "occupancy_mlp.3.weight": (H,), "occupancy_mlp.3.bias": (H,)})

    def mlp(prefix, fan_in):
        for i, f in ((0, fan_in), (2, H), (4, H)):
            k[f"{prefix}.{i}.weight"] = (H, f)
            k[f"{prefix}.{i}.bias"] = (H,)

    def common(prefix, norms):
        for j in norms:
            k[f"{prefix}.norm{j}.weight"] = (H,)
            k[f"{prefix}.norm{j}.bias"] = (H,)
        k[f"{prefix}.dense.out.0.weight"] = (4 * H, H)
        k[f"{prefix}.dense.out.0.bias"] = (4 * H,)
        k[f"{prefix}.dense.out.2.weight"] = (H, 4 * H)
        k[f"{prefix}.dense.out.2.bias"] = (H,)

    for l in range(n_enc):
        mlp(f"encoder_layers.{l}.out_v", 3 * H)
        mlp(f"encoder_layers.{l}.out_e", 3 * H)
        common(f"encoder_layers.{l}", (1, 2, 3))
    for l in range(n_dec):
        mlp(f"decoder_layers.{l}.out_v", 4 * H)
        common(f"decoder_layers.{l}", (1, 2))
    return k


@pytest.mark.parametrize("kw", [dict(), dict(use_virtual_center=True, occupancy_cutoff=8.0, num_encoder_layers=1,
                                             num_decoder_layers=2)])
def test_state_dict_layout_is_the_reference_class_layout(kw):
    from mpnn_amd.sparse import ProteinMPNN

    m = ProteinMPNN(**kw)
    want = expected_keys(kw.get("num_encoder_layers", 3), kw.get("num_decoder_layers", 3),
                         6 if kw.get("use_virtual_center") else 5, kw.get("occupancy_cutoff") is not None)
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == want


def test_constructor_rejects_what_the_kernels_do_not_cover():
    from mpnn_amd.sparse import ProteinMPNN

    with pytest.raises(ValueError):
        ProteinMPNN(hidden_dim=64)
    with pytest.raises(ValueError):
        ProteinMPNN(num_neighbors=65)
    with pytest.raises(ValueError):
        ProteinMPNN(vocab_size=33)


def test_no_cpu_fallback():
    from mpnn_amd.sparse import ProteinMPNN
    from mpnn_amd.synthetic import synthetic_flat_batch

    m = ProteinMPNN().eval()
    b = synthetic_flat_batch([12, 9], seed=1)
    b.pop("decoding_order")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(**b)


def test_dense_layout():
    from mpnn_amd.sparse import to_dense_layout

    B, L, slot = to_dense_layout(torch.tensor([0, 0, 0, 1, 2, 2]))
    assert (B, L) == (3, 3) and slot.tolist() == [0, 1, 2, 3, 6, 7]


def test_oracle_is_a_distribution_and_ignores_unresolved_coordinates():
    from mpnn_amd.sparse import ProteinMPNN
    from mpnn_amd.synthetic import synthetic_flat_batch

    torch.manual_seed(0)
    kw = dict(num_neighbors=8, num_encoder_layers=1, num_decoder_layers=1)
    sd = ProteinMPNN(**kw).state_dict()
    cfg = so.Config(num_neighbors=8, num_encoder_layers=1, num_decoder_layers=1)
    b = synthetic_flat_batch([20, 6], seed=2, n_chains=2, unresolved_every=5)
    order = b.pop("decoding_order")
    with torch.no_grad():
        lp, ei = so.forward(sd, cfg, **b, decoding_order=order, return_graph=True)
    assert lp.shape == (26, 21) and bool(torch.isfinite(lp).all())
    assert float((lp.exp().sum(-1) - 1).abs().max()) < 1e-5
    # graph on the resolved residues only; 8 neighbours, or the whole graph when it is smaller (self included)
    valid = b["mask"] > 0
    assert bool(valid[ei].all())
    deg = torch.bincount(ei[1], minlength=26)
    n_valid = [int(valid[b["batch"] == g].sum()) for g in range(2)]
    for v in range(26):
        want = min(8, n_valid[int(b["batch"][v])]) if valid[v] else 0
        assert int(deg[v]) == want


def test_dense_layout_rejects_unsorted_or_empty_batches():
    from mpnn_amd.sparse import to_dense_layout

    with pytest.raises(ValueError):
        to_dense_layout(torch.tensor([0, 1, 0]))
    with pytest.raises(ValueError):
        to_dense_layout(torch.zeros(0, dtype=torch.long))
    B, L, slot = to_dense_layout(torch.tensor([0, 0, 2]))  # an empty graph in between is allowed (bincount gives it 0 nodes)
    assert (B, L) == (3, 2) and slot.tolist() == [0, 1, 4]
