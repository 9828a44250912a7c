"""GPU tests of the sparse (edge-list) ProteinMPNN (reference src/mpnn/models/protein_mpnn.py) against
oracle/sparse_oracle.py.  PARITY UNPINNED for the model as a whole (the reference class needs torch_geometric, absent
here; see the oracle's header) - the bar is the dense forward's: |d log p| <= 1e-5 + 1e-4 |log p| at EVERY node,
unresolved ones included (the reference returns them too)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _inference_unless_asked():
    """The tests here exercise the INFERENCE kernels unless they open torch.enable_grad() themselves (with grad mode on
    the module takes its differentiable path, as any nn.Module does)."""
    with torch.no_grad():
        yield


def randomized(model, seed):
    """Every parameter away from its initial value (biases and LayerNorm vectors are otherwise 0 / 1)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() > 1:
                p.mul_(1.5)
            elif name.endswith("weight"):  # the 1-D weights are the LayerNorm scales
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    return model


CASES = {
    "knn30_ragged": (dict(num_neighbors=30), dict(lengths=[70, 45, 20], seed=3, n_chains=2, design_fraction=0.5,
                                                  unresolved_every=9)),
    "knn48_vc_occ": (dict(num_neighbors=48, use_virtual_center=True, occupancy_cutoff=10.0),
                     dict(lengths=[90, 60], seed=5, n_chains=3, design_fraction=0.7, unresolved_every=0)),
    "radius9": (dict(edge_cutoff=9.0), dict(lengths=[80, 33], seed=7, n_chains=1, unresolved_every=11)),
    "knn16_dec_only": (dict(num_neighbors=16, num_encoder_layers=0, num_decoder_layers=2),
                       dict(lengths=[40, 40, 7], seed=9, n_chains=2, unresolved_every=6)),
    "knn20_enc_only": (dict(num_neighbors=20, num_encoder_layers=2, num_decoder_layers=0),
                       dict(lengths=[50], seed=11, unresolved_every=0)),
    # four column tiles per residue: the tile-wise message kernels (k > 48)
    "knn64_tiles": (dict(num_neighbors=64, num_encoder_layers=2, num_decoder_layers=2),
                    dict(lengths=[100, 70], seed=13, n_chains=2, unresolved_every=13)),
}


def oracle_cfg(kw):
    from oracle.sparse_oracle import Config

    return Config(num_neighbors=kw.get("num_neighbors", 30), num_encoder_layers=kw.get("num_encoder_layers", 3),
                  num_decoder_layers=kw.get("num_decoder_layers", 3), edge_cutoff=kw.get("edge_cutoff"),
                  use_virtual_center=kw.get("use_virtual_center", False), occupancy_cutoff=kw.get("occupancy_cutoff"))


@pytest.mark.parametrize("name", list(CASES))
def test_sparse_forward_matches_oracle(name):
    from mpnn_amd.sparse import ProteinMPNN
    from mpnn_amd.synthetic import synthetic_flat_batch
    from oracle import sparse_oracle as so

    kw, bkw = CASES[name]
    torch.manual_seed(1)
    model = randomized(ProteinMPNN(**kw), seed=2).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    b = synthetic_flat_batch(**bkw)
    order = b.pop("decoding_order")
    with torch.no_grad():
        ref = so.forward(sd, oracle_cfg(kw), **b, decoding_order=order)
    model = model.to("cuda")
    g = {k: v.to("cuda") for k, v in b.items()}
    with torch.no_grad():                                  # the inference kernels
        out = model(**g, decoding_order=order.to("cuda"))
    with torch.enable_grad():                              # grad mode on: the differentiable path (training kernels)
        out_grad = model(**g, decoding_order=order.to("cuda"))
    assert not out.requires_grad and out_grad.requires_grad
    for what, o in (("inference", out), ("differentiable", out_grad.detach())):
        assert o.shape == ref.shape
        err = (o.cpu().double() - ref.double()).abs()
        tol = 1e-5 + 1e-4 * ref.double().abs()
        worst = int((err - tol).max(-1).values.argmax())
        assert bool((err <= tol).all()), (f"{name} ({what}): max |d log p| {float(err.max()):.3e} at node {worst} "
                                          f"(mask {float(b['mask'][worst])}, graph {int(b['batch'][worst])})")


@pytest.mark.parametrize("name", ["knn30_ragged", "knn48_vc_occ", "radius9", "knn16_dec_only", "knn64_tiles"])
def test_sparse_gradients_vs_oracle_autograd(name):
    """Training of the edge-list model (protein_mpnn.py:450-515 under autograd): a masked NLL loss and EVERY parameter
    gradient of the HIP path against autograd through oracle/sparse_oracle.py, at the dense model's bar (1e-3 relative
    L2 per tensor; parity unpinned as the oracle's header says).  Dropout and backbone noise off (they cannot share
    torch's random stream)."""
    from mpnn_amd.sparse import ProteinMPNN
    from mpnn_amd.synthetic import synthetic_flat_batch
    from oracle import sparse_oracle as so

    kw, bkw = CASES[name]
    torch.manual_seed(1)
    model = randomized(ProteinMPNN(dropout=0.0, augment_eps=0.0, **kw), seed=2).train()
    b = synthetic_flat_batch(**bkw)
    order = b.pop("decoding_order")
    w = b["mask"] * b["chain_mask_all"]

    def loss_of(lp, S, w_):
        return -(lp.gather(-1, S.unsqueeze(-1)).squeeze(-1) * w_).sum() / 2000.0

    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    with torch.enable_grad():
        ref_loss = loss_of(so.forward(sd, oracle_cfg(kw), **b, decoding_order=order), b["chain_seq_label"], w)
        ref_loss.backward()
    want = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in sd.items() if v.requires_grad}

    model = model.to("cuda")
    g = {k: v.to("cuda") for k, v in b.items()}
    with torch.enable_grad():
        lp = model(**g, decoding_order=order.to("cuda"))
        loss = loss_of(lp, g["chain_seq_label"], w.to("cuda"))
        loss.backward()
    assert abs(loss.item() - ref_loss.item()) <= 1e-5 * abs(ref_loss.item()), (loss.item(), ref_loss.item())
    total = sum(float(v.double().norm() ** 2) for v in want.values()) ** 0.5
    bad = []
    for n_, p_ in model.named_parameters():
        got = p_.grad.detach().cpu().double() if p_.grad is not None else torch.zeros_like(want[n_]).double()
        err, ref = float((got - want[n_].double()).norm()), float(want[n_].double().norm())
        if err > 1e-3 * ref + 1e-6 * total:
            bad.append(f"{n_}: err {err:.3e} vs norm {ref:.3e}")
    assert not bad, "; ".join(bad[:8])


def test_sparse_training_steps_reduce_loss():
    """A few AdamW steps in training mode with the reference's defaults (dropout 0.1, backbone noise 0.2,
    protein_mpnn.py:136-144, :198-200, :461-462) on the in-tree optimizer: finite, decreasing loss."""
    from mpnn_amd import training
    from mpnn_amd.sparse import ProteinMPNN
    from mpnn_amd.synthetic import synthetic_flat_batch

    torch.manual_seed(0)
    model = ProteinMPNN(num_neighbors=30).train().to("cuda")
    b = {k: v.to("cuda") for k, v in synthetic_flat_batch([96, 70, 64], seed=21, n_chains=2, unresolved_every=17).items()}
    order = b.pop("decoding_order")
    opt, _ = training.make_optimizer(model, lr=1e-3, max_grad_norm=1.0)
    w = b["mask"] * b["chain_mask_all"]
    losses = []
    for _ in range(8):
        opt.zero_grad(set_to_none=True)
        with torch.enable_grad():
            lp = model(**b, decoding_order=order)
            _, loss = training.loss_smoothed(b["chain_seq_label"], lp, w)
            loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(l == l and abs(l) < 1e6 for l in losses), losses
    assert min(losses[-2:]) < losses[0], losses
    assert all(torch.isfinite(p).all() for p in model.parameters())


def test_sparse_random_order_checkpoint_round_trip_and_guards():
    from mpnn_amd.sparse import ProteinMPNN
    from mpnn_amd.synthetic import synthetic_flat_batch

    torch.manual_seed(4)
    src = randomized(ProteinMPNN(num_neighbors=24), seed=5)
    model = ProteinMPNN(num_neighbors=24).eval()
    model.load_state_dict(src.state_dict())
    model = model.to("cuda")
    b = {k: v.to("cuda") for k, v in synthetic_flat_batch([64, 50], seed=13, n_chains=2).items()}
    order = b.pop("decoding_order")
    lp = model(**b)  # decoding order drawn inside, as in the reference
    assert lp.shape == (114, 21) and bool(torch.isfinite(lp).all())
    assert float((lp.exp().sum(-1) - 1).abs().max()) < 1e-4
    a1, a2 = model(**b, decoding_order=order), model(**b, decoding_order=order)
    assert torch.equal(a1, a2)
    # a weight change is picked up (packed weights are keyed on parameter versions)
    with torch.no_grad():
        model.output.bias.add_(1.0)
    a3 = model(**b, decoding_order=order)
    assert not torch.equal(a1, a3)
    with pytest.raises(ValueError):
        model(**b, decoding_order=order[:, :-1])
    with pytest.raises(RuntimeError, match="grad mode"):
        model.train()(**b)   # training mode (dropout, noise) lives on the differentiable path only


def test_sparse_radius_degree_limit_is_loud():
    from mpnn_amd.sparse import ProteinMPNN
    from mpnn_amd.synthetic import synthetic_flat_batch

    model = ProteinMPNN(edge_cutoff=60.0).eval().to("cuda")  # everything within reach: degree = graph size > 64
    b = {k: v.to("cuda") for k, v in synthetic_flat_batch([100], seed=3).items()}
    b.pop("decoding_order")
    with pytest.raises(RuntimeError, match="at most 64"):
        model(**b)


def test_sparse_at_size_graphs_are_independent():
    """Size-independent property at the headline shape (16 graphs x 512 residues, 48 neighbours): a graph's log-probs
    do not depend on what else is in the batch - the graph evaluated alone gives the same numbers (rows of different
    lengths, so the padded layouts differ)."""
    from mpnn_amd.sparse import ProteinMPNN
    from mpnn_amd.synthetic import synthetic_flat_batch

    torch.manual_seed(6)
    model = randomized(ProteinMPNN(num_neighbors=48), seed=7).eval().to("cuda")
    lengths = [512] * 15 + [301]
    b = {k: v.to("cuda") for k, v in synthetic_flat_batch(lengths, seed=17, n_chains=2, unresolved_every=50).items()}
    order = b.pop("decoding_order")
    lp = model(**b, decoding_order=order)
    assert lp.shape == (sum(lengths), 21) and bool(torch.isfinite(lp).all())
    assert float((lp.exp().sum(-1) - 1).abs().max()) < 1e-4
    for g in (0, 15):
        sel = b["batch"] == g
        one = {k: v[sel] for k, v in b.items()}
        one["batch"] = torch.zeros_like(one["batch"])
        alone = model(**one, decoding_order=order[g:g + 1, :lengths[g]] if lengths[g] == 512 else
                      torch.argsort(torch.argsort(order[g:g + 1])[:, :lengths[g]]))
        assert float((alone - lp[sel]).abs().max()) <= 1e-6


def test_sparse_f16_products_stay_within_the_reduced_precision_bar():
    """model.precision = "f16": one f16 product per fp32 product (include/mpnn_hip.h); same bar as the dense model's
    reduced-precision forward: max |d log p| <= 5e-2 against the fp32-class path, same graph."""
    from mpnn_amd.sparse import ProteinMPNN
    from mpnn_amd.synthetic import synthetic_flat_batch

    torch.manual_seed(8)
    model = randomized(ProteinMPNN(num_neighbors=30), seed=9).eval().to("cuda")
    b = {k: v.to("cuda") for k, v in synthetic_flat_batch([128, 90], seed=19, n_chains=2).items()}
    order = b.pop("decoding_order")
    ref = model(**b, decoding_order=order)
    model.precision = "f16"
    low = model(**b, decoding_order=order)
    model.precision = "fp32"
    d = float((low - ref).abs().max())
    assert 0.0 < d <= 5e-2, d
    assert float((low.argmax(-1) == ref.argmax(-1)).float().mean()) >= 0.97


@pytest.mark.parametrize("seed", range(12))
def test_sparse_random_configurations(seed):
    """Random constructor arguments, ragged graphs, unresolved residues and decoding orders against the oracle."""
    import numpy as np

    from mpnn_amd.sparse import ProteinMPNN
    from mpnn_amd.synthetic import synthetic_flat_batch
    from oracle import sparse_oracle as so

    rng = np.random.default_rng(1000 + seed)
    kw = dict(num_neighbors=int(rng.choice([4, 16, 30, 33, 48, 64])), num_encoder_layers=int(rng.integers(0, 4)),
              num_decoder_layers=int(rng.integers(0, 4)), use_virtual_center=bool(rng.random() < 0.4))
    if rng.random() < 0.3:
        kw["edge_cutoff"] = float(rng.uniform(6.0, 10.0))
    if rng.random() < 0.3:
        kw["occupancy_cutoff"] = float(rng.uniform(5.0, 10.0))
    lengths = [int(v) for v in rng.integers(3, 120, size=int(rng.integers(1, 5)))]
    bkw = dict(lengths=lengths, seed=int(rng.integers(0, 1 << 20)), n_chains=int(rng.integers(1, 4)),
               design_fraction=float(rng.uniform(0.2, 1.0)), unresolved_every=int(rng.choice([0, 0, 5, 17])))
    torch.manual_seed(seed)
    model = randomized(ProteinMPNN(**kw), seed=seed + 50).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    b = synthetic_flat_batch(**bkw)
    order = b.pop("decoding_order")
    with torch.no_grad():
        ref = so.forward(sd, oracle_cfg(kw), **b, decoding_order=order)
    out = model.to("cuda")(**{k: v.to("cuda") for k, v in b.items()}, decoding_order=order.to("cuda"))
    err = (out.cpu().double() - ref.double()).abs()
    assert bool((err <= 1e-5 + 1e-4 * ref.double().abs()).all()), (kw, bkw, float(err.max()))
