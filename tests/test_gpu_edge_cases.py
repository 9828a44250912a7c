"""GPU edge cases against the oracle: tiny lengths (K' = min(k, L)), fully masked proteins, odd batch shapes,
non-contiguous / integer-typed inputs, sampling with nothing or everything designable."""
import numpy as np
import pytest
import torch

from test_gpu_parity import LP_ATOL, LP_RTOL, assert_close, build_model, run_hip, run_oracle, same

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,L,k", [(1, 1, 48), (2, 2, 48), (1, 5, 3), (3, 17, 48), (2, 33, 32), (5, 49, 48)])
def test_tiny_lengths(B, L, k):
    from mpnn_amd.synthetic import synthetic_batch

    kw = dict(k_neighbors=k)
    model = build_model(kw)
    b = synthetic_batch(B=B, L=L, seed=100 + L, n_chains=1 if L < 4 else 2, design_fraction=0.5)
    rec = {n: v.numpy() for n, v in b.items()}
    out = run_hip(model, rec)
    ref = run_oracle(model, kw, rec)
    same(out["E_idx"].cpu().numpy(), ref["E_idx"].numpy(), "E_idx")
    same(out["mask_bw"].cpu().numpy(), ref["mask_bw"].numpy(), "mask_bw")
    assert_close(out["log_probs"].cpu().numpy(), ref["log_probs"].numpy(), LP_RTOL, LP_ATOL, "log_probs")


def test_fully_masked_protein_and_ragged_batch():
    """One protein entirely padding, the others ragged: finite outputs, valid rows match the oracle."""
    from mpnn_amd.synthetic import synthetic_batch

    kw = dict(k_neighbors=30)
    model = build_model(kw)
    b = synthetic_batch(B=4, L=70, seed=61, lengths=[70, 0, 41, 1], n_chains=2, design_fraction=0.6)
    rec = {n: v.numpy() for n, v in b.items()}
    out = run_hip(model, rec)
    ref = run_oracle(model, kw, rec)
    lp = out["log_probs"].cpu().numpy()
    assert np.isfinite(lp).all()
    same(out["E_idx"].cpu().numpy(), ref["E_idx"].numpy(), "E_idx")
    m = rec["mask"].astype(bool)
    assert_close(lp[m], ref["log_probs"].numpy()[m], LP_RTOL, LP_ATOL, "log_probs (valid rows)")
    assert_close(lp[~m], ref["log_probs"].numpy()[~m], 1e-5, 1e-5, "log_probs (padded rows)")


def test_input_dtypes_and_strides():
    """mask / chain_M as bool or int, int32 indices, non-contiguous X: same result as the canonical inputs."""
    from mpnn_amd.synthetic import synthetic_batch

    model = build_model(dict(k_neighbors=24))
    b = {k: v.cuda() for k, v in synthetic_batch(B=2, L=50, seed=7, lengths=[50, 37], n_chains=2).items()}
    with torch.no_grad():
        ref = model(b["X"], b["S"], b["mask"], b["chain_M"], b["residue_idx"], b["chain_encoding_all"],
                    fix_order=b["fix_order"])
        Xw = torch.zeros(2, 50, 4, 6, device="cuda")[..., ::2]
        Xw.copy_(b["X"])
        assert not Xw.is_contiguous()
        alt = model(Xw.double(), b["S"].int(), b["mask"].bool(), b["chain_M"].long(), b["residue_idx"].int(),
                    b["chain_encoding_all"].int(), fix_order=b["fix_order"].int())
    assert torch.equal(ref, alt)


def test_sampling_nothing_or_everything_designable():
    from mpnn_amd.synthetic import synthetic_batch

    model = build_model(dict(k_neighbors=16))
    b = {k: v.cuda() for k, v in synthetic_batch(B=2, L=30, seed=9, lengths=[30, 22]).items()}
    args = (b["X"], b["mask"], torch.zeros_like(b["chain_M"]), b["residue_idx"], b["chain_encoding_all"])
    out = model.sample(*args, S_true=b["S"], n_samples=2, temperature=1.0, seed=1)
    assert torch.equal(out["S"][0], b["S"]) and torch.equal(out["S"][1], b["S"])
    args = (b["X"], b["mask"], torch.ones_like(b["chain_M"]), b["residue_idx"], b["chain_encoding_all"])
    out = model.sample(*args, n_samples=3, temperature=1.0, seed=2)
    with torch.no_grad():
        ref = model(b["X"], out["S"][2], b["mask"], torch.ones_like(b["chain_M"]), b["residue_idx"],
                    b["chain_encoding_all"], fix_order=out["order"][2])
    assert float((ref - out["log_probs"][2]).abs().max()) <= 1e-5


def test_training_on_ragged_tiny_batch():
    """Backward on a batch with a fully padded protein and L < 16: finite gradients that match the oracle."""
    from test_gpu_train import build, compare_grads, hip_loss_and_grads, oracle_loss_and_grads
    from mpnn_amd.synthetic import synthetic_batch

    kw = dict(k_neighbors=48)
    b = synthetic_batch(B=3, L=13, seed=17, lengths=[13, 0, 6], n_chains=1, design_fraction=0.8)
    model = build(kw)
    loss, lp, grads = hip_loss_and_grads(model, b)
    oloss, olp, ograds = oracle_loss_and_grads(model, kw, b)
    assert abs(loss - oloss) <= 1e-5 * abs(oloss) + 1e-9
    assert all(torch.isfinite(g).all() for g in grads.values())
    compare_grads(grads, ograds)


def test_hip_graph_replay_is_bit_identical():
    """The inference forward is capturable (no host sync, no allocation outside torch's graph pool) and a replay on
    new inputs of the same shape returns exactly what eager launches return."""
    from mpnn_amd.graphs import GraphedForward
    from mpnn_amd.synthetic import synthetic_batch

    model = build_model(dict(k_neighbors=32))
    keys = ("X", "S", "mask", "chain_M", "residue_idx", "chain_encoding_all", "fix_order")
    b1 = {k: v.cuda() for k, v in synthetic_batch(B=2, L=80, seed=1, lengths=[80, 66]).items()}
    b2 = {k: v.cuda() for k, v in synthetic_batch(B=2, L=80, seed=2, lengths=[71, 80]).items()}
    g = GraphedForward(model, *[b1[k] for k in keys])
    for b in (b2, b1):
        with torch.no_grad():
            ref = model(*[b[k] for k in keys[:6]], fix_order=b["fix_order"])
        assert torch.equal(g(*[b[k] for k in keys]), ref)
    model.refresh_weights()   # e.g. after a training step: the captured graph reads the old packed weights
    with pytest.raises(RuntimeError):
        g(*[b[k] for k in keys])
    # a parameter change WITHOUT an eager forward in between (load_state_dict / in-place copy_): the packed set still
    # has its old identity, the replay must notice through the parameter versions
    g2 = GraphedForward(model, *[b1[k] for k in keys])
    g2(*[b2[k] for k in keys])
    sd = {k: v.clone() * 1.01 for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    with pytest.raises(RuntimeError):
        g2(*[b2[k] for k in keys])
    g3 = GraphedForward(model, *[b1[k] for k in keys])
    with torch.no_grad():
        model.W_out.bias.add_(0.5)
    with pytest.raises(RuntimeError):
        g3(*[b2[k] for k in keys])


@pytest.mark.parametrize("kw,B,L", [
    (dict(k_neighbors=64), 2, 100),                               # four full neighbour tiles (tile-wise message kernel)
    (dict(k_neighbors=56, use_virtual_center=True), 2, 90),       # partial fourth tile, 36 atom pairs
    (dict(k_neighbors=64, edge_cutoff=16.0), 1, 60),              # K' = min(k, L) = 60 with an edge mask
])
def test_more_than_48_neighbours(kw, B, L):
    """The largest neighbourhoods the module accepts (k <= 64): every kernel's fourth column tile."""
    from mpnn_amd.synthetic import synthetic_batch

    model = build_model(kw)
    b = synthetic_batch(B=B, L=L, seed=300 + L, lengths=[L] + [L - 13] * (B - 1), n_chains=2, design_fraction=0.5)
    rec = {n: v.numpy() for n, v in b.items()}
    out = run_hip(model, rec)
    ref = run_oracle(model, kw, rec)
    same(out["E_idx"].cpu().numpy(), ref["E_idx"].numpy(), "E_idx")
    same(out["mask_bw"].cpu().numpy(), ref["mask_bw"].numpy(), "mask_bw")
    same(out["mask_fw"].cpu().numpy(), ref["mask_fw"].numpy(), "mask_fw")
    m = rec["mask"].astype(bool)
    assert_close(out["log_probs"].cpu().numpy()[m], ref["log_probs"].numpy()[m], LP_RTOL, LP_ATOL, "log_probs")
    for got, want in zip(out["h_V_dec"], ref["h_V_dec"]):
        assert_close(got.cpu().numpy()[m], want.numpy()[m], 1e-4, 1e-5, "h_V_dec")
    # the differentiable and the sampling path take the same neighbourhoods (gradients / contract checked in
    # test_gpu_train.py::k64_L80 and test_gpu_sample.py::k64): here only that their log-probs agree with the forward
    d = {k: v.cuda() for k, v in b.items()}
    args = (d["X"], d["S"], d["mask"], d["chain_M"], d["residue_idx"], d["chain_encoding_all"])
    lp = model(*args, fix_order=d["fix_order"])        # grad mode on: differentiable path
    assert lp.requires_grad
    assert_close(lp.detach().cpu().numpy()[m], ref["log_probs"].numpy()[m], LP_RTOL, LP_ATOL, "log_probs (train path)")
    out_s = model.sample(d["X"], d["mask"], d["chain_M"], d["residue_idx"], d["chain_encoding_all"], seed=5)
    assert torch.isfinite(out_s["log_probs"]).all()

