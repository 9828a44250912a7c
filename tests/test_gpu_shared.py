"""Shared-structure forward: many sequences / decoding orders on few structures, featurizer and
encoder run once per distinct structure.  Bars: bit-identical to the same call on the materialised batch (the decoder
arithmetic is the same, only the indexing of the encoder-side tensors differs), and log-probs within 1e-4 relative
(+1e-5 absolute) of the oracle on the materialised batch."""
import numpy as np
import pytest
import torch

from test_gpu_parity import LP_ATOL, LP_RTOL, assert_close, build_model
from conftest import oracle_cfg

pytestmark = pytest.mark.gpu


def make_case(n, m, L, seed, lengths=None, dev="cuda"):
    from mpnn_amd.synthetic import synthetic_batch

    b = synthetic_batch(n, L, seed=seed, lengths=lengths, n_chains=2, device=dev)
    g = torch.Generator().manual_seed(seed + 1)
    B = n * m
    S = torch.randint(0, 21, (B, L), generator=g).to(dev)
    chain_M = (torch.rand(B, L, generator=g) < 0.8).float().to(dev)
    order = torch.argsort(torch.rand(B, L, generator=g), dim=-1).to(dev)
    return b, S, chain_M, order


def tiled(t, B):
    return t.repeat(B // t.shape[0], *([1] * (t.dim() - 1)))


@pytest.mark.parametrize("kw,n,m,L,lengths", [
    (dict(k_neighbors=48), 1, 5, 96, None),
    (dict(k_neighbors=30, use_virtual_center=True, edge_cutoff=14.0, occupancy_cutoff=8.0), 2, 3, 80, [80, 57]),
    (dict(k_neighbors=48, num_encoder_layers=1, num_decoder_layers=2), 3, 2, 64, [64, 50, 40]),
])
def test_shared_equals_materialised_and_oracle(kw, n, m, L, lengths):
    from oracle import mpnn_oracle as orc

    model = build_model(kw)
    b, S, chain_M, order = make_case(n, m, L, seed=11, lengths=lengths)
    B = n * m
    side = (b["X"], b["mask"], b["residue_idx"], b["chain_encoding_all"])
    with torch.no_grad():
        shared = model(side[0], S, side[1], chain_M, side[2], side[3], fix_order=order)
        Xf, mf, rf, cf = (tiled(t, B).contiguous() for t in side)
        full = model(Xf, S, mf, chain_M, rf, cf, fix_order=order)
    assert shared.shape == (B, L, 21)
    assert torch.equal(shared, full), f"max |d| {(shared - full).abs().max().item():.3e}"
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        ref = orc.forward(sd, oracle_cfg(kw), Xf.cpu(), S.cpu(), mf.cpu(), chain_M.cpu(), rf.cpu(), cf.cpu(),
                          fix_order=order.cpu(), formulation="edge", keep=False)
    ref = ref["log_probs"] if isinstance(ref, dict) else ref
    valid = mf.cpu().bool().numpy()
    assert_close(shared.cpu().numpy()[valid], np.asarray(ref)[valid], LP_RTOL, LP_ATOL, "shared log_probs vs oracle")


def test_expanded_views_are_detected_and_chain_m_broadcasts():
    model = build_model(dict(k_neighbors=48))
    b, S, chain_M, order = make_case(1, 4, 72, seed=5)
    B = 4
    X, mask, ridx, chain = (t.expand(B, *t.shape[1:]) for t in
                            (b["X"], b["mask"], b["residue_idx"], b["chain_encoding_all"]))
    with torch.no_grad():
        a = model(X, S, mask, chain_M[:1], ridx, chain, fix_order=order)                      # stride-0 views
        c = model(X.contiguous(), S, mask.contiguous(), chain_M[:1].repeat(B, 1), ridx.contiguous(),
                  chain.contiguous(), fix_order=order)                                      # the reference's repeat()
    assert torch.equal(a, c)


def test_shared_random_order_and_f16_mode_run():
    model = build_model(dict(k_neighbors=48))
    b, S, chain_M, _ = make_case(1, 3, 64, seed=9)
    args = (b["X"], S, b["mask"], chain_M, b["residue_idx"], b["chain_encoding_all"])
    with torch.no_grad():
        lp = model(*args)                      # internal random order, one draw per row
        assert torch.isfinite(lp).all() and lp.shape == (3, 64, 21)
        assert torch.allclose(lp.exp().sum(-1), torch.ones(3, 64, device="cuda"), atol=1e-4)
        order = torch.argsort(torch.rand(3, 64), dim=-1).cuda()
        ref = model(*args, fix_order=order)
        model.set_precision("f16")
        fast = model(*args, fix_order=order)
        model.set_precision("fp32")
    assert (fast - ref).abs().max().item() <= 5e-2   # the reduced-precision bar of include/mpnn_hip.h


def test_training_mode_materialises_and_bad_shapes_raise():
    import mpnn_amd
    from mpnn_amd.synthetic import fill_parameters_
    from mpnn_amd.training import loss_smoothed

    model = mpnn_amd.ProteinMPNN(k_neighbors=32, augment_eps=0.0, dropout=0.0).train()
    fill_parameters_(model, seed=0)
    model = model.cuda()
    b, S, chain_M, order = make_case(1, 2, 48, seed=3)
    lp = model(b["X"], S, b["mask"], chain_M, b["residue_idx"], b["chain_encoding_all"], fix_order=order)
    assert lp.requires_grad and lp.shape == (2, 48, 21)
    _, loss = loss_smoothed(S, lp, chain_M * b["mask"])
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    b3, S5, cm5, _ = make_case(3, 1, 48, seed=4)
    with pytest.raises(ValueError):  # 5 sequence rows on 3 structures
        model.eval()(b3["X"], torch.zeros(5, 48, dtype=torch.long, device="cuda"), b3["mask"], cm5[:1],
                     b3["residue_idx"], b3["chain_encoding_all"])


def test_per_row_backbone_noise_is_not_shared():
    """fix_backbone_noise with one row per SEQUENCE row on an expanded (stride-0) structure: every row is then its own
    noisy structure (reference model_utils.py:311-312 adds the noise per row), so the shared path must not be taken;
    noise given once per structure keeps it."""
    model = build_model(dict(k_neighbors=32))
    b, S, chain_M, order = make_case(1, 3, 56, seed=21)
    B = 3
    g = torch.Generator().manual_seed(5)
    noise = (0.2 * torch.randn(B, 56, 4, 3, generator=g)).cuda()
    X, mask, ridx, chain = (t.expand(B, *t.shape[1:]) for t in
                            (b["X"], b["mask"], b["residue_idx"], b["chain_encoding_all"]))
    with torch.no_grad():
        got = model(X, S, mask, chain_M, ridx, chain, fix_order=order, fix_backbone_noise=noise)
        want = model((X + noise).contiguous(), S, mask.contiguous(), chain_M, ridx.contiguous(), chain.contiguous(),
                     fix_order=order)
        assert torch.allclose(got, want, atol=1e-5), float((got - want).abs().max())
        assert float((got[0] - got[1]).abs().max()) > 1e-3 or not torch.equal(S[0], S[1])
        # n-row structure side with B-row noise
        got2 = model(b["X"], S, b["mask"], chain_M, b["residue_idx"], b["chain_encoding_all"], fix_order=order,
                     fix_backbone_noise=noise)
        assert torch.equal(got2, got)
        # one noise row per structure: shared path, equal to the materialised call
        one = model(b["X"], S, b["mask"], chain_M, b["residue_idx"], b["chain_encoding_all"], fix_order=order,
                    fix_backbone_noise=noise[:1])
        full = model(X.contiguous(), S, mask.contiguous(), chain_M, ridx.contiguous(), chain.contiguous(),
                     fix_order=order, fix_backbone_noise=noise[:1].repeat(B, 1, 1, 1))
        assert torch.equal(one, full)
