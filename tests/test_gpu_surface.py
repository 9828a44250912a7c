"""The small public surfaces of the reference's model file on the HIP library: the four gather helpers
(model_utils.py:13-45), PositionWiseFeedForward.forward (:151-155) and PositionalEncodings.forward (:168-177),
against plain-torch restatements on the CPU (index copies are bit-exact; the FFN holds the forward's 1e-4 bar)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref_gather_nodes(nodes, idx):
    B, N, K = idx.shape
    flat = idx.reshape(B, N * K, 1).expand(-1, -1, nodes.shape[-1])
    return torch.gather(nodes, 1, flat).reshape(B, N, K, -1)


@pytest.mark.parametrize("C", [128, 3, 20])
def test_gather_helpers_are_exact(C):
    import mpnn_amd

    g = torch.Generator().manual_seed(C)
    B, N, K = 3, 37, 11
    nodes = torch.randn(B, N, C, generator=g)
    edges = torch.randn(B, N, N, C, generator=g)
    nb = torch.randn(B, N, K, 256 if C % 4 == 0 else 5, generator=g)
    idx = torch.randint(0, N, (B, N, K), generator=g)
    idx_t = torch.randint(0, N, (B, K), generator=g)
    d = "cuda"
    assert torch.equal(mpnn_amd.gather_nodes(nodes.to(d), idx.to(d)).cpu(), _ref_gather_nodes(nodes, idx))
    want_e = torch.gather(edges, 2, idx.unsqueeze(-1).expand(-1, -1, -1, C))
    assert torch.equal(mpnn_amd.gather_edges(edges.to(d), idx.to(d)).cpu(), want_e)
    want_t = torch.gather(nodes, 1, idx_t.unsqueeze(-1).expand(-1, -1, C))
    assert torch.equal(mpnn_amd.gather_nodes_t(nodes.to(d), idx_t.to(d)).cpu(), want_t)
    want_c = torch.cat([nb, _ref_gather_nodes(nodes, idx)], -1)
    assert torch.equal(mpnn_amd.cat_neighbors_nodes(nodes.to(d), nb.to(d), idx.to(d)).cpu(), want_c)
    # the [B,L,L] scalar-pair form the featurizer uses (C = 1 via unsqueeze), int32 indices, non-contiguous input
    pair = torch.randn(B, N, N, generator=g)
    got = mpnn_amd.gather_edges(pair.to(d).unsqueeze(-1), idx.to(d).int()).squeeze(-1).cpu()
    assert torch.equal(got, torch.gather(pair, 2, idx))
    with pytest.raises(IndexError):
        mpnn_amd.gather_nodes(nodes.to(d), (idx + N).to(d))
    with pytest.raises(RuntimeError):
        mpnn_amd.gather_nodes(nodes, idx)            # CPU tensors: no fallback


def test_gather_gradients():
    import mpnn_amd

    g = torch.Generator().manual_seed(1)
    B, N, K, C = 2, 19, 7, 128
    nodes = torch.randn(B, N, C, generator=g)
    nb = torch.randn(B, N, K, 128, generator=g)
    idx = torch.randint(0, N, (B, N, K), generator=g)
    w = torch.randn(B, N, K, 256, generator=g)
    a, b = nodes.clone().requires_grad_(True), nb.clone().requires_grad_(True)
    (torch.cat([b, _ref_gather_nodes(a, idx)], -1) * w).sum().backward()
    x, y = nodes.cuda().requires_grad_(True), nb.cuda().requires_grad_(True)
    (mpnn_amd.cat_neighbors_nodes(x, y, idx.cuda()) * w.cuda()).sum().backward()
    assert torch.allclose(x.grad.cpu(), a.grad, rtol=1e-5, atol=1e-5) and torch.equal(y.grad.cpu(), b.grad)


def test_position_wise_feed_forward_standalone():
    import mpnn_amd
    from mpnn_amd.synthetic import fill_parameters_

    ffn = mpnn_amd.PositionWiseFeedForward(128, 512)
    fill_parameters_(ffn, seed=3)
    g = torch.Generator().manual_seed(2)
    for shape in ((2, 45, 128), (1, 128), (3, 5, 7, 128)):
        x = torch.randn(*shape, generator=g) * 2.0
        with torch.no_grad():
            want = F.linear(F.gelu(F.linear(x, ffn.W_in.weight, ffn.W_in.bias)), ffn.W_out.weight, ffn.W_out.bias)
            got = ffn.cuda()(x.cuda()).cpu()
        ffn.cpu()
        assert got.shape == want.shape
        err = (got - want).abs() - (1e-5 + 1e-4 * want.abs())
        assert float(err.max()) <= 0, float((got - want).abs().max())


def test_positional_encodings_standalone():
    import mpnn_amd
    from mpnn_amd.synthetic import fill_parameters_

    pe = mpnn_amd.PositionalEncodings(16)
    fill_parameters_(pe, seed=4)
    g = torch.Generator().manual_seed(5)
    offset = torch.randint(-90, 90, (2, 30, 12), generator=g)
    mask = torch.randint(0, 2, (2, 30, 12), generator=g)
    d = torch.clip(offset + 32, 0, 64) * mask + (1 - mask) * 65
    with torch.no_grad():
        want = F.linear(F.one_hot(d, 66).float(), pe.linear.weight, pe.linear.bias)
        got = pe.cuda()(offset.cuda(), mask.cuda()).cpu()
    assert got.shape == want.shape and torch.allclose(got, want, rtol=0, atol=1e-6)


def test_enc_message_one_row_for_every_node():
    """mpnn_enc_message with ldpq = 0 (one P | Q row for every node: h_V = 0 before the first encoder layer,
    model_utils.py:448, so the projections are the bias row) returns exactly what the materialised [N, 256] buffer
    gives - the form the forward uses instead of a projection launch."""
    from test_gpu_parity import build_model
    from mpnn_amd import engine
    from mpnn_amd._lib import call, ptr, stream_ptr
    from mpnn_amd.synthetic import synthetic_batch

    model = build_model(dict(k_neighbors=30))
    b = {k: v.cuda() for k, v in synthetic_batch(B=2, L=70, seed=3, lengths=[70, 51], n_chains=2).items()}
    pw, fm = engine.packed(model), model.features
    mask = b["mask"].float().contiguous()
    B, L = mask.shape
    N, H = B * L, 128
    atoms, _, _, E32, _, emask, K = engine._geometry(fm, b["X"], mask, None, False)
    hE, _ = engine._edge_features(pw, atoms, E32, b["residue_idx"].long(), b["chain_encoding_all"].long(), B, L, K,
                                  False)
    lw, row = pw.enc[0], pw.g_init[1]
    assert row.numel() == 2 * H and float(row[:H].abs().max()) > 0
    outs = []
    for PQ, ld in ((row.view(1, -1).repeat(N, 1).contiguous(), 2 * H), (row, 0)):
        gsum = torch.full((N, H), float("nan"), device="cuda")
        cnt = torch.full((N,), float("nan"), device="cuda")
        call("mpnn_enc_message", ptr(hE), ptr(PQ), ld, 0, H, ptr(E32), ptr(mask), ptr(emask), ptr(lw["W1b"]),
             ptr(lw["W2"]), ptr(lw["b2"]), B, L, K, ptr(gsum), ptr(cnt), stream_ptr())
        torch.cuda.synchronize()
        outs.append((gsum, cnt))
    assert torch.isfinite(outs[0][0]).all() and float(outs[0][0].abs().max()) > 0
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    with pytest.raises(RuntimeError):   # any other stride below one row of P stays an error
        call("mpnn_enc_message", ptr(hE), ptr(row), 64, 0, H, ptr(E32), ptr(mask), ptr(emask), ptr(lw["W1b"]),
             ptr(lw["W2"]), ptr(lw["b2"]), B, L, K, ptr(outs[0][0]), ptr(outs[0][1]), stream_ptr())
