"""GPU parity of the training step: loss and every parameter gradient of the HIP path (edge kernels forward +
backward through the C ABI, node-level ops on the device) against autograd through the CPU oracle, and against
the loss / gradient summary the reference itself produced (tests/golden/h_train_B2_L64.npz).

Bar (BASELINE north_star): gradients within 1e-3 relative L2 per parameter tensor; loss within 1e-5 relative."""
import numpy as np
import pytest
import torch

from conftest import load_golden, oracle_cfg

pytestmark = pytest.mark.gpu

T = torch.from_numpy
GRAD_RTOL = 1e-3


def build(kw, device="cuda", dropout=0.0, augment_eps=0.0):
    import mpnn_amd
    from mpnn_amd.synthetic import fill_parameters_

    model = mpnn_amd.ProteinMPNN(augment_eps=augment_eps, dropout=dropout, **kw).train()
    fill_parameters_(model, seed=0)
    return model.to(device)


def hip_loss_and_grads(model, b):
    from mpnn_amd import training

    d = "cuda"
    model.zero_grad(set_to_none=True)
    lp = training.forward_train(model, b["X"].to(d), b["S"].to(d), b["mask"].to(d), b["chain_M"].to(d),
                                b["residue_idx"].to(d), b["chain_encoding_all"].to(d), fix_order=b["fix_order"].to(d))
    _, loss = training.loss_smoothed(b["S"].to(d), lp, (b["mask"] * b["chain_M"]).to(d))
    loss.backward()
    torch.cuda.synchronize()
    grads = {n: (p.grad.detach().cpu() if p.grad is not None else torch.zeros_like(p).cpu())
             for n, p in model.named_parameters()}
    return loss.item(), lp.detach().cpu(), grads


def oracle_loss_and_grads(model, kw, b):
    from oracle import mpnn_oracle as orc

    sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    lp = orc.forward(sd, oracle_cfg(kw), b["X"], b["S"], b["mask"], b["chain_M"], b["residue_idx"],
                     b["chain_encoding_all"], fix_order=b["fix_order"], formulation="edge")
    _, loss = orc.loss_smoothed(b["S"], lp, b["mask"] * b["chain_M"])
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in sd.items() if v.requires_grad}
    return loss.item(), lp.detach(), grads


def compare_grads(got, want):
    worst = []
    total = sum(float(v.double().norm() ** 2) for v in want.values()) ** 0.5
    for name, w in want.items():
        g = got[name]
        err = float((g.double() - w.double()).norm())
        ref = float(w.double().norm())
        # a tensor whose true gradient is (numerically) zero is held to the scale of the whole gradient
        bound = GRAD_RTOL * ref + 1e-6 * total
        worst.append((err / (ref + 1e-30), name, err, ref, err <= bound))
    bad = [w for w in worst if not w[4]]
    worst.sort(reverse=True)
    assert not bad, "gradient mismatch: " + "; ".join(f"{n}: err {e:.3e} vs norm {r:.3e}" for _, n, e, r, _ in bad[:8])
    return worst[0]


CASES = {
    "k48_L64": (dict(k_neighbors=48), dict(B=2, L=64, seed=9, lengths=[64, 52], n_chains=2, design_fraction=0.5)),
    "k20_L40": (dict(k_neighbors=20), dict(B=3, L=40, seed=4, lengths=[40, 33, 17], n_chains=2, design_fraction=0.7)),
    "k30_cutoff_occ_vc": (dict(k_neighbors=30, edge_cutoff=12.0, occupancy_cutoff=10.0, use_virtual_center=True),
                          dict(B=2, L=72, seed=5, lengths=[72, 50], n_chains=3, design_fraction=0.6)),
    # four column tiles: the K > 48 instances of the training kernels (streaming featurizer, KTM = 4 weight gradients)
    "k64_L80": (dict(k_neighbors=64), dict(B=2, L=80, seed=11, lengths=[80, 70], n_chains=2, design_fraction=0.5)),
    "k56_L96_vc": (dict(k_neighbors=56, use_virtual_center=True),
                   dict(B=2, L=96, seed=12, lengths=[96, 61], n_chains=2, design_fraction=0.6)),
    # BASELINE configs[2]'s protein length: hub residues with hundreds of inverse-list entries, L^2 slot maps
    "k48_L512": (dict(k_neighbors=48), dict(B=2, L=512, seed=14, lengths=[512, 400], n_chains=2, design_fraction=0.6)),
}


@pytest.mark.parametrize("name", list(CASES))
def test_gradients_vs_oracle_autograd(name):
    from mpnn_amd.synthetic import synthetic_batch

    kw, bkw = CASES[name]
    b = synthetic_batch(**bkw)
    model = build(kw)
    loss, lp, grads = hip_loss_and_grads(model, b)
    oloss, olp, ograds = oracle_loss_and_grads(model, kw, b)
    assert abs(loss - oloss) <= 1e-5 * abs(oloss), (loss, oloss)
    assert float((lp - olp).abs().max()) <= 1e-4
    worst = compare_grads(grads, ograds)
    print(f"{name}: loss {loss:.6f}, worst relative gradient error {worst[0]:.2e} ({worst[1]})")


def test_gradients_at_configs2_size():
    """BASELINE configs[2] at size: ONE step's batch of 20 x 512 residues (ragged, two chains each).  The weight-gradient
    kernels' split-K partition over ~10 k residues, their partial-sum workspace, the per-wave power-of-two gradient
    scaling and the B L^2 inverse-list slot map only show at this size.  Proteins are independent and the loss
    normaliser is a constant (train.py:72), so the oracle's loss and gradients are accumulated over chunks of four
    proteins (bounded host memory); the HIP path sees the whole batch at once."""
    from mpnn_amd.synthetic import synthetic_batch
    from oracle import mpnn_oracle as orc

    kw = dict(k_neighbors=48)
    B, L = 20, 512
    lengths = [512, 480, 512, 350, 512, 505, 300, 512, 444, 512, 512, 391, 512, 470, 512, 333, 512, 512, 420, 512]
    b = synthetic_batch(B=B, L=L, seed=33, lengths=lengths, n_chains=2, design_fraction=0.7)
    model = build(kw)
    loss, lp, grads = hip_loss_and_grads(model, b)

    sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    oloss = 0.0
    keys = ("X", "S", "mask", "chain_M", "residue_idx", "chain_encoding_all", "fix_order")
    for r0 in range(0, B, 4):
        c = {k: b[k][r0:r0 + 4] for k in keys}
        olp = orc.forward(sd, oracle_cfg(kw), c["X"], c["S"], c["mask"], c["chain_M"], c["residue_idx"],
                          c["chain_encoding_all"], fix_order=c["fix_order"], formulation="edge")
        _, l_ = orc.loss_smoothed(c["S"], olp, c["mask"] * c["chain_M"])
        l_.backward()                      # .grad accumulates over the chunks
        oloss += l_.item()
        m = c["mask"].bool()
        err = (lp[r0:r0 + 4] - olp.detach()).abs()[m]
        assert bool((err <= 1e-5 + 1e-4 * olp.detach().abs()[m]).all()), (r0, err.max().item())
    ograds = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in sd.items() if v.requires_grad}
    assert abs(loss - oloss) <= 1e-5 * abs(oloss), (loss, oloss)
    worst = compare_grads(grads, ograds)
    print(f"20 x 512: loss {loss:.6f}, worst relative gradient error {worst[0]:.2e} ({worst[1]})")


def test_gradients_vs_reference_golden():
    """The reference's own loss and per-parameter gradient norms / leading entries (make_golden.py training_case)."""
    rec, _ = load_golden("h_train_B2_L64")
    b = {k: T(rec[k]) for k in ("X", "S", "mask", "chain_M", "residue_idx", "chain_encoding_all", "fix_order")}
    model = build(dict(k_neighbors=48))
    loss, lp, grads = hip_loss_and_grads(model, b)
    assert abs(loss - float(rec["loss"])) <= 1e-5 * abs(float(rec["loss"]))
    assert float((lp - T(rec["log_probs"])).abs().max()) <= 1e-4
    total = sum(float(rec["gnorm/" + n]) ** 2 for n in grads) ** 0.5
    for n, g in grads.items():
        ref_norm = float(rec["gnorm/" + n])
        assert abs(float(g.double().norm()) - ref_norm) <= GRAD_RTOL * ref_norm + 1e-6 * total, n
        head = g.reshape(-1)[:16].double().numpy()
        assert np.abs(head - rec["ghead/" + n]).max() <= GRAD_RTOL * ref_norm + 1e-6 * total, n


def test_wgrad_and_gather_kernels():
    """mpnn_edge_wgrad / mpnn_gather_rows against ATen on random data (both operand layouts, GELU on/off)."""
    from mpnn_amd import engine, training
    from mpnn_amd._lib import call, ptr, stream_ptr

    torch.manual_seed(0)
    d = "cuda"
    N, K = 70, 40
    Kp = 48
    A_nat = torch.randn(N, K, 128, device=d)
    B_nat = torch.randn(N, K, 128, device=d)
    A_frag, B_frag = engine.edges_from_natural(A_nat, N, K), engine.edges_from_natural(B_nat, N, K)
    A_rows = torch.zeros(N, Kp, 128, device=d)
    A_rows[:, :K] = A_nat
    for a_rows, scale in ((0, 1.0), (1, 3e-7)):   # gradient-sized operands too (f16 split needs the scaling)
        for gelu in (0, 1):
            dW = torch.zeros(128, 128, device=d)
            db = torch.zeros(128, device=d)
            src = (A_rows if a_rows else A_frag) * scale
            amax = src.abs().max().reshape(1)
            ws = torch.empty(37 * 16512, device=d)  # room for 37 workgroup partials only
            call("mpnn_edge_wgrad", ptr(src), a_rows, ptr(amax), ptr(B_frag), gelu, N, K, ptr(dW), ptr(db), ptr(ws),
                 ws.numel(), stream_ptr())
            dW, db = dW / scale, db / scale
            Bx = torch.nn.functional.gelu(B_nat) if gelu else B_nat
            want = torch.einsum("nko,nki->oi", A_nat.double(), Bx.double())
            assert float((dW.double() - want).norm() / want.norm()) < 2e-6
            assert float((db.double() - A_nat.double().sum((0, 1))).abs().max()) < 1e-3
    # gather: random neighbour lists
    Bb, L = 2, 35
    E_idx = torch.stack([torch.stack([torch.randperm(L)[:K] if K <= L else torch.randint(0, L, (K,)) for _ in range(L)])
                         for _ in range(Bb)]).to(d)
    K2 = E_idx.shape[-1]
    g = training.Graph(E_idx, None, None, None, K2)
    rows = torch.randn(g.N * g.Kp, 128, device=d)
    rank = torch.stack([torch.randperm(L) for _ in range(Bb)]).reshape(-1).to(d).to(torch.int32)
    g.set_order(rank)
    outA, outB = training._gather(rows, g, g.sel)
    real = rows.view(g.N, g.Kp, 128)[:, :K2].reshape(-1, 128).double()
    before = (rank.long()[g.src] > rank.long()[g.dest]).double().unsqueeze(-1)
    wantA = torch.zeros(g.N, 128, device=d, dtype=torch.float64).index_add_(0, g.dest, real * before)
    wantB = torch.zeros(g.N, 128, device=d, dtype=torch.float64).index_add_(0, g.dest, real * (1 - before))
    assert float((outA.double() - wantA).abs().max()) < 1e-4
    assert float((outB.double() - wantB).abs().max()) < 1e-4


@pytest.mark.parametrize("B,L,K", [(2, 35, 30), (1, 100, 48), (3, 512, 64), (2, 70, 1), (1, 16, 16), (2, 131, 48)])
def test_inverse_lists_match_stable_sort(B, L, K):
    """mpnn_inverse_lists / mpnn_inverse_sel against the lists a stable sort of the edges by their target gives
    (the ATen construction they replace): identical offsets, rows, chunk offsets and selectors."""
    from mpnn_amd import training

    g0 = torch.Generator().manual_seed(1000 * L + K)
    d = "cuda"
    # hub-heavy neighbour lists: residues near the start of a protein are picked far more often
    w = 1.0 / (1.0 + torch.arange(L, dtype=torch.float64)) ** 0.7
    E_idx = torch.stack([torch.stack([torch.multinomial(w, K, replacement=False, generator=g0) for _ in range(L)])
                         for _ in range(B)]).to(d)
    g = training.Graph(E_idx, None, None, None, K)
    ref = training.Graph.__new__(training.Graph)
    ref.__dict__.update(B=B, L=L, K=K, N=B * L, Kp=g.Kp, E_idx64=E_idx, max_chunks=g.max_chunks)
    ref._lists_by_sort(E_idx)
    torch.cuda.synchronize()
    assert torch.equal(g.inv_ptr, ref.inv_ptr) and int(g.inv_ptr[-1]) == B * L * K
    assert torch.equal(g.chunk_ptr, ref.chunk_ptr)
    nc = int(g.chunk_ptr[-1])
    assert nc <= g.max_chunks and torch.equal(g.chunk_node[:nc], ref.chunk_node[:nc])
    assert torch.equal(g.inv_row, ref.inv_row)
    rank = torch.stack([torch.randperm(L, generator=g0) for _ in range(B)]).reshape(-1).to(d).to(torch.int32)
    g.set_order(rank)
    r = rank.long()
    perm = torch.argsort(g.dest, stable=True)
    want = (r[g.src] <= r[g.dest]).to(torch.uint8)[perm]
    assert torch.equal(g.sel, want)


def test_inverse_lists_with_repeated_or_bad_neighbours():
    """Caller-supplied neighbour lists (Graph(E_idx, None, ...)) may name a neighbour twice: the slot-map construction
    keeps one edge per (target, centre) pair, reports the short list through its status word and Graph falls back to
    the sort - the gather's gradient then counts both edges.  An id outside the protein is refused."""
    from mpnn_amd import training

    d = "cuda"
    B, L, K = 2, 40, 12
    g0 = torch.Generator().manual_seed(5)
    E_idx = torch.randint(0, L, (B, L, K), generator=g0).to(d)      # with repetitions
    assert any(len(set(row.tolist())) < K for row in E_idx[0])
    g = training.Graph(E_idx, None, None, None, K)
    assert int(g.inv_ptr[-1]) == B * L * K
    rows = torch.randn(g.N * g.Kp, 128, device=d)
    out, _ = training._gather(rows, g, None)
    real = rows.view(g.N, g.Kp, 128)[:, :K].reshape(-1, 128).double()
    want = torch.zeros(g.N, 128, device=d, dtype=torch.float64).index_add_(0, g.dest, real)
    assert float((out.double() - want).abs().max()) < 1e-4
    bad = E_idx.clone()
    bad[1, 3, 2] = L
    with pytest.raises(ValueError, match="neighbour ids"):
        training.Graph(bad, None, None, None, K)


def test_gather_is_bit_reproducible_and_needs_no_zero_fill():
    """mpnn_gather_rows writes every output row once, in a fixed summation order (hub residues with several 64-entry
    chunks included): two runs into differently pre-filled outputs agree bit for bit."""
    from mpnn_amd import training
    from mpnn_amd._lib import call, ptr, stream_ptr

    d = "cuda"
    B, L, K = 2, 300, 48
    g0 = torch.Generator().manual_seed(11)
    w = 1.0 / (1.0 + torch.arange(L, dtype=torch.float64)) ** 0.9     # hubs: in-degrees far above 64
    E_idx = torch.stack([torch.stack([torch.multinomial(w, K, replacement=False, generator=g0) for _ in range(L)])
                         for _ in range(B)]).to(d)
    g = training.Graph(E_idx, None, None, None, K)
    deg = g.inv_ptr[1:] - g.inv_ptr[:-1]
    assert int(deg.max()) > 128 and int(deg.min()) < 64   # several chunks for the hubs, one for most
    rank = torch.stack([torch.randperm(L, generator=g0) for _ in range(B)]).reshape(-1).to(d).to(torch.int32)
    g.set_order(rank)
    rows = torch.randn(g.N * g.Kp, 128, device=d)
    outs = []
    for fill in (0.0, 7.0, float("nan")):
        outA = torch.full((g.N, 128), fill, device=d)
        outB = torch.full((g.N, 128), fill, device=d)
        part = torch.full((g.max_chunks * 256,), fill, device=d)
        call("mpnn_gather_rows", ptr(rows), ptr(g.inv_ptr), ptr(g.inv_row), ptr(g.sel), ptr(g.chunk_ptr),
             ptr(g.chunk_node), g.N, g.max_chunks, ptr(outA), ptr(outB), ptr(part), stream_ptr())
        outs.append((outA, outB))
    for a, b in outs[1:]:
        assert torch.equal(a, outs[0][0]) and torch.equal(b, outs[0][1])
    real = rows.view(g.N, g.Kp, 128)[:, :K].reshape(-1, 128).double()
    before = (rank.long()[g.src] > rank.long()[g.dest]).double().unsqueeze(-1)
    wantA = torch.zeros(g.N, 128, device=d, dtype=torch.float64).index_add_(0, g.dest, real * before)
    assert float((outs[0][0].double() - wantA).abs().max()) < 2e-4


@pytest.mark.parametrize("dropout,augment_eps", [(0.0, 0.0), (0.1, 0.1)])
def test_train_steps_reduce_loss(dropout, augment_eps):
    """Eight AdamW steps on one synthetic batch through ProteinMPNN.forward in training mode (the reference's defaults
    dropout = 0.1, augment_eps = 0.1 included): finite, decreasing loss; weights stay finite (train.py:375-392)."""
    from mpnn_amd import training
    from mpnn_amd.synthetic import synthetic_batch

    torch.manual_seed(0)
    b = synthetic_batch(B=4, L=96, seed=21, lengths=[96, 80, 64, 50], n_chains=2, design_fraction=0.8)
    model = build(dict(k_neighbors=48), dropout=dropout, augment_eps=augment_eps)
    # the reference's recipe when dropout is on: fused AdamW + warm-up/cosine schedule + gradient-norm clipping
    opt, sched = training.make_optimizer(model, lr=1e-3, total_steps=100) if dropout else \
        (torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-2), None)
    d = "cuda"
    batch = tuple(b[k].to(d) for k in ("X", "S", "mask", "chain_M", "residue_idx", "chain_encoding_all"))
    losses = []
    for _ in range(8):
        loss, lp = training.train_step(model, opt, batch, fix_order=b["fix_order"].to(d), scheduler=sched,
                                       max_grad_norm=1.0 if dropout else 0.0)
        losses.append(float(loss))
    _, nll, hits = training.loss_nll(batch[1], lp, batch[2] * batch[3])
    assert np.isfinite(float(nll)) and hits.shape == (4, 96)
    assert all(np.isfinite(losses)), losses
    assert min(losses[-2:]) < losses[0], losses
    assert all(torch.isfinite(p).all() for p in model.parameters())
    # the nn.Module surface takes the same path: forward in training mode returns a differentiable tensor
    lp = model(*batch, fix_order=b["fix_order"].to(d))
    assert lp.requires_grad and lp.shape == (4, 96, 21)


def _dropout_keep_mask(N, K, p, seed, device):
    """The kernels' counter-based keep decision (mpnn_edge_common.h dropout_keep), replicated with integer tensor ops:
    element id = ((node * 8 + t) * KT + c) * 256 + (16 g + col) * 4 + r for edge e = 16 c + col, feature f = 16 t + 4 g + r."""
    KT = (K + 15) // 16
    node = torch.arange(N, device=device).view(N, 1, 1)
    e = torch.arange(K, device=device).view(1, K, 1)
    f = torch.arange(128, device=device).view(1, 1, 128)
    t, g, r = f // 16, (f % 16) // 4, f % 4
    c, col = e // 16, e % 16
    M = 0xFFFFFFFF
    idx = (((node * 8 + t) * KT + c) * 256 + (16 * g + col) * 4 + r) & M
    h = ((idx * 0x9E3779B1) & M) ^ seed
    h = h ^ (h >> 16)
    h = (h * 0x85EBCA6B) & M
    h = h ^ (h >> 13)
    h = (h * 0xC2B2AE35) & M
    h = h ^ (h >> 16)
    return h >= int(p * 4294967296.0)


@pytest.mark.parametrize("p_drop", [0.0, 0.1, 0.5])
def test_edge_update_with_dropout_vs_aten(p_drop):
    """EncLayer edge update with dropout3 (:90-95): forward value and every gradient of the HIP Function against the
    same maths in ATen with the same keep mask; the dropped fraction is p."""
    import torch.nn.functional as F
    from mpnn_amd import engine, training

    torch.manual_seed(3)
    d = "cuda"
    B, L, K, seed = 2, 37, 40, 12345
    N = B * L
    E_idx = torch.stack([torch.stack([torch.randperm(L)[:K] if K <= L else torch.randint(0, L, (K,)) for _ in range(L)])
                         for _ in range(B)]).to(d)
    Kp = 48
    idx32 = torch.zeros(N, Kp, dtype=torch.int32, device=d)
    idx32[:, :K] = E_idx.reshape(N, K).to(torch.int32)
    g = training.Graph(E_idx, idx32.reshape(-1).contiguous(), torch.ones(N, device=d), None, K)
    leaf = lambda *s, sc=1.0: (sc * torch.randn(*s, device=d)).requires_grad_(True)
    hE_nat, P, Q = leaf(N, K, 128), leaf(N, 128), leaf(N, 128)
    W11b, W12, W13 = leaf(128, 128, sc=0.09), leaf(128, 128, sc=0.09), leaf(128, 128, sc=0.09)
    b12, b13, lnw, lnb = leaf(128, sc=0.1), leaf(128, sc=0.1), leaf(128), leaf(128, sc=0.1)
    params = [hE_nat, P, Q, W11b, W12, b12, W13, b13, lnw, lnb]
    upstream = torch.randn(N, K, 128, device=d) * 1e-4   # gradient-sized, like a real backward pass

    training._ZEROS.reset(torch.device(d, torch.cuda.current_device()))
    hE = training.FromNaturalFn.apply(g, hE_nat)
    out = training.EdgeUpdateFn.apply(g, training.GradSlot(), training.GradSlot(), p_drop, seed, hE, P, Q, W11b, W12, b12, W13, b13, lnw, lnb)
    out_nat = engine.edges_to_natural(out.detach(), N, K)
    out.backward(engine.edges_from_natural(upstream, N, K))
    got = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None

    keep = _dropout_keep_mask(N, K, p_drop, seed, d) if p_drop > 0 else torch.ones(N, K, 128, dtype=torch.bool, device=d)
    if p_drop > 0:
        assert abs(1.0 - keep.float().mean().item() - p_drop) < 0.01
    dest = g.dest.view(N, K)
    a1 = P.unsqueeze(1) + Q[dest] + F.linear(hE_nat, W11b)
    msg = F.linear(F.gelu(F.linear(F.gelu(a1), W12, b12)), W13, b13)
    msg = msg * keep / (1.0 - p_drop)
    want = F.layer_norm(hE_nat + msg, (128,), lnw, lnb, 1e-5)
    assert float((out_nat - want.detach()).abs().max()) < 2e-5
    want.backward(upstream)
    names = "hE P Q W11b W12 b12 W13 b13 ln_w ln_b".split()
    for n, a, p in zip(names, got, params):
        err = float((a.double() - p.grad.double()).norm() / (p.grad.double().norm() + 1e-30))
        assert err < 1e-4, f"d{n}: relative error {err:.2e}"


def test_eval_mode_with_grad_enabled_is_differentiable():
    """Fine-tuning in eval mode (no dropout, no backbone noise) must still reach the parameters: with grad mode on the
    module takes the differentiable path whatever its training flag; under no_grad it takes the inference kernels,
    and both give the same log-probs."""
    from mpnn_amd import training
    from mpnn_amd.synthetic import synthetic_batch

    kw = dict(k_neighbors=30)
    model = build(kw, dropout=0.1, augment_eps=0.1).eval()   # eval: the dropout / noise settings must not act
    b = synthetic_batch(2, 72, seed=12, lengths=[72, 60], n_chains=2)
    g = {k: v.cuda() for k, v in b.items()}
    args = (g["X"], g["S"], g["mask"], g["chain_M"], g["residue_idx"], g["chain_encoding_all"])
    lp = model(*args, fix_order=g["fix_order"])
    assert lp.requires_grad
    with torch.no_grad():
        lp_inf = model(*args, fix_order=g["fix_order"])
    assert not lp_inf.requires_grad
    assert (lp.detach() - lp_inf).abs().max().item() <= 1e-5
    _, loss = training.loss_smoothed(g["S"], lp, g["mask"] * g["chain_M"])
    loss.backward()
    got = {n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None}
    loss_ref, _, want = oracle_loss_and_grads(model, kw, b)
    assert abs(loss.item() - loss_ref) <= 1e-5 * max(1.0, abs(loss_ref))
    compare_grads({n: got.get(n, torch.zeros_like(v)) for n, v in want.items()}, want)


@pytest.mark.parametrize("N,n_letters", [(1, 21), (70, 21), (1000, 21), (130, 7)])
def test_head_backward_and_fused_loss_vs_aten(N, n_letters):
    """mpnn_logits_log_softmax_bwd and mpnn_loss_smoothed (through their autograd wrappers) against the ATen ops the
    reference uses (model_utils.py :527-528, train.py:63-73): values to 1e-5, gradients to 1e-5 relative."""
    import torch.nn.functional as F

    from mpnn_amd import training

    g = torch.Generator().manual_seed(N)
    hV = torch.randn(N, 128, generator=g).cuda().requires_grad_(True)
    W = (0.2 * torch.randn(n_letters, 128, generator=g)).cuda().requires_grad_(True)
    b = (0.1 * torch.randn(n_letters, generator=g)).cuda().requires_grad_(True)
    S = torch.randint(0, n_letters, (N,), generator=g).cuda()
    mask = (torch.rand(N, generator=g) < 0.8).float().cuda()

    lp = training.HeadFn.apply(hV, W, b)
    loss_vec, loss = training.loss_smoothed_fused(S, lp, mask)
    loss.backward()
    got = [t.grad.clone() for t in (hV, W, b)]
    for t in (hV, W, b):
        t.grad = None
    lp_ref = F.log_softmax(F.linear(hV, W, b), dim=-1)
    loss_vec_ref, loss_ref = training.loss_smoothed(S, lp_ref, mask)
    loss_ref.backward()
    want = [t.grad.clone() for t in (hV, W, b)]
    assert (lp - lp_ref).abs().max().item() <= 1e-5   # fp32 summation order over the 128 features
    assert (loss_vec - loss_vec_ref).abs().max().item() <= 1e-5
    assert abs(loss.item() - loss_ref.item()) <= 1e-6 + 1e-5 * abs(loss_ref.item())
    for name, a, r in zip(("d_hV", "dW", "db"), got, want):
        err = (a - r).norm().item() / max(r.norm().item(), 1e-30)
        assert err <= 1e-5, (name, err)


def test_inference_sees_weights_updated_by_a_fused_optimizer():
    """torch's fused AdamW does not bump Parameter._version, which keys the inference kernels' packed-weight cache:
    after a training step the eval-mode forward must use the NEW weights (checked against the oracle on them)."""
    from mpnn_amd import training
    from mpnn_amd.synthetic import synthetic_batch

    kw = dict(k_neighbors=30)
    model = build(kw)
    b = synthetic_batch(2, 64, seed=8, lengths=[64, 51], n_chains=2)
    g = {k: v.cuda() for k, v in b.items()}
    args = (g["X"], g["S"], g["mask"], g["chain_M"], g["residue_idx"], g["chain_encoding_all"])
    model.eval()
    with torch.no_grad():
        before = model(*args, fix_order=g["fix_order"])       # fills the packed-weight cache
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, fused=True)
    training.train_step(model, opt, args, fix_order=g["fix_order"])
    model.eval()
    with torch.no_grad():
        after = model(*args, fix_order=g["fix_order"])
    assert (after - before).abs().max().item() > 1e-3, "the step must be visible to the inference path"
    from oracle import mpnn_oracle as orc
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        ref = orc.forward(sd, oracle_cfg(kw), b["X"], b["S"], b["mask"], b["chain_M"], b["residue_idx"],
                          b["chain_encoding_all"], fix_order=b["fix_order"], formulation="edge")
    m = b["mask"].bool()
    err = (after.cpu() - ref).abs()[m]
    assert bool((err <= 1e-5 + 1e-4 * ref.abs()[m]).all()), err.max().item()


def test_autocast_context_does_not_retype_the_path():
    """The reference's loop may wrap forward + loss in torch.autocast(bfloat16) (train.py:380-384) and nothing stops a
    caller from running backward inside it too: outputs and gradients must be those of the plain call."""
    from mpnn_amd import training
    from mpnn_amd.synthetic import synthetic_batch

    model = build(dict(k_neighbors=30))
    b = synthetic_batch(2, 64, seed=4, lengths=[64, 50], n_chains=2)
    g = {k: v.cuda() for k, v in b.items()}
    args = (g["X"], g["S"], g["mask"], g["chain_M"], g["residue_idx"], g["chain_encoding_all"])

    def run(autocast):
        model.zero_grad(set_to_none=True)
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=autocast):
            lp = model(*args, fix_order=g["fix_order"])
            _, loss = training.loss_smoothed(g["S"], lp.float(), g["mask"] * g["chain_M"])
            loss.backward()
        return lp.detach(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    lp0, g0 = run(False)
    lp1, g1 = run(True)
    assert lp1.dtype == torch.float32 and torch.equal(lp0, lp1)
    assert g0.keys() == g1.keys()
    for n in g0:  # no sum of the backward pass meets by atomics: two runs agree bit for bit
        assert torch.equal(g0[n], g1[n]), n
    model.eval()
    with torch.no_grad(), torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        lp2 = model(*args, fix_order=g["fix_order"])
    with torch.no_grad():
        lp3 = model(*args, fix_order=g["fix_order"])
    assert torch.equal(lp2, lp3)


def test_energy_style_loss_gradients_vs_oracle():
    """EnergyMPNN's use of the model (energy_mpnn.py:42-68, finetune loop): one structure repeated for B mutant
    sequences, dG = sum(one_hot(S) * log_probs), a regression loss on dG - gradients of an arbitrary function of the
    log-probs, not of loss_smoothed, against autograd through the oracle."""
    import torch.nn.functional as F

    from mpnn_amd.synthetic import synthetic_batch
    from oracle import mpnn_oracle as orc

    kw = dict(k_neighbors=32)
    model = build(kw).eval()                      # fine-tuning without dropout / noise; grad mode is on
    one = synthetic_batch(1, 60, seed=17, n_chains=2)
    B = 4
    gen = torch.Generator().manual_seed(3)
    S = torch.randint(0, 21, (B, 60), generator=gen)
    order = torch.argsort(torch.rand(B, 60, generator=gen), dim=-1)
    target = torch.randn(B, generator=gen)
    rep = {k: one[k].repeat(B, *([1] * (one[k].dim() - 1))) for k in ("X", "mask", "chain_M", "residue_idx",
                                                                      "chain_encoding_all")}

    def dG_loss(lp, S_):
        dG = torch.sum(F.one_hot(S_, 21).float() * lp, dim=(1, 2))
        return torch.mean((dG - target.to(lp.device)) ** 2)

    model.zero_grad(set_to_none=True)
    lp = model(X=rep["X"].cuda(), S=S.cuda(), mask=rep["mask"].cuda(), chain_M=rep["chain_M"].cuda(),
               residue_idx=rep["residue_idx"].cuda(), chain_encoding_all=rep["chain_encoding_all"].cuda(),
               fix_order=order.cuda())
    loss = dG_loss(lp, S.cuda())
    loss.backward()
    got = {n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None}

    sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    lp_ref = orc.forward(sd, oracle_cfg(kw), rep["X"], S, rep["mask"], rep["chain_M"], rep["residue_idx"],
                         rep["chain_encoding_all"], fix_order=order, formulation="edge")
    loss_ref = dG_loss(lp_ref, S)
    loss_ref.backward()
    want = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in sd.items() if v.requires_grad}
    assert abs(loss.item() - loss_ref.item()) <= 1e-4 * max(1.0, abs(loss_ref.item()))
    compare_grads({n: got.get(n, torch.zeros_like(v)) for n, v in want.items()}, want)


@pytest.mark.parametrize("n_enc", [3, 0])
def test_second_backward_over_the_same_graph(n_enc):
    """retain_graph=True and a second backward: the shared per-edge gradient buffers must start afresh (with no encoder
    every decoder message kernel accumulates into ONE buffer that the featurizer's backward consumes)."""
    from mpnn_amd import training
    from mpnn_amd.synthetic import synthetic_batch

    kw = dict(k_neighbors=24, num_encoder_layers=n_enc)
    b = {k: v.cuda() for k, v in synthetic_batch(B=2, L=40, seed=31, lengths=[40, 29], n_chains=2).items()}
    model = build(kw)
    lp = training.forward_train(model, b["X"], b["S"], b["mask"], b["chain_M"], b["residue_idx"],
                                b["chain_encoding_all"], fix_order=b["fix_order"])
    _, loss = training.loss_smoothed(b["S"], lp, b["mask"] * b["chain_M"])
    loss.backward(retain_graph=True)
    first = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    loss.backward()
    for n, p in model.named_parameters():
        assert torch.allclose(p.grad, first[n], rtol=1e-5, atol=1e-9), n


def test_backward_is_bit_reproducible():
    """run twice, bit-compare.  No sum of the training step meets by floating-point atomics (the
    neighbour-gather gradient, weight / bias / LayerNorm-parameter gradients, the sequence-term table, the loss): two
    forward + backward passes over the same batch give identical losses, log-probs and `.grad`s, on a ragged
    multi-chain batch with hub residues and with the optional featurizer branches on."""
    from mpnn_amd import training
    from mpnn_amd.synthetic import synthetic_batch

    for kw, bkw in ((dict(k_neighbors=48), dict(B=3, L=160, seed=23, lengths=[160, 131, 77], n_chains=2, design_fraction=0.6)),
                    (dict(k_neighbors=30, edge_cutoff=12.0, occupancy_cutoff=10.0, use_virtual_center=True),
                     dict(B=2, L=72, seed=5, lengths=[72, 50], n_chains=3, design_fraction=0.6))):
        b = synthetic_batch(**bkw)
        model = build(kw)
        runs = [hip_loss_and_grads(model, b) for _ in range(2)]
        assert runs[0][0] == runs[1][0]
        assert torch.equal(runs[0][1], runs[1][1])
        for n in runs[0][2]:
            assert torch.equal(runs[0][2][n], runs[1][2][n]), n
    # and the clipped optimizer step that follows: same parameters after two identical steps from identical states
    b = synthetic_batch(B=2, L=96, seed=3, lengths=[96, 64], n_chains=2)
    d = "cuda"
    batch = tuple(b[k].to(d) for k in ("X", "S", "mask", "chain_M", "residue_idx", "chain_encoding_all"))
    after = []
    for _ in range(2):
        model = build(dict(k_neighbors=32))
        opt, _ = training.make_optimizer(model, lr=1e-3, max_grad_norm=0.05)
        for _ in range(2):
            training.train_step(model, opt, batch, fix_order=b["fix_order"].to(d))
        after.append({n: p.detach().clone() for n, p in model.named_parameters()})
    for n in after[0]:
        assert torch.equal(after[0][n], after[1][n]), n


def test_fused_backward_kernels_match_the_unfused_path():
    """mpnn_message_bwd_fused / mpnn_enc_edge_update_bwd_fused (K <= 48: chain + weight-gradient contractions in one
    kernel, operand scales from bounds) against mpnn_message_bwd / mpnn_enc_edge_update_bwd + mpnn_edge_wgrad (exact
    per-tensor maxima) on one model and batch: same loss, every gradient within 2e-5 relative L2 (both are fp32-class
    split-f16 evaluations of the same sums in different orders), and the fused entry points are the ones that ran."""
    from mpnn_amd import _lib, training
    from mpnn_amd.synthetic import synthetic_batch

    kw = dict(k_neighbors=48)
    b = synthetic_batch(B=3, L=96, seed=21, lengths=[96, 80, 57], n_chains=2, design_fraction=0.6)
    model = build(kw)
    seen = {}
    saved = (training.FUSED_MSG_BWD, training.FUSED_UPD_BWD)
    try:
        res = {}
        for tag, flag in (("fused", True), ("unfused", False)):
            training.FUSED_MSG_BWD = training.FUSED_UPD_BWD = flag
            prof = {}
            _lib.set_profiler(prof)
            res[tag] = hip_loss_and_grads(model, b)
            _lib.set_profiler(None)
            seen[tag] = set(prof)
    finally:
        _lib.set_profiler(None)
        training.FUSED_MSG_BWD, training.FUSED_UPD_BWD = saved
    assert {"mpnn_message_bwd_fused", "mpnn_enc_edge_update_bwd_fused"} <= seen["fused"]
    assert not ({"mpnn_message_bwd", "mpnn_enc_edge_update_bwd"} & seen["fused"])
    assert {"mpnn_message_bwd", "mpnn_enc_edge_update_bwd"} <= seen["unfused"]
    assert not ({"mpnn_message_bwd_fused", "mpnn_enc_edge_update_bwd_fused"} & seen["unfused"])
    (lf, _, gf), (lu, _, gu) = res["fused"], res["unfused"]
    assert lf == lu                      # the forward pass is the same kernels
    total = sum(float(v.double().norm() ** 2) for v in gu.values()) ** 0.5
    for n in gu:
        err = float((gf[n].double() - gu[n].double()).norm())
        assert err <= 2e-5 * float(gu[n].double().norm()) + 1e-7 * total, (n, err, float(gu[n].double().norm()))


def test_fused_edge_update_backward_scale_bound_holds_for_extreme_layernorms():
    """The fused edge-update backward scales its f16 operand splits from BOUNDS (13.3 rstd_max max|ln_w| max|dy|, then
    1.13 ||W^T|| per GEMM).  A bound that were too small would overflow the f16 range (inf / NaN gradients), one that is
    far too large would starve the lo halves: near-constant edge rows (rstd close to 1 / sqrt(eps)) and large LayerNorm
    weights must still give the unfused path's gradients."""
    from mpnn_amd import engine, training

    torch.manual_seed(5)
    d = "cuda"
    B, L, K = 1, 40, 30
    N = B * L
    E_idx = torch.stack([torch.stack([torch.randperm(L)[:K] for _ in range(L)]) for _ in range(B)]).to(d)
    idx32 = torch.zeros(N, 32, dtype=torch.int32, device=d)
    idx32[:, :K] = E_idx.reshape(N, K).to(torch.int32)
    g = training.Graph(E_idx, idx32.reshape(-1).contiguous(), torch.ones(N, device=d), None, K)
    mk = lambda *s, sc=1.0: (sc * torch.randn(*s, device=d))
    base = dict(hE=mk(N, K, 128, sc=1e-3) + 0.7, P=mk(N, 128, sc=1e-3), Q=mk(N, 128, sc=1e-3),
                W11b=mk(128, 128, sc=0.02), W12=mk(128, 128, sc=0.02), b12=mk(128, sc=0.01), W13=mk(128, 128, sc=0.02),
                b13=mk(128, sc=0.01), lnw=mk(128) * 20.0, lnb=mk(128, sc=0.1))
    upstream = engine.edges_from_natural(torch.randn(N, K, 128, device=d) * 3e-5, N, K)
    saved = training.FUSED_UPD_BWD
    out = {}
    try:
        for flag in (True, False):
            training.FUSED_UPD_BWD = flag
            training._ZEROS.reset(torch.device(d, torch.cuda.current_device()))
            p = {k: v.clone().requires_grad_(True) for k, v in base.items()}
            hE = training.FromNaturalFn.apply(g, p["hE"])
            y = training.EdgeUpdateFn.apply(g, training.GradSlot(), training.GradSlot(), 0.0, 0, hE, p["P"], p["Q"], p["W11b"],
                                            p["W12"], p["b12"], p["W13"], p["b13"], p["lnw"], p["lnb"])
            y.backward(upstream)
            out[flag] = {k: v.grad.clone() for k, v in p.items()}
    finally:
        training.FUSED_UPD_BWD = saved
    for k in base:
        a, w = out[True][k].double(), out[False][k].double()
        assert bool(torch.isfinite(a).all()), k
        assert float((a - w).norm()) <= 1e-4 * float(w.norm()) + 1e-12, (k, float((a - w).norm()), float(w.norm()))


@pytest.mark.parametrize("n_letters,N", [(21, 777), (32, 64)])
def test_sequence_term_kernels_vs_aten(n_letters, N):
    """SeqTermFn (mpnn_seq_table / mpnn_add_table_rows / mpnn_seq_table_bwd; reference :478-483, h_S = W_s(S) entering the
    third column block of the decoder's W1): A = A0 + (W_s W1c^T)[S] and the gradients of A0, W_s and W1c against fp64
    ATen autograd.  W1c is a strided column block of a [128, 512] weight, as in the model."""
    from mpnn_amd import training as T

    gen = torch.Generator().manual_seed(5)
    W1 = (torch.randn(128, 512, generator=gen) * 0.1).cuda().requires_grad_(True)
    Ws = torch.randn(n_letters, 128, generator=gen).cuda().requires_grad_(True)
    A0 = torch.randn(N, 128, generator=gen).cuda().requires_grad_(True)
    S = torch.randint(0, n_letters, (N,), generator=gen).cuda()
    dA = (torch.randn(N, 128, generator=gen) * 1e-4).cuda()      # gradient-sized
    onehot = torch.nn.functional.one_hot(S, 32 * ((n_letters + 31) // 32)).float()
    T._ZEROS.reset(torch.device("cuda", 0))
    A = T.SeqTermFn.apply(A0, Ws, W1.split(128, dim=1)[2], S, onehot)
    A.backward(dA)
    got = [A.detach().double().cpu(), A0.grad.double().cpu(), Ws.grad.double().cpu(), W1.grad.double().cpu()]
    W1d, Wsd, A0d = [t.detach().double().cpu().requires_grad_(True) for t in (W1, Ws, A0)]
    Ad = A0d + (Wsd @ W1d[:, 256:384].t())[S.cpu()]
    Ad.backward(dA.double().cpu())
    want = [Ad.detach(), A0d.grad, Wsd.grad, W1d.grad]
    for name, g_, w_ in zip(("A", "dA0", "dW_s", "dW1"), got, want):
        err = (g_ - w_).norm() / max(w_.norm().item(), 1e-30)
        assert err < 2e-6, f"{name}: relative error {err:.2e}"
    assert torch.equal(W1.grad[:, :256], torch.zeros_like(W1.grad[:, :256]))   # only the third column block moves


@pytest.mark.parametrize("kw,L", [(dict(k_neighbors=48), 96), (dict(k_neighbors=30, use_virtual_center=True), 64),
                                  (dict(k_neighbors=64), 80)])
def test_feature_wgrad_recomputes_the_chunks(kw, L):
    """mpnn_feature_wgrad (the featurizer's weight gradient with the raw features computed inside the contraction) against
    the materialised form it replaces: mpnn_feature_chunks followed by one mpnn_edge_wgrad per chunk.  Same feature
    arithmetic, same contraction kernel, same order of partial sums: equal to rounding (the chunk kernel is compiled
    without FMA contraction, as the whole featurizer source is for the neighbour search; the squared distances differ in
    the last bit)."""
    import mpnn_amd
    from mpnn_amd import engine, training as T
    from mpnn_amd._lib import call, ptr, stream_ptr
    from mpnn_amd.synthetic import synthetic_batch

    dev = torch.device("cuda", 0)
    model = mpnn_amd.ProteinMPNN(augment_eps=0.0, dropout=0.0, **kw).to(dev)
    b = synthetic_batch(B=2, L=L, seed=9, lengths=[L, L - 7], n_chains=2)
    X, mask = b["X"].to(dev), b["mask"].to(dev).float()
    ridx, chain = b["residue_idx"].to(dev), b["chain_encoding_all"].to(dev)
    with torch.no_grad():
        atoms, _, E64, E32, _, emask, K = engine._geometry(model.features, X, mask, None, False)
    g = T.Graph(E64, E32, mask.reshape(-1).contiguous(), emask, K)
    na = 6 if kw.get("use_virtual_center") else 5
    nch = (na * na + 5 + 7) // 8
    gen = torch.Generator().manual_seed(3)
    dE0 = (torch.randn(g.frag_numel(), generator=gen) * 1e-4).to(dev)
    amax = dE0.abs().max().reshape(1)
    mu = torch.linspace(2.0, 22.0, 16, device=dev)
    T._ZEROS.reset(dev)
    ws = T._workspace(dev)
    st = stream_ptr()
    dF = torch.zeros(nch, 128, 128, device=dev)
    db = torch.zeros(128, device=dev)
    call("mpnn_feature_wgrad", ptr(dE0), ptr(amax), ptr(atoms), ptr(g.E_idx32), ptr(ridx), ptr(chain), ptr(mu), g.B,
         g.L, g.K, na, ptr(dF), ptr(db), ptr(ws), ws.numel(), st)
    chunks = torch.empty(nch * g.frag_numel(), device=dev)
    call("mpnn_feature_chunks", ptr(atoms), ptr(g.E_idx32), ptr(ridx), ptr(chain), ptr(mu), g.B, g.L, g.K, na,
         ptr(chunks), st)
    n = g.frag_numel()
    for c in range(nch):
        dW, dbc = T._wgrad(dE0, False, amax, chunks[c * n:(c + 1) * n], False, g, c == 0)
        err = (dW - dF[c]).abs().max().item()
        assert err <= 1e-5 * max(dW.abs().max().item(), 1e-30), f"chunk {c}: max |d| {err:.3e}"
        if c == 0:
            assert torch.equal(dbc, db)
    assert dF.abs().max().item() > 0


def test_node_update_backward_with_an_unused_projection():
    """NodeUpdateFn when one of its projections (and, separately, h' itself) receives no gradient: the backward kernel gets
    a NULL entry in its pointer table (mpnn_node_update_bwd_each) instead of a zero tensor.  Gradients equal those of the
    same graph with an explicit zero contribution."""
    import mpnn_amd
    from mpnn_amd import training as T

    dev = torch.device("cuda", 0)
    torch.manual_seed(4)
    model = mpnn_amd.ProteinMPNN(k_neighbors=16, augment_eps=0.0, dropout=0.0).to(dev)
    lyr = model.encoder_layers[0]
    N = 80
    hV = torch.randn(N, 128, device=dev)
    gsum = torch.randn(N, 128, device=dev, requires_grad=True)
    cnt = torch.full((N,), 16.0, device=dev)
    mask = torch.ones(N, device=dev)
    Wa = (torch.randn(128, 128, device=dev) * 0.05).requires_grad_(True)
    Wb = (torch.randn(128, 128, device=dev) * 0.05).requires_grad_(True)
    ba = torch.randn(128, device=dev, requires_grad=True)

    def run(use_b, use_h):
        for p in list(lyr.parameters()) + [Wa, Wb, ba, gsum]:
            p.grad = None
        T._ZEROS.reset(dev)
        h, (pa, pb) = T._node_update(lyr, hV, gsum, cnt, mask, 0.0, [(Wa, ba), (Wb, None)])
        loss = (pa * pa).sum()
        if use_b:
            loss = loss + (pb * 0.0).sum()   # a zero gradient, delivered as a tensor
        if use_h:
            loss = loss + (h * 0.0).sum()
        loss.backward()
        return {n: (p.grad.clone() if p.grad is not None else None)
                for n, p in [("gsum", gsum), ("Wa", Wa), ("ba", ba), ("Wb", Wb), ("W3", lyr.W3.weight),
                             ("Win", lyr.dense.W_in.weight), ("ln2", lyr.norm2.weight)]}

    ref = run(True, True)         # every output receives a (zero) gradient tensor
    for use_b, use_h in ((False, True), (False, False)):
        got = run(use_b, use_h)
        for n in ref:
            if n == "Wb":
                assert got[n] is None or float(got[n].abs().max()) == 0.0
                continue
            assert torch.equal(got[n], ref[n]), n
