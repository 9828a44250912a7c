"""GPU tests of the autoregressive sampler (BASELINE configs[4]).  The reference has no sampler; the contract
 is: a teacher-forced forward on the sampled sequence with the same decoding order returns, at every
residue, the log-probs the sampler drew from.  The teacher-forced HIP forward is itself pinned to the oracle and the
reference's golden vectors in test_gpu_parity.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def build(kw):
    import mpnn_amd
    from mpnn_amd.synthetic import fill_parameters_

    model = mpnn_amd.ProteinMPNN(augment_eps=0.0, dropout=0.0, **kw).eval()
    fill_parameters_(model, seed=0)
    return model.to("cuda")


CASES = {
    "k48_vc": (dict(k_neighbors=48, use_virtual_center=True), dict(B=2, L=96, seed=3, lengths=[96, 70], n_chains=2,
                                                                   design_fraction=0.7), 3),
    "k30_cutoff": (dict(k_neighbors=30, edge_cutoff=14.0), dict(B=1, L=130, seed=8, n_chains=1, design_fraction=1.0), 4),
    "k16_short": (dict(k_neighbors=16), dict(B=3, L=40, seed=5, lengths=[40, 31, 12], n_chains=2, design_fraction=0.5), 2),
    # four column tiles: message / node update / head launches per level (the fused level kernels stop at K = 48)
    "k64": (dict(k_neighbors=64), dict(B=2, L=90, seed=13, lengths=[90, 75], n_chains=2, design_fraction=0.8), 2),
}


@pytest.mark.parametrize("name", list(CASES))
def test_sampler_matches_teacher_forced_forward(name):
    from mpnn_amd.synthetic import synthetic_batch

    kw, bkw, n_samples = CASES[name]
    model = build(kw)
    b = {k: v.to("cuda") for k, v in synthetic_batch(**bkw).items()}
    out = model.sample(b["X"], b["mask"], b["chain_M"], b["residue_idx"], b["chain_encoding_all"], S_true=b["S"],
                       n_samples=n_samples, temperature=0.5, seed=11)
    S, lp, order = out["S"], out["log_probs"], out["order"]
    assert S.shape == (n_samples,) + tuple(b["S"].shape) and int(S.min()) >= 0 and int(S.max()) < 21
    fixed = (b["chain_M"] * b["mask"]) == 0
    for s in range(n_samples):
        assert torch.equal(S[s][fixed], b["S"][fixed]), "non-designed residues keep their input letter"
        with torch.no_grad():
            ref = model(b["X"], S[s], b["mask"], b["chain_M"], b["residue_idx"], b["chain_encoding_all"],
                        fix_order=order[s])
        err = float((ref - lp[s]).abs().max())
        assert err <= 1e-5, f"sample {s}: teacher-forced log-probs differ from the sampler's by {err:.2e}"
    # different samples use different orders and draws
    if n_samples > 1:
        assert not torch.equal(order[0], order[1])
    print(f"{name}: {out['n_levels']} dependency levels for L={S.shape[-1]}")


def _oracle_log_probs(model, kw, b, S, order):
    """Teacher-forced forward of the CPU oracle (reference model_utils.py:427-529 with fix_order) on one sample."""
    from conftest import oracle_cfg
    from oracle import mpnn_oracle as orc

    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        return orc.forward(sd, oracle_cfg(kw), b["X"], S.cpu(), b["mask"], b["chain_M"], b["residue_idx"],
                           b["chain_encoding_all"], fix_order=order.cpu(), formulation="edge")


def _assert_lp_close(lp, ref, what):
    err = (lp.cpu().double() - ref.double()).abs()
    tol = 1e-5 + 1e-4 * ref.double().abs()   # the forward's bar: 1e-4 relative + 1e-5 absolute
    assert bool((err <= tol).all()), f"{what}: max |d log p| {float(err.max()):.3e} vs the oracle"


@pytest.mark.parametrize("name", list(CASES))
def test_sampler_matches_oracle(name):
    """SURVEY 8 A14's contract against the ORACLE (not the HIP forward): the reference's teacher-forced forward on the
    sampled sequence with the sampler's decoding order returns the log-probs the sampler drew from."""
    from mpnn_amd.synthetic import synthetic_batch

    kw, bkw, n_samples = CASES[name]
    model = build(kw)
    b = synthetic_batch(**bkw)
    g = {k: v.to("cuda") for k, v in b.items()}
    out = model.sample(g["X"], g["mask"], g["chain_M"], g["residue_idx"], g["chain_encoding_all"], S_true=g["S"],
                       n_samples=n_samples, temperature=0.5, seed=11)
    for s in range(n_samples):
        ref = _oracle_log_probs(model, kw, b, out["S"][s], out["order"][s])
        _assert_lp_close(out["log_probs"][s], ref, f"{name} sample {s}")


def test_sampler_configs4_matches_oracle_at_size():
    """BASELINE configs[4]: one L = 512 structure, virtual centre on, k = 48, 64 samples.  The oracle's teacher-forced
    forward is run on three of the 64 samples (first, middle, last); every sample must be a valid sequence."""
    from mpnn_amd.synthetic import synthetic_batch

    kw = dict(k_neighbors=48, use_virtual_center=True)
    model = build(kw)
    b = synthetic_batch(B=1, L=512, seed=21, design_fraction=1.0)
    g = {k: v.to("cuda") for k, v in b.items()}
    out = model.sample(g["X"], g["mask"], g["chain_M"], g["residue_idx"], g["chain_encoding_all"], n_samples=64,
                       temperature=0.1, seed=4)
    S, lp, order = out["S"], out["log_probs"], out["order"]
    assert S.shape == (64, 1, 512) and int(S.min()) >= 0 and int(S.max()) < 21
    assert bool(torch.isfinite(lp).all())
    assert bool((torch.sort(order, dim=-1).values == torch.arange(512, device="cuda")).all())
    assert len({tuple(o.tolist()) for o in order[:, 0, :8].cpu()}) > 1, "samples draw their own decoding orders"
    for s in (0, 31, 63):
        ref = _oracle_log_probs(model, kw, b, S[s], order[s])
        _assert_lp_close(lp[s], ref, f"configs[4] sample {s}")


def test_low_temperature_is_argmax_and_seed_reproducible():
    from mpnn_amd.synthetic import synthetic_batch

    model = build(dict(k_neighbors=32))
    b = {k: v.to("cuda") for k, v in synthetic_batch(B=2, L=64, seed=2, design_fraction=1.0).items()}
    args = (b["X"], b["mask"], b["chain_M"], b["residue_idx"], b["chain_encoding_all"])
    a = model.sample(*args, n_samples=2, temperature=1e-4, seed=5)
    designed = ((b["chain_M"] * b["mask"]) > 0).unsqueeze(0).expand_as(a["S"])
    # where the two best letters are further apart than 100 temperatures the draw IS the argmax (a closer runner-up
    # keeps a real share of the distribution: exp(-gap / T))
    top2 = a["log_probs"].topk(2, dim=-1).values
    clear = designed & ((top2[..., 0] - top2[..., 1]) > 1e-2)
    assert float(clear.sum()) > 0.8 * float(designed.sum())
    assert torch.equal(a["S"][clear], a["log_probs"].argmax(-1)[clear])
    a2 = model.sample(*args, n_samples=2, temperature=1e-4, seed=5)
    assert torch.equal(a["S"], a2["S"]) and torch.equal(a["order"], a2["order"])
    c = model.sample(*args, n_samples=2, temperature=1.0, seed=6)
    assert not torch.equal(c["S"], a["S"])


@pytest.mark.parametrize("mode", ["run", "run/head", "layer", "layer/head", "split"])
def test_every_launch_mode_is_bit_reproducible(monkeypatch, mode):
    """No sum of the sampler meets by atomics (a residue's column-tile sums are added in tile order, in LDS or from
    per-tile rows in HBM): the same call twice returns the same sequences AND the same log-probs, bit for bit, in
    every launch mode; k = 56 takes the four-tile split launches whatever the mode."""
    from mpnn_amd.synthetic import synthetic_batch

    monkeypatch.setenv("MPNN_SAMPLE_LAUNCHES", mode)
    for k in (48, 56):
        model = build(dict(k_neighbors=k, use_virtual_center=True))
        b = {n: v.to("cuda") for n, v in synthetic_batch(B=2, L=90, seed=4, lengths=[90, 71], n_chains=2,
                                                         design_fraction=0.7).items()}
        args = (b["X"], b["mask"], b["chain_M"], b["residue_idx"], b["chain_encoding_all"])
        a = model.sample(*args, S_true=b["S"], n_samples=6, temperature=0.5, seed=11)
        c = model.sample(*args, S_true=b["S"], n_samples=6, temperature=0.5, seed=11)
        assert torch.equal(a["S"], c["S"]) and torch.equal(a["log_probs"], c["log_probs"]), (mode, k)


def test_draw_frequencies_follow_the_model_distribution():
    """Residues decoded first see no sequence context: across samples with a FIXED order their letters are i.i.d. draws
    from one known distribution; compare frequencies over 4096 draws (256 samples x 16 first residues of equal
    context would not be identical, so pool per residue and test each with a loose chi-square bound)."""
    from mpnn_amd.synthetic import synthetic_batch

    model = build(dict(k_neighbors=24))
    b = {k: v.to("cuda") for k, v in synthetic_batch(B=1, L=48, seed=12, design_fraction=1.0).items()}
    n = 2048
    out = model.sample(b["X"], b["mask"], b["chain_M"], b["residue_idx"], b["chain_encoding_all"], n_samples=n,
                       temperature=1.0, seed=3, fix_order=b["fix_order"])
    first = int(b["fix_order"][0, 0])
    p = out["log_probs"][0, 0, first].exp().double().cpu().numpy()
    assert float((out["log_probs"][:, 0, first] - out["log_probs"][0, 0, first]).abs().max()) < 1e-6
    counts = np.bincount(out["S"][:, 0, first].cpu().numpy(), minlength=21).astype(np.float64)
    chi2 = float((((counts - n * p) ** 2) / (n * p + 1e-12)).sum())
    assert chi2 < 60.0, f"chi-square {chi2:.1f} over 20 degrees of freedom (99.999 % quantile ~ 52.4 + slack)"


def test_launch_modes_agree(monkeypatch):
    """The sampler's launch forms - the whole level loop in one library call (mpnn_decode_run, the default), the same
    launches issued from Python, the draw fused into the last layer's launch or in a launch of its own, and the
    message / node update / head form - run the same arithmetic; a residue's three column-tile sums are added in tile
    order everywhere, so the fused forms agree bit for bit and the split form to rounding (its tiles meet after a
    trip through HBM in another association).  Two structures, ragged, so that levels span the
    residues-per-workgroup choices (n_samples 2 and 40)."""
    from mpnn_amd.synthetic import synthetic_batch

    model = build(dict(k_neighbors=48, use_virtual_center=True))
    b = {k: v.to("cuda") for k, v in synthetic_batch(B=2, L=120, seed=9, lengths=[120, 83], n_chains=2,
                                                     design_fraction=0.6).items()}
    args = (b["X"], b["mask"], b["chain_M"], b["residue_idx"], b["chain_encoding_all"])
    for n_samples in (2, 40):
        outs = {}
        for mode in ("run", "run/head", "layer", "layer/head", "split"):
            monkeypatch.setenv("MPNN_SAMPLE_LAUNCHES", mode)
            outs[mode] = model.sample(*args, S_true=b["S"], n_samples=n_samples, temperature=0.3, seed=7)
        for other in ("run/head", "layer", "layer/head"):
            assert torch.equal(outs["run"]["S"], outs[other]["S"]), other
            assert float((outs["run"]["log_probs"] - outs[other]["log_probs"]).abs().max()) <= 1e-5, other
        same = (outs["run"]["S"] == outs["split"]["S"]).float().mean()
        assert float(same) > 0.99  # a draw can flip where u lands within an ulp of a CDF step
        agree = (outs["run"]["S"] == outs["split"]["S"]).all(-1).all(-1)  # samples drawn identically
        assert bool(agree.any())
        d = (outs["run"]["log_probs"][agree] - outs["split"]["log_probs"][agree]).abs().max()
        assert float(d) <= 1e-5
    monkeypatch.setenv("MPNN_SAMPLE_LAUNCHES", "level")
    with pytest.raises(ValueError, match="MPNN_SAMPLE_LAUNCHES"):
        model.sample(*args, S_true=b["S"], n_samples=2, temperature=0.3, seed=7)
