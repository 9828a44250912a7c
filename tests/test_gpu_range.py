"""Operand range of the f16-split matrix-pipe arithmetic (include/mpnn_hip.h, OPERAND RANGE): large and tiny weights
against the oracle, and the guards that turn an out-of-range operand into an error instead of silent garbage."""
import pytest
import torch

from test_gpu_parity import LP_ATOL, LP_RTOL, assert_close, build_model, run_hip, run_oracle, same

pytestmark = pytest.mark.gpu


def _batch():
    from mpnn_amd.synthetic import synthetic_batch

    b = synthetic_batch(B=2, L=72, seed=41, lengths=[72, 55], n_chains=2, design_fraction=0.6)
    return {n: v.numpy() for n, v in b.items()}


def _scale(model, pick, factor):
    with torch.no_grad():
        for n, p in model.named_parameters():
            if pick(n):
                p.mul_(factor)
    return model


def test_layernorm_weights_times_50_match_the_oracle():
    """Activations 50x the usual size (every LayerNorm gain x 50: |h_E|, |h_V| up to a few hundred, pre-activations in
    the thousands): still inside the f16 range, and the split keeps fp32-class accuracy there."""
    kw = dict(k_neighbors=48)
    model = _scale(build_model(kw), lambda n: "norm" in n and n.endswith("weight"), 50.0)
    rec = _batch()
    out, ref = run_hip(model, rec), run_oracle(model, kw, rec)
    same(out["E_idx"].cpu().numpy(), ref["E_idx"].numpy(), "E_idx")
    m = rec["mask"].astype(bool)
    big = float(ref["h_E_enc"][-1].abs().max())
    assert big > 100.0, big
    for l in range(3):   # tolerances relative to the 50x larger values
        assert_close(out["h_V_enc"][l].cpu().numpy()[m], ref["h_V_enc"][l].numpy()[m], 1e-4, 50 * 5e-5, f"enc{l} h_V")
    assert_close(out["log_probs"].cpu().numpy()[m], ref["log_probs"].numpy()[m], LP_RTOL, 10 * LP_ATOL, "log_probs")
    model.check_numerics()


def test_tiny_weights_match_the_oracle():
    """Every matrix scaled to ~1e-6: operands far below the f16 normal range keep only 3e-8 ABSOLUTE resolution, which
    is what matters for the sums they enter; log-probs stay within the usual bar of the oracle."""
    kw = dict(k_neighbors=32)
    model = _scale(build_model(kw), lambda n: n.endswith("weight") and "norm" not in n and "W_s" not in n, 1e-5)
    rec = _batch()
    out, ref = run_hip(model, rec), run_oracle(model, kw, rec)
    m = rec["mask"].astype(bool)
    assert_close(out["log_probs"].cpu().numpy()[m], ref["log_probs"].numpy()[m], LP_RTOL, LP_ATOL, "log_probs")
    for l in range(3):
        assert_close(out["h_V_dec"][l].cpu().numpy()[m], ref["h_V_dec"][l].numpy()[m], 1e-4, 5e-5, f"dec{l} h_V")


def test_weight_beyond_f16_range_is_refused_when_packed():
    model = build_model(dict(k_neighbors=16))
    with torch.no_grad():
        model.encoder_layers[1].W2.weight[3, 5] = 7.0e4
    rec = _batch()
    with pytest.raises(RuntimeError, match="65504"):
        run_hip(model, rec, keep=False)
    with torch.no_grad():
        model.encoder_layers[1].W2.weight[3, 5] = float("nan")
    with pytest.raises(RuntimeError):
        run_hip(model, rec, keep=False)


def _overflowing_model():
    model = _scale(build_model(dict(k_neighbors=16)), lambda n: "dense.W_in.weight" in n or "W1.weight" in n, 2.0e3)
    return _scale(model, lambda n: "dense.W_out.weight" in n or n.endswith("W2.weight"), 2.0e3)


def test_activation_overflow_is_reported_not_silent():
    """Weights in range, activations not (FFN output weights x 3e4 push h_V past 131008 before the LayerNorm): the
    outputs are inf / NaN, and the wrapper says so - check_numerics() raises, naming the call by its number."""
    model = _overflowing_model()
    rec = _batch()
    out = run_hip(model, rec, keep=False)
    assert not bool(torch.isfinite(out).all()), "this model must overflow the f16-split range"
    with pytest.raises(RuntimeError, match=r"call #1 .*inf / NaN"):
        model.check_numerics()
    model.check_numerics()                       # reported once, then clean


def test_bad_call_does_not_abort_the_next_good_call():
    """A serving loop: request 1 overflows, request 2 is fine.  The overflow of call #1 surfaces at call #2 WITHOUT a
    blocking wait - as a NumericsWarning that names call #1 - and call #2 still runs and returns its (finite, correct)
    result; check_numerics() then raises once for #1 only."""
    import warnings

    from mpnn_amd.engine import NumericsWarning

    kw = dict(k_neighbors=16)
    model = build_model(kw)
    rec = _batch()
    good = run_hip(model, rec, keep=False)                      # call #1: fine
    state = {n: p.detach().clone() for n, p in model.named_parameters()}
    bad_model = _overflowing_model()
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(dict(bad_model.named_parameters())[n].to(p.device))
    out = run_hip(model, rec, keep=False)                       # call #2: overflows
    assert not bool(torch.isfinite(out).all())
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(state[n])
    torch.cuda.synchronize()                                    # (so that the non-blocking look at call #2 sees it)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        again = run_hip(model, rec, keep=False)                 # call #3: fine, and must not be aborted
    msgs = [str(w.message) for w in caught if issubclass(w.category, NumericsWarning)]
    assert len(msgs) == 1 and "call #2" in msgs[0], msgs
    assert torch.equal(again, good)
    with pytest.raises(RuntimeError, match=r"call #2 "):
        model.check_numerics()
    model.check_numerics()


def test_sparse_model_has_check_numerics():
    from mpnn_amd import sparse

    assert callable(getattr(sparse.ProteinMPNN, "check_numerics", None))


def test_f16_mode_overflows_loudly_too():
    """The reduced-precision mode rounds each operand ONCE, to nearest even, with an instruction that overflows to inf
    (round 2 truncated with v_cvt_pkrtz, which saturates at 65504 and so returned finite, wrong numbers): the same
    overflowing model is reported by check_numerics() in f16 mode."""
    model = _overflowing_model()
    model.set_precision("f16")
    rec = _batch()
    out = run_hip(model, rec, keep=False)
    model.set_precision("fp32")
    assert not bool(torch.isfinite(out).all())
    with pytest.raises(RuntimeError, match="inf / NaN"):
        model.check_numerics()
