"""mpnn_amd.optim: the multi-tensor AdamW kernel against torch.optim.AdamW (the reference's optimizer,
train.py:217) and the clip against torch.nn.utils.clip_grad_norm_ (train.py:389-390), through the C ABI."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(128, 384), (128,), (21, 128), (512, 128), (128, 512), (16, 66), (1,), (3, 5, 7)] * 8   # 64 tensors: two launches


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(*s, generator=g) * 0.3 for s in SHAPES]


@pytest.mark.parametrize("max_norm", [0.0, 0.5])
def test_adamw_matches_torch(max_norm):
    from mpnn_amd.optim import AdamW

    ours = [torch.nn.Parameter(p.clone().cuda()) for p in _params(0)]
    ref = [torch.nn.Parameter(p.clone().double()) for p in _params(0)]
    opt = AdamW(ours, lr=3e-3, weight_decay=1e-2, max_grad_norm=max_norm)
    opt_ref = torch.optim.AdamW(ref, lr=3e-3, weight_decay=1e-2)
    for step in range(1, 8):
        grads = _params(100 + step)
        for p, q, g in zip(ours, ref, grads):
            p.grad = (g * 0.05).cuda()
            q.grad = (g * 0.05).double()
        if step == 4:                      # a tensor without gradient is skipped, as torch does
            ours[3].grad = None
            ref[3].grad = None
        if max_norm:
            torch.nn.utils.clip_grad_norm_(ref, max_norm)
        v0 = ours[0]._version
        opt.step()
        opt_ref.step()
        assert ours[0]._version > v0, "the update must be visible to version-keyed caches"
        if step == 2:                      # a scheduler changes the rate between steps
            for o in (opt, opt_ref):
                o.param_groups[0]["lr"] = 1e-3
    for p, q in zip(ours, ref):
        err = (p.detach().cpu().double() - q.detach()).abs().max()
        assert float(err) <= 2e-6 * float(q.detach().abs().max()) + 1e-7, float(err)
    sd = opt.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 7.0
    assert float(sd["state"][3]["step"]) == 6.0
    # the reference's checkpoint format round-trips through torch's own optimizer
    opt_ref.load_state_dict({"state": {k: {n: (t.cpu().double() if torch.is_tensor(t) and t.dim() else t)
                                          for n, t in v.items()} for k, v in sd["state"].items()},
                             "param_groups": opt_ref.state_dict()["param_groups"]})


def test_clip_grad_norm_matches_torch():
    from mpnn_amd.optim import clip_grad_norm_

    for scale, max_norm in ((1.0, 0.7), (1e-3, 5.0)):
        ours = [torch.nn.Parameter(p.clone().cuda()) for p in _params(1)]
        ref = [torch.nn.Parameter(p.clone()) for p in _params(1)]
        for p, q, g in zip(ours, ref, _params(2)):
            p.grad, q.grad = (g * scale).cuda(), g * scale
        total = clip_grad_norm_(ours, max_norm)
        total_ref = torch.nn.utils.clip_grad_norm_(ref, max_norm)
        assert abs(float(total) - float(total_ref)) <= 1e-5 * float(total_ref)
        for p, q in zip(ours, ref):
            assert torch.allclose(p.grad.cpu(), q.grad, rtol=2e-6, atol=1e-9)


def test_clip_with_non_finite_gradients_matches_torch():
    """One NaN gradient element makes the norm NaN: torch's clip then turns EVERY gradient into NaN (clamp(nan, max=1)
    is nan) - an obvious failure - and so must the kernels, in the stand-alone clip and inside the AdamW pass.  One
    infinite element gives coefficient 0 in both: finite gradients are zeroed, the infinite one becomes NaN."""
    from mpnn_amd.optim import AdamW, clip_grad_norm_

    for poison in (float("nan"), float("inf")):
        ours = [torch.nn.Parameter(p.clone().cuda()) for p in _params(1)]
        ref = [torch.nn.Parameter(p.clone()) for p in _params(1)]
        for p, q, g in zip(ours, ref, _params(2)):
            p.grad, q.grad = g.clone().cuda(), g.clone()
        ours[5].grad.view(-1)[3] = poison
        ref[5].grad.view(-1)[3] = poison
        clip_grad_norm_(ours, 0.7)
        torch.nn.utils.clip_grad_norm_(ref, 0.7)
        for p, q in zip(ours, ref):
            assert torch.equal(torch.isnan(p.grad.cpu()), torch.isnan(q.grad))
            assert torch.allclose(p.grad.cpu(), q.grad, rtol=2e-6, atol=1e-9, equal_nan=True)
        # folded into the update: the parameters end up NaN exactly where torch's do
        ours = [torch.nn.Parameter(p.clone().cuda()) for p in _params(1)]
        ref = [torch.nn.Parameter(p.clone()) for p in _params(1)]
        for p, q, g in zip(ours, ref, _params(2)):
            p.grad, q.grad = g.clone().cuda(), g.clone()
        ours[5].grad.view(-1)[3] = poison
        ref[5].grad.view(-1)[3] = poison
        opt, opt_ref = AdamW(ours, lr=1e-3, max_grad_norm=0.7), torch.optim.AdamW(ref, lr=1e-3)
        torch.nn.utils.clip_grad_norm_(ref, 0.7)
        opt.step()
        opt_ref.step()
        for p, q in zip(ours, ref):
            assert torch.equal(torch.isnan(p.detach().cpu()), torch.isnan(q.detach())), poison


def test_train_step_max_grad_norm_is_per_call():
    """train_step(max_grad_norm=x) sets the in-tree optimizer's clip for that step; a later 0 switches it off again
    (it used to stick), None leaves the optimizer's own setting."""
    import mpnn_amd
    from mpnn_amd import training
    from mpnn_amd.synthetic import synthetic_batch

    model = mpnn_amd.ProteinMPNN(k_neighbors=16, augment_eps=0.0, dropout=0.0).train().cuda()
    opt, _ = training.make_optimizer(model, lr=1e-3, max_grad_norm=0.5)
    b = synthetic_batch(B=1, L=32, seed=3, device="cuda")
    batch = (b["X"], b["S"], b["mask"], b["chain_M"], b["residue_idx"], b["chain_encoding_all"])
    training.train_step(model, opt, batch, fix_order=b["fix_order"])
    assert opt.max_grad_norm == 0.5
    training.train_step(model, opt, batch, fix_order=b["fix_order"], max_grad_norm=1.0)
    assert opt.max_grad_norm == 1.0
    training.train_step(model, opt, batch, fix_order=b["fix_order"], max_grad_norm=0.0)
    assert opt.max_grad_norm == 0.0


def test_train_step_uses_the_in_tree_optimizer():
    import mpnn_amd
    from mpnn_amd import optim, training
    from mpnn_amd.synthetic import synthetic_batch

    torch.manual_seed(0)
    model = mpnn_amd.ProteinMPNN(k_neighbors=24, augment_eps=0.0, dropout=0.0).train().cuda()
    opt, sched = training.make_optimizer(model, lr=1e-3, total_steps=200, max_grad_norm=1.0)
    assert isinstance(opt, optim.AdamW)
    b = synthetic_batch(B=2, L=48, seed=3, device="cuda")
    batch = (b["X"], b["S"], b["mask"], b["chain_M"], b["residue_idx"], b["chain_encoding_all"])
    losses = [float(training.train_step(model, opt, batch, scheduler=sched, fix_order=b["fix_order"],
                                        max_grad_norm=1.0)[0]) for _ in range(6)]
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses
    assert opt.last_grad_sumsq is not None and float(opt.last_grad_sumsq) > 0
    with torch.no_grad():                      # the inference path sees the updated weights
        lp = model.eval()(*batch, fix_order=b["fix_order"])
    assert torch.isfinite(lp).all()
