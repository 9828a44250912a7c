"""CPU-side checks: state_dict surface, C ABI exports, argument rejection (no GPU needed)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def test_state_dict_matches_reference_layout():
    import mpnn_amd
    from mpnn_amd.params import param_shapes

    for kw in (dict(), dict(use_virtual_center=True), dict(occupancy_cutoff=6.0, edge_cutoff=15.0)):
        m = mpnn_amd.ProteinMPNN(**kw)
        want = param_shapes(use_virtual_center=kw.get("use_virtual_center", False),
                            occupancy_cutoff=kw.get("occupancy_cutoff"))
        got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        assert got == dict(want)
    assert sum(p.numel() for p in mpnn_amd.ProteinMPNN().parameters()) == 1660485


def test_state_dict_matches_the_reference_generated_layout():
    """Against key names / shapes / dtypes the REFERENCE module itself reported (tests/golden/state_dict_layout.json,
    written by make_golden.py from model_utils.ProteinMPNN(**kwargs).state_dict()), in the reference's key order."""
    import json

    import mpnn_amd

    layout = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_layout.json")))
    assert len(layout) >= 6
    for name, rec in layout.items():
        m = mpnn_amd.ProteinMPNN(**rec["kwargs"])
        got = [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()]
        assert got == rec["keys"], name
        assert sum(p.numel() for p in m.parameters()) == rec["n_params"], name


def test_reference_format_checkpoint_loads(tmp_path):
    """The reference saves {"epoch", "step", "num_edges", "noise_level", "model_state_dict", "optimizer_state_dict",
    "scheduler_state_dict"} (train.py:548-568) and loads `ckpt["model_state_dict"]` (train.py:227), or a bare
    state_dict (finetune.py:496-500).  Build such a file with the reference's key layout and load it strictly."""
    import json

    import mpnn_amd

    layout = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_layout.json")))["train_py_defaults"]
    g = torch.Generator().manual_seed(0)
    sd = {k: torch.randn(shape, generator=g) for k, shape, _ in layout["keys"]}
    opt_state = {"state": {}, "param_groups": [{"lr": 1e-3, "params": list(range(len(sd)))}]}
    path = tmp_path / "last_epoch.pt"
    torch.save({"epoch": 3, "step": 1234, "num_edges": 48, "noise_level": 0.1, "model_state_dict": sd,
                "optimizer_state_dict": opt_state, "scheduler_state_dict": {}}, path)
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    model = mpnn_amd.ProteinMPNN(**layout["kwargs"])
    res = model.load_state_dict(ckpt["model_state_dict"], strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in model.state_dict().items():
        assert torch.equal(v, sd[k]), k
    bare = tmp_path / "proteinmpnn.pt"                      # finetune.py also accepts the bare dict
    torch.save(sd, bare)
    model2 = mpnn_amd.ProteinMPNN(**layout["kwargs"])
    model2.load_state_dict(torch.load(bare, map_location="cpu", weights_only=True))
    assert torch.equal(model2.W_out.weight, sd["W_out.weight"])
    # a checkpoint of a different variant must be refused, not silently accepted
    with pytest.raises(RuntimeError):
        mpnn_amd.ProteinMPNN(use_virtual_center=True).load_state_dict(sd)


def test_library_exports_every_declared_symbol():
    from mpnn_amd import _lib

    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "mpnn_hip.h")).read()
    hdr_nocomment = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = list(dict.fromkeys(re.findall(r"\b(mpnn_\w+)\s*\(", hdr_nocomment)))
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n
        m = re.search(r"\b" + n + r"\s*\(([^;]*?)\);", hdr_nocomment, re.S)
        args = [a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"]
        if n in _lib.SIGNATURES:
            assert len(args) == len(_lib.SIGNATURES[n]), n
    assert lib.mpnn_version() >= 1


def test_workspace_query():
    from mpnn_amd import _lib

    lib = _lib.load()
    part = 16512 * 4
    assert lib.mpnn_workspace_bytes(b"mpnn_edge_wgrad", 16384, 1) == 768 * part
    assert lib.mpnn_workspace_bytes(b"mpnn_edge_wgrad", 100, 1) == 100 * part
    assert lib.mpnn_workspace_bytes(b"mpnn_node_wgrad", 10240, 4) == 256 * 4 * part
    assert lib.mpnn_workspace_bytes(b"mpnn_node_wgrad_batch", 64, 11) == 2 * 11 * part
    assert lib.mpnn_workspace_bytes(b"mpnn_enc_message", 16384, 0) == 0
    assert lib.mpnn_workspace_bytes(b"no_such_entry", 1, 1) == -1
    assert lib.mpnn_workspace_bytes(None, 1, 1) == -1


def test_argument_rejection_without_gpu():
    """Launchers validate before touching the device: bad arguments return -1 with a message."""
    from mpnn_amd import _lib

    lib = _lib.load()
    rc = lib.mpnn_knn_topk(None, 1, 8, 4, 0.0, None, None, None, None, None)
    assert rc == -1 and b"mpnn_knn_topk" in lib.mpnn_last_error()
    rc = lib.mpnn_knn_topk(ctypes.c_void_p(16), 1, 8, 100, 0.0, None, ctypes.c_void_p(16), None, None, None)
    assert rc == -1 and b"K" in lib.mpnn_last_error()


def test_cpu_input_fails_loudly():
    import mpnn_amd
    from mpnn_amd.synthetic import synthetic_batch

    m = mpnn_amd.ProteinMPNN(k_neighbors=8).eval()
    b = synthetic_batch(1, 16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        with torch.no_grad():
            m(b["X"], b["S"], b["mask"], b["chain_M"], b["residue_idx"], b["chain_encoding_all"])


def test_optimizer_schedule_and_nll_match_the_reference_recipe():
    """train.py:217-222: AdamW(wd 1e-2), LinearLR(1e-8 -> 1 over 1 % of the steps) then cosine to 0; train.py:51-60."""
    import math

    import torch

    from mpnn_amd import training

    lin = torch.nn.Linear(4, 3)
    opt, sched = training.make_optimizer(lin, lr=2e-3, total_steps=1000)
    assert isinstance(opt, torch.optim.AdamW) and opt.defaults["weight_decay"] == 1e-2   # CPU module: torch's AdamW
    lrs = []
    for _ in range(1000):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
    assert abs(lrs[0] - 2e-3 * 1e-8) < 1e-12 and abs(lrs[10] - 2e-3) < 1e-9      # warm-up = 10 steps
    assert all(lrs[i] < lrs[i + 1] for i in range(10)) and all(lrs[i] >= lrs[i + 1] for i in range(10, 999))
    assert abs(lrs[10 + 495] - 2e-3 * 0.5 * (1 + math.cos(math.pi * 495 / 990))) < 1e-9
    assert training.make_optimizer(lin)[1] is None
    g = torch.Generator().manual_seed(0)
    lp = torch.log_softmax(torch.randn(2, 5, 21, generator=g), -1)
    S = torch.randint(0, 21, (2, 5), generator=g)
    mask = (torch.rand(2, 5, generator=g) < 0.7).float()
    loss, loss_av, hits = training.loss_nll(S, lp, mask)
    ref = torch.nn.NLLLoss(reduction="none")(lp.view(-1, 21), S.view(-1)).view(2, 5)
    assert torch.equal(loss, ref) and torch.equal(hits, lp.argmax(-1) == S)
    assert torch.allclose(loss_av, (ref * mask).sum() / mask.sum())
