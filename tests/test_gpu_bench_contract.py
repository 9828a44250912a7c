"""bench.py's output contract on a tiny workload: exactly one JSON line on stdout with the fields the driver reads."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2",
           "--length", "96", "--train-batch", "2", "--samples", "2"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["value"] > 0 and abs(d["value"] - 2 * 96 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-2
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["value"] > 0
    for leg in ("train_step", "sampling", "shared_structure", "batch_assembly", "kernel_rooflines"):
        assert leg in d, leg


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without torch.distributed.run: bench.py starts the two ranks itself.  On a one-GPU box
    both ranks share the card and the collectives go over gloo (BENCH_BACKEND=gloo); the driver's 8-GPU run uses RCCL."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
           "--length", "96", "--train-batch", "2", "--samples", "4", "--no-f16-leg"]
    env = dict(os.environ, BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"]["world_size"] == 2 and d["ranks"]["backend"] == "gloo"
    assert d["rccl_ranks"] == 0 and d["config"]["global_batch"] == 4
    assert abs(d["value"] - 2 * 2 * 96 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-2   # whole-job residues/s
    assert d["train_step"]["n_gpus"] == 2 and d["sampling"]["n_gpus"] == 2
    # the N > 1 line explains itself: per-rank step / host-enqueue / all-reduce times
    t = d["train_step"]
    for key in ("per_rank_ms_per_step", "per_rank_host_enqueue_ms_per_step", "per_rank_allreduce_ms_per_step"):
        assert len(t[key]) == 2 and all(v >= 0 for v in t[key]), key
    assert t["allreduce_ms_per_step"] > 0 and t["allreduce_bytes"] == 4 * 1660485
    assert max(t["per_rank_ms_per_step"]) <= t["ms_per_step"] * 1.05


def test_one_rank_through_rccl():
    """The `nccl` (= RCCL) code path with the one GPU this box has: bench.py under torch.distributed.run with ONE rank,
    BENCH_FORCE_DIST=1 (process group with device_id, barriers, max-over-ranks, parameter broadcast, all-gather of the
    sampled sequences) and MPNN_FORCE_ALLREDUCE=1 (the flat gradient all-reduce and the re-pointed gradient views) -
    every collective call the 8-GPU run makes, on RCCL itself."""
    import socket

    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = str(s_.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch",
           "2", "--length", "96", "--train-batch", "2", "--samples", "4", "--no-f16-leg", "--no-cpu-baseline"]
    env = dict(os.environ, BENCH_FORCE_DIST="1", MPNN_FORCE_ALLREDUCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("BENCH_BACKEND", None)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["ranks"]["backend"].startswith("nccl") and d["rccl_ranks"] == 1
    t = d["train_step"]
    assert len(t["per_rank_allreduce_ms_per_step"]) == 1 and t["allreduce_ms_per_step"] > 0
    assert t["final_loss"] == t["final_loss"]   # finite: the optimizer stepped on the reduced views
