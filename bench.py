#!/usr/bin/env python3
"""Throughput bench of the ProteinMPNN hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one encoder+decoder forward (featurizer included, teacher-forced decoder, fixed random
decoding order) over one batch of synthetic backbones of BASELINE.json configs[1]:
B=32 x L=512, k=48, hidden 128, fp32, inputs resident in HBM.  Proteins are independent, so ranks
take their own batch (weak scaling, no data-path collective).  Rank 0 prints ONE JSON line with the
whole-job residues/s, the roofline of the dominant kernel (HIP-event timed on its launch stream) and
the CPU baseline (the oracle's reference formulation on a bounded sample, rank 0, N=1 only).
"""
import argparse
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic FLOPs per residue as the reference writes the maths (SURVEY.md 8d, K=48, H=128)
FLOP_FWD_PER_RES = 83.86e6
FLOP_AS_WRITTEN = {  # per residue, per launch of the named entry point
    "mpnn_enc_edge_update": 2 * 48 * (384 * 128 + 2 * 128 * 128),   # W11, W12, W13 (reference :93)
    "mpnn_enc_message": 2 * 48 * (384 * 128 + 2 * 128 * 128),       # W1, W2, W3     (reference :78)
    "mpnn_dec_message": 2 * 48 * (512 * 128 + 2 * 128 * 128),       # W1, W2, W3     (reference :124)
    "mpnn_edge_features": 2 * 48 * (416 * 128 + 128 * 128 + 66 * 16),  # edge_embedding, W_e, positional
}
# what the kernel itself issues to the matrix pipe, per residue: the per-edge 128x128 products only (W1a/W1c and W3
# live in the node kernel), each evaluated as THREE f16 MFMA products (hi*hi + hi*lo + lo*hi, fp32 accumulate)
FLOP_F16_ISSUED = {
    "mpnn_enc_edge_update": 3 * 2 * 48 * 3 * 128 * 128,
    "mpnn_enc_message": 3 * 2 * 48 * 2 * 128 * 128,
    "mpnn_dec_message": 3 * 2 * 48 * 2 * 128 * 128,
    "mpnn_edge_features": 3 * 2 * 48 * (416 * 128 + 128 * 128),
}


def algorithmic_bytes(name, K, Kp):
    """Minimum HBM bytes per residue and launch: every tensor the launch must read or write
    counted once; gathered neighbour rows counted at full size although most of them hit in L2."""
    F = 512 * Kp  # one residue's hE block (fragment layout, padded to 16-neighbour tiles)
    return {
        "mpnn_backbone_atoms": 48 + 112,
        "mpnn_knn_topk": 16 + 12 * K,                         # Ca + mask in; E_idx (i64 API + i32 internal) out
        "mpnn_edge_features": 112 + K * (112 + 4) + F,        # own + gathered atom rows, indices; hE out
        "mpnn_enc_message": F + 512 * K + 4 * K + 512 + 516,  # hE, gathered Q rows, indices, P row; gsum + cnt out
        "mpnn_dec_message": F + 512 * K + 8 * K + 512 + 512,  # hE, one gathered row per edge, indices + ranks; gsum out
        "mpnn_enc_edge_update": 2 * F + 512 * K + 4 * K + 512,  # hE in and out, gathered Q' rows, P' row
        "mpnn_node_update": 6 * 512,                          # h_V, gsum in; h_V and up to three projections out
    }.get(name)


PEAK_HBM_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak
PEAK_F32_MFMA_TFLOPS = 157.3   # for reference: the f32-input MFMA / f32 vector peak


def host_cpu_info():
    """(model string, physical cores of the machine, hardware threads of the machine, threads this process may use)."""
    model, cores, phys = "unknown", set(), None
    try:
        pid = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and model == "unknown":
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("physical id"):
                pid = ln.split(":", 1)[1].strip()
            elif ln.startswith("core id"):
                cores.add((pid, ln.split(":", 1)[1].strip()))
        phys = len(cores) or None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    try:
        allowed = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = logical
    # the GPU box hands a one-GPU job a share of the host (OMP_NUM_THREADS is set to it): never oversubscribe that share
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    usable = min(allowed, share) if share > 0 else allowed
    return model, phys or logical, logical, max(1, usable)


def cpu_baseline(B_cpu, L, K, iters, warm=3):
    """The oracle in the reference's formulation (same [B,L,L] temporaries and O(L^3) mask einsum), timed on this
    host's cores (>= 3 warm-up + >= 5 timed, median): forward on every core this job may use (the headline
    `value`), forward on one thread, one training step (forward + autograd backward), and the C1 workload (100
    synthetic chains, log-normal lengths, token-budget batches).  Bounded samples, ~25 s in total."""
    from mpnn_amd import ProteinMPNN
    from mpnn_amd.synthetic import synthetic_batch
    from oracle import mpnn_oracle as orc

    cpu_model, phys, logical, threads = host_cpu_info()
    model = ProteinMPNN(k_neighbors=K, augment_eps=0.0, dropout=0.0).eval()
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    cfg = orc.Config(k_neighbors=K)

    def fwd_time(nthreads, Bc, n, nwarm):
        torch.set_num_threads(nthreads)
        b = synthetic_batch(Bc, L, seed=123)
        args = (b["X"], b["S"], b["mask"], b["chain_M"], b["residue_idx"], b["chain_encoding_all"])
        times = []
        with torch.no_grad():
            for _ in range(nwarm):
                orc.forward(sd, cfg, *args, fix_order=b["fix_order"], formulation="reference")
            for _ in range(n):
                t0 = time.perf_counter()
                orc.forward(sd, cfg, *args, fix_order=b["fix_order"], formulation="reference")
                times.append(time.perf_counter() - t0)
        times.sort()
        return times[len(times) // 2]

    def train_time(nthreads, Bc, n):
        torch.set_num_threads(nthreads)
        b = synthetic_batch(Bc, L, seed=124)
        args = (b["X"], b["S"], b["mask"], b["chain_M"], b["residue_idx"], b["chain_encoding_all"])
        sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        times = []
        for i in range(n + 1):
            t0 = time.perf_counter()
            lp = orc.forward(sdg, cfg, *args, fix_order=b["fix_order"], formulation="reference")
            _, loss = orc.loss_smoothed(b["S"], lp, b["mask"] * b["chain_M"])
            loss.backward()
            if i:
                times.append(time.perf_counter() - t0)
            for v in sdg.values():
                v.grad = None
        times.sort()
        return times[len(times) // 2]

    med = fwd_time(threads, B_cpu, iters, warm)
    one = fwd_time(1, 1, 3, 1)
    trn = train_time(threads, 1, 3)
    torch.set_num_threads(threads)
    c1 = c1_workload()
    t0 = time.perf_counter()
    with torch.no_grad():
        for b in c1["batches"]:
            orc.forward(sd, cfg, b["X"], b["S"], b["mask"], b["chain_M"], b["residue_idx"], b["chain_encoding_all"],
                        fix_order=b["fix_order"], formulation="reference")
    c1_s = time.perf_counter() - t0
    return {"value": B_cpu * L / med, "unit": "residues/s", "cores": threads, "kind": "port",
            "cpu_model": cpu_model, "host_physical_cores": phys, "host_hardware_threads": logical,
            "sample": f"oracle (reference formulation, fp32, torch CPU, {threads} threads = every core this job may use "
                      f"of the host's {phys} physical cores) forward on B={B_cpu} x L={L}, k={K}: median of {iters} "
                      f"runs after {warm} warm-ups, {med * 1e3:.0f} ms each",
            "forward_single_thread": {"value": round(L / one, 1), "cores": 1,
                                      "sample": f"same, B=1 x L={L}, median of 3 after 1 warm-up, {one * 1e3:.0f} ms each"},
            "train_step": {"value": round(L / trn, 1), "cores": threads,
                           "sample": f"forward + loss_smoothed + autograd backward, B=1 x L={L}, median of 3 after 1 "
                                     f"warm-up, {trn * 1e3:.0f} ms each"},
            "c1_forward": {"value": round(c1["residues"] / c1_s, 1), "unit": "valid residues/s", "cores": threads,
                           "sample": f"BASELINE configs[0] stand-in: {c1['describe']}; one pass, {c1_s:.1f} s"}}


def c1_workload(device="cpu", n_chains=100, budget=10000, seed=77):
    """BASELINE configs[0]: ~100 chains with log-normal lengths (median 250, clipped to 40..1000) standing
    in for the unavailable pdb_2021aug02_sample, grouped into padded batches under a 10000-token budget exactly as the
    reference's StructureLoader does (protein_mpnn_dataset.py:60-87; the entry that closes a batch is kept here)."""
    import numpy as np

    from mpnn_amd.data import TokenBudgetBatcher
    from mpnn_amd.synthetic import hash_normal, synthetic_batch

    lengths = np.clip(np.exp(np.log(250.0) + 0.6 * hash_normal(n_chains, seed)), 40, 1000).astype(np.int64)
    groups = [g for g in TokenBudgetBatcher(lengths, batch_size=budget, shuffle=False, drop_overflow=False).clusters if g]
    batches = []
    for gi, ids in enumerate(groups):
        ls = [int(lengths[i]) for i in ids]
        batches.append(synthetic_batch(len(ids), max(ls), seed=seed * 100 + gi, lengths=ls, device=device))
    return {"batches": batches, "residues": int(lengths.sum()), "padded": int(sum(b["mask"].numel() for b in batches)),
            "describe": f"{n_chains} synthetic chains, log-normal lengths (median 250, 40..1000; {int(lengths.sum())} "
                        f"residues), {len(batches)} token-budget batches <= {budget} padded residues, k=48"}


def library_sha256():
    """Hash of the kernel library in use: PMC traffic numbers are only quoted for the build they were collected on."""
    import hashlib

    from mpnn_amd import _lib
    try:
        return hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()
    except OSError:
        return None


def max_over_ranks(x, dev, dist):
    on_dev = dist.get_backend() == "nccl"
    t = torch.tensor([x], device=dev if on_dev else "cpu", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def log(msg):
    if os.environ.get("BENCH_VERBOSE"):
        print(f"[bench rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def quiesce_gc():
    """Before a timed region: one full cyclic-GC pass of CPython scans every object torch and the earlier legs left
    behind (~40 ms here, during which nothing is enqueued and the GPU drains); whether one falls into a 70 ms timed
    region depends on the allocation count at that moment.  Collect now and move the survivors out of the collector's
    sight, so the timed steps only ever trigger young-generation passes."""
    gc.collect()
    gc.freeze()


def train_leg(a, dev, rank, world, dist, barrier, _lib):
    """BASELINE configs[2]: full training step (forward, label-smoothed loss, backward, gradient all-reduce over
    RCCL when world > 1, AdamW) on ~10k-residue batches per rank; reported beside the forward metric."""
    from mpnn_amd import ProteinMPNN, distributed, training
    from mpnn_amd.synthetic import synthetic_batch

    B, L, K = a.train_batch, a.length, a.k
    # objects of the earlier legs (models, thousands of timing events) would be re-scanned by every cyclic-GC pass the
    # autograd graph triggers (measured: 24 instead of 17 ms per step): collect once, then exclude them from scanning
    quiesce_gc()
    torch.manual_seed(1234)
    # the reference's training defaults (train.py:199-211): dropout 0.1 (node- and edge-level), backbone noise 0.1 A
    model = ProteinMPNN(k_neighbors=K, augment_eps=0.1, dropout=0.1).train().to(dev)
    if dist is not None:
        log("train: broadcasting parameters")
        distributed.broadcast_parameters(model.parameters())
        log("train: broadcast done")
    # AdamW(lr, weight_decay=1e-2) as train.py:217, on the in-tree multi-tensor kernel
    from mpnn_amd.optim import AdamW
    opt = AdamW(model.parameters(), lr=1e-3, weight_decay=1e-2)   # mpnn_adamw_step: 3 launches for the 118 tensors
    b = synthetic_batch(B, L, seed=2000 + rank, device=dev)
    batch = (b["X"], b["S"], b["mask"], b["chain_M"], b["residue_idx"], b["chain_encoding_all"])
    steps, warm = max(3, min(a.steps, 20)), 2   # (the GPU starts idle: a ~1 ms pipeline fill is spread over these)

    ar_events = []

    def step():
        return training.train_step(model, opt, batch, world_size=world, allreduce_timings=ar_events)

    for _ in range(warm):
        step()
        log("train: warm-up step done")
    quiesce_gc()
    barrier()
    del ar_events[:]
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = step()
    enqueue = time.perf_counter() - t0
    torch.cuda.synchronize()
    own = time.perf_counter() - t0          # this rank's own steps, before it waits for the others
    barrier()
    elapsed = time.perf_counter() - t0
    ranks = None
    if dist is not None:
        elapsed = max_over_ranks(elapsed, dev, dist)
        # what a lost percent of scaling would be made of, per rank: its own step time (the barrier aside), its host
        # enqueue time, the gradient all-reduce (HIP events around the collective, div included)
        ar_ms = sum(e0.elapsed_time(e1) for e0, e1 in ar_events) / max(len(ar_events), 1)
        mine = torch.tensor([own / steps * 1e3, enqueue / steps * 1e3, ar_ms], dtype=torch.float64,
                            device=dev if dist.get_backend() == "nccl" else "cpu")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        ranks = {"per_rank_ms_per_step": [round(float(t[0]), 3) for t in every],
                 "per_rank_host_enqueue_ms_per_step": [round(float(t[1]), 3) for t in every],
                 "per_rank_allreduce_ms_per_step": [round(float(t[2]), 3) for t in every],
                 "allreduce_ms_per_step": round(max(float(t[2]) for t in every), 3),
                 "allreduce_bytes": 4 * sum(p.numel() for p in model.parameters())}
    assert torch.isfinite(loss)
    prof = {}
    _lib.set_profiler(prof)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    _lib.set_profiler(None)
    kern = {n: round(sum(e0.elapsed_time(e1) for e0, e1 in evs) / 3, 3) for n, evs in sorted(prof.items())}
    hip_ms = sum(kern.values())
    # HBM bytes this design moves per residue and step (per kernel confirmed by the PMC passes in
    # profiles/r03_pmc_traffic_train.txt): in units of one residue's fragment tensor F = 8 * ceil(K/16) KiB.
    #   K <= 48 (fused backward kernels): encoder layer 25 F (message forward 3, edge-update forward 6, fused edge-update
    #   backward 6 + W11b contraction 2 + gather 1, fused message backward 6 + gather 1), decoder layer 9.67 F,
    #   featurizer 12 F (forward 3, backward 3, W_e contraction 2, embedding contraction 4 reads of dE0 - the raw features
    #   are recomputed, not read)  =>  116 F for the 3 + 3 layer model;  K > 48 (unfused): 36 / 14 / 12  =>  162 F
    frag = 8 * ((K + 15) // 16) * 1024
    step_bytes = (116 if K <= 48 else 162) * frag * B * L
    tbps = step_bytes / (elapsed / steps) / 1e12
    return {**(ranks or {}),
            "metric": "residues/sec (train step: fwd + bwd + AdamW) at L=512,k=48",
            "value": round(B * L * world * steps / elapsed, 1), "unit": "residues/s", "n_gpus": world, "steps": steps,
            "warmup": warm, "ms_per_step": round(elapsed / steps * 1e3, 3),
            "host_enqueue_ms_per_step": round(enqueue / steps * 1e3, 3), "scaling": "weak",
            "config": {"workload": f"BASELINE configs[2]: {B} x L={L} = {B * L} residues per rank, k={K}, dropout 0.1, "
                                   "backbone noise 0.1, random decoding order per step, AdamW, one flat gradient "
                                   "all-reduce per step", "global_batch": B * world},
            "roofline": {"bound": "hbm", "achieved": round(tbps * 1e3, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(tbps / 8.0, 4), "bytes_per_step": step_bytes,
                         "note": "whole step, design bytes (116 fragment tensors per residue at K <= 48: the fused "
                                 "backward kernels keep da2 / dz in the workgroup and read each saved tensor once); the "
                                 "fused backward kernels themselves are LDS- / issue-bound in lock-step (2.3-3.5 TB/s), "
                                 "the forward-with-saves kernels and the weight-gradient passes run at 4-5 TB/s"},
            "final_loss": round(float(loss), 6),
            "hip_kernels_ms_per_step": kern, "hip_kernels_total_ms": round(hip_ms, 3),
            "note": "forward, backward and weight gradients of the featurizer, edge and node stages, the output head, "
                    "the smoothed loss, the sequence-term table and the AdamW update are HIP kernels (no library GEMM on "
                    "the path); what is left to ATen is glue: one_hot(S), cat / zero "
                    "fills / copies of autograd, the decoding order's argsort; every sum is order-fixed (no "
                    "floating-point atomics): two runs give identical gradients"}


def assembly_leg(a, dev, with_cpu):
    """padded batches of ~10k residues out of a shard resident in HBM (mpnn_assemble_batch) beside the
    reference's way (host featurize + host-to-device copies; the oracle's numpy restatement, one thread)."""
    from mpnn_amd.data import StructureShard, TokenBudgetBatcher
    from mpnn_amd.synthetic import synthetic_entries

    entries = synthetic_entries(400, seed=77)
    shard = StructureShard.from_entries(entries, shuffle=False).to(dev)
    batches = [b for b in TokenBudgetBatcher(shard.lengths, batch_size=10000, shuffle=False, drop_overflow=False).clusters
               if b]
    for ids in batches[:2]:
        shard.assemble(ids)
    quiesce_gc()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        for ids in batches:
            out = shard.assemble(ids)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (reps * len(batches))
    slots = sum(len(b) * int(shard.lengths[b].max()) for b in batches) / len(batches)
    res = {"workload": f"{len(batches)} batches under a 10000-residue padded budget from 400 synthetic multi-chain "
                       f"structures (mean {slots:.0f} residue slots per batch)",
           "gpu_us_per_batch": round(dt * 1e6, 1), "residue_slots_per_s": round(slots / dt, 1),
           "algorithmic_GBps": round(slots * 141 / dt / 1e9, 2),
           "note": "one launch per batch, ~61 B read + ~80 B written per slot: bound by launch latency, not by HBM"}
    if with_cpu:
        from oracle import featurize_oracle as fo
        orders = [fo.chain_order(e, shuffle=False) for e in entries]
        t0 = time.perf_counter()
        n = 0
        for ids in batches[:6]:
            ref = fo.featurize([entries[i] for i in ids], orders=[orders[i] for i in ids])
            moved = [torch.from_numpy(ref[k]).to(dev) for k in (0, 1, 2, 4, 5, 7)]
            n += 1
        torch.cuda.synchronize()
        res["cpu_port_ms_per_batch"] = round((time.perf_counter() - t0) / n * 1e3, 2)
        res["cpu_port"] = "oracle/featurize_oracle.py (numpy restatement of data_utils.py:51-183, 1 thread) + H2D copies"
    return res


def sampling_leg(a, dev, rank, world, dist, barrier):
    """BASELINE configs[4]: autoregressive sampling, random decoding order, virtual-centre node; 64 samples of one
    L=512 structure, sharded over the ranks (each rank encodes the structure once and draws its share)."""
    from mpnn_amd import ProteinMPNN
    from mpnn_amd.synthetic import synthetic_batch

    L, K = a.length, a.k
    total = a.samples
    mine = total // world + (1 if rank < total % world else 0)
    torch.manual_seed(1234)
    model = ProteinMPNN(k_neighbors=K, augment_eps=0.0, dropout=0.0, use_virtual_center=True).eval().to(dev)
    b = synthetic_batch(1, L, seed=3000, device=dev)  # the same structure on every rank
    args = (b["X"], b["mask"], b["chain_M"], b["residue_idx"], b["chain_encoding_all"])
    iters = 3

    def step(i):
        return model.sample(*args, n_samples=max(mine, 1), temperature=0.1, seed=100 * i + rank)

    out = step(0)
    quiesce_gc()
    barrier()
    t0 = time.perf_counter()
    for i in range(iters):
        out = step(i + 1)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        elapsed = max_over_ranks(elapsed, dev, dist)
        # the job's product: every rank's sequences gathered on all ranks (the path's only collective)
        if mine * world == total:
            mine_S = out["S"].contiguous() if dist.get_backend() == "nccl" else out["S"].cpu()
            gathered = [torch.empty_like(mine_S) for _ in range(world)]
            dist.all_gather(gathered, mine_S)
            assert torch.equal(gathered[rank], mine_S)
    assert int(out["S"].min()) >= 0 and torch.isfinite(out["log_probs"]).all()
    res = {"metric": "sampled residues/sec (autoregressive, random order, virtual centre) at L=512,k=48",
           "value": round(total * L * iters / elapsed, 1), "unit": "residues/s", "n_gpus": world, "steps": iters,
           "ms_per_step": round(elapsed / iters * 1e3, 3), "scaling": "strong",
           "config": {"workload": f"BASELINE configs[4]: 1 structure x L={L}, k={K}, {total} samples "
                                  f"({mine} on rank 0), temperature 0.1", "dependency_levels": out["n_levels"]}}
    if rank == 0:
        from mpnn_amd import _lib
        n_dec = len(model.decoder_layers)

        def timed(n, reps=3):
            """(wall ms per call, device ms of the decoder's level loop per call, levels) for n samples."""
            model.sample(*args, n_samples=n, temperature=0.1, seed=7)
            torch.cuda.synchronize()
            prof = {}
            t0_ = time.perf_counter()
            for i in range(reps):
                o = model.sample(*args, n_samples=n, temperature=0.1, seed=8 + i)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0_) / reps * 1e3
            _lib.set_profiler(prof)
            o = model.sample(*args, n_samples=n, temperature=0.1, seed=8)
            torch.cuda.synchronize()
            _lib.set_profiler(None)
            run = [e0.elapsed_time(e1) for name, evs in prof.items() if name.startswith("mpnn_decode_run")
                   for e0, e1 in evs]
            return wall, (sum(run) if run else None), o["n_levels"]

        # the decoder is a chain of dependent launches (levels x decoder layers): its bound is their latency, not HBM or
        # the matrix pipe.  achieved = device time of the level loop; the model = levels x layers x the time of ONE
        # dependent launch at that level size (measured here: loop time / launches)
        wall64, loop64, lv64 = timed(max(mine, 1))
        wall8, loop8, lv8 = timed(8)
        wall1, loop1, lv1 = timed(1)
        if loop64 is not None:
            res["roofline"] = {
                "bound": "latency", "unit": "us per dependent launch",
                "levels": lv64, "dependent_launches": lv64 * n_dec,
                "achieved": round(loop64 * 1e3 / (lv64 * n_dec), 2),
                "floor": round(loop1 * 1e3 / (lv1 * n_dec), 2),
                "frac": round((loop1 / (lv1 * n_dec)) / (loop64 / (lv64 * n_dec)), 4),
                "level_loop_ms": round(loop64, 3), "call_ms": round(wall64, 3),
                "note": "floor = the same launch at ONE sample (a level of ~7 residues: one workgroup per residue, nothing "
                        "to share): what a level costs when only its chain of dependent loads, weight streams (0.8 MB "
                        "per workgroup from L2) and barriers is left; host enqueue is not on the critical path "
                        "(mpnn_decode_run issues the whole loop from C)"}
        res["samples_per_rank_8"] = {"ms_per_call": round(wall8, 3), "level_loop_ms": round(loop8, 3) if loop8 else None,
                                     "levels": lv8, "sampled_residues_per_s": round(8 * L / wall8 * 1e3, 1),
                                     "note": "what each GPU of BASELINE configs[4] (64 samples over 8 GPUs) sees"}
        res["one_sequence"] = {"ms_per_call": round(wall1, 3), "level_loop_ms": round(loop1, 3) if loop1 else None,
                               "levels": lv1}
    return res


def self_launch(n, json_fd):
    """`python bench.py --gpus N` without a launcher: run the same command line under torch.distributed.run with N
    local ranks (RCCL by default, BENCH_BACKEND=gloo to rehearse on fewer GPUs).  Nothing in this parent process has
    initialised the GPU.  Returns the exit code (non-zero if any rank failed)."""
    import socket
    import subprocess

    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    lines = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        print(f"bench.py: the {n}-rank run failed (exit code {p.returncode}, {len(lines)} JSON lines)", file=sys.stderr)
        return p.returncode or 1
    os.write(json_fd, (lines[-1] + "\n").encode())
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--length", type=int, default=512)
    ap.add_argument("--k", type=int, default=48)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f16-leg", action="store_true", help="skip the reduced-precision forward leg")
    ap.add_argument("--train-batch", type=int, default=20,
                    help="proteins per rank in the training leg (BASELINE configs[2]: ~10k-residue batches); 0 = skip")
    ap.add_argument("--samples", type=int, default=64,
                    help="sequences drawn per structure in the sampling leg (BASELINE configs[4]); 0 = skip")
    a = ap.parse_args()

    # stdout carries exactly one line, the JSON: anything a library prints there (RCCL's start-up banner does) goes to
    # stderr instead, by pointing fd 1 at stderr until the line is written
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # started as plain `python bench.py --gpus N`: launch the N ranks ourselves (one process per GPU under
        # torch.distributed.run) BEFORE this process touches the GPU, relay rank 0's JSON line, fail if any rank fails
        sys.exit(self_launch(a.gpus, json_fd))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: start {a.gpus} ranks (or run `python bench.py --gpus "
              f"{a.gpus}` without a launcher, which starts them itself)", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no ROCm device visible; the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    # BENCH_BACKEND=gloo lets several ranks share one GPU: a rehearsal of the multi-rank code path on a 1-GPU box
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    # BENCH_FORCE_DIST=1 (under torch.distributed.run with one rank) sends the one-rank run through the RCCL code path too
    if world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    from mpnn_amd import ProteinMPNN, _lib
    from mpnn_amd.synthetic import synthetic_batch

    B, L, K = a.batch, a.length, a.k
    torch.manual_seed(1234)  # same random-init weights on every rank
    model = ProteinMPNN(k_neighbors=K, augment_eps=0.0, dropout=0.0).eval().to(dev)
    b = synthetic_batch(B, L, seed=1000 + rank, device=dev)
    args = (b["X"], b["S"], b["mask"], b["chain_M"], b["residue_idx"], b["chain_encoding_all"])

    def step():
        with torch.no_grad():
            return model(*args, fix_order=b["fix_order"])

    log("forward warm-up")
    for _ in range(a.warmup):
        step()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    quiesce_gc()
    barrier()
    log("forward timed region")
    prof_host = None
    if os.environ.get("BENCH_CPROFILE"):  # debugging aid: where the host time of the timed region goes
        import cProfile
        prof_host = cProfile.Profile()
        prof_host.enable()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    enqueue = time.perf_counter() - t0
    if prof_host is not None:
        prof_host.disable()
        import pstats
        pstats.Stats(prof_host, stream=sys.stderr).sort_stats("tottime").print_stats(14)
    barrier()
    elapsed = time.perf_counter() - t0
    log(f"forward done: {elapsed / a.steps * 1e3:.3f} ms per step, host enqueue {enqueue / a.steps * 1e3:.3f} ms")
    if dist is not None:
        elapsed = max_over_ranks(elapsed, dev, dist)
    assert os.environ.get("BENCH_NO_CHECK") or torch.isfinite(out).all()  # BENCH_NO_CHECK: timing-only ablation builds

    # ---- per-kernel device time (HIP events on the launch stream), separate from the timed region
    prof = {}
    _lib.set_profiler(prof)
    psteps = max(3, min(10, a.steps))
    for _ in range(psteps):
        step()
    torch.cuda.synchronize()
    _lib.set_profiler(None)
    per_kernel = {}
    for name, evs in prof.items():
        ms = [e0.elapsed_time(e1) for e0, e1 in evs]
        per_kernel[name] = {"launches_per_step": len(ms) / psteps, "avg_ms": sum(ms) / len(ms),
                            "ms_per_step": sum(ms) / psteps}
    # every forward kernel against both roofs: algorithmic HBM bytes and issued f16 MFMA FLOPs over its launch time
    kernel_rooflines = {}
    for name, v in sorted(per_kernel.items()):
        sec = v["avg_ms"] * 1e-3
        row = {"launches_per_step": round(v["launches_per_step"], 2), "avg_launch_ms": round(v["avg_ms"], 4)}
        nbytes = algorithmic_bytes(name, min(K, L), 16 * ((min(K, L) + 15) // 16))
        if nbytes:
            row["hbm_GBps"] = round(nbytes * B * L / sec / 1e9, 1)
            row["hbm_frac"] = round(nbytes * B * L / sec / 1e9 / PEAK_HBM_GBPS, 4)
        if name in FLOP_F16_ISSUED:
            tf = FLOP_F16_ISSUED[name] * (16 * ((min(K, L) + 15) // 16) / 48) * B * L / sec / 1e12  # tables are per K = 48
            row["mfma_f16_TFLOPs"] = round(tf, 1)
            row["mfma_frac"] = round(tf / PEAK_F16_MFMA_TFLOPS, 4)
        kernel_rooflines[name] = row
    dominant = max(per_kernel, key=lambda n: per_kernel[n]["ms_per_step"])
    dk = per_kernel[dominant]
    roofline = None
    traffic = None
    try:  # HBM bytes per launch from the committed PMC passes: only for the same workload AND the same library build
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if (pm["workload"] == {"B": B, "L": L, "K": K} and dominant in pm["kernels"]
                and pm.get("library_sha256") == library_sha256()):
            kk = pm["kernels"][dominant]
            traffic = {"bytes_per_launch": int((2 * kk["FETCH_SIZE_KB"] + kk["WRITE_SIZE_KB"]) * 1024),
                       "algorithmic_bytes_per_launch": 2 * B * L * 8 * ((K + 15) // 16) * 1024 if "update" in dominant
                       else B * L * 8 * ((K + 15) // 16) * 1024,
                       "source": pm["source"], "correction": pm["correction"], "library_sha256": pm["library_sha256"]}
    except (OSError, KeyError, ValueError):
        pass
    if dominant in FLOP_AS_WRITTEN:
        sec = dk["avg_ms"] * 1e-3
        ach = FLOP_F16_ISSUED[dominant] * B * L / sec / 1e12
        asw = FLOP_AS_WRITTEN[dominant] * B * L / sec / 1e12
        roofline = {"bound": "mfma", "kernel": dominant, "achieved": round(ach, 2), "peak": PEAK_F16_MFMA_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / PEAK_F16_MFMA_TFLOPS, 4),
                    "useful_frac": round(ach / 3.0 / PEAK_F16_MFMA_TFLOPS, 4), "traffic": traffic,
                    "avg_launch_ms": round(dk["avg_ms"], 4),
                    "as_written_f32_tflops": round(asw, 2),
                    "as_written_vs_f32_mfma_peak": round(asw / PEAK_F32_MFMA_TFLOPS, 4),
                    "note": "achieved = f16 MFMA FLOPs the kernel issues (3 split products per fp32 product; "
                            "per-edge 128x128 terms only) / HIP-event launch time, against the dense f16 MFMA peak; "
                            "useful_frac = the fp32-equivalent FLOPs executed (a third of the issued ones) / same peak; "
                            "as_written = the reference's fp32 FLOPs for the op this launch replaces / same time. "
                            "The kernel is VALU-issue bound (GELU + f16 splitting do not co-issue with MFMA here), "
                            "see DESIGN.md section 4"}

    if roofline is None:
        # tiny workloads can make a non-matrix kernel (node update, k-NN) the dominant one: price it against HBM
        Kq = min(K, L)
        nbytes = algorithmic_bytes(dominant, Kq, 16 * ((Kq + 15) // 16)) or 0
        ach = nbytes * B * L / (dk["avg_ms"] * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(ach, 2), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                    "frac": round(ach / PEAK_HBM_GBPS, 4), "traffic": traffic, "avg_launch_ms": round(dk["avg_ms"], 4),
                    "note": "achieved = algorithmic bytes of the launch (DESIGN.md section 4.1) / HIP-event launch time"}

    # ---- secondary: reduced-precision forward (one f16 product per fp32 product), same workload and the L=4096 shape
    f16 = {}
    quiesce_gc()
    for tag, (Bq, Lq) in ({} if a.no_f16_leg else {"B%dxL%d" % (B, L): (B, L), "B4xL4096": (4, 4096)}).items():
        bq = b if (Bq, Lq) == (B, L) else synthetic_batch(Bq, Lq, seed=4000 + rank, device=dev)
        aq = (bq["X"], bq["S"], bq["mask"], bq["chain_M"], bq["residue_idx"], bq["chain_encoding_all"])
        res = {}
        for mode in ("fp32", "f16"):
            model.set_precision(mode)
            with torch.no_grad():
                for _ in range(2):
                    o = model(*aq, fix_order=bq["fix_order"])
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    o = model(*aq, fix_order=bq["fix_order"])
                torch.cuda.synchronize()
            res[mode] = (Bq * Lq * 5 / (time.perf_counter() - t1), o)
        model.set_precision("fp32")
        f16[tag] = {"residues_per_s_fp32": round(res["fp32"][0], 1), "residues_per_s_f16": round(res["f16"][0], 1),
                    "max_abs_dlogp": round(float((res["f16"][1] - res["fp32"][1]).abs().max()), 5)}

    log("f16 leg done")
    # ---- secondary: many sequences on ONE structure (the EnergyMPNN workload, energy_mpnn.py:42-62): the same B x L rows,
    # structure side passed once, against the same call with the structure repeated B times as the reference does
    shared = None
    if not a.no_f16_leg and B > 1:
        one = synthetic_batch(1, L, seed=5000 + rank, device=dev)
        side1 = (one["X"], one["mask"], one["residue_idx"], one["chain_encoding_all"])
        sideB = tuple(t.repeat(B, *([1] * (t.dim() - 1))) for t in side1)
        cmB = one["chain_M"].repeat(B, 1)
        rates = {}
        for tag, sd in (("shared", side1), ("repeated", sideB)):
            with torch.no_grad():
                for _ in range(2):
                    o = model(sd[0], b["S"], sd[1], cmB, sd[2], sd[3], fix_order=b["fix_order"])
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    o = model(sd[0], b["S"], sd[1], cmB, sd[2], sd[3], fix_order=b["fix_order"])
                torch.cuda.synchronize()
            rates[tag] = (B * L * 5 / (time.perf_counter() - t1), o)
        shared = {"workload": f"{B} sequences x L={L} on one structure, k={K}, teacher-forced forward (per rank)",
                  "residues_per_s_shared": round(rates["shared"][0], 1),
                  "residues_per_s_repeated": round(rates["repeated"][0], 1),
                  "bit_identical": bool(torch.equal(rates["shared"][1], rates["repeated"][1])),
                  "note": "featurizer + encoder once per distinct structure (mpnn_dec_message_shared); SURVEY 8f N2"}
    log("shared-structure leg done")
    # ---- secondary: the same forward replayed as ONE HIP graph (mpnn_amd.graphs.GraphedForward): what the ~26 launch gaps cost
    graphed = None
    if not a.no_f16_leg:
        try:
            from mpnn_amd.graphs import GraphedForward
            gf = GraphedForward(model, *args, b["fix_order"])
            for _ in range(3):
                gf.graph.replay()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                gf.graph.replay()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / a.steps
            graphed = {"residues_per_s": round(B * L / dt, 1), "ms_per_step": round(dt * 1e3, 3),
                       "bit_identical_to_eager": bool(torch.equal(gf.out, out)),
                       "note": "per rank; one graph launch per forward instead of ~26 kernel launches; `value` above is the eager path"}
            del gf
        except Exception as e:  # the graph path is an extra: never let it take the line down
            graphed = {"error": repr(e)[:200]}
    log("graph-replay leg done")
    # ---- secondary: BASELINE configs[0] stand-in (C1): ragged token-budget batches of 100 chains, forward only
    c1 = None
    if not a.no_f16_leg:
        w = c1_workload(device=dev)

        def c1_pass():
            with torch.no_grad():
                for bb in w["batches"]:
                    o = model(bb["X"], bb["S"], bb["mask"], bb["chain_M"], bb["residue_idx"], bb["chain_encoding_all"],
                              fix_order=bb["fix_order"])
            return o
        c1_pass()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            o = c1_pass()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / 5
        c1 = {"workload": w["describe"], "valid_residues_per_s": round(w["residues"] / dt, 1),
              "padded_residues_per_s": round(w["padded"] / dt, 1), "ms_per_pass": round(dt * 1e3, 3),
              "note": "per rank; the CPU side of the same workload is cpu_baseline.c1_forward"}
        assert torch.isfinite(o).all()
    log("C1 leg done")
    # ---- secondary: the edge-list model family on the same B x L proteins in the flat layout
    sparse = None
    if not a.no_f16_leg:
        from mpnn_amd.sparse import ProteinMPNN as SparseMPNN
        from mpnn_amd.synthetic import synthetic_flat_batch

        torch.manual_seed(0)
        sm = SparseMPNN(num_neighbors=K).eval().to(dev)
        fb = synthetic_flat_batch([L] * B, seed=6000 + rank, device=dev)
        so_ = fb.pop("decoding_order")
        with torch.no_grad():
            for _ in range(2):
                o = sm(**fb, decoding_order=so_)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                o = sm(**fb, decoding_order=so_)
            torch.cuda.synchronize()
        assert torch.isfinite(o).all()
        sparse = {"workload": f"mpnn_amd.sparse.ProteinMPNN (protein_mpnn.py), {B} graphs x {L} nodes, knn {K} with self "
                              "loops, 3 enc + 3 dec, flat layout in / out (per rank)",
                  "residues_per_s": round(B * L * 5 / (time.perf_counter() - t1), 1),
                  "note": "same kernels as the dense path (mpnn_edge_features_biased, mpnn_dec_message_valid); includes "
                          "the flat <-> padded scatter and one host read of the largest graph size"}
    log("sparse leg done")
    train = None
    if a.train_batch > 0:
        train = train_leg(a, dev, rank, world, dist, barrier, _lib)

    log("train leg done")
    sampling = None
    if a.samples > 0:
        sampling = sampling_leg(a, dev, rank, world, dist, barrier)

    log("sampling leg done")
    if rank == 0:
        total_res = B * L * world * a.steps
        value = total_res / elapsed
        line = {
            "metric": "residues/sec (enc+dec fwd) at L=512,k=48",
            "value": round(value, 1), "unit": "residues/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3),
            "host_enqueue_ms_per_step": round(enqueue / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 (products as 3x f16-split MFMA, fp32 accumulate)", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: synthetic backbones, batch {B} x L={L}, k={K}, hidden 128, "
                                   "3 enc + 3 dec layers, fp32 forward + teacher-forced decoder",
                       "global_batch": B * world, "seq_len": L, "parallelism": f"dp{world} (proteins sharded, no collective)"},
            "ranks": {"world_size": dist.get_world_size() if dist is not None else 1,
                      "backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else ""))
                      if dist is not None else "none (single process)"},
            "rccl_ranks": dist.get_world_size() if (dist is not None and dist.get_backend() == "nccl") else 0,
            "whole_forward_tflops_as_written": round(FLOP_FWD_PER_RES * value / 1e12, 2),
            "roofline": roofline,
            "kernels_ms_per_step": {k: round(v["ms_per_step"], 3) for k, v in sorted(per_kernel.items())},
            "kernel_rooflines": kernel_rooflines,
            "shared_structure": shared,
            "c1_forward": c1,
            "sparse_forward": sparse,
        }
        if graphed:
            line["forward_hip_graph"] = graphed
        line["forward_f16_products"] = dict(f16, note="per-rank; one f16 MFMA product per fp32 product instead of three; "
                                            "fp32 accumulate / GELU / LayerNorm; tolerance in include/mpnn_hip.h")
        if train is not None:
            line["train_step"] = train
        if sampling is not None:
            line["sampling"] = sampling
        if not a.no_f16_leg:
            line["batch_assembly"] = assembly_leg(a, dev, with_cpu=not a.no_cpu_baseline and world == 1)
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(2, L, K, iters=5, warm=3)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
