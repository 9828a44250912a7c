#!/usr/bin/env python3
"""Forward throughput at other shapes (BASELINE configs[3]: L=4096 complexes; short chains), one JSON line each."""
import json, sys, time
import torch
sys.path.insert(0, ".")
from mpnn_amd import ProteinMPNN
from mpnn_amd.synthetic import synthetic_batch

dev = torch.device("cuda", 0)
for B, L, K in [(1, 4096, 48), (4, 4096, 48), (128, 128, 48), (64, 256, 48), (32, 512, 48), (8, 2048, 48)]:
    model = ProteinMPNN(k_neighbors=K, augment_eps=0.0, dropout=0.0).eval().to(dev)
    b = synthetic_batch(B, L, seed=5, device=dev)
    args = (b["X"], b["S"], b["mask"], b["chain_M"], b["residue_idx"], b["chain_encoding_all"])
    with torch.no_grad():
        for _ in range(3):
            model(*args, fix_order=b["fix_order"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            model(*args, fix_order=b["fix_order"])
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(json.dumps({"B": B, "L": L, "K": K, "ms_per_forward": round(dt * 1e3, 3), "residues_per_s": round(B * L / dt)}), flush=True)
