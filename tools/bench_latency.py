#!/usr/bin/env python3
"""Small-batch forward latency: eager launches vs HIP-graph replay (mpnn_amd.graphs.GraphedForward)."""
import json, sys, time
import torch
sys.path.insert(0, ".")
from mpnn_amd import ProteinMPNN
from mpnn_amd.graphs import GraphedForward
from mpnn_amd.synthetic import synthetic_batch

dev = torch.device("cuda", 0)
model = ProteinMPNN(k_neighbors=48, augment_eps=0.0, dropout=0.0).eval().to(dev)
for B, L in [(1, 100), (1, 256), (1, 512), (4, 256), (8, 512)]:
    b = synthetic_batch(B, L, seed=5, device=dev)
    args = (b["X"], b["S"], b["mask"], b["chain_M"], b["residue_idx"], b["chain_encoding_all"])

    def timeit(fn, n=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    with torch.no_grad():
        eager = timeit(lambda: model(*args, fix_order=b["fix_order"]))
        ref = model(*args, fix_order=b["fix_order"]).clone()
    g = GraphedForward(model, *args, b["fix_order"])
    graphed = timeit(lambda: g(*args, b["fix_order"]))
    same = bool(torch.equal(g(*args, b["fix_order"]), ref))
    print(json.dumps({"B": B, "L": L, "eager_ms": round(eager * 1e3, 3), "graph_ms": round(graphed * 1e3, 3),
                      "bit_identical": same}), flush=True)
