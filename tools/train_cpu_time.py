#!/usr/bin/env python3
"""Host enqueue time vs total time of the training step (is the step CPU- or GPU-bound?)."""
import sys, time
import torch
sys.path.insert(0, ".")
from mpnn_amd import ProteinMPNN, training
from mpnn_amd.synthetic import synthetic_batch

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = ProteinMPNN(k_neighbors=48, augment_eps=0.0, dropout=0.0).train().to(dev)
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-2)
b = synthetic_batch(20, 512, seed=1, device=dev)
batch = (b["X"], b["S"], b["mask"], b["chain_M"], b["residue_idx"], b["chain_encoding_all"])
for _ in range(3):
    training.train_step(model, opt, batch, fix_order=b["fix_order"])
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    training.train_step(model, opt, batch, fix_order=b["fix_order"])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / n:.2f} ms/step, total {1e3 * (t2 - t0) / n:.2f} ms/step")
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    training.train_step(model, opt, batch, fix_order=b["fix_order"])
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
