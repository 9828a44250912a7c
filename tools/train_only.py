#!/usr/bin/env python3
"""Training steps alone (bench.py's train_step leg without the other legs), for rocprofv3 kernel tables:
   rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -- python3 tools/train_only.py [steps [B L]]
(B = 1, L = 64 makes the GPU work negligible: the time per step is then the host's enqueue time)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mpnn_amd import ProteinMPNN, training
from mpnn_amd.optim import AdamW
from mpnn_amd.synthetic import synthetic_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = ProteinMPNN(k_neighbors=48, augment_eps=0.1, dropout=0.1).train().to(dev)
opt = AdamW(model.parameters(), lr=1e-3, weight_decay=1e-2)
Bt, Lt = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (20, 512)
b = synthetic_batch(Bt, Lt, seed=2000, device=dev)
batch = (b["X"], b["S"], b["mask"], b["chain_M"], b["residue_idx"], b["chain_encoding_all"])
for _ in range(2):
    training.train_step(model, opt, batch, world_size=1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    training.train_step(model, opt, batch, world_size=1)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step over {steps} steps (+ 2 warm-up steps in the trace); "
      f"host enqueue {t_enq / steps * 1e3:.3f} ms per step")
