"""Soak of the training loop: python tools/soak_train.py [steps]
Token-budget batches of changing shape from a synthetic shard, the reference's recipe (dropout 0.1, noise 0.1, AdamW +
warm-up/cosine, clipping); reports throughput, loss trend and device memory (a leak or a shape-dependent cache shows
as growth of the allocated bytes after the first epoch)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpnn_amd  # noqa: E402
from mpnn_amd import training  # noqa: E402
from mpnn_amd.data import StructureShard, TokenBudgetBatcher  # noqa: E402
from mpnn_amd.synthetic import synthetic_entries  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.manual_seed(0)
shard = StructureShard.from_entries(synthetic_entries(600, seed=5, max_len=800), shuffle=False).to("cuda")
batcher = TokenBudgetBatcher(shard.lengths, batch_size=10000, drop_overflow=False, rng=np.random.default_rng(0))
model = mpnn_amd.ProteinMPNN(k_neighbors=48).cuda().train()
opt, sched = training.make_optimizer(model, lr=1e-3, total_steps=steps)
done, res, t0, log = 0, 0, time.perf_counter(), []
while done < steps:
    for ids in batcher:
        X, S, mask, _, chain_M, ridx, _, cenc = shard.assemble(ids)
        loss, _ = training.train_step(model, opt, (X, S, mask, chain_M, ridx, cenc), scheduler=sched, max_grad_norm=1.0)
        res += int(shard.lengths[ids].sum())
        done += 1
        if done % 50 == 0 or done == steps:
            per = float(loss) * 2000.0 / max(float((mask * chain_M).sum()), 1.0)
            torch.cuda.synchronize()
            log.append((done, per, torch.cuda.memory_allocated() >> 20, torch.cuda.max_memory_allocated() >> 20))
            print(f"step {done:4d}: loss/residue {per:.4f}, allocated {log[-1][2]} MiB (peak {log[-1][3]} MiB), "
                  f"{res / (time.perf_counter() - t0) / 1e3:.0f} k valid residues/s", flush=True)
        if done >= steps:
            break
assert all(np.isfinite(p) for _, p, _, _ in log)
assert all(torch.isfinite(p).all() for p in model.parameters())
assert log[-1][3] <= 1.05 * log[1][3] + 64, "peak memory keeps growing"
print("soak ok")
