#!/bin/bash
# shader clock the driver reports while the sampler's small level kernels run, and while a throughput kernel runs
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
echo "--- during sampling (1 sample, L=512)"
python - <<'PY' &
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from mpnn_amd import ProteinMPNN
from mpnn_amd.synthetic import synthetic_batch
dev = torch.device("cuda:0")
model = ProteinMPNN(k_neighbors=48, augment_eps=0.0, dropout=0.0, use_virtual_center=True).eval().to(dev)
b = synthetic_batch(1, 512, seed=3000, device=dev)
args = (b["X"], b["mask"], b["chain_M"], b["residue_idx"], b["chain_encoding_all"])
t0 = time.time()
while time.time() - t0 < 12:
    model.sample(*args, n_samples=1, seed=1)
torch.cuda.synchronize()
bb = synthetic_batch(32, 512, seed=1, device=dev)
a2 = (bb["X"], bb["S"], bb["mask"], bb["chain_M"], bb["residue_idx"], bb["chain_encoding_all"])
print("FORWARD PHASE", flush=True)
t0 = time.time()
with torch.no_grad():
    while time.time() - t0 < 8:
        for _ in range(20):
            model(*a2, fix_order=bb["fix_order"])
        torch.cuda.synchronize()
PY
sleep 6
for i in 1 2 3; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -2; sleep 1; done
sleep 6
echo "--- during the B32xL512 forward"
for i in 1 2 3; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -2; sleep 1; done
wait
