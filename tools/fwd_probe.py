"""Forward at B32 x L512: host enqueue time vs total per step (python tools/fwd_probe.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd import ProteinMPNN
from mpnn_amd.synthetic import synthetic_batch

dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = ProteinMPNN(k_neighbors=48, augment_eps=0.0, dropout=0.0).eval().to(dev)
b = synthetic_batch(32, 512, seed=1000, device=dev)
args = (b["X"], b["S"], b["mask"], b["chain_M"], b["residue_idx"], b["chain_encoding_all"])
with torch.no_grad():
    for _ in range(5):
        model(*args, fix_order=b["fix_order"])
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            model(*args, fix_order=b["fix_order"])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"rep {rep}: enqueue {(t1 - t0) / 20 * 1e3:.3f} ms/step, total {(t2 - t0) / 20 * 1e3:.3f} ms/step", flush=True)
