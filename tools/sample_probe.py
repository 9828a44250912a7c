"""Host-enqueue vs device time of mpnn_amd sampling (one L=512 structure, virtual centre): python tools/sample_probe.py [n_samples]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd import ProteinMPNN
from mpnn_amd.synthetic import synthetic_batch

dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = ProteinMPNN(k_neighbors=48, augment_eps=0.0, dropout=0.0, use_virtual_center=True).eval().to(dev)
b = synthetic_batch(1, 512, seed=3000, device=dev)
args = (b["X"], b["mask"], b["chain_M"], b["residue_idx"], b["chain_encoding_all"])
for ns in [int(v) for v in sys.argv[1:]] or [1, 8, 64]:
    model.sample(*args, n_samples=ns, seed=1)
    torch.cuda.synchronize()
    enq = tot = 0.0
    for i in range(5):
        t0 = time.perf_counter()
        out = model.sample(*args, n_samples=ns, seed=2 + i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        enq += t1 - t0
        tot += t2 - t0
    print(f"n_samples {ns:3d}: levels {out['n_levels']}, host enqueue {enq / 5 * 1e3:.2f} ms, total {tot / 5 * 1e3:.2f} ms, "
          f"{ns * 512 * 5 / tot / 1e6:.2f} M sampled residues/s", flush=True)
