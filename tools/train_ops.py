"""ATen operators (and their device kernels) inside one training step: python tools/train_ops.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
from mpnn_amd import ProteinMPNN, training
from mpnn_amd.optim import AdamW
from mpnn_amd.synthetic import synthetic_batch

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ProteinMPNN(k_neighbors=48, augment_eps=0.1, dropout=0.1).train().to(dev)
opt = AdamW(model.parameters(), lr=1e-3, weight_decay=1e-2)
b = synthetic_batch(20, 512, seed=1, device=dev)
batch = (b["X"], b["S"], b["mask"], b["chain_M"], b["residue_idx"], b["chain_encoding_all"])
for _ in range(3):
    training.train_step(model, opt, batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(2):
        training.train_step(model, opt, batch)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=60))
