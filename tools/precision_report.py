#!/usr/bin/env python3
"""Max error of the HIP path vs the CPU oracle per golden fixture (run on the GPU box)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden_names, load_golden
from test_gpu_parity import build_model, run_hip, run_oracle
for name in golden_names():
    rec, kw = load_golden(name)
    model = build_model(kw)
    out, ref = run_hip(model, rec), run_oracle(model, kw, rec)
    mask = rec["mask"].astype(bool)
    lp, rl = out["log_probs"].cpu().numpy(), ref["log_probs"].numpy()
    d = np.abs(lp - rl)
    rel = (d / np.maximum(np.abs(rl), 1e-1)).max()
    he = np.abs(out["h_E_enc"][-1].cpu().numpy()[mask] - ref["h_E_enc"][-1].numpy()[mask]).max()
    hv = max(np.abs(a.cpu().numpy() - b.numpy()).max() for a, b in zip(out["h_V_dec"], ref["h_V_dec"]))
    print(f"{name:24s} max|dlogp| {d.max():.2e}  max rel {rel:.2e}  h_E(enc3) {he:.2e}  h_V(dec) {hv:.2e}")
