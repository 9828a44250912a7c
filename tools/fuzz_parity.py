"""Randomised parity sweep on the GPU: python tools/fuzz_parity.py [n_cases] [seed]

Each case draws a model configuration (k, cutoff, occupancy, virtual centre, layer counts), a ragged multi-chain batch and
a decoding order, and checks the HIP path against the oracle to the bars of tests/test_gpu_parity.py (indices and masks
bit-exact, log-probs 1e-4 relative + 1e-5), for every third case loss and gradients (relative L2 <= 1e-3), and for
every fourth case the sampler contract (a teacher-forced forward on the drawn sequences returns the sampler's log-probs);
every fifth case the f16-product mode (max |d log p| <= 5e-2), every seventh the shared-structure forward (bit-identical).
Prints one line per case and a summary; exit status 1 on any failure."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import mpnn_amd  # noqa: E402
from mpnn_amd import engine, training  # noqa: E402
from mpnn_amd.synthetic import fill_parameters_, synthetic_batch  # noqa: E402
from oracle import mpnn_oracle as orc  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for case in range(n_cases):
        B = int(rng.integers(1, 5))
        L = int(rng.integers(2, 150))
        kw = dict(k_neighbors=int(rng.choice([3, 16, 30, 32, 48, 56, 64])))
        if rng.random() < 0.3:
            kw["edge_cutoff"] = float(rng.uniform(8, 20))
        if rng.random() < 0.3:
            kw["occupancy_cutoff"] = float(rng.uniform(5, 10))
        if rng.random() < 0.4:
            kw["use_virtual_center"] = True
        if rng.random() < 0.3:
            kw["num_encoder_layers"] = int(rng.integers(1, 4))
            kw["num_decoder_layers"] = int(rng.integers(1, 4))
        lengths = [L] + [int(rng.integers(0 if rng.random() < 0.1 else 1, L + 1)) for _ in range(B - 1)]
        b = synthetic_batch(B=B, L=L, seed=int(rng.integers(1 << 30)), lengths=lengths,
                            n_chains=int(rng.integers(1, 4)) if L >= 6 else 1, design_fraction=float(rng.uniform(0.3, 1)))
        model = mpnn_amd.ProteinMPNN(augment_eps=0.0, dropout=0.0, **kw).eval()
        fill_parameters_(model, seed=int(rng.integers(100)))
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        cfg = orc.Config(k_neighbors=kw["k_neighbors"], edge_cutoff=kw.get("edge_cutoff"),
                         use_virtual_center=kw.get("use_virtual_center", False),
                         occupancy_cutoff=kw.get("occupancy_cutoff"),
                         num_encoder_layers=kw.get("num_encoder_layers", 3),
                         num_decoder_layers=kw.get("num_decoder_layers", 3))
        model = model.cuda()
        g = {k: v.cuda() for k, v in b.items()}
        args = (g["X"], g["S"], g["mask"], g["chain_M"], g["residue_idx"], g["chain_encoding_all"])
        msgs = []
        with torch.no_grad():
            out = engine.forward(model, *args, fix_order=g["fix_order"], return_intermediates=True)
            ref = orc.forward(sd, cfg, b["X"], b["S"], b["mask"], b["chain_M"], b["residue_idx"],
                              b["chain_encoding_all"], fix_order=b["fix_order"], formulation="edge", keep=True)
        for name in ("E_idx", "mask_bw", "mask_fw"):
            if not torch.equal(out[name].cpu(), ref[name]):
                msgs.append(f"{name} differs")
        m = b["mask"].bool()
        err = (out["log_probs"].cpu() - ref["log_probs"]).abs()[m]
        tol = 1e-5 + 1e-4 * ref["log_probs"].abs()[m]
        if err.numel() and not bool((err <= tol).all()):
            msgs.append(f"log_probs max |d| {err.max().item():.2e}")
        if not bool(torch.isfinite(out["log_probs"]).all()):
            msgs.append("non-finite log_probs")
        grad_note = ""
        if case % 3 == 0 and m.any():
            model.train()
            lp = model(*args, fix_order=g["fix_order"])
            _, loss = training.loss_smoothed(g["S"], lp, g["mask"] * g["chain_M"])
            loss.backward()
            sdg = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}
            lp_ref = orc.forward(sdg, cfg, b["X"], b["S"], b["mask"], b["chain_M"], b["residue_idx"],
                                 b["chain_encoding_all"], fix_order=b["fix_order"], formulation="edge")
            _, loss_ref = orc.loss_smoothed(b["S"], lp_ref, b["mask"] * b["chain_M"])
            loss_ref.backward()
            total = sum(float(v.grad.double().norm() ** 2) for v in sdg.values() if v.grad is not None) ** 0.5
            worst = 0.0
            for n, p in model.named_parameters():
                want = sdg[n].grad if sdg[n].grad is not None else torch.zeros_like(sdg[n])
                got = p.grad.cpu() if p.grad is not None else torch.zeros_like(want)
                worst = max(worst, float((got - want).double().norm()) / max(total, 1e-30))
            if abs(loss.item() - loss_ref.item()) > 1e-5 * max(1.0, abs(loss_ref.item())) or worst > 1e-3:
                msgs.append(f"training: loss {loss.item():.6g} vs {loss_ref.item():.6g}, worst gradient error {worst:.2e}")
            grad_note = f" grad {worst:.1e}"
        if case % 4 == 1 and m.any():
            # sampler contract: the teacher-forced forward on the drawn sequences, same orders,
            # returns the sampler's log-probs
            model.eval()
            ns = int(rng.integers(1, 4))
            with torch.no_grad():
                smp = model.sample(g["X"], g["mask"], g["chain_M"], g["residue_idx"], g["chain_encoding_all"],
                                   S_true=g["S"], n_samples=ns, temperature=float(rng.choice([0.1, 1.0])),
                                   seed=int(rng.integers(1 << 20)))
                worst_s = 0.0
                for q in range(ns):
                    lp_tf = model(g["X"], smp["S"][q], g["mask"], g["chain_M"], g["residue_idx"],
                                  g["chain_encoding_all"], fix_order=smp["order"][q])
                    d = (lp_tf - smp["log_probs"][q]).abs()[g["mask"].bool()]
                    worst_s = max(worst_s, float(d.max()) if d.numel() else 0.0)
                fixed = (g["chain_M"] * g["mask"]) == 0
                if not torch.equal(smp["S"][0][fixed], g["S"][fixed]):
                    msgs.append("sampler changed a fixed residue")
            if worst_s > 1e-4:
                msgs.append(f"sampler vs teacher-forced forward: {worst_s:.2e}")
            grad_note += f" sample {worst_s:.1e}"
        if case % 5 == 2 and m.any():
            # reduced-precision forward (one f16 product): bar of include/mpnn_hip.h, max |d log p| <= 5e-2
            model.eval()
            model.set_precision("f16")
            with torch.no_grad():
                fast = model(*args, fix_order=g["fix_order"])
            model.set_precision("fp32")
            d16 = float((fast.cpu() - out["log_probs"].cpu()).abs()[m].max())
            if d16 > 5e-2:
                msgs.append(f"f16 products: {d16:.2e}")
            grad_note += f" f16 {d16:.1e}"
        if case % 7 == 3:
            # shared-structure forward: three sequences on structure 0, given once vs repeated - bit-identical
            model.eval()
            S3 = torch.randint(0, 21, (3, L), generator=torch.Generator().manual_seed(case)).cuda()
            o3 = torch.argsort(torch.rand(3, L, generator=torch.Generator().manual_seed(case + 1)), dim=-1).cuda()
            side = [t[:1] for t in (g["X"], g["mask"], g["residue_idx"], g["chain_encoding_all"])]
            with torch.no_grad():
                a3 = model(side[0], S3, side[1], g["chain_M"][:1], side[2], side[3], fix_order=o3)
                b3 = model(side[0].repeat(3, 1, 1, 1), S3, side[1].repeat(3, 1), g["chain_M"][:1].repeat(3, 1),
                           side[2].repeat(3, 1), side[3].repeat(3, 1), fix_order=o3)
            if not torch.equal(a3, b3):
                msgs.append("shared-structure forward differs from the repeated batch")
            grad_note += " shared ok"
        status = "ok" if not msgs else "FAIL " + "; ".join(msgs)
        bad += bool(msgs)
        print(f"case {case:3d} B={B} L={L:3d} lengths={lengths} {kw}: {status}"
              f" (max |d logp| {err.max().item() if err.numel() else 0:.1e}{grad_note})", flush=True)
    print(f"{n_cases - bad} of {n_cases} cases passed", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
