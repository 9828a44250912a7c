#!/bin/bash
# On the GPU box, from the repo root: HBM traffic counters of the forward kernels (separate --pmc passes, nothing else traced).
#   -> gpurun_out/pmc_traffic.txt (per kernel and dispatch) and gpurun_out/pmc_traffic.json (what bench.py's roofline.traffic reads)
set -e
export TMPDIR=/tmp
F="--steps 3 --warmup 1 --no-cpu-baseline --no-f16-leg --train-batch 0 --samples 0"
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 bench.py $F > /dev/null 2> gpurun_out/pmc_$c.err
  echo "pass $c done"
done
python3 - <<'PY'
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"/tmp/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
entry = {"k_enc_edge_update": "mpnn_enc_edge_update", "k_edge_features": "mpnn_edge_features", "k_knn_topk": "mpnn_knn_topk",
         "k_node_update": "mpnn_node_update"}
lines = ["# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 3 --warmup 1 forward leg only "
         "(B=32 x L=512, k=48), average per dispatch, unit KB as reported",
         "# gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts wide streaming reads at 1/2 -> "
         "HBM-side bytes ~= 2*FETCH_SIZE + WRITE_SIZE (requests that hit the Infinity Cache are included)"]
kern = {}
for k, cs in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
    if k.startswith("void at::") or k.startswith("__amd"):
        continue
    n = max(len(v) for v in cs.values())
    avg = {c: sum(v) / len(v) for c, v in cs.items()}
    short = k.replace("(anonymous namespace)::", "").split("(")[0][:70]
    lines.append(f"{short:72s} n={n:3d} " + " ".join(f"{c}={v:.4g}" for c, v in sorted(avg.items())))
    name = None
    if "k_message" in k and "ILi3ELb1" in k: name = "mpnn_dec_message"
    elif "k_message" in k and "ILi3ELb0" in k: name = "mpnn_enc_message"
    else:
        for pat, e in entry.items():
            if pat in k and ("ILi3" in k or "ILi" not in k): name = e
    if name and name not in kern:
        kern[name] = {"FETCH_SIZE_KB": avg.get("FETCH_SIZE", 0.0), "WRITE_SIZE_KB": avg.get("WRITE_SIZE", 0.0)}
open("gpurun_out/pmc_traffic.txt", "w").write("\n".join(lines) + "\n")
import hashlib
json.dump({"workload": {"B": 32, "L": 512, "K": 48},
           "library_sha256": hashlib.sha256(open("mpnn_amd/libmpnn_hip.so", "rb").read()).hexdigest(),
           "source": "profiles/pmc_traffic.txt (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes; tools/pmc_traffic.sh)",
           "correction": "bytes = 2 * FETCH_SIZE (gfx950 counts wide streaming reads at half) + WRITE_SIZE, KB -> bytes x 1024",
           "kernels": kern}, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
print("\n".join(lines[:12]))
PY
