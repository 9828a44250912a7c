#!/bin/bash
# A/B kernel variants on the training step: per-kernel ms/step for each variants/libmpnn_<tag>.so
for so in variants/libmpnn_*.so; do
  tag=$(basename $so .so); tag=${tag#libmpnn_}
  BENCH_NO_CHECK=1 MPNN_HIP_LIB=$PWD/$so timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f16-leg --samples 0 2>/dev/null > gpurun_out/abt_$tag.json || { echo "$tag FAILED"; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/abt_$tag.json"))["train_step"]
k=d["hip_kernels_ms_per_step"]
print(f"{'$tag':10s} step {d['ms_per_step']:6.3f} ms | " + " ".join(f"{n.replace('mpnn_','')}={v:.3f}" for n,v in k.items() if v>0.3))
PY
done
