#!/bin/bash
# per-kernel averages of the sampler (8 samples of one L=512 structure) for each variants/libmpnn_<tag>.so
export TMPDIR=/tmp
for so in variants/libmpnn_*.so; do
  tag=$(basename $so .so); tag=${tag#libmpnn_}
  rm -rf /tmp/ps_$tag
  MPNN_HIP_LIB=$PWD/$so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$tag -- python3 tools/sample_probe.py 8 > gpurun_out/sp_$tag.log 2>&1 || { echo "$tag FAILED"; continue; }
  python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("/tmp/ps_$tag/**/*kernel_stats.csv", recursive=True)[0])))
print("$tag".ljust(10), " ".join(f"{r['Name'].replace('(anonymous namespace)::','').split('(')[0][-22:]}={float(r['AverageNs'])/1e3:.1f}us" for r in rows[:3]),
      open("gpurun_out/sp_$tag.log").read().strip().splitlines()[-3][:90] if False else "")
PY
  grep "n_samples" gpurun_out/sp_$tag.log | tail -1
done
