#!/bin/bash
# On the GPU box: per-kernel durations of the sampler in each launch mode: tools/sample_trace.sh <n_samples> [modes...]
export TMPDIR=/tmp
ns=$1; shift
for m in ${@:-level layer}; do
  rm -rf /tmp/ps_$m
  MPNN_SAMPLE_LAUNCHES=$m timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$m -- python3 tools/sample_probe.py $ns > gpurun_out/sp_$m.log 2>&1 || { echo "$m failed"; exit 1; }
  echo "== $m"; python3 tools/kstats.py /tmp/ps_$m 6; grep n_samples gpurun_out/sp_$m.log
done
