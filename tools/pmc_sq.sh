#!/bin/bash
# On the GPU box, from the repo root: SQ counter passes over the forward-only bench (one rocprofv3 --pmc run per pass).
#   tools/pmc_sq.sh <tag> "<counters of pass 1>" ["<counters of pass 2>" ...]   -> gpurun_out/pmc_sq_<tag>.txt
export TMPDIR=/tmp
tag=$1; shift
F="--steps 3 --warmup 1 --no-cpu-baseline --no-f16-leg --train-batch 0 --samples 0"
mkdir -p gpurun_out
: > gpurun_out/pmc_sq_$tag.txt
i=0
for pass in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmc_sq_$i
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_sq_$i -- python3 bench.py $F > /dev/null 2> gpurun_out/pmc_sq_$tag.err || { echo "pass $i failed"; tail -5 gpurun_out/pmc_sq_$tag.err; exit 1; }
  python3 tools/pmc_summary.py /tmp/pmc_sq_$i >> gpurun_out/pmc_sq_$tag.txt
  echo "pass $i done"
done
