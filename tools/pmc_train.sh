#!/bin/bash
# On the GPU box, from the repo root: SQ counter passes over the training step alone (one rocprofv3 --pmc run per pass).
#   tools/pmc_train.sh <tag> "<counters of pass 1>" ["<counters of pass 2>" ...]   -> gpurun_out/pmc_train_<tag>.txt
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out
: > gpurun_out/pmc_train_$tag.txt
i=0
for pass in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmc_tr_$i
  timeout -s KILL 150 rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_tr_$i -- python3 tools/train_only.py 2 > /dev/null 2> gpurun_out/pmc_train_$tag.err || { echo "pass $i failed"; tail -5 gpurun_out/pmc_train_$tag.err; continue; }
  python3 tools/pmc_summary.py /tmp/pmc_tr_$i >> gpurun_out/pmc_train_$tag.txt
  echo "pass $i done"
done
