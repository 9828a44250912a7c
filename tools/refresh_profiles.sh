#!/bin/bash
# On the GPU box, from the repo root: refresh the round's evidence under gpurun_out/ (copied into profiles/ afterwards).
#   full default bench line, rocprofv3 kernel stats of the same command, training step at the token-budget shapes
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python bench.py > gpurun_out/full_bench.json 2> gpurun_out/full_bench.err
echo "bench done"
rm -rf /tmp/prof_stats
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_bench.json 2> gpurun_out/prof_bench.err
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats.csv
rm -rf /tmp/prof_fwd
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fwd -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f16-leg --train-batch 0 --samples 0 > gpurun_out/prof_fwd_bench.json 2> gpurun_out/prof_fwd_bench.err
cp $(find /tmp/prof_fwd -name "*kernel_stats.csv" | head -1) gpurun_out/fwd_kernel_stats.csv
echo "rocprof done"
: > gpurun_out/train_shapes.jsonl
for shape in "128 78" "256 39" "512 19"; do
  set -- $shape
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --length $1 --batch $2 --train-batch $2 --samples 0 --no-f16-leg --no-cpu-baseline 2>/dev/null \
    | python -c "
import json, sys
d = json.loads(sys.stdin.read()); t = d['train_step']
print(json.dumps({'L': $1, 'B': $2, 'forward_residues_per_s': d['value'], 'train_residues_per_s': t['value'], 'train_ms_per_step': t['ms_per_step'], 'host_enqueue_ms_per_step': t['host_enqueue_ms_per_step']}))" >> gpurun_out/train_shapes.jsonl
done
cat gpurun_out/train_shapes.jsonl
