#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/gpu_check.sh [tag]
tag=${1:-run}
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -q --tb=line --timeout 180 --timeout-method thread > gpurun_out/pytest_$tag.log 2>&1
rc=$?
tail -4 gpurun_out/pytest_$tag.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -E "^E |FAILED|Error" gpurun_out/pytest_$tag.log | cut -c1-300 | head -20; exit $rc; fi
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/bench_$tag.err > gpurun_out/bench_$tag.json || { tail -5 gpurun_out/bench_$tag.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/bench_$tag.json"))
print("res/s", d["value"], "ms/step", d["ms_per_step"])
print("roofline", d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"])
print(d["kernels_ms_per_step"])
PY
