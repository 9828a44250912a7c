#!/bin/bash
# GPU check of the training path: gradient parity tests, then the bench line's training leg
timeout -k 10 400 python -u -m pytest tests/test_gpu_train.py -q --timeout 180 --timeout-method thread -s > gpurun_out/train_test.log 2>&1
grep -E "worst relative|passed|failed|Error|mismatch|assert" gpurun_out/train_test.log | cut -c1-600 | tail -12
grep -q " failed" gpurun_out/train_test.log && exit 1
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_train.json 2>gpurun_out/bench_train.err || { tail -5 gpurun_out/bench_train.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/bench_train.json')); t=d['train_step']
print('fwd res/s', d['value'], 'train res/s', t['value'], 'ms/step', t['ms_per_step'], 'hip ms', t['hip_kernels_total_ms'])
print(t['hip_kernels_ms_per_step'])"
