#!/bin/bash
# On the GPU box: counter passes over the sampler (tools/sample_probe.py <n_samples>), one rocprofv3 --pmc run per pass.
#   tools/pmc_sample.sh <tag> <n_samples> "<counters of pass 1>" ["<counters of pass 2>" ...] -> gpurun_out/pmc_sample_<tag>.txt
export TMPDIR=/tmp
tag=$1; ns=$2; shift 2
mkdir -p gpurun_out
: > gpurun_out/pmc_sample_$tag.txt
i=0
for pass in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmc_s_$i
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_s_$i -- python3 tools/sample_probe.py $ns > /dev/null 2> gpurun_out/pmc_sample_$tag.err || { echo "pass $i failed"; tail -5 gpurun_out/pmc_sample_$tag.err; exit 1; }
  python3 tools/pmc_summary.py /tmp/pmc_s_$i | grep -E "decode_level|dec_layer_level|k_sample|k_message<3, false|k_node_update" >> gpurun_out/pmc_sample_$tag.txt
  echo "pass $i done"
done
cat gpurun_out/pmc_sample_$tag.txt
