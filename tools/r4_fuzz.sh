#!/bin/bash
# round 4: the large-scale exactness runs (tools/fuzz.py: millions of random rays BVH vs exhaustive loop, frames BVH vs brute;
# tools/soak.py: full-size frames repeated) on every tree form round 4 added: Q8 records, hybrid-built trees, both together
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for env in ${FUZZ_FORMS:-"RT_X=0" "RT_NODES=q8" "RT_BVH_GPU=1" "RT_BVH_GPU=2" "RT_NODES=q8#RT_BVH_GPU=2" "RT_NODES=q8#RT_BVH_GPU=1"}; do
  env=${env//#/ }
  echo "== fuzz [$env]" | tee -a gpurun_out/r4_fuzz.log
  env $env timeout -k 10 500 python tools/fuzz.py 23 2>&1 | tee -a gpurun_out/r4_fuzz.log | tail -1
done
for env in "RT_X=0" "RT_NODES=q8"; do
  echo "== soak [$env]" | tee -a gpurun_out/r4_fuzz.log
  env $env timeout -k 10 500 python tools/soak.py 2>&1 | tee -a gpurun_out/r4_fuzz.log | grep -c True
done
