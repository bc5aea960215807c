#!/bin/bash
# round 4: the whole GPU suite re-run with a tree form forced for every context of every test process:
#   RT_NODES=q8 (one-request records), RT_BVH_GPU=2 (hybrid builder on all scenes), RT_BVH_GPU=1 (device builder)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for env in "RT_NODES=q8" "RT_BVH_GPU=2" "RT_BVH_GPU=1"; do
  tag=$(echo $env | tr '=' '_')
  env $env timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r4_suite_$tag.log 2>&1; echo "[$env] rc $?: $(tail -1 gpurun_out/r4_suite_$tag.log)"
  grep "^FAILED" gpurun_out/r4_suite_$tag.log | head -8
done
