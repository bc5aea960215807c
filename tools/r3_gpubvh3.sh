#!/bin/bash
cd "$GRAFT_REPO_ROOT"
{
for wl in C2 C4 C5; do
  st=3; [ $wl = C4 ] && st=2
  tools/exp.sh "RT_BVH_GPU=1 RT_BVH_GPU_BINS=32 :: --workload $wl --no-pmc --steps $st" "RT_BVH_GPU=1 RT_BVH_GPU_BINS=64 :: --workload $wl --no-pmc --steps $st" "RT_BVH_GPU=1 RT_BVH_GPU_TOP=morton :: --workload $wl --no-pmc --steps $st"
done
} > gpurun_out/ab_gpubvh2.log 2>&1
cat gpurun_out/ab_gpubvh2.log
