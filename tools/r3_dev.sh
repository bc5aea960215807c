#!/bin/bash
# host builder: an origin term in the split cost (rays that start inside the range)
cd "$GRAFT_REPO_ROOT"
{
timeout -k 10 1100 tools/exp.sh "RT_BVH_ORIGIN=0.05 :: --workload C2 --no-pmc --steps 10" "RT_BVH_ORIGIN=0.15 :: --workload C2 --no-pmc --steps 10" "RT_BVH_ORIGIN=0.4 :: --workload C2 --no-pmc --steps 10" "RT_BVH_ORIGIN=1 :: --workload C2 --no-pmc --steps 10" \
  "RT_BVH_ORIGIN=0.05 :: --workload C4 --no-pmc --steps 2" "RT_BVH_ORIGIN=0.15 :: --workload C4 --no-pmc --steps 2" "RT_BVH_ORIGIN=0.4 :: --workload C4 --no-pmc --steps 2" "RT_BVH_ORIGIN=1 :: --workload C4 --no-pmc --steps 2" \
  "RT_BVH_ORIGIN=0.15 :: --workload C5 --no-pmc --steps 3" "RT_BVH_ORIGIN=0.4 :: --workload C5 --no-pmc --steps 3"
} > gpurun_out/ab_origin.log 2>&1
cat gpurun_out/ab_origin.log
