#!/bin/bash
# host builder: subtree reinsertion passes before the rotations
cd "$GRAFT_REPO_ROOT"
{
timeout -k 10 1100 tools/exp.sh ":: --workload C2 --no-pmc --steps 10" "RT_BVH_REINSERT=1 :: --workload C2 --no-pmc --steps 10" "RT_BVH_REINSERT=4 :: --workload C2 --no-pmc --steps 10" "RT_BVH_REINSERT=4 RT_BVH_ROT=0 :: --workload C2 --no-pmc --steps 10" \
  ":: --workload C4 --no-pmc --steps 2" "RT_BVH_REINSERT=1 :: --workload C4 --no-pmc --steps 2" "RT_BVH_REINSERT=4 :: --workload C4 --no-pmc --steps 2" \
  "RT_BVH_REINSERT=4 :: --workload C5 --no-pmc --steps 3" "RT_BVH_REINSERT=4 :: --workload C1 --no-pmc --steps 20" "RT_BVH_REINSERT=4 :: --workload C3 --no-pmc --steps 5"
} > gpurun_out/ab_reinsert.log 2>&1
cat gpurun_out/ab_reinsert.log
