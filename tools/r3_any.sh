#!/bin/bash
cd "$GRAFT_REPO_ROOT"
tools/exp.sh "RT_BVH_CHILD_ORDER=5 :: --workload C5 --no-pmc" "RT_BVH_CHILD_ORDER=5 :: --workload C5x8 --no-pmc" "RT_BVH_CHILD_ORDER=5 :: --workload C2 --no-pmc" "RT_BVH_CHILD_ORDER=5 :: --workload C4 --no-pmc --steps 2" > gpurun_out/ab_any4.log 2>&1
cat gpurun_out/ab_any4.log
