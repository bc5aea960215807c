#!/bin/bash
# round 3: wide-vs-binary A/B (tests first, then bench lines without PMC)
cd "$GRAFT_REPO_ROOT"
tools/run_guarded.sh gpurun_out/t_wide.log 900 python3 -m pytest tests/test_gpu_wide.py tests/test_gpu_multi.py -x -q -m gpu || exit 1
{
tools/exp.sh "RT_BVH_WIDE=0 :: --workload C5 --no-pmc" "RT_BVH_WIDE=1 :: --workload C5 --no-pmc" \
  "RT_BVH_WIDE=1 RT_BVH_WIDE_BUDGET=25 :: --workload C5 --no-pmc" "RT_BVH_WIDE=1 RT_BVH_WIDE_BUDGET=30 :: --workload C5 --no-pmc" \
  "RT_BVH_WIDE=0 :: --workload C5x8 --no-pmc" "RT_BVH_WIDE=1 :: --workload C5x8 --no-pmc" "RT_BVH_WIDE=1 RT_BVH_WIDE_BUDGET=30 :: --workload C5x8 --no-pmc" \
  "RT_BVH_WIDE=0 :: --workload C4 --no-pmc --steps 2" "RT_BVH_WIDE=1 :: --workload C4 --no-pmc --steps 2" "RT_BVH_WIDE=1 RT_BVH_WIDE_BUDGET=30 :: --workload C4 --no-pmc --steps 2" \
  "RT_BVH_WIDE=0 :: --workload C2 --no-pmc" "RT_BVH_WIDE=1 :: --workload C2 --no-pmc" "RT_BVH_WIDE=1 RT_BVH_WIDE_BUDGET=30 :: --workload C2 --no-pmc"
} > gpurun_out/ab_wide.log 2>&1
cat gpurun_out/ab_wide.log
