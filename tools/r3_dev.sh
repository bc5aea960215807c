#!/bin/bash
# host builder: the size axis in the swept ranges only / the binned ranges only / with a bias
cd "$GRAFT_REPO_ROOT"
{
timeout -k 10 1100 tools/exp.sh "RT_BVH_SIZEAXIS=1 :: --workload C5 --no-pmc --steps 3" "RT_BVH_SIZEAXIS=2 :: --workload C5 --no-pmc --steps 3" "RT_BVH_SIZEBIAS=1.2 :: --workload C5 --no-pmc --steps 3" "RT_BVH_SIZEBIAS=1.5 :: --workload C5 --no-pmc --steps 3" \
 "RT_BVH_SIZEAXIS=1 :: --workload C4 --no-pmc --steps 2" "RT_BVH_SIZEAXIS=2 :: --workload C4 --no-pmc --steps 2" "RT_BVH_SIZEBIAS=1.2 :: --workload C4 --no-pmc --steps 2" "RT_BVH_SIZEBIAS=1.5 :: --workload C4 --no-pmc --steps 2" \
 "RT_BVH_SIZEBIAS=1.2 :: --workload C2 --no-pmc --steps 10" "RT_BVH_SIZEBIAS=1.5 :: --workload C2 --no-pmc --steps 10" "RT_BVH_SIZEBIAS=0.8 :: --workload C2 --no-pmc --steps 10" "RT_BVH_SIZEBIAS=1.2 :: --workload C5x8 --no-pmc --steps 3" "RT_BVH_SIZEAXIS=2 :: --workload C5x8 --no-pmc --steps 3"
} > gpurun_out/ab_sizeaxis2.log 2>&1
cat gpurun_out/ab_sizeaxis2.log
