#!/bin/bash
# pool thresholds on the headline scene after the leaf got cheaper (short reciprocals)
cd "$GRAFT_REPO_ROOT"
{
timeout -k 10 1000 tools/exp.sh ":: --workload C2 --no-pmc --steps 20" "RT_LEAFT=8 :: --workload C2 --no-pmc --steps 20" "RT_LEAFT=16 :: --workload C2 --no-pmc --steps 20" "RT_LEAFMUL=16 :: --workload C2 --no-pmc --steps 20" "RT_LEAFMUL=28 :: --workload C2 --no-pmc --steps 20" "RT_REFILLT=16 :: --workload C2 --no-pmc --steps 20" "RT_REFILLT=32 :: --workload C2 --no-pmc --steps 20" "RT_STEALT=4 :: --workload C2 --no-pmc --steps 20" "RT_STEALT=12 :: --workload C2 --no-pmc --steps 20"
} > gpurun_out/ab_thr5.log 2>&1
cat gpurun_out/ab_thr5.log
