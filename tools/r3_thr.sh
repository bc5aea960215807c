#!/bin/bash
# pool thresholds: HEAD defaults on every workload, and the old third-of-live rule beside them
cd "$GRAFT_REPO_ROOT"
{
timeout -k 10 1000 tools/exp.sh ":: --workload C2 --no-pmc --steps 20" "RT_LEAFMUL=21 :: --workload C2 --no-pmc --steps 20" "RT_LEAFMUL=24 :: --workload C2 --no-pmc --steps 20" ":: --workload C4 --no-pmc --steps 3" "RT_LEAFMUL=24 :: --workload C4 --no-pmc --steps 3" ":: --workload C1 --no-pmc --steps 20" ":: --workload C3 --no-pmc --steps 5" \
  ":: --workload C5 --no-pmc --steps 3" ":: --workload C5x8 --no-pmc --steps 3" "RT_REFILLT=40 :: --workload C5x8 --no-pmc --steps 3" "RT_REFILLT=24 :: --workload C5x8 --no-pmc --steps 3"
} > gpurun_out/ab_thr4.log 2>&1
cat gpurun_out/ab_thr4.log
