#!/bin/bash
# HISTORICAL: the variant this script measured was dropped and its switch is no longer in the code (results: profiles/r03_*.txt).
cd "$GRAFT_REPO_ROOT"
tools/exp.sh "RT_PF_TRIS=0 :: --workload C2 --no-pmc" "RT_PF_TRIS=1 :: --workload C2 --no-pmc" "RT_PF_TRIS=0 :: --workload C4 --no-pmc --steps 2" "RT_PF_TRIS=1 :: --workload C4 --no-pmc --steps 2" "RT_PF_TRIS=1 :: --workload C5 --no-pmc" > gpurun_out/ab_pf.log 2>&1
cat gpurun_out/ab_pf.log
