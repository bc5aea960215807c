#!/bin/bash
# HISTORICAL: the variant this script measured was dropped and its switch is no longer in the code (results: profiles/r03_*.txt).
cd "$GRAFT_REPO_ROOT"
tools/exp.sh "RT_LIGHT_PERM=210 :: --workload C2 --no-pmc --steps 5" "RT_LIGHT_PERM=120 :: --workload C2 --no-pmc --steps 5" "RT_LIGHT_PERM=201 :: --workload C2 --no-pmc --steps 5" "RT_LIGHT_PERM=021 :: --workload C2 --no-pmc --steps 5" "RT_LIGHT_PERM=102 :: --workload C2 --no-pmc --steps 5" "RT_LIGHT_PERM=012 :: --workload C2 --no-pmc --steps 5" > gpurun_out/ab_lp.log 2>&1
cat gpurun_out/ab_lp.log
