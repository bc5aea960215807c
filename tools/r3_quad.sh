#!/bin/bash
# HISTORICAL: the variant this script measured was dropped and its switch is no longer in the code (results: profiles/r03_*.txt).
cd "$GRAFT_REPO_ROOT"
tools/run_guarded.sh gpurun_out/t_quad.log 1100 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_big.py -q -m gpu -x -k "stress or big or full_size or policies or pool" || exit 1
tools/exp.sh "RT_QUAD=0 :: --workload C5 --no-pmc" "RT_QUAD=1 :: --workload C5 --no-pmc" "RT_QUAD=0 :: --workload C5x8 --no-pmc" "RT_QUAD=1 :: --workload C5x8 --no-pmc" > gpurun_out/ab_quad.log 2>&1
cat gpurun_out/ab_quad.log
