#!/bin/bash
# HISTORICAL: the variant this script measured was dropped and its switch is no longer in the code (results: profiles/r03_*.txt).
cd "$GRAFT_REPO_ROOT"
tools/run_guarded.sh gpurun_out/t_lock.log 600 env RT_LOCK_LIGHTS=2 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "frame" || exit 1
tools/exp.sh "RT_LOCK_LIGHTS=0 :: --workload C2 --no-pmc --steps 5" "RT_LOCK_LIGHTS=1 :: --workload C2 --no-pmc --steps 5" "RT_LOCK_LIGHTS=2 :: --workload C2 --no-pmc --steps 5" "RT_LOCK_LIGHTS=3 :: --workload C2 --no-pmc --steps 5" "RT_LOCK_LIGHTS=2 :: --workload C4 --no-pmc --steps 2" > gpurun_out/ab_lock.log 2>&1
cat gpurun_out/ab_lock.log
