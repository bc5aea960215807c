#!/bin/bash
cd "$GRAFT_REPO_ROOT"
tools/run_guarded.sh gpurun_out/t_cus.log 1100 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_big.py -x -q -m gpu || exit 1
{
tools/exp.sh "RT_CUSHARE=0 :: --workload C2 --no-pmc" "RT_CUSHARE=1 :: --workload C2 --no-pmc" \
  "RT_CUSHARE=0 :: --workload C4 --no-pmc --steps 2" "RT_CUSHARE=1 :: --workload C4 --no-pmc --steps 2" \
  "RT_CUSHARE=0 :: --workload C5 --no-pmc" "RT_CUSHARE=1 :: --workload C5 --no-pmc" \
  "RT_CUSHARE=0 :: --workload C1 --no-pmc" "RT_CUSHARE=1 :: --workload C1 --no-pmc"
} > gpurun_out/ab_cus.log 2>&1
cat gpurun_out/ab_cus.log
