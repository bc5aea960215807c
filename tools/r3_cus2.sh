#!/bin/bash
cd "$GRAFT_REPO_ROOT"
tools/run_guarded.sh gpurun_out/t_cus.log 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_big.py -x -q -m gpu -k "frame or policies or schedule or pool" || exit 1
RT_CUSHARE=1 tools/run_guarded.sh gpurun_out/phase_cus.log 500 tools/phase_timing.sh C2 || exit 1
cp gpurun_out/phase_C2.json gpurun_out/phase_C2_cus.json
{
tools/exp.sh "RT_CUSHARE=0 :: --workload C2 --no-pmc" "RT_CUSHARE=1 :: --workload C2 --no-pmc" "RT_CUSHARE=1 RT_REFILLT=16 :: --workload C2 --no-pmc" "RT_CUSHARE=1 RT_REFILLT=32 :: --workload C2 --no-pmc" \
  "RT_CUSHARE=1 :: --workload C4 --no-pmc --steps 2" "RT_CUSHARE=1 :: --workload C5 --no-pmc"
} > gpurun_out/ab_cus.log 2>&1
cat gpurun_out/ab_cus.log
