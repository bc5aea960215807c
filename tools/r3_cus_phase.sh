#!/bin/bash
cd "$GRAFT_REPO_ROOT"
RT_CUSHARE=0 tools/run_guarded.sh gpurun_out/phase_nocus.log 500 tools/phase_timing.sh C2 || exit 1
cp gpurun_out/phase_C2.json gpurun_out/phase_C2_nocus.json
RT_CUSHARE=1 tools/run_guarded.sh gpurun_out/phase_cus.log 500 tools/phase_timing.sh C2 || exit 1
cp gpurun_out/phase_C2.json gpurun_out/phase_C2_cus.json
