#!/bin/bash
cd "$GRAFT_REPO_ROOT"
RT_CUSHARE=0 tools/run_guarded.sh gpurun_out/pmc_c2_nocus.log 400 tools/pmc.sh c2nocus --workload C2 --spp 16 || exit 1
RT_CUSHARE=1 tools/run_guarded.sh gpurun_out/pmc_c2_cus.log 400 tools/pmc.sh c2cus --workload C2 --spp 16 || exit 1
