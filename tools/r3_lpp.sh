#!/bin/bash
cd "$GRAFT_REPO_ROOT"
tools/run_guarded.sh gpurun_out/t_all.log 1100 python3 -m pytest tests -q -m gpu || exit 1
tools/exp.sh ":: --workload C2 --no-pmc --steps 10" ":: --workload C4 --no-pmc --steps 2" ":: --workload C5 --no-pmc" ":: --workload C5x8 --no-pmc" ":: --workload C1 --no-pmc" ":: --workload C3 --no-pmc" > gpurun_out/ab_lpp3.log 2>&1
cat gpurun_out/ab_lpp3.log
