#!/bin/bash
cd "$GRAFT_REPO_ROOT"
tools/run_guarded.sh gpurun_out/pmc_c3.log 500 tools/pmc.sh c3 --workload C3 || exit 1
tools/run_guarded.sh gpurun_out/pmcta_c3.log 500 tools/pmc_ta.sh c3 --workload C3 || exit 1
tools/run_guarded.sh gpurun_out/pmcis_c3.log 500 tools/pmc_issue.sh c3 --workload C3 || exit 1
python3 bench.py --workload C3 --no-pmc --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | grep '^{' > gpurun_out/bench_c3.json
