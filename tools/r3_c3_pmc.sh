#!/bin/bash
cd "$GRAFT_REPO_ROOT"
tools/pmc_ta.sh c3 --workload C3 > gpurun_out/c3_pmc_ta.txt 2>&1
tools/pmc_issue.sh c3 --workload C3 > gpurun_out/c3_pmc_issue.txt 2>&1
tail -n 22 gpurun_out/c3_pmc_ta.txt; tail -n 45 gpurun_out/c3_pmc_issue.txt
