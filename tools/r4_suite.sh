#!/bin/bash
# round 4: the whole GPU suite, then one bench line per workload (no PMC passes)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r4_suite.log 2>&1; echo "gpu suite rc $?"; tail -5 gpurun_out/r4_suite.log
for wl in ${WL:-C1 C2 C3 C4 C5 C5x8}; do
  timeout -k 10 400 bash tools/ab.sh "RT_X=0" $wl 2>&1 | tee -a gpurun_out/r4_suite_bench.log
done
