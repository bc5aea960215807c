#!/bin/bash
# round 4: host / device / hybrid BVH builders — parity on the hybrid tree, then time and quality side by side
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
RT_BVH_GPU=2 timeout -k 10 900 python -m pytest tests/test_gpu_parity_big.py tests/test_gpu_q8.py -x -q -k "trace_big or frames_vs_oracle or goldens or q8_hits or q8_frames" > gpurun_out/r4_hybrid_tests.log 2>&1; echo "hybrid-tree tests rc $?"; tail -2 gpurun_out/r4_hybrid_tests.log
timeout -k 10 600 python tools/builders.py ${SCENES:-lowres hires stress stress8} 2>&1 | tee gpurun_out/r4_builders.log
for wl in ${WL:-C5 C5x8 C4}; do
  for b in 3 2; do
    timeout -k 10 400 bash tools/ab.sh "RT_BVH_GPU=$b" $wl 2>&1 | tee -a gpurun_out/r4_builders_ab.log
  done
done
