#!/bin/bash
# round 4: the one-request (Q8) node records against the 32-byte ones — parity tests, per-ray counters, frames
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_q8.py -x -q > gpurun_out/r4_q8_tests.log 2>&1; echo "q8 tests rc $?"; tail -3 gpurun_out/r4_q8_tests.log
timeout -k 10 300 python tools/q8_stats.py stress hires 2>&1 | tee gpurun_out/r4_q8_stats.log
for wl in ${WL:-C5 C5x8}; do
  for fmt in f16 q8; do
    timeout -k 10 400 bash tools/ab.sh "RT_NODES=$fmt" $wl 2>&1 | tee -a gpurun_out/r4_q8_ab.log
  done
done
