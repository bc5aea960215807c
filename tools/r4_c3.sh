#!/bin/bash
# round 4: photon k-NN walk over the explicit kd topology — parity tests, then the C3 frame
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kdbuild.py tests/test_gpu_parity_big.py -x -q -k "knn or photon or config3 or kd" > gpurun_out/r4_c3_tests.log 2>&1; echo "knn tests rc $?"; tail -3 gpurun_out/r4_c3_tests.log
timeout -k 10 400 bash tools/ab.sh "RT_X=0" C3 2>&1 | tee -a gpurun_out/r4_c3_ab.log
