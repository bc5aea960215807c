#!/bin/bash
# (RT_KNN_WAIT belonged to the insert-batching variant of knn_query, measured and removed: profiles/r03_c3_experiments.txt)
cd "$GRAFT_REPO_ROOT"
tools/run_guarded.sh gpurun_out/t_knn.log 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_big.py tests/test_gpu_kdbuild.py -x -q -m gpu -k "knn or photon or config3 or kd" || exit 1
tools/exp.sh ":: --workload C3 --no-pmc --steps 5" > gpurun_out/ab_c3.log 2>&1
cat gpurun_out/ab_c3.log
