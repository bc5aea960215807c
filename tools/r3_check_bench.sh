#!/bin/bash
cd "$GRAFT_REPO_ROOT"
tools/run_guarded.sh gpurun_out/t_app.log 1100 python3 -m pytest tests/test_gpu_app.py tests/test_gpu_multi.py -x -q -m gpu || exit 1
tools/run_guarded.sh gpurun_out/bench_default.log 600 python3 bench.py --steps 10 --warmup 3 --cpu-baseline-full || exit 1
grep '^{' gpurun_out/bench_default.log > gpurun_out/bench_default.json
