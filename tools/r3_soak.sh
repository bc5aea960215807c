#!/bin/bash
cd "$GRAFT_REPO_ROOT"
tools/run_guarded.sh gpurun_out/soak.log 900 python3 tools/soak.py || exit 1
tools/run_guarded.sh gpurun_out/fuzz.log 900 python3 tools/fuzz.py 31 || exit 1
RT_BVH_GPU=1 tools/run_guarded.sh gpurun_out/fuzz_gpubvh.log 900 python3 tools/fuzz.py 32 || exit 1
RT_FUZZ_TUNE=400 tools/run_guarded.sh gpurun_out/fuzz_tuned.log 900 python3 tools/fuzz.py 33 || exit 1
tail -n 12 gpurun_out/fuzz_tuned.log
tail -n 12 gpurun_out/soak.log gpurun_out/fuzz.log gpurun_out/fuzz_gpubvh.log
