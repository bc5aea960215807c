#!/bin/bash
cd "$GRAFT_REPO_ROOT"
tools/run_guarded.sh gpurun_out/t_bvhbuild.log 900 python3 -m pytest tests/test_gpu_bvhbuild.py -x -q -m gpu
