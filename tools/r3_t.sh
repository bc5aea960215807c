#!/bin/bash
cd "$GRAFT_REPO_ROOT"
tools/run_guarded.sh gpurun_out/t_all.log 1100 python3 -m pytest tests -q -m gpu || exit 1
RT_BVH_GPU=1 tools/run_guarded.sh gpurun_out/t_all_gpubvh.log 1100 python3 -m pytest tests -q -m gpu || exit 1
