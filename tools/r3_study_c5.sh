#!/bin/bash
# round 3: what binds the 1 M-triangle frame?  gather microbenchmark + PMC sets of the binary and the wide tree
cd "$GRAFT_REPO_ROOT"
tools/run_guarded.sh gpurun_out/gather.json 300 ray-tracing-engine_amd/bin/gather_bench || exit 1
RT_BVH_WIDE=0 tools/run_guarded.sh gpurun_out/pmc_c5_bin.log 500 tools/pmc.sh c5bin --workload C5 --spp 32 || exit 1
RT_BVH_WIDE=1 tools/run_guarded.sh gpurun_out/pmc_c5_wide.log 500 tools/pmc.sh c5wide --workload C5 --spp 32 || exit 1
