#!/bin/bash
# Diagnostic: where does a wave of the pooled k_render spend its shader clocks?
# Builds lib/librt_amd_timing.so (-DRT_PHASE_TIMING: s_memtime stamps at wave-uniform
# points, +~10 % run time) and renders one counted frame per workload.
# usage (GPU box): tools/phase_timing.sh [C2 C4 C5]   -> gpurun_out/phase_<wl>.json
set -e
cd "$(dirname "$0")/.."
PKG=ray-tracing-engine_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -DRT_PHASE_TIMING \
  -Wno-unused-function -Iinclude -I$PKG/csrc -I$PKG/host -pthread -shared -o $PKG/lib/librt_amd_timing.so \
  $PKG/csrc/rt_kernels.hip $PKG/csrc/kd_build.hip $PKG/csrc/bvh_gpu.hip $PKG/csrc/rt_api.cpp $PKG/csrc/bvh_build.cpp
mkdir -p gpurun_out
for wl in ${@:-C2}; do
  RT_AMD_LIB=$PWD/$PKG/lib/librt_amd_timing.so RT_PHASE_DUMP=1 python3 tools/phase_report.py $wl > gpurun_out/phase_$wl.json
  cat gpurun_out/phase_$wl.json
done
