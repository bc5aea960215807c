#!/bin/bash
# HISTORICAL: RT_BVH_EMULATE_DEVICE (the host builder emulating the device top) was an experiment switch and is no longer in the code (results: profiles/r03_bvh_emulation_morton_top_exact_bottom.txt).
cd "$GRAFT_REPO_ROOT"
{
for wl in C2 C4 C5; do
  st=3; [ $wl = C4 ] && st=2
  tools/exp.sh " :: --workload $wl --no-pmc --steps $st" "RT_BVH_EMULATE_DEVICE=64 :: --workload $wl --no-pmc --steps $st" "RT_BVH_EMULATE_DEVICE=256 :: --workload $wl --no-pmc --steps $st" "RT_BVH_EMULATE_DEVICE=1024 :: --workload $wl --no-pmc --steps $st" "RT_BVH_EMULATE_DEVICE=4096 :: --workload $wl --no-pmc --steps $st" "RT_BVH_GPU=1 :: --workload $wl --no-pmc --steps $st"
done
} > gpurun_out/ab_emu.log 2>&1
cat gpurun_out/ab_emu.log
