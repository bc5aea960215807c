#!/bin/bash
# measured-cost BVH tuning: C4 with probes of growing size; C2 with a short budget; C5 (refused? no: capped)
cd "$GRAFT_REPO_ROOT"
{
for cfg in "hires 2048 64 256 1 60" "hires 2048 64 512 1 120" "lowres 1024 128 128 1 0.4" "lowres 1024 128 128 1 1.0" "cubes 256 8 64 2 1"; do
  set -- $cfg
  echo "== $1 ${2}x$2 x $3 spp: probe $4 x $5 spp, budget $6 s"
  RT_BVH_VERBOSE=1 timeout -k 10 400 python3 tools/tune_try.py --scene $1 --size $2 --spp $3 --probe $4 --probe-spp $5 --budget $6 2>&1 | grep "before\|after\|tune:\|referee"
done
} > gpurun_out/tune_ab2.log 2>&1
cat gpurun_out/tune_ab2.log
