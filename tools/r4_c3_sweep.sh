#!/bin/bash
# round 4: C3 (photon k-NN over the explicit topology): samples of a pixel per wave, then the counter passes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for lpp in 4 8 16 32 64; do
  r=$(python3 bench.py --no-cpu-baseline --no-pmc --steps 5 --warmup 2 --workload C3 --lpp $lpp 2>&1 | tail -1)
  echo "C3 lpp=$lpp $(echo "$r" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.0f Mrays/s  %.2f ms"%(d["value"], d["ms_per_step"]))')" | tee -a gpurun_out/r4_c3_lpp.log
done
bash tools/pmc.sh r04_c3 --workload C3 > /dev/null 2>&1
cat gpurun_out/pmc_r04_c3/summary.txt | awk '{print $3, $4, $6}'
