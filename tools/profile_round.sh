#!/bin/bash
# Round measurements on the GPU box: bench lines for C1-C5, rocprofv3 kernel stats of the
# default bench command, HBM traffic (PMC) of C2 and C5.  Output: gpurun_out/round/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/round; mkdir -p $out
python3 bench.py > $out/bench_C2.log 2>&1; grep '^{' $out/bench_C2.log > $out/bench_C2.json
for w in C1 C3 C4 C5; do
  python3 bench.py --workload $w --no-cpu-baseline > $out/bench_$w.log 2>&1; grep '^{' $out/bench_$w.log > $out/bench_$w.json
done
echo bench done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --no-cpu-baseline > $out/prof.log 2>&1
find $out/prof -name '*kernel_stats.csv' -exec cp {} $out/c2_kernel_stats.csv \;
echo prof done
tools/traffic.sh C2 > $out/traffic_C2.log 2>&1
tools/traffic.sh C5 > $out/traffic_C5.log 2>&1
cp gpurun_out/traffic_C2.json gpurun_out/traffic_C5.json $out/ 2>/dev/null
echo traffic done
