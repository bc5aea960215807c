#!/bin/bash
# A round's measurements on the GPU box (one gpurun call): bench lines of every workload (each
# with its own in-run rocprofv3 PMC passes), rocprofv3 kernel stats of the default bench
# command, the wide PMC sets of C2 / C3 / C5, section clocks of the pooled kernel, the builders'
# report, the gather microbenchmark (the vector L1's divergent request ceiling).  Output:
# gpurun_out/round/ (copy the summaries to profiles/ with `tools/collect_profiles.py r04`).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/round; mkdir -p $out
python3 bench.py --steps 20 --warmup 5 --cpu-baseline-full > $out/bench_C2.log 2>&1; grep '^{' $out/bench_C2.log > $out/bench_C2.json
python3 bench.py --tune-probes 1500 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_C2_tuned.log 2>&1; grep '^{' $out/bench_C2_tuned.log > $out/bench_C2_tuned.json
echo "C2 done"
for w in C1 C3 C4 C5 C5x8; do
  python3 bench.py --workload $w --no-cpu-baseline --steps 5 --warmup 2 > $out/bench_$w.log 2>&1; grep '^{' $out/bench_$w.log > $out/bench_$w.json
  echo "$w done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --no-cpu-baseline --no-pmc --steps 20 --warmup 5 > $out/prof.log 2>&1
find $out/prof -name '*kernel_stats.csv' -exec cp {} $out/c2_kernel_stats.csv \;
grep '^{' $out/prof.log > $out/bench_C2_under_rocprof.json
echo "kernel stats done"
tools/pmc.sh c2 --spp 16 > /dev/null 2>&1; cp gpurun_out/pmc_c2/summary.txt $out/c2_pmc_spp16.txt
tools/pmc.sh c5 --workload C5 --spp 32 > /dev/null 2>&1; cp gpurun_out/pmc_c5/summary.txt $out/c5_pmc_spp32.txt
tools/pmc.sh c5x8 --workload C5x8 --spp 8 > /dev/null 2>&1; cp gpurun_out/pmc_c5x8/summary.txt $out/c5x8_pmc_spp8.txt
tools/pmc.sh c3 --workload C3 > /dev/null 2>&1; cp gpurun_out/pmc_c3/summary.txt $out/c3_pmc.txt
tools/pmc_ta.sh c2 --workload C2 --spp 16 > $out/c2_pmc_ta.txt 2>&1
tools/pmc_ta.sh c5 --workload C5 --spp 32 > $out/c5_pmc_ta.txt 2>&1
tools/pmc_issue.sh c2 --workload C2 --spp 16 > $out/c2_pmc_issue.txt 2>&1
tools/pmc_issue.sh c3 --workload C3 > $out/c3_pmc_issue.txt 2>&1
tools/pmc_ta.sh c3 --workload C3 > $out/c3_pmc_ta.txt 2>&1
echo "pmc done"
tools/phase_timing.sh C2 C4 C5 > $out/phase.log 2>&1; for w in C2 C4 C5; do [ -s gpurun_out/phase_$w.json ] && cp gpurun_out/phase_$w.json $out/; done
echo "phase done"
python3 -m pytest tests/test_gpu_bvhbuild.py tests/test_gpu_kdbuild.py -m gpu -q -s -k "report or built_on_the_device" > $out/builders.txt 2>&1
python3 tools/builders.py 2>/dev/null | grep -E "^(lowres|hires|stress|stress8) " >> $out/builders.txt
tools/kernel_resources.sh > $out/kernel_resources.txt 2>&1
echo "all done"
