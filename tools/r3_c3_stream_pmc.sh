#!/bin/bash
# photon query stream against the fused photon kernel: instructions, lane utilisation, waits (one PMC pass each)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for mode in stream fused; do
  d=gpurun_out/pmc_c3s_$mode; rm -rf $d
  if [ $mode = fused ]; then export RT_PHOTON_STREAM=0; else unset RT_PHOTON_STREAM; fi
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $d -- python3 bench.py --pmc-child --workload C3 > $d.log 2>&1
  echo "== $mode"
  python3 - "$d" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "k_knn_stream" if "k_knn_stream" in n else "k_render<photon>" if "k_render<false, true" in n else None
        if k: acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in acc.items():
    print(k, {x: "%.4g" % v for x, v in c.items()}, "lanes %.3f" % (c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_INSTS_VALU"]) if c.get("SQ_INSTS_VALU") else 0))
PY
done
