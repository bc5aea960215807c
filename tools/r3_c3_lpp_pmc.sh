#!/bin/bash
# C3: 8 against 16 samples of a pixel per wave (one PMC pass each)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lpp in 8 16; do
  d=gpurun_out/pmc_c3lpp_$lpp; rm -rf $d
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $d -- python3 bench.py --pmc-child --workload C3 --lpp $lpp > $d.log 2>&1
  echo "== lpp $lpp"
  python3 - "$d" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_render<false, true" in n: acc[n.split("(")[0][-40:]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in acc.items():
    print(k, {x: "%.4g" % v for x, v in c.items()}, "lanes %.3f" % (c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_INSTS_VALU"]) if c.get("SQ_INSTS_VALU") else 0))
PY
done
