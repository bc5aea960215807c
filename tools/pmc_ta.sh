#!/bin/bash
# TA / TCP busy and stall counters of the timed render kernel (what does the vector-memory path do?)
# usage: tools/pmc_ta.sh <tag> <bench args...>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
out=gpurun_out/pmcta_$tag; mkdir -p $out
i=0
# (TA and TCP have 2 counter slots per pass; a set that does not fit makes rocprofv3 abort and
# then hang in its finaliser, so every pass runs under a short timeout)
for set in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum" "TA_FLAT_READ_WAVEFRONTS_sum" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  i=$((i+1))
  echo "pass $i: $set"
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 bench.py --no-cpu-baseline --no-pmc --steps 1 --warmup 0 "$@" > $out/p$i.log 2>&1 || echo "pass $i failed" >> $out/fail.log
done
python3 - "$out" <<'PY'
import sys,glob,csv,collections
out=sys.argv[1]; agg=collections.OrderedDict()
for f in sorted(glob.glob(out+"/p*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        # the timed (STATS = false) render kernel: persistent pooled, or one-wave-per-workgroup (photon / brute / sequential)
        timed = "k_render_persist<false" in n or "k_render_persist5<false" in n or ("k_render<" in n and n.split("k_render<")[1].split(">")[0].split(", ")[3] == "false")
        if not timed: continue
        agg.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
for c,v in agg.items(): print("%-44s %.6g"%(c,v[-1]))
PY
