#!/bin/bash
# PMC passes over one bench configuration (own runs, --kernel-trace only; see gpurun rules).
# usage: tools/pmc.sh <tag> <bench args...>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
out=gpurun_out/pmc_$tag
mkdir -p $out
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
  "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_BRANCH" \
  "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" \
  "GRBM_GUI_ACTIVE GRBM_COUNT" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 bench.py --no-cpu-baseline --no-pmc --steps 1 --warmup 0 "$@" > $out/p$i.log 2>&1 || echo "pass $i failed" >> $out/fail.log
done
python3 - "$out" <<'PY'
import sys,glob,csv,collections
out=sys.argv[1]
agg=collections.OrderedDict()
for f in sorted(glob.glob(out+"/p*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "k_render_persist<" in n or "k_render_persist5<" in n:
            if n.split("<")[1].split(">")[0].split(", ")[0] == "true": continue  # the STATS build
        elif "k_render<" in n:
            if n.split("k_render<")[1].split(">")[0].split(", ")[3] == "true": continue
        else: continue
        key=(n.split("(")[0][-40:], r["Counter_Name"])
        agg.setdefault(key,[]).append(float(r["Counter_Value"]))
with open(out+"/summary.txt","w") as fo:
    for (k,c),v in agg.items():
        fo.write("%-42s %-32s n=%d last=%.6g\n"%(k,c,len(v),v[-1]))
print(open(out+"/summary.txt").read())
PY
