#!/bin/bash
# Instruction-issue side of the timed render kernel: instruction fetch / I-cache, scalar cycles, FP64 and
# transcendental shares (why is a wave not issuing?).  usage: tools/pmc_issue.sh <tag> <bench args...>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
out=gpurun_out/pmcis_$tag; mkdir -p $out
i=0
for set in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_INSTS_VSKIPPED SQ_CYCLES SQ_BUSY_CU_CYCLES" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQC_TC_STALL" \
           "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_FMA_F16 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  echo "pass $i: $set"
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 bench.py --no-cpu-baseline --no-pmc --steps 1 --warmup 0 "$@" > $out/p$i.log 2>&1 || echo "pass $i failed" >> $out/fail.log
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
