#!/bin/bash
# HBM-side bytes of one random 32-/64-byte record (gather microbenchmark under a FETCH_SIZE pass)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/fetchcal; mkdir -p $out
for cfg in "1048576 32" "1048576 64" "16384 32"; do
  set -- $cfg
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/t$1_r$2 -- ray-tracing-engine_amd/bin/gather_bench one $1 $2 > $out/t$1_r$2.log 2>&1
done
python3 - $out <<'PY'
import sys,glob,csv,json
out=sys.argv[1]
for d in sorted(glob.glob(out+"/t*_r*/")):
    log=json.loads([l for l in open(d.rstrip("/")+".log") if l.startswith("{")][-1])
    vals=[]
    for f in glob.glob(d+"**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_chase" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE": vals.append(float(r["Counter_Value"]))
    # two dispatches: warm-up (8 steps) and the timed one (1024 steps): the larger value is the timed one
    kib=max(vals) if vals else float("nan")
    print(json.dumps({"table_kb":log["table_kb"],"rec_bytes":log["rec_bytes"],"records":log["records"],"grec_per_s":log["grec_per_s"],
                      "FETCH_SIZE_KiB":kib,"hbm_side_bytes_per_record_x2":2*kib*1024/log["records"],"raw_bytes_per_record":kib*1024/log["records"]}))
PY
