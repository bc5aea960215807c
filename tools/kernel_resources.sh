#!/bin/bash
# Registers, scratch and LDS of every kernel instance (compiler remarks; runs without a GPU).
# usage: tools/kernel_resources.sh [filter-regex]  -> table on stdout
cd "$(dirname "$0")/../ray-tracing-engine_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -I../include -Icsrc -Ihost \
  -c csrc/rt_kernels.hip -o /tmp/rk_res.o -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import re,sys,subprocess
rows=[];cur=None
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m:
        cur={"name":m.group(1)};rows.append(cur);continue
    for key,pat in (("vgpr",r" VGPRs: (\d+)"),("agpr",r"AGPRs: (\d+)"),("sgpr",r" SGPRs: (\d+)"),("scratch",r"ScratchSize \[bytes/lane\]: (\d+)"),("occ",r"Occupancy \[waves/SIMD\]: (\d+)"),("vspill",r"VGPRs Spill: (\d+)"),("sspill",r"SGPRs Spill: (\d+)"),("lds",r"LDS Size \[bytes/block\]: (\d+)")):
        m=re.search(pat,l)
        if m and cur is not None: cur[key]=int(m.group(1))
names=subprocess.run(["c++filt"]+[r["name"] for r in rows],capture_output=True,text=True).stdout.splitlines()
flt=re.compile(sys.argv[1]) if len(sys.argv)>1 else None
print("%-72s %5s %7s %6s %6s %4s"%("kernel","VGPR","scratch","vspill","sspill","occ"))
for r,n in zip(rows,names):
    n=n.split("(")[0].replace("void rtk::","")
    if flt and not flt.search(n): continue
    print("%-72s %5d %7d %6d %6d %4d"%(n[:72],r.get("vgpr",0),r.get("scratch",0),r.get("vspill",0),r.get("sspill",0),r.get("occ",0)))
' "$@"
