#!/usr/bin/env python3
"""Copy the round's measurement summaries from gpurun_out/round2/ into profiles/ (tracked),
and write profiles/r02_pmc_<WL>.json — the counters bench.py falls back to (marked as such,
and only for the same kernel sources) when rocprofv3 cannot run."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

src = os.path.join(ROOT, "gpurun_out", "round2")
dst = os.path.join(ROOT, "profiles")
for wl in ("C1", "C2", "C3", "C4", "C5", "C5x8"):
    p = os.path.join(src, "bench_%s.json" % wl)
    if not os.path.exists(p) or os.path.getsize(p) == 0:
        print("missing", p)
        continue
    d = json.loads(open(p).read().strip().splitlines()[-1])
    json.dump(d, open(os.path.join(dst, "r02_bench_%s.json" % wl), "w"), indent=1)
    c = d["roofline"].get("counters_per_launch")
    if c:
        json.dump({"workload": wl, "kernel_source_hash": d["roofline"]["kernel_source_hash"], "counters": c,
                   "source": "bench.py in-run rocprofv3 passes, round 2"},
                  open(os.path.join(dst, "r02_pmc_%s.json" % wl), "w"), indent=1)
    print(wl, round(d["value"]), "Mrays/s", round(d["ms_per_step"], 2), "ms", d["roofline"]["bound"], d["roofline"]["frac"])
for name, to in (("c2_kernel_stats.csv", "r02_c2_kernel_stats.csv"), ("c2_pmc_spp16.txt", "r02_c2_pmc_spp16.txt"),
                 ("c5_pmc_spp32.txt", "r02_c5_pmc_spp32.txt"), ("phase_C2.json", "r02_phase_C2.json"),
                 ("phase_C4.json", "r02_phase_C4.json"), ("phase_C5.json", "r02_phase_C5.json"), ("builders.txt", "r02_builders.txt"),
                 ("stream_vs_megakernel.json", "r02_stream_vs_megakernel.json"),
                 ("bench_C2_under_rocprof.json", "r02_bench_C2_under_rocprof.json"),
                 ("c2_pmc_ta.txt", "r02_c2_pmc_ta.txt"), ("c5_pmc_ta.txt", "r02_c5_pmc_ta.txt"),
                 ("c2_pmc_issue.txt", "r02_c2_pmc_issue.txt")):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, to))
    else:
        print("missing", p)
print("kernel source hash now:", bench.kernel_source_hash())
