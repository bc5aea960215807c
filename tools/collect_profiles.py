#!/usr/bin/env python3
"""Copy the round's measurement summaries from gpurun_out/round/ into profiles/ (tracked),
and write profiles/<round>_pmc_<WL>.json — the counters bench.py falls back to (marked as such,
and only for the same kernel sources) when rocprofv3 cannot run.  usage: collect_profiles.py r04"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

R = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", "round")
dst = os.path.join(ROOT, "profiles")
for wl in ("C1", "C2", "C3", "C4", "C5", "C5x8"):
    p = os.path.join(src, "bench_%s.json" % wl)
    if not os.path.exists(p) or os.path.getsize(p) == 0:
        print("missing", p)
        continue
    d = json.loads(open(p).read().strip().splitlines()[-1])
    json.dump(d, open(os.path.join(dst, "%s_bench_%s.json" % (R, wl)), "w"), indent=1)
    c = d["roofline"].get("counters_per_launch")
    if c:
        json.dump({"workload": wl, "kernel_source_hash": d["roofline"]["kernel_source_hash"], "counters": c,
                   "source": "bench.py in-run rocprofv3 passes, " + R},
                  open(os.path.join(dst, "%s_pmc_%s.json" % (R, wl)), "w"), indent=1)
    print(wl, round(d["value"]), "Mrays/s", round(d["ms_per_step"], 2), "ms", d["roofline"]["bound"], d["roofline"]["frac"])
for name, to in (("c2_kernel_stats.csv", R + "_c2_kernel_stats.csv"), ("c2_pmc_spp16.txt", R + "_c2_pmc_spp16.txt"),
                 ("c5_pmc_spp32.txt", R + "_c5_pmc_spp32.txt"), ("phase_C2.json", R + "_phase_C2.json"),
                 ("phase_C4.json", R + "_phase_C4.json"), ("phase_C5.json", R + "_phase_C5.json"), ("builders.txt", R + "_builders.txt"),
                 ("stream_vs_megakernel.json", R + "_stream_vs_megakernel.json"),
                 ("bench_C2_under_rocprof.json", R + "_bench_C2_under_rocprof.json"), ("bench_C2_tuned.json", R + "_bench_C2_tuned.json"),
                 ("c2_pmc_ta.txt", R + "_c2_pmc_ta.txt"), ("c5_pmc_ta.txt", R + "_c5_pmc_ta.txt"),
                 ("c2_pmc_issue.txt", R + "_c2_pmc_issue.txt"), ("c5x8_pmc_spp8.txt", R + "_c5x8_pmc_spp8.txt"),
                 ("c3_pmc.txt", R + "_c3_pmc.txt"), ("c3_pmc_issue.txt", R + "_c3_pmc_issue.txt"), ("c3_pmc_ta.txt", R + "_c3_pmc_ta.txt"),
                 ("kernel_resources.txt", R + "_kernel_resources.txt")):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, to))
    else:
        print("missing", p)
print("kernel source hash now:", bench.kernel_source_hash())
