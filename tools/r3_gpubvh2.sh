#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python3 - > gpurun_out/create_times.log 2>&1 <<'PY'
import sys, time
sys.path.insert(0, "ray-tracing-engine_amd")
import pyrt
pyrt.Context(pyrt.Scene("cubes", 32, 32)).close()  # HIP initialisation
for kind in ("lowres", "hires", "stress", "stress8"):
    s = pyrt.Scene(kind, 256, 256)
    for name, b in (("host", pyrt.BVH_HOST), ("device", pyrt.BVH_DEVICE), ("device", pyrt.BVH_DEVICE)):
        t0 = time.perf_counter(); ctx = pyrt.Context(s, bvh_builder=b); dt = time.perf_counter() - t0
        bi = ctx.bvh_info()
        print("%-8s %-6s rt_create %.1f ms (tree %.1f ms) nodes %d depth %d" % (kind, name, dt * 1e3, bi.build_ms, bi.n_nodes, bi.max_depth), flush=True)
        ctx.close()
PY
cat gpurun_out/create_times.log
{
for wl in C2 C4 C5 C5x8; do
  st=3; [ $wl = C4 ] && st=2
  tools/exp.sh " :: --workload $wl --no-pmc --steps $st" "RT_BVH_GPU=1 :: --workload $wl --no-pmc --steps $st" "RT_BVH_GPU=1 RT_BVH_SLACK=5 :: --workload $wl --no-pmc --steps $st"
done
} > gpurun_out/ab_gpubvh.log 2>&1
cat gpurun_out/ab_gpubvh.log
RT_BVH_GPU=1 tools/run_guarded.sh gpurun_out/t_all_gpubvh.log 1100 python3 -m pytest tests -x -q -m gpu
