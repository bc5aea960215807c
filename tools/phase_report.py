#!/usr/bin/env python3
"""Renders one frame of a bench workload with the diagnostic (RT_PHASE_TIMING) library and
prints the share of wave time per section of the pooled kernel.  See tools/phase_timing.sh."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-engine_amd"))
sys.path.insert(0, ROOT)
NAMES = ["setup", "primary_misc", "vertex_setup", "pool_fill", "handout", "steal", "descent", "leaf", "pool_misc", "bsdf",
         "accum", "n_rounds", "n_steps", "n_pools", "n_tail_rounds", "foreign"]


def main():
    import bench
    import pyrt
    wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
    kind, w, h, spp, mode, nph, k = bench.WORKLOADS[wl]
    spp = int(os.environ.get("RT_SPP", min(spp, 32)))
    scene = pyrt.Scene(kind, w, h)
    ctx = pyrt.Context(scene)
    p = pyrt.make_params(w, h, spp, mode=mode, seed=1)
    ctx.render(p, want_accum=False)  # warm-up
    # the dump goes to stderr (fd 2): capture it around one render
    with tempfile.TemporaryFile() as tmp:
        old = os.dup(2)
        os.dup2(tmp.fileno(), 2)
        try:
            _, _, st = ctx.render(p, want_accum=False)
        finally:
            os.dup2(old, 2)
        tmp.seek(0)
        lines = [l for l in tmp.read().decode().splitlines() if "phase_clocks" in l]
    clocks = json.loads(lines[-1])["phase_clocks"]
    d = dict(zip(NAMES, clocks))
    tot = sum(clocks[:11]) + d.get("foreign", 0)
    rays = st.rays_closest + st.rays_shadow
    out = {"workload": wl, "spp": spp, "kernel_ms": st.kernel_ms, "rays": rays,
           "share": {n: round(d[n] / tot, 4) for n in NAMES[:11] + ["foreign"]},
           "rounds_per_pool": d["n_rounds"] / max(d["n_pools"], 1), "steps_per_round": d["n_steps"] / max(d["n_rounds"], 1),
           "tail_round_frac": d["n_tail_rounds"] / max(d["n_rounds"], 1),
           "clocks_per_step_descent": d["descent"] / max(d["n_steps"], 1), "clocks_per_round_leaf": d["leaf"] / max(d["n_rounds"], 1),
           "wave_clocks_per_64rays": tot / (rays / 64.0)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
