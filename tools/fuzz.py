"""One-off large-scale exactness check on the GPU: BVH traversal vs the exhaustive loop,
random rays (closest + any) and whole frames (BVH accel vs brute accel).  Prints mismatches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-engine_amd"))
import numpy as np, pyrt
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 11)
# RT_FUZZ_TUNE=probes: every context's tree goes through rt_bvh_tune first (measured-cost tuning: subtree moves, slot flips)
TUNE = int(os.environ.get("RT_FUZZ_TUNE", "0"))
def tuned(ctx, kind):
    if TUNE and not os.environ.get("RT_BVH_GPU"):
        rep = ctx.tune(pyrt.make_params(64, 64, 1, seed=5), 120.0, TUNE)
        print(kind, "tuned: %d probes, %d changes kept" % (rep.probes, rep.accepted), flush=True)
    return ctx
bad = 0
for kind, n in (("cubes", 4_000_000), ("lowres", 2_000_000), ("hires", 400_000), ("stress", 6_000)):
    s = pyrt.Scene(kind, 64, 64); ctx = tuned(pyrt.Context(s), kind)
    rays = np.zeros(n, pyrt.RAY_DTYPE)
    o = rng.uniform(-1.45, 1.45, (n, 3)).astype(np.float32)
    o[: n // 4] = rng.uniform(-3, 3, (n // 4, 3))            # outside the box too
    d = rng.normal(0, 1, (n, 3)).astype(np.float32)
    d[::7, rng.integers(0, 3)] = 0.0                         # axis-parallel components
    d[::11] *= np.float32(1e-3)                              # unnormalised, like shadow rays
    # origins ON surfaces: hit points of a first batch
    rays["origin"], rays["direction"] = o, d
    h0 = ctx.trace(rays[: n // 2])
    ok = h0["hit"] != 0
    on = rays["origin"][: n // 2][ok] + rays["direction"][: n // 2][ok] * h0["d"][ok, None]
    m = min(len(on), n // 4)
    rays["origin"][n // 2 : n // 2 + m] = on[:m]
    for any_hit in (False, True):
        kindT = pyrt.TRACE_ANY if any_hit else pyrt.TRACE_CLOSEST
        a = ctx.trace(rays, pyrt.ACCEL_BVH, kindT)
        b = ctx.trace(rays, pyrt.ACCEL_BRUTE, kindT)
        if any_hit:
            mis = int((a["hit"] != b["hit"]).sum())
        else:
            mis = int((a.view(np.uint8).reshape(n, -1) != b.view(np.uint8).reshape(n, -1)).any(1).sum())
        bad += mis
        print(kind, "any" if any_hit else "closest", n, "rays, mismatches", mis, flush=True)
    ctx.close()
for kind, w, spp in (("lowres", 192, 8), ("cubes", 256, 16), ("hires", 96, 4)):
    s = pyrt.Scene(kind, w, w); ctx = tuned(pyrt.Context(s), kind)
    _, a, _ = ctx.render(pyrt.make_params(w, w, spp, seed=21))
    _, b, _ = ctx.render(pyrt.make_params(w, w, spp, seed=21, accel=pyrt.ACCEL_BRUTE))
    mis = int((a.view(np.uint32) != b.view(np.uint32)).any(-1).sum())
    bad += mis
    print(kind, "frame %dx%dx%d BVH vs brute: differing pixels" % (w, w, spp), mis, flush=True)
    ctx.close()
print("TOTAL MISMATCHES", bad)
