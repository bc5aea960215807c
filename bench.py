#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X path-tracing hot path.

Workload (BASELINE.json configs[1]): 1024x1024, -m 1 (path), -N 128 spp, Cornell
box with meshes/example_low_res.off in slot 3 (1,222 triangles), pixel-RNG seed 1.
A "step" is one whole frame: zero the accumulator, integrate every sample of every
owned pixel (one HIP launch per rank), reduce over ranks (N>1), resolve on rank 0.
Metric: Mrays/s = (closest-hit + shadow rays actually cast) / wall time, whole job.

  python bench.py --gpus N --steps K --warmup W
N>1 is launched by torch.distributed.run, one rank per GPU (RCCL); the frame is
tile-sharded over ranks, i.e. total work is fixed: "scaling": "strong".
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-engine_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    # name: (scene, width, height, spp, mode, photons, k)
    "C2": ("lowres", 1024, 1024, 128, 1, 0, 0),
    "C1": ("cubes", 256, 256, 8, 1, 0, 0),
    "C3": ("cubes", 1024, 1024, 16, 0, 50000, 10),
    "C4": ("hires", 2048, 2048, 512, 1, 0, 0),
    "C5": ("stress", 1024, 1024, 256, 1, 0, 0),
}


def cpu_baseline(scene_kind, mode, spp_full):
    """The REAL reference (oracle/_ref/ref_harness = reference sources + our driver)
    timed single-threaded on a bounded sample of the same workload: same scene, same
    mode, 8 spp at 96x96 (~74k samples, ~10 s; brute force costs ~0.13 ms per sample on
    the low-res scene).  Rays are counted by the oracle's legacy mode on the same input
    (its image is byte-identical to the reference's, so the counts are the reference's)."""
    import orc
    import pyrt
    w = h = 96
    n = 8
    meshes = pyrt.MESH_DIR
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    scene = pyrt.Scene(scene_kind, w, h)
    p = pyrt.make_params(w, h, n, mode=mode, rng_mode=pyrt.RNG_LEGACY)
    t0 = time.perf_counter()
    _, _, st = orc.render(scene, p, math_mode=orc.MATH_LIBM)
    t_port = time.perf_counter() - t0
    rays = st.rays_closest + st.rays_shadow
    # the same sample on ALL host cores: the CPU restatement in pixel-RNG mode (independent
    # pixels, OpenMP), brute force like the reference.  (The reference itself cannot run
    # multi-threaded: one global RNG.)
    pp = pyrt.make_params(w, h, n, mode=mode, rng_mode=pyrt.RNG_PIXEL)
    t0 = time.perf_counter()
    _, _, stp = orc.render(scene, pp, math_mode=orc.MATH_DET, threads=0)
    t_all = time.perf_counter() - t0
    all_cores = {"port_all_cores_value": (stp.rays_closest + stp.rays_shadow) / t_all / 1e6,
                 "port_all_cores_threads": len(os.sched_getaffinity(0))}
    sample = "%s scene, %dx%d, -m %d -N %d, legacy RNG seed 1 (%d rays)" % (scene_kind, w, h, mode, n, rays)
    if os.path.exists(harness):
        try:
            with tempfile.TemporaryDirectory() as tmp:
                r = subprocess.run([harness, "time", meshes, scene_kind, str(w), str(h), str(mode), str(n), "0", "0"],
                                   cwd=tmp, capture_output=True, text=True, check=True, timeout=300)
            secs = json.loads(r.stdout.strip().splitlines()[-1])["seconds"]
            return {"value": rays / secs / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "reference", "sample": sample,
                    "seconds": secs, "port_value": rays / t_port / 1e6, **all_cores}
        except Exception as e:  # the baseline must never cost the bench line: fall back to the port
            sample += " [reference harness failed: %s]" % type(e).__name__
    return {"value": rays / t_port / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port", "sample": sample,
            "seconds": t_port, **all_cores}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (debug only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--accel", default="bvh", choices=["bvh", "brute"])
    ap.add_argument("--leaf", type=int, default=0, help="BVH leaf size override (debug)")
    ap.add_argument("--lpp", type=int, default=0, help="samples of a pixel per wave override (debug)")
    args = ap.parse_args()

    import torch
    import pyrt
    from pyrt import dist as rdist

    # RT_DIST_BACKEND=gloo + RT_SHARE_GPU=1: rehearsal of the N-rank flow on ONE GPU
    backend = os.environ.get("RT_DIST_BACKEND", "nccl")
    rank, world, local = rdist.init_from_env(backend)
    if os.environ.get("RT_SHARE_GPU") == "1":
        local = 0
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d (launch with torch.distributed.run)" % (world, args.gpus))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    kind, w, h, spp, mode, nph, k = WORKLOADS[args.workload]
    if args.spp:
        spp = args.spp
    scene = pyrt.Scene(kind, w, h)
    ctx = pyrt.Context(scene, device=local, bvh_leaf_max=args.leaf)  # raises if the HIP library / a gfx950 device is missing
    accel = pyrt.ACCEL_BRUTE if args.accel == "brute" else pyrt.ACCEL_BVH
    use_ph = 1 if nph else 0
    if nph:
        pos, dr, wt = ctx.emit_photons(nph, seed=1)
        kp, kd_, _ = pyrt.kd_order(pos, dr, wt)
        ctx.set_photons(kp, kd_)
    params = pyrt.make_params(w, h, spp, mode=mode, seed=1, accel=accel, rank=rank, world=world, tile=32,
                              use_photons=use_ph, k=k, photons_requested=nph, lanes_per_pixel=args.lpp)

    accum = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)
    bg = torch.from_numpy(pyrt.background(w, h)).to(dev)
    out = torch.empty((h, w, 3), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        accum.zero_()
        ctx.render_device(params, accum.data_ptr(), stream)
        rdist.reduce_frame(accum, dst=0)
        if rank == 0:
            ctx.resolve_device(w, h, spp, accum.data_ptr(), bg.data_ptr(), out.data_ptr(), stream)

    # one counted pass (untimed): rays, BVH node fetches and triangle tests are
    # deterministic per frame, so they are measured once
    params.collect_stats = 1
    accum.zero_()
    st = ctx.render_device(params, accum.data_ptr(), stream, stats=True)
    params.collect_stats = 0
    local_counts = [st.rays_closest, st.rays_shadow, st.nodes_visited, st.tris_tested, st.samples, st.knn_queries,
                    st.kd_visited]
    tot = rdist.sum_over_ranks(local_counts, dev)
    rays_per_frame = tot[0] + tot[1]

    for _ in range(args.warmup):
        step()
    rdist.barrier()
    torch.cuda.synchronize()
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    rdist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = rdist.max_over_ranks(elapsed, dev)
    kernel_ms, launches = ctx.profile_collect()

    # roofline of the dominant kernel (k_render) on THIS rank: algorithmic bytes per
    # launch = bytes per unit x units of this launch: one BVH node record
    # per node fetched (32 B: the packed f16 node this build traverses; §8d priced a
    # 64-B float node), 48 B per triangle record tested, 16 B per pixel-sample for the
    # accumulator, 32 B (position + direction) per kd node visited
    alg_bytes = 32 * local_counts[2] + 48 * local_counts[3] + 16 * local_counts[4] + 32 * local_counts[6]
    avg_ms = kernel_ms / max(launches, 1)
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0

    # HBM traffic of the same launch from PMC counters (tools/traffic.sh, committed under
    # profiles/): FETCH_SIZE (doubled, gfx950 correction) + WRITE_SIZE
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_traffic_%s.json" % args.workload)
    if os.path.exists(tpath) and not args.spp and world == 1 and args.accel == "bvh":
        traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")

    if rank == 0:
        res = {
            "metric": "Mrays/s (primary+secondary) at 1024x1024/128spp path trace",
            "value": rays_per_frame * args.steps / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %s scene (%d triangles), %dx%d, -m %d -N %d, pixel RNG seed 1, %s"
                                   % (args.workload, kind, scene.desc.n_triangles, w, h, mode, spp, args.accel),
                       "parallelism": "tiles32x%d" % world,
                       "rays_per_frame": rays_per_frame, "samples_per_frame": tot[4],
                       "knn_queries_per_frame": tot[5]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         # what HBM actually carried (PMC bytes / the same kernel time): the honest utilisation
                         "traffic_gbs": (traffic / (avg_ms * 1e-3) / 1e9) if traffic and avg_ms > 0 else None,
                         "traffic_frac": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and avg_ms > 0 else None,
                         "kernel": "k_render", "kernel_ms_avg": avg_ms, "launches": launches,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "nodes_per_ray": tot[2] / max(rays_per_frame, 1), "tris_per_ray": tot[3] / max(rays_per_frame, 1),
                         # lanes doing a node step / a leaf test per wave-level step of the traversal
                         # loop (rank 0's share; diagnostics of the counted pass)
                         "lanes_per_node_step": st.nodes_visited / max(st.reserved[0], 1),
                         "leaf_phases_per_node_step": st.reserved[1] / max(st.reserved[0], 1),
                         "lanes_at_leaf_per_node_step": st.reserved[2] / max(st.reserved[0], 1),
                         "lanes_without_ray_per_node_step": st.reserved[3] / max(st.reserved[0], 1),
                         "note": "scene is %.2f MB (L2/Infinity-Cache resident): achieved is the ALGORITHMIC byte rate, "
                                 "served mostly by caches; traffic = measured HBM bytes per launch"
                                 % ((32 * ctx.bvh_info().n_nodes + 48 * scene.desc.n_triangles) / 1e6)},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(kind, mode, spp)
        print(json.dumps(res), flush=True)
    ctx.close()
    rdist.shutdown()


if __name__ == "__main__":
    main()
