#!/usr/bin/env python3
"""Feasibility of the wavefront trace stage: rays/s of k_trace_stream (lean persistent
traversal over a ray queue in HBM) on a path-vertex-like ray mix of a bench scene:
per vertex 3 shadow rays (any-hit, towards the lights) + 1 bounce ray (closest hit,
random hemisphere direction).  usage: tools/stream_bench.py [C2|C4|C5] [vertices]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-engine_amd"))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    import pyrt
    wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
    nv = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
    kind = bench.WORKLOADS[wl][0]
    scene = pyrt.Scene(kind, 1024, 1024)
    ctx = pyrt.Context(scene)
    a = scene.arrays()
    rng = np.random.default_rng(1)
    # path vertices: first hits of camera rays, then one random bounce from them
    cam = a["camera"]
    u, v = rng.random(nv, np.float32), rng.random(nv, np.float32)
    d = cam[1] + u[:, None] * cam[2] + v[:, None] * cam[3] - cam[0]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros(nv, pyrt.RAY_DTYPE)
    rays["origin"], rays["direction"] = cam[0], d.astype(np.float32)
    pts = []
    for depth in range(2):
        h = ctx.trace(rays)
        ok = h["hit"] == 1
        p = rays["origin"][ok] + rays["direction"][ok] * h["d"][ok, None]
        pts.append(p.astype(np.float32))
        nd = rng.normal(size=(len(p), 3)).astype(np.float32)
        nd /= np.linalg.norm(nd, axis=1, keepdims=True)
        rays = np.zeros(len(p), pyrt.RAY_DTYPE)
        rays["origin"], rays["direction"] = p, nd
    P = np.concatenate(pts)[:nv]
    m = len(P)
    lights = a["lights"][:, 0:3]
    O = np.zeros((m, 4, 4), np.float32)
    D = np.zeros((m, 4, 4), np.float32)
    O[:, :, 0:3] = P[:, None, :]
    for li in range(3):
        D[:, li, 0:3] = lights[li] - P + rng.normal(0, 0.01, (m, 3)).astype(np.float32)
        O[:, li, 3] = np.array([1], np.uint32).view(np.float32)[0]  # any-hit
    bd = rng.normal(size=(m, 3)).astype(np.float32)
    bd /= np.linalg.norm(bd, axis=1, keepdims=True)
    D[:, 3, 0:3] = bd
    n = 4 * m
    dO = torch.from_numpy(O.reshape(n, 4)).cuda()
    dD = torch.from_numpy(D.reshape(n, 4)).cuda()
    res = torch.zeros((n, 2), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    ctx.trace_stream_device(dO.data_ptr(), dD.data_ptr(), n, res.data_ptr(), stream)
    torch.cuda.synchronize()
    # correctness on a sample against rt_trace
    r = res.cpu().numpy().view(np.uint32).reshape(m, 4, 2)
    sub = slice(0, 20000)
    for kk in range(4):
        rr = np.zeros(20000, pyrt.RAY_DTYPE)
        rr["origin"], rr["direction"] = O[sub, kk, 0:3], D[sub, kk, 0:3]
        if kk < 3:
            hh = ctx.trace(rr, pyrt.ACCEL_BVH, pyrt.TRACE_ANY)
            assert np.array_equal(hh["hit"].astype(np.uint32), r[sub, kk, 0]), kk
        else:
            hh = ctx.trace(rr)
            hit = hh["hit"] == 1
            assert np.array_equal(r[sub, kk, 0][hit], hh["d"][hit].view(np.uint32))
            assert (r[sub, kk, 0][~hit] == 0xFFFFFFFF).all()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        ctx.trace_stream_device(dO.data_ptr(), dD.data_ptr(), n, res.data_ptr(), stream)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(json.dumps({"workload": wl, "rays": n, "ms": ms, "Grays_per_s": n / ms / 1e6,
                      "shadow_occluded": float(r[:, 0:3, 0].mean()), "bounce_hit": float((r[:, 3, 0] != 0xFFFFFFFF).mean())}))


if __name__ == "__main__":
    main()
