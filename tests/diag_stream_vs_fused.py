#!/usr/bin/env python3
"""k_trace_stream on the REAL secondary rays of a frame (dumped by the oracle in casting
order), queued the way a wavefront integrator would queue them: one queue per path depth,
[vertex][3 shadow rays + bounce ray].  Prints Grays/s per stage and overall, next to the
megakernel's whole-frame rate on the same frame.  usage: diag_stream_vs_fused.py [scene] [w] [spp]"""
import json
import os
import sys
import time

import numpy as np

# (lives under tests/ because it uses the ORACLE to dump the frame's rays: test infrastructure,
# never imported by the product or by tools/)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-engine_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import orc
    import pyrt
    kind = sys.argv[1] if len(sys.argv) > 1 else "lowres"
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 640
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    scene = pyrt.Scene(kind, w, w)
    p = pyrt.make_params(w, w, spp, seed=1)
    t0 = time.time()
    rays = orc.dump_rays(scene, p)
    tag = rays[:, 3].view(np.uint32)
    kindv, depth = tag & 255, tag >> 8
    ctx = pyrt.Context(scene)
    _, _, st = ctx.render(p, want_accum=False)
    _, _, st = ctx.render(p, want_accum=False)
    mega = (st.rays_closest + st.rays_shadow) / st.kernel_ms / 1e6
    out = {"scene": kind, "w": w, "spp": spp, "rays_total": len(rays), "dump_s": time.time() - t0, "megakernel_Grays": mega,
           "megakernel_ms": st.kernel_ms, "stages": []}
    stream = torch.cuda.current_stream().cuda_stream
    total_ms, total_rays = 0.0, 0
    for d in range(3):
        sel = ((kindv == 1) & (depth == d)) | ((kindv == 0) & (depth == d + 1))
        q = rays[sel]
        n = len(q)
        if n == 0:
            continue
        O = q[:, 0:4].copy()
        O[:, 3] = (q[:, 3].view(np.uint32) & 1).view(np.float32)  # kind bit only
        D = q[:, 4:8].copy()
        reps = max(1, 24_000_000 // n)  # repeat the queue to a realistic launch size
        dO = torch.from_numpy(np.tile(O, (reps, 1))).cuda()
        dD = torch.from_numpy(np.tile(D, (reps, 1))).cuda()
        nn = n * reps
        res = torch.zeros((nn, 2), dtype=torch.int32, device="cuda")
        ctx.trace_stream_device(dO.data_ptr(), dD.data_ptr(), nn, res.data_ptr(), stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            ctx.trace_stream_device(dO.data_ptr(), dD.data_ptr(), nn, res.data_ptr(), stream)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3 / reps  # per original queue
        out["stages"].append({"depth": d, "rays": n, "ms": ms, "Grays": n / ms / 1e6})
        total_ms += ms
        total_rays += n
    prim = int(((kindv == 0) & (depth == 0)).sum())
    out["secondary_rays"] = total_rays
    out["stream_secondary_ms"] = total_ms
    out["stream_secondary_Grays"] = total_rays / total_ms / 1e6
    out["primary_rays"] = prim
    print(json.dumps(out))


if __name__ == "__main__":
    main()
