#!/usr/bin/env python3
"""Measured-cost tuning of the BVH (rt_bvh_tune) tried on a workload: probe size / spp / budget -> full-frame time."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-engine_amd"))
import numpy as np
import torch
import pyrt
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="lowres"); ap.add_argument("--size", type=int, default=1024); ap.add_argument("--spp", type=int, default=128)
ap.add_argument("--probe", type=int, default=128); ap.add_argument("--probe-spp", type=int, default=1); ap.add_argument("--budget", type=float, default=5.0)
ap.add_argument("--mode", type=int, default=1)
a = ap.parse_args()
s = pyrt.Scene(a.scene, a.size, a.size)
ctx = pyrt.Context(s)
p = pyrt.make_params(a.size, a.size, a.spp, mode=a.mode, seed=1)
acc = torch.zeros((a.size * a.size, 4), dtype=torch.float32, device="cuda")
def frame(stats=False):
    acc.zero_(); torch.cuda.synchronize()
    pp = pyrt.make_params(a.size, a.size, a.spp, mode=a.mode, seed=1, collect_stats=1 if stats else 0)
    st = ctx.render_device(pp, acc.data_ptr(), 0, stats=True)
    return st
for _ in range(2): frame()
st0 = frame(); ref = acc.clone()
sc0 = frame(True)
rays = sc0.rays_closest + sc0.rays_shadow
print("before: %.2f ms  nodes/ray %.3f tris/ray %.3f" % (st0.kernel_ms, sc0.nodes_visited / rays, sc0.tris_tested / rays), flush=True)
probe = pyrt.make_params(a.probe, a.probe, a.probe_spp, mode=a.mode, seed=7)
rep = ctx.tune(probe, a.budget)
print("tune: %d probes, %d accepted, cost %.0f -> %.0f (%.1f %%), %.2f s" % (rep.probes, rep.accepted, rep.cost_before, rep.cost_after, 100 * (rep.cost_after / rep.cost_before - 1), rep.seconds), flush=True)
for _ in range(2): frame()
st1 = frame()
same = bool(torch.equal(acc.view(torch.int32), ref.view(torch.int32)))
sc1 = frame(True)
print("after:  %.2f ms  nodes/ray %.3f tris/ray %.3f  frame identical: %s" % (st1.kernel_ms, sc1.nodes_visited / rays, sc1.tris_tested / rays, same), flush=True)
