"""The oracle's own CPU BVH (rt_oracle.cpp OBvh) must be a pure accelerator: the
restatement of RayTracer.h:27-53 gives the same hit records, frames and ray counts
with and without it.  (It is what lets the GPU parity tests put the ORACLE beside the
HIP path on the 11.7k- and 1M-triangle scenes, and what bench.py times as the CPU BVH
baseline.)  Also re-checks the committed pixel-mode frame goldens where that is cheap."""
import os

import numpy as np
import pytest

import orc
import pyrt
from raybatch import ray_batch


@pytest.mark.parametrize("kind,n", [("cubes", 100000), ("lowres", 30000), ("hires", 12000), ("stress", 250)])
def test_obvh_equals_exhaustive_loop_on_rays(kind, n):
    s = pyrt.Scene(kind, 256, 256)
    rays = ray_batch(s, n, 1234)
    a = orc.trace(s, rays)
    b = orc.trace(s, rays, orc.ACCEL_OBVH)
    assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    c = orc.trace(s, rays, orc.ACCEL_OBVH, pyrt.TRACE_ANY)
    assert np.array_equal(a["hit"], c["hit"]) and 0.3 < a["hit"].mean() < 1.0


def test_obvh_far_origins_and_scaled_scene():
    """Origins 1e3..1e5 scene extents away (the float Moller-Trumbore error grows with |o|)."""
    s = pyrt.Scene("lowres", 64, 64)
    rng = np.random.default_rng(8)
    n = 20000
    rays = np.zeros(n, pyrt.RAY_DTYPE)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dist = (10.0 ** rng.uniform(1, 5, n)).astype(np.float32)
    target = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    rays["origin"] = target - d * dist[:, None]
    rays["direction"] = d
    a = orc.trace(s, rays)
    b = orc.trace(s, rays, orc.ACCEL_OBVH)
    assert np.array_equal(a.view(np.uint8), b.view(np.uint8)) and a["hit"].mean() > 0.5


@pytest.mark.parametrize("kind,w,h,spp,mode", [("cubes", 48, 40, 6, 1), ("lowres", 40, 32, 3, 1), ("hires", 12, 12, 2, 1),
                                              ("cubes", 32, 32, 4, 0)])
def test_obvh_frames_equal_loop_frames(kind, w, h, spp, mode):
    s = pyrt.Scene(kind, w, h)
    for rng_mode, mm in ((pyrt.RNG_PIXEL, orc.MATH_DET), (pyrt.RNG_LEGACY, orc.MATH_LIBM)):
        p = pyrt.make_params(w, h, spp, mode=mode, seed=3, rng_mode=rng_mode)
        _, a, sa = orc.render(s, p, math_mode=mm)
        _, b, sb = orc.render(s, p, math_mode=mm, accel=orc.ACCEL_OBVH)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert (sa.rays_closest, sa.rays_shadow) == (sb.rays_closest, sb.rays_shadow)
        assert sb.tris_tested < sa.tris_tested and sb.nodes_visited > 0


def test_obvh_photon_frame():
    s = pyrt.Scene("cubes", 32, 32)
    p = pyrt.make_params(32, 32, 2, mode=0, seed=4, use_photons=1, k=5, photons_requested=2000, rng_mode=pyrt.RNG_LEGACY)
    _, a, _ = orc.render(s, p)
    _, b, _ = orc.render(s, p, accel=orc.ACCEL_OBVH)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_pixel_frame_goldens_are_the_oracles(golden):
    """tests/golden/pixel_frames.npz (make_pixel_goldens.py) against the oracle again —
    the cheap cases with the exhaustive loop, the rest through OBvh."""
    z = np.load(os.path.join(golden["dir"], "pixel_frames.npz"))
    names = sorted({k.split("/")[0] for k in z.files})
    assert len(names) >= 6
    for name in names:
        kind = name.split("_")[0]
        w, h, spp, mode, seed = (int(v) for v in z[name + "/cfg"])
        s = pyrt.Scene(kind, w, h)
        p = pyrt.make_params(w, h, spp, mode=mode, seed=seed)
        accel = orc.ACCEL_LOOP if kind in ("cubes", "lowres") else orc.ACCEL_OBVH
        _, acc, st = orc.render(s, p, math_mode=orc.MATH_DET, accel=accel)
        assert np.array_equal(acc.view(np.uint32), z[name + "/acc"]), name
        assert [st.rays_closest, st.rays_shadow] == z[name + "/rays"].tolist()
