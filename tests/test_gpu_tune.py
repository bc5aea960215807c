"""rt_bvh_tune — measured-cost tuning of the host-built BVH (csrc/bvh_build.cpp tuneMeasured): subtree moves and child
slot orders are kept only where a probe frame's counters fell.  The tree changes, the picture must not: frames, hits and
ray counts stay bit-identical (the reference's rayTrace is the exhaustive loop, RayTracer.h:27-53 — any tree over the
same triangles must return its hits), the tree stays a valid tree, and a probe LIMIT (not a time limit) makes it
deterministic."""
import os
import subprocess

import numpy as np
import pytest

import orc
import pyrt
from raybatch import ray_batch

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(bool(os.environ.get("RT_BVH_GPU")) or bool(os.environ.get("RT_NODES")),
                                 reason="RT_BVH_GPU / RT_NODES force a builder / node format: the tuner works on a host-built tree of 32-byte records")]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def walk(nodes, n_tri_records):
    """Every triangle record referenced exactly once, every node reached exactly once; returns the deepest leaf level."""
    child = nodes[:, 12:14].view(np.int32)
    seen_t = np.zeros(n_tri_records, np.int32)
    seen_n = np.zeros(len(nodes), np.int32)
    deepest, stack = 0, [(0, 0)]
    while stack:
        i, d = stack.pop()
        seen_n[i] += 1
        for c in child[i]:
            if c >= 0:
                stack.append((int(c), d + 1))
            else:
                code = ~int(c)
                first, cnt = code >> 3, (code & 7) + 1
                seen_t[first:first + cnt] += 1
                deepest = max(deepest, d + 1)
    assert (seen_n == 1).all() and (seen_t == 1).all()
    return deepest


def test_tuned_tree_same_frame_same_hits_fewer_visits_and_deterministic():
    s = pyrt.Scene("lowres", 96, 96)
    p = pyrt.make_params(96, 96, 4, seed=3, collect_stats=1)
    rays = ray_batch(s, 40000, 5)
    exports = []
    for run in range(2):
        ctx = pyrt.Context(s)
        _, acc0, st0 = ctx.render(p)
        depth0 = ctx.bvh_info().max_depth
        want = ctx.trace(rays, pyrt.ACCEL_BRUTE)
        rep = ctx.tune(pyrt.make_params(64, 64, 1, seed=7), 120.0, 500)
        assert 0 < rep.accepted and rep.probes <= 501 and rep.cost_after < rep.cost_before
        _, acc1, st1 = ctx.render(p)
        assert np.array_equal(bits(acc0), bits(acc1))
        assert (st0.rays_closest, st0.rays_shadow) == (st1.rays_closest, st1.rays_shadow)
        assert st1.nodes_visited + 1.5 * st1.tris_tested < st0.nodes_visited + 1.5 * st0.tris_tested
        got = ctx.trace(rays, pyrt.ACCEL_BVH)
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8))
        assert np.array_equal(ctx.trace(rays, pyrt.ACCEL_BVH, pyrt.TRACE_ANY)["hit"], want["hit"])
        nodes, tris = ctx.bvh_export()
        assert walk(nodes, len(tris)) == ctx.bvh_info().max_depth <= depth0  # no leaf deeper than before: the stacks were sized for it
        exports.append(nodes.copy())
        ctx.close()
    assert np.array_equal(exports[0], exports[1])  # a probe limit gives the same tree on every run


def test_tuned_frame_vs_oracle_path_and_ray_modes():
    s = pyrt.Scene("cubes", 64, 48)
    ctx = pyrt.Context(s)
    ctx.tune(pyrt.make_params(64, 48, 2, seed=11), 60.0, 300)
    for mode in (pyrt.MODE_RAY, pyrt.MODE_PATH):
        p = pyrt.make_params(64, 48, 5, mode=mode, seed=2)
        _, acc, st = ctx.render(p)
        _, ref, rst = orc.render(s, p, math_mode=orc.MATH_DET)
        assert np.array_equal(bits(acc), bits(ref)) and (st.rays_closest, st.rays_shadow) == (rst.rays_closest, rst.rays_shadow)
    ctx.close()


def test_tune_refuses_what_it_cannot_do():
    s = pyrt.Scene("lowres", 32, 32)
    dev = pyrt.Context(s, bvh_builder=pyrt.BVH_DEVICE)
    with pytest.raises(pyrt.RtError):
        dev.tune(pyrt.make_params(32, 32, 1), 1.0, 10)  # no host-side float tree to move subtrees in
    dev.close()
    ctx = pyrt.Context(s)
    with pytest.raises(pyrt.RtError):
        ctx.tune(pyrt.make_params(32, 32, 1, accel=pyrt.ACCEL_BRUTE), 1.0, 10)
    with pytest.raises(pyrt.RtError):
        ctx.tune(pyrt.make_params(0, 32, 1), 1.0, 10)
    rep = ctx.tune(pyrt.make_params(32, 32, 1), 0.0, 10)  # no budget: nothing happens
    assert rep.probes == 0 and rep.accepted == 0
    ctx.close()


def test_application_tune_flag_writes_the_same_picture(tmp_path):
    app = os.path.join(pyrt.ROOT, "ray-tracing-engine_amd", "bin", "RayTracer")
    outs = []
    for extra in ([], ["-tune", "0.5"]):
        d = tmp_path / ("t" + str(len(extra)))
        d.mkdir()
        r = subprocess.run([app, "-width", "64", "-height", "48", "-m", "1", "-N", "4", "-scene", "lowres", "-meshdir", pyrt.MESH_DIR, "-o", "o.ppm"] + extra,
                           cwd=d, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        if extra:
            assert "BVH tuned on 64x48 probe frames" in r.stdout
        outs.append((d / "o.ppm").read_bytes())
    assert outs[0] == outs[1]
