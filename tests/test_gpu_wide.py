"""The 4-wide BVH on the GPU (rt_options.bvh_width = 4: rtbvh::Node4x16 records, Trav::step_wide).
Same exactness bar as the binary tree — BVH == exhaustive loop == oracle, bit for bit, closest and
any hit, on all four scenes — and the frames of the pooled render kernel on it must equal the oracle's
and the binary tree's.  The wide form is opt-in (measured slower on MI355X: DESIGN.md §4.4)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import orc
import pyrt
from raybatch import ray_batch

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("kind,n_loop,n_bvh", [("cubes", 60000, 0), ("lowres", 40000, 300000), ("hires", 30000, 300000),
                                               ("stress", 1500, 150000)])
def test_wide_tree_is_exact(kind, n_loop, n_bvh):
    s = pyrt.Scene(kind, 256, 256)
    ctx = pyrt.Context(s, bvh_width=4)
    two = pyrt.Context(s, bvh_width=2)
    bi, b2 = ctx.bvh_info(), two.bvh_info()
    assert 0 < bi.n_wide_nodes < bi.n_nodes and b2.n_wide_nodes == 0
    rays = ray_batch(s, n_loop, 4321)
    ref = orc.trace(s, rays)
    h = ctx.trace(rays, pyrt.ACCEL_BVH)
    assert np.array_equal(h.view(np.uint8), ref.view(np.uint8))
    assert np.array_equal(ctx.trace(rays, pyrt.ACCEL_BVH, pyrt.TRACE_ANY)["hit"], ref["hit"])
    assert np.array_equal(two.trace(rays, pyrt.ACCEL_BVH).view(np.uint8), ref.view(np.uint8))
    if n_bvh:
        rays = ray_batch(s, n_bvh, 99)
        want = orc.trace(s, rays, orc.ACCEL_OBVH)
        assert np.array_equal(ctx.trace(rays, pyrt.ACCEL_BVH).view(np.uint8), want.view(np.uint8))
        assert np.array_equal(ctx.trace(rays, pyrt.ACCEL_BVH, pyrt.TRACE_ANY)["hit"], want["hit"])
    ctx.close(), two.close()


@pytest.mark.parametrize("kind,w,h,spp,mode", [("cubes", 45, 37, 5, 1), ("lowres", 64, 48, 6, 1), ("hires", 48, 48, 4, 1),
                                               ("stress", 32, 32, 2, 1), ("lowres", 40, 40, 3, 0)])
def test_frames_on_the_wide_tree_vs_oracle(kind, w, h, spp, mode):
    s = pyrt.Scene(kind, w, h)
    ctx = pyrt.Context(s, bvh_width=4)
    bg = pyrt.background(w, h)
    p = pyrt.make_params(w, h, spp, mode=mode, seed=5)
    out, acc, st = ctx.render(p, bg)
    ref_out, ref_acc, ref_st = orc.render(s, p, math_mode=orc.MATH_DET, bg=bg, accel=orc.ACCEL_OBVH)
    assert np.array_equal(bits(acc), bits(ref_acc)) and np.array_equal(bits(out), bits(ref_out))
    assert (st.rays_closest, st.rays_shadow) == (ref_st.rays_closest, ref_st.rays_shadow)
    ctx.close()


@pytest.mark.parametrize("kind,w,spp", [("lowres", 512, 16), ("hires", 384, 8), ("stress", 256, 6)])
def test_wide_and_binary_trees_render_the_same_frame(kind, w, spp):
    """Pooled persistent kernel on the wide tree vs the same kernel on the binary tree vs sequential
    shading (one wave per workgroup, binary tree), larger frames: accumulators and ray counts equal."""
    s = pyrt.Scene(kind, w, w)
    res = []
    for width in (4, 2):
        ctx = pyrt.Context(s, bvh_width=width)
        _, a, sa = ctx.render(pyrt.make_params(w, w, spp, seed=3, collect_stats=1))
        res.append((a, sa.rays_closest, sa.rays_shadow, sa.nodes_visited / max(sa.rays_closest + sa.rays_shadow, 1)))
        if width == 4:
            _, b, sb = ctx.render(pyrt.make_params(w, w, spp, seed=3, no_pool=True))
            assert np.array_equal(bits(a), bits(b)) and (sa.rays_closest, sa.rays_shadow) == (sb.rays_closest, sb.rays_shadow)
        ctx.close()
    assert np.array_equal(bits(res[0][0]), bits(res[1][0])) and res[0][1:3] == res[1][1:3]
    print("\n%s: node records per ray wide %.2f, binary %.2f" % (kind, res[0][3], res[1][3]))
    assert res[0][3] < res[1][3]


def test_wide_tree_stack_budgets_and_scaled_scenes(tmp_path):
    """The collapse under other stack budgets (RT_BVH_WIDE_BUDGET: read once per process) and on
    scaled geometry (other f16 plane scales): frames and hits unchanged."""
    script = tmp_path / "wb.py"
    script.write_text('''
import sys, numpy as np
sys.path.insert(0, sys.argv[1] + "/ray-tracing-engine_amd"); sys.path.insert(0, sys.argv[1] + "/tests")
import pyrt
from raybatch import ray_batch
out = {}
for kind, w, spp in (("lowres", 96, 6), ("hires", 80, 4), ("stress", 40, 2)):
    s = pyrt.Scene(kind, w, w); ctx = pyrt.Context(s, bvh_width=4)
    _, acc, st = ctx.render(pyrt.make_params(w, w, spp, seed=17))
    out[kind] = acc; out[kind + "_rays"] = np.array([st.rays_closest, st.rays_shadow, ctx.bvh_info().n_wide_nodes])
    out[kind + "_hits"] = ctx.trace(ray_batch(s, 50000, 7)).view(np.uint8)
    ctx.close()
np.savez(sys.argv[2], **out)
''')
    runs = {}
    for name, env in (("default", {}), ("b30", {"RT_BVH_WIDE_BUDGET": "30"}), ("b26_nocompact", {"RT_BVH_WIDE_BUDGET": "26", "RT_COMPACT": "0"}),
                      ("binary", {"RT_BVH_WIDE": "0"})):  # (RT_BVH_WIDE overrides rt_options.bvh_width)
        out = tmp_path / (name + ".npz")
        r = subprocess.run([sys.executable, str(script), pyrt.ROOT, str(out)], env=dict(os.environ, **env), capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        runs[name] = np.load(out)
    assert runs["binary"]["stress_rays"][2] == 0 and runs["default"]["stress_rays"][2] > 0
    assert runs["b30"]["hires_rays"][2] < runs["default"]["hires_rays"][2]  # a looser budget merges more
    for name in ("b30", "b26_nocompact", "binary"):
        for k in runs["default"].files:
            a, b = runs["default"][k], runs[name][k]
            if k.endswith("_rays"):
                assert np.array_equal(a[:2], b[:2]), (name, k)
            else:
                assert np.array_equal(bits(a) if a.dtype == np.float32 else a, bits(b) if b.dtype == np.float32 else b), (name, k)


@pytest.mark.parametrize("scale", [1e-3, 37.0, 1e4])
def test_wide_tree_exact_on_scaled_scenes(scale):
    a = pyrt.Scene("lowres", 64, 64).arrays()
    cam = a["camera"] * np.float32(scale)
    lights = a["lights"].copy()
    lights[:, 0:3] *= np.float32(scale)
    sc = pyrt.ArrayScene(a["pos"] * np.float32(scale), a["nrm"], a["tri"], a["tri_begin"], a["vtx_begin"], a["materials"], lights, cam)
    ctx = pyrt.Context(sc, bvh_width=4)
    assert ctx.bvh_info().n_wide_nodes > 0
    rays = ray_batch(pyrt.Scene("lowres", 64, 64), 60000, 5)
    rays["origin"] *= np.float32(scale)
    got = ctx.trace(rays, pyrt.ACCEL_BVH)
    want = ctx.trace(rays, pyrt.ACCEL_BRUTE)
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))
    ctx.close()
