"""The BVH built on the device (csrc/bvh_gpu.hip, rt_options.bvh_builder = RT_BVH_DEVICE — what RT_BVH_AUTO takes from
8,192 triangles): the host builder's split rules as kernels.  Same arrays, same exactness bar as the host SAH builder —
BVH == exhaustive loop == oracle on all four scenes — plus the structural invariants, the depth cap, determinism, and the
claim itself: the HOST builder's tree (node count, depth, node visits and triangle tests per ray), in a fraction of its
time.  (The whole GPU suite also runs on device-built trees with RT_BVH_GPU=1.)"""
import os

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
def test_device_built_tree_is_exact(kind, n_loop, n_bvh):
    s = pyrt.Scene(kind, 256, 256)
    ctx = pyrt.Context(s, bvh_builder=pyrt.BVH_DEVICE)
    bi = ctx.bvh_info()
    want = int(os.environ.get("RT_BVH_GPU", pyrt.BVH_DEVICE))  # (the variable overrides the option: the suite runs under each builder)
    if kind == "cubes":  # (34 triangles: a single part — the hybrid builder hands such a scene to the device builder's own path)
        assert bi.builder in (want, pyrt.BVH_DEVICE) or s.desc.n_triangles < 16
    else:
        assert bi.builder == want
    rays = ray_batch(s, n_loop, 4321)
    ref = orc.trace(s, rays)
    h = ctx.trace(rays, pyrt.ACCEL_BVH)
    assert np.array_equal(h.view(np.uint8), ref.view(np.uint8))
    assert np.array_equal(ctx.trace(rays, pyrt.ACCEL_BVH, pyrt.TRACE_ANY)["hit"], ref["hit"])
    assert np.array_equal(ctx.trace(rays, pyrt.ACCEL_BRUTE).view(np.uint8), ref.view(np.uint8))  # (its trisRef too)
    if n_bvh:
        rays = ray_batch(s, n_bvh, 99)
        assert np.array_equal(ctx.trace(rays, pyrt.ACCEL_BVH).view(np.uint8), orc.trace(s, rays, orc.ACCEL_OBVH).view(np.uint8))
    ctx.close()


@pytest.mark.parametrize("kind", ["lowres", "hires"])
def test_device_tree_structure_and_determinism(kind):
    s = pyrt.Scene(kind, 64, 64)
    ctx = pyrt.Context(s, bvh_builder=pyrt.BVH_DEVICE)
    bi = ctx.bvh_info()
    nodes, tris = ctx.bvh_export()
    host = pyrt.Context(s)
    hb = host.bvh_info()
    assert bi.n_tri_records == s.desc.n_triangles and bi.leaf_max == 2 and bi.pad == hb.pad
    assert bi.max_depth <= hb.max_depth + 1 and bi.max_depth < 32  # the same depth cap binds both builders
    ids = tris[:, 9]
    assert sorted(ids.tolist()) == list(range(s.desc.n_triangles))
    seen = np.zeros(bi.n_tri_records, bool)
    lo = nodes[:, [0, 1, 2, 6, 7, 8]].view(np.float32).reshape(-1, 2, 3)
    hi = nodes[:, [3, 4, 5, 9, 10, 11]].view(np.float32).reshape(-1, 2, 3)
    tf = tris.view(np.float32)
    reached = np.zeros(bi.n_nodes, bool)
    reached[0] = True
    for ni in range(bi.n_nodes):
        for c in range(2):
            ch = int(np.int32(nodes[ni, 12 + c]))
            if ch < 0:
                code = (~ch) & 0xFFFFFFFF
                first, cnt = code >> 3, (code & 7) + 1
                assert cnt <= bi.leaf_max and not seen[first:first + cnt].any()
                seen[first:first + cnt] = True
                p0 = tf[first:first + cnt, 0:3]
                verts = np.stack([p0, p0 + tf[first:first + cnt, 3:6], p0 + tf[first:first + cnt, 6:9]], 1)
                assert (verts >= lo[ni, c] - 1e-6).all() and (verts <= hi[ni, c] + 1e-6).all()
            else:
                assert ni < ch < bi.n_nodes and not reached[ch]  # breadth-first numbering: children after parents
                reached[ch] = True
                assert (lo[ch] >= lo[ni, c] - 1e-6).all() and (hi[ch] <= hi[ni, c] + 1e-6).all()
    assert seen.all() and reached.all()
    again = pyrt.Context(s, bvh_builder=pyrt.BVH_DEVICE)
    n2, t2 = again.bvh_export()
    assert np.array_equal(n2, nodes) and np.array_equal(t2, tris)  # same arrays on every build
    for c in (ctx, host, again):
        c.close()


@pytest.mark.parametrize("kind,w,h,spp", [("lowres", 64, 48, 6), ("hires", 48, 48, 4), ("stress", 32, 32, 2)])
def test_frames_on_device_built_tree_vs_oracle(kind, w, h, spp):
    s = pyrt.Scene(kind, w, h)
    ctx = pyrt.Context(s, bvh_builder=pyrt.BVH_DEVICE)
    p = pyrt.make_params(w, h, spp, seed=31)
    _, acc, st = ctx.render(p)
    _, ref, rst = orc.render(s, p, math_mode=orc.MATH_DET, accel=orc.ACCEL_OBVH)
    assert np.array_equal(bits(acc), bits(ref)) and (st.rays_closest, st.rays_shadow) == (rst.rays_closest, rst.rays_shadow)
    ctx.close()


@pytest.mark.parametrize("slack", [None, "0", "2"])
@pytest.mark.parametrize("kind", ["lowres", "hires", "stress"])
def test_all_three_builders_build_the_same_tree(kind, slack, monkeypatch):
    """The claim of csrc/bvh_gpu.hip: the device builder (the host builder's split rules as kernels, exact subtrees with the
    host's four sweep axes and id tie-breaks, the host's rotation passes as kernels on the host's slot order) and the hybrid
    builder (the same below the host's own top) produce THE HOST BUILDER'S TREE — the same boxes over the same triangle sets
    all the way down, whatever the node numbering and the child slots (tests/treedigest.py).  Also with the depth budget
    binding (RT_BVH_SLACK = 0: a balanced tree, the builders' median splits everywhere; 2: the budget of the biggest scenes),
    where the depth checks of the splits and of the rotations decide."""
    from treedigest import context_digest
    if slack is not None:
        monkeypatch.setenv("RT_BVH_SLACK", slack)
    s = pyrt.Scene(kind, 32, 32)
    got = {}
    for name, b in (("host", pyrt.BVH_HOST), ("device", pyrt.BVH_DEVICE), ("hybrid", pyrt.BVH_HYBRID)):
        ctx = pyrt.Context(s, bvh_builder=b)
        got[name] = (context_digest(ctx), ctx.bvh_info().n_nodes, ctx.bvh_info().max_depth)
        ctx.close()
    if not os.environ.get("RT_BVH_GPU"):  # (the variable forces one builder for all three)
        assert got["device"] == got["host"] and got["hybrid"] == got["host"], got


def test_device_builder_variants_are_exact(tmp_path):
    """The builders' other configurations (environment knobs, read once per process): the host builder's variants, the
    hybrid builder, the device builder without its rotation passes / final numbering.  Every one must give the exhaustive
    loop's hits and the default tree's frame."""
    import subprocess
    import sys
    script = tmp_path / "v.py"
    script.write_text('''
import os, sys, numpy as np
sys.path.insert(0, sys.argv[1] + "/ray-tracing-engine_amd"); sys.path.insert(0, sys.argv[1] + "/tests")
import pyrt
from raybatch import ray_batch
out = {}
for kind, w, spp, n in (("lowres", 64, 4, 60000), ("hires", 48, 3, 60000), ("stress", 32, 2, 30000)):
    s = pyrt.Scene(kind, w, w); ctx = pyrt.Context(s, bvh_builder=pyrt.BVH_HOST if os.environ.get("RT_TEST_HOST_BUILDER") else pyrt.BVH_DEVICE)
    rays = ray_batch(s, n, 11)
    got, want = ctx.trace(rays, pyrt.ACCEL_BVH), ctx.trace(rays, pyrt.ACCEL_BRUTE)
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), kind
    assert np.array_equal(ctx.trace(rays, pyrt.ACCEL_BVH, pyrt.TRACE_ANY)["hit"], want["hit"]), kind
    _, acc, st = ctx.render(pyrt.make_params(w, w, spp, seed=9))
    bi = ctx.bvh_info()
    out[kind] = acc; out[kind + "_n"] = np.array([st.rays_closest, st.rays_shadow, bi.n_nodes, bi.max_depth])
    ctx.close()
np.savez(sys.argv[2], **out)
''')
    runs = {}
    for name, env in (("default", {}), ("no_rotations", {"RT_BVH_GPU_ROT": "0"}), ("breadth_first_numbering", {"RT_BVH_GPU_PREORDER": "0"}),
                      # ... and the host builder's: its default, without the size axis, with an unbiased size axis,
                      # without the rotation passes
                      ("host", {"RT_TEST_HOST_BUILDER": "1"}), ("host_no_size_axis", {"RT_TEST_HOST_BUILDER": "1", "RT_BVH_SIZEAXIS": "0"}),
                      ("host_size_axis_unbiased", {"RT_TEST_HOST_BUILDER": "1", "RT_BVH_SIZEBIAS": "1"}),
                      ("host_no_rotations", {"RT_TEST_HOST_BUILDER": "1", "RT_BVH_ROT": "0"}),
                      # ... and the hybrid builder (RT_BVH_GPU=2 overrides the option): the host's top, exact subtrees on the device
                      ("hybrid", {"RT_BVH_GPU": "2"})):
        out = tmp_path / (name + ".npz")
        r = subprocess.run([sys.executable, str(script), pyrt.ROOT, str(out)], env=dict(os.environ, **env), capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        runs[name] = np.load(out)
    for name in runs:
        for kind in ("lowres", "hires", "stress"):
            assert np.array_equal(bits(runs[name][kind]), bits(runs["default"][kind])), (name, kind)
            assert np.array_equal(runs[name][kind + "_n"][:2], runs["default"][kind + "_n"][:2]), (name, kind)
            assert runs[name][kind + "_n"][3] < 32


def test_build_time_and_tree_quality_report(capsys):
    """The device builder restates the host builder's splits, so it must produce the host builder's TREE (the digests of
    test_all_three_builders_build_the_same_tree): the same node count and depth and — on a counted 256x256x4 frame — the same
    node visits and triangle tests per ray, and it must build the 1 M-triangle scene several times sooner (22 ms against
    160-180)."""
    rows = []
    for kind, n in (("lowres", 200000), ("hires", 200000), ("stress", 200000)):
        s = pyrt.Scene(kind, 256, 256)
        out = {}
        for name, b in (("host", pyrt.BVH_HOST), ("device", pyrt.BVH_DEVICE), ("hybrid", pyrt.BVH_HYBRID)):
            ctx = pyrt.Context(s, bvh_builder=b)
            bi = ctx.bvh_info()
            p = pyrt.make_params(256, 256, 4, seed=2, collect_stats=1)
            _, _, st = ctx.render(p, want_accum=False)
            rays = st.rays_closest + st.rays_shadow
            out[name] = (bi.build_ms, bi.n_nodes, bi.max_depth, st.nodes_visited / rays, st.tris_tested / rays, st.kernel_ms)
            ctx.close()
        rows.append((kind, out))
    with capsys.disabled():
        for kind, out in rows:
            print("\n%-7s host: build %8.1f ms nodes %7d depth %2d  %.2f nodes/ray %.2f tris/ray kernel %.2f ms | "
                  "device: build %7.1f ms nodes %7d depth %2d  %.2f nodes/ray %.2f tris/ray kernel %.2f ms | "
                  "hybrid: build %7.1f ms nodes %7d depth %2d  %.2f nodes/ray %.2f tris/ray kernel %.2f ms"
                  % ((kind,) + out["host"] + out["device"] + out["hybrid"]), end="")
        print()
    for kind, out in rows:
        for name in ("device", "hybrid"):
            assert out[name][1:3] == out["host"][1:3], (kind, name, out)  # nodes, depth
            # visits and tests per ray: equal — except on the one-request records (RT_NODES=q8 forced), whose 8-bit planes are
            # quantised in the frames of 16-KiB blocks of the node ARRAY, so they depend on the numbering too
            tol = 1e-2 if os.environ.get("RT_NODES") else 0.0
            assert abs(out[name][3] / out["host"][3] - 1) <= tol and abs(out[name][4] / out["host"][4] - 1) <= tol, (kind, name, out)
    stress = dict(rows)["stress"]
    if not os.environ.get("RT_BVH_GPU") and not os.environ.get("RT_NODES"):  # (forced builders / the host-side Q8 packing in every build)
        assert 3 * stress["device"][0] < stress["host"][0] and stress["device"][0] < stress["hybrid"][0] < stress["host"][0]
