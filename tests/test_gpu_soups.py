"""Builders under geometry they were not tuned on.  rt_create picks the device builder by itself from 8,192 triangles
(RT_BVH_AUTO), so it has to survive what a caller may hand it: coincident triangles (no split separates anything), collinear
centroids with sizes six orders of magnitude apart, a dense cluster beside scene-sized triangles, a flat sheet, a random soup
(tests/soups.py).  Every builder and both node formats must return the exhaustive loop's hits — which are the oracle's."""
import numpy as np
import pytest

import orc
import pyrt
import soups

pytestmark = pytest.mark.gpu

N = 140000


@pytest.mark.parametrize("n", [N, 9000, 3000])  # (swept and binned ranges at the top; just above RT_BVH_AUTO's threshold; swept only)
@pytest.mark.parametrize("kind", soups.KINDS)
def test_every_builder_is_exact_on_hard_soups(kind, n):
    from treedigest import context_digest
    s = soups.soup(kind, n)
    rays = soups.soup_rays(s, 3000)
    want = None
    digests = {}
    for builder, fmt in ((pyrt.BVH_AUTO, pyrt.NODES_AUTO), (pyrt.BVH_HOST, pyrt.NODES_F16), (pyrt.BVH_DEVICE, pyrt.NODES_F16),
                         (pyrt.BVH_HYBRID, pyrt.NODES_Q8), (pyrt.BVH_DEVICE, pyrt.NODES_Q8)):
        ctx = pyrt.Context(s, bvh_builder=builder, node_format=fmt)
        bi = ctx.bvh_info()
        assert bi.n_tri_records == n and bi.max_depth < 32
        if want is None:
            want = ctx.trace(rays, pyrt.ACCEL_BRUTE)
            assert want["hit"].sum() > len(rays) // 4  # (the rays are aimed at triangles)
            # ... and the exhaustive loop is the oracle's (a sample: up to 140 k tests per ray on one host core)
            ref = orc.trace(s, rays[:200])
            assert np.array_equal(want[:200].view(np.uint8), ref.view(np.uint8))
        got = ctx.trace(rays, pyrt.ACCEL_BVH)
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (kind, builder, fmt)
        assert np.array_equal(ctx.trace(rays, pyrt.ACCEL_BVH, pyrt.TRACE_ANY)["hit"], want["hit"]), (kind, builder, fmt)
        if fmt == pyrt.NODES_F16 and builder in (pyrt.BVH_HOST, pyrt.BVH_DEVICE):
            digests[builder] = (context_digest(ctx), bi.n_nodes, bi.max_depth)
        ctx.close()
    # ... and the device builder builds the host builder's tree on these too (coincident triangles: median splits by id all the
    # way down, the binned ones through the radix selection of bvh_gpu.hip)
    import os
    if not os.environ.get("RT_BVH_GPU"):
        assert digests[pyrt.BVH_DEVICE] == digests[pyrt.BVH_HOST], (kind, n, digests)


@pytest.mark.parametrize("kind", ["corner", "random"])
def test_frames_on_hard_soups_agree_across_builders(kind):
    """One small frame per builder (pooled kernel, the stacks sized by each tree's own depth): the same accumulators."""
    s = soups.soup(kind, N)
    p = pyrt.make_params(32, 32, 2, seed=3)
    frames = []
    for builder in (pyrt.BVH_AUTO, pyrt.BVH_HOST, pyrt.BVH_DEVICE):
        ctx = pyrt.Context(s, bvh_builder=builder)
        _, acc, st = ctx.render(p)
        frames.append((acc.copy(), st.rays_closest, st.rays_shadow))
        ctx.close()
    for f in frames[1:]:
        assert np.array_equal(f[0].view(np.uint32), frames[0][0].view(np.uint32)) and f[1:] == frames[0][1:]


@pytest.mark.parametrize("leaf", [1, 3, 8])
def test_device_builder_with_other_leaf_sizes(leaf):
    """rt_options.bvh_leaf_max 1 ... 8 through the device builder (parts, top leaves and the depth budget all depend on it):
    exact, and the same tree as the host builder's."""
    from treedigest import context_digest
    import os
    if leaf > 2 and os.environ.get("RT_NODES") == "q8":
        pytest.skip("the one-request records hold leaves of at most 2 triangles")
    for s in (pyrt.Scene("hires", 16, 16), soups.soup("random", 20000)):
        rays = soups.soup_rays(s, 2000) if isinstance(s, pyrt.ArrayScene) else None
        got = {}
        for builder in (pyrt.BVH_HOST, pyrt.BVH_DEVICE):
            ctx = pyrt.Context(s, bvh_leaf_max=leaf, bvh_builder=builder)
            bi = ctx.bvh_info()
            assert bi.leaf_max == leaf and bi.max_depth < 32
            if rays is not None:
                want = ctx.trace(rays, pyrt.ACCEL_BRUTE)
                assert np.array_equal(ctx.trace(rays, pyrt.ACCEL_BVH).view(np.uint8), want.view(np.uint8)), (leaf, builder)
            got[builder] = (context_digest(ctx), bi.n_nodes, bi.max_depth)
            ctx.close()
        import os
        if not os.environ.get("RT_BVH_GPU"):
            assert got[pyrt.BVH_DEVICE] == got[pyrt.BVH_HOST], (leaf, got)


def test_device_builder_rejects_what_the_host_builder_rejects():
    """Scenes big enough for RT_BVH_AUTO's device builder are validated by rtbvh::planSceneExact (the host builder's checks, on
    several threads): a non-finite vertex or a triangle pointing outside its mesh fails rt_create with the same errors."""
    s = soups.soup("random", 20000)
    a = s.arrays()
    bad = a["pos"].copy()
    bad[31337, 2] = np.inf
    with pytest.raises(pyrt.RtError) as e:
        pyrt.Context(pyrt.ArrayScene(bad, a["nrm"], a["tri"], a["tri_begin"], a["vtx_begin"], a["materials"], a["lights"], a["camera"]))
    assert e.value.code == 1 and "non-finite" in str(e.value)
    tri = a["tri"].copy()
    tri[19999, 1] = 60000
    with pytest.raises(pyrt.RtError) as e:
        pyrt.Context(pyrt.ArrayScene(a["pos"], a["nrm"], tri, a["tri_begin"], a["vtx_begin"], a["materials"], a["lights"], a["camera"]))
    assert e.value.code == 1 and "outside its mesh" in str(e.value)
