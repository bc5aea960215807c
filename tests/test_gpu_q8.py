"""The ONE-REQUEST node records on the GPU (rt_options.node_format = RT_NODES_Q8: rtbvh::Slot16 — 16-byte records with 8-bit
box planes in the frame of their 16-KiB block, nodes and triangle records in one array; rt_kernels.hip Trav with LT_Q8).
Held to the bar of the 32-byte binary16 records: RayTracer::rayTrace (reference source/RayTracer.h:27-53: closest positive t,
strict '<', lowest (mesh, triangle) on ties) bit for bit against the exhaustive GPU loop and the CPU oracle, closest and any
hit, on all four scenes, on scaled scenes, from far origins, on device-built trees; frames bit-identical to the oracle's and to
the frames the 32-byte records give, with the same ray counts."""
import os

import numpy as np
import pytest

import orc
import pyrt
from raybatch import ray_batch

pytestmark = pytest.mark.gpu
# (RT_NODES / RT_BVH_GPU override the options of every context — the whole suite is run under each form, tools/r4_suite_forms.sh;
# the tests that hold two forms against each other need both)
two_forms = pytest.mark.skipif(bool(os.environ.get("RT_NODES")), reason="RT_NODES forces one node format on every context")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("kind,n,n_orc", [("cubes", 200000, 20000), ("lowres", 200000, 20000), ("hires", 300000, 20000), ("stress", 200000, 3000)])
def test_q8_hits_equal_the_exhaustive_loop_and_the_oracle(kind, n, n_orc):
    s = pyrt.Scene(kind, 256, 256)
    ctx = pyrt.Context(s, node_format=pyrt.NODES_Q8)
    assert ctx.bvh_info().node_format == pyrt.NODES_Q8
    rays = ray_batch(s, n, 77)
    want = ctx.trace(rays, pyrt.ACCEL_BRUTE)
    got = ctx.trace(rays, pyrt.ACCEL_BVH)
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), kind
    assert np.array_equal(ctx.trace(rays, pyrt.ACCEL_BVH, pyrt.TRACE_ANY)["hit"], want["hit"]), kind
    assert 0.3 < want["hit"].mean() < 1.0
    ref = orc.trace(s, rays[:n_orc])  # the restated reference loop itself
    assert np.array_equal(got[:n_orc].view(np.uint8), ref.view(np.uint8)), kind
    ctx.close()


@two_forms
@pytest.mark.parametrize("kind,w,h,spp,mode", [("cubes", 64, 64, 4, 1), ("lowres", 48, 40, 6, 1), ("hires", 96, 64, 4, 1), ("stress", 64, 48, 3, 1),
                                              ("hires", 48, 48, 16, 0), ("stress", 40, 40, 2, 0)])
def test_q8_frames_equal_the_oracle_and_the_f16_frames(kind, w, h, spp, mode):
    s = pyrt.Scene(kind, w, h)
    bg = pyrt.background(w, h)
    p = pyrt.make_params(w, h, spp, mode=mode, seed=29, collect_stats=1)
    ref_out, ref_acc, ref_st = orc.render(s, p, math_mode=orc.MATH_DET, bg=bg, accel=orc.ACCEL_OBVH)
    frames = {}
    for fmt in (pyrt.NODES_F16, pyrt.NODES_Q8):
        ctx = pyrt.Context(s, node_format=fmt)
        out, acc, st = ctx.render(p, bg)
        assert np.array_equal(bits(acc), bits(ref_acc)) and np.array_equal(bits(out), bits(ref_out)), (kind, fmt)
        assert (st.rays_closest, st.rays_shadow) == (ref_st.rays_closest, ref_st.rays_shadow)
        frames[fmt] = st
        # the uncounted (timed) instance, other samples per wave, the sequential shading: the same frame
        for kw in (dict(), dict(lanes_per_pixel=1), dict(lanes_per_pixel=64), dict(no_pool=True)):
            _, acc2, st2 = ctx.render(pyrt.make_params(w, h, spp, mode=mode, seed=29, **kw))
            assert np.array_equal(bits(acc2), bits(ref_acc)), (kind, fmt, kw)
            assert (st2.rays_closest, st2.rays_shadow) == (ref_st.rays_closest, ref_st.rays_shadow)
        ctx.close()
    q, f = frames[pyrt.NODES_Q8], frames[pyrt.NODES_F16]
    assert q.frame_fetches > 0 and f.frame_fetches == 0
    # coarser boxes cost visits, but not many; and most visits stay inside the block whose frame the lane holds
    assert q.nodes_visited <= 1.35 * f.nodes_visited and q.tris_tested <= 1.5 * f.tris_tested
    assert q.frame_fetches < q.nodes_visited


@pytest.mark.parametrize("scale", [1e-3, 1.0, 37.0, 1e4, 1e10])
def test_q8_on_scaled_scenes_and_from_far_origins(scale):
    """The frames' grid follows the scene's scale (the step is 1 / 255 of a block's extent), so the exactness argument
    must hold at 1e-3 ... 1e10 scene units (hits AND frames, either side of the short-reciprocal bound); rays starting far
    outside the scene take the exhaustive loop beyond the padding reference, as with the 32-byte records."""
    a = pyrt.Scene("lowres", 64, 64).arrays()
    f = np.float32(scale)
    lights = a["lights"].copy()
    lights[:, 0:3] *= f  # (rt_light: position[3] ... intensity, side at word 16)
    lights[:, 16] *= f
    sc = pyrt.ArrayScene(a["pos"] * f, a["nrm"], a["tri"], a["tri_begin"], a["vtx_begin"], a["materials"], lights, a["camera"] * f)
    ctx = pyrt.Context(sc, node_format=pyrt.NODES_Q8)
    rays = ray_batch(sc, 60000, 5)
    rays["origin"][len(rays) // 2 + len(rays) // 16:] = a["camera"][0] * f
    want = ctx.trace(rays, pyrt.ACCEL_BRUTE)
    got = ctx.trace(rays, pyrt.ACCEL_BVH)
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), scale
    assert np.array_equal(ctx.trace(rays, pyrt.ACCEL_BVH, pyrt.TRACE_ANY)["hit"], want["hit"]), scale
    assert 0.2 < want["hit"].mean()
    far = rays.copy()
    far["origin"][::2] += np.float32(3000.0) * f
    assert np.array_equal(ctx.trace(far, pyrt.ACCEL_BVH).view(np.uint8), ctx.trace(far, pyrt.ACCEL_BRUTE).view(np.uint8)), scale
    p = pyrt.make_params(32, 32, 3, seed=13)
    _, acc, st = ctx.render(p)
    _, ref, rst = orc.render(sc, p, math_mode=orc.MATH_DET)
    assert np.array_equal(bits(acc), bits(ref)) and (st.rays_closest, st.rays_shadow) == (rst.rays_closest, rst.rays_shadow)
    ctx.close()


@pytest.mark.parametrize("kind", ["lowres", "stress"])
def test_q8_over_a_device_built_tree(kind):
    s = pyrt.Scene(kind, 64, 64)
    ctx = pyrt.Context(s, bvh_builder=pyrt.BVH_DEVICE, node_format=pyrt.NODES_Q8)
    bi = ctx.bvh_info()
    assert bi.builder == int(os.environ.get("RT_BVH_GPU", pyrt.BVH_DEVICE)) and bi.node_format == pyrt.NODES_Q8
    rays = ray_batch(s, 100000, 3)
    want = ctx.trace(rays, pyrt.ACCEL_BRUTE)
    assert np.array_equal(ctx.trace(rays, pyrt.ACCEL_BVH).view(np.uint8), want.view(np.uint8))
    p = pyrt.make_params(64, 64, 3, seed=5)
    _, acc, _ = ctx.render(p)
    host = pyrt.Context(s)
    _, acc0, _ = host.render(p)
    assert np.array_equal(bits(acc), bits(acc0))
    ctx.close(), host.close()


@two_forms
def test_q8_refuses_what_it_cannot_express():
    s = pyrt.Scene("lowres", 32, 32)
    with pytest.raises(pyrt.RtError):
        pyrt.Context(s, bvh_leaf_max=4, node_format=pyrt.NODES_Q8)  # one count bit per child: leaves of 1 or 2 triangles
    with pytest.raises(pyrt.RtError):
        pyrt.Context(s, node_format=7)


def test_q8_at_full_size_on_the_stress_scene():
    """BASELINE config 5 (1 M triangles, 1024 x 1024) at 16 of its 256 spp: the frame of the one-request records is the frame
    of the 32-byte records, bit for bit, with the same ray counts, and a 2-way tile split of it adds up to it."""
    w = h = 1024
    spp = 16
    s = pyrt.Scene("stress", w, h)
    f16, q8 = pyrt.Context(s, node_format=pyrt.NODES_F16), pyrt.Context(s, node_format=pyrt.NODES_Q8)
    p = pyrt.make_params(w, h, spp, seed=1)
    _, a, sa = f16.render(p)
    _, b, sb = q8.render(p)
    assert np.array_equal(bits(a), bits(b)) and (sa.rays_closest, sa.rays_shadow) == (sb.rays_closest, sb.rays_shadow)
    acc = None
    for rank in range(2):
        _, part, _ = q8.render(pyrt.make_params(w, h, spp, seed=1, rank=rank, world=2, tile=32))
        acc = part if acc is None else acc + part
    assert np.array_equal(bits(acc), bits(a))
    f16.close(), q8.close()
