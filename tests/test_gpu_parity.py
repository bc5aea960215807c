"""Parity tests proper: the HIP path, called through the C ABI, against the CPU
oracle (oracle/liboracle.so) on the same seeded inputs and against the golden
vectors produced by the real reference.  Bit-exact unless a tolerance is stated."""
import ctypes as C

import os

import numpy as np
import pytest

import orc
import pyrt

pytestmark = pytest.mark.gpu


def f32(u):
    return np.array(u, dtype=np.uint32).view(np.float32)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def cubes():
    s = pyrt.Scene("cubes", 256, 256)
    return s, pyrt.Context(s)


@pytest.fixture(scope="module")
def lowres():
    s = pyrt.Scene("lowres", 256, 256)
    return s, pyrt.Context(s)


# ------------------------------------------------------------------ building blocks
def test_detmath_device_equals_host():
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(0, 1.0000000279, 200000), [0.0, 1.0, 0.5, 0.975, 1.0000000278, 1e-9, 0.4999999]])
    got = pyrt.unit(pyrt.UNIT_ASIN, x)[:, 0]
    ref = np.array([orc.lib().orc_det_asin(float(v)) for v in x[:20000]] )
    assert np.array_equal(got[:20000].view(np.uint64), ref.view(np.uint64))
    # nan for |x|>1 on both
    assert np.isnan(got[-3]) and np.isnan(orc.lib().orc_det_asin(1.0000000278))
    a = np.concatenate([rng.uniform(0, 6.2831860, 200000), [0.0, np.pi / 2, np.pi, 3 * np.pi / 2, 2 * np.pi]]).astype(np.float32)
    for which, fn in ((pyrt.UNIT_SINF, orc.lib().orc_det_sinf), (pyrt.UNIT_COSF, orc.lib().orc_det_cosf)):
        got = pyrt.unit(which, a)[:, 0]
        ref = np.array([fn(float(v)) for v in a[:20000]], np.float32)
        assert np.array_equal(bits(got[:20000]), bits(ref))
        # and close to libm (not bit-equal: see tests/test_oracle_images.py)
        lib = np.sin(a.astype(np.float64)) if which == pyrt.UNIT_SINF else np.cos(a.astype(np.float64))
        assert np.abs(got - lib).max() < 1.2e-7
    k = rng.integers(0, 2**32, (5000, 4), dtype=np.uint64).astype(np.uint32)
    k[:, 1] &= 1
    got = pyrt.unit(pyrt.UNIT_STREAM_SEED, k)[:, 0]
    ref = np.array([orc.lib().orc_stream_seed(*map(int, r)) for r in k], np.uint32)
    assert np.array_equal(got, ref) and got.min() >= 1 and got.max() <= 2147483646


def test_triangle_intersect_golden(golden):
    v = np.array(golden["vectors"]["triangleIntersect"], np.uint32).reshape(-1, 19)
    init = np.full((len(v), 4), -7.0, np.float32)
    out = pyrt.unit(pyrt.UNIT_TRIANGLE, f32(v[:, :15]), init)
    assert np.array_equal(out[:, 0].astype(np.uint32), v[:, 15])
    assert np.array_equal(bits(out[:, 1:4]), v[:, 16:19])


def test_bsdf_golden(golden):
    """The golden values were computed by the reference with libm pow(); the device
    uses x*x and (x*x)*(x*x)*x (rt_pixelmode.h).  Both are within 1 ulp in double,
    which survives the narrowing to float with probability ~1e-8: expect bit
    equality on all 300 vectors."""
    v = np.array(golden["vectors"]["bsdf"], np.uint32).reshape(-1, 20)
    inp = np.concatenate([v[:, 9:17], v[:, 0:9]], axis=1)  # kd alpha albedo f0 | n wi wo
    out = pyrt.unit(pyrt.UNIT_BSDF, f32(inp))
    assert np.array_equal(bits(out), v[:, 17:20])


def test_ray_at_and_light_golden(golden, cubes):
    s, _ = cubes
    V = golden["vectors"]
    v = np.array(V["rayAt"], np.uint32).reshape(-1, 8)
    cam = np.frombuffer(bytes(s.desc.camera), np.uint32)
    inp = np.concatenate([np.tile(cam, (len(v), 1)), v[:, :2]], axis=1)
    out = pyrt.unit(pyrt.UNIT_RAY_AT, f32(inp))
    assert np.array_equal(bits(out), v[:, 2:8])
    lv = np.array(V["evaluateLight"], np.uint32).reshape(3, 20, 6)
    lights = np.frombuffer(C.string_at(s.desc.lights, 84 * 3), np.uint32).reshape(3, 21)
    for li in range(3):
        inp = np.concatenate([np.tile(lights[li], (20, 1)), lv[li, :, :3]], axis=1)
        out = pyrt.unit(pyrt.UNIT_LIGHT_EVAL, f32(inp))
        assert np.array_equal(bits(out), lv[li, :, 3:6])


def test_samplers_against_oracle(cubes):
    s, _ = cubes
    rng = np.random.default_rng(11)
    n = 4000
    lights = np.frombuffer(C.string_at(s.desc.lights, 84 * 3), np.uint32).reshape(3, 21)
    inp = np.zeros((n, 28), np.uint32)
    inp[:, 0] = rng.integers(1, 2147483646, n)
    N = rng.choice([1, 2, 3, 4, 8, 16, 128], n)
    inp[:, 2] = N
    inp[:, 1] = rng.integers(0, 1 << 30, n) % N
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    nrm[::7] = [0, 1, 0]
    nrm[1::7] = [0, 0, -1]
    inp[:, 4:7] = bits(nrm)
    li = rng.integers(0, 3, n)
    inp[:, 7:28] = lights[li]
    out = pyrt.unit(pyrt.UNIT_SAMPLERS, inp)
    xy, hs, ls = np.zeros(2, np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32)
    L = orc.lib()
    for i in range(n):
        st = C.c_uint32(int(inp[i, 0]))
        L.orc_jitter(C.byref(st), int(inp[i, 1]), int(inp[i, 2]), orc._p(xy))
        nn = nrm[i].copy()
        L.orc_hsphere(C.byref(st), orc.MATH_DET, orc._p(nn), orc._p(hs))
        L.orc_rand_area(C.byref(st), C.byref(s.desc.lights[int(li[i])]), orc._p(ls))
        assert np.array_equal(out[i, 0:2], bits(xy)), i
        assert np.array_equal(out[i, 2:5], bits(hs)), i
        assert np.array_equal(out[i, 5:8], bits(ls)), i
        assert out[i, 8] == st.value


def test_jitter_and_light_sample_golden(golden, cubes):
    """Sequences drawn from a fresh seed-1 engine by the reference itself."""
    s, _ = cubes
    V = golden["vectors"]
    lights = np.frombuffer(C.string_at(s.desc.lights, 84 * 3), np.uint32).reshape(3, 21)
    state = 1
    for N, i, bx, by in np.array(V["jitterSample_seq"], np.uint32).reshape(-1, 4):
        inp = np.zeros((1, 28), np.uint32)
        inp[0, 0], inp[0, 1], inp[0, 2] = state, i, N
        inp[0, 4:7] = bits(np.array([0, 1, 0], np.float32))
        inp[0, 7:28] = lights[0]
        out = pyrt.unit(pyrt.UNIT_SAMPLERS, inp)[0]
        assert (out[0], out[1]) == (bx, by)
        st = C.c_uint32(state)  # advance by the 4 engine calls of jitterSample only
        for _ in range(4):
            orc.lib().orc_engine_next(C.byref(st))
        state = st.value
    # LightSource.h:46-49 randAreaPosition: 4 rounds over the 3 lights from a fresh seed-1
    # engine, each call continuing the engine where the previous one left it
    state = 1
    for j, row in enumerate(np.array(V["randAreaPosition_seq"], np.uint32).reshape(-1, 3)):
        inp = np.zeros((1, 22), np.uint32)
        inp[0, 0], inp[0, 1:22] = state, lights[j % 3]
        out = pyrt.unit(pyrt.UNIT_LIGHT_SAMPLE, inp)[0]
        assert np.array_equal(out[0:3], row), j
        state = int(out[3])
    st = C.c_uint32(1)
    for _ in range(2 * 12):  # 2 float draws per call = 1 engine call each
        orc.lib().orc_engine_next(C.byref(st))
    assert state == st.value


def test_detmath_ulp_bounds_vs_libm():
    """include/rt_pixelmode.h is shared by the oracle and the device, so GPU-vs-oracle
    parity cannot see a defect in it: bound every pinned function against libm (numpy,
    float64 reference) on the GPU.  Stated tolerances: asin within 4 ulp(double) and its
    float narrowing within 1 ulp(float); pow2 exact, pow5 = (x*x)*(x*x)*x within 3
    ulp(double) (three roundings), both equal to libm after the narrowing to float the
    reference performs (sin/cos: 1.2e-7 absolute, test above)."""
    rng = np.random.default_rng(21)
    x = np.concatenate([rng.uniform(0, 1, 300000), 1 - 10.0 ** rng.uniform(-12, -1, 50000), 10.0 ** rng.uniform(-12, -1, 50000),
                        [0.0, 1.0, 0.5]])
    got = pyrt.unit(pyrt.UNIT_ASIN, x)[:, 0]
    ref = np.arcsin(x)
    # double result: relative error below 4 ulp(double); after the narrowing the reference does
    # (RayTracer.h:102 `float theta = asin(..)`) at most one float ulp, and equal for > 99.9999 %
    assert np.abs(got - ref).max() <= 4 * np.spacing(ref).max()
    gf, rf = got.astype(np.float32), ref.astype(np.float32)
    assert (np.abs(gf.view(np.int32).astype(np.int64) - rf.view(np.int32).astype(np.int64)) <= 1).all()
    assert (gf != rf).mean() < 1e-6
    y = np.concatenate([rng.uniform(0, 1, 300000), rng.uniform(1, 2, 50000), [0.0, 1.0, 2.0]])
    pw = pyrt.unit(pyrt.UNIT_POW, y)
    for col, e, ulps in ((0, 2.0, 0), (1, 5.0, 3)):
        r = np.power(y, e)
        assert (np.abs(pw[:, col] - r) <= ulps * np.spacing(r)).all()
        assert (pw[:, col].astype(np.float32) != r.astype(np.float32)).mean() < 1e-6


def test_short_reciprocal_and_square_root_are_the_ieee_ones_bit_for_bit():
    """Ray.cpp:14 `inv_det = 1.0f / det`: the default render instances compute it as v_rcp_f32 + one Newton step
    (rt_device.h recip_fast).  tools/microbench/recip_exact.hip checked ALL 2^32 inputs on gfx950 (0 mismatches for
    2^-100 <= |x| < 2^101, profiles/r03_recip_exact.json); this is the regression guard: 4 M samples of that range — every
    binade, the mantissa extremes, both signs — against the GPU's own division and against numpy's (IEEE) float32 one."""
    rng = np.random.default_rng(5)
    e = rng.integers(-100, 100, 4_000_000)
    m = rng.integers(0, 1 << 23, 4_000_000).astype(np.uint32)
    m[:2000] = 0
    m[2000:4000] = (1 << 23) - 1
    m[4000:6000] = (1 << 22)
    x = (((e + 127).astype(np.uint32) << 23) | m).view(np.float32)
    x[::2] *= np.float32(-1)
    x = np.concatenate([x, np.float32([1.0, -1.0, 3.0, 1e-6, 2.0 ** -100, 2.0 ** 100, 0.9999999, 1.0000001])]).astype(np.float32)
    got = pyrt.unit(pyrt.UNIT_RECIP, x)
    with np.errstate(all="ignore"):
        want = (np.float32(1.0) / x).astype(np.float32)
    assert np.array_equal(got[:, 1].view(np.uint32), want.view(np.uint32))  # the GPU's division is IEEE's
    assert np.array_equal(got[:, 0].view(np.uint32), want.view(np.uint32))  # ... and the short form is the division
    # the same for the five-instruction square root (rt_device.h sqrt_fast: the lengths of the normalisations)
    pos = np.abs(x)
    gs = pyrt.unit(pyrt.UNIT_RECIP, pos)
    ws = np.sqrt(pos.astype(np.float32)).astype(np.float32)  # (numpy's float32 sqrt is correctly rounded)
    assert np.array_equal(gs[:, 3].view(np.uint32), ws.view(np.uint32))
    assert np.array_equal(gs[:, 2].view(np.uint32), ws.view(np.uint32))


# ------------------------------------------------------------------ closest hit / any hit
from raybatch import ray_batch as _ray_batch  # noqa: E402


@pytest.mark.parametrize("which", ["cubes", "lowres"])
def test_trace_golden(golden, which, cubes, lowres):
    s, ctx = cubes if which == "cubes" else lowres
    v = np.array(golden["vectors"]["rayTrace_" + which], np.uint32).reshape(-1, 14)
    rays = np.zeros(len(v), pyrt.RAY_DTYPE)
    rays["origin"], rays["direction"] = f32(v[:, 0:3]), f32(v[:, 3:6])
    for accel in (pyrt.ACCEL_BRUTE, pyrt.ACCEL_BVH):
        h = ctx.trace(rays, accel)
        assert np.array_equal(h["hit"].astype(np.uint32), v[:, 6])
        m = v[:, 6] == 1
        assert np.array_equal(h["mesh"][m], v[m, 7]) and np.array_equal(h["vtx"][m], v[m, 8:11])
        assert np.array_equal(bits(h["u"][m]), v[m, 11]) and np.array_equal(bits(h["v"][m]), v[m, 12])
        assert np.array_equal(bits(h["d"][m]), v[m, 13])
        a = ctx.trace(rays, accel, pyrt.TRACE_ANY)
        assert np.array_equal(a["hit"].astype(np.uint32), v[:, 6])


@pytest.mark.parametrize("which,n", [("cubes", 200000), ("lowres", 40000)])
def test_trace_bvh_equals_brute_equals_oracle(which, n, cubes, lowres):
    s, ctx = cubes if which == "cubes" else lowres
    rays = _ray_batch(s, n, 1234)
    ref = orc.trace(s, rays)
    for accel in (pyrt.ACCEL_BRUTE, pyrt.ACCEL_BVH):
        h = ctx.trace(rays, accel)
        assert np.array_equal(h.view(np.uint8), ref.view(np.uint8)), accel  # every field, every bit
        a = ctx.trace(rays, accel, pyrt.TRACE_ANY)
        assert np.array_equal(a["hit"], ref["hit"])
    assert 0.3 < ref["hit"].mean() < 1.0


def test_trace_large_bvh_vs_brute_on_gpu(lowres):
    """2M rays: the BVH must agree with the exhaustive loop on every field (GPU vs GPU,
    the oracle is too slow here)."""
    s, ctx = lowres
    rays = _ray_batch(s, 2_000_000, 99)
    a = ctx.trace(rays, pyrt.ACCEL_BRUTE)
    b = ctx.trace(rays, pyrt.ACCEL_BVH)
    assert np.array_equal(a.view(np.uint8), b.view(np.uint8))


def test_bvh_structure(lowres):
    s, ctx = lowres
    bi = ctx.bvh_info()
    nodes, tris = ctx.bvh_export()
    assert bi.n_tri_records == s.desc.n_triangles and bi.max_depth < 32 and bi.pad > 0
    ids = tris[:, 9]
    assert sorted(ids.tolist()) == list(range(s.desc.n_triangles))  # every triangle exactly once
    seen = np.zeros(bi.n_tri_records, bool)
    lo = nodes[:, [0, 1, 2, 6, 7, 8]].view(np.float32).reshape(-1, 2, 3)
    hi = nodes[:, [3, 4, 5, 9, 10, 11]].view(np.float32).reshape(-1, 2, 3)
    tf = tris.view(np.float32)
    for ni in range(bi.n_nodes):
        for c in range(2):
            ch = int(nodes[ni, 12 + c].view(np.int32)) if False else int(np.int32(nodes[ni, 12 + c]))
            if ch < 0:
                code = (~ch) & 0xFFFFFFFF
                first, cnt = code >> 3, (code & 7) + 1
                assert cnt <= bi.leaf_max and not seen[first:first + cnt].any()
                seen[first:first + cnt] = True
                p0 = tf[first:first + cnt, 0:3]
                verts = np.stack([p0, p0 + tf[first:first + cnt, 3:6], p0 + tf[first:first + cnt, 6:9]], 1)
                assert (verts >= lo[ni, c] - 1e-6).all() and (verts <= hi[ni, c] + 1e-6).all()
            else:
                assert 0 < ch < bi.n_nodes
                assert (lo[ch] >= lo[ni, c] - 1e-6).all() and (hi[ch] <= hi[ni, c] + 1e-6).all()
    assert seen.all()


# ------------------------------------------------------------------ whole frames
CASES = [("cubes", 64, 64, 8, pyrt.MODE_PATH), ("cubes", 64, 64, 4, pyrt.MODE_RAY), ("cubes", 96, 64, 3, pyrt.MODE_PATH),
         ("cubes", 37, 29, 5, pyrt.MODE_PATH), ("lowres", 40, 40, 4, pyrt.MODE_PATH), ("lowres", 32, 32, 2, pyrt.MODE_RAY)]


@pytest.mark.parametrize("kind,w,h,spp,mode", CASES)
def test_frame_bit_exact_vs_oracle(kind, w, h, spp, mode):
    s = pyrt.Scene(kind, w, h)
    ctx = pyrt.Context(s)
    bg = pyrt.background(w, h)
    p = pyrt.make_params(w, h, spp, mode=mode, seed=17, collect_stats=1)
    ref_out, ref_acc, ref_st = orc.render(s, p, math_mode=orc.MATH_DET, bg=bg)
    for accel in (pyrt.ACCEL_BVH, pyrt.ACCEL_BRUTE):
        p.accel = accel
        out, acc, st = ctx.render(p, bg)
        assert np.array_equal(bits(acc), bits(ref_acc)), (kind, accel)
        assert np.array_equal(bits(out), bits(ref_out))
        assert (st.rays_closest, st.rays_shadow, st.samples) == (ref_st.rays_closest, ref_st.rays_shadow, w * h * spp)
        if accel == pyrt.ACCEL_BRUTE:
            # closest-hit casts test every triangle like the reference; the any-hit
            # (shadow) loop leaves at its first accepted triangle, the reference's does not
            T = s.desc.n_triangles
            assert st.rays_closest * T <= st.tris_tested <= ref_st.tris_tested == (st.rays_closest + st.rays_shadow) * T
    assert orc.ppm_md5(out) == orc.ppm_md5(ref_out)
    ctx.close()


def test_frame_is_deterministic_and_split_invariant():
    """Same frame rendered (a) twice, (b) as 3 sample ranges, (c) as 2 x 3 tile shards
    summed afterwards: all bit-identical (the per-pixel float sum order never changes)."""
    w, h, spp = 128, 96, 6
    s = pyrt.Scene("lowres", w, h)
    ctx = pyrt.Context(s)
    p = pyrt.make_params(w, h, spp, seed=5)
    _, full, _ = ctx.render(p)
    _, again, _ = ctx.render(p)
    assert np.array_equal(bits(full), bits(again))
    import torch
    acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    for b, c in ((0, 1), (1, 3), (4, 2)):
        q = pyrt.make_params(w, h, spp, seed=5, spp_begin=b, spp_count=c)
        ctx.render_device(q, acc.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(bits(acc.cpu().numpy()), bits(full))
    for world, tile in ((2, 8), (3, 16), (5, 32)):
        total = np.zeros((h, w, 4), np.float32)
        owned = np.zeros((h, w), np.int32)
        for r in range(world):
            q = pyrt.make_params(w, h, spp, seed=5, rank=r, world=world, tile=tile)
            _, part, st = ctx.render(q)
            owned += (part[..., :3].sum(-1) + part[..., 3] > 0)
            total += part  # adding zeros is exact
        assert np.array_equal(bits(total), bits(full)), (world, tile)
        assert owned.max() <= 1
    ctx.close()


def test_parameter_validation(cubes):
    s, ctx = cubes
    for kw, code in ((dict(rng_mode=pyrt.RNG_LEGACY), 4), (dict(max_depth=0), 4), (dict(max_depth=4), 4),
                     (dict(use_photons=1, k=5, photons_requested=100), 5), (dict(rank=2, world=2), 1),
                     (dict(tile=12), 1), (dict(spp_begin=3, spp_count=4), 1)):
        with pytest.raises(pyrt.RtError) as e:
            ctx.render(pyrt.make_params(16, 16, 4, **kw))
        assert e.value.code == code, kw
    with pytest.raises(pyrt.RtError):
        ctx.render(pyrt.make_params(0, 16, 4))
    with pytest.raises(pyrt.RtError) as e:
        ctx.knn(np.zeros((1, 3), np.float32), 3)
    assert e.value.code == 5 and "tree is empty" in str(e.value)


# ------------------------------------------------------------------ photon map
def test_photon_emission_bit_exact_vs_oracle(cubes):
    s, ctx = cubes
    pos, dir_, w = ctx.emit_photons(6000, seed=9)
    ref, _, _ = orc.emit_photons(s, 6000, pyrt.RNG_PIXEL, seed=9, math_mode=orc.MATH_DET)
    assert len(pos) == len(ref) > 3000
    assert np.array_equal(bits(pos), bits(ref[:, 0:3])) and np.array_equal(bits(dir_), bits(ref[:, 3:6]))
    assert np.array_equal(bits(w), bits(ref[:, 6]))


def test_knn_golden_and_oracle(golden, cubes):
    """k-NN on the REFERENCE's photon list and kd order (golden), results as the
    reference's kdtree::knearest returned them; then a big random query set vs the oracle."""
    s, ctx = cubes
    V = golden["vectors"]
    ph = np.array(V["photons_cubes_3000"], np.uint32).view(np.float32).reshape(-1, 7)
    kp, kd_, kw = pyrt.kd_order(ph[:, 0:3], ph[:, 3:6], ph[:, 6])
    assert np.array_equal(bits(kp), np.array(V["kdtree_order_pos"], np.uint32).reshape(-1, 3))
    ctx.set_photons(kp, kd_)
    q = np.array(V["knearest"], np.uint32)
    i = 0
    by_k = {}
    while i < len(q):
        k = int(q[i + 3])
        by_k.setdefault(k, []).append((f32(q[i:i + 3]), int(q[i + 4]), q[i + 5:i + 5 + 6 * k].reshape(k, 6)))
        i += 5 + 6 * k
    for k, items in by_k.items():
        idx, dist, vis = ctx.knn(np.stack([it[0] for it in items]), k)
        for j, (_, visited, res) in enumerate(items):
            assert vis[j] <= visited  # the GPU walk skips subtrees that cannot change the result
            assert np.array_equal(bits(kp[idx[j]]), res[:, 0:3]) and np.array_equal(bits(kd_[idx[j]]), res[:, 3:6])
    rng = np.random.default_rng(3)
    qs = rng.uniform([-1.5, -1, -1.5], [1.5, 1.5, 1.5], (50000, 3)).astype(np.float32)
    qs[::4] = kp[rng.integers(0, len(kp), len(qs[::4]))] + rng.normal(0, 0.02, (len(qs[::4]), 3)).astype(np.float32)
    qs[7] = kp[11]  # exact hit
    kd7 = np.concatenate([kp, kd_, kw[:, None]], 1)
    for k in (1, 2, 5, 10, 16):
        idx, dist, vis = ctx.knn(qs, k)
        ri, rd, rv = orc.knn(kd7, qs, k)
        assert np.array_equal(idx, ri) and np.array_equal(bits(dist), bits(rd))
        assert (vis <= rv).all() and (k < 5 or vis.sum() < 0.6 * rv.sum())
    ctx.set_photons(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))


@pytest.mark.parametrize("kind,w,h,spp,mode,nph,k", [("cubes", 64, 64, 3, pyrt.MODE_RAY, 5000, 10),
                                                     ("cubes", 48, 40, 2, pyrt.MODE_PATH, 3000, 5),
                                                     ("lowres", 32, 32, 2, pyrt.MODE_RAY, 3000, 10)])
def test_photon_frame_bit_exact_vs_oracle(kind, w, h, spp, mode, nph, k):
    s = pyrt.Scene(kind, w, h)
    ctx = pyrt.Context(s)
    pos, dir_, wt = ctx.emit_photons(nph, seed=2)
    kp, kd_, kw = pyrt.kd_order(pos, dir_, wt)
    ctx.set_photons(kp, kd_)
    bg = pyrt.background(w, h)
    p = pyrt.make_params(w, h, spp, mode=mode, seed=23, use_photons=1, k=k, photons_requested=nph, collect_stats=1)
    out, acc, st = ctx.render(p, bg)
    ext = np.concatenate([kp, kd_, kw[:, None]], 1)
    ref_out, ref_acc, ref_st = orc.render(s, p, math_mode=orc.MATH_DET, bg=bg, ext_photons=ext)
    assert np.array_equal(bits(acc), bits(ref_acc))
    assert np.array_equal(bits(out), bits(ref_out))
    assert (st.knn_queries, st.rays_shadow) == (ref_st.knn_queries, 0) and 0 < st.kd_visited <= ref_st.kd_visited
    ctx.close()


@pytest.mark.parametrize("kind,n", [("hires", 300000), ("stress", 4000)])
def test_bvh_equals_exhaustive_loop_on_big_scenes(kind, n):
    """11.7k-triangle and 1M-triangle scenes: every field of every hit record from the
    BVH kernel equals the exhaustive (reference-order) kernel's."""
    s = pyrt.Scene(kind, 256, 256)
    ctx = pyrt.Context(s)
    rays = _ray_batch(s, n, 4321)
    a = ctx.trace(rays, pyrt.ACCEL_BRUTE)
    b = ctx.trace(rays, pyrt.ACCEL_BVH)
    assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    assert np.array_equal(ctx.trace(rays, pyrt.ACCEL_BRUTE, pyrt.TRACE_ANY)["hit"], ctx.trace(rays, pyrt.ACCEL_BVH, pyrt.TRACE_ANY)["hit"])
    bi = ctx.bvh_info()
    assert bi.max_depth < 32
    ctx.close()


def test_stress_frame_bvh_equals_brute_on_a_tile():
    """C5 scene, a 16x16-pixel corner rendered with the BVH and with the exhaustive loop."""
    s = pyrt.Scene("stress", 16, 16)
    ctx = pyrt.Context(s)
    p = pyrt.make_params(16, 16, 2, seed=3)
    _, a, _ = ctx.render(p)
    p.accel = pyrt.ACCEL_BRUTE
    _, b, _ = ctx.render(p)
    assert np.array_equal(bits(a), bits(b))
    ctx.close()


def test_frame_independent_of_samples_per_wave():
    """lane = (pixel, sample): however many samples of a pixel a wave runs side by side
    (1..64), the owner lane adds them in sample order, so the frame never changes —
    also for sample counts that are not a multiple, sample sub-ranges and odd sizes."""
    w, h, spp = 45, 37, 11
    s = pyrt.Scene("cubes", w, h)
    ctx = pyrt.Context(s)
    ref = None
    for lpp in (1, 2, 4, 8, 16, 32, 64, 0):
        _, acc, st = ctx.render(pyrt.make_params(w, h, spp, seed=9, lanes_per_pixel=lpp))
        assert st.samples == w * h * spp
        if ref is None:
            ref = acc
            _, oacc, _ = orc.render(s, pyrt.make_params(w, h, spp, seed=9), math_mode=orc.MATH_DET)
            assert np.array_equal(bits(acc), bits(oacc))
        assert np.array_equal(bits(acc), bits(ref)), lpp
    import torch
    acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    for (b, c, lpp) in ((0, 3, 4), (3, 5, 16), (8, 3, 2)):
        ctx.render_device(pyrt.make_params(w, h, spp, seed=9, spp_begin=b, spp_count=c, lanes_per_pixel=lpp), acc.data_ptr(),
                          torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(bits(acc.cpu().numpy()), bits(ref))
    # photon and brute-force variants share the exchange code
    pos, dr, wt = ctx.emit_photons(2000, seed=1)
    kp, kd_, _ = pyrt.kd_order(pos, dr, wt)
    ctx.set_photons(kp, kd_)
    base = None
    for lpp in (1, 8):
        for kw in (dict(use_photons=1, k=5, photons_requested=2000), dict(accel=pyrt.ACCEL_BRUTE)):
            _, a, _ = ctx.render(pyrt.make_params(w, h, 5, seed=2, lanes_per_pixel=lpp, **kw))
            key = tuple(sorted(kw))
            base = base or {}
            if key in base:
                assert np.array_equal(bits(a), bits(base[key]))
            base[key] = a
    ctx.close()


@pytest.mark.parametrize("scale", [1e-3, 1.0, 37.0, 1e4])
def test_bvh_exact_on_scaled_scenes(scale, lowres):
    """The packed (binary16) boxes are stored as coordinate x a power of two chosen
    from the scene extent; BVH hits must stay identical to the exhaustive loop for
    scenes three orders of magnitude smaller / four larger than the Cornell box."""
    s, _ = lowres
    a = s.arrays()
    cam = a["camera"] * np.float32(scale)
    scaled = pyrt.ArrayScene(a["pos"] * np.float32(scale), a["nrm"], a["tri"], a["tri_begin"], a["vtx_begin"],
                             a["materials"], a["lights"], cam)
    ctx = pyrt.Context(scaled)
    rays = _ray_batch(scaled, 60000, 77)
    rays["origin"][len(rays) // 2 + len(rays) // 16:] = cam[0]
    b = ctx.trace(rays, pyrt.ACCEL_BVH)
    e = ctx.trace(rays, pyrt.ACCEL_BRUTE)
    assert np.array_equal(b.view(np.uint8), e.view(np.uint8))
    ref = orc.trace(scaled, rays[:3000])
    assert np.array_equal(b[:3000].view(np.uint8), ref.view(np.uint8))
    assert 0.2 < e["hit"].mean()
    ctx.close()


@pytest.mark.parametrize("scale", [1e-3, 37.0, 1e4, 1e9, 1e10, 1e15])
def test_frames_exact_on_scaled_scenes_either_side_of_the_reciprocal_bound(scale, lowres):
    """The default render instances take 1 / det, the lengths and 1 / length in short forms that are the IEEE operations'
    bits only inside 2^-100 ... 2^100 (rt_device.h recip_fast / sqrt_fast); rt_create decides from the scene's magnitudes
    whether a launch may use them (rt_api.cpp create_ctx: at the Cornell box's size x 1e9 it still may, from x 3e9 on |det|
    is no longer bounded, from 1e14 on the inputs themselves are too large) and picks the dividing instances otherwise.
    Whatever it picks, the frame must be the oracle's and the exhaustive loop's, bit for bit."""
    s, _ = lowres
    a = s.arrays()
    f = np.float32(scale)
    lights = a["lights"].copy()
    lights[:, 0:3] *= f  # (rt_light: position[3] ... intensity, side at word 16)
    lights[:, 16] *= f
    scaled = pyrt.ArrayScene(a["pos"] * f, a["nrm"], a["tri"], a["tri_begin"], a["vtx_begin"], a["materials"], lights, a["camera"] * f)
    ctx = pyrt.Context(scaled)
    p = pyrt.make_params(32, 32, 3, seed=13)
    _, acc, st = ctx.render(p)
    _, brute, _ = ctx.render(pyrt.make_params(32, 32, 3, seed=13, accel=pyrt.ACCEL_BRUTE))
    _, ref, rst = orc.render(scaled, p, math_mode=orc.MATH_DET)
    assert np.array_equal(bits(acc), bits(brute))
    assert np.array_equal(bits(acc), bits(ref)) and (st.rays_closest, st.rays_shadow) == (rst.rays_closest, rst.rays_shadow)
    ctx.close()


def test_degenerate_inputs_are_rejected_or_harmless():
    s = pyrt.Scene("cubes", 16, 16)
    a = s.arrays()
    bad = a["pos"].copy()
    bad[3, 1] = np.nan
    with pytest.raises(pyrt.RtError) as e:
        pyrt.Context(pyrt.ArrayScene(bad, a["nrm"], a["tri"], a["tri_begin"], a["vtx_begin"], a["materials"], a["lights"],
                                     a["camera"]))
    assert e.value.code == 1 and "non-finite" in str(e.value)
    tri = a["tri"].copy()
    tri[0, 0] = 9999
    with pytest.raises(pyrt.RtError):
        pyrt.Context(pyrt.ArrayScene(a["pos"], a["nrm"], tri, a["tri_begin"], a["vtx_begin"], a["materials"], a["lights"],
                                     a["camera"]))
    # a single-triangle scene and zero-area triangles still trace exactly
    one = pyrt.ArrayScene(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [2, 2, 2], [2, 2, 2], [2, 2, 2]], np.float32),
                          np.tile(np.array([0, 0, 1], np.float32), (6, 1)), np.array([[0, 1, 2], [3, 4, 5]], np.uint32),
                          [0, 2], [0, 6], a["materials"][:1], a["lights"], a["camera"])
    ctx = pyrt.Context(one)
    rays = np.zeros(3, pyrt.RAY_DTYPE)
    rays["origin"] = [[0.2, 0.2, 1], [0.2, 0.2, 1], [2, 2, 3]]
    rays["direction"] = [[0, 0, -1], [0, 0, 1], [0, 0, -1]]
    h = ctx.trace(rays)
    assert h["hit"].tolist() == [1, 0, 0] and h["d"][0] == 1.0
    assert np.array_equal(h.view(np.uint8), ctx.trace(rays, pyrt.ACCEL_BRUTE).view(np.uint8))
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,w,h,spp", [("lowres", 1024, 1024, 4), ("hires", 512, 512, 6), ("stress", 256, 256, 3)])
def test_pool_and_stealing_never_change_the_frame(kind, w, h, spp):
    """Size-independent property at BASELINE-sized frames, where the oracle is too slow:
    the vertex pool (worker lanes, tail work stealing, shared atomic-min hit keys) is a
    SCHEDULE — the frame must equal, bit for bit, the one the sequential per-lane shading
    of the same kernel family produces, and the ray counts must agree."""
    s = pyrt.Scene(kind, w, h)
    ctx = pyrt.Context(s)
    _, a, sa = ctx.render(pyrt.make_params(w, h, spp, seed=5))
    _, b, sb = ctx.render(pyrt.make_params(w, h, spp, seed=5, no_pool=True))
    assert (sa.rays_closest, sa.rays_shadow, sa.samples) == (sb.rays_closest, sb.rays_shadow, sb.samples)
    assert np.array_equal(bits(a), bits(b))
    ctx.close()
