"""Per-function pins: every vector in tests/golden/ref_vectors.json was produced by
calling the reference's own functions (oracle/ref_harness.cpp).  The oracle's
restatement must reproduce them bit for bit; the host layer's scene script must
reproduce the reference's Scene object bit for bit."""
import ctypes as C

import numpy as np
import pytest

import orc
import pyrt


def f32(u):
    return np.array(u, dtype=np.uint32).view(np.float32)


def u32(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("kind", ["cubes", "lowres"])
def test_host_scene_equals_reference_scene(golden, kind):
    V = golden["vectors"]
    s = pyrt.Scene(kind, 256, 256)
    a = s.arrays()
    assert np.array_equal(u32(a["pos"]).ravel(), np.array(V[kind + "_pos"], np.uint32))
    assert np.array_equal(u32(a["nrm"]).ravel(), np.array(V[kind + "_nrm"], np.uint32))
    tri_local = a["tri"].copy()
    for m in range(len(a["tri_begin"]) - 1):
        tri_local[a["tri_begin"][m]:a["tri_begin"][m + 1]] -= a["vtx_begin"][m]
    assert np.array_equal(tri_local.ravel(), np.array(V[kind + "_tri_local"], np.uint32))
    assert np.array_equal(a["tri_begin"], np.array(V[kind + "_tri_off"], np.uint32))
    assert np.array_equal(a["vtx_begin"], np.array(V[kind + "_vtx_off"], np.uint32))
    assert np.array_equal(u32(a["materials"]).ravel(), np.array(V[kind + "_mat"], np.uint32))
    assert np.array_equal(u32(a["camera"]).ravel(), np.array(V[kind + "_cam"], np.uint32))
    assert np.array_equal(u32(a["lights"]).ravel(), np.array(V[kind + "_lights"], np.uint32))


def test_camera_frame_non_square(golden):
    s = pyrt.Scene("cubes", 380, 270)
    assert np.array_equal(u32(s.arrays()["camera"]).ravel(), np.array(golden["vectors"]["cam_380x270"], np.uint32))


def test_ray_at(golden):
    s = pyrt.Scene("cubes", 256, 256)
    v = np.array(golden["vectors"]["rayAt"], np.uint32).reshape(-1, 8)
    o, d = np.zeros(3, np.float32), np.zeros(3, np.float32)
    for row in v:
        uu, vv = f32(row[:2])
        orc.lib().orc_ray_at(C.byref(s.desc.camera), float(uu), float(vv), orc._p(o), orc._p(d))
        assert np.array_equal(o.view(np.uint32), row[2:5]) and np.array_equal(d.view(np.uint32), row[5:8])


def test_triangle_intersect(golden):
    v = np.array(golden["vectors"]["triangleIntersect"], np.uint32).reshape(-1, 19)
    nh = 0
    for row in v:
        f = f32(row[:15]).copy()
        uvt = np.full(3, -7.0, np.float32)
        hit = orc.lib().orc_tri_intersect(orc._p(f[0:3]), orc._p(f[3:6]), orc._p(f[6:9]), orc._p(f[9:12]),
                                          orc._p(f[12:15]), orc._p(uvt))
        assert hit == row[15]
        assert np.array_equal(uvt.view(np.uint32), row[16:19])
        nh += int(hit)
    assert 50 < nh < len(v)


@pytest.mark.parametrize("kind", ["cubes", "lowres"])
def test_ray_trace(golden, kind):
    s = pyrt.Scene(kind, 256, 256)
    v = np.array(golden["vectors"]["rayTrace_" + kind], np.uint32).reshape(-1, 14)
    rays = np.zeros(len(v), pyrt.RAY_DTYPE)
    rays["origin"] = f32(v[:, 0:3])
    rays["direction"] = f32(v[:, 3:6])
    hits = orc.trace(s, rays)
    assert np.array_equal(hits["hit"].astype(np.uint32), v[:, 6])
    m = v[:, 6] == 1
    assert m.sum() > 100
    assert np.array_equal(hits["mesh"][m], v[m, 7])
    assert np.array_equal(hits["vtx"][m], v[m, 8:11])
    assert np.array_equal(u32(hits["u"][m]), v[m, 11])
    assert np.array_equal(u32(hits["v"][m]), v[m, 12])
    assert np.array_equal(u32(hits["d"][m]), v[m, 13])


def test_bsdf(golden):
    v = np.array(golden["vectors"]["bsdf"], np.uint32).reshape(-1, 20)
    out = np.zeros(3, np.float32)
    for row in v:
        f = f32(row).copy()
        m = pyrt.Material()
        m.kd, m.alpha = float(f[9]), float(f[10])
        for c in range(3):
            m.albedo[c], m.f0[c] = float(f[11 + c]), float(f[14 + c])
        orc.lib().orc_bsdf(C.byref(m), orc.MATH_LIBM, orc._p(f[0:3]), orc._p(f[3:6]), orc._p(f[6:9]), orc._p(out))
        assert np.array_equal(out.view(np.uint32), row[17:20]), (f[:17], out, f[17:20])
    # SURVEY App. C known answer
    assert any(abs(f32(r[17]) - 0.218821779) < 1e-9 for r in v)


def test_evaluate_light(golden):
    s = pyrt.Scene("cubes", 256, 256)
    v = np.array(golden["vectors"]["evaluateLight"], np.uint32).reshape(3, 20, 6)
    out = np.zeros(3, np.float32)
    for li in range(3):
        for row in v[li]:
            p = f32(row[:3]).copy()
            orc.lib().orc_eval_light(C.byref(s.desc.lights[li]), orc._p(p), orc._p(out))
            assert np.array_equal(out.view(np.uint32), row[3:6])


def test_engine_and_samplers(golden):
    V = golden["vectors"]
    L = orc.lib()
    st = C.c_uint32(1)
    assert [L.orc_engine_next(C.byref(st)) for _ in range(8)] == V["engine_first8"]
    assert V["engine_first8"][:3] == [16807, 282475249, 1622650073]

    st = C.c_uint32(1)
    xy = np.zeros(2, np.float32)
    for N, i, bx, by in np.array(V["jitterSample_seq"], np.uint32).reshape(-1, 4):
        L.orc_jitter(C.byref(st), int(i), int(N), orc._p(xy))
        assert np.array_equal(xy.view(np.uint32), [bx, by])

    s = pyrt.Scene("cubes", 256, 256)
    st = C.c_uint32(1)
    out = np.zeros(3, np.float32)
    for j, row in enumerate(np.array(V["randAreaPosition_seq"], np.uint32).reshape(-1, 3)):
        L.orc_rand_area(C.byref(st), C.byref(s.desc.lights[j % 3]), orc._p(out))
        assert np.array_equal(out.view(np.uint32), row)

    st = C.c_uint32(1)
    for row in np.array(V["hsphere_seq"], np.uint32).reshape(-1, 6):
        n = f32(row[:3]).copy()
        L.orc_hsphere(C.byref(st), orc.MATH_LIBM, orc._p(n), orc._p(out))
        assert np.array_equal(out.view(np.uint32), row[3:6])


def test_photon_emission_and_kdtree(golden):
    V = golden["vectors"]
    s = pyrt.Scene("cubes", 256, 256)
    ph, state, rays = orc.emit_photons(s, 3000, pyrt.RNG_LEGACY)
    ref = np.array(V["photons_cubes_3000"], np.uint32).reshape(-1, 7)
    assert np.array_equal(ph.view(np.uint32), ref)
    st = C.c_uint32(state)
    assert [orc.lib().orc_engine_next(C.byref(st)) for _ in range(4)] == V["photons_cubes_3000_engine_after"]

    kd = orc.kd_build(ph)
    order = np.array(V["kdtree_order_pos"], np.uint32).reshape(-1, 3)
    assert np.array_equal(kd[:, :3].view(np.uint32), order)
    # the product's host builder must produce the same order
    hp, hd, hw = pyrt.kd_order(ph[:, 0:3], ph[:, 3:6], ph[:, 6])
    assert np.array_equal(hp.view(np.uint32), order)
    assert np.array_equal(hd, kd[:, 3:6]) and np.array_equal(hw, kd[:, 6])

    q = np.array(V["knearest"], np.uint32)
    i = 0
    while i < len(q):
        pos = f32(q[i:i + 3]).copy()
        k, visited = int(q[i + 3]), int(q[i + 4])
        res = q[i + 5:i + 5 + 6 * k].reshape(k, 6)
        i += 5 + 6 * k
        idx, dist, vis = orc.knn(kd, pos[None, :], k)
        assert vis[0] == visited
        assert np.array_equal(kd[idx[0], :6].view(np.uint32), res)
