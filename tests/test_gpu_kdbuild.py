"""The photon kd-tree order built on the device (csrc/kd_build.hip) against
std::nth_element itself: the reference's golden photon list and tree order
(tests/golden/ref_vectors.json, produced by the reference's kdtree::make_tree), the
oracle's / host's restatement on tie-heavy inputs (distinct keys make the final order
canonical — only ties expose the selection algorithm), forced depth limits (the
__heap_select path), and the whole device pipeline emission -> compaction -> order."""
import numpy as np
import pytest

import orc
import pyrt

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _tie_heavy(rng, n, kind):
    pos = rng.uniform(-1.5, 1.5, (n, 3)).astype(np.float32)
    if kind == 1:  # coarse grid: every key value shared by hundreds of elements
        pos = (np.round(pos * 6) / 6).astype(np.float32)
    elif kind == 2:  # photons on axis-aligned walls: exact ties on one coordinate
        wall = rng.integers(0, 5, n)
        pos[wall == 0, 1] = -1.0
        pos[wall == 1, 0] = -1.51
        pos[wall == 2, 0] = 1.51
        pos[wall == 3, 2] = -1.51
    elif kind == 3:  # all equal on two axes
        pos[:, 0] = 0.25
        pos[:, 2] = -0.5
    elif kind == 4:  # sorted / reverse-sorted runs
        pos = np.sort(pos, axis=0)
        pos[n // 2:] = pos[n // 2:][::-1]
    return pos


def test_golden_kdtree_order(golden):
    """The reference's own photon list (3000 requested, legacy RNG) and the order its
    kdtree::make_tree left it in."""
    V = golden["vectors"]
    ph = np.array(V["photons_cubes_3000"], np.uint32).view(np.float32).reshape(-1, 7)
    perm, ms = pyrt.kd_order_device(ph[:, 0:3])
    assert np.array_equal(bits(ph[perm, 0:3]), np.array(V["kdtree_order_pos"], np.uint32).reshape(-1, 3))


@pytest.mark.parametrize("kind", [0, 1, 2, 3, 4])
def test_device_order_equals_nth_element_order(kind):
    rng = np.random.default_rng(100 + kind)
    for n in (1, 2, 3, 4, 5, 64, 191, 192, 193, 257, 1000, 4097, 35807, 100000):
        pos = _tie_heavy(rng, n, kind)
        perm, ms = pyrt.kd_order_device(pos)
        ref = orc.kd_order_depth(pos, -1)
        assert np.array_equal(perm, ref), (kind, n, int((perm != ref).sum()))
        # and the product's host builder (std::nth_element) agrees with both
        hp, _, _ = pyrt.kd_order(pos, np.zeros_like(pos))
        assert np.array_equal(bits(hp), bits(pos[perm]))


@pytest.mark.parametrize("depth", [0, 1, 2, 5])
def test_heap_select_path(depth):
    """std::__introselect with its depth limit forced low: __heap_select + the final swap,
    on ranges of every size (large ones on global memory, small ones in LDS)."""
    rng = np.random.default_rng(7 + depth)
    for n, kind in ((50, 1), (500, 2), (3000, 1), (20000, 2)):
        pos = _tie_heavy(rng, n, kind)
        perm, _ = pyrt.kd_order_device(pos, depth_limit=depth)
        assert np.array_equal(perm, orc.kd_order_depth(pos, depth)), (depth, n)


def test_photon_map_built_on_the_device():
    """rt_build_photon_map = emission + compaction + order, nothing on the host: the same
    map as rt_emit_photons + the host's std::nth_element order, and the same frame."""
    w, h, nph, k = 64, 48, 20000, 10
    s = pyrt.Scene("cubes", w, h)
    a, b = pyrt.Context(s), pyrt.Context(s)
    pos, dir_, wt = a.emit_photons(nph, seed=5)
    kp, kd_, kw = pyrt.kd_order(pos, dir_, wt)
    a.set_photons(kp, kd_)
    n, ms = b.build_photon_map(nph, seed=5)
    assert n == len(kp) > 10000
    gp, gd, gw = b.get_photons(n)
    assert np.array_equal(bits(gp), bits(kp)) and np.array_equal(bits(gd), bits(kd_)) and np.array_equal(bits(gw), bits(kw))
    p = pyrt.make_params(w, h, 2, mode=pyrt.MODE_RAY, seed=3, use_photons=1, k=k, photons_requested=nph)
    _, fa, _ = a.render(p)
    _, fb, _ = b.render(p)
    assert np.array_equal(bits(fa), bits(fb))
    # the oracle agrees with the frame
    ext = np.concatenate([kp, kd_, kw[:, None]], 1)
    _, fo, _ = orc.render(s, p, math_mode=orc.MATH_DET, ext_photons=ext)
    assert np.array_equal(bits(fb), bits(fo))
    # rebuilding replaces the map; zero photons leaves none
    n2, _ = b.build_photon_map(3000, seed=6)
    assert 0 < n2 < n
    assert b.build_photon_map(0)[0] == 0
    with pytest.raises(pyrt.RtError):
        b.render(p)
    print("photon map of %d on the device: emit %.2f ms, kd order %.2f ms" % (n, ms[0], ms[1]))
    a.close()
    b.close()
