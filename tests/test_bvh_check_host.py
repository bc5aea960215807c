"""The host-built BVH in its device node formats, checked without a GPU: rt_bvh_check_host builds, packs and
verifies the structure itself (every node and every triangle record reachable exactly once, every stored child
box containing its padded geometry, the depth as reported and within the cap)."""
import numpy as np
import pytest

import pyrt

FORMATS = [pyrt.NODES_F16, pyrt.NODES_Q8]


@pytest.mark.parametrize("kind", ["cubes", "lowres", "hires"])
@pytest.mark.parametrize("fmt", FORMATS)
def test_packed_tree_is_a_valid_tree(kind, fmt):
    s = pyrt.Scene(kind, 32, 32)
    r = pyrt.bvh_check_host(s, 0, fmt)
    assert r["nodes"] >= 1 and r["depth"] >= 1
    # the packed boxes contain the float boxes, so the surface-area estimate of node visits can only grow — and the
    # encodings are fine enough that it grows by little
    assert r["visits_float"] <= r["visits_packed"] <= 1.25 * r["visits_float"]
    if fmt == pyrt.NODES_Q8:
        assert r["slots"] >= r["nodes"] + r["added_nodes"] and r["blocks"] >= 1


@pytest.mark.parametrize("fmt", FORMATS)
def test_leaf_sizes_and_tiny_scenes(fmt):
    for leaf in (1, 2, 4, 8):
        if fmt == pyrt.NODES_Q8 and leaf > 2:
            with pytest.raises(pyrt.RtError):  # the 16-byte record has one count bit per child
                pyrt.bvh_check_host(pyrt.Scene("lowres", 32, 32), leaf, fmt)
            continue
        r = pyrt.bvh_check_host(pyrt.Scene("lowres", 32, 32), leaf, fmt)
        assert r["nodes"] >= 1
    # a scene of one, two and three triangles: the root still has two (leaf) children
    a = pyrt.Scene("cubes", 32, 32).arrays()
    for ntri in (1, 2, 3):
        tri = a["tri"][:ntri]
        sc = pyrt.ArrayScene(a["pos"], a["nrm"], tri, np.array([0, ntri], np.uint32), np.array([0, len(a["pos"])], np.uint32),
                             a["materials"][:1], a["lights"], a["camera"])
        r = pyrt.bvh_check_host(sc, 0, fmt)
        assert r["nodes"] >= 1 and r["depth"] >= 1


@pytest.mark.parametrize("fmt", FORMATS)
def test_big_lattice_scene_packs_and_checks(fmt):
    """The 1 M-triangle stress scene (BASELINE config 5) is what the Q8 format is for: thousands of 16-KiB blocks."""
    r = pyrt.bvh_check_host(pyrt.Scene("stress", 32, 32), 0, fmt)
    assert r["nodes"] > 400000
    if fmt == pyrt.NODES_Q8:
        assert r["blocks"] > 1000 and r["visits_packed"] <= 1.15 * r["visits_float"]


@pytest.mark.parametrize("kind", ["cubes", "lowres"])
def test_tuner_machinery_on_a_host_computable_cost(kind, monkeypatch):
    """bvh_build.cpp tuneMeasured (what rt_bvh_tune drives with a probe frame's counters) run on a cost the host can
    compute, the summed surface area of the child boxes (RT_BVH_TUNE_AREA=probes): proposals, undo of rejected moves,
    slot flips and the depth bound all run without a GPU, and the structure check must still pass — every triangle once,
    boxes containing their padded geometry, no leaf deeper than before."""
    s = pyrt.Scene(kind, 32, 32)
    monkeypatch.setenv("RT_BVH_ROT", "0")
    base = pyrt.bvh_check_host(s, 0, pyrt.NODES_F16)
    monkeypatch.setenv("RT_BVH_TUNE_AREA", "400")
    tuned = pyrt.bvh_check_host(s, 0, pyrt.NODES_F16)
    assert tuned["nodes"] == base["nodes"] and tuned["depth"] <= base["depth"]
    assert tuned["visits_float"] <= base["visits_float"]
    if kind == "lowres":
        assert tuned["visits_float"] < 0.995 * base["visits_float"]


@pytest.mark.parametrize("kind,cutoff", [("lowres", 64), ("lowres", 1024), ("hires", 256), ("hires", 1024), ("stress", 1024)])
def test_hybrid_builders_host_top_is_a_valid_partial_tree(kind, cutoff):
    """rtbvh::buildTop (what rt_create runs on the host for RT_BVH_HYBRID): the host builder's own splits down to parts of at
    most `cutoff` triangles.  Parts and top leaves cover the order exactly once, every part has one referrer whose box contains
    its padded geometry, and there is depth left for every part's subtree."""
    s = pyrt.Scene(kind, 32, 32)
    r = pyrt.bvh_top_check_host(s, 0, cutoff)
    n = s.desc.n_triangles
    assert r["parts"] >= 2 and 2 < r["largest_part"] <= cutoff and r["top_nodes"] >= r["parts"] // 2
    assert r["deepest_part"] < r["depth_cap"]
    if kind == "stress":
        assert 1000 < r["parts"] < 4000 and r["top_nodes"] < 4000  # (1,450 parts for 1 M triangles: the device does the rest)
    with pytest.raises(pyrt.RtError):
        pyrt.bvh_top_check_host(s, 0, max(n, 3))  # a single part: nothing to build on the host


@pytest.mark.parametrize("kind", ["identical", "line", "corner", "flat", "random"])
def test_host_builders_survive_hard_soups(kind):
    """tests/soups.py: coincident triangles, collinear centroids with wildly different sizes, a dense cluster beside scene-sized
    triangles, a flat sheet, a random soup — at a size where RT_BVH_AUTO takes the hybrid builder.  The host top, the float tree
    and both packed formats must be valid trees within the depth cap (the GPU side: tests/test_gpu_soups.py)."""
    import soups
    s = soups.soup(kind, 140000)
    top = pyrt.bvh_top_check_host(s, 0, 1024)
    assert top["parts"] >= 137 and top["largest_part"] <= 1024 and top["deepest_part"] < top["depth_cap"]
    for fmt in (pyrt.NODES_F16, pyrt.NODES_Q8):
        r = pyrt.bvh_check_host(s, 0, fmt)
        assert 70000 <= r["nodes"] < 140000 and r["depth"] < 32
