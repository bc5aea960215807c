"""The C-ABI libraries load on a machine without a GPU and export every symbol
include/rt_amd.h and include/rt_host.h declare; compute entry points refuse to
run (no CPU fallback) when no gfx950 device is present."""
import ctypes as C
import os
import re

import pytest

import pyrt


def _declared(header):
    txt = open(os.path.join(pyrt.ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", txt)))


def test_amd_symbols_match_header():
    decl = _declared("rt_amd.h")
    assert decl == sorted(pyrt.AMD_SYMBOLS)
    L = pyrt.amd()
    for s in decl:
        assert hasattr(L, s), s
    assert L.rt_abi_version() == 2


def test_host_symbols_match_header():
    decl = [s for s in _declared("rt_host.h")]
    assert decl == sorted(pyrt.HOST_SYMBOLS)
    L = pyrt.host()
    for s in decl:
        assert hasattr(L, s), s


def test_struct_sizes():
    assert C.sizeof(pyrt.Material) == 32 and C.sizeof(pyrt.Light) == 84 and C.sizeof(pyrt.Camera) == 48
    assert C.sizeof(pyrt.Params) == 96 and C.sizeof(pyrt.Stats) == 104
    assert pyrt.RAY_DTYPE.itemsize == 24 and pyrt.HIT_DTYPE.itemsize == 36


def _has_gpu():
    import torch
    return torch.cuda.device_count() > 0


@pytest.mark.skipif(_has_gpu(), reason="a GPU is present")
def test_no_device_is_a_loud_error_not_a_fallback():
    s = pyrt.Scene("cubes", 32, 32)
    with pytest.raises(pyrt.RtError) as e:
        pyrt.Context(s)
    assert e.value.code == 2 and "no CPU path" in str(e.value)
    import numpy as np
    with pytest.raises(pyrt.RtError):
        pyrt.unit(pyrt.UNIT_SINF, np.zeros(4, np.float32))
    with pytest.raises(pyrt.RtError) as e:
        pyrt.Group(s, [0, 0])
    assert e.value.code == 2
    with pytest.raises(pyrt.RtError) as e:
        pyrt.kd_order_device(np.zeros((8, 3), np.float32))
    assert e.value.code == 2
    with pytest.raises(pyrt.RtError) as e:
        pyrt.Context(s, bvh_builder=pyrt.BVH_DEVICE)
    assert e.value.code == 2


def test_scene_rejects_unknown_kind_and_missing_mesh(tmp_path):
    with pytest.raises(pyrt.RtError):
        pyrt.Scene("nope", 8, 8)
    with pytest.raises(pyrt.RtError):
        pyrt.Scene("cubes", 8, 8, mesh_dir=str(tmp_path))


def test_stress_scene_shape():
    s = pyrt.Scene("stress", 64, 64)
    assert s.desc.n_triangles == 6 + 2 + 2 + 1000008 + 12
    a = s.arrays()
    lo, hi = a["pos"][a["vtx_begin"][3]:a["vtx_begin"][4]].min(0), a["pos"][a["vtx_begin"][3]:a["vtx_begin"][4]].max(0)
    assert (lo > [-1.41, -0.96, -1.41]).all() and (hi < [1.41, 1.41, 1.01]).all()


def test_off_loader_fan_triangulates_polygons_and_skips_comments(tmp_path):
    """Mesh::loadOFF (reference source/Mesh.h:57-90): header comment lines, polygons with
    more than 3 corners become a fan around their first vertex."""
    import shutil
    import numpy as np
    shutil.copy(os.path.join(pyrt.MESH_DIR, "cube_tri2.off"), tmp_path / "cube_tri2.off")
    (tmp_path / "poly.off").write_text("OFF\n# a comment line\n6 2 0\n0 0 0\n1 0 0\n1 1 0\n0 1 0\n-0.5 0.5 0\n0.5 -0.5 0\n"
                                       "4 0 1 2 3\n5 0 5 1 2 4\n")
    s = pyrt.Scene("file:poly.off", 16, 16, mesh_dir=str(tmp_path))
    a = s.arrays()
    b, e = a["tri_begin"][3], a["tri_begin"][4]
    assert e - b == 2 + 3
    local = a["tri"][b:e] - a["vtx_begin"][3]
    assert local.tolist() == [[0, 1, 2], [0, 2, 3], [0, 5, 1], [0, 1, 2], [0, 2, 4]]
    n = a["nrm"][a["vtx_begin"][3]:a["vtx_begin"][4]]
    assert np.allclose(np.abs(n[:, 2]), 1.0)  # planar polygon: all vertex normals are +-z


def test_host_bvh_build_does_not_depend_on_thread_count():
    """The fork-join SAH build (bvh_build.cpp) must produce the same arrays — float
    nodes, packed f16 nodes, triangle records — for every thread count."""
    for kind in ("cubes", "lowres", "hires"):
        s = pyrt.Scene(kind, 32, 32)
        runs = [pyrt.bvh_build_host(s, 0, t) for t in (1, 2, 5, 0)]
        assert len({d for _, d, _ in runs}) == 1, kind
        info = runs[0][0]
        assert info.n_tri_records == s.desc.n_triangles and info.max_depth <= 31 and info.leaf_max == 2
    # a range large enough to fork several levels deep
    s = pyrt.Scene("stress", 32, 32)
    a, b = pyrt.bvh_build_host(s, 0, 1), pyrt.bvh_build_host(s, 0, 8)
    # (19 balanced levels + 5 spare: the deepest tree that still leaves 14 waves of LDS, bvh_build.h)
    assert a[1] == b[1] and a[0].n_nodes == b[0].n_nodes and a[0].max_depth == b[0].max_depth == 24
