"""ctypes view of oracle/liboracle.so — the CPU restatement used as the CHECKER.

Test infrastructure only (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
"""
import ctypes as C
import hashlib
import os

import numpy as np

import pyrt

ROOT = pyrt.ROOT
MATH_LIBM, MATH_DET = 0, 1


class Opts(C.Structure):
    _fields_ = [("math_mode", C.c_int32), ("threads", C.c_int32), ("ext_photons", C.c_void_p),
                ("n_ext_photons", C.c_uint32), ("engine_state", C.c_uint32), ("accel", C.c_uint32),
                ("ray_dump", C.c_void_p), ("ray_dump_cap", C.c_uint64), ("ray_dump_count", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        L.orc_render.argtypes = [C.POINTER(pyrt.SceneDesc), C.POINTER(pyrt.Params), C.POINTER(Opts), C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.POINTER(pyrt.Stats)]
        L.orc_trace.argtypes = [C.POINTER(pyrt.SceneDesc), C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_trace2.argtypes = [C.POINTER(pyrt.SceneDesc), C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]
        L.orc_emit_photons.argtypes = [C.POINTER(pyrt.SceneDesc), C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32,
                                       C.POINTER(C.c_uint32), C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32),
                                       C.POINTER(C.c_uint64)]
        L.orc_kd_build.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_kd_order_depth.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p]
        L.orc_knn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                              C.c_void_p]
        L.orc_ppm_bytes.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64]
        L.orc_ppm_bytes.restype = C.c_uint64
        L.orc_fill_background.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.orc_fill_background.restype = None
        L.orc_tri_intersect.argtypes = [C.c_void_p] * 6
        L.orc_ray_at.argtypes = [C.POINTER(pyrt.Camera), C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.orc_ray_at.restype = None
        L.orc_bsdf.argtypes = [C.POINTER(pyrt.Material), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_bsdf.restype = None
        L.orc_eval_light.argtypes = [C.POINTER(pyrt.Light), C.c_void_p, C.c_void_p]
        L.orc_eval_light.restype = None
        L.orc_engine_next.argtypes = [C.POINTER(C.c_uint32)]
        L.orc_engine_next.restype = C.c_uint32
        L.orc_jitter.argtypes = [C.POINTER(C.c_uint32), C.c_int32, C.c_int32, C.c_void_p]
        L.orc_jitter.restype = None
        L.orc_rand_area.argtypes = [C.POINTER(C.c_uint32), C.POINTER(pyrt.Light), C.c_void_p]
        L.orc_rand_area.restype = None
        L.orc_hsphere.argtypes = [C.POINTER(C.c_uint32), C.c_int32, C.c_void_p, C.c_void_p]
        L.orc_hsphere.restype = None
        L.orc_det_asin.argtypes = [C.c_double]
        L.orc_det_asin.restype = C.c_double
        L.orc_det_sinf.argtypes = [C.c_float]
        L.orc_det_sinf.restype = C.c_float
        L.orc_det_cosf.argtypes = [C.c_float]
        L.orc_det_cosf.restype = C.c_float
        L.orc_stream_seed.argtypes = [C.c_uint32] * 4
        L.orc_stream_seed.restype = C.c_uint32
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def background(w, h):
    bg = np.empty((h, w, 3), np.float32)
    lib().orc_fill_background(_p(bg), w, h)
    return bg


ACCEL_LOOP, ACCEL_OBVH = 0, 1


def render(scene, params, math_mode=MATH_LIBM, threads=0, ext_photons=None, bg=None, engine_state=1, accel=ACCEL_LOOP):
    """Returns (out_rgb or None, accum[h,w,4], stats).  accel=ACCEL_OBVH runs the same
    restatement over the oracle's own CPU BVH (identical results, see rt_oracle.cpp OBvh)."""
    w, h = params.width, params.height
    o = Opts()
    o.math_mode, o.threads, o.engine_state, o.accel = math_mode, threads, engine_state, accel
    keep = None
    if ext_photons is not None:
        keep = np.ascontiguousarray(ext_photons, np.float32).reshape(-1, 7)
        o.ext_photons, o.n_ext_photons = keep.ctypes.data, len(keep)
    out = np.empty((h, w, 3), np.float32) if bg is not None else None
    acc = np.empty((h, w, 4), np.float32)
    st = pyrt.Stats()
    bgc = None if bg is None else np.ascontiguousarray(bg, np.float32)
    rc = lib().orc_render(scene.desc_ptr, C.byref(params), C.byref(o), _p(bgc), _p(out), _p(acc), C.byref(st))
    if rc != 0:
        raise RuntimeError("orc_render failed: %d" % rc)
    return out, acc, st


def dump_rays(scene, params, cap=40_000_000):
    """Every ray the pixel-mode frame casts, in casting order: array [n][8] = o.xyz, tag, d.xyz, 0
    with tag (uint32 bits) = kind (0 closest, 1 shadow) | path depth << 8.  Single-threaded."""
    o = Opts()
    o.math_mode, o.threads, o.engine_state, o.accel = MATH_DET, 1, 1, ACCEL_OBVH
    buf = np.zeros((cap, 8), np.float32)
    o.ray_dump, o.ray_dump_cap = buf.ctypes.data, cap
    acc = np.empty((params.height, params.width, 4), np.float32)
    st = pyrt.Stats()
    rc = lib().orc_render(scene.desc_ptr, C.byref(params), C.byref(o), None, None, _p(acc), C.byref(st))
    assert rc == 0
    return buf[:o.ray_dump_count].copy()


def trace(scene, rays, accel=ACCEL_LOOP, kind=pyrt.TRACE_CLOSEST):
    rays = np.ascontiguousarray(rays, pyrt.RAY_DTYPE)
    hits = np.zeros(len(rays), pyrt.HIT_DTYPE)
    if accel == ACCEL_LOOP and kind == pyrt.TRACE_CLOSEST:
        lib().orc_trace(scene.desc_ptr, _p(rays), len(rays), _p(hits))
    else:
        lib().orc_trace2(scene.desc_ptr, _p(rays), len(rays), _p(hits), accel, kind)
    return hits


def emit_photons(scene, n_requested, rng_mode, seed=1, math_mode=MATH_LIBM, engine_state=1):
    """Returns (photons[n,7] in emission order, engine_state_after, emission_rays)."""
    out = np.zeros((max(n_requested, 1), 7), np.float32)
    st = C.c_uint32(engine_state)
    n = C.c_uint32()
    rays = C.c_uint64()
    rc = lib().orc_emit_photons(scene.desc_ptr, n_requested, rng_mode, seed, math_mode, C.byref(st), _p(out), len(out),
                                C.byref(n), C.byref(rays))
    assert rc == 0
    return out[:n.value].copy(), st.value, rays.value


def kd_build(photons7):
    a = np.ascontiguousarray(photons7, np.float32).copy()
    lib().orc_kd_build(_p(a), len(a))
    return a


def kd_order_depth(pos, depth_limit=-1):
    """Tree-slot -> input-index permutation of kdtree::make_tree, with std::__introselect's
    depth limit forced (>= 0) or the library's own (< 0)."""
    pos = np.ascontiguousarray(pos, np.float32)
    perm = np.zeros(len(pos), np.uint32)
    lib().orc_kd_order_depth(_p(pos), len(pos), depth_limit, _p(perm))
    return perm


def knn(photons7_kd, queries, k):
    ph = np.ascontiguousarray(photons7_kd, np.float32)
    q = np.ascontiguousarray(queries, np.float32)
    idx = np.zeros((len(q), k), np.uint32)
    dist = np.zeros((len(q), k), np.float32)
    vis = np.zeros(len(q), np.uint32)
    rc = lib().orc_knn(_p(ph), len(ph), _p(q), len(q), k, _p(idx), _p(dist), _p(vis))
    if rc != 0:
        raise RuntimeError("orc_knn failed: %d" % rc)
    return idx, dist, vis


def ppm_bytes(rgb):
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w = rgb.shape[:2]
    n = lib().orc_ppm_bytes(_p(rgb), w, h, None, 0)
    buf = C.create_string_buffer(n)
    lib().orc_ppm_bytes(_p(rgb), w, h, buf, n)
    return buf.raw[:n]


def ppm_md5(rgb):
    return hashlib.md5(ppm_bytes(rgb)).hexdigest()
