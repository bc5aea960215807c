"""pyrt — thin ctypes view of the C ABI (include/rt_amd.h, include/rt_host.h).

Plumbing for tests and bench.py only: every call goes straight through the
shared libraries built by ray-tracing-engine_amd/Makefile.  There is no Python
implementation of anything and no fallback: if librt_amd.so is missing, or no
gfx950 device is present, the calls raise RtError.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(_PKG))
LIB_DIR = os.path.join(os.path.dirname(_PKG), "lib")
MESH_DIR = os.path.join(ROOT, "tests", "golden", "meshes")

RT_OK = 0
MODE_RAY, MODE_PATH = 0, 1
RNG_LEGACY, RNG_PIXEL = 0, 1
ACCEL_BVH, ACCEL_BRUTE = 0, 1
BVH_AUTO, BVH_DEVICE, BVH_HYBRID, BVH_HOST = 0, 1, 2, 3
TRACE_CLOSEST, TRACE_ANY = 0, 1
(UNIT_ASIN, UNIT_SINF, UNIT_COSF, UNIT_STREAM_SEED, UNIT_TRIANGLE, UNIT_BSDF, UNIT_RAY_AT, UNIT_LIGHT_EVAL,
 UNIT_SAMPLERS, UNIT_LIGHT_SAMPLE, UNIT_POW, UNIT_RECIP) = range(12)
KMAX = 16


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rt error %d: %s" % (code, msg))
        self.code = code


class Material(C.Structure):
    _fields_ = [("kd", C.c_float), ("alpha", C.c_float), ("albedo", C.c_float * 3), ("f0", C.c_float * 3)]


class Light(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("color", C.c_float * 3), ("vertical", C.c_float * 3),
                ("horizontal", C.c_float * 3), ("normal", C.c_float * 3), ("intensity", C.c_float),
                ("side", C.c_float), ("factor", C.c_float), ("ac", C.c_float), ("al", C.c_float), ("aq", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("lower_left", C.c_float * 3), ("horizontal", C.c_float * 3),
                ("vertical", C.c_float * 3)]


class SceneDesc(C.Structure):
    _fields_ = [("n_meshes", C.c_uint32), ("n_vertices", C.c_uint32), ("n_triangles", C.c_uint32),
                ("n_lights", C.c_uint32), ("vertex_pos", C.POINTER(C.c_float)), ("vertex_nrm", C.POINTER(C.c_float)),
                ("tri_vtx", C.POINTER(C.c_uint32)), ("mesh_tri_begin", C.POINTER(C.c_uint32)),
                ("mesh_vtx_begin", C.POINTER(C.c_uint32)), ("materials", C.POINTER(Material)),
                ("lights", C.POINTER(Light)), ("camera", Camera)]


class Options(C.Structure):
    _fields_ = [("device", C.c_int32), ("bvh_leaf_max", C.c_uint32), ("bvh_builder", C.c_uint32), ("node_format", C.c_uint32), ("reserved", C.c_uint32 * 4)]


class Params(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in
                ("width", "height", "spp", "mode", "max_depth", "seed", "rng_mode", "accel", "use_photons", "k",
                 "photons_requested", "spp_begin", "spp_count", "rank", "world", "tile", "collect_stats")] + \
               [("reserved", C.c_uint32 * 7)]


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("knn_queries", C.c_uint64), ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64),
                ("kd_visited", C.c_uint64), ("frame_fetches", C.c_uint64), ("kernel_ms", C.c_double), ("reserved", C.c_uint64 * 4)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "reserved"}


class TuneReport(C.Structure):
    _fields_ = [("probes", C.c_uint32), ("accepted", C.c_uint32), ("cost_before", C.c_double), ("cost_after", C.c_double),
                ("seconds", C.c_double), ("reserved", C.c_uint64 * 4)]


class BvhInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_tri_records", C.c_uint32), ("max_depth", C.c_uint32),
                ("leaf_max", C.c_uint32), ("pad", C.c_float), ("build_ms", C.c_float), ("builder", C.c_uint32),
                ("node_format", C.c_uint32), ("flags", C.c_uint32)]


RAY_DTYPE = np.dtype([("origin", "<f4", 3), ("direction", "<f4", 3)])
HIT_DTYPE = np.dtype([("hit", "<i4"), ("mesh", "<u4"), ("tri", "<u4"), ("vtx", "<u4", 3), ("u", "<f4"),
                      ("v", "<f4"), ("d", "<f4")])

# every symbol include/rt_amd.h / include/rt_host.h declares
AMD_SYMBOLS = ["rt_abi_version", "rt_last_error", "rt_create", "rt_destroy", "rt_set_photons", "rt_emit_photons",
               "rt_render", "rt_render_passes", "rt_render_device", "rt_resolve_device", "rt_trace", "rt_knn", "rt_bvh_info_get",
               "rt_bvh_export", "rt_bvh_build_host", "rt_bvh_check_host", "rt_bvh_top_check_host", "rt_bvh_tune", "rt_profile_reset", "rt_profile_collect", "rt_test_unit",
               "rt_trace_stream_device", "rt_build_photon_map", "rt_get_photons", "rt_test_kd_order", "rt_owned_granules", "rt_pack_owned_device", "rt_unpack_owned_device", "rt_group_create", "rt_group_destroy",
               "rt_group_size", "rt_group_uses_rccl", "rt_group_ctx", "rt_group_set_photons", "rt_group_render"]
HOST_SYMBOLS = ["rt_host_scene_build", "rt_host_scene_desc", "rt_host_scene_free", "rt_host_last_error",
                "rt_host_fill_background", "rt_host_save_ppm", "rt_host_kd_order"]

_amd = None
_host = None


def make_params(width, height, spp, mode=MODE_PATH, seed=1, accel=ACCEL_BVH, max_depth=3, rng_mode=RNG_PIXEL,
                use_photons=0, k=0, photons_requested=0, spp_begin=0, spp_count=0, rank=0, world=1, tile=8,
                collect_stats=0, lanes_per_pixel=0, no_pool=False, wavefront=False):
    p = Params()
    p.width, p.height, p.spp, p.mode, p.max_depth, p.seed = width, height, spp, mode, max_depth, seed
    p.rng_mode, p.accel, p.use_photons, p.k, p.photons_requested = rng_mode, accel, use_photons, k, photons_requested
    p.spp_begin, p.spp_count, p.rank, p.world, p.tile, p.collect_stats = spp_begin, spp_count, rank, world, tile, collect_stats
    p.reserved[0] = lanes_per_pixel  # 0 = automatic; power of two <= 64: samples of a pixel a wave runs side by side
    p.reserved[1] = 1 if no_pool else 0  # schedule only: sequential shading instead of the wave's ray pool
    p.reserved[2] = 1 if wavefront else 0  # schedule only: the queue-based integrator
    return p


def amd():
    """librt_amd.so (HIP kernels + C ABI).  Loads without a GPU; compute calls then fail."""
    global _amd
    if _amd is None:
        # RT_AMD_LIB: load a diagnostic build of the same library instead (tools/phase_timing.sh)
        path = os.environ.get("RT_AMD_LIB") or os.path.join(LIB_DIR, "librt_amd.so")
        if not os.path.exists(path):
            raise RtError(-1, "%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
        L = C.CDLL(path, mode=C.RTLD_GLOBAL)
        L.rt_last_error.restype = C.c_char_p
        L.rt_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(Options), C.POINTER(C.c_void_p)]
        L.rt_destroy.argtypes = [C.c_void_p]
        L.rt_destroy.restype = None
        L.rt_set_photons.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.rt_emit_photons.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.POINTER(C.c_uint32)]
        L.rt_render.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        L.rt_render_passes.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        L.rt_render_device.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        L.rt_resolve_device.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]
        L.rt_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.rt_knn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_bvh_info_get.argtypes = [C.c_void_p, C.POINTER(BvhInfo)]
        L.rt_bvh_export.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_bvh_tune.argtypes = [C.c_void_p, C.POINTER(Params), C.c_double, C.c_uint32, C.POINTER(TuneReport)]
        L.rt_bvh_build_host.argtypes = [C.POINTER(SceneDesc), C.c_uint32, C.c_uint32, C.POINTER(BvhInfo),
                                        C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        L.rt_bvh_check_host.argtypes = [C.POINTER(SceneDesc), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.rt_bvh_top_check_host.argtypes = [C.POINTER(SceneDesc), C.c_uint32, C.c_uint32, C.c_void_p]
        L.rt_profile_reset.argtypes = [C.c_void_p]
        L.rt_profile_collect.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
        L.rt_test_unit.argtypes = [C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]
        L.rt_trace_stream_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.rt_build_photon_map.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
        L.rt_get_photons.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.rt_test_kd_order.argtypes = [C.c_int32, C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p, C.POINTER(C.c_double)]
        L.rt_owned_granules.argtypes = [C.POINTER(Params), C.c_uint32, C.POINTER(C.c_uint32)]
        L.rt_pack_owned_device.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_unpack_owned_device.argtypes = [C.c_void_p, C.POINTER(Params), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_group_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_int32), C.c_uint32, C.POINTER(Options),
                                      C.POINTER(C.c_void_p)]
        L.rt_group_destroy.argtypes = [C.c_void_p]
        L.rt_group_destroy.restype = None
        L.rt_group_size.argtypes = [C.c_void_p]
        L.rt_group_size.restype = C.c_uint32
        L.rt_group_uses_rccl.argtypes = [C.c_void_p]
        L.rt_group_ctx.argtypes = [C.c_void_p, C.c_uint32]
        L.rt_group_ctx.restype = C.c_void_p
        L.rt_group_set_photons.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.rt_group_render.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        _amd = L
    return _amd


def host():
    """librt_host.so (scene script, OFF loader, flattener, background, kd order)."""
    global _host
    if _host is None:
        amd()  # dependency, resolved through rpath as well
        path = os.path.join(LIB_DIR, "librt_host.so")
        if not os.path.exists(path):
            raise RtError(-1, "%s not built" % path)
        L = C.CDLL(path)
        L.rt_host_last_error.restype = C.c_char_p
        L.rt_host_scene_build.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        L.rt_host_scene_desc.argtypes = [C.c_void_p]
        L.rt_host_scene_desc.restype = C.POINTER(SceneDesc)
        L.rt_host_scene_free.argtypes = [C.c_void_p]
        L.rt_host_scene_free.restype = None
        L.rt_host_fill_background.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.rt_host_fill_background.restype = None
        L.rt_host_save_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.rt_host_kd_order.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        _host = L
    return _host


def _check(rc):
    if rc != RT_OK:
        raise RtError(rc, amd().rt_last_error().decode())


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Scene:
    """A flattened preset scene (host/ScenePresets.h) owned by librt_host."""

    def __init__(self, kind, width, height, mesh_dir=MESH_DIR):
        h = C.c_void_p()
        rc = host().rt_host_scene_build(kind.encode(), mesh_dir.encode(), width, height, C.byref(h))
        if rc != RT_OK:
            raise RtError(rc, host().rt_host_last_error().decode())
        self._h = h
        self.kind, self.width, self.height = kind, width, height
        self.desc_ptr = host().rt_host_scene_desc(h)
        self.desc = self.desc_ptr.contents

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _host is not None:
            _host.rt_host_scene_free(h)

    def __del__(self):
        try:  # at interpreter exit module globals / ctypes may already be torn down
            self.close()
        except Exception:
            pass

    def arrays(self):
        d = self.desc
        nv, nt, nm = d.n_vertices, d.n_triangles, d.n_meshes
        return dict(
            pos=np.ctypeslib.as_array(d.vertex_pos, (nv, 3)).copy(),
            nrm=np.ctypeslib.as_array(d.vertex_nrm, (nv, 3)).copy(),
            tri=np.ctypeslib.as_array(d.tri_vtx, (nt, 3)).copy(),
            tri_begin=np.ctypeslib.as_array(d.mesh_tri_begin, (nm + 1,)).copy(),
            vtx_begin=np.ctypeslib.as_array(d.mesh_vtx_begin, (nm + 1,)).copy(),
            materials=np.frombuffer(C.string_at(d.materials, 32 * nm), dtype="<f4").reshape(nm, 8).copy(),
            lights=np.frombuffer(C.string_at(d.lights, 84 * d.n_lights), dtype="<f4").reshape(d.n_lights, 21).copy(),
            camera=np.frombuffer(bytes(d.camera), dtype="<f4").reshape(4, 3).copy())


def background(width, height):
    bg = np.empty((height, width, 3), np.float32)
    host().rt_host_fill_background(_ptr(bg), width, height)
    return bg


def bvh_build_host(scene, leaf_max=0, threads=0):
    """The host BVH build alone (no GPU): (BvhInfo, digest, seconds)."""
    info, dig, sec = BvhInfo(), C.c_uint64(0), C.c_double(0)
    _check(amd().rt_bvh_build_host(C.byref(scene.desc), leaf_max, threads, C.byref(info), C.byref(dig), C.byref(sec)))
    return info, dig.value, sec.value


NODES_AUTO, NODES_F16, NODES_Q8 = 0, 1, 2


def bvh_check_host(scene, leaf_max=0, node_format=NODES_F16):
    """Host build + packing into the given device node format + structural check (no GPU).  Returns a dict of the
    shape numbers; raises RtError when the packed tree is not a valid tree over the scene's triangles."""
    out = np.zeros(8, np.uint32)
    est = np.zeros(2, np.float64)
    _check(amd().rt_bvh_check_host(C.byref(scene.desc), leaf_max, node_format, _ptr(out), _ptr(est)))
    return dict(nodes=int(out[0]), slots=int(out[1]), blocks=int(out[2]), depth=int(out[3]), added_nodes=int(out[4]),
                visits_float=float(est[0]), visits_packed=float(est[1]))


def bvh_top_check_host(scene, leaf_max=0, cutoff=1024):
    """The hybrid builder's host half (rtbvh::buildTop) alone, validated (no GPU).  Returns a dict of the shape numbers."""
    out = np.zeros(8, np.uint32)
    _check(amd().rt_bvh_top_check_host(C.byref(scene.desc), leaf_max, cutoff, _ptr(out)))
    return dict(top_nodes=int(out[0]), parts=int(out[1]), largest_part=int(out[2]), deepest_part=int(out[3]), depth_cap=int(out[4]),
                top_leaves=int(out[5]))


def kd_order(pos, dir_, weight=None):
    """kdtree::make_tree order (in place on copies); returns (pos, dir, weight)."""
    pos = np.ascontiguousarray(pos, np.float32).copy()
    dir_ = np.ascontiguousarray(dir_, np.float32).copy()
    w = None if weight is None else np.ascontiguousarray(weight, np.float32).copy()
    rc = host().rt_host_kd_order(_ptr(pos), _ptr(dir_), _ptr(w), len(pos))
    if rc != RT_OK:
        raise RtError(rc, host().rt_host_last_error().decode())
    return pos, dir_, w


class Context:
    """rt_ctx: the scene resident in HBM on one gfx950 device."""

    def __init__(self, scene, device=0, bvh_leaf_max=0, bvh_builder=0, node_format=0):
        self.scene = scene
        opt = Options()
        opt.device, opt.bvh_leaf_max, opt.bvh_builder, opt.node_format = device, bvh_leaf_max, bvh_builder, node_format
        h = C.c_void_p()
        _check(amd().rt_create(scene.desc_ptr, C.byref(opt), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            amd().rt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:  # at interpreter exit module globals / ctypes may already be torn down
            self.close()
        except Exception:
            pass

    def bvh_info(self):
        bi = BvhInfo()
        _check(amd().rt_bvh_info_get(self._h, C.byref(bi)))
        return bi

    def tune(self, probe_params, budget_seconds, max_probes=0):
        """rt_bvh_tune: measured-cost tuning of the host-built tree against the rays of `probe_params`."""
        rep = TuneReport()
        _check(amd().rt_bvh_tune(self._h, C.byref(probe_params), float(budget_seconds), int(max_probes), C.byref(rep)))
        return rep

    def bvh_export(self):
        bi = self.bvh_info()
        nodes = np.empty((bi.n_nodes, 16), np.uint32)
        tris = np.empty((bi.n_tri_records, 12), np.uint32)
        _check(amd().rt_bvh_export(self._h, _ptr(nodes), _ptr(tris)))
        return nodes, tris

    def set_photons(self, pos, dir_):
        pos = np.ascontiguousarray(pos, np.float32)
        dir_ = np.ascontiguousarray(dir_, np.float32)
        _check(amd().rt_set_photons(self._h, _ptr(pos), _ptr(dir_), len(pos)))

    def emit_photons(self, n_requested, seed=1):
        pos = np.zeros((max(n_requested, 1), 3), np.float32)
        dir_ = np.zeros_like(pos)
        w = np.zeros(max(n_requested, 1), np.float32)
        n = C.c_uint32()
        _check(amd().rt_emit_photons(self._h, n_requested, seed, _ptr(pos), _ptr(dir_), _ptr(w), C.byref(n)))
        return pos[:n.value].copy(), dir_[:n.value].copy(), w[:n.value].copy()

    def build_photon_map(self, n_requested, seed=1):
        """Emission + compaction + kd order on the device; returns (n_stored, [emit_ms, kd_ms])."""
        n = C.c_uint32()
        ms = (C.c_double * 2)()
        _check(amd().rt_build_photon_map(self._h, n_requested, seed, C.byref(n), ms))
        return n.value, [ms[0], ms[1]]

    def get_photons(self, cap):
        pos = np.zeros((max(cap, 1), 3), np.float32)
        dir_ = np.zeros_like(pos)
        w = np.zeros(max(cap, 1), np.float32)
        n = C.c_uint32()
        _check(amd().rt_get_photons(self._h, _ptr(pos), _ptr(dir_), _ptr(w), cap, C.byref(n)))
        return pos[:n.value].copy(), dir_[:n.value].copy(), w[:n.value].copy()

    def render(self, params, bg=None, want_accum=True):
        w, h = params.width, params.height
        out = np.empty((h, w, 3), np.float32) if bg is not None else None
        acc = np.empty((h, w, 4), np.float32) if want_accum else None
        st = Stats()
        bgc = None if bg is None else np.ascontiguousarray(bg, np.float32)
        _check(amd().rt_render(self._h, C.byref(params), _ptr(bgc), _ptr(out), _ptr(acc), C.byref(st)))
        return out, acc, st

    def render_passes(self, params, bg, accum_io):
        """rt_render_passes: integrate params.spp_begin/spp_count on top of accum_io (in place)
        and resolve the running estimate; returns (out_rgb, stats)."""
        w, h = params.width, params.height
        out = np.empty((h, w, 3), np.float32)
        st = Stats()
        bgc = np.ascontiguousarray(bg, np.float32)
        assert accum_io.dtype == np.float32 and accum_io.flags["C_CONTIGUOUS"] and accum_io.shape == (h, w, 4)
        _check(amd().rt_render_passes(self._h, C.byref(params), _ptr(bgc), _ptr(accum_io), _ptr(out), C.byref(st)))
        return out, st

    def render_device(self, params, d_accum_ptr, stream=0, stats=False):
        st = Stats() if stats else None
        _check(amd().rt_render_device(self._h, C.byref(params), C.c_void_p(d_accum_ptr), C.c_void_p(stream),
                                      C.byref(st) if stats else None))
        return st

    def resolve_device(self, width, height, spp, d_accum_ptr, d_bg_ptr, d_out_ptr, stream=0):
        _check(amd().rt_resolve_device(self._h, width, height, spp, C.c_void_p(d_accum_ptr), C.c_void_p(d_bg_ptr),
                                       C.c_void_p(d_out_ptr), C.c_void_p(stream)))

    def trace_stream_device(self, d_ray_o, d_ray_d, n, d_res, stream=0):
        _check(amd().rt_trace_stream_device(self._h, C.c_void_p(d_ray_o), C.c_void_p(d_ray_d), n, C.c_void_p(d_res),
                                            C.c_void_p(stream)))

    def pack_owned(self, params, d_accum_ptr, d_packed_ptr, stream=0):
        _check(amd().rt_pack_owned_device(self._h, C.byref(params), C.c_void_p(d_accum_ptr), C.c_void_p(d_packed_ptr),
                                          C.c_void_p(stream)))

    def unpack_owned(self, params, from_rank, d_packed_ptr, d_accum_ptr, stream=0):
        _check(amd().rt_unpack_owned_device(self._h, C.byref(params), from_rank, C.c_void_p(d_packed_ptr),
                                            C.c_void_p(d_accum_ptr), C.c_void_p(stream)))

    def profile_reset(self):
        _check(amd().rt_profile_reset(self._h))

    def profile_collect(self):
        ms, n = C.c_double(), C.c_uint32()
        _check(amd().rt_profile_collect(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def trace(self, rays, accel=ACCEL_BVH, kind=TRACE_CLOSEST):
        rays = np.ascontiguousarray(rays, RAY_DTYPE)
        hits = np.zeros(len(rays), HIT_DTYPE)
        _check(amd().rt_trace(self._h, _ptr(rays), len(rays), accel, kind, _ptr(hits)))
        return hits

    def knn(self, queries, k):
        q = np.ascontiguousarray(queries, np.float32)
        idx = np.zeros((len(q), k), np.uint32)
        dist = np.zeros((len(q), k), np.float32)
        vis = np.zeros(len(q), np.uint32)
        _check(amd().rt_knn(self._h, _ptr(q), len(q), k, _ptr(idx), _ptr(dist), _ptr(vis)))
        return idx, dist, vis


def kd_order_device(pos, depth_limit=-1, device=0):
    """rt_test_kd_order: permutation (tree slot -> input index) built on the GPU, and its ms."""
    pos = np.ascontiguousarray(pos, np.float32)
    perm = np.zeros(len(pos), np.uint32)
    ms = C.c_double()
    _check(amd().rt_test_kd_order(device, _ptr(pos), len(pos), depth_limit, _ptr(perm), C.byref(ms)))
    return perm, ms.value


def owned_granules(params, rank):
    """Number of 8x8-pixel granules `rank` owns in the tile-sharded frame `params` describes."""
    n = C.c_uint32()
    _check(amd().rt_owned_granules(C.byref(params), rank, C.byref(n)))
    return n.value


class Group:
    """rt_group: one process driving N devices (RCCL / peer copies inside librt_amd.so)."""

    def __init__(self, scene, devices, bvh_leaf_max=0, node_format=0):
        self.scene = scene
        opt = Options()
        opt.bvh_leaf_max, opt.node_format = bvh_leaf_max, node_format
        devs = (C.c_int32 * len(devices))(*devices)
        h = C.c_void_p()
        _check(amd().rt_group_create(scene.desc_ptr, devs, len(devices), C.byref(opt), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            amd().rt_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def size(self):
        return amd().rt_group_size(self._h)

    @property
    def uses_rccl(self):
        return bool(amd().rt_group_uses_rccl(self._h))

    def set_photons(self, pos, dir_):
        pos = np.ascontiguousarray(pos, np.float32)
        dir_ = np.ascontiguousarray(dir_, np.float32)
        _check(amd().rt_group_set_photons(self._h, _ptr(pos), _ptr(dir_), len(pos)))

    def render(self, params, bg=None, want_accum=True):
        w, h = params.width, params.height
        out = np.empty((h, w, 3), np.float32) if bg is not None else None
        acc = np.empty((h, w, 4), np.float32) if want_accum else None
        st = Stats()
        bgc = None if bg is None else np.ascontiguousarray(bg, np.float32)
        _check(amd().rt_group_render(self._h, C.byref(params), _ptr(bgc), _ptr(out), _ptr(acc), C.byref(st)))
        return out, acc, st


_UNIT_IO = {UNIT_ASIN: (np.float64, 1, np.float64, 1), UNIT_SINF: (np.float32, 1, np.float32, 1),
            UNIT_COSF: (np.float32, 1, np.float32, 1), UNIT_STREAM_SEED: (np.uint32, 4, np.uint32, 1),
            UNIT_TRIANGLE: (np.float32, 15, np.float32, 4), UNIT_BSDF: (np.float32, 17, np.float32, 3),
            UNIT_RAY_AT: (np.float32, 14, np.float32, 6), UNIT_LIGHT_EVAL: (np.float32, 24, np.float32, 3),
            UNIT_SAMPLERS: (np.uint32, 28, np.uint32, 12), UNIT_LIGHT_SAMPLE: (np.uint32, 22, np.uint32, 4),
            UNIT_POW: (np.float64, 1, np.float64, 2), UNIT_RECIP: (np.float32, 1, np.float32, 4)}


def unit(which, inp, out_init=None, device=0):
    """rt_test_unit: evaluate one device building block on n packed inputs."""
    it, iw, ot, ow = _UNIT_IO[which]
    a = np.ascontiguousarray(inp, it).reshape(-1, iw)
    out = np.zeros((len(a), ow), ot) if out_init is None else np.ascontiguousarray(out_init, ot).reshape(len(a), ow).copy()
    _check(amd().rt_test_unit(device, which, _ptr(a), _ptr(out), len(a)))
    return out


class ArrayScene:
    """A scene assembled from numpy arrays (tests: scaled / synthetic geometry)."""

    def __init__(self, pos, nrm, tri, tri_begin, vtx_begin, materials, lights, camera):
        self._keep = [np.ascontiguousarray(pos, np.float32), np.ascontiguousarray(nrm, np.float32),
                      np.ascontiguousarray(tri, np.uint32), np.ascontiguousarray(tri_begin, np.uint32),
                      np.ascontiguousarray(vtx_begin, np.uint32), np.ascontiguousarray(materials, np.float32),
                      np.ascontiguousarray(lights, np.float32), np.ascontiguousarray(camera, np.float32)]
        k = self._keep
        d = SceneDesc()
        d.n_meshes, d.n_vertices, d.n_triangles, d.n_lights = len(k[3]) - 1, len(k[0]), len(k[2]), len(k[6])
        d.vertex_pos = k[0].ctypes.data_as(C.POINTER(C.c_float))
        d.vertex_nrm = k[1].ctypes.data_as(C.POINTER(C.c_float))
        d.tri_vtx = k[2].ctypes.data_as(C.POINTER(C.c_uint32))
        d.mesh_tri_begin = k[3].ctypes.data_as(C.POINTER(C.c_uint32))
        d.mesh_vtx_begin = k[4].ctypes.data_as(C.POINTER(C.c_uint32))
        d.materials = k[5].ctypes.data_as(C.POINTER(Material))
        d.lights = k[6].ctypes.data_as(C.POINTER(Light))
        C.memmove(C.byref(d.camera), k[7].ctypes.data, 48)
        self.desc = d
        self.desc_ptr = C.pointer(d)

    def arrays(self):
        k = self._keep
        return dict(pos=k[0], nrm=k[1], tri=k[2], tri_begin=k[3], vtx_begin=k[4], materials=k[5], lights=k[6], camera=k[7])
