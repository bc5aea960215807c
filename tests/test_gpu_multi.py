"""Multi-GPU behind the C ABI (VERDICT r01 missing 4): rt_group drives N 'ranks' from one
process, each integrates its tiles, only OWNED granules travel to rank 0, rank 0
resolves.  On the one-GPU test box the ranks share device 0 (peer-copy exchange; the RCCL
exchange needs distinct devices, its loading and communicator set-up are still checked).
Every assembled frame must equal the single-context frame bit for bit."""
import os
import subprocess

import numpy as np
import pytest

import orc
import pyrt
from pyrt import dist as rdist

pytestmark = pytest.mark.gpu

APP = os.path.join(pyrt.ROOT, "ray-tracing-engine_amd", "bin", "RayTracer")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("kind,w,h,spp,mode,ranks", [("lowres", 200, 136, 6, 1, 2), ("cubes", 45, 37, 5, 1, 3),
                                                     ("hires", 256, 256, 4, 1, 4), ("cubes", 64, 64, 3, 0, 6),
                                                     ("lowres", 256, 192, 2, 1, 8)])  # (8: the driver's --gpus 8 shape)
def test_group_frame_equals_single_context_frame(kind, w, h, spp, mode, ranks):
    s = pyrt.Scene(kind, w, h)
    bg = pyrt.background(w, h)
    ctx = pyrt.Context(s)
    p = pyrt.make_params(w, h, spp, mode=mode, seed=11)
    ref_out, ref_acc, ref_st = ctx.render(p, bg)
    ctx.close()
    g = pyrt.Group(s, [0] * ranks)
    assert g.size == ranks and not g.uses_rccl  # ranks share a device: peer copies
    for tile in (0, 8, 16):
        p.tile = tile
        out, acc, st = g.render(p, bg)
        assert np.array_equal(bits(acc), bits(ref_acc)), (ranks, tile)
        assert np.array_equal(bits(out), bits(ref_out))
        assert (st.rays_closest, st.rays_shadow, st.samples) == (ref_st.rays_closest, ref_st.rays_shadow, w * h * spp)
    g.close()


def test_group_create_fails_cleanly_when_a_later_rank_has_no_device():
    """a context at rank >= 1 that cannot be created (device ordinal out of range)
    must surface as RtError — the failure path used to walk frame vectors not yet sized."""
    s = pyrt.Scene("cubes", 32, 32)
    for devs in ([0, 99], [0, 0, 99], [99]):
        with pytest.raises(pyrt.RtError) as e:
            pyrt.Group(s, devs)
        assert e.value.code == 2 and "out of range" in str(e.value)  # RT_ERR_NO_DEVICE
    g = pyrt.Group(s, [0, 0])  # the library is still usable afterwards
    assert g.size == 2
    g.close()


def test_group_photon_frame_and_oracle():
    w, h, spp, nph, k = 48, 40, 2, 3000, 5
    s = pyrt.Scene("cubes", w, h)
    g = pyrt.Group(s, [0, 0])
    one = pyrt.Context(s)
    pos, dir_, wt = one.emit_photons(nph, seed=2)
    kp, kd_, kw = pyrt.kd_order(pos, dir_, wt)
    g.set_photons(kp, kd_)
    p = pyrt.make_params(w, h, spp, mode=pyrt.MODE_RAY, seed=23, use_photons=1, k=k, photons_requested=nph)
    _, acc, st = g.render(p)
    ext = np.concatenate([kp, kd_, kw[:, None]], 1)
    _, ref_acc, ref_st = orc.render(s, p, math_mode=orc.MATH_DET, ext_photons=ext)
    assert np.array_equal(bits(acc), bits(ref_acc)) and st.knn_queries == ref_st.knn_queries
    one.close()
    g.close()


def test_rccl_loads_and_initialises():
    """The RCCL side of rt_group (dlopen of librccl.so, ncclCommInitAll / ncclCommDestroy) on
    the devices this box has; with one device the exchange loop has nothing to move."""
    s = pyrt.Scene("cubes", 32, 32)
    os.environ["RT_GROUP_FORCE_RCCL"] = "1"
    try:
        g = pyrt.Group(s, [0])
        assert g.uses_rccl
        ctx = pyrt.Context(s)
        p = pyrt.make_params(32, 32, 2, seed=3)
        _, a, _ = g.render(p)
        _, b, _ = ctx.render(p)
        assert np.array_equal(bits(a), bits(b))
        ctx.close()
        g.close()
    finally:
        del os.environ["RT_GROUP_FORCE_RCCL"]


def test_pack_unpack_kernels_follow_the_documented_order():
    """rt_pack_owned_device / rt_unpack_owned_device against the numpy statement of the
    granule order (pyrt.dist.owned_granule_index), odd image size."""
    import torch
    w, h, world, tile = 45, 37, 3, 16
    s = pyrt.Scene("cubes", w, h)
    ctx = pyrt.Context(s)
    frame = torch.arange(w * h * 4, dtype=torch.float32, device="cuda").reshape(h, w, 4) + 1
    total = torch.zeros_like(frame)
    for r in range(world):
        p = pyrt.make_params(w, h, 1, rank=r, world=world, tile=tile)
        n = pyrt.owned_granules(p, r)
        packed = torch.full((n * 64, 4), -1.0, dtype=torch.float32, device="cuda")
        ctx.pack_owned(p, frame.data_ptr(), packed.data_ptr(), torch.cuda.current_stream().cuda_stream)
        ix = rdist.owned_granule_index(w, h, r, world, tile).reshape(-1)
        want = np.zeros((n * 64, 4), np.float32)
        want[ix >= 0] = frame.cpu().numpy().reshape(-1, 4)[ix[ix >= 0]]
        assert np.array_equal(packed.cpu().numpy(), want)
        ctx.unpack_owned(p, r, packed.data_ptr(), total.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(total, frame)  # the ranks' granules partition the frame
    ctx.close()


def test_application_with_devices_list(tmp_path):
    """`RayTracer -devices 0,0,0`: Renderer::render through rt_group writes the same PPM."""
    from test_gpu_app import _expected
    out = tmp_path / "o.ppm"
    r = subprocess.run([APP, "-width", "72", "-height", "48", "-m", "1", "-N", "3", "-scene", "lowres", "-devices", "0,0,0",
                        "-meshdir", pyrt.MESH_DIR, "-o", str(out)], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "on 3 GPUs (peer copies)" in r.stdout
    assert out.read_bytes() == _expected("lowres", 72, 48, 3, 1)
