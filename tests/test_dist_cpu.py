"""N>1 path on CPU: gloo ranks shard the frame by tiles exactly as bench.py's ranks do
(rt_params.rank/world/tile), the oracle stands in for the GPU kernel, and the frame is
assembled on rank 0 both ways: FrameAssembler (what bench.py runs: one gather of each
rank's OWNED granules) and the reference exchange (a full-frame SUM reduce).  Either way
the assembled frame must be bit-identical to the single-rank frame."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import pyrt

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, os.path.join(sys.argv[1], "ray-tracing-engine_amd")); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
    import numpy as np, torch
    import pyrt, orc
    from pyrt import dist as rdist
    rank, world, local = rdist.init_from_env("gloo")
    w, h, spp, tile = 45, 37, 3, int(sys.argv[3])
    scene = pyrt.Scene("cubes", w, h)
    p = pyrt.make_params(w, h, spp, seed=4, rank=rank, world=world, tile=tile)
    _, acc, st = orc.render(scene, p, math_mode=orc.MATH_DET, threads=2)
    t = torch.from_numpy(acc.copy())
    owned = int((t.abs().sum(-1) > 0).sum())
    g = torch.from_numpy(acc.copy())
    rdist.FrameAssembler(None, p, rank, world, "cpu").assemble(g)   # what bench.py runs
    rdist.reduce_frame(t, dst=0)                                    # reference exchange
    if rank == 0:
        assert torch.equal(g.view(torch.int32), t.view(torch.int32)), "gather and reduce disagree"
    tot = rdist.sum_over_ranks([st.rays_closest + st.rays_shadow, owned], "cpu")
    mx = rdist.max_over_ranks(float(rank), "cpu")
    if rank == 0:
        np.save(sys.argv[2], t.numpy())
        print("RAYS", tot[0], "OWNED", tot[1], "MAXRANK", mx)
    rdist.barrier()
    rdist.shutdown()
""")


@pytest.mark.parametrize("world,tile", [(2, 8), (2, 16), (3, 8)])
def test_tile_sharded_frame_equals_single_rank(tmp_path, world, tile):
    import orc
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "frame.npy"
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(29600 + world * 10 + tile), str(script),
                        pyrt.ROOT, str(out), str(tile)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    w, h, spp = 45, 37, 3
    scene = pyrt.Scene("cubes", w, h)
    _, full, st = orc.render(scene, pyrt.make_params(w, h, spp, seed=4), math_mode=orc.MATH_DET)
    got = np.load(out)
    assert np.array_equal(got.view(np.uint32), full.view(np.uint32))
    line = [l for l in r.stdout.splitlines() if l.startswith("RAYS")][0].split()
    assert int(line[1]) == st.rays_closest + st.rays_shadow
    assert float(line[5]) == world - 1


def test_ownership_partitions_the_frame():
    """owner(tile) = (tx + ty) mod world: every pixel has exactly one owner and the
    shares are balanced (the diagonal pattern avoids column stripes)."""
    w, h = 1024, 1024
    for world, tile in ((2, 32), (4, 32), (8, 32), (8, 8), (3, 64)):
        ys, xs = np.mgrid[0:h, 0:w]
        owner = ((xs // tile) + (ys // tile)) % world
        counts = np.bincount(owner.ravel(), minlength=world)
        assert counts.sum() == w * h and counts.max() - counts.min() <= 0.02 * w * h / world


def test_owned_granule_order_matches_the_library():
    """pyrt.dist.owned_granule_index (numpy, used on CPU tensors) and rt_owned_granules (the
    library's count, host-only) agree; the granules of all ranks partition the image."""
    from pyrt import dist as rdist
    for w, h, world, tile in ((45, 37, 2, 8), (1024, 1024, 8, 32), (96, 64, 3, 16), (17, 9, 5, 8)):
        seen = np.zeros(w * h, np.int32)
        for r in range(world):
            p = pyrt.make_params(w, h, 1, rank=r, world=world, tile=tile)
            ix = rdist.owned_granule_index(w, h, r, world, tile)
            assert len(ix) == pyrt.owned_granules(p, r)
            np.add.at(seen, ix[ix >= 0], 1)
            assert (np.diff(ix[:, 0]) > 0).all()  # row-major granule order
        assert (seen == 1).all()


EXCHANGE8 = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, os.path.join(sys.argv[1], "ray-tracing-engine_amd"))
    import numpy as np, torch
    import pyrt
    from pyrt import dist as rdist
    rank, world, local = rdist.init_from_env("gloo")
    w = h = int(sys.argv[2]); tile = int(sys.argv[3])
    p = pyrt.make_params(w, h, 1, rank=rank, world=world, tile=tile)
    # what a rank's render leaves behind: its OWN pixels filled (here: a function of the pixel), zeros elsewhere
    ys, xs = np.mgrid[0:h, 0:w]
    mine = ((xs // tile) + (ys // tile)) % world == rank
    pattern = np.stack([(ys * w + xs).astype(np.float32) + c * 0.25 for c in range(4)], -1)
    acc = torch.from_numpy(np.where(mine[..., None], pattern, np.float32(0)).astype(np.float32))
    fa = rdist.FrameAssembler(None, p, rank, world, "cpu")
    share = fa.counts[rank] / (w * h / 64.0 / world)
    assert abs(share - 1.0) <= 0.01, "rank %d owns %.4f of an even share" % (rank, share)
    assert fa.packed.shape[0] == fa.counts[rank] * 64  # exact buffers, nothing padded to the largest share
    for _ in range(2):  # (a second frame through the same assembler)
        g = acc.clone()
        fa.assemble(g)
    if rank == 0:
        assert np.array_equal(g.numpy().view(np.uint32), pattern.view(np.uint32)), "assembled frame is not the whole frame"
        print("ASSEMBLED", fa.counts)
    rdist.barrier()
    rdist.shutdown()
""")


def test_eight_rank_exchange_on_config4_shape(tmp_path):
    """BASELINE config 4's frame (2048 x 2048, 32-pixel ownership tiles) split over EIGHT gloo ranks — the driver's
    --gpus 8 shape, rehearsed without GPUs: every rank's granule count within 1 % of an even share, exact-size
    point-to-point buffers, and the frame assembled on rank 0 from synthetic per-rank accumulators is whole."""
    script = tmp_path / "x8.py"
    script.write_text(EXCHANGE8)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
                        "127.0.0.1", "--master-port", "29788", str(script), pyrt.ROOT, "2048", "32"], capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "ASSEMBLED" in r.stdout
