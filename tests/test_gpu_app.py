"""The host layer end to end on the GPU: our RayTracer application and the
reference's OWN Main.cpp compiled unchanged against our host headers
(oracle/_ref/RayTracer_dropin) must write the PPM the oracle predicts for pixel
RNG seed 1, byte for byte."""
import os
import subprocess

import numpy as np
import pytest

import orc
import pyrt

pytestmark = pytest.mark.gpu

APP = os.path.join(pyrt.ROOT, "ray-tracing-engine_amd", "bin", "RayTracer")
DROPIN = os.path.join(pyrt.ROOT, "oracle", "_ref", "RayTracer_dropin")


def _expected(kind, w, h, spp, mode, nph=0, k=0):
    s = pyrt.Scene(kind, w, h)
    bg = pyrt.background(w, h)
    p = pyrt.make_params(w, h, spp, mode=mode, seed=1)
    ext = None
    if nph:
        ph, _, _ = orc.emit_photons(s, nph, pyrt.RNG_PIXEL, seed=1, math_mode=orc.MATH_DET)
        ext = orc.kd_build(ph)
        p.use_photons, p.k, p.photons_requested = 1, k, nph
    out, _, _ = orc.render(s, p, math_mode=orc.MATH_DET, bg=bg, ext_photons=ext)
    return orc.ppm_bytes(out)


@pytest.mark.parametrize("args,exp", [
    (["-width", "64", "-height", "48", "-m", "1", "-N", "4"], ("cubes", 64, 48, 4, 1)),
    (["-w", "40", "-h", "40", "-m", "0", "-n", "3", "-scene", "lowres"], ("lowres", 40, 40, 3, 0)),
    (["-width", "48", "-height", "48", "-m", "0", "-N", "2", "-p", "3000", "-k", "10"], ("cubes", 48, 48, 2, 0, 3000, 10)),
    (["-width", "32", "-height", "32", "-m", "1", "-N", "2", "-p", "2000", "-k", "5", "-accel", "1"], ("cubes", 32, 32, 2, 1, 2000, 5)),
])
def test_application_writes_the_predicted_ppm(tmp_path, args, exp):
    out = tmp_path / "o.ppm"
    r = subprocess.run([APP] + args + ["-meshdir", pyrt.MESH_DIR, "-o", str(out)], cwd=tmp_path, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert out.read_bytes() == _expected(*exp)
    assert (tmp_path / "update.ppm").read_bytes() == out.read_bytes()  # Renderer.cpp:268-269
    assert "Mrays/s" in r.stdout
    if "-p" in args:
        # Main.cpp:216 + SURVEY App. A.9: the cloud saved BEFORE render() is empty
        assert "POINTS 0" in (tmp_path / "pointcloud.pcd").read_text()


def test_application_usage_errors(tmp_path):
    r = subprocess.run([APP, "-width"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 1 and "Missing argument" in r.stderr and "USAGE" in r.stderr
    r = subprocess.run([APP, "-bogus", "1"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 1 and "Unknown argument <-bogus>" in r.stderr
    r = subprocess.run([APP, "-scene", "lowres", "-meshdir", str(tmp_path), "-N", "1"], capture_output=True, text=True,
                       cwd=tmp_path)
    assert r.returncode == 1 and "Error loading OFF file" in r.stderr


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="oracle/_ref/RayTracer_dropin is built only where /root/reference exists")
def test_reference_main_unchanged_runs_on_the_gpu(tmp_path):
    """reference source/Main.cpp (compiled as is) + our Renderer::render: config 1 of
    BASELINE.json, hard-coded ../meshes paths and all."""
    (tmp_path / "build").mkdir()
    os.symlink(pyrt.MESH_DIR, tmp_path / "meshes")
    r = subprocess.run([DROPIN, "-width", "256", "-height", "256", "-m", "1", "-N", "8"], cwd=tmp_path / "build",
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    got = (tmp_path / "build" / "output.ppm").read_bytes()
    assert got == _expected("cubes", 256, 256, 8, 1)
    # and it is the same picture as the reference's up to Monte Carlo noise
    ref = np.array(open(os.path.join(pyrt.ROOT, "tests", "golden", "ppm", "cubes_64_m1_N4.ppm")).read().split()[4:], float)
    mine = np.array(got.split()[4:], float).reshape(256, 256, 3)
    mine64 = mine.reshape(64, 4, 64, 4, 3).mean((1, 3))
    assert abs(mine64.mean() - ref.mean()) / ref.mean() < 0.03


def test_bench_two_ranks_rehearsal_on_one_gpu(tmp_path):
    """bench.py's N-rank flow (tile sharding, reduce onto rank 0, resolve, max-over-ranks
    timing, summed counters) rehearsed with 2 gloo ranks sharing this GPU; the ray count
    must equal the 1-rank count (every pixel is integrated by exactly one rank)."""
    import json
    import sys
    env = dict(os.environ, RT_DIST_BACKEND="gloo", RT_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29711", os.path.join(pyrt.ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--workload", "C1"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    two = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    r1 = subprocess.run([sys.executable, os.path.join(pyrt.ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--workload", "C1",
                         "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, cwd=tmp_path)
    one = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][-1])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert two["cpu_baseline"]["value"] > 0  # (rank 0 times the CPU beside it at every N)
    assert two["config"]["rays_per_frame"] == one["config"]["rays_per_frame"] == 5526901
    for k in ("metric", "value", "unit", "ms_per_step", "roofline", "dtype", "data", "vs_baseline", "higher_is_better"):
        assert k in two and k in one
    assert one["roofline"]["bound"] == "valu_issue" and one["roofline"]["hbm"]["bound"] == "hbm"
    if one["roofline"]["frac"] is not None:  # (None only when rocprofv3 is unavailable)
        assert 0 < one["roofline"]["frac"] <= 1 and 0 < one["roofline"]["hbm"]["frac"] <= 1


def test_bench_spawns_its_own_launcher(tmp_path):
    """`python bench.py --gpus N` with no launcher in the environment starts
    torch.distributed.run itself (as a child, before touching the GPU) and rank 0 still prints the
    one JSON line.  Rehearsed with 3 gloo ranks sharing this GPU."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(RT_DIST_BACKEND="gloo", RT_SHARE_GPU="1", MASTER_PORT="29733")
    r = subprocess.run([sys.executable, os.path.join(pyrt.ROOT, "bench.py"), "--gpus", "3", "--steps", "2", "--warmup", "1",
                        "--workload", "C1"], capture_output=True, text=True, env=env, timeout=900, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["config"]["rays_per_frame"] == 5526901 and d["value"] > 0
    assert "owned tiles sent" in d["config"]["parallelism"]
    # the N > 1 line is complete and attributable — a per-rank roofline measured by
    # rank 0's own PMC child passes on its shard, the CPU baseline, and where the frame time goes
    roof, cfg = d["roofline"], d["config"]
    assert roof["frac"] is not None and 0 < roof["frac"] <= 1 and "rank 0's shard of 3" in roof["traffic_source"]
    assert roof["valu_issue"]["frac"] is not None and roof["traffic"] is not None and "1/3" in roof["scope"]
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] == 1
    assert cfg["assemble_ms"] > 0 and cfg["kernel_ms_max_over_ranks"] >= cfg["kernel_ms_min_over_ranks"] > 0
    assert "sends of each rank's owned 8x8-pixel granules" in cfg["exchange"]
    # round 4: the all-cores CPU figures say how many threads the oracle's OpenMP loop really ran on
    assert d["cpu_baseline"]["all_cores_threads"] >= 1 and d["cpu_baseline"]["port_bvh_all_cores_value"] > d["cpu_baseline"]["port_bvh_value"]


def test_progressive_update_ppm_matches_the_reference_semantics(tmp_path):
    """-progress 1: an update.ppm after every pass (Renderer.cpp:261-269).  The final
    image is the one-launch image, and stopping after pass j gives the j-sample estimate
    resolved with j (checked against the oracle for j = 2 of N = 3... via a 2-sample run
    being a different jitter grid, so instead: final image identical, file rewritten)."""
    base = ["-width", "40", "-height", "32", "-m", "1", "-N", "5", "-meshdir", pyrt.MESH_DIR]
    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    ra = subprocess.run([APP] + base + ["-o", "o.ppm"], cwd=tmp_path / "a", capture_output=True, text=True, timeout=120)
    rb = subprocess.run([APP] + base + ["-o", "o.ppm", "-progress", "2"], cwd=tmp_path / "b", capture_output=True, text=True,
                        timeout=120)
    assert ra.returncode == 0 and rb.returncode == 0, ra.stderr + rb.stderr
    assert (tmp_path / "a" / "o.ppm").read_bytes() == (tmp_path / "b" / "o.ppm").read_bytes() == _expected("cubes", 40, 32, 5, 1)
    assert (tmp_path / "b" / "update.ppm").read_bytes() == (tmp_path / "b" / "o.ppm").read_bytes()
