#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X path-tracing hot path.

Default workload (BASELINE.json configs[1]): 1024x1024, -m 1 (path), -N 128 spp, Cornell
box with meshes/example_low_res.off in slot 3 (1,222 triangles), pixel-RNG seed 1.
A "step" is one whole frame: zero the accumulator, integrate every sample of every
owned pixel (one HIP launch per rank), assemble the frame over ranks (N>1), resolve on
rank 0.  Metric: Mrays/s = (closest-hit + shadow rays actually cast) / wall time, whole job.

  python bench.py --gpus N --steps K --warmup W
With N>1 and no WORLD_SIZE in the environment the script starts its own
`torch.distributed.run` (one rank per GPU, RCCL) as a child process, before anything
touches the GPU.  The frame is tile-sharded over ranks, i.e. total work is fixed:
"scaling": "strong".

The JSON line carries
  roofline      the roof that BINDS the dominant kernel: VALU issue rate for the cache-
                resident scenes (C1-C4; SQ_INSTS_VALU of the timed kernel / its time against
                1024 SIMDs x 2.4 GHz / 2 cycles per wave64 op), HBM bytes for the 1M-triangle
                scene (C5).  Counters are measured IN THIS RUN by rocprofv3 child passes
                (N=1 only; --no-pmc skips them and falls back to profiles/, marked as such).
  cpu_baseline  the reference's own code timed on the host beside it, plus the CPU
                restatement with and without a BVH and the GPU exhaustive kernel, so that
                the hardware gain and the algorithmic gain can be read separately.
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-engine_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SIMDS = 256 * 4           # 256 CUs x 4 SIMD-32
CLOCK_GHZ = 2.4           # max shader clock (same guide)
VALU_PEAK_GINSTR = SIMDS * CLOCK_GHZ / 2.0  # one wave64 VALU op per 2 cycles per SIMD -> 1228.8 G wave-instr/s

WORKLOADS = {
    # name: (scene, width, height, spp, mode, photons, k)
    "C2": ("lowres", 1024, 1024, 128, 1, 0, 0),
    "C1": ("cubes", 256, 256, 8, 1, 0, 0),
    "C3": ("cubes", 1024, 1024, 16, 0, 50000, 10),
    "C4": ("hires", 2048, 2048, 512, 1, 0, 0),
    "C5": ("stress", 1024, 1024, 256, 1, 0, 0),
    # supplementary: 8x the C5 lattice (8M triangles, ~0.9 GB of nodes + triangle records: beyond
    # the 256 MB Infinity Cache, so FETCH_SIZE is HBM traffic proper)
    "C5x8": ("stress8", 1024, 1024, 32, 1, 0, 0),
}
HBM_BOUND = ("C5", "C5x8")
# The vector L1 (TCP) of a CU serves a bounded number of DIVERGENT 16-byte requests per clock, whatever their hit
# level: measured by tools/microbench/gather.hip (profiles/r03_gather_microbench.json): 0.93 per CU-clock when every
# request hits L1, 0.72 when they miss to L2 (3.9 when all lanes of a wave ask for the same address).  BVH traversal
# beyond the LDS-resident scenes is bound by exactly this unit.
TCP_DIVERGENT_PEAK = 0.93
TCP_DIVERGENT_PEAK_L1_MISS = 0.72
NUM_CUS = 256


def kernel_source_hash():
    """Identity of the device code the counters belong to (profiles/*.json carry it)."""
    h = hashlib.sha1()
    for rel in ("ray-tracing-engine_amd/csrc/rt_kernels.hip", "ray-tracing-engine_amd/csrc/rt_device.h",
                "ray-tracing-engine_amd/csrc/rt_kernels.h", "include/rt_pixelmode.h"):
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()[:16]


def metric_text(wl):
    kind, w, h, spp, mode, nph, k = WORKLOADS[wl]
    what = "path trace" if mode == 1 else "ray trace"
    if nph:
        what += " + photon map"
    return "Mrays/s (primary+secondary) at %dx%d/%dspp %s" % (w, h, spp, what)


# ----------------------------------------------------------------------------- PMC passes
PMC_PASSES = [["SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVES",
               "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"],
              ["FETCH_SIZE"], ["WRITE_SIZE"], ["GRBM_GUI_ACTIVE"],
              ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum"]]


def is_timed_render_kernel(name):
    """The integrate kernel of the TIMED launches (not the one STATS pass)."""
    for fam in ("k_render_persist<", "k_render_persist5<"):  # (the 16-wave workgroups, and the 4-wave ones of the 20-wave plan)
        if fam in name:
            return name.split(fam)[1].split(">")[0].split(",")[0].strip() in ("false", "0")
    if "k_render<" in name:
        return name.split("k_render<")[1].split(">")[0].split(",")[3].strip() in ("false", "0")
    return False


def pmc_measure(args, rank=0, world=1):
    """rocprofv3 --pmc child passes over ONE frame of the same workload — at N > 1 over rank `rank`'s
    SHARD of it, on device 0 of this process's view — (program after `--` is python3 itself;
    --kernel-trace only).  Returns {counter: value per launch} or None."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    out = {}
    base = tempfile.mkdtemp(prefix="rt_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for i, group in enumerate(PMC_PASSES):
            d = os.path.join(base, "p%d" % i)
            cmd = [rocprof, "--kernel-trace", "--pmc"] + group + ["--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", args.workload, "--accel", args.accel]
            if args.spp:
                cmd += ["--spp", str(args.spp)]
            # the child must run THIS run's kernel configuration (integrator, samples per wave, leaf size)
            cmd += ["--integrator", args.integrator, "--lpp", str(args.lpp), "--leaf", str(args.leaf), "--tune-probes", str(args.tune_probes),
                    "--pmc-rank", str(rank), "--pmc-world", str(world)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
            rows = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if is_timed_render_kernel(row["Kernel_Name"]):
                        rows.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            if not rows:
                return None, "pass %d produced no counters (rc %d): %s" % (i, r.returncode, (r.stderr or r.stdout)[-300:])
            for c, v in rows.items():
                out[c] = sum(v) / len(v)
    except Exception as e:  # the counters must never cost the bench line
        return None, "%s: %s" % (type(e).__name__, e)
    finally:
        shutil.rmtree(base, ignore_errors=True)
    return out, "rocprofv3 --pmc child passes of this run (1 frame each%s)" % (
        "" if world == 1 else "; rank %d's shard of %d" % (rank, world))


def pmc_child(args):
    """One untimed frame of the workload: what the rocprofv3 passes wrap."""
    import torch
    import pyrt
    kind, w, h, spp, mode, nph, k = WORKLOADS[args.workload]
    spp = args.spp or spp
    scene = pyrt.Scene(kind, w, h)
    ctx = pyrt.Context(scene, device=0, bvh_leaf_max=args.leaf)
    tune_tree(ctx, pyrt, args, mode, nph)  # (the same deterministic tuning as the timed run's)
    params = build_params(ctx, pyrt, args, w, h, spp, mode, nph, k, args.pmc_rank, args.pmc_world)
    accum = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
    ctx.render_device(params, accum.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ctx.close()


def tune_tree(ctx, pyrt, args, mode, nph):
    """--tune-probes N: rt_bvh_tune on 128x128x1 probe frames of this workload's camera and integrator."""
    if not args.tune_probes or nph or args.accel != "bvh":
        return None
    probe = pyrt.make_params(128, 128, 1, mode=mode, seed=7)
    rep = ctx.tune(probe, 600.0, args.tune_probes)
    return {"probes": rep.probes, "changes_kept": rep.accepted, "probe_cost_before": rep.cost_before,
            "probe_cost_after": rep.cost_after, "seconds": rep.seconds,
            "note": "rt_bvh_tune: subtree moves / child slot orders kept only where a probe frame's node visits + 1.5 x triangle tests fell; same image"}


# ----------------------------------------------------------------------------- CPU baseline
def effective_cores():
    """Cores this process can actually keep busy: the affinity mask, capped by the cgroup's CPU quota (a GPU box of this pool
    shows 256 hardware threads and grants its one-GPU share, 16 cores' worth: 128 OpenMP threads then scale 13.5 x, not 128 x)."""
    n = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    eff = n if quota is None else max(1, min(n, int(quota + 0.5)))
    return eff, n, quota


def cpu_baseline(ctx, scene_kind, mode, device, full=False):
    """The REAL reference (oracle/_ref/ref_harness = reference sources + our driver), timed
    single-threaded on a bounded sample of the same scene and mode (96x96, 8 spp, ~10 s:
    brute force costs ~0.13 ms per sample on the low-res scene; BASELINE config 1 is the same
    command at 256x256 — 7x the samples, same rate).  Beside it, for the decomposition of the
    GPU/CPU ratio: the CPU restatement (oracle) single-threaded with the exhaustive loop and
    with its own BVH on config 1's full 256x256x8, both again on all host cores, and the GPU's
    exhaustive kernel.  hardware gain = like for like (loop/loop, BVH/BVH); the rest is the
    algorithm."""
    import numpy as np
    import orc
    import pyrt
    w = h = 96
    n = 8
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    scene = pyrt.Scene(scene_kind, w, h)
    p = pyrt.make_params(w, h, n, mode=mode, rng_mode=pyrt.RNG_LEGACY)
    t0 = time.perf_counter()
    _, _, st = orc.render(scene, p, math_mode=orc.MATH_LIBM)
    t_port = time.perf_counter() - t0
    rays = st.rays_closest + st.rays_shadow
    out = {"unit": "Mrays/s", "cores": 1,
           "sample": "%s scene, %dx%d, -m %d -N %d, legacy RNG seed 1 (%d rays; BASELINE config 1 is this command at "
                     "256x256: same rate, 7x the work)" % (scene_kind, w, h, mode, n, rays),
           "port_loop_value": rays / t_port / 1e6}
    # the restatement through its CPU BVH, single thread, on config 1's full size
    big = pyrt.Scene(scene_kind, 256, 256)
    pb = pyrt.make_params(256, 256, 8, mode=mode, rng_mode=pyrt.RNG_LEGACY)
    t0 = time.perf_counter()
    _, _, sb = orc.render(big, pb, math_mode=orc.MATH_LIBM, accel=orc.ACCEL_OBVH)
    t_bvh = time.perf_counter() - t0
    out["port_bvh_value"] = (sb.rays_closest + sb.rays_shadow) / t_bvh / 1e6
    out["port_bvh_sample"] = "%s scene, 256x256, -m %d -N 8 (BASELINE config 1's size), oracle CPU BVH, 1 thread" % (scene_kind, mode)
    # all host cores (pixel RNG mode: independent pixels, OpenMP over runs of 64 pixels; the reference itself cannot run
    # multi-threaded: one global RNG).  "All cores" = the cores this process may use: the affinity mask capped by the
    # cgroup's CPU quota (effective_cores).  The samples are sized for them: a 512 x 512 frame is 4,096 work items (>= 16 per
    # thread on a 256-thread box) and the spp is chosen for about 4 s at half-linear scaling, so every thread has work for the
    # whole measurement; the thread count OpenMP actually used comes back from the oracle.
    threads, hw_threads, quota = effective_cores()
    per_sample = rays / float(w * h * n)
    big_w = 512

    def all_cores(rate1, accel):
        # a one-spp probe frame first: the usable share of the host is not always visible (a CPU quota may be enforced outside
        # this process' cgroup files), so the sample is sized from the rate the probe actually reaches, for about 4 s
        sc = pyrt.Scene(scene_kind, big_w, big_w)
        probe_spp = 1
        t0 = time.perf_counter()
        _, _, sp = orc.render(sc, pyrt.make_params(big_w, big_w, probe_spp, mode=mode, rng_mode=pyrt.RNG_PIXEL), math_mode=orc.MATH_DET,
                              threads=threads, accel=accel)
        probe_rate = (sp.rays_closest + sp.rays_shadow) / max(time.perf_counter() - t0, 1e-6)
        spp = int(min(64, max(1, round(probe_rate * 4.0 / (big_w * big_w * per_sample)))))
        pa = pyrt.make_params(big_w, big_w, spp, mode=mode, rng_mode=pyrt.RNG_PIXEL)
        t0 = time.perf_counter()
        _, _, sa = orc.render(sc, pa, math_mode=orc.MATH_DET, threads=threads, accel=accel)
        dt = time.perf_counter() - t0
        return (sa.rays_closest + sa.rays_shadow) / dt / 1e6, int(sa.reserved[0]), "%dx%dx%d spp, %.1f s" % (big_w, big_w, spp, dt)

    out["port_loop_all_cores_value"], used1, out["port_loop_all_cores_sample"] = all_cores(out["port_loop_value"], orc.ACCEL_LOOP)
    out["port_bvh_all_cores_value"], used2, out["port_bvh_all_cores_sample"] = all_cores(out["port_bvh_value"], orc.ACCEL_OBVH)
    out["all_cores_threads"] = min(used1, used2)
    out["all_cores_threads_available"] = hw_threads
    out["cpu_quota_cores"] = quota  # (cgroup CPU quota of this process if one is visible; None = none visible)
    # what the host really gave: the all-cores rate over the one-thread rate of the same code
    out["all_cores_speedup_loop"] = out["port_loop_all_cores_value"] / out["port_loop_value"]
    out["all_cores_speedup_bvh"] = out["port_bvh_all_cores_value"] / out["port_bvh_value"]
    # the GPU's exhaustive kernel (the reference algorithm itself on the GPU), bounded frame
    try:
        gp = pyrt.make_params(512, 512, 8, mode=mode, seed=1, accel=pyrt.ACCEL_BRUTE)
        ctx.render(gp, want_accum=False)
        _, _, sg = ctx.render(gp, want_accum=False)
        out["gpu_loop_value"] = (sg.rays_closest + sg.rays_shadow) / (sg.kernel_ms * 1e-3) / 1e6
        out["gpu_loop_sample"] = "same scene, 512x512, 8 spp, --accel brute (kernel time)"
    except Exception as e:
        out["gpu_loop_value"] = None
        out["gpu_loop_sample"] = "failed: %s" % e
    if os.path.exists(harness):
        try:
            with tempfile.TemporaryDirectory() as tmp:
                r = subprocess.run([harness, "time", pyrt.MESH_DIR, scene_kind, str(w), str(h), str(mode), str(n), "0", "0"],
                                   cwd=tmp, capture_output=True, text=True, check=True, timeout=300)
            secs = json.loads(r.stdout.strip().splitlines()[-1])["seconds"]
            out.update({"value": rays / secs / 1e6, "kind": "reference", "seconds": secs})
            if full and scene_kind in ("cubes", "lowres"):  # (the exhaustive loop over a bigger scene would take hours)
                # BASELINE.json configs[0] exactly: 256x256, -m 1 -N 8 on this scene (same code, 7x the work)
                pf = pyrt.make_params(256, 256, 8, mode=mode, rng_mode=pyrt.RNG_LEGACY)
                _, _, sf = orc.render(pyrt.Scene(scene_kind, 256, 256), pf, math_mode=orc.MATH_LIBM, accel=orc.ACCEL_OBVH)
                with tempfile.TemporaryDirectory() as tmp:
                    r = subprocess.run([harness, "time", pyrt.MESH_DIR, scene_kind, "256", "256", str(mode), "8", "0", "0"],
                                       cwd=tmp, capture_output=True, text=True, check=True, timeout=900)
                fsecs = json.loads(r.stdout.strip().splitlines()[-1])["seconds"]
                frays = sf.rays_closest + sf.rays_shadow
                out.update({"full_config1_value": frays / fsecs / 1e6, "full_config1_seconds": fsecs, "full_config1_rays": frays,
                            "full_config1_sample": "%s scene, 256x256, -m %d -N 8, legacy RNG seed 1: BASELINE.json configs[0] at "
                                                   "its stated size, reference code, 1 thread" % (scene_kind, mode)})
            return out
        except Exception as e:  # the baseline must never cost the bench line: fall back to the port
            out["sample"] += " [reference harness failed: %s]" % type(e).__name__
    out.update({"value": rays / t_port / 1e6, "kind": "port", "seconds": t_port})
    return out


def build_params(ctx, pyrt, args, w, h, spp, mode, nph, k, rank, world):
    accel = pyrt.ACCEL_BRUTE if args.accel == "brute" else pyrt.ACCEL_BVH
    if nph:
        ctx.build_photon_map(nph, seed=1)  # emission + kd order on the device
    return pyrt.make_params(w, h, spp, mode=mode, seed=1, accel=accel, rank=rank, world=world, tile=32,
                            use_photons=1 if nph else 0, k=k, photons_requested=nph, lanes_per_pixel=args.lpp,
                            wavefront=args.integrator == "wavefront")


def second_build_ms(pyrt, scene, local, args):
    """rt_create's build time once the process is warm (rank 0, after the timed region)."""
    c2 = pyrt.Context(scene, device=local, bvh_leaf_max=args.leaf)
    ms = c2.bvh_info().build_ms
    c2.close()
    return ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (debug only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true", default=True,
                    help="also time the reference's code on BASELINE config 1 at its stated size (256x256, -m 1 -N 8: ~65 s of "
                         "one host core; the default since round 3: VERDICT r02 item 8)")
    ap.add_argument("--no-cpu-baseline-full", dest="cpu_baseline_full", action="store_false",
                    help="only the bounded 96x96 sample (~10 s)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-rank", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--pmc-world", type=int, default=1, help=argparse.SUPPRESS)
    ap.add_argument("--accel", default="bvh", choices=["bvh", "brute"])
    ap.add_argument("--leaf", type=int, default=0, help="BVH leaf size override (debug)")
    ap.add_argument("--lpp", type=int, default=0, help="samples of a pixel per wave override (debug)")
    ap.add_argument("--tune-probes", type=int, default=0,
                    help="measured-cost BVH tuning before the frames (rt_bvh_tune): this many probe frames of the workload's camera "
                         "at 128x128x1 (deterministic: a probe count, not a time limit); 0 = off (the default)")
    ap.add_argument("--integrator", default="fused", choices=["fused", "wavefront"],
                    help="wavefront = the opt-in queue-based integrator (same image; DESIGN.md section 8)")
    args = ap.parse_args()

    if args.pmc_child:
        return pmc_child(args)

    # N > 1 without a launcher: become the launcher (a CHILD process, before any GPU call)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29531"),
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    rank_env = int(os.environ.get("RANK", "0"))
    # counters of the timed kernel, measured now, by child processes, before this process
    # initialises the GPU.  At N > 1 rank 0 measures ITS shard (the counters describe one rank's
    # launch; the others wait for it in the rendezvous), so the line carries a per-rank roofline.
    pmc, pmc_source = None, "skipped (--no-pmc)" if args.no_pmc else "measured by rank 0 only"
    # (no other rank may touch a GPU while the counter passes run — on a shared-GPU rehearsal their warm-up
    # kernels would land in rank 0's device-wide counters, and under RCCL they would sit in their first collective for as
    # long as the passes take.  The ranks of one launcher share its pid as parent: a marker file keyed by it is the gate.)
    gate = os.path.join(tempfile.gettempdir(), "rt_bench_pmc_%d_%s.done" % (os.getppid(), os.environ.get("MASTER_PORT", "0")))
    if args.integrator != "fused":
        pmc_source = "skipped: the counters are defined for the fused kernel (the wavefront integrator is several kernels)"
    elif rank_env == 0 and not args.no_pmc:
        if os.path.exists(gate):
            os.remove(gate)
        try:
            pmc, pmc_source = pmc_measure(args, 0, world_env)
        finally:
            if world_env > 1:
                open(gate, "w").close()
    elif world_env > 1 and not args.no_pmc:
        t_gate = time.time()
        while not os.path.exists(gate) and time.time() - t_gate < 3600:
            time.sleep(0.2)
    if world_env > 1:
        # every rank builds the same host BVH: share the host's cores instead of 16 builder threads each
        os.environ.setdefault("RT_BVH_THREADS", str(max(2, len(os.sched_getaffinity(0)) // world_env)))

    import torch
    import pyrt
    from pyrt import dist as rdist

    # RT_DIST_BACKEND=gloo + RT_SHARE_GPU=1: rehearsal of the N-rank flow on ONE GPU
    backend = os.environ.get("RT_DIST_BACKEND", "nccl")
    rank, world, local = rdist.init_from_env(backend)
    if os.environ.get("RT_SHARE_GPU") == "1":
        local = 0
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    kind, w, h, spp, mode, nph, k = WORKLOADS[args.workload]
    if args.spp:
        spp = args.spp
    scene = pyrt.Scene(kind, w, h)
    ctx = pyrt.Context(scene, device=local, bvh_leaf_max=args.leaf)  # raises if the HIP library / a gfx950 device is missing
    tuned = tune_tree(ctx, pyrt, args, mode, nph)
    params = build_params(ctx, pyrt, args, w, h, spp, mode, nph, k, rank, world)

    accum = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)
    bg = torch.from_numpy(pyrt.background(w, h)).to(dev)
    out = torch.empty((h, w, 3), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    frame = rdist.FrameAssembler(ctx, params, rank, world, dev)  # owned tiles -> rank 0 (one gather)

    asm_events = []  # (before, after) the frame assembly on this rank's stream, timed steps only

    def step(timed=False):
        accum.zero_()
        ctx.render_device(params, accum.data_ptr(), stream)
        if timed and world > 1:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            frame.assemble(accum, stream)
            e1.record()
            asm_events.append((e0, e1))
        else:
            frame.assemble(accum, stream)
        if rank == 0:
            ctx.resolve_device(w, h, spp, accum.data_ptr(), bg.data_ptr(), out.data_ptr(), stream)

    # one counted pass (untimed): rays, BVH node fetches and triangle tests are
    # deterministic per frame, so they are measured once
    params.collect_stats = 1
    accum.zero_()
    st = ctx.render_device(params, accum.data_ptr(), stream, stats=True)
    params.collect_stats = 0
    local_counts = [st.rays_closest, st.rays_shadow, st.nodes_visited, st.tris_tested, st.samples, st.knn_queries,
                    st.kd_visited]
    tot = rdist.sum_over_ranks(local_counts, dev)
    rays_per_frame = tot[0] + tot[1]

    for _ in range(args.warmup):
        step()
    rdist.barrier()
    torch.cuda.synchronize()
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(timed=True)
    torch.cuda.synchronize()
    rdist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = rdist.max_over_ranks(elapsed, dev)
    kernel_ms, launches = ctx.profile_collect()
    avg_ms = kernel_ms / max(launches, 1)
    secs = avg_ms * 1e-3
    # per-rank kernel time (slowest / fastest rank: the load balance of the tile sharding) and the
    # time rank 0's stream spends in the assembly (pack + gather + scatter; it includes waiting for
    # the slowest rank's granules, i.e. skew + exchange)
    kernel_ms_max = rdist.max_over_ranks(avg_ms, dev)
    kernel_ms_min = -rdist.max_over_ranks(-avg_ms, dev)
    assemble_ms = (sum(a.elapsed_time(b) for a, b in asm_events) / len(asm_events)) if asm_events else 0.0
    exchange = ("none (one rank)" if world == 1 else
                "%s point-to-point sends of each rank's owned 8x8-pixel granules to rank 0 (%.2f MiB per rank and frame)" % (
                    backend, max(frame.counts) * 64 * 16 / 2**20))

    # ALGORITHMIC bytes per launch (SURVEY §8d per-unit figures x the units of this launch): 32 B
    # per node record fetched (the packed node this build traverses; §8d priced a 64-B float
    # node), 48 B per triangle record tested, 16 B per pixel-sample (accumulator), 32 B per kd node
    alg_bytes = 32 * local_counts[2] + 48 * local_counts[3] + 16 * local_counts[4] + 32 * local_counts[6]
    scene_mb = (32 * ctx.bvh_info().n_nodes + 48 * scene.desc.n_triangles) / 1e6

    # counters: this run's, or (fallback) a committed profile of the SAME device code
    khash = kernel_source_hash()
    if pmc is None:
        ppath = os.path.join(ROOT, "profiles", "r04_pmc_%s.json" % args.workload)
        if os.path.exists(ppath) and not args.spp and world == 1 and args.accel == "bvh":
            saved = json.load(open(ppath))
            if saved.get("kernel_source_hash") == khash:
                pmc, pmc_source = saved["counters"], "profiles/r04_pmc_%s.json @ kernel hash %s [%s]" % (args.workload, khash, pmc_source)
            else:
                pmc_source = "profiles/r04_pmc_%s.json is for kernel hash %s, this build is %s: not used [%s]" % (
                    args.workload, saved.get("kernel_source_hash"), khash, pmc_source)
    hbm_bytes = valu = lane_util = None
    if pmc:
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            # rocprofv3 reports KiB; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B: guide, HBM section)
            hbm_bytes = 2 * pmc["FETCH_SIZE"] * 1024 + pmc["WRITE_SIZE"] * 1024
        valu = pmc.get("SQ_INSTS_VALU")
        if valu and pmc.get("SQ_THREAD_CYCLES_VALU"):
            # active lanes per VALU wave-instruction / 64
            lane_util = min(pmc["SQ_THREAD_CYCLES_VALU"] / (valu * 64.0), 1.0)

    if rank != 0:
        rdist.shutdown()  # (before rank 0 starts its CPU baseline: nobody waits on anybody after this)
    elif world > 1 and os.path.exists(gate):
        os.remove(gate)  # (every rank has passed it long ago: they all took part in the timed steps)
    if rank == 0:
        hbm_roof = {"bound": "hbm", "achieved": (hbm_bytes / secs / 1e9) if hbm_bytes and secs > 0 else None,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s"}
        hbm_roof["frac"] = hbm_roof["achieved"] / HBM_PEAK_GBS if hbm_roof["achieved"] is not None else None
        valu_roof = {"bound": "valu_issue", "achieved": (valu / secs / 1e9) if valu and secs > 0 else None,
                     "peak": VALU_PEAK_GINSTR, "unit": "G wave-instr/s"}
        valu_roof["frac"] = valu_roof["achieved"] / VALU_PEAK_GINSTR if valu_roof["achieved"] is not None else None
        # vector-L1 request rate: TCP cache accesses per CU-clock of the timed kernel (clock from GRBM_GUI_ACTIVE, which
        # rocprofv3 sums over the 8 XCDs)
        tcp = None
        if pmc and pmc.get("TCP_TOTAL_CACHE_ACCESSES_sum") and pmc.get("GRBM_GUI_ACTIVE"):
            tcp = pmc["TCP_TOTAL_CACHE_ACCESSES_sum"] / (pmc["GRBM_GUI_ACTIVE"] / 8.0 * NUM_CUS)
        tcp_roof = {"bound": "vector_l1_divergent_requests", "achieved": tcp, "peak": TCP_DIVERGENT_PEAK,
                    "peak_when_missing_l1": TCP_DIVERGENT_PEAK_L1_MISS, "unit": "cache accesses per CU-clock",
                    "frac": (tcp / TCP_DIVERGENT_PEAK) if tcp is not None else None,
                    "peak_source": "tools/microbench/gather.hip on MI355X: profiles/r03_gather_microbench.json"}
        primary = hbm_roof if args.workload in HBM_BOUND else valu_roof
        roof = dict(primary)
        roof.update({
            "traffic": hbm_bytes,  # measured HBM-side bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE)
            "traffic_source": pmc_source, "kernel_source_hash": khash,
            "scope": "one launch of rank 0's shard (1/%d of the pixels)" % world if world > 1 else "one launch (whole frame)",
            "kernel": "k_render_persist" if args.accel == "bvh" and not nph else "k_render",
            "kernel_ms_avg": avg_ms, "launches": launches,
            "valu_issue": valu_roof, "hbm": hbm_roof, "vector_l1": tcp_roof,
            "valu_lane_utilisation": lane_util,
            # what the caches served: SURVEY §8d's per-unit bytes x units; NOT an HBM fraction (the scene is
            # %.2f MB) — priced against HBM peak it may exceed 1, which only says the caches work
            "algorithmic_bytes_per_launch": alg_bytes,
            "algorithmic_gbs": alg_bytes / secs / 1e9 if secs > 0 else None,
            "scene_mb": scene_mb,
            "nodes_per_ray": tot[2] / max(rays_per_frame, 1), "tris_per_ray": tot[3] / max(rays_per_frame, 1),
            # lanes doing a node step / a leaf test per wave-level step of the traversal loop (rank 0's
            # share; diagnostics of the counted pass)
            "lanes_per_node_step": st.nodes_visited / max(st.reserved[0], 1),
            "leaf_phases_per_node_step": st.reserved[1] / max(st.reserved[0], 1),
            "lanes_at_leaf_per_node_step": st.reserved[2] / max(st.reserved[0], 1),
            "lanes_without_ray_per_node_step": st.reserved[3] / max(st.reserved[0], 1),
            "note": ("HBM-side reading: measured FETCH/WRITE bytes; a scene below 256 MB is partly served by the Infinity "
                     "Cache, whose hits these fabric-side counters still count.  roofline.vector_l1 (the CU's vector L1 against its "
                     "measured ceiling for divergent 16-B requests) is the unit nearest its limit, but round 4's one-request node "
                     "records showed it does not bind alone: 29 % fewer L1 accesses, the same wait cycles, an 18 % longer frame "
                     "(+64 % VALU).  What bounds a wave's traversal step is the memory latency of its slowest lane plus its own "
                     "instruction issue, at the 16 waves per CU that 128 VGPRs and the LDS stacks allow" if args.workload in HBM_BOUND else
                     "cache-resident scene: the binding unit is VALU issue under divergence, not any bandwidth; "
                     "frac = SQ_INSTS_VALU / s over 1024 SIMDs x 2.4 GHz / 2"),
        })
        if pmc:
            roof["counters_per_launch"] = pmc
        res = {
            "metric": metric_text(args.workload),
            "value": rays_per_frame * args.steps / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %s scene (%d triangles), %dx%d, -m %d -N %d, pixel RNG seed 1, %s"
                                   % (args.workload, kind, scene.desc.n_triangles, w, h, mode, spp,
                                      args.accel + (", wavefront integrator" if args.integrator == "wavefront" else "")),
                       "parallelism": "tiles32x%d, owned tiles sent to rank 0" % world,
                       "exchange": exchange, "assemble_ms": assemble_ms,
                       "kernel_ms_max_over_ranks": kernel_ms_max, "kernel_ms_min_over_ranks": kernel_ms_min,
                       "rays_per_frame": rays_per_frame, "samples_per_frame": tot[4],
                       "knn_queries_per_frame": tot[5],
                       # (outside the timed region: the scene's tree, who built it and how long rt_create's build took —
                       # the process's first rt_create, which also loads the code objects, and a second one)
                       "bvh": {"builder": {1: "device", 2: "hybrid", 3: "host"}.get(ctx.bvh_info().builder, "host"),
                               "build_ms_first_create": ctx.bvh_info().build_ms, "build_ms": second_build_ms(pyrt, scene, local, args),
                               "nodes": ctx.bvh_info().n_nodes, "max_depth": ctx.bvh_info().max_depth}},
            "roofline": roof,
        }
        if tuned:
            res["config"]["bvh_tune"] = tuned
        if not args.no_cpu_baseline:
            # (rank 0 only, after the timed region; at N > 1 the other ranks have left the job by now)
            rdist.shutdown()
            cb = cpu_baseline(ctx, kind, mode, dev, full=args.cpu_baseline_full)
            gpu_bvh = res["value"]
            cb["decomposition"] = {
                "gpu_bvh_over_reference_1thread": gpu_bvh / cb["value"],
                "hardware_loop_over_loop": (cb["gpu_loop_value"] / cb["value"]) if cb.get("gpu_loop_value") else None,
                "hardware_bvh_over_bvh_1thread": gpu_bvh / cb["port_bvh_value"],
                "hardware_bvh_over_bvh_all_cores": gpu_bvh / cb["port_bvh_all_cores_value"],
                "algorithm_bvh_over_loop_on_cpu": cb["port_bvh_value"] / cb["port_loop_value"],
            }
            res["cpu_baseline"] = cb
        print(json.dumps(res), flush=True)
    ctx.close()
    rdist.shutdown()


if __name__ == "__main__":
    main()
