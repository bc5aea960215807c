#!/usr/bin/env python3
"""Regenerate the golden fixtures from the REAL reference (build container only).

Runs oracle/_ref/RayTracer (the stock program, reference source/Main.cpp) and
oracle/_ref/ref_harness (our driver over the reference's own headers, see
oracle/ref_harness.cpp) and stores their OUTPUTS as data:

  tests/golden/ppm/*.ppm       small whole-image P3 outputs (legacy RNG, seed 1)
  tests/golden/manifest.json   command line -> md5 for every image (incl. big ones)
  tests/golden/ref_vectors.json  per-function input/output bit patterns

Nothing from /root/reference is copied except the .off meshes, which are data
(tests/golden/meshes).  Usage:  python tests/golden/make_golden.py [--slow]
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")
MESHES = os.path.join(HERE, "meshes")

# (name, scene, w, h, mode, N, p, k, keep_ppm, slow)
CASES = [
    ("cubes_64_m0_N4", "cubes", 64, 64, 0, 4, 0, 0, True, False),
    ("cubes_64_m1_N4", "cubes", 64, 64, 1, 4, 0, 0, True, False),
    ("cubes_64_m0_N2_p5000_k10", "cubes", 64, 64, 0, 2, 5000, 10, True, False),
    ("cubes_64_m1_N2_p5000_k10", "cubes", 64, 64, 1, 2, 5000, 10, True, False),
    ("cubes_96x64_m1_N3", "cubes", 96, 64, 1, 3, 0, 0, True, False),
    ("cubes_40x56_m0_N5_p2000_k5", "cubes", 40, 56, 0, 5, 2000, 5, True, False),
    ("lowres_48_m1_N4", "lowres", 48, 48, 1, 4, 0, 0, True, False),
    ("lowres_32_m0_N2_p3000_k10", "lowres", 32, 32, 0, 2, 3000, 10, True, False),
    ("cubes_256_m1_N8", "cubes", 256, 256, 1, 8, 0, 0, False, False),
    ("cubes_128_m0_N4_p50000_k10", "cubes", 128, 128, 0, 4, 50000, 10, False, False),
    ("lowres_256_m1_N8", "lowres", 256, 256, 1, 8, 0, 0, False, True),
]


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def main():
    slow = "--slow" in sys.argv
    os.makedirs(os.path.join(HERE, "ppm"), exist_ok=True)
    man_path = os.path.join(HERE, "manifest.json")
    manifest = json.load(open(man_path)) if os.path.exists(man_path) else {}
    with tempfile.TemporaryDirectory() as tmp:
        # the stock binary reads ../meshes relative to CWD (Main.cpp:186-187)
        os.makedirs(os.path.join(tmp, "build"))
        os.symlink(MESHES, os.path.join(tmp, "meshes"))
        cwd = os.path.join(tmp, "build")
        for name, scene, w, h, mode, n, p, k, keep, is_slow in CASES:
            if is_slow and not slow:
                continue
            out = os.path.join(cwd, name + ".ppm")
            cmd = [os.path.join(REF, "ref_harness"), "render", MESHES, scene,
                   str(w), str(h), str(mode), str(n), str(p), str(k), out]
            r = subprocess.run(cmd, cwd=cwd, check=True, capture_output=True, text=True)
            info = json.loads(r.stdout.strip().splitlines()[-1])
            entry = dict(scene=scene, w=w, h=h, mode=mode, N=n, p=p, k=k,
                         md5=md5(out), ref_seconds=round(info["seconds"], 3),
                         ppm=("ppm/%s.ppm" % name) if keep else None)
            if scene == "cubes":
                # cross-check the harness' scene script against the stock program
                out2 = os.path.join(cwd, name + "_stock.ppm")
                args = ["-width", str(w), "-height", str(h), "-m", str(mode), "-N", str(n), "-o", out2]
                if p:
                    args += ["-p", str(p), "-k", str(k)]
                subprocess.run([os.path.join(REF, "RayTracer")] + args, cwd=cwd, check=True,
                               capture_output=True)
                assert md5(out2) == entry["md5"], (name, "harness != stock binary")
                entry["stock_binary_agrees"] = True
            if keep:
                os.replace(out, os.path.join(HERE, "ppm", name + ".ppm"))
            manifest[name] = entry
            print(name, entry["md5"], entry["ref_seconds"], flush=True)
        vec = os.path.join(HERE, "ref_vectors.json")
        subprocess.run([os.path.join(REF, "ref_harness"), "vectors", MESHES, vec], cwd=cwd, check=True,
                       capture_output=True)
        print("vectors", os.path.getsize(vec), "bytes")
    json.dump(manifest, open(man_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
