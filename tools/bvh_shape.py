#!/usr/bin/env python3
"""Shape of the host- and device-built BVH of a preset scene (nodes, depth, build time).
usage (GPU box): python3 tools/bvh_shape.py stress8 [stress hires lowres]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ray-tracing-engine_amd"))
import pyrt  # noqa: E402

for kind in sys.argv[1:] or ["stress"]:
    sc = pyrt.Scene(kind, 64, 64)
    for builder, name in ((0, "host"), (1, "device")):
        ctx = pyrt.Context(sc, bvh_builder=builder)
        b = ctx.bvh_info()
        print(kind, name, {f: getattr(b, f) for f, _ in b._fields_ if f != "reserved"})
        ctx.close()
    sc.close()
