"""The trees of the three builders compared as TREES (tests/treedigest.py: a digest independent of node numbering and child
slots), and where two of them part.  usage: [RT_DIGEST_DIFF=1] python tools/tree_digest.py [scene ...]   (GPU box)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-engine_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyrt
from treedigest import context_digest, first_differences

if __name__ == "__main__":
    for kind in sys.argv[1:] or ["lowres", "hires", "stress"]:
        s = pyrt.Scene(kind, 64, 64)
        out, exp = {}, {}
        for name, b in (("host", pyrt.BVH_HOST), ("device", pyrt.BVH_DEVICE), ("hybrid", pyrt.BVH_HYBRID)):
            ctx = pyrt.Context(s, bvh_builder=b)
            out[name] = (context_digest(ctx), ctx.bvh_info().n_nodes)
            nodes, tris = ctx.bvh_export()
            exp[name] = (np.ascontiguousarray(nodes).view(np.uint32).reshape(len(nodes), 16).copy(),
                         np.ascontiguousarray(tris).view(np.uint32).reshape(len(tris), 12).copy())
            ctx.close()
        print(kind, " ".join("%s %s (%d nodes)" % (k, v[0], v[1]) for k, v in out.items()),
              "| device == host: %s, hybrid == host: %s" % (out["device"][0] == out["host"][0], out["hybrid"][0] == out["host"][0]), flush=True)
        if os.environ.get("RT_DIGEST_DIFF") and out["device"][0] != out["host"][0]:
            first_differences(exp["host"], exp["device"])

