import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT,"ray-tracing-engine_amd"))
import pyrt
for k in ("cubes","lowres","hires","stress"):
    s=pyrt.Scene(k,64,64); c=pyrt.Context(s); b=c.bvh_info(); print(k, "nodes", b.n_nodes, "depth", b.max_depth, "pad", b.pad); c.close()
