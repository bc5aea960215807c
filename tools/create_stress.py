import sys
sys.path.insert(0, "ray-tracing-engine_amd")
import pyrt
s = pyrt.Scene("stress", 64, 64)
for i in range(3):
    c = pyrt.Context(s)
    print(c.bvh_info().build_ms)
    c.close()
