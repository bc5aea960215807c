import sys, os, ctypes as C
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT,"ray-tracing-engine_amd"))
import numpy as np, pyrt, torch
for kind,w,spp,mode in (("lowres",512,16,1),("stress",512,4,1)):
    s=pyrt.Scene(kind,w,w); ctx=pyrt.Context(s)
    p=pyrt.make_params(w,w,spp,mode=mode,seed=1,collect_stats=1)
    acc=torch.zeros((w,w,4),device="cuda")
    st=pyrt.Stats()
    pyrt._check(pyrt.amd().rt_render_device(ctx._h, C.byref(p), C.c_void_p(acc.data_ptr()), None, C.byref(st)))
    # read raw counters through the fields rt_api filled before samples was overwritten: use reserved + hack fields
    r=st.reserved
    print(kind, "reserved", list(r), "samples", st.samples, "knn", st.knn_queries, "kd", st.kd_visited)
    ctx.close()
