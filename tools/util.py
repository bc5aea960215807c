import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT,"ray-tracing-engine_amd"))
import numpy as np, pyrt
for kind,w,spp,mode in (("lowres",512,16,1),("lowres",512,16,0),("cubes",512,16,1),("stress",512,4,1)):
    s=pyrt.Scene(kind,w,w); ctx=pyrt.Context(s)
    p=pyrt.make_params(w,w,spp,mode=mode,seed=1,collect_stats=1)
    _,_,st=ctx.render(p,want_accum=False)
    r=st.reserved
    print(kind,"mode",mode,"wave time in closest %.3f shadow %.3f other(shading, rng) %.3f | cycles/closest-cast %.0f /shadow-cast %.0f | ms %.2f"%(
      r[0]/r[2], r[1]/r[2], 1-(r[0]+r[1])/r[2], r[0]/(st.rays_closest/64), r[1]/(st.rays_shadow/64), st.kernel_ms))
    ctx.close()
