import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT,"ray-tracing-engine_amd"))
import numpy as np, pyrt
for kind,w,spp,mode in (("lowres",512,16,1),("cubes",512,16,1),("stress",512,4,1)):
    s=pyrt.Scene(kind,w,w); ctx=pyrt.Context(s)
    p=pyrt.make_params(w,w,spp,mode=mode,seed=1,collect_stats=1)
    _,_,st=ctx.render(p,want_accum=False)
    r=st.reserved
    rays=st.rays_closest+st.rays_shadow
    print(kind,"node steps/ray %.2f | node-loop lane util %.3f | rounds(wave)/ray*64 %.2f | leaf-phase lane util %.3f"%(
      st.nodes_visited/rays, st.nodes_visited/(64*r[0]), r[1]*64/rays, r[2]/(64*r[1])))
    ctx.close()
