import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT,"ray-tracing-engine_amd"))
import numpy as np, pyrt
for kind,w,spp,mode in (("lowres",512,16,1),("cubes",512,16,1),("stress",512,4,1)):
    s=pyrt.Scene(kind,w,w); ctx=pyrt.Context(s)
    p=pyrt.make_params(w,w,spp,mode=mode,seed=1,collect_stats=1)
    _,_,st=ctx.render(p,want_accum=False)
    r=st.reserved; tot=r[3]
    print(kind,"of wave lifetime: node loops %.3f | leaf phases %.3f | pool loop total %.3f (overhead %.3f) | everything else %.3f"%(
      r[0]/tot, r[1]/tot, r[2]/tot, (r[2]-r[0]-r[1])/tot if False else 0, 1-(r[0]+r[1])/tot))
    ctx.close()
