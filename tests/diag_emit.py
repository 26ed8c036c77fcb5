import sys, os, ctypes as C, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import oracle_lib as ol
from raytracer_amd import abi, host, scene
sd = scene.parse_scene_file(os.path.join(ol.GOLDEN, "final_input.txt"), flavour="final")
g = host.Scene(sd); o = ol.OracleScene(sd)
for which,count in ((0,200000),(1,200000)):
    e = host.final_emit_params(which, seed=99)
    cap=2*count+64; want=np.zeros(cap,abi.photon_dtype); nw=ol.oracle().orc_emit_photons(o.h,C.byref(e),0,count,want.ctypes.data,cap); want=want[:nw]
    got,ng=g.photonmap_emit(e,0,count)
    print("which",which,"counts",ng,nw)
    if ng==nw:
        for f in ("position","power","inDir"):
            d=np.abs(got[f].astype(np.float64)-want[f].astype(np.float64)).max(axis=1)
            print(f,"differ",(d>0).sum(),"max",d.max(),"quantiles of nonzero",np.quantile(d[d>0],[.5,.9,.99,1.0]) if (d>0).any() else None, ">1e-4:",(d>1e-4).sum())
        for f in ("phi","theta"):
            dd=np.abs(got[f].astype(int)-want[f].astype(int)); print(f,(dd>0).sum(),dd.max())
