#!/usr/bin/env python3
"""walks that start at the origin's leaf (RT_LANE_ORIGIN) against the walk from the root and against the oracle: the 1 M-triangle
scene under the drt_1m samplers at 480x270 x 4 spp, everything from depth 1 on in the path kernel"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from raytracer_amd import abi, host, synth
import oracle_lib as ol
sd = synth.mesh_scene(1000, 500, reflect=0.3)
res = (480, 270)
kw = dict(spp=4, lens_radius=0.01, focal_distance=1.3, light_radius=0.5, glossy=0.05, seed=3)
p = host.default_params("hw4", **kw)
cam = host.camera_for(sd, res)
out = {}
for origin in ("1", "0"):
    os.environ["RT_LANE_ORIGIN"] = origin
    g = host.Scene(sd)
    for pd in (1, 3):
        abi.check(g.L.rt_scene_set_option(g.h, abi.RT_OPT_PATH_DEPTH, pd))
        rgb, ids, st = g.render(cam, p, want_ids=True)
        rad, _ = g.render_radiance(cam, p)
        out[(origin, pd)] = (rad, (st.rays_primary, st.rays_reflect, st.rays_refract, st.rays_shadow))
        print("origin", origin, "path depth", pd, out[(origin, pd)][1], flush=True)
    g.close()
t0 = time.time()
o = ol.OracleScene(sd)
orgb, oids, orad = o.render_whitted(ol.oracle_camera(sd.eye, sd.view, sd.fov, *res), p, want_radiance=True)
orad = orad.reshape(res[1], res[0], 3)
print("oracle %.1f s" % (time.time() - t0), flush=True)
for k, (rad, cnt) in out.items():
    bad = (np.abs(rad - orad) > 1e-4 * np.maximum(1.0, np.abs(orad))).any(-1)
    print(k, "pixels off the oracle:", int(bad.sum()), "max", float(np.abs(rad - orad).max()))
a, b = out[("1", 1)][0], out[("0", 1)][0]
print("origin on vs off, pixels differing:", int((a != b).any(-1).sum()))
