#!/usr/bin/env python3
"""Fuzz of the Whitted / distributed-ray-tracing frame, not collected by pytest: random triangle soups with spheres, random
materials (mirrors, glass), lights, resolutions and sampler settings; rt_render on two streams (the default) and on one
stream must give the same bytes, and against the oracle (normals at their fixed point): hit ids, primary and shadow ray counts exact, radiance within 1e-4 * max(1, |L|)
on >= 99.5 % of the pixels, the 8-bit frame within 1 LSB on >= 99.5 %.  `python tests/fuzz_render.py [cases] [seed]`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib as ol
from raytracer_amd import abi, host
from raytracer_amd.scene import SceneData

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
only = int(os.environ.get("FUZZ_ONLY", "-1"))


def soup(rng):
    n = int(2 ** rng.uniform(3, 12))                                  # 8 .. 4096 triangles
    ns = int(rng.integers(0, 4))                                      # spheres outside the trees
    nobj = int(rng.integers(1, 5))
    c = rng.uniform(-0.45, 0.45, (n, 1, 3)); v = (c + rng.normal(scale=rng.uniform(0.02, 0.15), size=(n, 3, 3))).astype(np.float32)
    v[..., 2] += 0.6
    v[0] = [[-2, -0.5, -1], [-2, -0.5, 3], [2, -0.5, 3]]; v[1] = [[-2, -0.5, -1], [2, -0.5, 3], [2, -0.5, -1]]
    nrm = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]).astype(np.float32)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-20)
    prims = np.concatenate([v.reshape(n, 9), nrm], axis=1).astype(np.float32)
    ptype = np.zeros(n + ns, np.uint8)
    if ns:
        sp = np.zeros((ns, 12), np.float32)
        sp[:, :3] = rng.uniform(-0.4, 0.4, (ns, 3)); sp[:, 2] += 0.6; sp[:, 3] = rng.uniform(0.05, 0.25, ns)
        prims = np.concatenate([prims, sp]); ptype[n:] = 1
    nm = int(rng.integers(1, 6))
    mats = np.zeros((nm, 10), np.float32)
    mats[:, 0:3] = rng.uniform(0.1, 1.0, (nm, 3)); mats[:, 3:6] = rng.uniform(0.1, 0.9, (nm, 3))   # colour, Ka Kd Ks
    mats[:, 6] = rng.integers(1, 40, nm)                                                          # specularity
    mats[:, 7] = np.where(rng.random(nm) < 0.5, rng.uniform(0.1, 0.8, nm), 0)                     # reflect
    mats[:, 8] = np.where(rng.random(nm) < 0.3, rng.uniform(0.2, 0.9, nm), 0)                     # refract
    mats[:, 9] = rng.uniform(1.0, 1.8, nm)                                                        # Nr
    pmat = rng.integers(0, nm, n + ns).astype(np.int32)
    cuts = np.linspace(0, n, nobj + 1).astype(np.int64)
    lights = [dict(pos=tuple(rng.uniform(-1.5, 1.5, 3) + np.array([0, 1.5, -0.5]))) for _ in range(int(rng.integers(1, 4)))]
    return SceneData(prims=prims, prim_type=ptype, prim_material=pmat, materials=mats, obj_first=cuts[:-1],
                     obj_count=np.diff(cuts), lights=lights, eye=np.array([0, 0, -1], np.float32),
                     view=np.array([0, 0, 1], np.float32), fov=45.0, resx=64, resy=64)


bad = 0
for c in range(cases):
    sd = soup(rng)
    res = (int(rng.integers(17, 120)), int(rng.integers(9, 90)))
    kw = {}
    if rng.random() < 0.6:
        kw = dict(spp=int(rng.integers(1, 9)), seed=int(rng.integers(0, 1 << 30)))
        if rng.random() < 0.5: kw.update(lens_radius=float(rng.uniform(0.001, 0.03)), focal_distance=float(rng.uniform(0.8, 2.0)))
        if rng.random() < 0.5: kw.update(light_radius=float(rng.uniform(0.05, 0.6)))
        if rng.random() < 0.5: kw.update(glossy=float(rng.uniform(0.01, 0.2)))
    both = bool(((sd.materials[:, 7] > 0) & (sd.materials[:, 8] > 0)).any())      # a material that reflects AND refracts: 2^depth rays per sample
    if both:
        kw["max_depth"] = int(rng.integers(3, 8))
        if "spp" in kw: kw["spp"] = min(kw["spp"], 2)
    if only >= 0 and c != only:
        continue
    print(f"case {c}: {sd.prims.shape[0]} prims {res[0]}x{res[1]} ...", flush=True)
    g = host.Scene(sd); o = ol.OracleScene(sd)
    cam = host.camera_for(sd, res)
    p = host.default_params("hw4", **kw)
    abi.check(g.L.rt_scene_set_option(g.h, abi.RT_OPT_RENDER_OVERLAP, 0))
    one, ids1, st1 = g.render(cam, p, want_ids=True)
    abi.check(g.L.rt_scene_set_option(g.h, abi.RT_OPT_RENDER_OVERLAP, 1))
    two, ids2, st2 = g.render(cam, p, want_ids=True)
    rad, _ = g.render_radiance(cam, p)
    # walks from the origin's leaf (the default) against walks from the root (RT_LANE_ORIGIN=0), with every secondary ray in
    # the per-lane kernels (path kernel from depth 1, then passes per lane + path kernel from 2): the same bytes and counts
    os.environ["RT_LANE_ORIGIN"] = "0"
    g0 = host.Scene(sd)
    del os.environ["RT_LANE_ORIGIN"]
    origin_same = True
    for pd, ld in ((1, 1), (2, 1)):
        res_o = []
        for sc in (g, g0):
            abi.check(sc.L.rt_scene_set_option(sc.h, abi.RT_OPT_PATH_DEPTH, pd)); abi.check(sc.L.rt_scene_set_option(sc.h, abi.RT_OPT_LANE_DEPTH, ld))
            fo, io, so = sc.render(cam, p, want_ids=True)
            ro, _ = sc.render_radiance(cam, p)
            res_o.append((fo, io, ro, (so.rays_primary, so.rays_reflect, so.rays_refract, so.rays_shadow)))
        eqr = (res_o[0][2] == res_o[1][2]) | (np.isnan(res_o[0][2]) & np.isnan(res_o[1][2]))
        origin_same = origin_same and bool(np.array_equal(res_o[0][0], res_o[1][0]) and np.array_equal(res_o[0][1], res_o[1][1]) and eqr.all() and res_o[0][3] == res_o[1][3])
    g0.close()
    abi.check(g.L.rt_scene_set_option(g.h, abi.RT_OPT_PATH_DEPTH, 3)); abi.check(g.L.rt_scene_set_option(g.h, abi.RT_OPT_LANE_DEPTH, 2))
    # Triangle::GetNormal re-normalises the stored normal in place on every call (hw4.cpp:135): the reference's frame depends
    # on how often each triangle has been hit before.  The device uses the fixed point, so the oracle renders the
    # frame twice more: where the re-normalisation does not settle (it can alternate between two neighbours), the reference's
    # own consecutive frames differ, and in a soup of mirrors and glass ten bounces deep an ulp decides which sliver is hit.
    # That self-variation (pixels beyond the tolerance, hits) is the yardstick for the device's frame.
    ocam = ol.oracle_camera(sd.eye, sd.view, sd.fov, *res)
    o.render_whitted(ocam, p)
    o.L.orc_counters_reset(o.h)
    _, _, oprev = o.render_whitted(ocam, p, want_radiance=True)
    ocp = o.counters()
    o.L.orc_counters_reset(o.h)
    orgb, oids, orad = o.render_whitted(ocam, p, want_radiance=True)
    oc = o.counters()
    same = bool(np.array_equal(one, two) and np.array_equal(ids1, ids2))
    ids_ok = bool(np.array_equal(ids2, oids))
    # primary rays and hits (one shadow ray per hit and light) of every depth; the oracle also counts the children a hit at
    # the depth limit would have had (its recursion returns at once, hw4.cpp:796), the device does not spawn them
    L = max(1, len(sd.lights))
    self_hits = abs(oc.rays_shadow - ocp.rays_shadow) // L
    # (even when two oracle frames agree, a normal that alternates is read at one parity there and at its fixed value here:
    # a few hits in a thousand may differ ten bounces deep)
    slack = max(3 * self_hits + 2, (oc.rays_shadow // L) // 500)
    counts_ok = st2.rays_primary == oc.rays_primary and abs(st2.rays_shadow - oc.rays_shadow) // L <= slack and \
                st2.rays_reflect <= oc.rays_reflect + slack and st2.rays_refract <= oc.rays_refract + slack
    orad = orad.reshape(res[1], res[0], 3)
    tol = 1e-4 * np.maximum(1.0, np.abs(orad))
    off = (~(np.abs(rad - orad) <= tol) & ~(np.isnan(rad) & np.isnan(orad))).any(-1)
    d = np.abs(two.astype(int) - orgb.reshape(two.shape).astype(int)).max(-1)
    npx = res[0] * res[1]
    oprev = oprev.reshape(res[1], res[0], 3)
    self_off = int((~(np.abs(oprev - orad) <= tol) & ~(np.isnan(oprev) & np.isnan(orad))).any(-1).sum())
    rad_ok = bool(off.sum() <= max(1, 0.005 * npx, 3 * self_off) and (d > 1).sum() <= max(1, 0.005 * npx, 3 * self_off))
    ok = same and ids_ok and counts_ok and rad_ok and origin_same
    if only >= 0:
        prev = orad
        for rep in range(3):
            _, _, r2 = o.render_whitted(ocam, p, want_radiance=True)
            r2 = r2.reshape(res[1], res[0], 3)
            print("   oracle frame", rep + 3, "differs from the one before in", int((r2 != prev).any(-1).sum()), "px; from the GPU's in",
                  int((~(np.abs(rad - r2) <= tol)).any(-1).sum()), "px beyond tolerance,", int((rad != r2).any(-1).sum()), "px in any bit")
            prev = r2
        for mode in ("strict", "fast", "lane", "bvh"):
            g.set_traversal(mode)
            r3, i3, s3 = g.render(cam, p, want_ids=True)
            rd3, _ = g.render_radiance(cam, p)
            off3 = (~(np.abs(rd3 - orad) <= tol) & ~(np.isnan(rd3) & np.isnan(orad))).any(-1)
            print("   mode", mode, "rays", (s3.rays_primary, s3.rays_reflect, s3.rays_refract, s3.rays_shadow), "px off", int(off3.sum()),
                  "same bytes as default", bool(np.array_equal(r3, two)))
        ys, xs = np.nonzero(off)
        for y, x in list(zip(ys, xs))[:8]:
            print("   px", x, y, "gpu", rad[y, x], "oracle", orad[y, x], "id", ids2.reshape(res[1], res[0])[y, x])
        for md in (1, 2, 3, 5, 8):
            pm = host.default_params("hw4", **dict(kw, max_depth=md))
            g.set_traversal("bvh")
            _, _, sm = g.render(cam, pm, want_ids=True)
            o.reset()
            o.render_whitted(ol.oracle_camera(sd.eye, sd.view, sd.fov, *res), pm)
            om = o.counters()
            print("   max_depth", md, "gpu", (sm.rays_primary, sm.rays_reflect, sm.rays_refract, sm.rays_shadow),
                  "oracle", (om.rays_primary, om.rays_reflect, om.rays_refract, om.rays_shadow))
    if not counts_ok:
        print("   oracle counts", (oc.rays_primary, oc.rays_reflect, oc.rays_refract, oc.rays_shadow), "lights", len(sd.lights))
    bad += 0 if ok else 1
    print(f"case {c}: {sd.prims.shape[0]} prims {res[0]}x{res[1]} {kw} passes {st2.passes} rays {st2.rays_primary}/{st2.rays_reflect}/"
          f"{st2.rays_refract}/{st2.rays_shadow} streams={'same' if same else 'DIFFER'} ids={'ok' if ids_ok else 'DIFFER'} "
          f"counts={'ok' if counts_ok else 'DIFFER'} origin-walks={'same' if origin_same else 'DIFFER'} radiance={'ok' if rad_ok else 'DIFFER'} ({int(off.sum())} px off, {int((d > 1).sum())} px > 1 LSB; the oracle against itself: {self_off} px, {self_hits} hits)",
          flush=True)
    g.close(); o.close()
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
