#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE ITSELF (oracle/_ref, compiled from
/root/reference by oracle/Makefile).  TEST INFRASTRUCTURE ONLY; runs in the build
container (needs oracle/_ref).  The fixtures are data: inputs and the outputs the
unmodified reference functions produced for them.

    python oracle/gen_golden.py [--big]      (--big also makes the 3840x2160 hit-id digest)
"""
import argparse
import ctypes as C
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol                      # noqa: E402
from raytracer_amd import abi, scene         # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
f32 = np.float32


def run_ref_binary(scene_text, name="scene.txt"):
    """Run oracle/_ref/hw4_ref in a scratch cwd (it writes colorOutput.ppm and executionTime_<name>)."""
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, name), "w") as f:
            f.write(scene_text)
        subprocess.run([os.path.join(ol.REF_DIR, "hw4_ref"), name], cwd=d, check=True, stdout=subprocess.DEVNULL)
        ppm = open(os.path.join(d, "colorOutput.ppm"), "rb").read()
        log = open(os.path.join(d, "executionTime_" + name)).read()
    head, rest = ppm.split(b"\n", 1)
    _, w, h, _ = head.split()
    return np.frombuffer(rest, np.uint8).reshape(int(h), int(w), 3).copy(), log


def edge_rays():
    r = []
    for o in ([0, 0, -1], [0.1, 0.2, -1], [0, -0.5, 0.5], [0.25, -0.5, 0.25], [0, 0.1, 0.9]):
        for d in ([0, 0, 1], [0, 1, 0], [1, 0, 0], [0, -1, 0], [0, 0, -1], [0, 0.6, 0.8], [0.6, 0, 0.8], [0.6, 0.8, 0]):
            r.append(o + d)
    for a in np.linspace(-0.5, 0.5, 41):
        r.append([a, 1.0, a + 0.5, 0, -1, 0]); r.append([a, 1.0, a + 0.5, 1e-3, -1, 0])
    return np.array(r, f32)


def random_rays(n, seed, box=1.5):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-box, box, (n, 3)).astype(f32); o[:, 2] = rng.uniform(-1.5, 2.0, n)
    d = rng.normal(size=(n, 3)).astype(f32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(f32)
    return np.concatenate([o, d.astype(f32)], axis=1)


def ref_primary(sd, resx, resy, xs, ys):
    L = ol.ref_hw4()
    camv = np.zeros(21, f32); e = np.ascontiguousarray(sd.eye); v = np.ascontiguousarray(sd.view)
    L.ref_camera(e.ctypes.data, v.ctypes.data, sd.fov, resx, resy, camv.ctypes.data)
    xy = np.stack([xs.ravel(), ys.ravel()], axis=1).astype(f32)
    rays = np.zeros((xy.shape[0], 6), f32)
    L.ref_primary_rays(camv.ctypes.data, resx, resy, xy.ctypes.data, xy.shape[0], rays.ctypes.data)
    return camv, rays


def ref_trace_ids(rays):
    L = ol.ref_hw4()
    n = rays.shape[0]
    prim = np.zeros(n, np.int32); ln = np.zeros(n, f32); pt = np.zeros((n, 3), f32)
    rays = np.ascontiguousarray(rays, f32)
    L.ref_trace_ids(rays.ctypes.data, n, prim.ctypes.data, ln.ctypes.data, pt.ctypes.data)
    return prim, ln, pt


def gen_monkey(sd, text, big):
    # 1. full frame through the unmodified binary
    rgb, log = run_ref_binary(text, "monkey.txt")
    lines = dict(l.split(":", 1) for l in log.strip().splitlines())
    # the image the reference's authors rendered in 2018 (MinGW/x87): an output file the reference holds
    old = open("/root/reference/HW4/hw4/monkey256.ppm", "rb").read().split(b"\n", 1)[1]
    bundled = np.frombuffer(old, np.uint8).reshape(256, 256, 3).copy()
    np.savez_compressed(os.path.join(G, "monkey256_ref.npz"), rgb=rgb, bundled_rgb=bundled,
                        kd_nodes=int(lines["total kd-tree node count"]),
                        kd_depth=int(lines["max depth of kd-tree subdivision"]),
                        avg_check=float(lines["average ray-mesh Intersection check count"]))
    print("monkey256_ref: kd", lines["total kd-tree node count"], lines["max depth of kd-tree subdivision"],
          lines["average ray-mesh Intersection check count"])
    # 2. primary-ray hit ids / len / point through SimpleMeshRayTrace with ID-colour materials
    ol.ref_build_scene(sd, id_colour=True)
    xs, ys = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    camv, rays = ref_primary(sd, 256, 256, xs, ys)
    prim, ln, pt = ref_trace_ids(rays)
    np.savez_compressed(os.path.join(G, "monkey256_hits.npz"), cam=camv, rays=rays, prim=prim, len=ln, point=pt)
    print("monkey256_hits: hit pixels", int((prim >= 0).sum()), "distinct", len(np.unique(prim)) - 1)
    # 3. arbitrary rays (random + degenerate)
    r2 = np.concatenate([random_rays(6000, 7), edge_rays()])
    prim, ln, pt = ref_trace_ids(r2)
    np.savez_compressed(os.path.join(G, "monkey_rays_hits.npz"), rays=r2, prim=prim, len=ln, point=pt)
    # 4. IsInShadow on primary hit points of a 96x96 frame + the KD tree itself
    L = ol.ref_build_scene(sd)
    tree, nb, md = ol.ref_kd_tree()
    lohi, leaf, depth, skip, lists = ol.canonical_from_tree(tree)
    leafprims = np.concatenate([l for l in lists if l is not None]).astype(np.int32)
    np.savez_compressed(os.path.join(G, "monkey_kd.npz"), lohi=lohi, leaf=leaf, depth=depth, skip=skip,
                        leaf_prims=leafprims, nbuilt=nb, maxdepth=md)
    xs, ys = np.meshgrid(np.arange(0, 256, 2), np.arange(0, 256, 2), indexing="ij")
    _, rays = ref_primary(sd, 256, 256, xs, ys)
    ol.ref_build_scene(sd, id_colour=True)
    prim, ln, pt = ref_trace_ids(rays)
    L = ol.ref_build_scene(sd)
    ok = prim >= 0
    pts = pt[ok]; prs = prim[ok]; dirs = rays[ok, 3:6]
    light = np.array(sd.lights[0]["pos"], f32)
    nrm = sd.prims[prs, 9:12].copy()
    flip = (nrm * dirs).sum(1) > 0
    nrm[flip] *= -1
    sh = np.zeros(pts.shape[0], np.uint8)
    for i in range(pts.shape[0]):
        sh[i] = L.ref_in_shadow(pts[i].ctypes.data, light.ctypes.data, nrm[i].ctypes.data)
    np.savez_compressed(os.path.join(G, "monkey_shadow.npz"), point=pts, normal=nrm, light=light, shadow=sh)
    print("monkey_shadow:", int(sh.sum()), "of", sh.size, "in shadow")
    if big:
        ol.ref_build_scene(sd, id_colour=True)
        xs, ys = np.meshgrid(np.arange(3840), np.arange(2160), indexing="ij")
        camv, rays = ref_primary(sd, 3840, 2160, xs, ys)
        prim, ln, pt = ref_trace_ids(rays)
        ids = prim.reshape(3840, 2160)              # [x][y]
        np.savez_compressed(os.path.join(G, "monkey4k_hits_digest.npz"), cam=camv,
                            sha256_ids_xy=hashlib.sha256(ids.tobytes()).hexdigest(),
                            sha256_len_xy=hashlib.sha256(ln.tobytes()).hexdigest(),
                            sub_ids=ids[::8, ::8].copy(), hit_pixels=int((prim >= 0).sum()))
        print("monkey4k: hit pixels", int((prim >= 0).sum()))


def gen_units():
    L = ol.ref_hw4()
    rng = np.random.default_rng(3)
    n = 4000
    c = rng.uniform(-1, 1, (n, 3)).astype(f32); e = rng.uniform(0, 1, (n, 3)).astype(f32)
    e[::7, 1] = 0.0                                   # flat boxes (the floor quad has extend.y == 0)
    o = rng.uniform(-2, 2, (n, 3)).astype(f32); d = rng.normal(size=(n, 3)).astype(f32)
    d[::5, 0] = 0.0; d[::11, 1] = 0.0; d[::13, 2] = 0.0   # zero components: inf / NaN slabs
    o[::17] = c[::17]                                 # origin inside
    box = np.array([L.ref_box_test(c[i].ctypes.data, e[i].ctypes.data, o[i].ctypes.data, d[i].ctypes.data) for i in range(n)], np.uint8)
    v = rng.uniform(-1, 1, (n, 9)).astype(f32)
    to = rng.uniform(-2, 2, (n, 3)).astype(f32)
    cen = v.reshape(n, 3, 3).mean(1) + rng.normal(scale=0.3, size=(n, 3)).astype(f32)
    td = (cen - to).astype(f32); td /= np.linalg.norm(td, axis=1, keepdims=True).astype(f32)
    td[::19] = (v[::19, 3:6] - v[::19, 0:3])          # in-plane rays: det ~ 0
    tri = np.zeros(n, np.uint8); tp = np.zeros((n, 3), f32)
    for i in range(n):
        tri[i] = L.ref_tri_test(v[i].ctypes.data, to[i].ctypes.data, td[i].ctypes.data, tp[i].ctypes.data)
    sc = rng.uniform(-1, 1, (n, 3)).astype(f32); sr = rng.uniform(0.05, 0.8, n).astype(f32)
    so = rng.uniform(-2, 2, (n, 3)).astype(f32); so[::9] = sc[::9] + f32(0.01)   # inside the sphere
    sd_ = (sc + rng.normal(scale=0.4, size=(n, 3)).astype(f32) - so).astype(f32)
    sd_ /= np.linalg.norm(sd_, axis=1, keepdims=True).astype(f32)
    sph = np.zeros(n, np.uint8); sp = np.zeros((n, 3), f32)
    for i in range(n):
        sph[i] = L.ref_sphere_test(sc[i].ctypes.data, float(sr[i]), so[i].ctypes.data, sd_[i].ctypes.data, sp[i].ctypes.data)
    np.savez_compressed(os.path.join(G, "unit_vectors.npz"), box_c=c, box_e=e, box_o=o, box_d=d, box_hit=box,
                        tri_v=v, tri_o=to, tri_d=td, tri_hit=tri, tri_p=tp,
                        sph_c=sc, sph_r=sr, sph_o=so, sph_d=sd_, sph_hit=sph, sph_p=sp)
    print("unit vectors: box", int(box.sum()), "tri", int(tri.sum()), "sphere", int(sph.sum()), "of", n)


def gen_photon():
    R = ol.ref_photon()
    rng = np.random.default_rng(42)
    N, Q, K = 12000, 384, 100
    pos = rng.uniform(-0.5, 0.5, (N, 3)).astype(f32); pos[:, 2] += 0.5
    pos[:N // 2, 1] = -0.5                               # half of them on the floor plane (ties on y)
    d = rng.normal(size=(N, 3)).astype(f32)
    pw = rng.uniform(0, 1, (N, 3)).astype(f32)
    R.refp_create(N)
    R.refp_add(pos.ctypes.data, d.ctypes.data, pw.ctypes.data, N)
    stored = np.zeros(N + 1, abi.photon_dtype); R.refp_dump(stored.ctypes.data)
    R.refp_balance()
    R.refp_scale(1.0 / N)
    heap = np.zeros(N + 1, abi.photon_dtype); R.refp_dump(heap.ctypes.data)
    heap[0] = np.zeros(1, abi.photon_dtype)[0]; stored[0] = heap[0]
    qp = rng.uniform(-0.5, 0.5, (Q, 3)).astype(f32); qp[:, 2] += 0.5; qp[:Q // 2, 1] = -0.5
    qn = rng.normal(size=(Q, 3)).astype(f32)
    out = dict(pos=pos, dir=d, power=pw, stored=stored.view(np.uint8), heap=heap.view(np.uint8), qpos=qp, qnrm=qn, k=K)
    for tag, maxd in (("r1", 1.0), ("r01", 0.1), ("r002", 0.02)):
        idx = np.zeros((Q, K), np.int32); d2 = np.zeros((Q, K), f32); fnd = np.zeros(Q, np.int32); r2 = np.zeros(Q, f32)
        for i in range(Q):
            r = C.c_float()
            fnd[i] = R.refp_locate(qp[i].ctypes.data, K, maxd, idx[i].ctypes.data, d2[i].ctypes.data, C.byref(r)); r2[i] = r.value
        rgb = np.zeros((Q, 3), f32)
        R.refp_irradiance(qp.ctypes.data, qn.ctypes.data, Q, maxd, K, rgb.ctypes.data)
        out.update({f"idx_{tag}": idx, f"d2_{tag}": d2, f"found_{tag}": fnd, f"r2_{tag}": r2, f"rgb_{tag}": rgb,
                    f"maxd_{tag}": maxd})
    tabs = [np.zeros(256, f32) for _ in range(4)]
    R.refp_tables(*[t.ctypes.data for t in tabs])
    out["tables"] = np.stack(tabs)
    np.savez_compressed(os.path.join(G, "photon_golden.npz"), **out)
    print("photon golden: N", N, "Q", Q, "mean found r=0.1:", out["found_r01"].mean())


def final_emit_params(which, seed=7):
    """the two emission loops of final.cpp:1053-1123 as rt_emit_params"""
    e = abi.rt_emit_params(); e.which = which; e.light = 0
    e.power[:] = [3.0, 3.0, 3.0] if which == 0 else [0.2, 0.2, 0.2]      # final.cpp:1078 / :1115
    e.start_nr = 2.40; e.max_depth = 6; e.ncaustic_sphere = 1
    for k, v in enumerate((0.2, -0.3, 0.45, 0.2)):                         # causticSphere, final.cpp:754-758
        e.caustic_sphere[0][k] = v
    e.caustic_floor_y = -0.3; e.seed = seed
    return e


def gen_final():
    """Photon-mapping renderer: outputs of the reference's own PhotonTrace / PhotonTraceCaustics / AddPhoton /
    balance / TraceIrradiance / IsInShadow / GetSamplePoses / TestVisualizeLight / SimpleMeshRayTrace(pmap, cpmap)
    on the bundled Final scene, through oracle/ref_harness_final.cpp.  Random draws are the reference's own engine
    with the seed pinned by oracle/ref_time_shim.h (final.cpp's and photon.cpp's copies of mRandom both start there)."""
    sd = scene.parse_scene_file(os.path.join(G, "final_input.txt"), flavour="final")
    R = ol.ref_final_build_scene(sd)
    cs = np.array([0.2, -0.3, 0.45, 0.2], f32); R.reff_add_caustic_sphere(cs.ctypes.data)
    rng = np.random.default_rng(3)
    out = dict(seed=R.reff_seed())
    # light sampling + light quads
    smp = np.zeros((400, 3), f32); out["nsamples"] = R.reff_light_samples(0, smp.ctypes.data, 400); out["light_samples"] = smp
    q = np.zeros((2, 9), f32); R.reff_light_quads(q.ctypes.data); out["light_quads"] = q
    # emission: caller-given rays, the reference's russian roulette
    R.reff_maps_create(30000, 6000)
    for which, n in ((0, 30500), (1, 9000)):
        pos = np.zeros((n, 3), f32); pos[:, 0] = 0.3 + rng.uniform(-0.1, 0.1, n); pos[:, 1] = 0.49; pos[:, 2] = 0.7 + rng.uniform(-0.1, 0.1, n)
        if which == 0:
            d = rng.normal(size=(n, 3)); d[:, 1] = -np.abs(d[:, 1])
        else:
            d = cs[:3] + rng.normal(scale=0.1, size=(n, 3)) - pos
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(f32)
        col = np.array([3.0, 3.0, 3.0] if which == 0 else [0.2, 0.2, 0.2], f32)
        R.reff_photon_trace(which, pos.ctypes.data, d.ctypes.data, col.ctypes.data, 2.40, n)
        ns = R.reff_map_stored(which)
        st = np.zeros(ns + 1, abi.photon_dtype); R.reff_map_dump(which, st.ctypes.data); st[0] = np.zeros(1, abi.photon_dtype)[0]
        out[f"emit{which}_pos"] = pos; out[f"emit{which}_dir"] = d; out[f"emit{which}_stored"] = st.view(np.uint8)
        print("final: map", which, "emissions", n, "stored", ns)
    R.reff_map_balance_scale(0); R.reff_map_balance_scale(1)
    heaps = []
    for which in (0, 1):
        ns = R.reff_map_stored(which)
        h = np.zeros(ns + 1, abi.photon_dtype); R.reff_map_dump(which, h.ctypes.data); h[0] = np.zeros(1, abi.photon_dtype)[0]
        R.reff_map_load(which, h.ctypes.data, ns)
        heaps.append(h); out[f"heap{which}"] = h.view(np.uint8)
    # TraceIrradiance
    n = 1500
    ori = rng.uniform(-0.45, 0.45, (n, 3)).astype(f32); ori[:, 2] += 0.5
    d = rng.normal(size=(n, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(f32)
    out["ti_ori"] = ori; out["ti_dir"] = d
    for depth in (1, 2, 3, 4):
        rgb = np.zeros((n, 3), f32); R.reff_trace_irradiance(ori.ctypes.data, d.ctypes.data, depth, 2.4, n, rgb.ctypes.data)
        out[f"ti_rgb_d{depth}"] = rgb
    # IsInShadow
    pts = rng.uniform(-0.45, 0.45, (n, 3)).astype(f32); pts[:, 2] += 0.5
    nr = rng.normal(size=(n, 3)); nr = (nr / np.linalg.norm(nr, axis=1, keepdims=True)).astype(f32)
    lp = np.zeros((n, 3), f32); lp[:, 0] = 0.3 + rng.uniform(-.1, .1, n); lp[:, 1] = 0.49; lp[:, 2] = 0.7 + rng.uniform(-.1, .1, n)
    sh = np.zeros(n, np.uint8); R.reff_is_in_shadow(pts.ctypes.data, lp.ctypes.data, nr.ctypes.data, n, sh.ctypes.data)
    out.update(sh_point=pts, sh_normal=nr, sh_light=lp, sh_result=sh)
    # the eye path: 16x16 primary rays of the native camera, the reference's serial engine from its seed
    cam = ol.oracle_camera(sd.eye, sd.view, sd.fov, 16, 16)
    xs, ys = np.meshgrid(np.arange(16), np.arange(16), indexing="ij")
    rays = ol.oracle_primary_rays(cam, xs.ravel(), ys.ravel())
    o3 = np.ascontiguousarray(rays[:, :3]); d3 = np.ascontiguousarray(rays[:, 3:])
    rgb = np.zeros((256, 3), f32); ha = np.zeros(256, np.uint8)
    R.reff_eye_trace(o3.ctypes.data, d3.ctypes.data, 0, 2.4, 256, rgb.ctypes.data, ha.ctypes.data)
    sl = np.zeros(256, np.uint8); R.reff_sees_light(o3.ctypes.data, d3.ctypes.data, 256, sl.ctypes.data)
    out.update(eye_rays=rays, eye_rgb=rgb, eye_hit_any=ha, eye_sees_light=sl, eye_cam=ol.camera_to_array(cam))
    np.savez_compressed(os.path.join(G, "final_golden.npz"), **out)
    print("final golden: eye pixels hit", int(ha.sum()), "see light", int(sl.sum()), "shadowed", int(sh.sum()))


def gen_emission():
    """The reference's own emission loops (final.cpp:1053-1123), run once by PhotonMappingImageSynthesis with a 0 x 0 camera
    in a FRESH process (final.cpp's copy of the serial engine starts at the pinned seed): position and direction of every
    emission, noted on their way into PhotonTrace / PhotonTraceCaustics (oracle/ref_harness_final.cpp, --wrap observers)."""
    import tempfile
    sd = scene.parse_scene_file(os.path.join(G, "final_input.txt"), flavour="final")
    R = ol.ref_final_build_scene(sd)
    cs = np.array([0.2, -0.3, 0.45, 0.2], f32); R.reff_add_caustic_sphere(cs.ctypes.data)
    ng, nc = C.c_int64(0), C.c_int64(0)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)
        try:
            R.reff_run_emission_loops(C.byref(ng), C.byref(nc))
        finally:
            os.chdir(cwd)
    out = dict(seed=R.reff_seed(), n_global=ng.value, n_caustic=nc.value)
    for which, n in ((0, ng.value), (1, nc.value)):
        pd = np.zeros((n, 6), f32); sb = np.zeros(n, np.int32)
        R.reff_emission_log(which, pd.ctypes.data, sb.ctypes.data)
        out[f"posdir{which}"] = pd; out[f"stored_before{which}"] = sb
    # kept small: every global emission's ray is 24 B; the first 40 000 global + all caustic emissions pin the loop
    keep = min(ng.value, 40000)
    out["posdir0"] = out["posdir0"][:keep]; out["stored_before0"] = out["stored_before0"][:keep]
    np.savez_compressed(os.path.join(G, "emission_golden.npz"), **out)
    print("emission golden: global emissions", ng.value, "(kept", keep, ") caustic emissions", nc.value)


def gen_quantise():
    """AddPhoton's direction bytes (photon.cpp:846,869-882) for 60 000 random directions."""
    R = ol.ref_photon()
    n = 60000
    rng = np.random.default_rng(11)
    d = rng.normal(size=(n, 3)).astype(f32)
    d[:64] = np.eye(3, dtype=f32)[rng.integers(0, 3, 64)] * rng.choice([-1, 1], (64, 1)).astype(f32)   # axis-aligned
    z = np.zeros((n, 3), f32)
    R.refp_create(n); R.refp_add(z.ctypes.data, d.ctypes.data, z.ctypes.data, n)
    st = np.zeros(n + 1, abi.photon_dtype); R.refp_dump(st.ctypes.data)
    np.savez_compressed(os.path.join(G, "photon_quantise.npz"), dir=d, phi=st["phi"][1:], theta=st["theta"][1:])
    print("quantise golden:", n)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--only", default="", help="comma list: monkey,units,photon,final,quantise,emission")
    a = ap.parse_args()
    if not ol.have_ref():
        sys.exit("oracle/_ref is missing: run `make -C oracle ref` in the build container")
    text = open(os.path.join(G, "monkey.txt")).read()
    sd = scene.parse_scene_text(text)
    only = set(a.only.split(",")) if a.only else None
    if only == {"emission"}:
        gen_emission()
        sys.exit(0)
    if not only or "emission" in only:        # needs the reference's engine untouched: a process of its own
        import subprocess
        subprocess.run([sys.executable, os.path.abspath(__file__), "--only", "emission"], check=True)
    if not only or "monkey" in only: gen_monkey(sd, text, a.big)
    if not only or "units" in only: gen_units()
    if not only or "photon" in only: gen_photon()
    if not only or "final" in only: gen_final()
    if not only or "quantise" in only: gen_quantise()
