"""ctypes bindings of the CHECKERS: oracle/librt_oracle.so (the CPU restatement)
and, when present, oracle/_ref/*.so (the reference itself, compiled in the build
container).  Test infrastructure only -- nothing under raytracer_amd/ imports this.
"""
import ctypes as C
import os

import numpy as np

from raytracer_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "librt_oracle.so")
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
GOLDEN = os.path.join(ROOT, "tests", "golden")

vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int32, C.c_float


class orc_counters(C.Structure):
    _fields_ = [("rays_primary", i64), ("rays_shadow", i64), ("rays_reflect", i64), ("rays_refract", i64),
                ("box_tests", i64), ("tri_tests", i64), ("leaf_refs_read", i64), ("knn_lookups", i64),
                ("knn_node_visits", i64), ("knn_found", i64), ("average_check", C.c_double),
                ("node_pops", i64), ("rays_traced", i64), ("rays_gather", i64)]


class orc_kd_flat(C.Structure):
    _fields_ = [("nnode", i64), ("lo", vp), ("hi", vp), ("left", vp), ("right", vp), ("is_leaf", vp),
                ("leaf_first", vp), ("leaf_count", vp), ("nleafref", i64), ("leaf_prims", vp),
                ("nroot", i32), ("roots", vp)]


_O = None


def oracle():
    global _O
    if _O is None:
        if not os.path.exists(ORACLE_SO):
            raise RuntimeError(f"{ORACLE_SO} missing: run `make -C oracle`")
        L = C.CDLL(ORACLE_SO)
        L.orc_scene_create.restype = vp
        L.orc_scene_create.argtypes = [C.POINTER(abi.rt_scene_desc)]
        L.orc_scene_destroy.argtypes = [vp]
        L.orc_scene_info.argtypes = [vp, C.POINTER(abi.rt_scene_info)]
        L.orc_scene_reset_normals.argtypes = [vp]
        L.orc_counters_get.argtypes = [vp, C.POINTER(orc_counters)]
        L.orc_counters_reset.argtypes = [vp]
        L.orc_kd_flatten.argtypes = [vp, C.POINTER(orc_kd_flat)]
        L.orc_kd_flat_free.argtypes = [C.POINTER(orc_kd_flat)]
        L.orc_box_test.argtypes = [vp, vp, vp, vp]
        L.orc_tri_test.argtypes = [vp, vp, vp, vp]
        L.orc_sphere_test.argtypes = [vp, f32, vp, vp, vp]
        L.orc_trace_closest.argtypes = [vp, vp, i64, vp]
        L.orc_trace_any.argtypes = [vp, vp, vp, i64, vp]
        L.orc_shadow_ray.argtypes = [vp, vp, vp, f32, vp, vp]
        L.orc_camera_setup.argtypes = [vp, vp, f32, i32, i32, C.POINTER(abi.rt_camera)]
        L.orc_primary_ray.argtypes = [C.POINTER(abi.rt_camera), f32, f32, vp]
        L.orc_render_whitted.argtypes = [vp, C.POINTER(abi.rt_camera), C.POINTER(abi.rt_params), vp, vp, vp]
        L.orc_direction_tables.argtypes = [vp, vp, vp, vp]
        L.orc_photon_quantize_dir.argtypes = [vp, vp, vp, vp]
        L.orc_pmap_balance.argtypes = [vp, i64]
        L.orc_pmap_knn.argtypes = [vp, i64, vp, i64, i32, f32, vp, vp, vp, vp, vp]
        L.orc_pmap_irradiance.argtypes = [vp, i64, vp, vp, i64, i32, f32, vp]
        L.orc_rng.restype = f32
        L.orc_rng.argtypes = [C.c_uint64] * 4
        L.orc_light_samples.argtypes = [C.POINTER(abi.rt_light), vp, i32]
        L.orc_light_quad.argtypes = [C.POINTER(abi.rt_light), vp]
        L.orc_render_photon.argtypes = [vp, C.POINTER(abi.rt_camera), C.POINTER(abi.rt_params), vp, i64, vp, i64, vp, vp, vp]
        L.orc_emit_photons.restype = i64
        L.orc_emit_photons.argtypes = [vp, C.POINTER(abi.rt_emit_params), i64, i64, vp, i64]
        L.orc_minstd_next.restype = f32
        L.orc_minstd_next.argtypes = [C.POINTER(C.c_uint32)]
        L.orc_minstd_seed.restype = C.c_uint32
        L.orc_minstd_seed.argtypes = [C.c_uint32]
        L.orc_photon_trace_serial.restype = i64
        L.orc_photon_trace_serial.argtypes = [vp, C.POINTER(abi.rt_emit_params), vp, vp, i64, C.POINTER(C.c_uint32), vp, i64]
        L.orc_set_emission_cosf.argtypes = [i32]
        L.orc_emission_rays_serial.argtypes = [vp, i32, i64, C.POINTER(C.c_uint32), vp, vp]
        L.orc_trace_irradiance.argtypes = [vp, C.POINTER(abi.rt_params), vp, i64, vp, vp, i32, f32, i64, vp]
        L.orc_eye_trace.argtypes = [vp, C.POINTER(abi.rt_camera), C.POINTER(abi.rt_params), vp, i64, vp, i64, vp, vp, i32,
                                    f32, i64, C.POINTER(C.c_uint32), vp, vp]
        _O = L
    return _O


def flat_from_arrays(lo, hi, left, right, is_leaf, leaf_first, leaf_count, leaf_prims, roots):
    return dict(lo=lo, hi=hi, left=left, right=right, is_leaf=is_leaf, leaf_first=leaf_first,
                leaf_count=leaf_count, leaf_prims=leaf_prims, roots=roots)


def canonical_from_tree(t):
    """Tree (left/right naming) -> the traversal-order arrays the product uploads:
    pre-order with the right child first; returns (lohi[n][6], is_leaf[n], depth[n], skip[n], leaf prim lists)."""
    lohi, leaf, depth, skip, lists = [], [], [], [], []

    def rec(i, d):
        me = len(lohi)
        lohi.append(np.concatenate([t["lo"][i], t["hi"][i]])); leaf.append(int(t["is_leaf"][i])); depth.append(d)
        skip.append(0); lists.append(None)
        if t["is_leaf"][i]:
            f, c = int(t["leaf_first"][i]), int(t["leaf_count"][i])
            lists[me] = t["leaf_prims"][f:f + c].copy()
        else:
            rec(int(t["right"][i]), d + 1)
            rec(int(t["left"][i]), d + 1)
        skip[me] = len(lohi)

    import sys
    sys.setrecursionlimit(10000)
    for r in t["roots"]:
        rec(int(r), 0)
    return (np.array(lohi, np.float32).reshape(-1, 6), np.array(leaf, np.int32), np.array(depth, np.int32),
            np.array(skip, np.int32), lists)


class OracleScene:
    def __init__(self, sd):
        self.L = oracle()
        self.sd = sd
        d, self._keep = sd.make_desc()
        self.h = vp(self.L.orc_scene_create(C.byref(d)))

    def close(self):
        if self.h:
            self.L.orc_scene_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        i = abi.rt_scene_info(); self.L.orc_scene_info(self.h, C.byref(i)); return i

    def counters(self):
        c = orc_counters(); self.L.orc_counters_get(self.h, C.byref(c)); return c

    def reset(self):
        self.L.orc_counters_reset(self.h); self.L.orc_scene_reset_normals(self.h)

    def kd_tree(self):
        f = orc_kd_flat()
        self.L.orc_kd_flatten(self.h, C.byref(f))
        n, nl, nr = f.nnode, f.nleafref, f.nroot

        def arr(p, ct, shape):
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=shape).copy()
        out = flat_from_arrays(arr(f.lo, f32, (n, 3)), arr(f.hi, f32, (n, 3)), arr(f.left, i32, (n,)),
                               arr(f.right, i32, (n,)), arr(f.is_leaf, i32, (n,)), arr(f.leaf_first, i64, (n,)),
                               arr(f.leaf_count, i32, (n,)), arr(f.leaf_prims, i32, (max(nl, 1),))[:nl],
                               arr(f.roots, i32, (max(nr, 1),))[:nr])
        self.L.orc_kd_flat_free(C.byref(f))
        return out

    def trace_closest(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = np.zeros(rays.shape[0], abi.hit_dtype)
        self.L.orc_trace_closest(self.h, rays.ctypes.data, rays.shape[0], out.ctypes.data)
        return out

    def trace_any(self, rays, tmax=None):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = np.zeros(rays.shape[0], np.uint8)
        t = None if tmax is None else np.ascontiguousarray(tmax, np.float32)
        self.L.orc_trace_any(self.h, rays.ctypes.data, None if t is None else t.ctypes.data, rays.shape[0],
                             out.ctypes.data)
        return out

    def render_whitted(self, cam, params, want_radiance=False):
        n = cam.resx * cam.resy
        rgb = np.zeros(n * 3, np.uint8); ids = np.zeros(n, np.int32)
        rad = np.zeros((n, 3), np.float32) if want_radiance else None
        self.L.orc_render_whitted(self.h, C.byref(cam), C.byref(params), rgb.ctypes.data, ids.ctypes.data,
                                  None if rad is None else rad.ctypes.data)
        return rgb.reshape(cam.resy, cam.resx, 3), ids, rad

    def render_photon(self, cam, params, gheap, cheap):
        """PhotonMappingImageSynthesis' pixel loop (final.cpp:1146-1202) over the given balanced maps"""
        n = cam.resx * cam.resy
        rgb = np.zeros(n * 3, np.uint8); ids = np.zeros(n, np.int32); rad = np.zeros((n, 3), np.float32)
        g = np.ascontiguousarray(gheap); c = np.ascontiguousarray(cheap)
        self.L.orc_render_photon(self.h, C.byref(cam), C.byref(params), g.ctypes.data, g.shape[0] - 1, c.ctypes.data,
                                 c.shape[0] - 1, rgb.ctypes.data, ids.ctypes.data, rad.ctypes.data)
        return rgb.reshape(cam.resy, cam.resx, 3), ids, rad.reshape(cam.resy, cam.resx, 3)


def oracle_shadow_rays(points, light, normals, bias=1e-4):
    points = np.ascontiguousarray(points, np.float32); normals = np.ascontiguousarray(normals, np.float32)
    light = np.ascontiguousarray(light, np.float32)
    rays = np.zeros((points.shape[0], 6), np.float32); tl = np.zeros(points.shape[0], np.float32)
    L = oracle()
    for i in range(points.shape[0]):
        t = C.c_float()
        L.orc_shadow_ray(points[i].ctypes.data, light.ctypes.data, normals[i].ctypes.data, bias, rays[i].ctypes.data, C.byref(t))
        tl[i] = t.value
    return rays, tl


def oracle_camera(eye, view_raw, fov, resx, resy):
    cam = abi.rt_camera()
    e = np.ascontiguousarray(eye, np.float32); v = np.ascontiguousarray(view_raw, np.float32)
    oracle().orc_camera_setup(e.ctypes.data, v.ctypes.data, float(fov), resx, resy, C.byref(cam))
    return cam


def oracle_primary_rays(cam, xs, ys):
    xs = np.asarray(xs, np.float32).ravel(); ys = np.asarray(ys, np.float32).ravel()
    out = np.zeros((xs.size, 6), np.float32)
    L = oracle()
    for i in range(xs.size):
        L.orc_primary_ray(C.byref(cam), float(xs[i]), float(ys[i]), out[i].ctypes.data)
    return out


def camera_to_array(cam):
    return np.concatenate([np.array(cam.center[:], np.float32), np.array([list(c) for c in cam.corners], np.float32).ravel(),
                           np.array(cam.xaxis[:], np.float32), np.array(cam.yaxis[:], np.float32)])


# ---------------------------------------------------------------------------
# the reference itself (only where oracle/_ref was built)
def have_ref():
    return os.path.exists(os.path.join(REF_DIR, "libref_hw4.so"))


_R = None
_RP = None


def ref_hw4():
    global _R
    if _R is None:
        L = C.CDLL(os.path.join(REF_DIR, "libref_hw4.so"))
        L.ref_scene_build.argtypes = [i64, vp, vp, vp, vp, i32, vp, vp, i32, vp]
        L.ref_kd_stats.argtypes = [C.POINTER(i64), C.POINTER(i32), C.POINTER(i64), C.POINTER(i64)]
        L.ref_kd_dump.argtypes = [vp] * 9
        L.ref_box_test.argtypes = [vp] * 4
        L.ref_tri_test.argtypes = [vp] * 4
        L.ref_sphere_test.argtypes = [vp, f32, vp, vp, vp]
        L.ref_trace_colour.argtypes = [vp, vp, i32, vp, f32, vp]
        L.ref_trace_ids.argtypes = [vp, i64, vp, vp, vp]
        L.ref_in_shadow.argtypes = [vp, vp, vp]
        L.ref_camera.argtypes = [vp, vp, f32, i32, i32, vp]
        L.ref_primary_rays.argtypes = [vp, i32, i32, vp, i64, vp]
        _R = L
    return _R


def ref_photon():
    global _RP
    if _RP is None:
        L = C.CDLL(os.path.join(REF_DIR, "libref_photon.so"))
        L.refp_create.argtypes = [i32]
        L.refp_tables.argtypes = [vp] * 4
        L.refp_add.argtypes = [vp, vp, vp, i64]
        L.refp_scale.argtypes = [f32]
        L.refp_dump.argtypes = [vp]
        L.refp_load.argtypes = [vp, i32]
        L.refp_locate.argtypes = [vp, i32, f32, vp, vp, vp]
        L.refp_irradiance.argtypes = [vp, vp, i64, f32, i32, vp]
        _RP = L
    return _RP


_RF = None


def ref_final():
    """the reference's photon-mapping renderer (final.cpp + photon.cpp) behind oracle/ref_harness_final.cpp"""
    global _RF
    if _RF is None:
        L = C.CDLL(os.path.join(REF_DIR, "libref_final.so"))
        L.reff_add_sphere.argtypes = [vp, vp]
        L.reff_add_triangle.argtypes = [vp, vp]
        L.reff_add_rectlight.argtypes = [vp, f32, f32, i32, i32]
        L.reff_add_caustic_sphere.argtypes = [vp]
        L.reff_camera_center.argtypes = [vp]
        L.reff_light_quads.argtypes = [vp]
        L.reff_light_samples.argtypes = [i32, vp, i32]
        L.reff_is_in_shadow.argtypes = [vp, vp, vp, i64, vp]
        L.reff_sees_light.argtypes = [vp, vp, i64, vp]
        L.reff_maps_create.argtypes = [i32, i32]
        L.reff_map_stored.argtypes = [i32]
        L.reff_map_dump.argtypes = [i32, vp]
        L.reff_map_load.argtypes = [i32, vp, i32]
        L.reff_map_balance_scale.argtypes = [i32]
        L.reff_random.argtypes = [i32, vp]
        L.reff_photon_trace.argtypes = [i32, vp, vp, vp, f32, i64]
        L.reff_trace_irradiance.argtypes = [vp, vp, i32, f32, i64, vp]
        L.reff_eye_trace.argtypes = [vp, vp, i32, f32, i64, vp, vp]
        L.reff_run_emission_loops.argtypes = [C.POINTER(i64), C.POINTER(i64)]
        L.reff_emission_log.argtypes = [i32, vp, vp]
        _RF = L
    return _RF


def ref_final_build_scene(sd):
    """Create the reference's globals from a flavour-"final" SceneData, as its parser would."""
    L = ref_final()
    L.reff_clear()
    for i in range(sd.nprim):
        mat = np.ascontiguousarray(sd.materials[sd.prim_material[i]], np.float32)
        f = np.ascontiguousarray(sd.prims[i], np.float32)
        if sd.prim_type[i] == abi.RT_PRIM_SPHERE:
            L.reff_add_sphere(f.ctypes.data, mat.ctypes.data)
        else:
            L.reff_add_triangle(f.ctypes.data, mat.ctypes.data)
    for l in sd.lights:
        pos = np.ascontiguousarray(l["pos"], np.float32)
        L.reff_add_rectlight(pos.ctypes.data, float(l["size1"]), float(l["size2"]), int(l["n1"]), int(l["n2"]))
    e = np.ascontiguousarray(sd.eye, np.float32)
    L.reff_camera_center(e.ctypes.data)
    return L


def ref_build_scene(sd, id_colour=False, with_lights=True):
    """Build the reference's globals from a SceneData.  id_colour: every primitive gets
    surfColor.x = id+1, Ka = 1 and nothing else, so SimpleMeshRayTrace's colour is the hit id."""
    L = ref_hw4()
    prims = np.ascontiguousarray(sd.prims, np.float32)
    types = np.ascontiguousarray(sd.prim_type, np.uint8)
    if id_colour:
        n = sd.nprim
        mats = np.zeros((n, 10), np.float32)
        mats[:, 0] = np.arange(1, n + 1, dtype=np.float32); mats[:, 3] = 1.0; mats[:, 6] = 1.0; mats[:, 9] = 1.0
        matidx = np.arange(n, dtype=np.int32)
    else:
        mats = np.ascontiguousarray(sd.materials, np.float32); matidx = np.ascontiguousarray(sd.prim_material, np.int32)
    of = np.ascontiguousarray(sd.obj_first, np.int64); oc = np.ascontiguousarray(sd.obj_count, np.int64)
    lights = np.zeros((max(1, len(sd.lights)), 7), np.float32)
    nl = 0
    if with_lights and not id_colour:
        for l in sd.lights:
            lights[nl, 0:3] = l["pos"]; lights[nl, 3:6] = l.get("color", (1, 1, 1)); lights[nl, 6] = l.get("intensity", 1.0)
            nl += 1
    L.ref_scene_build(sd.nprim, prims.ctypes.data, types.ctypes.data, matidx.ctypes.data, mats.ctypes.data,
                      len(of), of.ctypes.data, oc.ctypes.data, nl, lights.ctypes.data)
    return L


def ref_kd_tree():
    L = ref_hw4()
    nb, md, nn, nl = i64(), i32(), i64(), i64()
    L.ref_kd_stats(C.byref(nb), C.byref(md), C.byref(nn), C.byref(nl))
    n, l = nn.value, nl.value
    lo = np.zeros((n, 3), np.float32); hi = np.zeros((n, 3), np.float32)
    left = np.zeros(n, np.int32); right = np.zeros(n, np.int32); leaf = np.zeros(n, np.int32)
    lf = np.zeros(n, np.int64); lc = np.zeros(n, np.int32); lp = np.zeros(max(l, 1), np.int32)
    roots = np.zeros(64, np.int32)
    L.ref_kd_dump(lo.ctypes.data, hi.ctypes.data, left.ctypes.data, right.ctypes.data, leaf.ctypes.data,
                  lf.ctypes.data, lc.ctypes.data, lp.ctypes.data, roots.ctypes.data)
    nroot = int((left[roots[:64]] >= -1).sum()) if n else 0
    # roots are the nodes no one points to
    pointed = set(left[left >= 0].tolist()) | set(right[right >= 0].tolist())
    rts = np.array([i for i in range(n) if i not in pointed], np.int32)
    return flat_from_arrays(lo, hi, left, right, leaf, lf, lc, lp[:l], rts), nb.value, md.value
