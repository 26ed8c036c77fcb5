"""Host-side handles over the C ABI (include/rt_hip.h, include/rt_host.h).

`Scene` owns an `rt_scene*`; its methods have the names and argument meaning of
the reference functions they replace (SimpleMeshRayTrace's closest-hit loop ->
trace_closest, IsInShadow -> trace_any, CalcIrradiance -> photon_irradiance,
GenerateColorImage -> render).  Everything computes on the GPU through
librt_hip.so; nothing here falls back to the CPU.
"""
import ctypes as C
import os

import numpy as np

from . import abi
from .scene import SceneData

_HOSTLIB = None


def host_lib():
    global _HOSTLIB
    if _HOSTLIB is None:
        p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "librt_host.so")
        if not os.path.exists(p):
            raise RuntimeError(f"{p} is missing: build it with `make -C raytracer_amd/host`")
        L = C.CDLL(p)
        L.rth_camera_setup.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_int32, C.POINTER(abi.rt_camera)]
        L.rth_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32]
        L.rth_write_execution_time.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_int64,
                                               C.c_int32, C.c_double]
        L.rth_ssaa_downsample.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        vpp = C.POINTER(C.c_void_p)
        L.rth_scene_parse_file.argtypes = [C.c_char_p, C.c_int, C.c_int, vpp]
        L.rth_scene_parse_text.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int, vpp]
        L.rth_scene_fill_desc.argtypes = [C.c_void_p, C.POINTER(abi.rt_scene_desc)]
        L.rth_scene_view.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int32)]
        L.rth_scene_save.argtypes = [C.c_void_p, C.c_char_p]
        L.rth_scene_load.argtypes = [C.c_char_p, vpp]
        L.rth_scene_free.argtypes = [C.c_void_p]
        L.rth_scene_free.restype = None
        _HOSTLIB = L
    return _HOSTLIB


HOST_SYMBOLS = ("rth_camera_setup", "rth_write_ppm", "rth_write_execution_time", "rth_ssaa_downsample",
                "rth_scene_parse_file", "rth_scene_parse_text", "rth_scene_fill_desc", "rth_scene_view", "rth_scene_save",
                "rth_scene_load", "rth_scene_free")
FLAVOURS = {"hw4": 0, "final": 1}


def _scene_from_handle(h, flavour):
    """copies what an rth_scene holds into a SceneData (the arrays the C++ front end hands to rt_scene_create)"""
    L = host_lib()
    d = abi.rt_scene_desc()
    if L.rth_scene_fill_desc(h, C.byref(d)):
        raise ValueError("rth_scene_fill_desc failed")
    n, nm, no, nl = int(d.nprim), int(d.nmaterial), int(d.nobject), int(d.nlight)

    def arr(ptr, count, dtype, cols=1):
        if count == 0:
            return np.zeros((0, cols) if cols > 1 else (0,), dtype)
        nbytes = count * cols * np.dtype(dtype).itemsize
        a = np.frombuffer(C.string_at(ptr, nbytes), dtype=dtype).copy()
        return a.reshape(count, cols) if cols > 1 else a
    lights = []
    if nl:
        la = C.cast(d.lights, C.POINTER(abi.rt_light * nl)).contents
        for l in la:
            lights.append(dict(type=int(l.type), pos=list(l.pos), color=list(l.color), intensity=float(l.intensity),
                               dir1=list(l.dir1), dir2=list(l.dir2), size1=float(l.size1), size2=float(l.size2),
                               n1=int(l.n1), n2=int(l.n2)))
    eye = np.zeros(3, np.float32); view = np.zeros(3, np.float32)
    fov = C.c_float(); rx = C.c_int32(); ry = C.c_int32()
    L.rth_scene_view(h, eye.ctypes.data, view.ctypes.data, C.byref(fov), C.byref(rx), C.byref(ry))
    return SceneData(prims=arr(d.prims, n, np.float32, 12), prim_type=arr(d.prim_type, n, np.uint8),
                     prim_material=arr(d.prim_material, n, np.int32), materials=arr(d.materials, nm, np.float32, 10),
                     obj_first=arr(d.obj_first, no, np.int64), obj_count=arr(d.obj_count, no, np.int64), lights=lights,
                     eye=eye, view=view, fov=float(fov.value), resx=int(rx.value), resy=int(ry.value), flavour=flavour)


def parse_scene_native(path=None, text=None, flavour="hw4", nthreads=0, cache_save=None, cache_load=None):
    """the C++ front end's parser (include/rt_host.h, rth_scene_*) -> SceneData; optional binary cache round trip"""
    L = host_lib()
    h = C.c_void_p()
    if cache_load is not None:
        rc = L.rth_scene_load(os.fsencode(cache_load), C.byref(h))
    elif path is not None:
        rc = L.rth_scene_parse_file(os.fsencode(path), FLAVOURS[flavour], nthreads, C.byref(h))
    else:
        b = text.encode() if isinstance(text, str) else bytes(text)
        rc = L.rth_scene_parse_text(b, len(b), FLAVOURS[flavour], nthreads, C.byref(h))
    if rc:
        raise ValueError(f"scene parse / load failed ({rc})")
    try:
        if cache_save is not None and L.rth_scene_save(h, os.fsencode(cache_save)):
            raise ValueError("rth_scene_save failed")
        return _scene_from_handle(h, flavour)
    finally:
        L.rth_scene_free(h)


def camera_setup(eye, view_raw, fov, resx, resy):
    """Camera::CalcPlaneParameters (hw4.cpp:242-266) -> rt_camera."""
    cam = abi.rt_camera()
    e = np.ascontiguousarray(eye, np.float32); v = np.ascontiguousarray(view_raw, np.float32)
    rc = host_lib().rth_camera_setup(e.ctypes.data, v.ctypes.data, float(fov), int(resx), int(resy), C.byref(cam))
    if rc:
        raise ValueError("rth_camera_setup: bad argument")
    return cam


def camera_for(sd: SceneData, resolution=None):
    rx, ry = resolution if resolution else (sd.resx, sd.resy)
    return camera_setup(sd.eye, sd.view, sd.fov, rx, ry)


def write_ppm(path, rgb8, resx, resy):
    a = np.ascontiguousarray(rgb8, np.uint8)
    if host_lib().rth_write_ppm(os.fsencode(path), a.ctypes.data, resx, resy):
        raise OSError(f"cannot write {path}")


def default_params(flavour="hw4", **kw):
    p = abi.rt_params()
    p.mode = abi.RT_RENDER_WHITTED if flavour == "hw4" else abi.RT_RENDER_PHOTON
    p.max_depth = 10 if flavour == "hw4" else 5
    p.ambient_intensity = 1.0
    p.background[:] = [0.0, 0.0, 0.0]
    p.start_nr = 1.0 if flavour == "hw4" else 2.4
    p.primary_only = 0
    p.count_work = 0
    p.tile_size = 0; p.tile_rank = 0; p.tile_nranks = 1
    p.gather_samples = 50; p.knn_k = 100; p.knn_radius_global = 1.0; p.knn_radius_caustic = 0.1
    p.irradiance_max_depth = 3; p.aa_taps = 0 if flavour == "hw4" else 6
    p.seed = 1
    p.spp = 0; p.lens_radius = 0.0; p.focal_distance = 0.0; p.light_radius = 0.0; p.glossy = 0.0
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def pixel_list(cam, params):
    """rt_render_pixel_list: the pixels (x + y * resx, -1 = outside the frame) of a render call in the order of
    rt_params.packed_output"""
    L = abi.load()
    n = C.c_int64(0)
    abi.check(L.rt_render_pixel_list(C.byref(cam), C.byref(params), None, 0, C.byref(n)))
    out = np.empty(n.value, np.int32)
    abi.check(L.rt_render_pixel_list(C.byref(cam), C.byref(params), out.ctypes.data, n.value, C.byref(n)))
    return out


def final_emit_params(which, seed=1, light=0):
    """the two emission loops of final.cpp:1053-1123: global photons of colour 3 (:1078), caustic photons of
    colour 0.2 (:1115) aimed at the hard-coded proxy sphere (:754-758), Nr 2.40, depth 6 (parameter.h:26)."""
    e = abi.rt_emit_params(); e.which = which; e.light = light
    e.power[:] = [3.0, 3.0, 3.0] if which == abi.RT_PMAP_GLOBAL else [0.2, 0.2, 0.2]
    e.start_nr = 2.40; e.max_depth = 6; e.ncaustic_sphere = 1
    for k, v in enumerate((0.2, -0.3, 0.45, 0.2)):
        e.caustic_sphere[0][k] = v
    e.caustic_floor_y = -0.3; e.seed = seed
    return e


def host_flatten(sd: SceneData):
    """The KD-trees exactly as rt_scene_create uploads them, built on the host (no GPU)."""
    L = abi.load()
    d, keep = sd.make_desc()
    h = C.c_void_p(); info = abi.rt_scene_info()
    abi.check(L.rt_host_flatten(C.byref(d), C.byref(h), C.byref(info)))
    try:
        def get(which, dtype, cols):
            n = L.rt_host_flat_get(h, which, None, 0)
            abi.check(n)
            a = np.zeros((n, cols) if cols > 1 else (n,), dtype)
            abi.check(L.rt_host_flat_get(h, which, a.ctypes.data, n))
            return a
        out = dict(lohi=get(0, np.float32, 6), skip=get(1, np.int32, 1), meta=get(2, np.int32, 1),
                   leaf_first=get(3, np.int32, 1), leaf_count=get(4, np.int32, 1), leaf_prim=get(5, np.int32, 1),
                   loose=get(6, np.int32, 1), leaf_tri=get(7, np.float32, 9), info=info)
    finally:
        L.rt_host_flat_free(h)
    return out


class Scene:
    """rt_scene handle (GPU).  Raises RtError when no gfx950 device is usable."""

    def __init__(self, sd: SceneData, device=-1):
        self.L = abi.load()
        self.sd = sd
        d, keep = sd.make_desc(device)
        h = C.c_void_p()
        abi.check(self.L.rt_scene_create(C.byref(d), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.rt_scene_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_traversal(self, mode):
        """"strict" (reference order, no culling), "fast" (reference tree, ordered + culled), "lane" (per lane) or "bvh"
        (the build's own hierarchy, the default): include/rt_hip.h."""
        abi.check(self.L.rt_scene_set_option(self.h, abi.RT_OPT_TRAVERSAL,
                                             {"strict": abi.RT_TRAVERSAL_STRICT, "fast": abi.RT_TRAVERSAL_FAST,
                                              "lane": abi.RT_TRAVERSAL_LANE, "bvh": abi.RT_TRAVERSAL_BVH}[mode]))

    def knn_stats(self, on=None):
        """work counters of the wave-shared radiance-estimate kernels (include/rt_hip.h, RT_OPT_KNN_STATS): on=True zeroes and
        starts them, on=False stops; returns the 16 counters"""
        if on is not None:
            abi.check(self.L.rt_scene_set_option(self.h, abi.RT_OPT_KNN_STATS, 1 if on else 0))
        out = np.zeros(16, np.int64)
        abi.check(self.L.rt_scene_knn_stats(self.h, out.ctypes.data))
        return out

    def set_knn(self, mode):
        """"strict" (the reference's list order, bit-identical irradiance) or "fast" (same photon set; default)."""
        abi.check(self.L.rt_scene_set_option(self.h, abi.RT_OPT_KNN, abi.RT_KNN_STRICT if mode == "strict" else abi.RT_KNN_FAST))

    def info(self):
        i = abi.rt_scene_info()
        abi.check(self.L.rt_scene_info_get(self.h, C.byref(i)))
        return i

    # ---- host-buffer entry points ------------------------------------------
    def trace_closest(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = np.zeros(rays.shape[0], abi.hit_dtype)
        abi.check(self.L.rt_trace_closest(self.h, rays.ctypes.data, rays.shape[0], out.ctypes.data))
        return out

    def trace_any(self, rays, tmax=None):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = np.zeros(rays.shape[0], np.uint8)
        t = None if tmax is None else np.ascontiguousarray(tmax, np.float32)
        abi.check(self.L.rt_trace_any(self.h, rays.ctypes.data, None if t is None else t.ctypes.data,
                                      rays.shape[0], out.ctypes.data))
        return out

    def trace_work(self, d_rays_ptr, n, any_hit=False, d_tmax_ptr=None, stream_ptr=None):
        """work counters of the traversal on device-resident rays: (box tests, triangle records read,
        exact tests, record fetches) -- measurement only."""
        c = (C.c_int64 * 4)()
        abi.check(self.L.rt_trace_work_dev(self.h, d_rays_ptr, d_tmax_ptr, n, 1 if any_hit else 0, c, stream_ptr))
        return tuple(int(x) for x in c)

    def render(self, cam, params, want_ids=False):
        n = cam.resx * cam.resy
        rgb = np.zeros(n * 3, np.uint8)
        ids = np.zeros(n, np.int32) if want_ids else None
        st = abi.rt_stats()
        abi.check(self.L.rt_render(self.h, C.byref(cam), C.byref(params), rgb.ctypes.data,
                                   None if ids is None else ids.ctypes.data, C.byref(st)))
        return rgb.reshape(cam.resy, cam.resx, 3), ids, st

    def photonmap_emit_until(self, e, target, batch=None):
        """The `while (stored < max)` loops of final.cpp:1057 / :1090: emission indices 0,1,2,... in batches until
        `target` photons are stored; returns the first `target` of them in emission order (what AddPhoton keeps,
        photon.cpp:837) and the number of emission indices consumed."""
        batch = int(batch or max(4096, target // 4))
        parts, have, first = [], 0, 0
        while have < target:
            ph, n = self.photonmap_emit(e, first, batch, cap=2 * batch + 64)
            if n > ph.size:
                raise abi.RtError(abi.RT_E_LIMIT if hasattr(abi, "RT_E_LIMIT") else -4, "emission batch overflowed its buffer")
            parts.append(ph); have += ph.size; first += batch
            if first > 1000 * max(target, 1) + 10 * batch:
                raise RuntimeError("photon emission stores (almost) nothing: no target reachable")
        out = np.concatenate(parts)[:target] if parts else np.zeros(0, abi.photon_dtype)
        return out, first

    def render_radiance(self, cam, params):
        """the un-quantised frame (float [resy][resx][3])"""
        rad = np.zeros(cam.resx * cam.resy * 3, np.float32)
        st = abi.rt_stats()
        abi.check(self.L.rt_render_radiance(self.h, C.byref(cam), C.byref(params), rad.ctypes.data, C.byref(st)))
        return rad.reshape(cam.resy, cam.resx, 3), st

    def photonmap_upload(self, which, heap_order):
        a = np.ascontiguousarray(heap_order)
        assert a.dtype == abi.photon_dtype
        abi.check(self.L.rt_photonmap_upload(self.h, which, a.ctypes.data, a.shape[0] - 1))

    def photonmap_build(self, which, photons, power_scale=1.0):
        """photons[1..n] as AddPhoton leaves them -> balanced (+ scaled as final.cpp:1126 does) heap, uploaded."""
        a = np.ascontiguousarray(photons)
        assert a.dtype == abi.photon_dtype
        abi.check(self.L.rt_photonmap_build(self.h, which, a.ctypes.data, a.shape[0] - 1, float(power_scale)))
        return a

    def photonmap_build_dev(self, which, d_photons, n, power_scale=1.0, stream=None):
        """the same build with the photons resident on the device: d_photons = address of rt_photon[n + 1] (slot 0 unused),
        left in the reference's heap order (include/rt_hip.h: rt_photonmap_build_dev)."""
        abi.check(self.L.rt_photonmap_build_dev(self.h, which, int(d_photons), int(n), float(power_scale), stream))

    def photonmap_emit(self, e, first, count, cap=None):
        """photons stored by emission indices [first, first+count), in emission order (0-based array)."""
        cap = int(count) * 2 + 64 if cap is None else int(cap)
        out = np.zeros(cap, abi.photon_dtype)
        n = C.c_int64(0)
        abi.check(self.L.rt_photonmap_emit(self.h, C.byref(e), int(first), int(count), out.ctypes.data, cap, C.byref(n)))
        return out[:min(n.value, cap)], n.value

    def photon_knn(self, which, pos, k, max_dist):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        q = pos.shape[0]
        idx = np.zeros((q, k), np.int32); d2 = np.zeros((q, k), np.float32)
        found = np.zeros(q, np.int32); r2 = np.zeros(q, np.float32)
        abi.check(self.L.rt_photon_knn(self.h, which, pos.ctypes.data, q, k, float(max_dist), idx.ctypes.data,
                                       d2.ctypes.data, found.ctypes.data, r2.ctypes.data))
        return idx, d2, found, r2

    def photon_knn_set(self, which, pos, k, max_dist):
        """the photons of photon_knn in unspecified order (the scene's kNN option decides the kernel)"""
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        q = pos.shape[0]
        idx = np.zeros((q, k), np.int32); d2 = np.zeros((q, k), np.float32)
        found = np.zeros(q, np.int32); r2 = np.zeros(q, np.float32)
        abi.check(self.L.rt_photon_knn_set(self.h, which, pos.ctypes.data, q, k, float(max_dist), idx.ctypes.data,
                                           d2.ctypes.data, found.ctypes.data, r2.ctypes.data))
        return idx, d2, found, r2

    def photon_irradiance(self, which, pos, nrm, k, max_dist):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        nrm = np.ascontiguousarray(nrm, np.float32).reshape(-1, 3)
        rgb = np.zeros((pos.shape[0], 3), np.float32)
        abi.check(self.L.rt_photon_irradiance(self.h, which, pos.ctypes.data, nrm.ctypes.data, pos.shape[0], k,
                                              float(max_dist), rgb.ctypes.data))
        return rgb
