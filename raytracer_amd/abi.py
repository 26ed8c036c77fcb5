"""ctypes mirror of include/rt_hip.h and loader of librt_hip.so.

The library is the product; this module only binds it.  There is no fallback:
if the shared object is missing, loading raises, and every compute entry point
of the library itself fails with RT_E_NODEVICE when no gfx950 GPU is usable.
"""
import ctypes as C
import os
import sys

import numpy as np

RT_ABI_VERSION = 3
RT_PRIM_TRIANGLE, RT_PRIM_SPHERE, RT_PRIM_DISABLED = 0, 1, 255
RT_LIGHT_POINT, RT_LIGHT_RECT = 0, 1
RT_SCENE_CULLING, RT_SCENE_KD_REFERENCE = 1, 2
RT_RENDER_WHITTED, RT_RENDER_PHOTON = 0, 1
RT_PMAP_GLOBAL, RT_PMAP_CAUSTIC = 0, 1
RT_OPT_TRAVERSAL, RT_TRAVERSAL_STRICT, RT_TRAVERSAL_FAST, RT_TRAVERSAL_LANE, RT_TRAVERSAL_BVH = 1, 0, 1, 2, 3
RT_OPT_KNN, RT_KNN_STRICT, RT_KNN_FAST = 2, 0, 1
RT_OPT_LANE_DEPTH = 3
RT_OPT_KNN_STATS = 4
RT_OPT_FUSED_PRIMARY = 5
RT_OPT_RAY_SORT = 6
RT_OPT_RENDER_OVERLAP = 7
RT_OPT_PATH_DEPTH = 8
RT_NOALLOC_MAX_QUERIES = 262144
RT_E_INVALID, RT_E_NODEVICE, RT_E_NOMEM, RT_E_LIMIT, RT_E_STATE, RT_E_HIP = -1, -2, -3, -4, -5, -6

_f, _i32, _i64, _u8 = C.c_float, C.c_int32, C.c_int64, C.c_uint8


class rt_material(C.Structure):
    _fields_ = [("surf", _f * 3), ("Ka", _f), ("Kd", _f), ("Ks", _f), ("specularity", _f),
                ("reflect", _f), ("refract", _f), ("Nr", _f)]


class rt_light(C.Structure):
    _fields_ = [("type", _i32), ("pos", _f * 3), ("color", _f * 3), ("intensity", _f),
                ("dir1", _f * 3), ("dir2", _f * 3), ("size1", _f), ("size2", _f), ("n1", _i32), ("n2", _i32)]


class rt_scene_desc(C.Structure):
    _fields_ = [("abi_version", _i32), ("flags", C.c_uint32), ("nprim", _i64), ("prims", C.c_void_p),
                ("prim_type", C.c_void_p), ("prim_material", C.c_void_p), ("nmaterial", _i32),
                ("materials", C.c_void_p), ("nobject", _i32), ("obj_first", C.c_void_p),
                ("obj_count", C.c_void_p), ("nlight", _i32), ("lights", C.c_void_p),
                ("tolerance_depth", C.c_double), ("shadow_bias", _f), ("refract_bias", _f), ("device", _i32)]


class rt_scene_info(C.Structure):
    _fields_ = [("kd_nodes_built", _i64), ("kd_max_depth", _i32), ("flat_nodes", _i64), ("leaf_refs", _i64),
                ("ntri", _i64), ("nsphere", _i64), ("nloose", _i64), ("device_bytes", _i64)]


class rt_camera(C.Structure):
    _fields_ = [("center", _f * 3), ("corners", (_f * 3) * 4), ("xaxis", _f * 3), ("yaxis", _f * 3),
                ("resx", _i32), ("resy", _i32)]


class rt_params(C.Structure):
    _fields_ = [("mode", _i32), ("max_depth", _i32), ("ambient_intensity", _f), ("background", _f * 3),
                ("start_nr", _f), ("primary_only", _i32), ("tile_size", _i32), ("tile_rank", _i32),
                ("tile_nranks", _i32), ("gather_samples", _i32), ("knn_k", _i32), ("knn_radius_global", _f),
                ("knn_radius_caustic", _f), ("irradiance_max_depth", _i32), ("aa_taps", _i32),
                ("count_work", _i32), ("seed", C.c_uint64), ("spp", _i32), ("lens_radius", _f), ("focal_distance", _f),
                ("light_radius", _f), ("glossy", _f), ("packed_output", _i32)]


class rt_emit_params(C.Structure):
    _fields_ = [("which", _i32), ("light", _i32), ("power", _f * 3), ("start_nr", _f), ("max_depth", _i32),
                ("ncaustic_sphere", _i32), ("caustic_sphere", (_f * 4) * 4), ("caustic_floor_y", _f),
                ("seed", C.c_uint64)]


class rt_stats(C.Structure):
    _fields_ = [("rays_primary", _i64), ("rays_shadow", _i64), ("rays_reflect", _i64), ("rays_refract", _i64),
                ("rays_gather", _i64), ("knn_lookups", _i64), ("passes", _i64), ("kernel_ms", C.c_double),
                ("trace_closest_ms", C.c_double), ("trace_any_ms", C.c_double),
                ("trace_closest_launches", _i64), ("trace_any_launches", _i64),
                ("closest_work", _i64 * 4), ("any_work", _i64 * 4),
                ("knn_ms", C.c_double), ("knn_launches", _i64), ("lane_closest_ms", C.c_double), ("lane_any_ms", C.c_double),
                ("lane_closest_launches", _i64), ("lane_any_launches", _i64), ("lane_closest_rays", _i64),
                ("lane_any_rays", _i64), ("lane_closest_work", _i64 * 4), ("lane_any_work", _i64 * 4),
                ("path_ms", C.c_double), ("path_launches", _i64), ("path_roots", _i64), ("path_closest_rays", _i64),
                ("path_shadow_rays", _i64), ("path_work", _i64 * 4)]


# numpy views of the POD records
ray_dtype = np.dtype([("o", np.float32, 3), ("d", np.float32, 3)])
hit_dtype = np.dtype([("prim", np.int32), ("len", np.float32), ("p", np.float32, 3)])
photon_dtype = np.dtype([("position", np.float32, 3), ("inDir", np.float32, 3), ("power", np.float32, 3),
                         ("phi", np.uint8), ("theta", np.uint8), ("plane", np.int16)])
assert ray_dtype.itemsize == 24 and hit_dtype.itemsize == 20 and photon_dtype.itemsize == 40

# every symbol include/rt_hip.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "rt_init", "rt_last_error", "rt_abi_version",
    "rt_scene_create", "rt_scene_destroy", "rt_scene_info_get", "rt_scene_set_option", "rt_scene_knn_stats",
    "rt_host_flatten", "rt_host_flat_get", "rt_host_flat_free", "rt_host_photon_balance",
    "rt_trace_closest", "rt_trace_closest_dev", "rt_trace_any", "rt_trace_any_dev", "rt_trace_work_dev",
    "rt_photonmap_upload", "rt_photonmap_build", "rt_photonmap_build_dev", "rt_photonmap_emit", "rt_photonmap_emit_dev",
    "rt_photonmap_image_bytes", "rt_photonmap_export_dev", "rt_photonmap_import_dev",
    "rt_photon_knn", "rt_photon_knn_dev", "rt_photon_knn_set", "rt_photon_irradiance", "rt_photon_irradiance_dev",
    "rt_photon_irradiance_noalloc_dev",
    "rt_render", "rt_render_dev", "rt_render_radiance", "rt_render_pixel_list", "rt_generate_primary_dev",
]

_LIB = None


def lib_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "librt_hip.so")


def load():
    """Load librt_hip.so (built in-tree by raytracer_amd/csrc/Makefile).  Raises if absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # PyTorch-ROCm wheels bundle their own HIP runtime under the same soname; whichever copy is
    # mapped first serves the whole process, and torch cannot initialise on top of the system
    # copy.  Host programs that also use torch must therefore import it first -- do it for them.
    if "torch" not in sys.modules and not os.environ.get("RT_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    p = lib_path()
    if not os.path.exists(p):
        raise RuntimeError(f"{p} is missing: build it with `make -C raytracer_amd/csrc` "
                           "(there is no CPU fallback)")
    L = C.CDLL(p)
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
    L.rt_last_error.restype = C.c_char_p
    L.rt_init.argtypes = [i32]
    L.rt_scene_create.argtypes = [C.POINTER(rt_scene_desc), C.POINTER(vp)]
    L.rt_scene_destroy.argtypes = [vp]
    L.rt_scene_info_get.argtypes = [vp, C.POINTER(rt_scene_info)]
    L.rt_scene_set_option.argtypes = [vp, i32, i32]
    L.rt_scene_knn_stats.argtypes = [vp, vp]
    L.rt_host_flatten.argtypes = [C.POINTER(rt_scene_desc), C.POINTER(vp), C.POINTER(rt_scene_info)]
    L.rt_host_flat_get.argtypes = [vp, i32, vp, i64]
    L.rt_host_flat_get.restype = i64
    L.rt_host_flat_free.argtypes = [vp]
    L.rt_host_flat_free.restype = None
    L.rt_host_photon_balance.argtypes = [vp, i64]
    L.rt_trace_closest.argtypes = [vp, vp, i64, vp]
    L.rt_trace_closest_dev.argtypes = [vp, vp, i64, vp, vp]
    L.rt_trace_any.argtypes = [vp, vp, vp, i64, vp]
    L.rt_trace_any_dev.argtypes = [vp, vp, vp, i64, vp, vp]
    L.rt_trace_work_dev.argtypes = [vp, vp, vp, i64, i32, vp, vp]
    L.rt_generate_primary_dev.argtypes = [vp, C.POINTER(rt_camera), i64, i64, C.c_float, C.c_float, vp, vp]
    if hasattr(L, "rt_render"):
        L.rt_render.argtypes = [vp, C.POINTER(rt_camera), C.POINTER(rt_params), vp, vp, C.POINTER(rt_stats)]
        L.rt_render_radiance.argtypes = [vp, C.POINTER(rt_camera), C.POINTER(rt_params), vp, C.POINTER(rt_stats)]
        L.rt_render_dev.argtypes = [vp, C.POINTER(rt_camera), C.POINTER(rt_params), vp, vp, C.POINTER(rt_stats), vp]
        L.rt_render_pixel_list.argtypes = [C.POINTER(rt_camera), C.POINTER(rt_params), vp, i64, C.POINTER(i64)]
    if hasattr(L, "rt_photonmap_upload"):
        L.rt_photonmap_upload.argtypes = [vp, i32, vp, i64]
        L.rt_photonmap_build.argtypes = [vp, i32, vp, i64, C.c_float]
        L.rt_photonmap_build_dev.argtypes = [vp, i32, vp, i64, C.c_float, vp]
        L.rt_photonmap_emit.argtypes = [vp, C.POINTER(rt_emit_params), i64, i64, vp, i64, C.POINTER(i64)]
        L.rt_photonmap_emit_dev.argtypes = [vp, C.POINTER(rt_emit_params), i64, i64, vp, i64, C.POINTER(i64), vp]
        L.rt_photonmap_image_bytes.argtypes = [vp, i32, C.POINTER(i64), C.POINTER(i64)]
        L.rt_photonmap_export_dev.argtypes = [vp, i32, vp, i64, vp, vp]
        L.rt_photonmap_import_dev.argtypes = [vp, i32, vp, i64, i64, vp, vp]
        L.rt_photon_knn.argtypes = [vp, i32, vp, i64, i32, C.c_float, vp, vp, vp, vp]
        L.rt_photon_knn_set.argtypes = [vp, i32, vp, i64, i32, C.c_float, vp, vp, vp, vp]
        L.rt_photon_knn_dev.argtypes = [vp, i32, vp, i64, i32, C.c_float, vp, vp, vp, vp, vp]
        L.rt_photon_irradiance.argtypes = [vp, i32, vp, vp, i64, i32, C.c_float, vp]
        L.rt_photon_irradiance_dev.argtypes = [vp, i32, vp, vp, i64, i32, C.c_float, vp, vp]
        L.rt_photon_irradiance_noalloc_dev.argtypes = [vp, i32, vp, vp, i64, i32, C.c_float, vp, vp]
    _LIB = L
    return L


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"librt_hip error {code}: {msg}")
        self.code = code


def check(rc):
    if rc < 0:
        raise RtError(rc, load().rt_last_error().decode("utf-8", "replace"))
    return rc
