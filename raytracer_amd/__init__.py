"""raytracer_amd: MI355X-native hot path of clotha87762/RayTracer (KD-tree ray
traversal, ray/triangle + ray/sphere intersection, Whitted shading, photon-map
kNN radiance estimate) behind the C ABI of include/rt_hip.h.

Python here is host plumbing (scene parsing, ctypes bindings); all tracing runs
in librt_hip.so on a gfx950 GPU.  There is no CPU fallback.
"""
from . import abi, scene, host  # noqa: F401
from .scene import SceneData, parse_scene_file, parse_scene_text  # noqa: F401
from .host import Scene, camera_setup, camera_for, default_params  # noqa: F401

__all__ = ["abi", "scene", "host", "SceneData", "parse_scene_file", "parse_scene_text", "Scene",
           "camera_setup", "camera_for", "default_params"]
