"""Time the HOST half of rt_scene_create (reference KD-trees + the build's own BVH, no device) on a synthetic mesh of about
N triangles; RT_BUILD_TIMING=1 prints the phases.  Runs without a GPU.  Prints a digest of every array built (the same
for every RT_HOST_THREADS: tests/test_host_build.py).
usage: time_host_build.py N [repeats]"""
import ctypes as C, os, resource, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytracer_amd import synth, abi
n = int(float(sys.argv[1])); rep = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nv = max(4, int((n / 4) ** 0.5)); nu = max(4, n // (2 * nv))
sd = synth.mesh_scene(nu, nv)
L = abi.load()
L.rt_debug_host_build.argtypes = [C.POINTER(abi.rt_scene_desc), C.POINTER(C.c_uint64)]; L.rt_debug_host_build.restype = C.c_int
d, keep = sd.make_desc()
for _ in range(rep):
    dig = C.c_uint64(0)
    r0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.perf_counter()
    abi.check(L.rt_debug_host_build(C.byref(d), C.byref(dig) if os.environ.get('RT_DIGEST', '1') != '0' else None))
    t1 = time.perf_counter(); r1 = resource.getrusage(resource.RUSAGE_SELF)
    print(f"triangles {2 * nu * nv}: host build {t1 - t0:.2f} s wall, {r1.ru_utime - r0.ru_utime + r1.ru_stime - r0.ru_stime:.2f} s CPU "
          f"(user {r1.ru_utime - r0.ru_utime:.2f}), digest {dig.value:016x}")
