"""Time rt_scene_create on a synthetic mesh of about N triangles (RT_BUILD_TIMING=1 prints the builders' phases).  A small
scene is created first, so that the HIP context and the code object are loaded before the clock starts.
usage: time_scene_build.py N [repeats]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytracer_amd import synth, host
n = int(float(sys.argv[1])); rep = int(sys.argv[2]) if len(sys.argv) > 2 else 1
assert n <= 20_000_000
t0 = time.perf_counter(); host.Scene(synth.mesh_scene(8, 8)).close(); print(f"first (small) scene, with the HIP start-up: {time.perf_counter() - t0:.2f} s")
nv = max(4, int((n / 4) ** 0.5)); nu = max(4, n // (2 * nv))
t0 = time.perf_counter(); sd = synth.mesh_scene(nu, nv); t1 = time.perf_counter()
print(f"triangles {2 * nu * nv}: mesh {t1 - t0:.2f} s")
for _ in range(rep):
    t1 = time.perf_counter(); sc = host.Scene(sd); t2 = time.perf_counter()
    print(f"triangles {2 * nu * nv}: rt_scene_create {t2 - t1:.2f} s")
    sc.close()
