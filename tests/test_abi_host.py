"""CPU tests (no GPU): the C-ABI library loads and exports every symbol of include/rt_hip.h,
fails loudly without a device, and the host-side logic above/below the boundary (parser,
camera, KD builder, photon balance, PPM writer) agrees with the reference's fixtures."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as ol
from raytracer_amd import abi, host, scene

ROOT = ol.ROOT


def _load(name):
    with np.load(os.path.join(ol.GOLDEN, name)) as z:
        return {k: np.ascontiguousarray(z[k]) for k in z.files}


def test_library_exports_every_declared_symbol():
    L = abi.load()
    hdr = open(os.path.join(ROOT, "include", "rt_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(abi.SYMBOLS), declared ^ set(abi.SYMBOLS)
    for s in sorted(declared):
        assert hasattr(L, s), f"librt_hip.so does not export {s}"
    assert L.rt_abi_version() == abi.RT_ABI_VERSION
    H = host.host_lib()
    hh = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rt_host.h")).read(), flags=re.S)
    assert set(re.findall(r"\b(rth_[a-z0-9_]+)\s*\(", hh)) == set(host.HOST_SYMBOLS)
    for s in host.HOST_SYMBOLS:
        assert hasattr(H, s)


def test_struct_sizes_match_header(tmp_path):
    # sizes printed by a C program that includes include/rt_hip.h (x86-64 SysV), compiled here with gcc
    import subprocess
    names = ["rt_material", "rt_camera", "rt_light", "rt_scene_desc", "rt_scene_info", "rt_params", "rt_stats",
             "rt_photon", "rt_ray", "rt_hit", "rt_prim"]
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "rt_hip.h"\nint main(void){' +
                   "".join(f'printf("%zu\\n", sizeof({n}));' for n in names) + "return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    want = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    got = [C.sizeof(getattr(abi, n)) for n in names[:7]] + [abi.photon_dtype.itemsize, abi.ray_dtype.itemsize,
                                                            abi.hit_dtype.itemsize, 48]
    assert got == want


def test_product_has_no_cpu_path_and_says_so(monkey_sd):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = abi.load()
    assert L.rt_init(1) == -2                                   # RT_E_NODEVICE
    assert b"no" in L.rt_last_error().lower()
    with pytest.raises(abi.RtError) as e:
        host.Scene(monkey_sd)
    assert e.value.code == -2


def test_parser_reads_monkey(monkey_sd):
    sd = monkey_sd
    assert sd.nprim == 970 and list(sd.obj_first) == [0, 2] and list(sd.obj_count) == [2, 968]
    assert (sd.resx, sd.resy, sd.fov) == (256, 256, 45.0) and len(sd.lights) == 1
    assert sd.materials.shape == (2, 10) and np.allclose(sd.materials[0][:3], [1, 1, 0.5])
    # normals are stored normalised once (hw4.cpp:637)
    n = sd.prims[:, 9:12]
    assert np.abs(np.linalg.norm(n, axis=1) - 1).max() < 1e-6


def test_parser_object_rules():
    base = "E 0 0 -1\nV 0 0 1\nF 45\nR 8 8\nM 1 1 1 1 1 1 1 0 0 1\n"
    tri = "T 0 0 0 1 0 0 0 1 0 0 0 1\n"
    sd = scene.parse_scene_text(base + tri + tri)               # no `O`: one object holds all (hw4.cpp:754)
    assert list(sd.obj_first) == [0] and list(sd.obj_count) == [2]
    sd = scene.parse_scene_text(base + "O\n" + tri + "O\n" + tri)   # empty `O` ignored; trailing mesh unreachable
    assert list(sd.obj_count) == [1] and sd.prim_type[1] == abi.RT_PRIM_DISABLED
    sd = scene.parse_scene_text(base + "S 0 0 1 0.5\n" + tri, flavour="final")
    assert sd.prim_type[0] == abi.RT_PRIM_SPHERE and len(sd.obj_first) == 0
    sd = scene.parse_scene_file(os.path.join(ol.GOLDEN, "final_input.txt"), flavour="final")
    assert sd.nprim == 14 and sd.lights[0]["type"] == abi.RT_LIGHT_RECT and (sd.lights[0]["n1"], sd.lights[0]["n2"]) == (20, 20)


def test_scene_text_round_trip(monkey_sd):
    sd2 = scene.parse_scene_text(scene.write_scene_text(monkey_sd))
    assert np.array_equal(sd2.prims[:, :9], monkey_sd.prims[:, :9]) and np.array_equal(sd2.obj_count, monkey_sd.obj_count)
    assert np.array_equal(sd2.materials[sd2.prim_material], monkey_sd.materials[monkey_sd.prim_material])


def test_host_camera_matches_reference(monkey_sd):
    for name, res in (("monkey256_hits.npz", (256, 256)), ("monkey4k_hits_digest.npz", (3840, 2160))):
        cam = host.camera_for(monkey_sd, res)
        assert np.array_equal(ol.camera_to_array(cam), _load(name)["cam"])


def test_host_kd_builder_matches_reference_tree(monkey_sd):
    g = _load("monkey_kd.npz")
    f = host.host_flatten(monkey_sd)
    assert (f["info"].kd_nodes_built, f["info"].kd_max_depth) == (138, 9)       # executionTime_monkey.txt:4-5
    assert np.array_equal(f["lohi"], g["lohi"]) and np.array_equal(f["skip"], g["skip"])
    assert np.array_equal(f["meta"] & 0xFF, g["depth"]) and np.array_equal((f["meta"] >> 8) & 1, g["leaf"])
    assert np.array_equal(f["leaf_prim"], g["leaf_prims"])
    leaves = np.flatnonzero(g["leaf"])
    assert np.array_equal(np.cumsum(f["leaf_count"][leaves])[:-1], f["leaf_first"][leaves][1:])
    # leaf records hold v0 and the two edges computed as hw4.cpp:177-178 does
    p = monkey_sd.prims[f["leaf_prim"]]
    assert np.array_equal(f["leaf_tri"][:, 0:3], p[:, 0:3])
    assert np.array_equal(f["leaf_tri"][:, 3:6], p[:, 3:6] - p[:, 0:3])
    assert np.array_equal(f["leaf_tri"][:, 6:9], p[:, 6:9] - p[:, 0:3])


def test_host_kd_builder_matches_oracle_on_random_soups():
    from raytracer_amd.scene import SceneData
    rng = np.random.default_rng(17)
    for n, nobj in ((1, 1), (2, 1), (37, 2), (1500, 3)):
        c = rng.uniform(-0.4, 0.4, (n, 1, 3)); v = (c + rng.normal(scale=0.05, size=(n, 3, 3))).astype(np.float32)
        if n > 30:
            v[: n // 3, :, 1] = -0.5                                  # a flat slab: zero-extent boxes
        prims = np.concatenate([v.reshape(n, 9), np.tile(np.float32([0, 1, 0]), (n, 1))], axis=1)
        cuts = np.linspace(0, n, nobj + 1).astype(np.int64)
        sd = SceneData(prims=prims, prim_type=np.zeros(n, np.uint8), prim_material=np.zeros(n, np.int32),
                       materials=np.array([[1, 1, 1, 1, 0, 0, 1, 0, 0, 1]], np.float32), obj_first=cuts[:-1],
                       obj_count=np.diff(cuts), eye=np.zeros(3, np.float32), view=np.array([0, 0, 1], np.float32))
        o = ol.OracleScene(sd)
        lohi, leaf, depth, skip, lists = ol.canonical_from_tree(o.kd_tree())
        f = host.host_flatten(sd)
        oi = o.info()
        assert (f["info"].kd_nodes_built, f["info"].kd_max_depth) == (oi.kd_nodes_built, oi.kd_max_depth)
        assert np.array_equal(f["lohi"], lohi) and np.array_equal(f["skip"], skip)
        assert np.array_equal(f["leaf_prim"], np.concatenate([l for l in lists if l is not None]))
        o.close()


def test_host_photon_balance_matches_reference():
    g = _load("photon_golden.npz")
    stored = g["stored"].view(abi.photon_dtype).copy()
    heap = g["heap"].view(abi.photon_dtype)
    n = stored.size - 1
    abi.check(abi.load().rt_host_photon_balance(stored.ctypes.data, n))
    for f in ("position", "inDir", "phi", "theta", "plane"):
        assert np.array_equal(stored[f][1:], heap[f][1:]), f
    for n_small in (0, 1, 2, 3, 7):                                     # degenerate maps do not crash
        a = np.zeros(n_small + 1, abi.photon_dtype); a["position"] = np.arange(3 * (n_small + 1)).reshape(-1, 3)
        abi.check(abi.load().rt_host_photon_balance(a.ctypes.data, n_small))


def test_ppm_writer_and_log(tmp_path):
    rgb = np.arange(4 * 3 * 3, dtype=np.uint8)
    p = tmp_path / "colorOutput.ppm"
    host.write_ppm(str(p), rgb, 4, 3)
    assert p.read_bytes() == b"P6 4 3 255\n" + rgb.tobytes()             # imageIO.cpp:64-74
    q = tmp_path / "executionTime_x.txt"
    host.host_lib().rth_write_execution_time(os.fsencode(str(q)), 256, 256, 1, 0.5, 138, 9, 133.332)
    assert q.read_text().splitlines() == ["resolution:256X256", "SSAA ratio:1", "elapse time:0.5 seconds",
                                          "total kd-tree node count:138", "max depth of kd-tree subdivision:9",
                                          "average ray-mesh Intersection check count:133.332"]
    src = np.random.default_rng(0).integers(0, 256, (6, 4, 3), dtype=np.uint8)   # 4x6 image, ratio 2
    dst = np.zeros((3, 2, 3), np.uint8)
    host.host_lib().rth_ssaa_downsample(src.ctypes.data, 4, 6, 2, dst.ctypes.data)
    want = src.reshape(3, 2, 2, 2, 3).astype(np.int32).transpose(0, 2, 1, 3, 4).reshape(3, 2, 4, 3).sum(2) // 4
    assert np.array_equal(dst, want)


def test_host_builders_do_not_depend_on_the_thread_count(tmp_path):
    """the KD build / flatten and the photon balance fork threads for large subtrees (RT_HOST_THREADS): the arrays must
    be the same whatever the schedule.  Separate processes because the thread count is read once."""
    import subprocess
    import sys
    code = r'''
import sys, hashlib
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
from raytracer_amd import abi, host, synth
f = host.host_flatten(synth.mesh_scene(300, 150))
h = hashlib.sha256()
for k in ("lohi", "skip", "meta", "leaf_first", "leaf_count", "leaf_prim", "loose", "leaf_tri"):
    h.update(np.ascontiguousarray(f[k]).tobytes())
rng = np.random.default_rng(1)
n = 300000
a = np.zeros(n + 1, abi.photon_dtype)
a["position"][1:] = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
a["position"][1:n // 10, 1] = 0.25
abi.check(abi.load().rt_host_photon_balance(a.ctypes.data, n))
h.update(a.tobytes())
print(f["info"].kd_nodes_built, f["info"].kd_max_depth, h.hexdigest())
''' % (ROOT, os.path.join(ROOT, "tests"))
    outs = []
    for nt in ("1", "3", "8"):
        r = subprocess.run([sys.executable, "-c", code], env={**os.environ, "RT_HOST_THREADS": nt}, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1] == outs[2], outs
