"""The C++ front end's scene parser (include/rt_host.h rth_scene_*, raytracer_amd/host/scene_parser.cpp) against the
Python mirror of the reference's token loops (raytracer_amd/scene.py; hw4.cpp:538-765, final.cpp:509-735): same arrays,
bit for bit, whatever the thread count; binary cache round trip; the drop-in programs exist and refuse to run without a GPU.
CPU only."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from raytracer_amd import host, scene, synth

ROOT = ol.ROOT
GOLD = os.path.join(ROOT, "tests", "golden")


def same(a: scene.SceneData, b: scene.SceneData):
    assert a.prims.shape == b.prims.shape
    assert a.prims.tobytes() == b.prims.tobytes()
    assert np.array_equal(a.prim_type, b.prim_type)
    assert np.array_equal(a.prim_material, b.prim_material)
    assert a.materials.tobytes() == np.ascontiguousarray(b.materials, np.float32).tobytes()
    assert np.array_equal(a.obj_first, b.obj_first) and np.array_equal(a.obj_count, b.obj_count)
    assert np.asarray(a.eye, np.float32).tobytes() == np.asarray(b.eye, np.float32).tobytes()
    assert np.asarray(a.view, np.float32).tobytes() == np.asarray(b.view, np.float32).tobytes()
    assert np.float32(a.fov) == np.float32(b.fov) and (a.resx, a.resy) == (b.resx, b.resy)
    assert len(a.lights) == len(b.lights)
    for la, lb in zip(a.lights, b.lights):
        assert la["type"] == lb.get("type", 0)
        for k, dflt in (("pos", None), ("color", (1, 1, 1)), ("dir1", (1, 0, 0)), ("dir2", (0, 0, 1))):
            assert np.array_equal(np.asarray(la[k], np.float32), np.asarray(lb.get(k, dflt), np.float32)), k
        assert np.float32(la["intensity"]) == np.float32(lb.get("intensity", 1.0))
        assert np.float32(la["size1"]) == np.float32(lb.get("size1", 0.0)) and np.float32(la["size2"]) == np.float32(lb.get("size2", 0.0))
        assert (la["n1"], la["n2"]) == (lb.get("n1", 1), lb.get("n2", 1))


@pytest.mark.parametrize("name,flavour", [("monkey.txt", "hw4"), ("final_input.txt", "final")])
def test_bundled_scenes_parse_like_the_mirror(name, flavour):
    p = os.path.join(GOLD, name)
    same(host.parse_scene_native(path=p, flavour=flavour), scene.parse_scene_file(p, flavour))


def test_object_rules_and_unknown_tokens():
    # an empty object is dropped, meshes after the last `O` are disabled (hw4.cpp:729-768), unknown tokens are skipped,
    # the default material applies before the first M, equal M lines share one material, E/V/F/R: the last one wins
    txt = """E 0 0 0  V 0 0 1 F 45 R 10 10 junk 1 2
    O
    T 0 0 1  1 0 1  0 1 1   0 0 -2
    M 1 0 0 .5 .5 .5 6 0 0 1
    T 0 0 2  1 0 2  0 1 2   0 0 -1   O  O
    M 0 1 0 .5 .5 .5 6 0 0 1  S 0 0 5 1
    M 1 0 0 .5 .5 .5 6 0 0 1  T 0 0 3  1 0 3  0 1 3   0 3 -1 O
    T 0 0 4  1 0 4  0 1 4   0 0 -1
    L 1 2 3 ML 4 5 6 .1 .2 .3 2.5 RectL 0 1 0 1 1 4 4 R 32 16 E 1 1 1"""
    for fl in ("hw4", "final"):
        same(host.parse_scene_native(text=txt, flavour=fl), scene.parse_scene_text(txt, fl))
    s = host.parse_scene_native(text=txt, flavour="hw4")
    assert list(s.obj_first) == [0, 2] and list(s.obj_count) == [2, 2] and s.prim_type[4] == 255
    assert list(s.prim_material) == [0, 1, 2, 1, 1] and (s.resx, s.resy) == (32, 16)
    same(host.parse_scene_native(text="", flavour="hw4"), scene.parse_scene_text("", "hw4"))
    same(host.parse_scene_native(text="T 1 2 3 4 5 6 7 8 9 0 0 1", flavour="hw4"), scene.parse_scene_text("T 1 2 3 4 5 6 7 8 9 0 0 1", "hw4"))


def test_numbers_are_read_like_atof():
    toks = ["1", "-1", "+2.5", ".5", "-.25", "3.", "1e-3", "1E+2", "-4.75e-1", "0.1", "0.3333333333333333", "123456789.125",
            "1.23456789012345678", "9007199254740993", "1e22", "1e23", "1e-22", "1e-23", "1e38", "1e39", "1e-46", "4.9e-324",
            "1e400", "-1e400", "000.001000", "0", "-0", "1.5abc", "abc", "1e", "1e+", "7.0000001", "16777217", "0.1e1",
            "3.4028235e38", "1.17549435e-38", "2.5e-45", "123456789012345678901234567890", "0.000000000000000000001234"]
    txt = "F 1 " + " ".join("S %s 0 0 1" % t for t in toks)
    a = host.parse_scene_native(text=txt, flavour="hw4")
    b = scene.parse_scene_text(txt, "hw4")
    assert a.prims[:, 0].tobytes() == b.prims[:, 0].tobytes(), [(t, x, y) for t, x, y in zip(toks, a.prims[:, 0], b.prims[:, 0]) if x != y]
    rng = np.random.default_rng(5)
    vals = np.concatenate([rng.normal(size=3000), rng.normal(size=3000) * 1e-4, rng.normal(size=3000) * 1e6])
    for fmt in ("%.9g", "%.17g", "%.6f", "%e", "%.3f"):
        t2 = "F 1 " + " ".join(("S " + fmt + " 0 0 1") % v for v in vals)
        a = host.parse_scene_native(text=t2, flavour="hw4")
        assert a.prims[:, 0].tobytes() == np.array([float(fmt % v) for v in vals], np.float32).tobytes()


def test_threads_do_not_change_the_result(tmp_path):
    sd = synth.mesh_scene(120, 60, seed=3)                       # 14 402 triangles, ~2 MB of text, 2 objects
    txt = scene.write_scene_text(sd)
    assert len(txt) > (1 << 20)
    one = host.parse_scene_native(text=txt, flavour="hw4", nthreads=1)
    assert one.nprim == sd.nprim and list(one.obj_count) == list(sd.obj_count)
    assert one.prims[:, :9].tobytes() == sd.prims[:, :9].tobytes()          # %.9g round-trips float32
    for nt in (2, 5, 16):
        same(host.parse_scene_native(text=txt, flavour="hw4", nthreads=nt), one)
    p = tmp_path / "mesh.txt"
    p.write_text(txt)
    same(host.parse_scene_native(path=str(p), flavour="hw4", nthreads=7), one)
    same(one, scene.parse_scene_text(txt, "hw4"))
    # material state and an open object crossing chunk boundaries: few, far-apart M / O lines
    lines = txt.splitlines()
    lines.insert(len(lines) // 3, "M 0.2 0.3 0.4 1 1 1 6 0 0 1")
    lines.insert(2 * len(lines) // 3, "O")
    t3 = "\n".join(lines)
    ref = scene.parse_scene_text(t3, "hw4")
    for nt in (1, 3, 8):
        same(host.parse_scene_native(text=t3, flavour="hw4", nthreads=nt), ref)


def test_arguments_that_look_like_keywords_do_not_split_statements():
    """the reference reads a fixed number of tokens after a keyword whatever they contain (hw4.cpp:538-747): `nan`, `inf`,
    junk that atof reads as 0, even a `T` in argument position.  A parallel chunk cut that lands on such a token must not
    end the statement early: every thread count gives the single-threaded parse (and the Python mirror's)."""
    sd = synth.mesh_scene(120, 60, seed=4)
    lines = scene.write_scene_text(sd).splitlines()
    junk = ["nan", "inf", "Tx", "T", "S", "abc", "M", "-inf"]
    out = []
    k = 0
    for ln in lines:
        tk = ln.split()
        if tk and tk[0] == "T" and len(tk) == 13:
            tk[1 + (k % 12)] = junk[k % len(junk)]              # one argument of every triangle: a letter-led token
            k += 1
            ln = " ".join(tk)
        out.append(ln)
    txt = "\n".join(out)
    assert len(txt) > (1 << 20) and k > 10_000
    one = host.parse_scene_native(text=txt, flavour="hw4", nthreads=1)
    ref = scene.parse_scene_text(txt, "hw4")
    assert one.nprim == sd.nprim == ref.nprim
    # NaNs may carry different payload bits (strtod / atof of "nan" against Python's float("nan")); everything else, and the
    # places of the NaNs, must be equal
    assert np.array_equal(one.prims, ref.prims, equal_nan=True)
    for nt in (2, 3, 8, 16):
        got = host.parse_scene_native(text=txt, flavour="hw4", nthreads=nt)
        assert got.nprim == one.nprim and got.prims.tobytes() == one.prims.tobytes()
        assert list(got.obj_count) == list(one.obj_count) and got.prim_material.tobytes() == one.prim_material.tobytes()


def test_binary_cache_round_trip(tmp_path):
    p = os.path.join(GOLD, "monkey.txt")
    c = str(tmp_path / "monkey.rtscene")
    a = host.parse_scene_native(path=p, flavour="hw4", cache_save=c)
    same(host.parse_scene_native(cache_load=c, flavour="hw4"), a)
    blob = open(c, "rb").read()
    open(c, "wb").write(blob[:-7])                                # truncated: refused, not read out of bounds
    with pytest.raises(ValueError):
        host.parse_scene_native(cache_load=c, flavour="hw4")
    open(c, "wb").write(b"nonsense" + blob[8:])
    with pytest.raises(ValueError):
        host.parse_scene_native(cache_load=c, flavour="hw4")
    with pytest.raises(ValueError):
        host.parse_scene_native(path=str(tmp_path / "missing.txt"), flavour="hw4")


def test_programs_exist_and_have_no_cpu_renderer(tmp_path):
    import torch
    for name in ("hw4", "photon"):
        exe = os.path.join(ROOT, "raytracer_amd", "bin", name)
        assert os.access(exe, os.X_OK), f"{exe} is missing: make -C raytracer_amd/host"
        r = subprocess.run([exe, str(tmp_path / "nope.txt")], cwd=tmp_path, capture_output=True, text=True, timeout=60)
        assert r.returncode != 0 and "can not open input file" in r.stdout        # hw4.cpp:511
    if not torch.cuda.is_available():
        r = subprocess.run([os.path.join(ROOT, "raytracer_amd", "bin", "hw4"), os.path.join(GOLD, "monkey.txt")],
                           cwd=tmp_path, capture_output=True, text=True, timeout=120)
        assert r.returncode == 2 and not (tmp_path / "colorOutput.ppm").exists(), (r.returncode, r.stderr)
