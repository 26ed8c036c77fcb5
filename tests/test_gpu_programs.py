"""The drop-in programs (raytracer_amd/bin/hw4, bin/photon: the reference's executables with the frame driver replaced by
librt_hip.so, raytracer_amd/host/rt_main.cpp) run as the reference's do -- `./hw4 monkey.txt` in a working directory --
and leave the reference's output files behind."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from raytracer_amd import abi, host, scene

pytestmark = pytest.mark.gpu
ROOT = ol.ROOT
GOLD = os.path.join(ROOT, "tests", "golden")
BIN = os.path.join(ROOT, "raytracer_amd", "bin")


def read_ppm(path):
    b = open(path, "rb").read()
    head, rest = b.split(b"\n", 1)
    magic, w, h, mx = head.split()
    assert magic == b"P6" and mx == b"255"                       # imageIO.cpp:64-74
    w, h = int(w), int(h)
    assert len(rest) == 3 * w * h
    return np.frombuffer(rest, np.uint8).reshape(h, w, 3)


def test_hw4_program_writes_the_reference_outputs(tmp_path):
    r = subprocess.run([os.path.join(BIN, "hw4"), os.path.join(GOLD, "monkey.txt")], cwd=tmp_path, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("elapse time:")                   # hw4.cpp:1128
    img = read_ppm(tmp_path / "colorOutput.ppm")
    ref = np.load(os.path.join(GOLD, "monkey256_ref.npz"))["rgb"]        # the reference binary's own frame
    d = np.abs(img.astype(int) - ref.reshape(img.shape).astype(int)).max(-1)
    assert d.max() <= 1 and (d > 0).mean() <= 0.002
    log = (tmp_path / "executionTime_monkey.txt").read_text().splitlines()
    assert log[0] == "resolution:256X256" and log[1] == "SSAA ratio:1" and log[2].startswith("elapse time:")
    assert log[3] == "total kd-tree node count:138" and log[4] == "max depth of kd-tree subdivision:9"   # the reference's run
    assert log[5].startswith("average ray-mesh Intersection check count:")
    # the same frame through the library from Python: identical bytes
    sd = scene.parse_scene_file(os.path.join(GOLD, "monkey.txt"))
    g = host.Scene(sd)
    rgb, _, _ = g.render(host.camera_for(sd), host.default_params("hw4"))
    g.close()
    assert np.array_equal(rgb.reshape(img.shape), img)
    # options: resolution, SSAA box filter, output name, scene cache
    r = subprocess.run([os.path.join(BIN, "hw4"), os.path.join(GOLD, "monkey.txt"), "--res", "64", "48", "--ssaa", "2",
                        "--out", "small.ppm"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert read_ppm(tmp_path / "small.ppm").shape == (48, 64, 3)
    assert (tmp_path / "executionTime_monkey.txt").read_text().splitlines()[:2] == ["resolution:64X48", "SSAA ratio:2"]


def test_photon_program_equals_the_library_pipeline(tmp_path):
    args = ["--res", "20", "16", "--photons", "20000", "400", "--seed", "11"]
    r = subprocess.run([os.path.join(BIN, "photon"), os.path.join(GOLD, "final_input.txt")] + args, cwd=tmp_path,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    img = read_ppm(tmp_path / "colorOutput.ppm")
    sd = scene.parse_scene_file(os.path.join(GOLD, "final_input.txt"), flavour="final")
    g = host.Scene(sd)
    for which, target in ((abi.RT_PMAP_GLOBAL, 20000), (abi.RT_PMAP_CAUSTIC, 400)):
        ph, _ = g.photonmap_emit_until(host.final_emit_params(which, seed=11), target, batch=16384 if target < 16384 else target)
        stored = np.zeros(target + 1, abi.photon_dtype); stored[1:] = ph
        g.photonmap_build(which, stored, power_scale=1.0 / np.float32(target))
    rgb, _, st = g.render(host.camera_for(sd, (20, 16)), host.default_params("final", seed=11))
    g.close()
    assert st.knn_lookups > 0 and st.rays_gather > 0
    assert np.array_equal(rgb.reshape(img.shape), img)
