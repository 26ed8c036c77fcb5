"""CPU tests (no GPU): the oracle (oracle/rt_oracle.c) against fixtures produced by the
reference itself (tests/golden/*.npz, made by oracle/gen_golden.py from oracle/_ref).
This is what pins the oracle; the GPU parity tests then compare the HIP path to it."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import oracle_lib as ol
from raytracer_amd import abi, host

G = ol.GOLDEN


def _load(name):
    """dict of live arrays (an NpzFile would hand out temporaries whose .ctypes.data dangles)"""
    with np.load(os.path.join(G, name)) as z:
        return {k: np.ascontiguousarray(z[k]) for k in z.files}


@pytest.fixture(scope="module")
def osc(monkey_sd):
    s = ol.OracleScene(monkey_sd)
    yield s
    s.close()


def test_unit_box_tri_sphere_known_answers():
    g = _load("unit_vectors.npz")
    L = ol.oracle()
    n = g["box_hit"].size
    box = np.array([L.orc_box_test(g["box_c"][i].ctypes.data, g["box_e"][i].ctypes.data, g["box_o"][i].ctypes.data,
                                   g["box_d"][i].ctypes.data) for i in range(n)], np.uint8)
    assert np.array_equal(box, g["box_hit"])
    p = np.zeros(3, np.float32)
    for i in range(n):
        h = L.orc_tri_test(g["tri_v"][i].ctypes.data, g["tri_o"][i].ctypes.data, g["tri_d"][i].ctypes.data, p.ctypes.data)
        assert h == g["tri_hit"][i], i
        if h:
            assert p.tobytes() == g["tri_p"][i].tobytes(), i
        h = L.orc_sphere_test(g["sph_c"][i].ctypes.data, float(g["sph_r"][i]), g["sph_o"][i].ctypes.data,
                              g["sph_d"][i].ctypes.data, p.ctypes.data)
        assert h == g["sph_hit"][i], i
        if h:
            assert p.tobytes() == g["sph_p"][i].tobytes(), i


def test_kd_tree_matches_reference(osc):
    g = _load("monkey_kd.npz")
    i = osc.info()
    assert (i.kd_nodes_built, i.kd_max_depth) == (g["nbuilt"].item(), g["maxdepth"].item()) == (138, 9)
    lohi, leaf, depth, skip, lists = ol.canonical_from_tree(osc.kd_tree())
    assert np.array_equal(lohi, g["lohi"]) and np.array_equal(leaf, g["leaf"])
    assert np.array_equal(depth, g["depth"]) and np.array_equal(skip, g["skip"])
    assert np.array_equal(np.concatenate([l for l in lists if l is not None]), g["leaf_prims"])


def test_camera_and_primary_rays(monkey_sd):
    g = _load("monkey256_hits.npz")
    cam = ol.oracle_camera(monkey_sd.eye, monkey_sd.view, monkey_sd.fov, 256, 256)
    assert np.array_equal(ol.camera_to_array(cam), g["cam"])
    xs, ys = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    sel = np.random.default_rng(0).choice(65536, 4000, replace=False)
    rays = ol.oracle_primary_rays(cam, xs.ravel()[sel], ys.ravel()[sel])
    assert rays.tobytes() == g["rays"][sel].tobytes()


def test_hit_ids_len_point_256(osc):
    g = _load("monkey256_hits.npz")
    h = osc.trace_closest(g["rays"])
    assert np.array_equal(h["prim"], g["prim"])
    hit = g["prim"] >= 0
    assert h["len"][hit].tobytes() == g["len"][hit].tobytes()
    assert h["p"][hit].tobytes() == g["point"][hit].tobytes()


def test_hit_ids_random_and_degenerate_rays(osc):
    g = _load("monkey_rays_hits.npz")
    h = osc.trace_closest(g["rays"])
    assert np.array_equal(h["prim"], g["prim"])
    hit = g["prim"] >= 0
    assert h["len"][hit].tobytes() == g["len"][hit].tobytes()
    assert h["p"][hit].tobytes() == g["point"][hit].tobytes()


def test_is_in_shadow(osc):
    g = _load("monkey_shadow.npz")
    rays, _ = ol.oracle_shadow_rays(g["point"], g["light"], g["normal"])
    assert np.array_equal(osc.trace_any(rays), g["shadow"])


def test_full_frame_byte_exact_and_stats(osc, monkey_sd):
    g = _load("monkey256_ref.npz")
    osc.reset()
    cam = ol.oracle_camera(monkey_sd.eye, monkey_sd.view, monkey_sd.fov, 256, 256)
    rgb, ids, _ = osc.render_whitted(cam, host.default_params("hw4"))
    assert np.array_equal(rgb, g["rgb"])                                # colorOutput.ppm of the reference binary
    c = osc.counters()
    assert f"{c.average_check:.6g}" == f"{g['avg_check'].item():.6g}" == "133.332"   # executionTime_monkey.txt:6
    # census of SURVEY.md 3.1 / BASELINE.md
    assert (c.rays_primary, c.rays_reflect, c.rays_shadow) == (65536, 9827, 28805)
    assert (c.box_tests, c.tri_tests) == (4298664, 16337755)
    h = _load("monkey256_hits.npz")
    assert np.array_equal(ids.reshape(256, 256).T.ravel(), h["prim"])   # ids are [y][x]; fixture is x-major


def test_bundled_2018_image_differs_in_six_pixels():
    """HW4/hw4/monkey256.ppm (rendered 2018 with MinGW/x87) vs the reference compiled here: SURVEY section 4."""
    g = _load("monkey256_ref.npz")
    old = g.get("bundled_rgb")
    if old is None:
        pytest.skip("bundled image not in the fixture")
    d = np.argwhere((old != g["rgb"]).any(-1))
    assert len(d) == 6


def test_photon_balance_and_quantisation():
    g = _load("photon_golden.npz")
    L = ol.oracle()
    stored = g["stored"].view(abi.photon_dtype).copy()
    heap = g["heap"].view(abi.photon_dtype)
    n = stored.size - 1
    for i in range(0, n, 37):
        phi, th = C.c_uint8(), C.c_uint8(); nd = np.zeros(3, np.float32)
        L.orc_photon_quantize_dir(g["dir"][i].ctypes.data, C.byref(phi), C.byref(th), nd.ctypes.data)
        assert (phi.value, th.value) == (stored["phi"][i + 1], stored["theta"][i + 1])
        assert nd.tobytes() == stored["inDir"][i + 1].tobytes()
    L.orc_pmap_balance(stored.ctypes.data, n)
    for f in ("position", "phi", "theta", "plane"):
        assert np.array_equal(stored[f][1:], heap[f][1:]), f
    # ScalePhotonPower(scale, 0, stored) (final.cpp:1126, photon.cpp:826-833) scales slots 0..n-1: the last
    # heap slot keeps its unscaled power (it is one of the slots LocatePhoton never reaches, photon.cpp:58,168)
    scaled = stored["power"].copy()
    scaled[:n] *= np.float32(1.0 / n)
    assert np.array_equal(scaled[1:], heap["power"][1:])
    t = [np.zeros(256, np.float32) for _ in range(4)]
    L.orc_direction_tables(*[a.ctypes.data for a in t])
    assert np.array_equal(np.stack(t), g["tables"])


@pytest.mark.parametrize("tag", ["r1", "r01", "r002"])
def test_photon_knn_and_irradiance(tag):
    g = _load("photon_golden.npz")
    L = ol.oracle()
    heap = g["heap"].view(abi.photon_dtype).copy()
    n = heap.size - 1
    qp = np.ascontiguousarray(g["qpos"]); qn = np.ascontiguousarray(g["qnrm"])
    q, k = qp.shape[0], g["k"].item(); maxd = g[f"maxd_{tag}"].item()
    idx = np.zeros((q, k), np.int32); d2 = np.zeros((q, k), np.float32)
    found = np.zeros(q, np.int32); r2 = np.zeros(q, np.float32); vis = C.c_int64(0)
    L.orc_pmap_knn(heap.ctypes.data, n, qp.ctypes.data, q, k, maxd, idx.ctypes.data, d2.ctypes.data,
                   found.ctypes.data, r2.ctypes.data, C.byref(vis))
    assert np.array_equal(idx, g[f"idx_{tag}"]) and np.array_equal(found, g[f"found_{tag}"])
    assert d2.tobytes() == g[f"d2_{tag}"].tobytes() and r2.tobytes() == g[f"r2_{tag}"].tobytes()
    rgb = np.zeros((q, 3), np.float32)
    L.orc_pmap_irradiance(heap.ctypes.data, n, qp.ctypes.data, qn.ctypes.data, q, k, maxd, rgb.ctypes.data)
    assert rgb.tobytes() == g[f"rgb_{tag}"].tobytes()


def test_4k_hit_id_digest_subsample(osc, monkey_sd):
    """BASELINE config 2 golden: the reference's primary hit ids at 3840x2160.  The full digest is
    checked on the GPU; here the oracle reproduces the 480x270 subsample."""
    g = _load("monkey4k_hits_digest.npz")
    cam = ol.oracle_camera(monkey_sd.eye, monkey_sd.view, monkey_sd.fov, 3840, 2160)
    assert np.array_equal(ol.camera_to_array(cam), g["cam"])
    xs, ys = np.meshgrid(np.arange(0, 3840, 8), np.arange(0, 2160, 8), indexing="ij")
    sel = np.random.default_rng(1).choice(xs.size, 20000, replace=False)
    rays = ol.oracle_primary_rays(cam, xs.ravel()[sel], ys.ravel()[sel])
    h = osc.trace_closest(rays)
    assert np.array_equal(h["prim"], g["sub_ids"].ravel()[sel])


@pytest.mark.ref
@pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_live_reference_random_scene():
    """Where the compiled reference is present: a random triangle soup, tree + hit ids, oracle vs reference."""
    from raytracer_amd.scene import SceneData
    rng = np.random.default_rng(5)
    n = 600
    c = rng.uniform(-0.4, 0.4, (n, 1, 3)); v = (c + rng.normal(scale=0.05, size=(n, 3, 3))).astype(np.float32)
    v[..., 2] += 0.6
    nrm = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    prims = np.concatenate([v.reshape(n, 9), nrm], axis=1).astype(np.float32)
    sd = SceneData(prims=prims, prim_type=np.zeros(n, np.uint8), prim_material=np.zeros(n, np.int32),
                   materials=np.array([[1, 1, 1, 1, 0, 0, 1, 0, 0, 1]], np.float32),
                   obj_first=np.array([0, 250], np.int64), obj_count=np.array([250, 350], np.int64),
                   eye=np.array([0, 0, -1], np.float32), view=np.array([0, 0, 1], np.float32), fov=45.0, resx=64, resy=64)
    o = ol.OracleScene(sd)
    ol.ref_build_scene(sd, id_colour=True)
    rt, nb, md = ol.ref_kd_tree()
    i = o.info()
    assert (i.kd_nodes_built, i.kd_max_depth) == (nb, md)
    a, b = ol.canonical_from_tree(rt), ol.canonical_from_tree(o.kd_tree())
    for x, y in zip(a[:4], b[:4]):
        assert np.array_equal(x, y)
    cam = ol.oracle_camera(sd.eye, sd.view, sd.fov, 64, 64)
    xs, ys = np.meshgrid(np.arange(64), np.arange(64), indexing="ij")
    rays = ol.oracle_primary_rays(cam, xs.ravel(), ys.ravel())
    prim = np.zeros(rays.shape[0], np.int32); ln = np.zeros(rays.shape[0], np.float32); pt = np.zeros((rays.shape[0], 3), np.float32)
    ol.ref_hw4().ref_trace_ids(rays.ctypes.data, rays.shape[0], prim.ctypes.data, ln.ctypes.data, pt.ctypes.data)
    h = o.trace_closest(rays)
    assert np.array_equal(h["prim"], prim) and (prim >= 0).sum() > 500
    o.close()
