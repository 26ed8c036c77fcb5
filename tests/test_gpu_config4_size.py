"""BASELINE configuration 4 AT ITS STATED SIZE: 10 M global + 1 M caustic photons on the bundled Final scene,
made the way the product makes them -- rt_photonmap_emit_dev until the target is stored, rt_photonmap_build_dev
(PhotonMap::balance + ScalePhotonPower, photon.cpp:133-254, 826-833, replayed by kernels), the photons never on the host --
then checked against the oracle on the DOWNLOADED heaps:

  * rt_photon_knn (LocatePhoton, photon.cpp:53-131) on 2 000 surface hit points: np.index lists, dist2 and found bit for
    bit, on both maps at the radii the renderer uses (final.cpp:909: 0.1 caustic; photon.cpp:780: 1.0 global);
  * rt_photon_irradiance (CalcIrradiance, photon.cpp:391-441; the default wave-shared estimate) within 1e-5 relative
    (it adds the same photons in another order) and the strict option bit for bit;
  * a 64x36 crop of the 1920x1080 photon-mapping frame (final.cpp:1045-1209; 1 tap, 50 gather rays, 20x20 light samples,
    k = 100) against the oracle's frame on the same maps and the same crop camera: hit ids equal, radiance within
    2e-4 * max(1, |L|) on >= 99.5 % of the pixels, the 8-bit frame within 1 LSB (tests/test_gpu_photon_render.py's bar).
The fixture-size tests (12 k - 700 k photons) are tests/test_gpu_photon.py / test_gpu_photon_render.py; this file is the
same bar at the size the configuration names."""
import ctypes as C
import os
import time

import numpy as np
import pytest

import oracle_lib as ol
from raytracer_amd import abi, host, scene

pytestmark = pytest.mark.gpu

N_GLOBAL, N_CAUSTIC = 10_000_000, 1_000_000


@pytest.fixture(scope="module")
def config4():
    import torch
    from raytracer_amd import dist
    sd = scene.parse_scene_file(os.path.join(ol.GOLDEN, "final_input.txt"), flavour="final")
    g = host.Scene(sd); o = ol.OracleScene(sd)
    t0 = time.perf_counter()
    maps = dist.build_photon_maps(g, torch, (N_GLOBAL, N_CAUSTIC), seed=1)
    print(f"config 4 maps: {N_GLOBAL} + {N_CAUSTIC} photons emitted, balanced, imaged and downloaded in {time.perf_counter() - t0:.1f} s "
          f"({maps[0]['emissions']} + {maps[1]['emissions']} emissions)")
    heaps = [np.ascontiguousarray(maps[0]["heap"]), np.ascontiguousarray(maps[1]["heap"])]
    assert heaps[0].size == N_GLOBAL + 1 and heaps[1].size == N_CAUSTIC + 1
    yield sd, g, o, heaps
    g.close(); o.close()


def _surface_queries(sd, g, n, seed):
    """hit points of rays shot from inside the box in random directions (what the renderer queries at: surface points),
    with a unit vector facing back along the ray as the query normal"""
    rng = np.random.default_rng(seed)
    ori = np.tile(np.float32([0.0, 0.0, 0.5]), (2 * n, 1)) + rng.uniform(-0.2, 0.2, (2 * n, 3)).astype(np.float32)
    d = rng.normal(size=(2 * n, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    h = g.trace_closest(np.concatenate([ori, d], axis=1))
    ok = h["prim"] >= 0
    q = np.ascontiguousarray(h["p"][ok][:n]); nr = np.ascontiguousarray(-d[ok][:n])
    assert q.shape[0] == n
    return q, nr


@pytest.mark.parametrize("which,radius", [(abi.RT_PMAP_GLOBAL, 1.0), (abi.RT_PMAP_CAUSTIC, 0.1)])
def test_config4_knn_lists_and_irradiance_vs_oracle(config4, which, radius):
    sd, g, o, heaps = config4
    heap = heaps[which]; n = heap.size - 1
    k, nq = 100, 2000
    q, nr = _surface_queries(sd, g, nq, seed=11 + which)
    if which == abi.RT_PMAP_CAUSTIC:        # the caustic map lives on the floor under the glass sphere: query there
        rng = np.random.default_rng(5)
        q[: nq // 2] = heap["position"][1:][rng.integers(0, n, nq // 2)] + rng.normal(scale=0.01, size=(nq // 2, 3)).astype(np.float32)
        nr[: nq // 2] = np.float32([0, 1, 0])
    L = ol.oracle()
    oi = np.zeros((nq, k), np.int32); od = np.zeros((nq, k), np.float32); of = np.zeros(nq, np.int32)
    orr = np.zeros(nq, np.float32); vis = C.c_int64(0)
    L.orc_pmap_knn(heap.ctypes.data, n, q.ctypes.data, nq, k, radius, oi.ctypes.data, od.ctypes.data, of.ctypes.data,
                   orr.ctypes.data, C.byref(vis))
    idx, d2, found, r2 = g.photon_knn(which, q, k, radius)
    assert np.array_equal(found, of) and np.array_equal(idx, oi), "np.index lists differ from the oracle's LocatePhoton"
    assert d2.tobytes() == od.tobytes() and r2.tobytes() == orr.tobytes()
    assert (found == k).sum() > nq // 4, "the queries do not exercise full lists"
    want = np.zeros((nq, 3), np.float32)
    L.orc_pmap_irradiance(heap.ctypes.data, n, q.ctypes.data, nr.ctypes.data, nq, k, radius, want.ctypes.data)
    got = g.photon_irradiance(which, q, nr, k, radius)
    assert np.allclose(got, want, rtol=1e-5, atol=1e-12), float((np.abs(got - want) / np.maximum(np.abs(want), 1e-30)).max())
    assert (want > 0).any()
    g.set_knn("strict")
    try:
        strict = g.photon_irradiance(which, q, nr, k, radius)
    finally:
        g.set_knn("fast")
    assert strict.tobytes() == want.tobytes(), "strict estimate is not the oracle's CalcIrradiance bit for bit"
    print(f"map {which}: {vis.value / nq:.0f} oracle visits per lookup, mean found {found.mean():.1f}")


def crop_camera(cam, x0, y0, w, h):
    """the camera of the w x h window at pixel (x0, y0) of `cam`'s frame: same eye, same pixel pitch (GetPixelCenter,
    hw4.cpp:268-276, derives the pitch from the corners and the resolution), corners moved to the window's"""
    c = [np.array(cam.corners[i][:], np.float32) for i in range(4)]
    X, Y = np.array(cam.xaxis[:], np.float32), np.array(cam.yaxis[:], np.float32)
    pw = np.float32(np.linalg.norm(c[1] - c[0]) / cam.resx); ph = np.float32(np.linalg.norm(c[0] - c[2]) / cam.resy)
    tl = c[0] + X * (pw * x0) + Y * (ph * y0)
    out = abi.rt_camera.from_buffer_copy(bytes(cam))
    for i, v in enumerate((tl, tl + X * (pw * w), tl + Y * (ph * h), tl + X * (pw * w) + Y * (ph * h))):
        for k2 in range(3):
            out.corners[i][k2] = float(np.float32(v[k2]))
    out.resx, out.resy = w, h
    return out


def test_config4_frame_crop_vs_oracle(config4):
    sd, g, o, heaps = config4
    full = host.camera_for(sd, (1920, 1080))
    # a window that holds the glass sphere's edge (prim 1, refract 1), the floor under it with the caustic (prim 3) and a
    # piece of the right wall (prim 10)
    cam = crop_camera(full, 1560, 900, 64, 36)
    p = host.default_params("final", aa_taps=1)
    rgb, ids, st = g.render(cam, p, want_ids=True)
    rad, _ = g.render_radiance(cam, p)
    t0 = time.perf_counter()
    orgb, oids, orad = o.render_photon(cam, p, heaps[0], heaps[1])
    print(f"oracle: 64x36 crop in {time.perf_counter() - t0:.1f} s; GPU frame {st.kernel_ms:.1f} ms, {st.knn_lookups} lookups")
    assert np.array_equal(ids, oids), "hit ids of the crop differ from the oracle's"
    assert len(np.unique(ids)) >= 3, "the crop window is too plain to be a test"
    npx = 64 * 36
    tol = 2e-4 * np.maximum(1.0, np.abs(orad))
    bad = (~(np.abs(rad - orad) <= tol) & ~(np.isnan(rad) & np.isnan(orad))).any(-1)
    assert bad.sum() <= 0.005 * npx, (int(bad.sum()), npx, float(np.nanmax(np.abs(rad - orad))))
    d = np.abs(rgb.astype(int) - np.asarray(orgb).reshape(rgb.shape).astype(int)).max(-1)
    assert (d > 1).sum() <= 0.005 * npx and (d > 0).sum() <= 0.05 * npx, (int((d > 1).sum()), int((d > 0).sum()))
    assert st.knn_lookups > 50 * npx // 2
