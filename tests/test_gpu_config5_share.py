"""One GPU's share of BASELINE configuration 5 ("10 M-triangle synthetic mesh, 200 M photons ... on 8x MI355X"), checked at
full size on the one GPU a test box has: the scene and the photon map are REPLICATED per GPU, so every rank
holds exactly these two structures and answers its own tiles' rays and queries from them.

Photon half (200 M photons on the walls of the Final scene's box, bench.py's stress_knn_200m map):
  1. rt_photonmap_build through the device builder (photon_build_dev.hip) leaves the same heap as the host builder
     (photon_build.cpp = PhotonMap::balance, photon.cpp:133-254), byte for byte, all 200 M slots;
  2. rt_photon_knn (LocatePhoton, photon.cpp:53-131) on that map against BRUTE FORCE over all 200 M photons for 16 queries:
     the k smallest dist2 in float32, computed as the reference computes them (photon.cpp:95-100), as a sorted multiset --
     which photon of several at the same distance is kept depends on the walk, the distances do not;
  3. the same lists against the oracle's LocatePhoton on that heap, index for index in np.index order, and
     rt_photon_irradiance_dev (the default, wave-shared estimate) against the oracle's CalcIrradiance (photon.cpp:391-441)
     within 1e-5 relative, on 2 000 queries (the oracle walks the heap, it does not scan it).
Triangle half (the 10 M-triangle mesh of bench.py's stress_mesh_10m, 3840x2160):
  4. primary hit ids of a strided 39 000-pixel sample of the 4K frame, and closest hits + occlusion of 20 000 random rays,
     against the oracle (KDNode::BuildKD + the closest-hit loop, hw4.cpp:393-490, 808-871) on the same 10 M triangles.
Host memory: three copies of the 8 GB photon array (the box allows 270 GB; the build container has no GPU anyway)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle_lib as ol                   # noqa: E402
import bench                              # noqa: E402
from raytracer_amd import abi, host, synth   # noqa: E402

pytestmark = pytest.mark.gpu

N = int(float(os.environ.get("RT_TEST_CONFIG5_PHOTONS", "200e6")))
K, RADIUS = 100, 1.0


def say(*a):
    print(time.strftime("%H:%M:%S"), *a, flush=True)


@pytest.fixture(scope="module")
def photon_share():
    a, rng = bench.knn_map(N)
    say(f"{N} photons generated")
    sd = bench.bench_scene()
    s_host, s_dev = host.Scene(sd), host.Scene(sd)
    old = os.environ.get("RT_PHOTON_BUILD")
    try:
        os.environ["RT_PHOTON_BUILD"] = "host"
        t0 = time.perf_counter()
        h = s_host.photonmap_build(abi.RT_PMAP_GLOBAL, a.copy())
        say(f"host build {time.perf_counter() - t0:.1f} s")
        s_host.close()
        os.environ["RT_PHOTON_BUILD"] = "device"
        t0 = time.perf_counter()
        d = s_dev.photonmap_build(abi.RT_PMAP_GLOBAL, a)          # in place: `a` is the heap from here on
        say(f"device build (host array in and out) {time.perf_counter() - t0:.1f} s")
    finally:
        if old is None:
            os.environ.pop("RT_PHOTON_BUILD", None)
        else:
            os.environ["RT_PHOTON_BUILD"] = old
    yield s_dev, h, d, rng
    s_dev.close()


def test_config5_device_heap_equals_host_heap(photon_share):
    s_dev, h, d, rng = photon_share
    step = 1 << 24
    for lo in range(1, N + 1, step):                          # slot 0 is not part of the heap
        assert h[lo:lo + step].tobytes() == d[lo:lo + step].tobytes(), f"heaps differ in slots [{lo}, {lo + step})"


def test_config5_knn_vs_brute_force(photon_share):
    s_dev, h, d, rng = photon_share
    nq = 16
    x = np.ascontiguousarray(d["position"][:, 0]); y = np.ascontiguousarray(d["position"][:, 1]); z = np.ascontiguousarray(d["position"][:, 2])
    x[0] = y[0] = z[0] = np.float32(1e18)                     # slot 0 is no photon
    q = np.random.default_rng(99).uniform(-0.5, 0.5, (nq, 3)).astype(np.float32); q[:, 2] += 0.5
    q[: nq // 2, 1] = -0.5                                    # half of them on the floor, among the coplanar photons
    q[nq // 2:, 0] = -0.5
    gi, gd, gf, gr = s_dev.photon_knn(abi.RT_PMAP_GLOBAL, q, K, RADIUS)
    t = np.empty_like(x); d2 = np.empty_like(x)
    for j in range(nq):
        # dist2 as photon.cpp:95-100 computes it: dx*dx, += dy*dy, += dz*dz in float32, no contraction
        np.subtract(q[j, 0], x, out=t); np.multiply(t, t, out=d2)
        np.subtract(q[j, 1], y, out=t); np.multiply(t, t, out=t); np.add(d2, t, out=d2)
        np.subtract(q[j, 2], z, out=t); np.multiply(t, t, out=t); np.add(d2, t, out=d2)
        inside = int((d2 < np.float32(RADIUS * RADIUS)).sum())
        want = np.sort(np.partition(d2, K)[:K])
        f = int(gf[j])
        assert f == min(K, inside), (j, f, inside)
        assert np.sort(gd[j, :f]).tobytes() == want[:f].tobytes(), f"query {j}: the k smallest dist2 differ from brute force"
        assert d2[gi[j, :f]].tobytes() == gd[j, :f].tobytes(), f"query {j}: the listed photons are not at the listed distances"
    say(f"{nq} queries equal brute force over {N} photons")


def test_config5_lists_and_estimate_vs_oracle(photon_share):
    import torch
    s_dev, h, d, rng = photon_share
    no = 2000
    r = np.random.default_rng(123)
    qo = r.uniform(-0.5, 0.5, (no, 3)).astype(np.float32); qo[:, 2] += 0.5
    face = r.integers(0, 3, no)
    qo[face == 0, 1] = -0.5; qo[face == 1, 0] = -0.5; qo[face == 2, 2] = 1.0
    nrm = np.zeros((no, 3), np.float32); nrm[face == 0, 1] = 1.0; nrm[face == 1, 0] = 1.0; nrm[face == 2, 2] = -1.0
    idx = np.zeros((no, K), np.int32); od2 = np.zeros((no, K), np.float32); fnd = np.zeros(no, np.int32); vis = C.c_int64(0)
    ol.oracle().orc_pmap_knn(d.ctypes.data, N, qo.ctypes.data, no, K, RADIUS, idx.ctypes.data, od2.ctypes.data, fnd.ctypes.data, None, C.byref(vis))
    gi, gd, gf, gr = s_dev.photon_knn(abi.RT_PMAP_GLOBAL, qo, K, RADIUS)
    assert np.array_equal(gi, idx) and np.array_equal(gf, fnd) and gd.tobytes() == od2.tobytes()
    say(f"oracle lists equal ({vis.value / no:.0f} visits per lookup)")
    orgb = np.zeros((no, 3), np.float32)
    ol.oracle().orc_pmap_irradiance(d.ctypes.data, N, qo.ctypes.data, nrm.ctypes.data, no, K, RADIUS, orgb.ctypes.data)
    dq, dn = torch.from_numpy(qo).cuda(), torch.from_numpy(nrm).cuda()
    rgb = torch.zeros((no, 3), dtype=torch.float32, device="cuda")
    abi.check(s_dev.L.rt_photon_irradiance_dev(s_dev.h, 0, dq.data_ptr(), dn.data_ptr(), no, K, RADIUS, rgb.data_ptr(), None))
    torch.cuda.synchronize()
    rgb = rgb.cpu().numpy()
    assert np.allclose(rgb, orgb, rtol=1e-5, atol=1e-12), float((np.abs(rgb - orgb) / np.maximum(np.abs(orgb), 1e-30)).max())
    assert (orgb > 0).any()


def test_config5_mesh_10m_hit_ids_vs_oracle():
    t0 = time.perf_counter()
    sd = synth.mesh_scene(3162, 1581)
    g = host.Scene(sd)
    say(f"10 M-triangle scene ({sd.nprim} triangles) generated, built and uploaded in {time.perf_counter() - t0:.1f} s")
    t0 = time.perf_counter()
    o = ol.OracleScene(sd)
    say(f"oracle KD build {time.perf_counter() - t0:.1f} s")
    gi, oi = g.info(), o.info()
    assert (gi.kd_nodes_built, gi.kd_max_depth) == (oi.kd_nodes_built, oi.kd_max_depth)     # the same tree (hw4.cpp:391-392's statistics)
    res = (3840, 2160)
    cam = host.camera_for(sd, res)
    rgb, ids, st = g.render(cam, host.default_params("hw4", primary_only=1), want_ids=True)
    pix = np.arange(17, res[0] * res[1], 211, dtype=np.int64)              # 39 k pixels, every row and column phase
    ocam = ol.oracle_camera(sd.eye, sd.view, sd.fov, *res)
    rays = ol.oracle_primary_rays(ocam, (pix % res[0]).astype(np.int64), (pix // res[0]).astype(np.int64))
    want = o.trace_closest(rays)
    assert np.array_equal(ids[pix], want["prim"]) and (want["prim"] >= 0).sum() > pix.size // 20
    # incoherent rays through the mesh from every side, closest hit (id, len, point bits) and occlusion
    r = np.random.default_rng(8)
    n = 20000
    ori = r.normal(size=(n, 3)); ori = (ori / np.linalg.norm(ori, axis=1, keepdims=True) * 1.2).astype(np.float32); ori[:, 2] += 0.55
    tgt = np.float32([0.0, 0.0, 0.55]) + r.uniform(-0.3, 0.3, (n, 3)).astype(np.float32)
    dd = tgt - ori; dd = (dd / np.linalg.norm(dd, axis=1, keepdims=True)).astype(np.float32)
    rr = np.ascontiguousarray(np.concatenate([ori, dd], axis=1))
    got, ref = g.trace_closest(rr), o.trace_closest(rr)
    assert np.array_equal(got["prim"], ref["prim"]) and got["len"].tobytes() == ref["len"].tobytes() and got["p"].tobytes() == ref["p"].tobytes()
    assert (ref["prim"] >= 0).sum() > n // 3
    assert np.array_equal(g.trace_any(rr), o.trace_any(rr))
    g.close(); o.close()
