"""GPU parity: photon-map kNN / irradiance (HIP, LDS max-heaps) against the reference's
fixture (LocatePhoton / CalcIrradiance called through oracle/_ref) and the CPU oracle.
Bar: bit-exact index lists (in the reference's np.index order), squared distances, radius
and -- with the strict kNN option -- irradiance; with the fast option (default) the index SETS, found and
radius are bit-exact and the irradiance, which adds the same photons in another order, is within 1e-5
relative (float association only)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol
from raytracer_amd import abi, host, scene

pytestmark = pytest.mark.gpu


def _load(name):
    with np.load(os.path.join(ol.GOLDEN, name)) as z:
        return {k: np.ascontiguousarray(z[k]) for k in z.files}


@pytest.fixture(scope="module", params=["fast", "fast-wave", "fast-lane", "strict"])
def gscene(request):
    """fast = the default choice between the wave-shared and the per-lane walk (by photons per query); the two are also
    forced through RT_KNN_WALK, which the library reads at every call"""
    sd = scene.parse_scene_file(os.path.join(ol.GOLDEN, "final_input.txt"), flavour="final")
    s = host.Scene(sd)
    mode, _, walk = request.param.partition("-")
    s.set_knn(mode)
    s.knn_mode = mode
    old = os.environ.pop("RT_KNN_WALK", None)
    if walk:
        os.environ["RT_KNN_WALK"] = walk
    yield s
    os.environ.pop("RT_KNN_WALK", None)
    if old is not None:
        os.environ["RT_KNN_WALK"] = old
    s.close()


def _irr_equal(s, got, want):
    if s.knn_mode == "strict":
        return got.tobytes() == want.tobytes()
    return bool(np.allclose(got, want, rtol=1e-5, atol=1e-12))


def _sets_equal(idx, d2, found, oi, od, of):
    """the same photons with the same squared distances, whatever the order"""
    if not np.array_equal(found, of):
        return False
    a = np.sort(np.where(idx >= 0, idx, np.iinfo(np.int32).max), axis=1)
    b = np.sort(np.where(oi >= 0, oi, np.iinfo(np.int32).max), axis=1)
    if not np.array_equal(a, b):
        return False
    oa, ob = np.argsort(idx, axis=1, kind="stable"), np.argsort(oi, axis=1, kind="stable")
    return np.take_along_axis(d2, oa, 1).tobytes() == np.take_along_axis(od, ob, 1).tobytes()


def _oracle_knn(heap, qp, k, maxd):
    L = ol.oracle()
    q = qp.shape[0]; n = heap.size - 1
    idx = np.zeros((q, k), np.int32); d2 = np.zeros((q, k), np.float32)
    found = np.zeros(q, np.int32); r2 = np.zeros(q, np.float32); vis = C.c_int64(0)
    L.orc_pmap_knn(heap.ctypes.data, n, qp.ctypes.data, q, k, maxd, idx.ctypes.data, d2.ctypes.data, found.ctypes.data,
                   r2.ctypes.data, C.byref(vis))
    return idx, d2, found, r2


def _oracle_irr(heap, qp, qn, k, maxd):
    rgb = np.zeros((qp.shape[0], 3), np.float32)
    ol.oracle().orc_pmap_irradiance(heap.ctypes.data, heap.size - 1, qp.ctypes.data, qn.ctypes.data, qp.shape[0], k, maxd,
                                    rgb.ctypes.data)
    return rgb


@pytest.mark.parametrize("tag", ["r1", "r01", "r002"])
def test_knn_and_irradiance_vs_reference_fixture(gscene, tag):
    g = _load("photon_golden.npz")
    heap = g["heap"].view(abi.photon_dtype).copy()
    gscene.photonmap_upload(abi.RT_PMAP_GLOBAL, heap)
    k, maxd = g["k"].item(), g[f"maxd_{tag}"].item()
    idx, d2, found, r2 = gscene.photon_knn(abi.RT_PMAP_GLOBAL, g["qpos"], k, maxd)
    assert np.array_equal(found, g[f"found_{tag}"]) and np.array_equal(idx, g[f"idx_{tag}"])
    assert d2.tobytes() == g[f"d2_{tag}"].tobytes() and r2.tobytes() == g[f"r2_{tag}"].tobytes()
    rgb = gscene.photon_irradiance(abi.RT_PMAP_GLOBAL, g["qpos"], g["qnrm"], k, maxd)
    assert _irr_equal(gscene, rgb, g[f"rgb_{tag}"])
    si, sd2, sf, sr = gscene.photon_knn_set(abi.RT_PMAP_GLOBAL, g["qpos"], k, maxd)
    assert _sets_equal(si, sd2, sf, g[f"idx_{tag}"], g[f"d2_{tag}"], g[f"found_{tag}"]) and sr.tobytes() == g[f"r2_{tag}"].tobytes()


def test_build_from_unbalanced_equals_reference_heap(gscene):
    g = _load("photon_golden.npz")
    stored = g["stored"].view(abi.photon_dtype).copy()
    heap = g["heap"].view(abi.photon_dtype)
    out = gscene.photonmap_build(abi.RT_PMAP_CAUSTIC, stored)
    for f in ("position", "phi", "theta", "plane"):
        assert np.array_equal(out[f][1:], heap[f][1:]), f
    idx, d2, found, r2 = gscene.photon_knn(abi.RT_PMAP_CAUSTIC, g["qpos"], g["k"].item(), 0.1)
    assert np.array_equal(idx, g["idx_r01"]) and np.array_equal(found, g["found_r01"])


def _random_map(n, seed):
    rng = np.random.default_rng(seed)
    a = np.zeros(n + 1, abi.photon_dtype)
    a["position"][1:] = rng.uniform(-0.5, 0.5, (n, 3)).astype(np.float32)
    a["position"][1: 1 + n // 3, 1] = -0.5
    a["power"][1:] = rng.uniform(0, 1e-4, (n, 3)).astype(np.float32)
    a["phi"][1:] = rng.integers(0, 256, n); a["theta"][1:] = rng.integers(0, 256, n)
    return a


def test_large_map_many_k_vs_oracle(gscene):
    n = 300_000
    a = _random_map(n, 21)
    heap = gscene.photonmap_build(abi.RT_PMAP_GLOBAL, a)           # balanced in place by the product
    chk = a.copy()
    rng = np.random.default_rng(22)
    for k, maxd, q in ((1, 1.0, 700), (8, 0.05, 1000), (50, 0.2, 1025), (100, 1.0, 2000), (100, 0.01, 500), (287, 0.3, 200)):
        qp = rng.uniform(-0.55, 0.55, (q, 3)).astype(np.float32)
        qp[: q // 4, 1] = -0.5
        qn = rng.normal(size=(q, 3)).astype(np.float32)
        gi, gd, gf, gr = gscene.photon_knn(abi.RT_PMAP_GLOBAL, qp, k, maxd)
        oi, od, of, orr = _oracle_knn(heap, qp, k, maxd)
        assert np.array_equal(gf, of) and np.array_equal(gi, oi), (k, maxd)
        assert gd.tobytes() == od.tobytes() and gr.tobytes() == orr.tobytes()
        rgb = gscene.photon_irradiance(abi.RT_PMAP_GLOBAL, qp, qn, k, maxd)
        assert _irr_equal(gscene, rgb, _oracle_irr(heap, qp, qn, k, maxd)), (k, maxd)
        si, sd2, sf, sr = gscene.photon_knn_set(abi.RT_PMAP_GLOBAL, qp, k, maxd)
        oi, od, of, orr = _oracle_knn(heap, qp, k, maxd)
        assert _sets_equal(si, sd2, sf, oi, od, of) and sr.tobytes() == orr.tobytes(), (k, maxd)


def test_tiny_and_empty_maps(gscene):
    rng = np.random.default_rng(5)
    qp = rng.uniform(-0.5, 0.5, (130, 3)).astype(np.float32)
    for n in (0, 1, 2, 3, 4, 5, 9, 17):
        a = _random_map(n, 30 + n)
        heap = gscene.photonmap_build(abi.RT_PMAP_GLOBAL, a)
        gi, gd, gf, gr = gscene.photon_knn(abi.RT_PMAP_GLOBAL, qp, 4, 2.0)
        oi, od, of, orr = _oracle_knn(heap, qp, 4, 2.0)
        assert np.array_equal(gf, of) and np.array_equal(gi, oi), n
        assert gd.tobytes() == od.tobytes() and gr.tobytes() == orr.tobytes()
    # ragged / empty query sets
    for q in (0, 1, 63, 65):
        gi, gd, gf, gr = gscene.photon_knn(abi.RT_PMAP_GLOBAL, qp[:q], 4, 2.0)
        assert gf.shape == (q,)


def test_error_paths(gscene):
    sd = scene.parse_scene_file(os.path.join(ol.GOLDEN, "final_input.txt"), flavour="final")
    fresh = host.Scene(sd)
    qp = np.zeros((4, 3), np.float32)
    with pytest.raises(abi.RtError) as e:
        fresh.photon_knn(abi.RT_PMAP_GLOBAL, qp, 10, 1.0)              # no map uploaded
    assert e.value.code == -5
    with pytest.raises(abi.RtError) as e:
        gscene.photon_knn(abi.RT_PMAP_GLOBAL, qp, 100000, 1.0)         # k beyond the LDS heap limit
    assert e.value.code == -4
    fresh.close()


def test_dev_entry_points(gscene):
    import torch
    g = _load("photon_golden.npz")
    heap = g["heap"].view(abi.photon_dtype).copy()
    gscene.photonmap_upload(abi.RT_PMAP_GLOBAL, heap)
    q, k = g["qpos"].shape[0], g["k"].item()
    dp = torch.from_numpy(g["qpos"]).cuda(); dn = torch.from_numpy(g["qnrm"]).cuda()
    di = torch.zeros((q, k), dtype=torch.int32, device="cuda"); dd = torch.zeros((q, k), dtype=torch.float32, device="cuda")
    df = torch.zeros(q, dtype=torch.int32, device="cuda"); dr = torch.zeros(q, dtype=torch.float32, device="cuda")
    dc = torch.zeros((q, 3), dtype=torch.float32, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    abi.check(gscene.L.rt_photon_knn_dev(gscene.h, 0, dp.data_ptr(), q, k, 1.0, di.data_ptr(), dd.data_ptr(), df.data_ptr(),
                                         dr.data_ptr(), st))
    abi.check(gscene.L.rt_photon_irradiance_dev(gscene.h, 0, dp.data_ptr(), dn.data_ptr(), q, k, 1.0, dc.data_ptr(), st))
    torch.cuda.synchronize()
    assert np.array_equal(di.cpu().numpy(), g["idx_r1"]) and _irr_equal(gscene, dc.cpu().numpy(), g["rgb_r1"])


def test_lattice_of_ties_goes_through_the_strict_redo(gscene):
    """photons on a regular lattice (and duplicated): most queries meet candidates that tie for the farthest place, where
    the surviving set depends on the reference heap's history -- the fast kernel must hand those to the strict one."""
    n1 = 14
    ax = (np.arange(n1, dtype=np.float32) / np.float32(n1) - np.float32(0.5))
    pts = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    pts = np.concatenate([pts, pts[::3]])                              # exact duplicates too
    n = pts.shape[0]
    a = np.zeros(n + 1, abi.photon_dtype)
    a["position"][1:] = pts; a["power"][1:] = np.float32(1.0 / n)
    rng = np.random.default_rng(4)
    a["phi"][1:] = rng.integers(0, 256, n); a["theta"][1:] = rng.integers(0, 256, n)
    heap = gscene.photonmap_build(abi.RT_PMAP_GLOBAL, a)
    qp = np.concatenate([pts[rng.choice(n1 ** 3, 300)],                # queries ON lattice points: symmetric ties
                         rng.uniform(-0.5, 0.4, (300, 3)).astype(np.float32)])
    qn = rng.normal(size=qp.shape).astype(np.float32)
    for k, maxd in ((20, 0.3), (100, 0.5), (7, 2.0)):
        oi, od, of, orr = _oracle_knn(heap, qp, k, maxd)
        si, sd2, sf, sr = gscene.photon_knn_set(abi.RT_PMAP_GLOBAL, qp, k, maxd)
        assert _sets_equal(si, sd2, sf, oi, od, of) and sr.tobytes() == orr.tobytes(), (k, maxd)
        rgb = gscene.photon_irradiance(abi.RT_PMAP_GLOBAL, qp, qn, k, maxd)
        assert _irr_equal(gscene, rgb, _oracle_irr(heap, qp, qn, k, maxd)), (k, maxd)


def test_strict_redo_by_keys_is_the_references_list(monkeypatch):
    """The strict redo finishes a long walk by KEYS (kernels_photon.hip, WALK 2): the photons that can still be accepted come
    from the second heap, are sorted by the key of their turn in the reference's visiting order and go through the
    reference's `consider` in that order.  Here EVERY query is sent to the strict kernel (the wave-shared walk with a budget
    of one node gives all of them up), the first strict launch hands every walk on as soon as the reference's heap exists
    (RT_KNN_REDO_GIVEUP=0) and the second switches to keys at once, after 37 and after 400 stack items
    (RT_KNN_REDO_SWITCH): the irradiance must be the stack walk's bits (RT_KNN_REDO_WALK=1) and the oracle's CalcIrradiance
    bit for bit -- on a map with a coplanar third (the caustic map's shape, where the walk is long), a lattice with
    duplicates (ties) and queries ON photons."""
    sd = scene.parse_scene_file(os.path.join(ol.GOLDEN, "final_input.txt"), flavour="final")
    s = host.Scene(sd)
    s.set_knn("fast")
    monkeypatch.setenv("RT_KNN_WALK", "wave"); monkeypatch.setenv("RT_KNN_BUDGET", "1")
    rng = np.random.default_rng(11)
    a = _random_map(60000, 5)
    n1 = 12
    ax = (np.arange(n1, dtype=np.float32) / np.float32(n1) - np.float32(0.5))
    pts = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    lat = np.zeros(pts.shape[0] + pts[::3].shape[0], abi.photon_dtype)
    lat["position"] = np.concatenate([pts, pts[::3]]); lat["power"] = np.float32(1e-5)
    lat["phi"] = rng.integers(0, 256, lat.size); lat["theta"] = rng.integers(0, 256, lat.size)
    a = np.concatenate([a, lat])
    n = a.size - 1
    heap = s.photonmap_build(abi.RT_PMAP_GLOBAL, a)
    qp = np.concatenate([rng.uniform(-0.5, 0.5, (500, 3)).astype(np.float32),
                         a["position"][1 + rng.integers(0, n, 300)],                     # ON photons (lattice points among them)
                         np.concatenate([rng.uniform(-0.5, 0.5, (200, 1)), np.full((200, 1), -0.5), rng.uniform(-0.5, 0.5, (200, 1))], 1).astype(np.float32)])
    qn = rng.normal(size=qp.shape).astype(np.float32)
    try:
        for k, maxd in ((100, 1.0), (100, 0.1), (30, 0.05), (112, 0.3), (8, 0.02)):
            want = _oracle_irr(heap, qp, qn, k, maxd)
            got = {}
            for walk, switch in (("1", None), ("2", "0"), ("2", "37"), ("2", "400"), ("2", None)):
                monkeypatch.setenv("RT_KNN_REDO_WALK", walk)
                if switch is None:
                    monkeypatch.delenv("RT_KNN_REDO_SWITCH", raising=False); monkeypatch.delenv("RT_KNN_REDO_GIVEUP", raising=False)
                else:        # the first launch gives every walk up as soon as the heap exists, the second switches to keys after `switch` items
                    monkeypatch.setenv("RT_KNN_REDO_SWITCH", switch); monkeypatch.setenv("RT_KNN_REDO_GIVEUP", "0")
                got[(walk, switch)] = s.photon_irradiance(abi.RT_PMAP_GLOBAL, qp, qn, k, maxd)
            for key, rgb in got.items():
                assert rgb.tobytes() == want.tobytes(), (k, maxd, key, int((rgb.view(np.uint32) != want.view(np.uint32)).any(-1).sum()))
            assert (want != 0).any(-1).sum() > 300
    finally:
        s.close()


def test_coherent_queries_share_a_walk(gscene):
    """the wave-shared radiance estimate (RT_KNN_FAST: 64 neighbouring queries per walk, lists of the k smallest dist2 in
    registers) on queries that ARE neighbours -- a dense patch on a wall and a dense patch in free space --
    at every list size the kernel is instantiated for and at their edges; against the oracle's CalcIrradiance."""
    n = 400_000
    a = _random_map(n, 31)
    heap = gscene.photonmap_build(abi.RT_PMAP_GLOBAL, a)
    g = np.stack(np.meshgrid(np.linspace(-0.2, -0.1, 48), np.linspace(0.1, 0.2, 48), indexing="ij"), -1).reshape(-1, 2).astype(np.float32)
    wall = np.stack([g[:, 0], np.full(len(g), -0.5, np.float32), g[:, 1]], 1)
    free = np.stack([g[:, 0], g[:, 1], g[:, 0] * 0.5], 1).astype(np.float32)
    rng = np.random.default_rng(32)
    for k, maxd in ((8, 1.0), (32, 1.0), (33, 0.5), (64, 1.0), (65, 1.0), (100, 1.0), (104, 1.0), (105, 0.3), (112, 1.0),
                    (100, 0.02), (100, 0.004)):
        for qp in (wall, free):
            qp = np.ascontiguousarray(qp[rng.permutation(len(qp))])          # the product sorts them itself
            qn = rng.normal(size=qp.shape).astype(np.float32)
            rgb = gscene.photon_irradiance(abi.RT_PMAP_GLOBAL, qp, qn, k, maxd)
            want = _oracle_irr(heap, qp, qn, k, maxd)
            assert _irr_equal(gscene, rgb, want), (k, maxd, float(np.abs(rgb - want).max()))
    # queries far outside the map, at infinity and NaN find nothing and disturb nobody
    qp = np.concatenate([wall[:500], np.full((70, 3), 3e18, np.float32), np.full((3, 3), np.nan, np.float32), wall[500:900] + 50.0]).astype(np.float32)
    qn = rng.normal(size=qp.shape).astype(np.float32)
    rgb = gscene.photon_irradiance(abi.RT_PMAP_GLOBAL, qp, qn, 100, 1.0)
    want = _oracle_irr(heap, qp, qn, 100, 1.0)
    assert _irr_equal(gscene, rgb[:500], want[:500]) and not rgb[500:].any() and not want[500:573].any()


def test_map_image_export_import_between_scenes():
    """one build per node (raytracer_amd/dist.py): a scene that built its map exports the device image, another scene
    imports it (what a rank does with the image it received over RCCL) and answers queries with the same bits; the
    receiving scene never ran the host balance."""
    import torch
    sd = scene.parse_scene_file(os.path.join(ol.GOLDEN, "final_input.txt"), flavour="final")
    a_scene, b_scene = host.Scene(sd), host.Scene(sd)
    a = _random_map(150_000, 91)
    heap = a_scene.photonmap_build(abi.RT_PMAP_GLOBAL, a)
    L = a_scene.L
    n, nb = C.c_int64(0), C.c_int64(0)
    abi.check(L.rt_photonmap_image_bytes(a_scene.h, 0, C.byref(n), C.byref(nb)))
    assert n.value == 150_000 and nb.value > 150_000 * 64
    img = torch.empty(nb.value, dtype=torch.uint8, device="cuda")
    box = (C.c_float * 6)()
    abi.check(L.rt_photonmap_export_dev(a_scene.h, 0, img.data_ptr(), nb.value, box, None))
    torch.cuda.synchronize()
    assert L.rt_photonmap_import_dev(b_scene.h, 0, img.data_ptr(), n.value, nb.value - 16, box, None) == abi.RT_E_INVALID   # wrong size
    abi.check(L.rt_photonmap_import_dev(b_scene.h, 0, img.data_ptr(), n.value, nb.value, box, None))
    del img
    rng = np.random.default_rng(92)
    qp = rng.uniform(-0.5, 0.5, (5000, 3)).astype(np.float32); qp[:2000, 1] = -0.5
    qn = rng.normal(size=qp.shape).astype(np.float32)
    for mode in ("fast", "strict"):
        a_scene.set_knn(mode); b_scene.set_knn(mode)
        ra = a_scene.photon_irradiance(0, qp, qn, 100, 0.3)
        rb = b_scene.photon_irradiance(0, qp, qn, 100, 0.3)
        assert ra.tobytes() == rb.tobytes() and np.abs(ra).max() > 0
    ia, da, fa, _ = a_scene.photon_knn(0, qp[:500], 50, 0.3)
    ib, db, fb, _ = b_scene.photon_knn(0, qp[:500], 50, 0.3)
    oi, od, of, _ = _oracle_knn(heap, qp[:500], 50, 0.3)
    assert np.array_equal(ia, ib) and np.array_equal(ib, oi) and db.tobytes() == od.tobytes() and np.array_equal(fb, of)
    a_scene.close(); b_scene.close()


def _map_image(sc, which=0):
    import torch
    n, nb = C.c_int64(0), C.c_int64(0)
    abi.check(sc.L.rt_photonmap_image_bytes(sc.h, which, C.byref(n), C.byref(nb)))
    img = torch.empty(nb.value, dtype=torch.uint8, device="cuda")
    box = (C.c_float * 6)()
    abi.check(sc.L.rt_photonmap_export_dev(sc.h, which, img.data_ptr(), nb.value, box, None))
    torch.cuda.synchronize()
    return n.value, img.cpu().numpy(), np.array(list(box), np.float32)


def _tie_map(n, seed, kind):
    """maps that exercise median_split's handling of equal keys: 'wall' = photons on axis-aligned planes with the few distinct
    coordinates a hit point's rounding leaves, 'grid' = every coordinate from a set of 7 values, 'same' = one point, 'sorted' /
    'reversed' = distinct keys in order (the quickselect's worst pivots)."""
    rng = np.random.default_rng(seed)
    a = _random_map(n, seed)
    p = a["position"][1:]
    if kind == "wall":
        third = n // 3
        p[:third, 1] = np.float32(-0.5) + rng.integers(-1, 2, third).astype(np.float32) * np.float32(5.9604645e-08)
        p[third:2 * third, 0] = np.float32(0.5)
        p[2 * third:, 2] = rng.integers(-1, 2, n - 2 * third).astype(np.float32) * np.float32(1.4e-45)
    elif kind == "grid":
        p[:] = rng.integers(-3, 4, (n, 3)).astype(np.float32) * np.float32(0.125)
    elif kind == "same":
        p[:] = np.float32(0.25)
    elif kind == "sorted":
        p[:] = (np.arange(n, dtype=np.float32) / np.float32(n))[:, None]
    elif kind == "reversed":
        p[:] = (np.arange(n, 0, -1, dtype=np.float32) / np.float32(n))[:, None]
    return a


@pytest.mark.parametrize("n,kind", [(1, "random"), (2, "random"), (3, "grid"), (6, "random"), (7, "grid"), (100, "wall"), (1023, "wall"),
                                    (1024, "grid"), (1025, "random"), (1025, "same"), (3000, "sorted"), (5000, "wall"), (9000, "reversed"),
                                    (20_000, "grid"), (20_000, "same"), (50_000, "sorted"), (150_000, "random"), (150_000, "wall"),
                                    (700_000, "wall"), (700_000, "grid")])
def test_device_build_equals_host_build(n, kind):
    """f2: rt_photonmap_build_dev (photon_build_dev.hip: the reference's balance replayed by kernels, both heaps, boxes, node
    records) leaves the same bytes as rt_photonmap_build on the host -- the photons in heap order with their planes and
    scaled powers, the device image, the box -- on maps with distinct keys and on maps where ties decide."""
    import torch
    sd = scene.parse_scene_file(os.path.join(ol.GOLDEN, "final_input.txt"), flavour="final")
    a_scene, b_scene = host.Scene(sd), host.Scene(sd)
    src = _tie_map(n, 1000 + n, kind) if kind != "random" else _random_map(n, 1000 + n)
    scale = 0.37
    os.environ["RT_PHOTON_BUILD"] = "host"
    try:
        h_heap = a_scene.photonmap_build(0, src.copy(), scale)
    finally:
        del os.environ["RT_PHOTON_BUILD"]
    d = torch.from_numpy(src.copy().view(np.uint8)).cuda()
    b_scene.photonmap_build_dev(0, d.data_ptr(), n, scale)
    torch.cuda.synchronize()
    d_heap = d.cpu().numpy().view(abi.photon_dtype)
    assert h_heap[1:].tobytes() == d_heap[1:].tobytes()
    assert h_heap["power"][0].tobytes() == d_heap["power"][0].tobytes()           # slot 0 is scaled too (photon.cpp:826-833)
    na, ia, ba = _map_image(a_scene); nb_, ib, bb = _map_image(b_scene)
    assert na == nb_ == n and ba.tobytes() == bb.tobytes()
    assert ia.shape == ib.shape
    if not np.array_equal(ia, ib):
        bad = np.flatnonzero(ia != ib)
        raise AssertionError(f"images differ at {bad.size} bytes, first at {bad[0]} of {ia.size} (record {bad[0] // 16})")
    a_scene.close(); b_scene.close()


def test_host_pointer_build_through_the_device(monkeypatch):
    """RT_PHOTON_BUILD=device: rt_photonmap_build on a host array goes through the device builder and returns the same heap."""
    sd = scene.parse_scene_file(os.path.join(ol.GOLDEN, "final_input.txt"), flavour="final")
    a_scene, b_scene = host.Scene(sd), host.Scene(sd)
    src = _tie_map(60_000, 5, "wall")
    monkeypatch.setenv("RT_PHOTON_BUILD", "host")
    h = a_scene.photonmap_build(0, src.copy(), 1.0)
    monkeypatch.setenv("RT_PHOTON_BUILD", "device")
    dv = b_scene.photonmap_build(0, src.copy(), 1.0)
    monkeypatch.delenv("RT_PHOTON_BUILD")
    assert h[1:].tobytes() == dv[1:].tobytes()
    qp = np.random.default_rng(6).uniform(-0.5, 0.5, (2000, 3)).astype(np.float32); qp[:800, 1] = -0.5
    qn = np.random.default_rng(7).normal(size=qp.shape).astype(np.float32)
    assert a_scene.photon_irradiance(0, qp, qn, 100, 0.3).tobytes() == b_scene.photon_irradiance(0, qp, qn, 100, 0.3).tobytes()
    a_scene.close(); b_scene.close()
