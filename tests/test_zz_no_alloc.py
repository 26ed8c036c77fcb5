"""GPU: rt_photon_irradiance_noalloc_dev (include/rt_hip.h) -- the radiance estimate for callers that cannot allocate
device memory.  It runs the kernels of rt_photon_irradiance_dev with the per-call workspace in a fixed array of the
library's code object, so its output must equal that entry point's bit for bit, and the oracle's CalcIrradiance
(Final/photon/photon.cpp:391-441 restated) within the float-association tolerance of the fast estimate (1e-5 relative;
bit-exact with the strict option).  Two calls in a row share the workspace, the second with a smaller input: whatever the
first call left there (sort keys, the strict kernel's list and its length) must not leak into the second.
No counterpart in the reference."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
from raytracer_amd import abi, host, scene
import os

pytestmark = pytest.mark.gpu


def _lattice_map(rng):
    """200 k photons on the walls of the Final scene's box plus a lattice with exact duplicates: the lattice makes exact
    ties, which the fast kernel hands to the strict one through the list that lives in the workspace"""
    n1 = 12
    ax = (np.arange(n1, dtype=np.float32) / np.float32(n1) - np.float32(0.5))
    pts = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    pts = np.concatenate([pts, pts[::3]])
    m = 200_000
    wall = rng.uniform(-0.5, 0.5, (m, 3)).astype(np.float32)
    face = rng.integers(0, 3, m)
    wall[face == 0, 1] = -0.5; wall[face == 1, 0] = -0.5; wall[face == 2, 2] = 0.5
    pos = np.concatenate([pts, wall]).astype(np.float32)
    n = pos.shape[0]
    a = np.zeros(n + 1, abi.photon_dtype)
    a["position"][1:] = pos
    a["power"][1:] = rng.uniform(0, 1e-4, (n, 3)).astype(np.float32)
    a["phi"][1:] = rng.integers(0, 256, n); a["theta"][1:] = rng.integers(0, 256, n)
    return a, pts


def _queries(rng, pts, q):
    qp = rng.uniform(-0.5, 0.5, (q, 3)).astype(np.float32)
    qp[: q // 3, 1] = -0.5                                    # on a wall
    nl = min(q // 5, 400)
    qp[q - nl:] = pts[rng.choice(pts.shape[0], nl)]           # ON lattice points: symmetric ties -> strict redo
    qn = rng.normal(size=(q, 3)).astype(np.float32)
    return np.ascontiguousarray(qp), np.ascontiguousarray(qn)


def _oracle_irr(heap, qp, qn, k, maxd):
    rgb = np.zeros((qp.shape[0], 3), np.float32)
    ol.oracle().orc_pmap_irradiance(heap.ctypes.data, heap.size - 1, qp.ctypes.data, qn.ctypes.data, qp.shape[0], k, maxd,
                                    rgb.ctypes.data)
    return rgb


@pytest.mark.parametrize("mode", ["fast", "strict"])
def test_noalloc_irradiance_equals_the_allocating_entry_point(mode):
    import torch
    L = abi.load()
    abi.check(L.rt_init(1))                                   # resolves the workspace's address on the device
    sd = scene.parse_scene_file(os.path.join(ol.GOLDEN, "final_input.txt"), flavour="final")
    s = host.Scene(sd, device=0)
    s.set_knn(mode)
    rng = np.random.default_rng(77)
    a, pts = _lattice_map(rng)
    heap = s.photonmap_build(abi.RT_PMAP_GLOBAL, a)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    k, maxd = 100, 0.25
    # two calls in a row on the shared workspace: 60 000 queries, then 4 097 (smaller, ragged last wave)
    for q in (60_000, 4_097):
        qp, qn = _queries(rng, pts, q)
        dp = torch.from_numpy(qp).cuda(); dn = torch.from_numpy(qn).cuda()
        want = torch.full((q, 3), -1.0, dtype=torch.float32, device="cuda")
        got = torch.full((q, 3), -2.0, dtype=torch.float32, device="cuda")
        abi.check(L.rt_photon_irradiance_noalloc_dev(s.h, 0, dp.data_ptr(), dn.data_ptr(), q, k, maxd, got.data_ptr(), st))
        abi.check(L.rt_photon_irradiance_dev(s.h, 0, dp.data_ptr(), dn.data_ptr(), q, k, maxd, want.data_ptr(), st))
        torch.cuda.synchronize()
        g, w = got.cpu().numpy(), want.cpu().numpy()
        assert g.tobytes() == w.tobytes(), (mode, q, float(np.abs(g - w).max()))
        # and against the oracle on a sample that includes the tie queries at the end
        sel = np.concatenate([np.arange(0, q - 400, max(1, (q - 400) // 600)), np.arange(q - 400, q)])
        o = _oracle_irr(heap, np.ascontiguousarray(qp[sel]), np.ascontiguousarray(qn[sel]), k, maxd)
        if mode == "strict":
            assert g[sel].tobytes() == o.tobytes()
        else:
            assert np.allclose(g[sel], o, rtol=1e-5, atol=1e-12), float(np.abs(g[sel] - o).max())
        assert np.abs(o).max() > 0
    s.close()


def test_noalloc_limits_and_errors():
    import torch
    L = abi.load()
    abi.check(L.rt_init(1))
    sd = scene.parse_scene_file(os.path.join(ol.GOLDEN, "final_input.txt"), flavour="final")
    s = host.Scene(sd, device=0)
    buf = torch.zeros(16, dtype=torch.float32, device="cuda")
    rc = L.rt_photon_irradiance_noalloc_dev(s.h, 0, buf.data_ptr(), buf.data_ptr(), 4, 100, 1.0, buf.data_ptr(), None)
    assert rc == abi.RT_E_STATE                               # no map uploaded
    a = np.zeros(1001, abi.photon_dtype)
    a["position"][1:] = np.random.default_rng(1).uniform(-0.5, 0.5, (1000, 3)).astype(np.float32)
    a["power"][1:] = np.float32(1e-3)
    a["phi"][1:] = np.arange(1000) % 256; a["theta"][1:] = (np.arange(1000) * 7) % 256
    s.photonmap_build(abi.RT_PMAP_GLOBAL, a)
    rc = L.rt_photon_irradiance_noalloc_dev(s.h, 0, buf.data_ptr(), buf.data_ptr(), abi.RT_NOALLOC_MAX_QUERIES + 1, 100, 1.0,
                                            buf.data_ptr(), None)
    assert rc == abi.RT_E_LIMIT and b"split the call" in L.rt_last_error()
    assert L.rt_photon_irradiance_noalloc_dev(s.h, 0, None, None, 0, 100, 1.0, None, None) == 0     # empty input
    assert L.rt_photon_irradiance_noalloc_dev(s.h, 0, None, buf.data_ptr(), 4, 100, 1.0, buf.data_ptr(), None) == abi.RT_E_INVALID
    # the largest call the workspace is stated for goes through (and the answer is the allocating entry point's)
    q = abi.RT_NOALLOC_MAX_QUERIES
    qp = torch.rand((q, 3), dtype=torch.float32, device="cuda") - 0.5
    qn = torch.ones((q, 3), dtype=torch.float32, device="cuda")
    g = torch.zeros((q, 3), dtype=torch.float32, device="cuda"); w = torch.zeros((q, 3), dtype=torch.float32, device="cuda")
    abi.check(L.rt_photon_irradiance_noalloc_dev(s.h, 0, qp.data_ptr(), qn.data_ptr(), q, 50, 0.3, g.data_ptr(), None))
    abi.check(L.rt_photon_irradiance_dev(s.h, 0, qp.data_ptr(), qn.data_ptr(), q, 50, 0.3, w.data_ptr(), None))
    torch.cuda.synchronize()
    assert torch.equal(g, w) and float(g.abs().max()) > 0.0
    s.close()


def test_noalloc_call_allocates_no_device_memory():
    """The property the entry point exists for: on a FRESH scene (no grow-only workspace yet) the largest call it accepts
    leaves the device's free memory (hipMemGetInfo) exactly where it was, while the allocating entry point on the same
    fresh scene lowers it -- so the check can tell the two apart.  Inputs and outputs are allocated before the first
    reading; nothing else runs on the device in between."""
    import torch
    L = abi.load()
    abi.check(L.rt_init(1))
    sd = scene.parse_scene_file(os.path.join(ol.GOLDEN, "final_input.txt"), flavour="final")
    rng = np.random.default_rng(5)
    a = np.zeros(150_001, abi.photon_dtype)
    a["position"][1:] = rng.uniform(-0.5, 0.5, (150_000, 3)).astype(np.float32)
    a["power"][1:] = np.float32(1e-4)
    a["phi"][1:] = rng.integers(0, 256, 150_000); a["theta"][1:] = rng.integers(0, 256, 150_000)
    q = abi.RT_NOALLOC_MAX_QUERIES
    qp = torch.rand((q, 3), dtype=torch.float32, device="cuda") - 0.5
    qn = torch.ones((q, 3), dtype=torch.float32, device="cuda")
    out = torch.zeros((q, 3), dtype=torch.float32, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    free = {}
    for name in ("noalloc", "allocating"):
        s = host.Scene(sd, device=0)
        s.photonmap_build(abi.RT_PMAP_GLOBAL, a.copy())
        torch.cuda.synchronize()
        before = torch.cuda.mem_get_info()[0]
        fn = L.rt_photon_irradiance_noalloc_dev if name == "noalloc" else L.rt_photon_irradiance_dev
        abi.check(fn(s.h, 0, qp.data_ptr(), qn.data_ptr(), q, 100, 0.2, out.data_ptr(), st))
        torch.cuda.synchronize()
        free[name] = (before, torch.cuda.mem_get_info()[0])
        assert float(out.abs().max()) > 0.0
        s.close()
    assert free["noalloc"][1] == free["noalloc"][0], f"free device memory moved by {free['noalloc'][0] - free['noalloc'][1]} B"
    assert free["allocating"][1] < free["allocating"][0], "the check has no teeth: the allocating entry point did not allocate"
