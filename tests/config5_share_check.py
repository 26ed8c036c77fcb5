#!/usr/bin/env python3
"""BASELINE configuration 5's per-GPU photon share, checked at full size (a result check, not a timing):

  1. rt_photonmap_build through the device builder (photon_build_dev.hip) leaves the same heap as the host builder
     (photon_build.cpp = PhotonMap::balance, photon.cpp:133-254), byte for byte, on N photons;
  2. rt_photon_knn (LocatePhoton, photon.cpp:53-131) on that map against brute force over all N photons for a sample of
     queries: the k smallest dist2 in float32, computed as the reference computes them (photon.cpp:95-100), as a sorted
     multiset -- which photon of several at the same distance is kept depends on the walk, the distances do not;
  3. the same lists against the oracle's LocatePhoton on that heap, index for index in np.index order, and
     rt_photon_irradiance_dev (the default, wave-shared estimate) against the oracle's CalcIrradiance (photon.cpp:391-441)
     within 1e-5 relative, on a larger sample (the oracle walks the heap, it does not scan it).
Not collected by pytest (8 GB per copy of the map; the box allows it, the build container does not).

    python3 tests/config5_share_check.py [N=200000000] [brute-force queries=16] [oracle queries=2000]
        -> gpurun_out/config5_share_check.json (kept as profiles/r02_config5_share_check.json)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C                        # noqa: E402
import oracle_lib as ol                   # noqa: E402
import bench                              # noqa: E402
from raytracer_amd import abi, host       # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 16
NO = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
K, RADIUS = 100, 1.0


def say(*a):
    print(time.strftime("%H:%M:%S"), *a, flush=True)


def main():
    import torch
    out = dict(photons=N, queries=NQ, k=K, radius=RADIUS)
    a, rng = bench.knn_map(N)
    say(f"{N} photons generated")
    sd = bench.bench_scene()
    s_host, s_dev = host.Scene(sd), host.Scene(sd)
    os.environ["RT_PHOTON_BUILD"] = "host"
    t0 = time.perf_counter()
    h = s_host.photonmap_build(abi.RT_PMAP_GLOBAL, a.copy())
    out["host_build_s"] = time.perf_counter() - t0
    say(f"host build {out['host_build_s']:.1f} s")
    os.environ["RT_PHOTON_BUILD"] = "device"
    t0 = time.perf_counter()
    d = s_dev.photonmap_build(abi.RT_PMAP_GLOBAL, a.copy())
    out["device_build_s"] = time.perf_counter() - t0
    del os.environ["RT_PHOTON_BUILD"]
    say(f"device build (host array in and out) {out['device_build_s']:.1f} s")
    same = True
    step = 1 << 24
    for lo in range(1, N + 1, step):                          # slot 0 is not part of the heap
        if h[lo:lo + step].tobytes() != d[lo:lo + step].tobytes():
            same = False
            break
    out["device_heap_equals_host_heap"] = bool(same)
    say("heaps equal:", same)
    s_host.close()
    del h

    # brute force on the heap-order positions (slot i of the lists is an index into this array)
    x = np.ascontiguousarray(d["position"][:, 0]); y = np.ascontiguousarray(d["position"][:, 1]); z = np.ascontiguousarray(d["position"][:, 2])
    x[0] = y[0] = z[0] = np.float32(1e18)                     # slot 0 is no photon
    q = rng.uniform(-0.5, 0.5, (NQ, 3)).astype(np.float32); q[:, 2] += 0.5
    q[: NQ // 2, 1] = -0.5                                    # half of them on the floor, among the coplanar photons
    q[NQ // 2:, 0] = -0.5
    gi, gd, gf, gr = s_dev.photon_knn(abi.RT_PMAP_GLOBAL, q, K, RADIUS)
    lists_ok, own_ok = True, True
    for j in range(NQ):
        dx = q[j, 0] - x; d2 = dx * dx
        dx = q[j, 1] - y; d2 += dx * dx
        dx = q[j, 2] - z; d2 += dx * dx
        inside = int((d2 < np.float32(RADIUS * RADIUS)).sum())
        want = np.sort(np.partition(d2, K)[:K])
        f = int(gf[j])
        got = np.sort(gd[j, :f])
        if f != min(K, inside) or got.tobytes() != want[:f].tobytes():
            lists_ok = False
        if d2[gi[j, :f]].tobytes() != gd[j, :f].tobytes():    # the listed photons are at the listed distances
            own_ok = False
        say(f"query {j}: found {f}, inside radius {inside}, lists {'ok' if lists_ok else 'DIFFER'}")
    del x, y, z
    # the oracle on the same heap: lists index for index, and the default estimate
    qo = rng.uniform(-0.5, 0.5, (NO, 3)).astype(np.float32); qo[:, 2] += 0.5
    face = rng.integers(0, 3, NO)
    qo[face == 0, 1] = -0.5; qo[face == 1, 0] = -0.5; qo[face == 2, 2] = 1.0
    no = np.zeros((NO, 3), np.float32); no[face == 0, 1] = 1.0; no[face == 1, 0] = 1.0; no[face == 2, 2] = -1.0
    idx = np.zeros((NO, K), np.int32); od2 = np.zeros((NO, K), np.float32); fnd = np.zeros(NO, np.int32); vis = C.c_int64(0)
    ol.oracle().orc_pmap_knn(d.ctypes.data, N, qo.ctypes.data, NO, K, RADIUS, idx.ctypes.data, od2.ctypes.data, fnd.ctypes.data, None, C.byref(vis))
    gi, gd, gf, gr = s_dev.photon_knn(abi.RT_PMAP_GLOBAL, qo, K, RADIUS)
    oracle_lists = bool(np.array_equal(gi, idx) and np.array_equal(gf, fnd) and gd.tobytes() == od2.tobytes())
    say(f"oracle lists equal: {oracle_lists} ({vis.value / NO:.0f} visits per lookup)")
    orgb = np.zeros((NO, 3), np.float32)
    ol.oracle().orc_pmap_irradiance(d.ctypes.data, N, qo.ctypes.data, no.ctypes.data, NO, K, RADIUS, orgb.ctypes.data)
    dq, dn = torch.from_numpy(qo).cuda(), torch.from_numpy(no).cuda()
    rgb = torch.zeros((NO, 3), dtype=torch.float32, device="cuda")
    abi.check(s_dev.L.rt_photon_irradiance_dev(s_dev.h, 0, dq.data_ptr(), dn.data_ptr(), NO, K, RADIUS, rgb.data_ptr(), None))
    torch.cuda.synchronize()
    rgb = rgb.cpu().numpy()
    rel = float((np.abs(rgb - orgb) / np.maximum(np.abs(orgb), 1e-30)).max())
    irr_ok = bool(np.allclose(rgb, orgb, rtol=1e-5, atol=1e-12))
    say(f"irradiance within 1e-5: {irr_ok} (max rel {rel:.2e})")
    out.update(knn_dist2_lists_equal_brute_force=bool(lists_ok), listed_photons_at_listed_distances=bool(own_ok),
               oracle_queries=NO, knn_index_lists_equal_oracle=oracle_lists, oracle_visits_per_lookup=vis.value / NO,
               irradiance_within_1e5_rel_of_oracle=irr_ok, irradiance_max_rel_err=rel,
               ok=bool(same and lists_ok and own_ok and oracle_lists and irr_ok))
    s_dev.close()
    path = os.path.join(ROOT, "gpurun_out", "config5_share_check.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out))
    return 0 if out["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
