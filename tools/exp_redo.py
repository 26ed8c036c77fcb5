#!/usr/bin/env python3
"""the strict redo's walks against each other: EVERY query is sent to the strict kernel (RT_KNN_BUDGET=1: every wave of the
fast kernel gives up) and the irradiance of RT_KNN_REDO_WALK = 2 (candidates by the second heap, the reference's order by
keys), 1 (stack walk) and 0 (stackless walk) must be the same bits; maps: the kNN line's (photons on the walls of a box)
and a coplanar one (the caustic map's shape: every photon on the floor); radii 1.0 and 0.1; also times one call each.
    python3 tools/exp_redo.py [photons=1e6] [queries=60000]"""
import ctypes as C, os, sys, time
os.environ["RT_KNN_BUDGET"] = "1"; os.environ["RT_KNN_WALK"] = "wave"     # the wave-shared walk has the budget
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from raytracer_amd import abi, host
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
nq = int(float(sys.argv[2])) if len(sys.argv) > 2 else 60_000
g = host.Scene(bench.bench_scene())
bad = 0
for shape in ("walls", "coplanar"):
    a, rng = bench.knn_map(n)
    if shape == "coplanar":
        a["position"][1:, 1] = np.float32(-0.5)
        a["position"][1:, 0] = (a["position"][1:, 0] * np.float32(0.3)).astype(np.float32)      # a dense patch
        a["position"][1:, 2] = (np.float32(0.5) + (a["position"][1:, 2] - np.float32(0.5)) * np.float32(0.3)).astype(np.float32)
    g.photonmap_build(abi.RT_PMAP_GLOBAL, a)
    q = rng.uniform(-0.5, 0.5, (nq, 3)).astype(np.float32); q[:, 2] += 0.5
    if shape == "coplanar":
        q[:, 1] = -0.5; q[:, 0] *= 0.35; q[:, 2] = 0.5 + (q[:, 2] - 0.5) * 0.35
        q[: nq // 8] = a["position"][1 + rng.integers(0, n, nq // 8)]          # queries ON photons: exact ties of symmetric neighbours, dist2 = 0
    else:
        qf = rng.integers(0, 3, nq); q[qf == 0, 1] = -0.5; q[qf == 1, 0] = -0.5; q[qf == 2, 2] = 1.0
    nrm = np.tile(np.float32([0, 1, 0]), (nq, 1))
    dq = torch.from_numpy(np.ascontiguousarray(q)).cuda(); dn = torch.from_numpy(nrm).cuda()
    for k, radius in ((100, 1.0), (100, 0.1), (30, 0.02), (112, 0.05)):
        res = {}
        for walk in ("2", "2/0", "2/37", "2/400", "1", "0"):
            if walk == "0" and (shape == "coplanar" and radius >= 0.1):
                continue                                                   # (the stackless walk of a coplanar map: minutes)
            os.environ["RT_KNN_REDO_WALK"] = walk[0]
            os.environ.pop("RT_KNN_REDO_SWITCH", None); os.environ.pop("RT_KNN_REDO_GIVEUP", None)
            if "/" in walk:      # every walk is given up as soon as the heap exists; the second launch switches to keys after that many stack items
                os.environ["RT_KNN_REDO_GIVEUP"] = "0"; os.environ["RT_KNN_REDO_SWITCH"] = walk.split("/")[1]
            out = torch.zeros((nq, 3), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize(); t0 = time.perf_counter()
            abi.check(g.L.rt_photon_irradiance_dev(g.h, 0, dq.data_ptr(), dn.data_ptr(), nq, k, radius, out.data_ptr(), None))
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            res[walk] = (out.cpu().numpy(), dt)
        ref = res["1"][0]
        line = f"{shape} k={k} r={radius}: " + " ".join(f"walk {w} {res[w][1] * 1e3:.1f} ms" for w in res)
        for w in res:
            same = np.array_equal(res[w][0].view(np.uint32), ref.view(np.uint32))
            if not same:
                bad += 1
                d = (res[w][0].view(np.uint32) != ref.view(np.uint32)).any(-1)
                line += f" | walk {w} DIFFERS in {int(d.sum())} queries (first {int(np.nonzero(d)[0][0])})"
        print(line + f" | nonzero {int((ref != 0).any(-1).sum())}", flush=True)
print("mismatching runs:", bad)
sys.exit(1 if bad else 0)
