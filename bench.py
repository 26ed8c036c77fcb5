#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X hot path on BASELINE.json's configuration 2:
the bundled monkey scene at 3840x2160, primary + shadow rays only (materials overridden to
reflect = refract = 0 as SURVEY 8d-2 says), traced by librt_hip.so through the C ABI.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one frame of the workload.  Everything the timed region reads (scene, KD-trees)
is resident in HBM before it starts; rays are generated on the GPU.  N > 1: the frame grows
with N (weak scaling: 3840x2160 pixels per GPU), 64x64 tiles are dealt round-robin to the
ranks (no collective in the tracing path) and every step ends with the RCCL gather of the
ranks' tiles to rank 0, as north_star prescribes.  Rank 0 prints ONE JSON line.

Extra objects on the line: `roofline` (closest-hit kernel: algorithmic bytes / HIP-event
time, vs 8 TB/s), `cpu_baseline` (the reference binary oracle/_ref/hw4_ref -- or the oracle
port when it is absent -- on a bounded sample of the same workload, 1 core), `secondary`
(photon-map kNN lookups/s on a 10 M-photon map, BASELINE configuration 4's shape).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8 TB/s (6.3 TB/s achievable)
BASE_RES = (3840, 2160)
TILE = 64
WEAK_FACTORS = {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (4, 2)}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def bench_scene():
    """monkey.txt with every material's reflect/refract set to 0."""
    from raytracer_amd import scene
    sd = scene.parse_scene_file(os.path.join(ROOT, "tests", "golden", "monkey.txt"))
    sd.materials = sd.materials.copy()
    sd.materials[:, 7] = 0.0
    sd.materials[:, 8] = 0.0
    return sd


def frame_for(n):
    fx, fy = WEAK_FACTORS.get(n, (n, 1))
    return BASE_RES[0] * fx, BASE_RES[1] * fy


def rank_pixel_index(resx, resy, rank, nranks, tile=TILE):
    """flat pixel indices (x + y*resx) of the tiles rank `rank` renders -- the same dealing as
    render.cpp:rank_microtiles (tiles row-major, t % nranks == rank), in tile order."""
    tw, th = (resx + tile - 1) // tile, (resy + tile - 1) // tile
    out = []
    for t in range(rank, tw * th, nranks):
        tx, ty = t % tw, t // tw
        xs = np.arange(tx * tile, min((tx + 1) * tile, resx), dtype=np.int64)
        ys = np.arange(ty * tile, min((ty + 1) * tile, resy), dtype=np.int64)
        out.append((ys[:, None] * resx + xs[None, :]).ravel())
    return np.concatenate(out) if out else np.zeros(0, np.int64)


def algorithmic_bytes_per_ray(gs, cam, params, sp):
    """per ray = 32 B ray in + 32 B x box tests + 36 B x triangle tests + 20 B hit out, with the
    test counts THIS kernel's algorithm performs per ray.  The counts come from one extra, untimed frame
    rendered with rt_params.count_work = 1: the same traversal code instantiated with per-lane counters
    (k_trace_work), on exactly the rays of the bench frame.  (The leaf-index term of SURVEY's formula is 0:
    leaf triangles are stored in leaf order, there is no index indirection.)"""
    from raytracer_amd import abi
    import copy
    import torch
    p2 = abi.rt_params.from_buffer_copy(bytes(params))
    p2.count_work = 1
    st = abi.rt_stats()
    frame = torch.zeros(cam.resx * cam.resy * 3, dtype=torch.uint8, device="cuda")
    abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(p2), frame.data_ptr(), None, C.byref(st), sp))
    rays = st.rays_primary + st.rays_reflect + st.rays_refract
    box, tri, exact, fetch = (x / rays for x in st.closest_work)
    srays = max(1, st.rays_shadow)
    abox, atri, aexact, afetch = (x / srays for x in st.any_work)
    return dict(bytes_per_ray=32 + 32 * box + 36 * tri + 20, box_tests_per_ray=box, tri_tests_per_ray=tri,
                exact_tri_tests_per_ray=exact, record_fetches_per_ray=fetch, rays=rays,
                any_bytes_per_ray=32 + 32 * abox + 36 * atri + 1, any_box_tests_per_ray=abox,
                any_tri_tests_per_ray=atri)


def _port_frame(args):
    """one oracle-port frame in a worker process (cpu_baseline's all-cores leg)"""
    sd, res = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from raytracer_amd import host
    o = ol.OracleScene(sd)
    t0 = time.perf_counter()
    o.render_whitted(ol.oracle_camera(sd.eye, sd.view, sd.fov, *res), host.default_params("hw4", primary_only=1))
    dt = time.perf_counter() - t0
    rays = o.counters().rays_traced
    o.close()
    return rays, dt


def cpu_port_all_cores(sd, res):
    """the build's CPU restatement on all host cores.  Rays are independent, so the cores each render the
    sample frame (one process per core, fork); throughput = all their rays over the wall time of the slowest."""
    import multiprocessing as mp
    cores = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or (os.cpu_count() or 1), os.cpu_count() or 1, 64))
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        res_list = pool.map(_port_frame, [(sd, res)] * cores)
    wall = time.perf_counter() - t0
    rays = sum(r for r, _ in res_list)
    return dict(value=rays / wall / 1e6, unit="Mrays/s", cores=cores, kind="port",
                sample=f"{cores} processes, each the sample frame {res[0]}x{res[1]} (oracle port)", seconds=wall)


def cpu_baseline(sd, res):
    """The reference binary (oracle/_ref/hw4_ref, compiled from the reference's own sources in the build
    container) on a bounded sample of the same workload: same scene, same materials, `R` lowered to `res`.
    Time = its own `elapse time` (render loop only, hw4.cpp:1094-1128), 1 core (it is single-threaded)."""
    from raytracer_amd import scene
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "hw4_ref")
    sample = f"monkey (970 tris, reflect=refract=0) at {res[0]}x{res[1]}, primary + shadow rays"
    # rays of the sample, counted by the oracle (one shadow ray per primary hit and light)
    o = ol.OracleScene(sd)
    cam = ol.oracle_camera(sd.eye, sd.view, sd.fov, *res)
    from raytracer_amd import host
    t0 = time.perf_counter()
    o.render_whitted(cam, host.default_params("hw4", primary_only=1))
    port_s = time.perf_counter() - t0
    rays = o.counters().rays_traced
    o.close()
    if os.path.exists(ref_bin):
        sd2 = scene.SceneData(**{**sd.__dict__})
        sd2.resx, sd2.resy = res
        with tempfile.TemporaryDirectory() as d:
            with open(os.path.join(d, "bench.txt"), "w") as f:
                f.write(scene.write_scene_text(sd2))
            subprocess.run([ref_bin, "bench.txt"], cwd=d, check=True, stdout=subprocess.DEVNULL)
            lines = dict(l.split(":", 1) for l in open(os.path.join(d, "executionTime_bench.txt")).read().strip().splitlines())
        secs = float(lines["elapse time"].split()[0])
        return dict(value=rays / secs / 1e6, unit="Mrays/s", cores=1, kind="reference", sample=sample,
                    seconds=secs, rays=int(rays))
    return dict(value=rays / port_s / 1e6, unit="Mrays/s", cores=1, kind="port", sample=sample, seconds=port_s,
                rays=int(rays))


def knn_secondary(scene_handle, torch, n_photons, n_queries, k=100, radius=1.0, steps=3, with_cpu=True):
    """BASELINE configuration 4's shape: kNN radiance lookups/s on a synthetic photon map (positions on
    the walls of the Final scene's unit box), queries on the floor/walls."""
    from raytracer_amd import abi
    rng = np.random.default_rng(1234)
    a = np.zeros(n_photons + 1, abi.photon_dtype)
    pos = rng.uniform(-0.5, 0.5, (n_photons, 3)).astype(np.float32)
    pos[:, 2] += 0.5
    face = rng.integers(0, 5, n_photons)
    pos[face == 0, 1] = -0.5; pos[face == 1, 1] = 0.5; pos[face == 2, 0] = -0.5; pos[face == 3, 0] = 0.5; pos[face == 4, 2] = 1.0
    a["position"][1:] = pos
    a["power"][1:] = np.float32(1.0 / n_photons)
    a["phi"][1:] = rng.integers(0, 256, n_photons); a["theta"][1:] = rng.integers(0, 256, n_photons)
    del pos, face
    log(f"kNN: {n_photons} synthetic photons generated, building the heap")
    t0 = time.perf_counter()
    scene_handle.photonmap_build(abi.RT_PMAP_GLOBAL, a)
    build_s = time.perf_counter() - t0
    log(f"kNN: heap built and uploaded in {build_s:.1f} s")
    if not with_cpu:
        del a
    q = rng.uniform(-0.5, 0.5, (n_queries, 3)).astype(np.float32); q[:, 2] += 0.5
    qf = rng.integers(0, 3, n_queries)
    q[qf == 0, 1] = -0.5; q[qf == 1, 0] = -0.5; q[qf == 2, 2] = 1.0
    # queries of neighbouring pixels are neighbours in space: sort them along a Morton-like key
    key = ((q[:, 0] + 0.5) * 1023).astype(np.int64) >> 4 << 12 | ((q[:, 1] + 0.5) * 1023).astype(np.int64) >> 4 << 6 | ((q[:, 2]) * 1023).astype(np.int64) >> 4
    q = np.ascontiguousarray(q[np.argsort(key, kind="stable")])
    nrm = np.tile(np.float32([0, 1, 0]), (n_queries, 1))
    dq = torch.from_numpy(q).cuda(); dn = torch.from_numpy(nrm).cuda()
    out = torch.zeros((n_queries, 3), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream()
    L = scene_handle.L

    def run():
        abi.check(L.rt_photon_irradiance_dev(scene_handle.h, 0, dq.data_ptr(), dn.data_ptr(), n_queries, k, radius,
                                             out.data_ptr(), C.c_void_p(st.cuda_stream)))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(steps):
        run()
    e1.record(st); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    # algorithmic bytes per lookup: 24 B in + 16 B x heap nodes visited + 12 B x found + 12 B out, with the visits
    # THIS kernel performs: one counted launch (untimed) -- phase-1 record fetches of every lane + the nodes of the shared walks
    # summed over the lanes that were inside the subtree.  The reference's own visit count (CPU oracle, a sample of the same
    # queries) is reported beside it.
    scene_handle.knn_stats(True)
    run(); torch.cuda.synchronize()
    ks = scene_handle.knn_stats(False)
    ref_visits, found_mean, cpu = None, float(k), None
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as ol
        ns = min(2000, n_queries)
        sel = np.linspace(0, n_queries - 1, ns).astype(np.int64)
        qs = np.ascontiguousarray(q[sel])
        idx = np.zeros((ns, k), np.int32); d2 = np.zeros((ns, k), np.float32); fnd = np.zeros(ns, np.int32); vis = C.c_int64(0)
        t0 = time.perf_counter()
        ol.oracle().orc_pmap_knn(a.ctypes.data, n_photons, qs.ctypes.data, ns, k, radius, idx.ctypes.data, d2.ctypes.data,
                                 fnd.ctypes.data, None, C.byref(vis))
        cpu_s = time.perf_counter() - t0
        ref_visits = vis.value / ns
        found_mean = float(fnd.mean())
        cpu = dict(value=ns / cpu_s, unit="lookups/s", cores=1, kind="port", sample=f"{ns} of the same queries, oracle LocatePhoton")
    if ks[6] > 0:          # the wave-shared kernels ran
        visits = float(ks[0] + ks[12]) / n_queries
        detail = dict(kernel="k_irr_main", waves=int(ks[6]), phase1_fetches_per_lookup=float(ks[0]) / n_queries,
                      walk_nodes_per_wave=float(ks[1] + ks[3]) / float(ks[6]), inserted_per_wave=float(ks[2]) / float(ks[6]),
                      strict_redo_queries=int(ks[5]), waves_gave_up=int(ks[11]), longest_wave_ms=float(int(ks[8]) >> 24) * 1e-5)
        tkey = "knn_10m.k_irr_main"
    else:
        visits, detail, tkey = (ref_visits or 0.0), dict(kernel="k_photon_knn"), "knn_10m.k_photon_knn"
    bytes_per = 24 + 16 * visits + 12 * found_mean + 12
    gbs = bytes_per * n_queries / (ms * 1e-3) / 1e9
    return dict(metric="kNN radiance lookups/s", value=n_queries / (ms * 1e-3), unit="lookups/s",
                config=dict(workload="synthetic photon map on the Final scene's box walls", photons=n_photons,
                            queries=n_queries, k=k, radius=radius),
                ms_per_step=ms, heap_build_host_s=build_s,
                roofline=dict(bound="hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS,
                              traffic=_traffic(tkey), visits_per_lookup=visits, bytes_per_lookup=bytes_per,
                              reference_visits_per_lookup=ref_visits, **detail),
                cpu_baseline=cpu)


def mesh_secondary(torch, local_rank, steps=5, nu=1000, nv=500, tag="mesh_1m"):
    """BASELINE configuration 3's scene: the synthetic 1 M-triangle mesh (raytracer_amd/synth.py) at 3840x2160,
    primary + shadow rays.  Here the KD-tree (224 k nodes, 2.4 M leaf records, 126 MB per octant-ordered copy set)
    no longer fits any cache, so the HBM roofline of the closest-hit kernel means something."""
    from raytracer_amd import abi, host, synth
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    t0 = time.perf_counter()
    sd = synth.mesh_scene(nu, nv)
    gen_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    gs = host.Scene(sd, device=local_rank)
    build_s = time.perf_counter() - t0
    inf = gs.info()
    cam = host.camera_for(sd, BASE_RES)
    params = host.default_params("hw4", primary_only=1)
    frame = torch.zeros(BASE_RES[0] * BASE_RES[1] * 3, dtype=torch.uint8, device="cuda")
    st = abi.rt_stats()
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    tot = dict(ms=0.0, closest=0.0, anyh=0.0, rays=0, prim=0, shadow=0, any_launches=0)
    for i in range(steps + 1):
        abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(params), frame.data_ptr(), None, C.byref(st), sp))
        if i:
            tot["ms"] += st.kernel_ms; tot["closest"] += st.trace_closest_ms; tot["anyh"] += st.trace_any_ms
            tot["rays"] += st.rays_primary + st.rays_shadow; tot["prim"] += st.rays_primary
            tot["shadow"] += st.rays_shadow; tot["any_launches"] += st.trace_any_launches
    alg = algorithmic_bytes_per_ray(gs, cam, params, sp)
    avg_ms = tot["closest"] / steps
    fused = tot["any_launches"] == 0 and tot["shadow"] > 0          # the whole frame is one kernel (k_primary_only)
    bytes_per_launch = alg["bytes_per_ray"] * (tot["prim"] / steps) + (alg["any_bytes_per_ray"] * (tot["shadow"] / steps) if fused else 0.0)
    gbs = bytes_per_launch / (avg_ms * 1e-3) / 1e9
    kernel = "k_primary_only" if fused else "k_trace_closest"
    traffic = _traffic(tag + "." + kernel)
    # CPU baseline: the oracle port on a strided sample of the same frame (the reference binary would need the scene as
    # 130 MB of text and minutes of KD build printing every node)
    o = ol.OracleScene(sd)
    res = (240, 135)
    t0 = time.perf_counter()
    o.render_whitted(ol.oracle_camera(sd.eye, sd.view, sd.fov, *res), host.default_params("hw4", primary_only=1))
    cpu_s = time.perf_counter() - t0
    crays = o.counters().rays_traced
    o.close(); gs.close()
    return dict(metric="Mrays/s (KD-tree traversal, primary + shadow rays)", value=tot["rays"] / (tot["ms"] * 1e-3) / 1e6,
                unit="Mrays/s", ms_per_step=tot["ms"] / steps,
                config=dict(workload=f"synthetic displaced-sphere mesh, {sd.nprim} triangles + floor, {BASE_RES[0]}x{BASE_RES[1]}, "
                                     "primary + shadow rays", kd_nodes=int(inf.flat_nodes), leaf_records=int(inf.leaf_refs),
                            scene_device_mb=inf.device_bytes / 1e6, scene_gen_s=gen_s, kd_build_upload_s=build_s),
                roofline=dict(bound="hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS, traffic=traffic,
                              kernel=kernel, avg_launch_ms=avg_ms,
                              rays_per_launch=(tot["prim"] + (tot["shadow"] if fused else 0)) / steps,
                              bytes_per_ray=alg["bytes_per_ray"], box_tests_per_ray=alg["box_tests_per_ray"],
                              tri_tests_per_ray=alg["tri_tests_per_ray"], any_hit_bytes_per_ray=alg["any_bytes_per_ray"],
                              any_hit_ms_per_step=tot["anyh"] / steps,
                              note=("one fused kernel traces the primary and the shadow rays: algorithmic bytes of both. " if fused else "") +
                                   "algorithmic bytes count every lane's node / triangle reads; the kernel fetches a record once "
                                   "per wave on the scalar path, so achieved can exceed what HBM itself delivers"),
                cpu_baseline=dict(value=crays / cpu_s / 1e6, unit="Mrays/s", cores=1, kind="port",
                                  sample=f"the same scene at {res[0]}x{res[1]}, primary + shadow rays, oracle port", seconds=cpu_s))


def drt_secondary(torch, local_rank, res=(1920, 1080), spp=64, nu=1000, nv=500):
    """BASELINE configuration 3: distributed ray tracing at 64 spp (jittered pixel samples, thin lens, ball light =
    soft shadows, glossy reflection) on the synthetic 1 M-triangle scene.  The samplers are build-defined (the reference
    has none); every ray type is counted."""
    from raytracer_amd import abi, host, synth
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    sd = synth.mesh_scene(nu, nv, reflect=0.3)
    gs = host.Scene(sd, device=local_rank)
    cam = host.camera_for(sd, res)
    kw = dict(spp=spp, lens_radius=0.01, focal_distance=1.3, light_radius=0.5, glossy=0.05, seed=3)
    p = host.default_params("hw4", **kw)
    frame = torch.zeros(res[0] * res[1] * 3, dtype=torch.uint8, device="cuda")
    st = abi.rt_stats()
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(p), frame.data_ptr(), None, C.byref(st), sp))
    rays = st.rays_primary + st.rays_shadow + st.rays_reflect + st.rays_refract
    o = ol.OracleScene(sd)
    cres, cspp = (48, 27), 8
    pc = host.default_params("hw4", **{**kw, "spp": cspp})
    t0 = time.perf_counter()
    o.render_whitted(ol.oracle_camera(sd.eye, sd.view, sd.fov, *cres), pc)
    cpu_s = time.perf_counter() - t0
    crays = o.counters().rays_traced
    o.close(); gs.close()
    return dict(metric="Mrays/s (distributed ray tracing, all ray types)", value=rays / (st.kernel_ms * 1e-3) / 1e6, unit="Mrays/s",
                ms_per_frame=st.kernel_ms, passes=st.passes,
                config=dict(workload=f"synthetic mesh, {sd.nprim} triangles, {res[0]}x{res[1]}, {spp} spp, lens 0.01 / focus 1.3, "
                                     "light ball 0.5, glossy 0.05, reflect 0.3 (sphere) / 0.5 (floor), depth <= 10"),
                rays=dict(primary=st.rays_primary, shadow=st.rays_shadow, reflect=st.rays_reflect, refract=st.rays_refract),
                trace_closest_ms=st.trace_closest_ms, trace_any_ms=st.trace_any_ms,
                cpu_baseline=dict(value=crays / cpu_s / 1e6, unit="Mrays/s", cores=1, kind="port",
                                  sample=f"the same scene and samplers at {cres[0]}x{cres[1]}, {cspp} spp, oracle port", seconds=cpu_s))


def photon_frame_secondary(torch, local_rank, n_global=1_000_000, n_caustic=100_000, res=(192, 192)):
    """BASELINE configuration 4's pipeline end to end on the bundled Final scene: GPU emission of both maps, host balance,
    then the photon-mapping frame (caustic kNN r=0.1 + 50 gather rays with global kNN r=1.0, k=100, 400 light samples)."""
    from raytracer_amd import abi, host, scene
    from raytracer_amd import dist as rdist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    sd = scene.parse_scene_file(os.path.join(ROOT, "tests", "golden", "final_input.txt"), flavour="final")
    gs = host.Scene(sd, device=local_rank)
    t0 = time.perf_counter()
    maps = rdist.build_photon_maps(gs, torch, (n_global, n_caustic), seed=1)
    maps_s = time.perf_counter() - t0
    cam = host.camera_for(sd, res)
    p = host.default_params("final", aa_taps=1)
    frame = torch.zeros(res[0] * res[1] * 3, dtype=torch.uint8, device="cuda")
    st = abi.rt_stats()
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(p), frame.data_ptr(), None, C.byref(st), sp))
    gs.knn_stats(True)                       # one more frame, untimed, with the kNN kernels' work counters on
    st2 = abi.rt_stats()
    abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(p), frame.data_ptr(), None, C.byref(st2), sp))
    ks = gs.knn_stats(False)
    rays = st.rays_primary + st.rays_shadow + st.rays_reflect + st.rays_refract + st.rays_gather
    # CPU baseline: the oracle's restatement of the same frame at 10x10 pixels, same maps
    o = ol.OracleScene(sd)
    cres = (10, 10)
    t0 = time.perf_counter()
    o.render_photon(ol.oracle_camera(sd.eye, sd.view, sd.fov, *cres), p, maps[0]["heap"], maps[1]["heap"])
    cpu_s = time.perf_counter() - t0
    c = o.counters()
    crays = c.rays_primary + c.rays_shadow + c.rays_reflect + c.rays_refract + c.rays_gather
    o.close(); gs.close()
    knn_detail = dict(waves=int(ks[6]), strict_redo_queries=int(ks[5]), waves_gave_up=int(ks[11]),
                      walk_nodes_per_wave=float(ks[1] + ks[3]) / max(1.0, float(ks[6])),
                      inserted_per_wave=float(ks[2]) / max(1.0, float(ks[6])), phase1_fetches_per_lookup=float(ks[0]) / max(1, st.knn_lookups))
    return dict(metric="photon-mapping frame: Mrays/s and kNN radiance lookups/s", value=rays / (st.kernel_ms * 1e-3) / 1e6, knn=knn_detail,
                unit="Mrays/s", lookups_per_s=st.knn_lookups / (st.kernel_ms * 1e-3), ms_per_frame=st.kernel_ms,
                config=dict(workload=f"bundled Final scene, {res[0]}x{res[1]}, 1 AA tap, 50 gather rays, 20x20 light samples, k=100",
                            photons_global=n_global, photons_caustic=n_caustic, emissions_global=maps[0]["emissions"],
                            emissions_caustic=maps[1]["emissions"], maps_emit_balance_upload_s=maps_s),
                rays=dict(primary=st.rays_primary, shadow=st.rays_shadow, reflect=st.rays_reflect, refract=st.rays_refract,
                          gather=st.rays_gather), knn_lookups=st.knn_lookups,
                cpu_baseline=dict(value=crays / cpu_s / 1e6, unit="Mrays/s", lookups_per_s=c.knn_lookups / cpu_s, cores=1,
                                  kind="port", sample=f"the same frame at {cres[0]}x{cres[1]}, same photon maps, oracle port",
                                  seconds=cpu_s))


def _traffic(key):
    """HBM bytes per launch from the committed PMC summary (tools/prof_pmc.sh -> profiles/traffic_r01.json)"""
    tf = os.path.join(ROOT, "profiles", "traffic_r01.json")
    if not os.path.exists(tf):
        return None
    try:
        d = json.load(open(tf))
        for part in key.split("."):
            d = d[part]
        return d.get("hbm_bytes_per_launch")
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-knn", action="store_true", help="skip the secondary lines (kNN, 1M-triangle mesh, photon frame)")
    ap.add_argument("--only-secondary", type=str, default="", help="comma list of secondary lines to run: knn,mesh_1m,drt_1m,photon_frame (and, on request only: stress_mesh_10m,stress_knn_200m)")
    ap.add_argument("--knn-photons", type=int, default=10_000_000)
    ap.add_argument("--knn-queries", type=int, default=1920 * 1080)
    ap.add_argument("--cpu-sample", type=str, default="960x540")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: exercise the sharding / gather logic only (tests)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        log(f"warning: WORLD_SIZE {world} != --gpus {args.gpus}; using WORLD_SIZE")
    n = world if world > 1 else 1
    dry = args.dry_run
    dev = torch.device("cpu") if dry else torch.device("cuda", local_rank)
    if not dry:
        torch.cuda.set_device(dev)
    if n > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo" if dry else "nccl")

    from raytracer_amd import abi, host
    sd = bench_scene()
    resx, resy = frame_for(n) if not dry else (256 * WEAK_FACTORS.get(n, (n, 1))[0], 128 * WEAK_FACTORS.get(n, (n, 1))[1])
    npix = resx * resy
    packed = n > 1                      # a rank's tiles leave the kernel as one contiguous buffer (rt_params.packed_output)
    cam = host.camera_for(sd, (resx, resy))
    if not dry:
        gs = host.Scene(sd, device=local_rank)
        params = host.default_params("hw4", primary_only=1, tile_size=TILE if n > 1 else 0, tile_rank=rank, tile_nranks=n,
                                     packed_output=1 if packed else 0)
        stream = torch.cuda.current_stream()
        sp = C.c_void_p(stream.cuda_stream)
    if packed:
        # pixel lists in the kernel's order (rt_render_pixel_list; -1 = a slot outside the frame)
        lists = [host.pixel_list(cam, host.default_params("hw4", primary_only=1, tile_size=TILE, tile_rank=r, tile_nranks=n))
                 for r in range(n)]
        counts = [int(l.size) for l in lists]
        my_pix = None
    else:
        my_pix = torch.from_numpy(rank_pixel_index(resx, resy, rank, n)).to(dev)
        counts = [len(rank_pixel_index(resx, resy, r, n)) for r in range(n)] if n > 1 else [npix]
    maxc = max(counts)
    frame = torch.zeros(npix * 3, dtype=torch.uint8, device=dev)
    sendbuf = [torch.zeros((maxc, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    recv = [[torch.zeros((maxc, 3), dtype=torch.uint8, device=dev) for _ in range(n)] for _ in range(2)] if (n > 1 and rank == 0) else None
    final = torch.zeros((npix, 3), dtype=torch.uint8, device=dev) if rank == 0 else None
    if packed:
        all_pix = [torch.from_numpy(l.astype(np.int64)).to(dev) for l in lists] if rank == 0 else None
    else:
        all_pix = [torch.from_numpy(rank_pixel_index(resx, resy, r, n)).to(dev) for r in range(n)] if (n > 1 and rank == 0) else None
    st = abi.rt_stats()
    tot = dict(rays=0, closest_ms=0.0, any_ms=0.0, closest_launches=0, closest_rays=0, any_launches=0, shadow_rays=0)
    pending = [None, None]

    def step(i, timed):
        b = i & 1
        last[0] = i
        if dry and packed:
            # what the kernel would write: slot k of this rank's pixel list gets the colour of its pixel (here a
            # function of the pixel index, the rank and the step, so that the assembled frame can be checked exactly)
            if pending[b] is not None:
                pending[b].wait(); pending[b] = None
            pix = torch.from_numpy(lists[rank].astype(np.int64))
            sendbuf[b][: counts[rank], 0] = (pix % 251).to(torch.uint8)
            sendbuf[b][: counts[rank], 1] = rank + 1
            sendbuf[b][: counts[rank], 2] = i & 255
        elif dry:
            frame.view(-1, 3)[my_pix] = torch.tensor([rank + 1, i & 255, 7], dtype=torch.uint8)
        else:
            if packed and pending[b] is not None:     # the kernel writes the send buffer itself: its gather of two steps
                pending[b].wait()                     # ago must have left
                pending[b] = None
            # timed steps ask for no rt_stats: the frames queue on the stream back to back (rt_stats needs a device -> host
            # copy and a synchronisation per frame).  Every step renders the same frame, so its ray counts are those of the
            # instrumented frames that follow the timed region.
            abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(params), sendbuf[b].data_ptr() if packed else frame.data_ptr(),
                                         None, C.byref(st) if timed == "stats" else None, sp))
            if timed == "stats":
                tot["rays"] += st.rays_primary + st.rays_shadow
                tot["closest_ms"] += st.trace_closest_ms; tot["any_ms"] += st.trace_any_ms
                tot["closest_launches"] += st.trace_closest_launches; tot["closest_rays"] += st.rays_primary
                tot["any_launches"] += st.trace_any_launches; tot["shadow_rays"] += st.rays_shadow
        if n > 1:
            if pending[b] is not None:                # the buffer pair of two steps ago must have left
                pending[b].wait()
            if not packed:
                sendbuf[b][: counts[rank]] = frame.view(-1, 3)[my_pix]
            pending[b] = dist.gather(sendbuf[b], recv[b] if rank == 0 else None, dst=0, async_op=True)

    last = [0]

    def drain():
        # wait for the gathers in flight; rank 0 then puts the newest frame's tiles into display order
        for b in (0, 1):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None
        if rank == 0:
            b = last[0] & 1
            for r in range(n):
                if packed:
                    ok = all_pix[r] >= 0
                    final[all_pix[r][ok]] = recv[b][r][: counts[r]][ok]
                else:
                    final[all_pix[r]] = recv[b][r][: counts[r]]

    def sync():
        if not dry:
            torch.cuda.synchronize()
        if n > 1:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i, False)
    if n > 1:
        drain()
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, True)
    if n > 1:
        drain()
    sync()
    dt = time.perf_counter() - t0
    # the same frame again with rt_stats, for the ray counts and the per-kernel HIP-event times of the roofline
    nstat = 0
    if not dry:
        nstat = max(1, min(args.steps, 20))
        for i in range(nstat):
            step(i, "stats")
        if n > 1:
            drain()
        sync()
        tot["rays"] = tot["rays"] / nstat * args.steps
    if n > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        r = torch.tensor([float(tot["rays"])], dtype=torch.float64, device=dev)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        total_rays = float(r.item())
    else:
        total_rays = float(tot["rays"])

    if dry:
        if rank == 0:
            if n > 1:
                ok = True
                for r in range(n):
                    pix = all_pix[r][all_pix[r] >= 0]
                    got = final[pix]
                    ok = ok and bool((got[:, 0] == (pix % 251).to(torch.uint8)).all()) and bool((got[:, 1] == r + 1).all()) \
                        and bool((got[:, 2] == (last[0] & 255)).all())
                ok = ok and sum(int((all_pix[r] >= 0).sum()) for r in range(n)) == npix      # every pixel has exactly one owner
            else:
                ok = True
            print(json.dumps({"dry_run": True, "n_gpus": n, "frame": [resx, resy], "assembled_ok": ok,
                              "pixels_per_rank": counts}))
        if n > 1:
            dist.destroy_process_group()
        return

    out = None
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = total_rays / dt / 1e6
        out = {"metric": "Mrays/s (KD-tree traversal, primary + shadow rays)", "value": value, "unit": "Mrays/s",
               "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "BASELINE configs[1]: bundled monkey scene (970 triangles, 138 KD nodes), "
                                      f"{resx}x{resy}, primary + shadow rays only, reflect=refract=0",
                          "frame": [resx, resy], "rays_per_step": total_rays / args.steps,
                          "parallelism": f"tiles{TILE}x{TILE} round-robin over {n} GPU(s), scene replicated, gather to rank 0"}}
        if n > 1:
            # the frame rank 0 assembled from the ranks' last gather against the same frame rendered whole on rank 0
            pw = host.default_params("hw4", primary_only=1)
            abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(pw), frame.data_ptr(), None, None, sp))
            torch.cuda.synchronize()
            out["config"]["assembled_frame_equals_one_gpu_frame"] = bool(torch.equal(final.view(-1), frame))
        # roofline of the dominant kernel (k_trace_closest): algorithmic bytes / HIP-event time
        alg = algorithmic_bytes_per_ray(gs, cam, host.default_params("hw4", primary_only=1, tile_size=TILE if n > 1 else 0, tile_rank=rank,
                                                                      tile_nranks=n), sp)
        launches = max(1, tot["closest_launches"])
        rays_per_launch = tot["closest_rays"] / launches
        avg_ms = tot["closest_ms"] / launches
        fused = tot["any_launches"] == 0 and tot["shadow_rays"] > 0       # the whole frame is ONE kernel (k_primary_only)
        shadow_per_launch = tot["shadow_rays"] / launches
        bytes_per_launch = alg["bytes_per_ray"] * rays_per_launch + (alg["any_bytes_per_ray"] * shadow_per_launch if fused else 0.0)
        gbs = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        kernel = "k_primary_only" if fused else "k_trace_closest"
        traffic = _traffic("monkey_4k." + kernel)
        out["roofline"] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                           "traffic": traffic, "kernel": kernel, "avg_launch_ms": avg_ms,
                           "rays_per_launch": rays_per_launch + (shadow_per_launch if fused else 0.0),
                           "bytes_per_ray": alg["bytes_per_ray"],
                           "box_tests_per_ray": alg["box_tests_per_ray"], "tri_tests_per_ray": alg["tri_tests_per_ray"],
                           "exact_tri_tests_per_ray": alg["exact_tri_tests_per_ray"],
                           "any_hit_bytes_per_ray": alg["any_bytes_per_ray"],
                           "any_hit_ms_per_step": tot["any_ms"] / max(1, nstat),
                           "note": ("one fused kernel traces the primary AND the shadow rays (generate, closest hit, any hit, shade, "
                                    "pixel): algorithmic bytes of both. " if fused else "") +
                                   "The scene (the build's BVH over the tree triangles, DESIGN 4.1e: 41 KB of nodes per octant "
                                   "layout + 46 KB of triangles) is cache resident: the algorithmic bytes are served by the scalar "
                                   "cache / L2, not HBM, and the kernel is bound by vector-instruction issue (SQ counters: the "
                                   "vector instructions alone account for the kernel time, profiles/r01_sq_monkey_4k.json); "
                                   "see DESIGN.md"}
        if not args.no_cpu_baseline and n == 1:      # the CPU leg is timed at N = 1 only
            w, h = (int(x) for x in args.cpu_sample.split("x"))
            out["cpu_baseline"] = cpu_baseline(sd, (w, h))
            try:
                out["cpu_baseline"]["all_cores"] = cpu_port_all_cores(sd, (w, h))
            except Exception as e:      # a reported extra, never a reason to lose the line
                out["cpu_baseline"]["all_cores"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_knn and n == 1:
            sec = {}
            for name, fn in (("knn", lambda: knn_secondary(gs, torch, args.knn_photons, args.knn_queries)),
                             ("mesh_1m", lambda: mesh_secondary(torch, local_rank)),
                             ("drt_1m", lambda: drt_secondary(torch, local_rank)),
                             ("photon_frame", lambda: photon_frame_secondary(torch, local_rank)),
                             # one GPU's share of BASELINE configuration 5 (scene and photon map are replicated per GPU)
                             ("stress_mesh_10m", lambda: mesh_secondary(torch, local_rank, steps=3, nu=3162, nv=1581, tag="stress_mesh_10m")),
                             ("stress_knn_200m", lambda: knn_secondary(gs, torch, 200_000_000, args.knn_queries, steps=2, with_cpu=False))):
                if args.only_secondary and name not in args.only_secondary.split(","):
                    continue
                if name.startswith("stress_") and name not in args.only_secondary.split(","):
                    continue                      # BASELINE configuration 5's sizes: minutes of host build, on request only
                try:
                    sec[name] = fn()
                except Exception as e:           # a secondary line must never take the headline down with it
                    sec[name] = {"error": f"{type(e).__name__}: {e}"}
            out["secondary"] = sec
    if n > 1 and not args.no_knn:
        # the metric's second half at N GPUs: every rank answers its own share of the radiance lookups (the queries
        # follow their pixels, the photon map is replicated, no exchange: SURVEY 8e), weak scaling like the frame
        try:
            r = knn_secondary(gs, torch, args.knn_photons, args.knn_queries, steps=3, with_cpu=False)
            ms = torch.tensor([r["ms_per_step"]], dtype=torch.float64, device="cuda")
        except Exception as e:
            log(f"rank {rank}: kNN line failed: {type(e).__name__}: {e}")
            r, ms = None, torch.tensor([float("inf")], dtype=torch.float64, device="cuda")
        dist.all_reduce(ms, op=dist.ReduceOp.MAX)
        if rank == 0:
            if r is not None and ms.item() != float("inf"):
                r["value"] = n * args.knn_queries / (ms.item() * 1e-3)
                r["ms_per_step"] = ms.item(); r["n_gpus"] = n
                r["config"]["queries"] = n * args.knn_queries
                r["note"] = "aggregate over the ranks: each answers its own queries on its replica of the map; time = slowest rank"
                out["secondary"] = {"knn": r}
            else:
                out["secondary"] = {"knn": {"error": "a rank failed"}}
    if n > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
