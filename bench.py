#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X hot path on BASELINE.json's configuration 3 (`configs[2]`, the configuration
the metric's "1/2/4/8 GPUs" is quoted on): distributed ray tracing at 64 spp (jittered pixel samples, thin lens, ball
light = soft shadows, glossy reflection, recursion depth <= 10) on the synthetic 1 M-triangle scene at 1920x1080, traced
by librt_hip.so through the C ABI.  Every ray type is counted (primary + shadow + reflection + refraction).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one frame of the workload.  Everything the timed region reads (scene, KD-trees, the build's hierarchy) is
resident in HBM before it starts; rays are generated on the GPU.  N > 1: the frame grows with N (weak scaling:
1920x1080 pixels x 64 spp per GPU), 64x64 tiles are dealt round-robin to the ranks (no collective in the tracing path)
and every step ends with the RCCL gather of the ranks' tiles to rank 0, as north_star prescribes; the same frame at a
FIXED size over N GPUs (strong scaling) is the line `secondary.drt_1m_strong`.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      the dominant kernel against the limiter its counters support: `bound` is "hbm" when HBM
                traffic is the largest of the three busy fractions, else "valu_issue" / "scalar_issue"; achieved / peak /
                frac belong to that bound and frac <= 1 by construction.  Beside them: the SURVEY 8d algorithmic bytes
                (per lane), the bytes the waves actually fetch (a record once per wave), the PMC HBM traffic, and
                `counters_commit` / `counters_stale` (the commit the counters were collected at; stale = the kernel
                sources have changed since).
  cpu_baseline  the CPU leg on a bounded sample of the same workload, cores stated.  For the headline it is the oracle
                port (`kind: "port"`): the reference has no distributed-ray-tracing samplers, so its binary cannot run
                this workload; the lines whose workload the reference CAN run (monkey_4k, mesh_1m, knn, photon_frame)
                time the reference itself (`kind: "reference"`).
  secondary     monkey_4k (BASELINE configs[1]), mesh_1m, knn, photon_frame, photon_frame_full (configs[3] at its stated
                size), stress_mesh_10m, stress_knn_200m (one GPU's share of configs[4]), each with its own roofline,
                cpu_baseline and a parity sample checked against the oracle inside this run.
Exit status: 0 = every line ran and its parity sample agreed; 3 = an N > 1 run stalled in a collective (the watchdog
printed what there was); 4 = a secondary line failed or a parity sample disagreed (the JSON line is printed first).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md: HBM3E 8 TB/s peak (6.3 achievable); 256 CUs x 4 SIMD-32 at 2.4 GHz, a wave64 vector instruction
# issues in 2 cycles (constants table: v_fma_f32 2 cyc; 4 only for a wave alone on its SIMD); one scalar unit per CU
HBM_PEAK_GBS = 8000.0
VALU_PEAK_GINST = 1024 * 2.4 / 2.0          # wave-instructions / ns  -> 1228.8 G wave-instructions / s
SCALAR_PEAK_GINST = 256 * 2.4               # SALU + SMEM + branch instructions, one per cycle and CU -> 614.4 G / s
BASE_RES = (3840, 2160)                     # the 4K frames of monkey_4k / mesh_1m / stress_mesh_10m
HEAD_RES, HEAD_SPP = (1920, 1080), 64       # the headline: BASELINE configs[2]
TILE = 64
WEAK_FACTORS = {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (4, 2)}
PROFILE_TAGS = ("r03", "r02")               # profiles/<tag>_counters.json, newest first


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def _jsonable(o):
    """numpy scalars / arrays that found their way into the line"""
    if isinstance(o, np.generic):
        return o.item()
    if isinstance(o, np.ndarray):
        return o.tolist()
    return str(o)


def dumps(line):
    return json.dumps(line, default=_jsonable)


def bench_scene():
    """monkey.txt with every material's reflect/refract set to 0."""
    from raytracer_amd import scene
    sd = scene.parse_scene_file(os.path.join(ROOT, "tests", "golden", "monkey.txt"))
    sd.materials = sd.materials.copy()
    sd.materials[:, 7] = 0.0
    sd.materials[:, 8] = 0.0
    return sd


def frame_for(n):
    """weak scaling of the headline: 1920x1080 pixels (x 64 spp) per GPU"""
    fx, fy = WEAK_FACTORS.get(n, (n, 1))
    return HEAD_RES[0] * fx, HEAD_RES[1] * fy


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


# ---------------------------------------------------------------------------------------------------------------------
# roofline

_COUNTERS = None
KERNEL_SOURCE_GLOBS = ("raytracer_amd/csrc/*.hip", "raytracer_amd/csrc/*.h", "raytracer_amd/csrc/*.cpp", "raytracer_amd/csrc/Makefile")


def kernel_source_sha():
    """sha256 over the library's sources (sorted by name): what the counters in profiles/ are stamped with
    (tools/prof_counters.py) and compared against, so that counters older than the kernels are flagged"""
    import glob
    h = hashlib.sha256()
    for g in KERNEL_SOURCE_GLOBS:
        for f in sorted(glob.glob(os.path.join(ROOT, g))):
            h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def counters(workload, kernel):
    """per-launch counters of `kernel` in `workload` from the committed rocprofv3 passes (tools/prof_all.sh ->
    tools/prof_counters.py -> profiles/<tag>_counters.json): HBM bytes (PMC, gfx950-corrected), L2 hit rate, SQ instruction
    counts.  They are properties of the kernel on this workload, not of a run; the time is measured live.  Returns
    (counters or None, provenance): provenance names the file, the commit and source hash the workload's counters were
    collected at, and whether the sources have changed since (`stale`)."""
    global _COUNTERS
    if _COUNTERS is None:
        _COUNTERS = {"_files": []}
        live = kernel_source_sha()
        for tag in PROFILE_TAGS:             # newest first: a workload's counters come from the newest file that has them
            p = os.path.join(ROOT, "profiles", f"{tag}_counters.json")
            if not os.path.exists(p):
                continue
            try:
                j = json.load(open(p))
            except Exception:
                continue
            meta = j.get("_meta", {})
            for w, ks in j.items():
                if w.startswith("_") or w in _COUNTERS:
                    continue
                m = meta.get(w, {})
                _COUNTERS[w] = dict(kernels=ks, prov=dict(source=f"profiles/{tag}_counters.json", commit=m.get("commit"),
                                                          kernel_source_sha=m.get("kernel_source_sha"),
                                                          stale=(m.get("kernel_source_sha") != live)))
    w = _COUNTERS.get(workload)
    if not w:
        return None, dict(source=None, commit=None, kernel_source_sha=None, stale=None)
    for k, v in w["kernels"].items():
        if k == kernel or k.startswith(kernel + "<"):
            return v, w["prov"]
    return None, w["prov"]


def make_roofline(workload, kernel, avg_ms, units_per_launch, alg_bytes, wave_bytes=None, counter_scale=1.0, **detail):
    """The dominant kernel against the three limiters the counters can speak for.
      hbm      useful HBM bytes / time / 8 TB/s, useful = min(algorithmic bytes, PMC traffic): what the algorithm needs
               out of what was moved (traffic above the algorithmic bytes is waste, algorithmic bytes above the traffic
               were served by the scalar cache / L2 and never were HBM's to deliver)
      valu     vector wave-instructions / time / (1024 SIMDs x 2.4 GHz / 2 cycles)
      scalar   SALU + SMEM + branch instructions / time / (256 CUs x 2.4 GHz)
    `bound` names the largest, achieved / peak / unit / frac are its numbers (<= 1 by construction).
    counter_scale: the committed counters are means over EVERY launch of the kernel in the profiled run; where that run also
    holds launches that are not the workload's (the photon frame's 10x10 parity sample: as many launches, next to no work)
    the caller gives the ratio all launches / the workload's launches, and the per-launch counts are scaled by it."""
    c, src = counters(workload, kernel)
    if c and counter_scale != 1.0:
        c = dict(c)
        for key in ("hbm_bytes_per_launch", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
            if c.get(key):
                c[key] = c[key] * counter_scale
    t = max(avg_ms, 1e-9) * 1e-3
    traffic = c.get("hbm_bytes_per_launch") if c else None
    useful = min(alg_bytes, traffic) if traffic else None
    fr = {}
    if useful is not None:
        fr["hbm"] = (useful / t / 1e9, HBM_PEAK_GBS, "GB/s")
    if c and c.get("SQ_INSTS_VALU"):
        fr["valu_issue"] = (c["SQ_INSTS_VALU"] / t / 1e9, VALU_PEAK_GINST, "G wave-instructions/s")
        sc = (c.get("SQ_INSTS_SALU", 0) or 0) + (c.get("SQ_INSTS_SMEM", 0) or 0) + (c.get("SQ_INSTS_BRANCH", 0) or 0)
        if sc:
            fr["scalar_issue"] = (sc / t / 1e9, SCALAR_PEAK_GINST, "G instructions/s")
    if fr:
        bound = max(fr, key=lambda k: fr[k][0] / fr[k][1])
        # a kernel whose HBM TRAFFIC (lines moved, useful or not) is a larger share of its peak than either issue rate is of
        # theirs is bound by memory, whatever share of those lines the algorithm asked for: `bound` says so, `achieved` stays
        # the useful bytes per second, and `hbm_traffic_frac` below is the traffic's own share
        if traffic and "hbm" in fr and traffic / t / 1e9 / HBM_PEAK_GBS > fr[bound][0] / fr[bound][1]:
            bound = "hbm"
        ach, peak, unit = fr[bound]
        note = None
    else:      # no counters committed for this kernel: the SURVEY 8d figure alone, capped so that it stays a fraction
        bound, ach, peak, unit = "hbm", min(alg_bytes / t / 1e9, HBM_PEAK_GBS), HBM_PEAK_GBS, "GB/s"
        note = "no PMC / SQ counters committed for this kernel: algorithmic bytes only (capped at the peak)"
    out = dict(bound=bound, achieved=ach, peak=peak, unit=unit, frac=ach / peak, traffic=traffic, kernel=kernel,
               avg_launch_ms=avg_ms, units_per_launch=units_per_launch,
               algorithmic_bytes_per_launch=alg_bytes, algorithmic_gbs=alg_bytes / t / 1e9,
               wave_fetched_bytes_per_launch=wave_bytes,
               hbm_traffic_gbs=(traffic / t / 1e9) if traffic else None,
               # the traffic itself against the peak (lines moved, useful or not): the 200 M-photon estimate and the path kernel
               # are limited by it (scattered 64-byte records / 128-byte lines), whatever `frac` says about the useful bytes
               hbm_traffic_frac=(traffic / t / 1e9 / HBM_PEAK_GBS) if traffic else None,
               fractions={k: v[0] / v[1] for k, v in fr.items()},
               l2_hit_rate=c.get("l2_hit_rate") if c else None,
               wait_inst_fraction=(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]) if c and c.get("SQ_WAVE_CYCLES") and c.get("SQ_WAIT_INST_ANY") else None,
               wait_memory_fraction=(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]) if c and c.get("SQ_WAVE_CYCLES") and c.get("SQ_WAIT_ANY") else None,
               counters_source=src.get("source"), counters_commit=src.get("commit"),
               counters_kernel_source_sha=src.get("kernel_source_sha"), counters_stale=src.get("stale"), **detail)
    if note:
        out["note"] = note
    return out


def traversal_bytes(work, rays, out_bytes=20):
    """SURVEY 8d per-lane algorithmic bytes of a traversal launch: 32 B ray in + 32 B x box tests + 36 B x triangle records
    + the result, summed over the launch, from the device work counters (box, tri, exact, fetch); and the bytes its waves
    fetch: `fetch` counts a wave-uniform record fetch once per lane, a record is <= 64 B (BVH node 64, triangle 48)."""
    box, tri, exact, fetch = (float(x) for x in work)
    alg = 32.0 * rays + 32.0 * box + 36.0 * tri + out_bytes * rays
    return alg, fetch / 64.0 * 64.0


def render_work(gs, cam, params, sp, torch):
    """one extra, untimed frame with rt_params.count_work: the traversal code with per-lane counters on the frame's rays"""
    from raytracer_amd import abi
    p2 = abi.rt_params.from_buffer_copy(bytes(params))
    p2.count_work = 1
    st = abi.rt_stats()
    frame = torch.zeros(cam.resx * cam.resy * 3, dtype=torch.uint8, device="cuda")
    abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(p2), frame.data_ptr(), None, C.byref(st), sp))
    return st


# ---------------------------------------------------------------------------------------------------------------------
# CPU legs.  Everything that forks (the reference binaries, the process pool) runs BEFORE the process touches the GPU.

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
    sample frame (one process per core, spawned); throughput = all their rays over the wall time of the slowest."""
    import multiprocessing as mp
    cores = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or (os.cpu_count() or 1), os.cpu_count() or 1, 64))
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(cores) as pool:
        res_list = pool.map_async(_port_frame, [(sd, res)] * cores).get(timeout=180)
    wall = time.perf_counter() - t0
    rays = sum(r for r, _ in res_list)
    busy = max(d for _, d in res_list)
    return dict(value=rays / busy / 1e6, unit="Mrays/s", cores=cores, kind="port",
                sample=f"{cores} processes, each the sample frame {res[0]}x{res[1]} (oracle port); time = the slowest "
                       "process's render loop", seconds=busy, wall_with_spawn_s=wall)


def _run_reference_hw4(sd, res, name):
    """the reference binary on `sd` at `res`: its own `elapse time` (render loop only, hw4.cpp:1094-1128) and KD log"""
    from raytracer_amd import scene
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "hw4_ref")
    if not os.path.exists(ref_bin):
        return None
    sd2 = scene.SceneData(**{**sd.__dict__})
    sd2.resx, sd2.resy = res
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, name + ".txt"), "w") as f:
            f.write(scene.write_scene_text(sd2))
        t0 = time.perf_counter()
        subprocess.run([ref_bin, name + ".txt"], cwd=d, check=True, stdout=subprocess.DEVNULL, timeout=600)
        wall = time.perf_counter() - t0
        lines = dict(l.split(":", 1) for l in open(os.path.join(d, "executionTime_" + name + ".txt")).read().strip().splitlines())
    return dict(seconds=float(lines["elapse time"].split()[0]), wall_s=wall,
                kd_nodes=int(lines["total kd-tree node count"]), kd_depth=int(lines["max depth of kd-tree subdivision"]),
                avg_check=float(lines["average ray-mesh Intersection check count"]))


def cpu_baseline_monkey(sd, res):
    """The reference binary (oracle/_ref/hw4_ref, compiled from the reference's own sources in the build container) on a
    bounded sample of the headline workload: same scene, same materials, `R` lowered to `res`; 1 core (single-threaded)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from raytracer_amd import host
    sample = f"monkey (970 tris, reflect=refract=0) at {res[0]}x{res[1]}, primary + shadow rays"
    o = ol.OracleScene(sd)
    t0 = time.perf_counter()
    o.render_whitted(ol.oracle_camera(sd.eye, sd.view, sd.fov, *res), host.default_params("hw4", primary_only=1))
    port_s = time.perf_counter() - t0
    rays = o.counters().rays_traced            # one shadow ray per primary hit and light
    o.close()
    ref = _run_reference_hw4(sd, res, "bench")
    if ref:
        return dict(value=rays / ref["seconds"] / 1e6, unit="Mrays/s", cores=1, kind="reference", sample=sample,
                    seconds=ref["seconds"], rays=int(rays), port_seconds=port_s)
    return dict(value=rays / port_s / 1e6, unit="Mrays/s", cores=1, kind="port", sample=sample, seconds=port_s, rays=int(rays))


def cpu_baseline_mesh(res=(320, 180), nu=1000, nv=500):
    """SURVEY 8d last bullet: the reference binary itself on the 1 M-triangle scene, written in its own text format
    (152 MB); render loop `elapse time` at `res`, rays counted by the oracle on the same frame.  Its KD log is kept: the
    reference builds the same tree the library uploads."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from raytracer_amd import host, synth
    sd = synth.mesh_scene(nu, nv)
    sample = f"the same scene ({sd.nprim} triangles) at {res[0]}x{res[1]}, primary + shadow rays"
    o = ol.OracleScene(sd)
    t0 = time.perf_counter()
    o.render_whitted(ol.oracle_camera(sd.eye, sd.view, sd.fov, *res), host.default_params("hw4", primary_only=1))
    port_s = time.perf_counter() - t0
    rays = o.counters().rays_traced
    oi = o.info()
    o.close()
    try:
        ref = _run_reference_hw4(sd, res, "mesh1m")
    except Exception as e:
        log(f"reference binary on the 1 M-triangle scene failed: {type(e).__name__}: {e}")
        ref = None
    if ref:
        return dict(value=rays / ref["seconds"] / 1e6, unit="Mrays/s", cores=1, kind="reference", sample=sample,
                    seconds=ref["seconds"], rays=int(rays), port_seconds=port_s, whole_run_wall_s=ref["wall_s"],
                    reference_kd_nodes=ref["kd_nodes"], reference_kd_depth=ref["kd_depth"],
                    reference_avg_check=ref["avg_check"],
                    kd_equals_library=bool(ref["kd_nodes"] == oi.kd_nodes_built and ref["kd_depth"] == oi.kd_max_depth))
    return dict(value=rays / port_s / 1e6, unit="Mrays/s", cores=1, kind="port", sample=sample + ", oracle port", seconds=port_s,
                rays=int(rays))


# ---------------------------------------------------------------------------------------------------------------------
# secondary workloads (one GPU)

def knn_map(n_photons, seed=1234):
    rng = np.random.default_rng(seed)
    from raytracer_amd import abi
    a = np.zeros(n_photons + 1, abi.photon_dtype)
    pos = rng.uniform(-0.5, 0.5, (n_photons, 3)).astype(np.float32)
    pos[:, 2] += 0.5
    face = rng.integers(0, 5, n_photons)
    pos[face == 0, 1] = -0.5; pos[face == 1, 1] = 0.5; pos[face == 2, 0] = -0.5; pos[face == 3, 0] = 0.5; pos[face == 4, 2] = 1.0
    a["position"][1:] = pos
    a["power"][1:] = np.float32(1.0 / n_photons)
    a["phi"][1:] = rng.integers(0, 256, n_photons); a["theta"][1:] = rng.integers(0, 256, n_photons)
    return a, rng


def knn_secondary(scene_handle, torch, n_photons, n_queries, k=100, radius=1.0, steps=3, with_cpu=True, cpu_queries=2000):
    """BASELINE configuration 4's shape: kNN radiance lookups/s on a synthetic photon map (positions on
    the walls of the Final scene's unit box), queries on the floor/walls."""
    from raytracer_amd import abi
    a, rng = knn_map(n_photons)
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
    # THIS kernel performs: one counted launch (untimed) -- phase-1 record fetches of every lane + the nodes of the walks
    # summed over the lanes that were inside the subtree.  The reference's own visit count (CPU oracle, a sample of the same
    # queries) is reported beside it.
    scene_handle.knn_stats(True)
    run(); torch.cuda.synchronize()
    ks = scene_handle.knn_stats(False)
    ref_visits, found_mean, cpu, parity = None, float(k), None, None
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as ol
        ns = min(cpu_queries, n_queries)
        sel = np.linspace(0, n_queries - 1, ns).astype(np.int64)
        qs = np.ascontiguousarray(q[sel]); qn = np.ascontiguousarray(nrm[sel])
        idx = np.zeros((ns, k), np.int32); d2 = np.zeros((ns, k), np.float32); fnd = np.zeros(ns, np.int32); vis = C.c_int64(0)
        t0 = time.perf_counter()
        ol.oracle().orc_pmap_knn(a.ctypes.data, n_photons, qs.ctypes.data, ns, k, radius, idx.ctypes.data, d2.ctypes.data,
                                 fnd.ctypes.data, None, C.byref(vis))
        cpu_s = time.perf_counter() - t0
        ref_visits = vis.value / ns
        found_mean = float(fnd.mean())
        cpu = dict(value=ns / cpu_s, unit="lookups/s", cores=1, kind="port", sample=f"{ns} of the same queries, oracle LocatePhoton",
                   seconds=cpu_s)
        ref_rgb = None
        if ol.have_ref():
            # the reference itself (oracle/_ref/libref_photon.so: the unmodified PhotonMap of photon.cpp) on the same heap:
            # CalcIrradiance = LocatePhoton + the sum, photon.cpp:342-443, for the same queries, one core
            R = ol.ref_photon()
            R.refp_create(n_photons)
            R.refp_load(a.ctypes.data, n_photons)
            ref_rgb = np.zeros((ns, 3), np.float32)
            t0 = time.perf_counter()
            R.refp_irradiance(qs.ctypes.data, qn.ctypes.data, ns, radius, k, ref_rgb.ctypes.data)
            ref_s = time.perf_counter() - t0
            R.refp_create(1)                       # gives the reference's copy of the map back
            cpu = dict(value=ns / ref_s, unit="lookups/s", cores=1, kind="reference",
                       sample=f"{ns} of the same queries on the same heap, the reference's own CalcIrradiance "
                              "(LocatePhoton + sum, oracle/_ref/libref_photon.so)", seconds=ref_s,
                       port_lookups_per_s=ns / cpu_s)
        # parity sample: the oracle's CalcIrradiance for those queries against what the timed kernel wrote for them, and the
        # strict kernel's index lists against the oracle's
        orgb = np.zeros((ns, 3), np.float32)
        ol.oracle().orc_pmap_irradiance(a.ctypes.data, n_photons, qs.ctypes.data, qn.ctypes.data, ns, k, radius, orgb.ctypes.data)
        grgb = out.cpu().numpy()[sel]
        gi, gd, gf, gr = scene_handle.photon_knn(abi.RT_PMAP_GLOBAL, qs[:500], k, radius)
        parity = dict(queries=ns, irradiance_within_1e5_rel=bool(np.allclose(grgb, orgb, rtol=1e-5, atol=1e-12)),
                      max_rel_err=float((np.abs(grgb - orgb) / np.maximum(np.abs(orgb), 1e-30)).max()),
                      index_lists_bit_exact=bool(np.array_equal(gi, idx[:500]) and np.array_equal(gf, fnd[:500]) and gd.tobytes() == d2[:500].tobytes()))
        if ref_rgb is not None:               # and against the reference's own answers
            parity["irradiance_within_1e5_rel_of_reference"] = bool(np.allclose(grgb, ref_rgb, rtol=1e-5, atol=1e-12))
            parity["oracle_equals_reference_bits"] = bool(orgb.tobytes() == ref_rgb.tobytes())
        parity["parity_sample_ok"] = bool(parity["irradiance_within_1e5_rel"] and parity["index_lists_bit_exact"] and
                                          parity.get("irradiance_within_1e5_rel_of_reference", True))
    if ks[6] > 0:          # the wave-shared kernels ran
        visits = float(ks[0] + ks[12]) / n_queries
        detail = dict(waves=int(ks[6]), phase1_fetches_per_lookup=float(ks[0]) / n_queries,
                      walk_nodes_per_wave=float(ks[1] + ks[3]) / float(ks[6]), inserted_per_wave=float(ks[2]) / float(ks[6]),
                      strict_redo_queries=int(ks[5]), waves_gave_up=int(ks[11]), longest_wave_ms=float(int(ks[8]) >> 24) * 1e-5)
        kernel = "k_irr_main"
    else:
        visits, detail, kernel = (ref_visits or 0.0), dict(), "k_photon_knn"
    bytes_per = 24 + 16 * visits + 12 * found_mean + 12
    rf = make_roofline({10_000_000: "knn_10m", 200_000_000: "stress_knn_200m"}.get(n_photons, f"knn_{n_photons}"), kernel, ms, n_queries, bytes_per * n_queries,
                       visits_per_lookup=visits, bytes_per_lookup=bytes_per, reference_visits_per_lookup=ref_visits,
                       time_is="whole step (query sort + k_irr_main + strict redo), HIP events", **detail)
    return dict(metric="kNN radiance lookups/s", value=n_queries / (ms * 1e-3), unit="lookups/s",
                config=dict(workload="synthetic photon map on the Final scene's box walls", photons=n_photons,
                            queries=n_queries, k=k, radius=radius),
                ms_per_step=ms, map_build_s=build_s, roofline=rf, cpu_baseline=cpu, parity=parity)


def mesh_secondary(torch, local_rank, steps=5, nu=1000, nv=500, tag="mesh_1m", cpu=None):
    """BASELINE configuration 3's scene: the synthetic 1 M-triangle mesh (raytracer_amd/synth.py) at 3840x2160,
    primary + shadow rays."""
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
    ids = torch.zeros(BASE_RES[0] * BASE_RES[1], dtype=torch.int32, device="cuda")
    st = abi.rt_stats()
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    tot = dict(ms=0.0, closest=0.0, anyh=0.0, rays=0, prim=0, shadow=0, any_launches=0)
    for i in range(steps + 1):
        abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(params), frame.data_ptr(), ids.data_ptr(), C.byref(st), sp))
        if i:
            tot["ms"] += st.kernel_ms; tot["closest"] += st.trace_closest_ms; tot["anyh"] += st.trace_any_ms
            tot["rays"] += st.rays_primary + st.rays_shadow; tot["prim"] += st.rays_primary
            tot["shadow"] += st.rays_shadow; tot["any_launches"] += st.trace_any_launches
    stw = render_work(gs, cam, params, sp, torch)
    avg_ms = tot["closest"] / steps
    fused = tot["any_launches"] == 0 and tot["shadow"] > 0          # the whole frame is one kernel (k_primary_only)
    alg_c, wave_c = traversal_bytes(stw.closest_work, stw.rays_primary)
    alg_a, wave_a = traversal_bytes(stw.any_work, stw.rays_shadow, out_bytes=1)
    kernel = "k_primary_only" if fused else "k_trace_closest_bvh"
    rf = make_roofline(tag, kernel, avg_ms, (tot["prim"] + (tot["shadow"] if fused else 0)) / steps,
                       alg_c + (alg_a if fused else 0.0), wave_c + (wave_a if fused else 0.0),
                       box_tests_per_ray=stw.closest_work[0] / max(1, stw.rays_primary),
                       tri_records_per_ray=stw.closest_work[1] / max(1, stw.rays_primary),
                       shadow_box_tests_per_ray=stw.any_work[0] / max(1, stw.rays_shadow),
                       shadow_tri_records_per_ray=stw.any_work[1] / max(1, stw.rays_shadow))
    # parity sample: primary hit ids of a strided pixel sample of THIS frame against the oracle
    parity = None
    try:
        o = ol.OracleScene(sd)
        ocam = ol.oracle_camera(sd.eye, sd.view, sd.fov, *BASE_RES)
        npx = BASE_RES[0] * BASE_RES[1]
        pix = np.arange(17, npx, 211, dtype=np.int64)                 # 39 k pixels, every row and column phase
        rays = ol.oracle_primary_rays(ocam, (pix % BASE_RES[0]).astype(np.int64), (pix // BASE_RES[0]).astype(np.int64))
        t0 = time.perf_counter()
        want = o.trace_closest(rays)["prim"]
        port_s = time.perf_counter() - t0
        got = ids.cpu().numpy()[pix]
        o.close()
        if cpu is None:          # no reference-binary leg for this scene (stress_mesh_10m: 1.5 GB of scene text): the oracle port, timed
            cpu = dict(value=pix.size / port_s / 1e6, unit="Mrays/s", cores=1, kind="port", seconds=port_s,
                       sample=f"{pix.size} primary rays of the same frame (every 211th pixel), oracle closest-hit traversal on one core, "
                              "KD build not counted")
        parity = dict(pixels=int(pix.size), hit=int((want >= 0).sum()), mismatches=int((got != want).sum()),
                      parity_sample_ok=bool(np.array_equal(got, want)))
    except Exception as e:
        parity = dict(error=f"{type(e).__name__}: {e}", parity_sample_ok=False)
    gs.close()
    return dict(metric="Mrays/s (KD-tree traversal, primary + shadow rays)", value=tot["rays"] / (tot["ms"] * 1e-3) / 1e6,
                unit="Mrays/s", ms_per_step=tot["ms"] / steps,
                config=dict(workload=f"synthetic displaced-sphere mesh, {sd.nprim} triangles + floor, {BASE_RES[0]}x{BASE_RES[1]}, "
                                     "primary + shadow rays", kd_nodes=int(inf.flat_nodes), leaf_records=int(inf.leaf_refs),
                            scene_device_mb=inf.device_bytes / 1e6, scene_gen_s=gen_s, kd_build_upload_s=build_s),
                roofline=rf, cpu_baseline=cpu, parity=parity)


DRT_KW = dict(lens_radius=0.01, focal_distance=1.3, light_radius=0.5, glossy=0.05, seed=3)


def drt_analysis(torch, gs, sd, cam, p, sp, with_cpu=True):
    """BASELINE configuration 3's frame, instrumented (after the timed region): ray counts and the frame's own kernel time
    from rt_stats; the per-kernel HIP-event times of the roofline from one more frame with every kernel on one stream; the
    work counters from a frame with rt_params.count_work; and, with `with_cpu`, the CPU leg + parity sample: the oracle's
    restatement of the same samplers on a small frame against the GPU on that frame.  The samplers are build-defined (the
    reference has none), so this leg is `kind: "port"` by necessity."""
    from raytracer_amd import abi, host
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    res = (cam.resx, cam.resy)
    frame = torch.zeros(res[0] * res[1] * 3, dtype=torch.uint8, device="cuda")
    st = abi.rt_stats()
    for _ in range(2):
        abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(p), frame.data_ptr(), None, C.byref(st), sp))
    rays = st.rays_primary + st.rays_shadow + st.rays_reflect + st.rays_refract
    frame_ms = st.kernel_ms
    counts = dict(primary=st.rays_primary, shadow=st.rays_shadow, reflect=st.rays_reflect, refract=st.rays_refract)
    passes = st.passes
    # the per-kernel times of the roofline come from one more frame with every kernel on one stream: in the timed frame the
    # shadow half of a pass shares the chip with the next pass's closest-hit kernel, and two kernels side by side time each other
    abi.check(gs.L.rt_scene_set_option(gs.h, abi.RT_OPT_RENDER_OVERLAP, 0))
    abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(p), frame.data_ptr(), None, C.byref(st), sp))
    one_stream_ms = st.kernel_ms
    abi.check(gs.L.rt_scene_set_option(gs.h, abi.RT_OPT_RENDER_OVERLAP, 1))
    stw = render_work(gs, cam, p, sp, torch)
    # the frame's traversal time by kernel family; the roofline goes to the largest
    # (the per-lane pass kernels are the single-step ones unless RT_LANE_STEP=0 asks for the while-while walk)
    lane_k = ("k_trace_closest_bvh_lanes", "k_trace_any_bvh_lanes") if os.environ.get("RT_LANE_STEP", "1") == "0" else \
             ("k_trace_closest_bvh_step", "k_trace_any_bvh_step")
    fam = {
        lane_k[0]: (st.lane_closest_ms, st.lane_closest_launches, st.lane_closest_rays, stw.lane_closest_work, 20),
        lane_k[1]: (st.lane_any_ms, st.lane_any_launches, st.lane_any_rays, stw.lane_any_work, 1),
        "k_trace_closest_bvh": (st.trace_closest_ms - st.lane_closest_ms, st.trace_closest_launches - st.lane_closest_launches,
                                st.rays_primary + st.rays_reflect + st.rays_refract - st.lane_closest_rays,
                                [a - b for a, b in zip(stw.closest_work, stw.lane_closest_work)], 20),
        "k_trace_any_bvh": (st.trace_any_ms - st.lane_any_ms, st.trace_any_launches - st.lane_any_launches,
                            st.rays_shadow - st.lane_any_rays, [a - b for a, b in zip(stw.any_work, stw.lane_any_work)], 1),
    }
    if st.path_launches > 0:
        # the persistent path kernel (RT_OPT_PATH_DEPTH): closest-hit and shadow walks of every deeper recursion level in one
        # launch; per ray 32 B in + 32 B x box tests + 36 B x triangle records + the result (20 B closest, 1 B shadow)
        pw = stw.path_work
        fam["k_paths_bvh_lanes"] = (st.path_ms, st.path_launches, st.path_closest_rays + st.path_shadow_rays,
                                    [pw[0] + pw[2], pw[1] + pw[3], 0, 0],
                                    (20.0 * st.path_closest_rays + 1.0 * st.path_shadow_rays) / max(1, st.path_closest_rays + st.path_shadow_rays))
    kernel = max(fam, key=lambda k: fam[k][0])
    ms, launches, krays, work, ob = fam[kernel]
    launches = max(1, launches)
    alg, wave = traversal_bytes(work, krays, out_bytes=ob)
    per_lane = kernel.endswith("_lanes") or kernel.endswith("_step")
    rf = make_roofline("drt_1m", kernel, ms / launches, krays / launches, alg / launches,
                       (alg if per_lane else wave) / launches,     # per-lane kernels: every lane fetches its own records
                       launches_per_frame=launches, ms_per_frame_in_this_kernel=ms, frame_ms_on_one_stream=one_stream_ms,
                       time_is="HIP events around the kernel's launches in a frame with every kernel on one stream "
                               "(RT_OPT_RENDER_OVERLAP 0); `value` is the two-stream frame",
                       box_tests_per_ray=work[0] / max(1, krays), tri_records_per_ray=work[1] / max(1, krays),
                       traversal_ms_by_kernel={k: v[0] for k, v in fam.items()},
                       traversal_launches_by_kernel={k: int(v[1]) for k, v in fam.items()},
                       traversal_rays_by_kernel={k: int(v[2]) for k, v in fam.items()})
    out = dict(rays=rays, frame_ms=frame_ms, counts=counts, passes=passes, trace_closest_ms=st.trace_closest_ms,
               trace_any_ms=st.trace_any_ms, roofline=rf, cpu_baseline=None, parity=None)
    if with_cpu:
        o = ol.OracleScene(sd)
        cres, cspp = (128, 72), 8
        pc = host.default_params("hw4", spp=cspp, **DRT_KW)
        t0 = time.perf_counter()
        orgb, oids, _ = o.render_whitted(ol.oracle_camera(sd.eye, sd.view, sd.fov, *cres), pc)
        cpu_s = time.perf_counter() - t0
        crays = o.counters().rays_traced
        o.close()
        grgb, gids, _ = gs.render(host.camera_for(sd, cres), pc, want_ids=True)
        d = np.abs(grgb.astype(int) - orgb.reshape(grgb.shape).astype(int)).max(-1)
        out["parity"] = dict(frame=list(cres), spp=cspp, hit_ids_equal=bool(np.array_equal(gids, oids)),
                             pixels_off_by_more_than_1lsb=int((d > 1).sum()),
                             parity_sample_ok=bool(np.array_equal(gids, oids) and (d > 1).sum() <= 0.005 * cres[0] * cres[1]),
                             note="parity unpinned: the samplers are build-defined, the checker is the oracle's restatement of them")
        out["cpu_baseline"] = dict(value=crays / cpu_s / 1e6, unit="Mrays/s", cores=1, kind="port",
                                   sample=f"the same scene and samplers at {cres[0]}x{cres[1]}, {cspp} spp, oracle port "
                                          "(the reference has no such samplers, so its binary cannot run this workload)",
                                   seconds=cpu_s, rays=int(crays))
    return out


def drt_workload(sd, res, spp):
    return (f"BASELINE configs[2]: synthetic mesh, {sd.nprim} triangles, {res[0]}x{res[1]}, {spp} spp, lens 0.01 / focus 1.3, "
            "light ball 0.5, glossy 0.05, reflect 0.3 (sphere) / 0.5 (floor), depth <= 10")


def monkey_secondary(torch, local_rank, steps=20, warmup=5, cpu=None):
    """BASELINE configs[1] (the headline of rounds 1-2): the bundled monkey scene at 3840x2160, primary + shadow rays only
    (materials' reflect / refract set to 0); the frame's hit-id image against the reference's own 4K digest."""
    from raytracer_amd import abi, host
    sd = bench_scene()
    gs = host.Scene(sd, device=local_rank)
    resx, resy = BASE_RES
    cam = host.camera_for(sd, BASE_RES)
    params = host.default_params("hw4", primary_only=1)
    frame = torch.zeros(resx * resy * 3, dtype=torch.uint8, device="cuda")
    hit_ids = torch.zeros(resx * resy, dtype=torch.int32, device="cuda")
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    st = abi.rt_stats()
    render = lambda ids, stats: abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(params), frame.data_ptr(),
                                                             hit_ids.data_ptr() if ids else None, C.byref(st) if stats else None, sp))
    for _ in range(warmup):
        render(False, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):           # no rt_stats: the frames queue on the stream back to back
        render(False, False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tot = dict(rays=0, closest_ms=0.0, launches=0, prim=0, shadow=0, any_launches=0)
    nstat = max(1, min(steps, 20))
    for _ in range(nstat):
        render(True, True)
        tot["rays"] += st.rays_primary + st.rays_shadow; tot["closest_ms"] += st.trace_closest_ms
        tot["launches"] += st.trace_closest_launches; tot["prim"] += st.rays_primary; tot["shadow"] += st.rays_shadow
        tot["any_launches"] += st.trace_any_launches
    rays_per_step = tot["rays"] / nstat
    try:
        g = np.load(os.path.join(ROOT, "tests", "golden", "monkey4k_hits_digest.npz"))
        ids = hit_ids.cpu().numpy()
        dig = hashlib.sha256(np.ascontiguousarray(ids.reshape(resy, resx).T.astype("<i4")).tobytes()).hexdigest()   # the reference's x-major order
        want_dig = str(g["sha256_ids_xy"].item())
        parity = dict(frame=[resx, resy], hit_id_sha256=dig, reference_sha256=want_dig, parity_sample_ok=bool(dig == want_dig))
    except Exception as e:
        parity = dict(error=f"{type(e).__name__}: {e}", parity_sample_ok=False)
    stw = render_work(gs, cam, params, sp, torch)
    launches = max(1, tot["launches"])
    fused = tot["any_launches"] == 0 and tot["shadow"] > 0       # the whole frame is ONE kernel (k_primary_only)
    alg_c, wave_c = traversal_bytes(stw.closest_work, stw.rays_primary)
    alg_a, wave_a = traversal_bytes(stw.any_work, stw.rays_shadow, out_bytes=1)
    kernel = "k_primary_only" if fused else "k_trace_closest_bvh"
    rf = make_roofline(
        "monkey_4k", kernel, tot["closest_ms"] / launches, (tot["prim"] + (tot["shadow"] if fused else 0)) / launches,
        alg_c + (alg_a if fused else 0.0), wave_c + (wave_a if fused else 0.0),
        box_tests_per_ray=stw.closest_work[0] / max(1, stw.rays_primary), tri_records_per_ray=stw.closest_work[1] / max(1, stw.rays_primary),
        exact_tri_tests_per_ray=stw.closest_work[2] / max(1, stw.rays_primary),
        shadow_box_tests_per_ray=stw.any_work[0] / max(1, stw.rays_shadow), shadow_tri_records_per_ray=stw.any_work[1] / max(1, stw.rays_shadow),
        note=("one fused kernel traces the primary AND the shadow rays (generate, closest hit, any hit, shade, pixel). " if fused else "") +
             "The scene (87 KB) lives in the scalar cache / L2: HBM sees the frame that is written and little else, so the "
             "kernel is priced against instruction issue; see DESIGN.md 4.1d / 5")
    gs.close()
    return dict(metric="Mrays/s (KD-tree traversal, primary + shadow rays)", value=rays_per_step * steps / dt / 1e6, unit="Mrays/s",
                ms_per_step=dt / steps * 1e3, steps=steps,
                config=dict(workload="BASELINE configs[1]: bundled monkey scene (970 triangles, 138 KD nodes), "
                                     f"{resx}x{resy}, primary + shadow rays only, reflect=refract=0", rays_per_step=rays_per_step),
                roofline=rf, cpu_baseline=cpu, parity=parity)


def photon_frame_secondary(torch, local_rank, n_global=1_000_000, n_caustic=100_000, res=(192, 192), tag="photon_frame"):
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
    # an instrumented frame first (ray counts, the per-call times of the radiance estimate: rt_stats keeps every call on one
    # stream), then the TIMED frames without rt_stats -- as a caller renders: the caustic estimate's strict redo runs on the
    # scene's second stream beside the gather rays (render_photon.cpp)
    abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(p), frame.data_ptr(), None, C.byref(st), sp))
    nframes = 1 if res[0] * res[1] >= 1 << 20 else 4
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nframes):
        abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(p), frame.data_ptr(), None, None, sp))
    torch.cuda.synchronize()
    frame_ms = (time.perf_counter() - t0) / nframes * 1e3
    gs.knn_stats(True)                       # one more frame, untimed, with the kNN kernels' work counters on
    st2 = abi.rt_stats()
    abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(p), frame.data_ptr(), None, C.byref(st2), sp))
    ks = gs.knn_stats(False)
    rays = st.rays_primary + st.rays_shadow + st.rays_reflect + st.rays_refract + st.rays_gather
    # roofline: the radiance estimate is the frame (k_irr_main ~80 % of it); algorithmic bytes from its work counters
    visits = float(ks[0] + ks[12]) / max(1, st.knn_lookups)
    bytes_per = 24 + 16 * visits + 12 * float(p.knn_k) + 12
    launches = max(1, st.knn_launches)
    # (the profiled run = this function: one instrumented frame, the timed ones, the counted one -- and the 10x10 parity sample,
    # whose radiance-estimate launches are as many and carry next to nothing)
    full_frames = nframes + 2
    rf = make_roofline(tag, "k_irr_main", st.knn_ms / launches, st.knn_lookups / launches, bytes_per * st.knn_lookups / launches,
                       counter_scale=(full_frames + 1) / full_frames, visits_per_lookup=visits, bytes_per_lookup=bytes_per, launches_per_frame=launches,
                       ms_per_frame_in_the_estimate=st.knn_ms,
                       time_is="all radiance-estimate calls of the frame (query sort + k_irr_main + strict redo), HIP events",
                       waves=int(ks[6]), strict_redo_queries=int(ks[5]), waves_gave_up=int(ks[11]),
                       walk_nodes_per_wave=float(ks[1] + ks[3]) / max(1.0, float(ks[6])),
                       inserted_per_wave=float(ks[2]) / max(1.0, float(ks[6])))
    # CPU baseline + parity sample: the oracle's restatement of the same frame at 10x10 pixels, same maps, and the GPU on it
    o = ol.OracleScene(sd)
    cres = (10, 10)
    t0 = time.perf_counter()
    orgb, oids, _ = o.render_photon(ol.oracle_camera(sd.eye, sd.view, sd.fov, *cres), p, maps[0]["heap"], maps[1]["heap"])
    cpu_s = time.perf_counter() - t0
    c = o.counters()
    crays = c.rays_primary + c.rays_shadow + c.rays_reflect + c.rays_refract + c.rays_gather
    o.close()
    grgb, gids, _ = gs.render(host.camera_for(sd, cres), p, want_ids=True)
    d = np.abs(grgb.astype(int) - np.asarray(orgb).reshape(grgb.shape).astype(int)).max(-1)
    parity = dict(frame=list(cres), hit_ids_equal=bool(np.array_equal(gids, np.asarray(oids).reshape(gids.shape))),
                  pixels_off_by_more_than_1lsb=int((d > 1).sum()),
                  parity_sample_ok=bool(np.array_equal(gids, np.asarray(oids).reshape(gids.shape)) and (d > 1).sum() <= 1))
    gs.close()
    cpu_line = dict(value=crays / cpu_s / 1e6, unit="Mrays/s", lookups_per_s=c.knn_lookups / cpu_s, cores=1,
                    kind="port", sample=f"the same frame at {cres[0]}x{cres[1]}, same photon maps, oracle port", seconds=cpu_s)
    if ol.have_ref() and n_global <= 20_000_000:
        # the reference itself (oracle/_ref/libref_final.so: the unmodified SimpleMeshRayTrace(pmap, cpmap, ...) of
        # final.cpp:785-983 with TraceIrradiance / CalcIrradiance / IsInShadow below it) on the same scene and the same two
        # heaps, the primary rays of an 8x8 frame, 1 tap; one core.  Its gather directions come from its own serial engine, so
        # its rays are counted by the oracle port on that frame and its radiance is not a parity sample.
        try:
            R = ol.ref_final_build_scene(sd)
            cs = np.array([0.2, -0.3, 0.45, 0.2], np.float32); R.reff_add_caustic_sphere(cs.ctypes.data)
            R.reff_maps_create(n_global, n_caustic)
            R.reff_map_load(0, maps[0]["heap"].ctypes.data, n_global); R.reff_map_load(1, maps[1]["heap"].ctypes.data, n_caustic)
            rres = (8, 8)
            rcam = ol.oracle_camera(sd.eye, sd.view, sd.fov, *rres)
            xs, ys = np.meshgrid(np.arange(rres[0]), np.arange(rres[1]), indexing="ij")
            eye = ol.oracle_primary_rays(rcam, xs.ravel(), ys.ravel())
            o3 = np.ascontiguousarray(eye[:, :3]); d3 = np.ascontiguousarray(eye[:, 3:])
            rrgb = np.zeros((eye.shape[0], 3), np.float32); ha = np.zeros(eye.shape[0], np.uint8)
            t0 = time.perf_counter()
            R.reff_eye_trace(o3.ctypes.data, d3.ctypes.data, 0, 2.4, eye.shape[0], rrgb.ctypes.data, ha.ctypes.data)
            ref_s = time.perf_counter() - t0
            R.reff_maps_create(1, 1)
            o2 = ol.OracleScene(sd)
            o2.render_photon(rcam, p, maps[0]["heap"], maps[1]["heap"])
            c2 = o2.counters()
            rrays = c2.rays_primary + c2.rays_shadow + c2.rays_reflect + c2.rays_refract + c2.rays_gather
            o2.close()
            cpu_line = dict(value=rrays / ref_s / 1e6, unit="Mrays/s", lookups_per_s=c2.knn_lookups / ref_s, cores=1, kind="reference",
                            sample=f"the same scene and photon maps, the primary rays of an {rres[0]}x{rres[1]} frame (1 tap) through the "
                                   "reference's own SimpleMeshRayTrace(pmap, cpmap) (oracle/_ref/libref_final.so); rays and lookups "
                                   "counted by the oracle port on that frame", seconds=ref_s,
                            port_mrays_per_s=crays / cpu_s / 1e6, port_lookups_per_s=c.knn_lookups / cpu_s)
        except Exception as e:
            log(f"reference leg of {tag} failed: {type(e).__name__}: {e}")
    return dict(metric="photon-mapping frame: Mrays/s and kNN radiance lookups/s", value=rays / (frame_ms * 1e-3) / 1e6,
                unit="Mrays/s", lookups_per_s=st.knn_lookups / (frame_ms * 1e-3), ms_per_frame=frame_ms,
                ms_per_frame_on_one_stream=st.kernel_ms, timed_frames=nframes,
                config=dict(workload=f"bundled Final scene, {res[0]}x{res[1]}, 1 AA tap, 50 gather rays, 20x20 light samples, k=100",
                            photons_global=n_global, photons_caustic=n_caustic, emissions_global=maps[0]["emissions"],
                            emissions_caustic=maps[1]["emissions"], maps_emit_balance_upload_s=maps_s),
                rays=dict(primary=st.rays_primary, shadow=st.rays_shadow, reflect=st.rays_reflect, refract=st.rays_refract,
                          gather=st.rays_gather), knn_lookups=st.knn_lookups, roofline=rf,
                cpu_baseline=cpu_line, parity=parity)


# ---------------------------------------------------------------------------------------------------------------------
# N > 1: the lines of the configurations that are meant to scale (strong scaling), after the weak-scaling headline

class Coll:
    """torch.distributed with device tensors; with the gloo backend (rehearsal of the N > 1 path on one GPU) tensors are
    staged through the host."""
    def __init__(self, dist, torch, backend):
        self.dist, self.torch, self.host = dist, torch, backend == "gloo"

    def all_ok(self, ok, dev):
        t = self.torch.tensor([1 if ok else 0], dtype=self.torch.int64, device="cpu" if self.host else dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(t.item())

    def max_float(self, x, dev):
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cpu" if self.host else dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_float(self, x, dev):
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cpu" if self.host else dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def gather0(self, send, recv_list, rank):
        if not self.host:
            self.dist.gather(send, recv_list if rank == 0 else None, dst=0)
            return
        s = send.cpu()
        r = [self.torch.empty_like(s) for _ in recv_list] if rank == 0 else None
        self.dist.gather(s, r, dst=0)
        if rank == 0:
            for a, b in zip(recv_list, r):
                a.copy_(b)


def drt_strong(torch, coll, rank, n, local_rank, res=(1920, 1080), spp=64, steps=2, gs=None, sd=None):
    """BASELINE configuration 3 over N GPUs, STRONG scaling: the fixed 1080p x 64 spp frame, 64x64 tiles round-robin, each
    rank's pixels written packed (rt_params.packed_output) and gathered to rank 0; time = slowest rank incl. the gather."""
    from raytracer_amd import abi, host, synth
    dev = torch.device("cuda", local_rank)
    cam = None
    ok = True
    own = gs is None                 # the headline's scene is this configuration's scene: reuse it
    try:
        if own:
            sd = synth.mesh_scene(1000, 500, reflect=0.3)
            gs = host.Scene(sd, device=local_rank)
        cam = host.camera_for(sd, res)
        plist = [host.pixel_list(cam, host.default_params("hw4", tile_size=TILE, tile_rank=r, tile_nranks=n)) for r in range(n)]
        counts = [int(l.size) for l in plist]
        maxc = max(counts)
        p = host.default_params("hw4", spp=spp, tile_size=TILE, tile_rank=rank, tile_nranks=n, packed_output=1, **DRT_KW)
        send = torch.zeros((maxc, 3), dtype=torch.uint8, device=dev)
        recv = [torch.zeros((maxc, 3), dtype=torch.uint8, device=dev) for _ in range(n)] if rank == 0 else []
        sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        st = abi.rt_stats()
    except Exception as e:
        log(f"rank {rank}: drt_strong set-up failed: {type(e).__name__}: {e}")
        ok = False
    if not coll.all_ok(ok, dev):
        return dict(error="set-up failed on a rank")
    rays, t_total = 0, 0.0
    for i in range(steps + 1):
        torch.cuda.synchronize(); coll.dist.barrier()
        t0 = time.perf_counter()
        rc = gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(p), send.data_ptr(), None, C.byref(st), sp)
        torch.cuda.synchronize()
        if not coll.all_ok(rc == 0, dev):
            return dict(error="render failed on a rank")
        coll.gather0(send, recv, rank)
        torch.cuda.synchronize(); coll.dist.barrier()
        dt = time.perf_counter() - t0
        if i:
            t_total += coll.max_float(dt, dev)
            rays += st.rays_primary + st.rays_shadow + st.rays_reflect + st.rays_refract
    total_rays = coll.sum_float(float(rays), dev)
    out = None
    if rank == 0:
        # the assembled frame against the same frame rendered whole on rank 0
        final = torch.zeros((res[0] * res[1], 3), dtype=torch.uint8, device=dev)
        for r in range(n):
            pl = torch.from_numpy(plist[r].astype(np.int64)).to(dev)
            okm = pl >= 0
            final[pl[okm]] = recv[r][: counts[r]][okm]
        whole = torch.zeros(res[0] * res[1] * 3, dtype=torch.uint8, device=dev)
        pw = host.default_params("hw4", spp=spp, **DRT_KW)
        abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(pw), whole.data_ptr(), None, None, sp))
        torch.cuda.synchronize()
        out = dict(metric="Mrays/s (distributed ray tracing, all ray types)", value=total_rays / t_total / 1e6, unit="Mrays/s",
                   n_gpus=n, scaling="strong", ms_per_frame=t_total / steps * 1e3, steps=steps,
                   config=dict(workload=f"BASELINE configs[2]: 1 M-triangle mesh, {res[0]}x{res[1]}, {spp} spp, tiles {TILE}x{TILE} "
                                        f"round-robin over {n} GPUs, packed gather to rank 0",
                               assembled_frame_equals_one_gpu_frame=bool(torch.equal(final.view(-1), whole))))
    if own:
        gs.close()
    return out


def emission_sharded(torch, dist, coll, rank, n, local_rank, targets=(1_000_000, 100_000)):
    """BASELINE configuration 4/5's exchange step over N GPUs: emission sharded by emission index, RCCL all-gather of the
    photons, ONE balance per node and the map's device image broadcast (raytracer_amd/dist.py).  Checked: rank 0's heap
    equals the heap it builds alone, and every rank's map answers a fixed set of radiance queries with the same bits."""
    from raytracer_amd import abi, host, scene
    from raytracer_amd import dist as rdist
    dev = torch.device("cuda", local_rank)
    sd = scene.parse_scene_file(os.path.join(ROOT, "tests", "golden", "final_input.txt"), flavour="final")
    gs = host.Scene(sd, device=local_rank)
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    if coll.host:       # rehearsal over gloo: every rank builds (the broadcast of device tensors needs RCCL)
        maps = rdist.build_photon_maps(gs, torch, targets, seed=1, rank=rank, world=n, dist=_HostDist(dist, torch), every_rank_builds=True)
    else:
        maps = rdist.build_photon_maps(gs, torch, targets, seed=1, rank=rank, world=n, dist=dist)
    torch.cuda.synchronize(); dist.barrier()
    dt = coll.max_float(time.perf_counter() - t0, dev)
    # every rank answers the same queries on its copy of the global map
    rng = np.random.default_rng(5)
    q = rng.uniform(-0.45, 0.45, (4096, 3)).astype(np.float32); q[:, 2] += 0.5; q[:2048, 1] = -0.5
    nr = np.tile(np.float32([0, 1, 0]), (4096, 1))
    rgb = gs.photon_irradiance(abi.RT_PMAP_GLOBAL, q, nr, 100, 1.0)
    h = int.from_bytes(hashlib.sha256(rgb.tobytes()).digest()[:7], "little")
    hmax, hmin = coll.max_float(float(h), dev), -coll.max_float(-float(h), dev)
    out = None
    if rank == 0:
        gs1 = host.Scene(sd, device=local_rank)
        t1 = time.perf_counter()
        alone = rdist.build_photon_maps(gs1, torch, targets, seed=1)
        alone_s = time.perf_counter() - t1
        same = all(m["heap"].tobytes() == a["heap"].tobytes() for m, a in zip(maps, alone))
        gs1.close()
        out = dict(metric="photon maps: sharded emission + all-gather + one balance per node", value=sum(targets) / dt, unit="photons/s",
                   n_gpus=n, scaling="strong", seconds=dt, one_rank_alone_seconds=alone_s,
                   config=dict(workload=f"bundled Final scene, {targets[0]} global + {targets[1]} caustic photons",
                               allgather_bytes=int(sum(m.get("allgather_bytes", 0) for m in maps)),
                               image_bytes=int(sum(m.get("image_bytes", 0) for m in maps)),
                               heap_equals_one_rank_heap=bool(same), every_rank_answers_alike=bool(hmax == hmin and np.abs(rgb).max() > 0)))
    gs.close()
    return out


class _HostDist:
    """all_gather of device tensors through the host (gloo rehearsal only)"""
    def __init__(self, dist, torch):
        self.dist, self.torch = dist, torch

    def all_gather(self, outs, t):
        s = t.cpu()
        r = [self.torch.empty_like(s) for _ in outs]
        self.dist.all_gather(r, s)
        for a, b in zip(outs, r):
            a.copy_(b)


# ---------------------------------------------------------------------------------------------------------------------

EXIT_STALLED, EXIT_LINE_FAILED = 3, 4


def line_failed(line):
    """a secondary line that raised, or whose parity sample disagreed"""
    if not isinstance(line, dict):
        return True
    if "error" in line:
        return True
    par = line.get("parity")
    if isinstance(par, dict) and par.get("parity_sample_ok") is False:
        return True
    cfg = line.get("config", {})
    for k in ("assembled_frame_equals_one_gpu_frame", "heap_equals_one_rank_heap", "every_rank_answers_alike"):
        if cfg.get(k) is False:
            return True
    return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-knn", action="store_true", help="skip the secondary lines")
    ap.add_argument("--no-full-size", action="store_true", help="skip photon_frame_full, stress_mesh_10m, stress_knn_200m (quick runs)")
    ap.add_argument("--only-secondary", type=str, default="", help="comma list of secondary lines to run: monkey_4k,knn,mesh_1m,photon_frame,photon_frame_full,stress_mesh_10m,stress_knn_200m (N > 1: knn,drt_1m_strong,emission_sharded)")
    ap.add_argument("--knn-photons", type=int, default=10_000_000)
    ap.add_argument("--knn-queries", type=int, default=1920 * 1080)
    ap.add_argument("--cpu-sample", type=str, default="960x540")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: exercise the sharding / gather logic only (tests)")
    ap.add_argument("--backend", type=str, default="nccl", help="nccl (= RCCL; default) or gloo (rehearsal of the N > 1 path with every rank on one GPU)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n = world if world > 1 else 1
    # RT_BENCH_FORCE_MULTI=1 (tests/test_gpu_dist_rccl.py): the N > 1 code path -- process group, tiles, packed gather, all-reduces,
    # the strong-scaling lines -- with however many ranks there are, ONE included: the only way to run it over RCCL on a one-GPU box
    multi = world > 1 or os.environ.get("RT_BENCH_FORCE_MULTI") == "1"
    dry = args.dry_run
    rehearse = multi and args.backend == "gloo" and not dry
    only = [x for x in args.only_secondary.split(",") if x]
    full_size = ("photon_frame_full", "stress_mesh_10m", "stress_knn_200m")
    want = lambda name: (not args.no_knn) and ((name in only) if only else not (args.no_full_size and name in full_size))

    # ---- CPU legs that fork (reference binaries, process pool): before anything touches the GPU ----------------------
    cpu = {}
    if rank == 0 and (not multi) and not dry and not args.no_cpu_baseline:
        w, h = (int(x) for x in args.cpu_sample.split("x"))
        t0 = time.perf_counter()
        if want("monkey_4k"):
            msd = bench_scene()
            try:
                cpu["monkey_4k"] = cpu_baseline_monkey(msd, (w, h))
            except Exception as e:
                cpu["monkey_4k"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                cpu["monkey_4k"]["all_cores"] = cpu_port_all_cores(msd, (w, h))
            except Exception as e:      # a reported extra, never a reason to lose the line
                cpu["monkey_4k"]["all_cores"] = {"error": f"{type(e).__name__}: {e}"}
        if want("mesh_1m"):
            try:
                cpu["mesh_1m"] = cpu_baseline_mesh()
            except Exception as e:
                cpu["mesh_1m"] = {"error": f"{type(e).__name__}: {e}"}
        log(f"CPU legs (reference binary, all-cores port): {time.perf_counter() - t0:.1f} s")

    import torch
    import torch.distributed as dist

    if world != args.gpus and world > 1:
        log(f"warning: WORLD_SIZE {world} != --gpus {args.gpus}; using WORLD_SIZE")
    if rehearse:
        local_rank = 0
    dev = torch.device("cpu") if dry else torch.device("cuda", local_rank)
    if not dry:
        torch.cuda.set_device(dev)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo" if (dry or rehearse) else "nccl")
    coll = Coll(dist, torch, "gloo" if (dry or rehearse) else "nccl") if multi else None

    # Every rank of an N > 1 run sits in collectives from here on; if one of them dies or stalls the others would wait for
    # ever.  After RT_BENCH_WATCHDOG_S seconds (default 360) without the run finishing, every rank gives up: rank 0 prints the
    # line as it stands, marked, and the process ends with a NON-ZERO status -- a stalled run must never read as a success.
    out = None
    watchdog = None
    if multi:
        import threading

        def give_up():
            if rank == 0:
                line = dict(out) if out else {"metric": "Mrays/s (distributed ray tracing, all ray types)", "value": None, "n_gpus": n}
                line["error"] = f"the N = {n} run did not finish within its time limit: a rank stalled in a collective"
                print(dumps(line), flush=True)
            log(f"rank {rank}: watchdog: giving up, exit status {EXIT_STALLED}")
            os._exit(EXIT_STALLED)
        watchdog = threading.Timer(float(os.environ.get("RT_BENCH_WATCHDOG_S", "360")), give_up)
        watchdog.daemon = True
        watchdog.start()

    from raytracer_amd import abi, host
    if dry:
        sd = bench_scene()                  # only the camera is used
        resx, resy = 256 * WEAK_FACTORS.get(n, (n, 1))[0], 128 * WEAK_FACTORS.get(n, (n, 1))[1]
    else:
        from raytracer_amd import synth
        t0 = time.perf_counter()
        sd = synth.mesh_scene(1000, 500, reflect=0.3)
        resx, resy = frame_for(n)
    npix = resx * resy
    packed = multi                      # a rank's tiles leave the kernel as one contiguous buffer (rt_params.packed_output)
    cam = host.camera_for(sd, (resx, resy))
    scene_s = None
    if not dry:
        gs = host.Scene(sd, device=local_rank)
        scene_s = time.perf_counter() - t0
        params = host.default_params("hw4", spp=HEAD_SPP, tile_size=TILE if multi else 0, tile_rank=rank, tile_nranks=n,
                                     packed_output=1 if packed else 0, **DRT_KW)
        stream = torch.cuda.current_stream()
        sp = C.c_void_p(stream.cuda_stream)
    if packed:
        # pixel lists in the kernel's order (rt_render_pixel_list; -1 = a slot outside the frame)
        lists = [host.pixel_list(cam, host.default_params("hw4", tile_size=TILE, tile_rank=r, tile_nranks=n)) for r in range(n)]
        counts = [int(l.size) for l in lists]
        my_pix = None
    else:
        my_pix = torch.from_numpy(rank_pixel_index(resx, resy, rank, n)).to(dev)
        counts = [len(rank_pixel_index(resx, resy, r, n)) for r in range(n)] if multi else [npix]
    maxc = max(counts)
    frame = torch.zeros(npix * 3, dtype=torch.uint8, device=dev)
    sendbuf = [torch.zeros((maxc, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    recv = [[torch.zeros((maxc, 3), dtype=torch.uint8, device=dev) for _ in range(n)] for _ in range(2)] if (multi and rank == 0) else None
    final = torch.zeros((npix, 3), dtype=torch.uint8, device=dev) if rank == 0 else None
    if packed:
        all_pix = [torch.from_numpy(l.astype(np.int64)).to(dev) for l in lists] if rank == 0 else None
    else:
        all_pix = [torch.from_numpy(rank_pixel_index(resx, resy, r, n)).to(dev) for r in range(n)] if (multi and rank == 0) else None
    st = abi.rt_stats()
    tot = dict(rays=0)
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
            # timed steps ask for no rt_stats (no event bookkeeping, no counter read-back).  Every step renders the same
            # frame, so its ray counts are those of the instrumented frames that follow the timed region.
            abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(params), sendbuf[b].data_ptr() if packed else frame.data_ptr(),
                                         None, C.byref(st) if timed == "stats" else None, sp))
            if timed == "stats":
                tot["rays"] += st.rays_primary + st.rays_shadow + st.rays_reflect + st.rays_refract
        if multi:
            if pending[b] is not None:                # the buffer pair of two steps ago must have left
                pending[b].wait()
            if not packed:
                sendbuf[b][: counts[rank]] = frame.view(-1, 3)[my_pix]
            if rehearse:
                coll.gather0(sendbuf[b], recv[b] if rank == 0 else [], rank)
            else:
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
        if multi:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i, False)
    if multi:
        drain()
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, True)
    if multi:
        drain()
    sync()
    dt = time.perf_counter() - t0
    # the same frame again with rt_stats, for the ray counts
    nstat = 0
    if not dry:
        nstat = max(1, min(args.steps, 2))
        for i in range(nstat):
            step(i, "stats")
        if multi:
            drain()
        sync()
        tot["rays"] = tot["rays"] / nstat * args.steps
    if multi:
        cdev = "cpu" if (dry or rehearse) else dev
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        r = torch.tensor([float(tot["rays"])], dtype=torch.float64, device=cdev)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        total_rays = float(r.item())
    else:
        total_rays = float(tot["rays"])

    if dry:
        if rank == 0:
            if multi:
                ok = True
                for r in range(n):
                    pix = all_pix[r][all_pix[r] >= 0]
                    got = final[pix]
                    ok = ok and bool((got[:, 0] == (pix % 251).to(torch.uint8)).all()) and bool((got[:, 1] == r + 1).all()) \
                        and bool((got[:, 2] == (last[0] & 255)).all())
                ok = ok and sum(int((all_pix[r] >= 0).sum()) for r in range(n)) == npix      # every pixel has exactly one owner
            else:
                ok = True
            out = {"dry_run": True, "n_gpus": n, "frame": [resx, resy], "assembled_ok": ok, "pixels_per_rank": counts}
        if multi:
            stall = os.environ.get("RT_BENCH_TEST_STALL_RANK")       # tests only: this rank never reaches the last barrier
            if stall is not None and int(stall) == rank:
                time.sleep(3600)
            dist.barrier()
            watchdog.cancel()
            dist.destroy_process_group()
        if rank == 0:
            print(dumps(out))
        return 0

    failed = []
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = total_rays / dt / 1e6
        out = {"metric": "Mrays/s (distributed ray tracing: primary + shadow + reflection + refraction rays)", "value": value,
               "unit": "Mrays/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": drt_workload(sd, (resx, resy), HEAD_SPP), "frame": [resx, resy], "spp": HEAD_SPP,
                          "rays_per_step": total_rays / args.steps, "scene_generate_build_upload_s": scene_s,
                          "parallelism": f"tiles{TILE}x{TILE} round-robin over {n} GPU(s), scene replicated, gather to rank 0"}}
        if rehearse:
            out["config"]["rehearsal"] = "gloo backend, every rank on GPU 0, collectives staged through the host: NOT a measurement"
        if multi:
            # the frame rank 0 assembled from the ranks' last gather against the same frame rendered whole on rank 0
            pw = host.default_params("hw4", spp=HEAD_SPP, **DRT_KW)
            abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(pw), frame.data_ptr(), None, None, sp))
            torch.cuda.synchronize()
            same = bool(torch.equal(final.view(-1), frame))
            out["config"]["assembled_frame_equals_one_gpu_frame"] = same
            if not same:
                failed.append("headline: assembled frame differs from the one-GPU frame")
        # roofline of the dominant kernel (counters of the committed profile, time live), CPU leg, parity sample: on this
        # rank's share of the frame (N = 1: the whole frame)
        try:
            pa = host.default_params("hw4", spp=HEAD_SPP, tile_size=TILE if multi else 0, tile_rank=rank, tile_nranks=n, **DRT_KW)
            an = drt_analysis(torch, gs, sd, cam, pa, sp, with_cpu=not args.no_cpu_baseline)
            out["roofline"] = an["roofline"]
            if an["cpu_baseline"]:
                out["cpu_baseline"] = an["cpu_baseline"]
            if an["parity"]:
                out["parity"] = an["parity"]
                if not an["parity"]["parity_sample_ok"]:
                    failed.append("headline: parity sample")
            out["config"].update(rays=an["counts"], passes=an["passes"], frame_kernel_ms=an["frame_ms"])
        except Exception as e:
            out["roofline"] = {"error": f"{type(e).__name__}: {e}"}
            failed.append("headline: instrumented frames")
        if (not multi) and not args.no_knn:
            sec = {}
            for name, fn in (("monkey_4k", lambda: monkey_secondary(torch, local_rank, cpu=cpu.get("monkey_4k"))),
                             ("mesh_1m", lambda: mesh_secondary(torch, local_rank, cpu=cpu.get("mesh_1m"))),
                             ("knn", lambda: knn_secondary(gs, torch, args.knn_photons, args.knn_queries)),
                             ("photon_frame", lambda: photon_frame_secondary(torch, local_rank)),
                             # BASELINE configuration 4 at its stated size: 10 M + 1 M photons from GPU emission, 1080p
                             ("photon_frame_full", lambda: photon_frame_secondary(torch, local_rank, 10_000_000, 1_000_000, (1920, 1080), tag="photon_frame_full")),
                             # one GPU's share of BASELINE configuration 5 (scene and photon map are replicated per GPU)
                             ("stress_mesh_10m", lambda: mesh_secondary(torch, local_rank, steps=3, nu=3162, nv=1581, tag="stress_mesh_10m")),
                             ("stress_knn_200m", lambda: knn_secondary(gs, torch, 200_000_000, args.knn_queries, steps=2, cpu_queries=500))):
                if not want(name):
                    continue
                try:
                    t0 = time.perf_counter()
                    sec[name] = fn()
                    log(f"{name}: {time.perf_counter() - t0:.1f} s")
                except Exception as e:           # a secondary line never takes the headline down with it, but it fails the run
                    sec[name] = {"error": f"{type(e).__name__}: {e}"}
                    log(f"{name} FAILED: {type(e).__name__}: {e}")
                if line_failed(sec[name]):
                    failed.append(name)
            out["secondary"] = sec
    if multi and not args.no_knn:
        sec = {}
        # the metric's second half at N GPUs: every rank answers its own share of the radiance lookups (the queries
        # follow their pixels, the photon map is replicated, no exchange: SURVEY 8e), weak scaling like the frame
        if not only or "knn" in only:
            try:
                r = knn_secondary(gs, torch, args.knn_photons, args.knn_queries, steps=3, with_cpu=False)
                ms = r["ms_per_step"]
            except Exception as e:
                log(f"rank {rank}: kNN line failed: {type(e).__name__}: {e}")
                r, ms = None, float("inf")
            ms = coll.max_float(ms, dev)
            if rank == 0:
                if r is not None and ms != float("inf"):
                    r["value"] = n * args.knn_queries / (ms * 1e-3)
                    r["ms_per_step"] = ms; r["n_gpus"] = n; r["scaling"] = "weak"
                    r["config"]["queries"] = n * args.knn_queries
                    r["note"] = "aggregate over the ranks: each answers its own queries on its replica of the map; time = slowest rank"
                    sec["knn"] = r
                else:
                    sec["knn"] = {"error": "a rank failed"}
        # the configurations that are meant to scale, strong scaling (every rank runs them: collectives inside)
        for name, fn in (("drt_1m_strong", lambda: drt_strong(torch, coll, rank, n, local_rank, gs=gs, sd=sd)),
                         ("emission_sharded", lambda: emission_sharded(torch, dist, coll, rank, n, local_rank))):
            if only and name not in only:
                continue
            try:
                res = fn()
            except Exception as e:
                log(f"rank {rank}: {name} failed: {type(e).__name__}: {e}")
                res = {"error": f"{type(e).__name__}: {e}"}
            if rank == 0:
                sec[name] = res
        if rank == 0:
            out["secondary"] = sec
            failed += [k for k, v in sec.items() if line_failed(v)]
    if multi:
        dist.barrier()
        watchdog.cancel()
        dist.destroy_process_group()
    rc = 0
    if rank == 0:
        if failed:
            out["failed_lines"] = failed
            rc = EXIT_LINE_FAILED
        print(dumps(out), flush=True)
    if multi and not dry:
        # every rank leaves with the same status
        pass
    return rc


if __name__ == "__main__":
    sys.exit(main())
