"""Multi-GPU host logic (one process per GPU, torch.distributed: backend "nccl" = RCCL over xGMI on the
GPUs, "gloo" in the CPU tests).  SURVEY 8e: the scene and the photon maps are replicated, rays follow their
image tiles, and the path has ONE exchange step -- photon emission is sharded over the ranks by emission
index and the ranks' photon arrays are all-gathered into the same map everywhere.

Nothing here computes photons or pixels: `emit_fn` is rt_photonmap_emit_dev on a GPU rank (the CPU tests pass
the oracle's emitter to exercise the same exchange over gloo).
"""
import numpy as np

from . import abi


def emit_photon_map(emit_fn, target, rank=0, world=1, batch=None, dist=None, device="cpu", stats=None, keep_on_device=False,
                    exchange_always=False):
    """The `while (stored < max)` emission loops of final.cpp:1057 / :1090, sharded.

    emit_fn(first, count) -> uint8 tensor [n, 40] on `device`: the photons stored by emission indices
    [first, first+count) in emission order.  Every emission index owns its random stream, so the result does
    not depend on `world`: round k hands rank r the indices [(k*world + r)*batch, (k*world + r + 1)*batch),
    the counts and then the (padded) payloads are all-gathered, and every rank appends the blocks in index
    order until `target` photons are there; the first `target` are kept (AddPhoton's capacity test,
    photon.cpp:837).  Returns (numpy photon_dtype[target], emission indices consumed); with keep_on_device the first element
    is a uint8 tensor [target, 40] on `device` instead (the photons never visit the host: rt_photonmap_build_dev takes them).
    exchange_always: run the collectives with one rank too (tests/dist_rccl_worker.py: the RCCL calls on the one-GPU box).
    """
    import torch
    batch = int(batch or max(4096, -(-target // (4 * world))))
    parts, have, rnd = [], 0, 0
    while have < target:
        first = (rnd * world + rank) * batch
        mine = emit_fn(first, batch)
        assert mine.dtype == torch.uint8 and mine.dim() == 2 and mine.shape[1] == 40
        if world > 1 or exchange_always:
            cnt = torch.tensor([mine.shape[0]], dtype=torch.int64, device=device)
            counts = [torch.zeros_like(cnt) for _ in range(world)]
            dist.all_gather(counts, cnt)
            counts = [int(c.item()) for c in counts]
            maxc = max(max(counts), 1)
            send = torch.zeros((maxc, 40), dtype=torch.uint8, device=device)
            send[: mine.shape[0]] = mine
            recv = [torch.empty_like(send) for _ in range(world)]
            dist.all_gather(recv, send)                       # RCCL all-gather over xGMI: world x maxc x 40 B
            if stats is not None:
                stats["allgather_bytes"] = stats.get("allgather_bytes", 0) + world * maxc * 40
            blocks = [recv[r][: counts[r]] for r in range(world)]
        else:
            blocks = [mine]
        for b in blocks:                                      # index order == rank order within a round
            if have >= target:
                break
            parts.append(b if keep_on_device else b.cpu().numpy())
            have += b.shape[0]
        rnd += 1
        if rnd * world * batch > 1000 * max(target, 1) + 10 * world * batch:
            raise RuntimeError("photon emission stores (almost) nothing: target unreachable")
    if keep_on_device:
        out = torch.cat(parts, dim=0)[:target] if parts else torch.zeros((0, 40), dtype=torch.uint8, device=device)
        return out.contiguous(), rnd * world * batch
    out = np.concatenate(parts, axis=0)[:target] if parts else np.zeros((0, 40), np.uint8)
    return np.ascontiguousarray(out).view(abi.photon_dtype).reshape(-1), rnd * world * batch


def gpu_emitter(scene, e, torch, stream=None):
    """emit_fn over rt_photonmap_emit_dev: photons stay on the GPU for the all-gather."""
    import ctypes as C

    def emit(first, count):
        cap = 2 * count + 64
        buf = torch.empty((cap, 40), dtype=torch.uint8, device="cuda")
        n = C.c_int64(0)
        sp = C.c_void_p(stream.cuda_stream if stream is not None else torch.cuda.current_stream().cuda_stream)
        abi.check(scene.L.rt_photonmap_emit_dev(scene.h, C.byref(e), int(first), int(count), buf.data_ptr(), cap,
                                                C.byref(n), sp))
        if n.value > cap:
            raise abi.RtError(-4, "emission batch overflowed its buffer")
        return buf[: n.value]
    return emit


def share_from_rank0(make_fn, rank, world, dist, device, torch, src=0, exchange_always=False):
    """One rank builds, everybody holds: rank `src` calls make_fn() -> (uint8 tensor on `device`, list of int64 header
    words); the header and then the payload are broadcast (RCCL over xGMI on the GPUs, gloo in the CPU tests).
    Returns (payload tensor, header list) on every rank.  Used for the balanced photon map: the balance
    (photon.cpp:133-254) is a serial-ish host job of seconds to tens of seconds, so ONE rank per node runs it and the
    others receive the result instead of every rank repeating it on the same host's cores."""
    payload, header = make_fn() if rank == src else (None, None)
    if world == 1 and not exchange_always:
        return payload, header
    nh = torch.tensor([len(header) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(nh, src)
    h = torch.tensor(header if rank == src else [0] * int(nh.item()), dtype=torch.int64, device=device)
    dist.broadcast(h, src)
    nbytes = torch.tensor([payload.numel() if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(nbytes, src)
    if rank != src:
        payload = torch.empty(int(nbytes.item()), dtype=torch.uint8, device=device)
    dist.broadcast(payload, src)
    return payload, [int(x) for x in h.tolist()]


def balance_on_rank0(photons, scale, rank, world, dist, torch, device="cpu"):
    """photons: photon_dtype[target + 1] as AddPhoton leaves them (slot 0 unused), identical on every rank after the
    all-gather.  Rank 0 balances (librt_hip's host builder = PhotonMap::balance) and scales (ScalePhotonPower(scale, 0,
    n): slots 0..n-1, photon.cpp:826-833); every rank returns the same heap-order array.  Host form of the exchange
    (CPU tests; the GPU path moves the device image instead: build_photon_maps)."""
    n = photons.shape[0] - 1

    def make():
        a = np.ascontiguousarray(photons).copy()
        abi.check(abi.load().rt_host_photon_balance(a.ctypes.data, n))
        if scale != 1.0:
            a["power"][:n] = a["power"][:n] * np.float32(scale)
        return torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).to(device), [n]
    payload, _ = share_from_rank0(make, rank, world, dist, device, torch)
    return np.ascontiguousarray(payload.cpu().numpy()).view(abi.photon_dtype).reshape(-1)


def build_photon_maps(scene, torch, targets, seed, rank=0, world=1, dist=None, batch=None, every_rank_builds=False, want_heap=True):
    """Both maps of PhotonMappingImageSynthesis (final.cpp:1047-1129): sharded emission, all-gather, then ONE rank
    balances + scales the array, builds the second heap and uploads (rt_photonmap_build), and broadcasts the map's
    device image to the other ranks' GPUs (rt_photonmap_export_dev / _import_dev) -- the host build is not repeated
    per rank on the same host's cores.  `heap` (the host array in heap order) is returned on rank 0 only.
    every_rank_builds=True keeps the old shape (each rank builds for itself).  targets = (global, caustic).
    The photons stay on the device from the emission to the finished map (rt_photonmap_emit_dev -> all-gather ->
    rt_photonmap_build_dev); `heap` is downloaded only if want_heap (the checks of bench.py / the tests)."""
    from . import host
    out = []
    for which, target in ((abi.RT_PMAP_GLOBAL, targets[0]), (abi.RT_PMAP_CAUSTIC, targets[1])):
        if target <= 0:
            out.append(None)
            continue
        e = host.final_emit_params(which, seed=seed)
        st = {}
        ph, used = emit_photon_map(gpu_emitter(scene, e, torch), target, rank, world, batch, dist, "cuda", st, keep_on_device=True)
        scale = float(np.float32(1.0) / np.float32(max(target, 1)))                                               # final.cpp:1126

        def build_here():
            """slot 0 unused (photon.cpp:852), balance + ScalePhotonPower + image on this rank's GPU; -> host heap or None"""
            d = torch.zeros((target + 1, 40), dtype=torch.uint8, device="cuda")
            d[1:] = ph
            scene.photonmap_build_dev(which, d.data_ptr(), target, power_scale=scale,
                                      stream=torch.cuda.current_stream().cuda_stream)
            if not want_heap:
                return None
            return np.ascontiguousarray(d.cpu().numpy()).view(abi.photon_dtype).reshape(-1)
        if world == 1 or every_rank_builds:
            out.append(dict(heap=build_here(), emissions=used, **st))
            continue
        # one build per node: rank 0 balances, builds the second heap and uploads; the others receive its device image
        import ctypes as C
        import time
        L, sp = scene.L, C.c_void_p(torch.cuda.current_stream().cuda_stream)
        heap = None
        t0 = time.perf_counter()

        def make():
            nonlocal heap
            heap = build_here()
            n, nb = C.c_int64(0), C.c_int64(0)
            abi.check(L.rt_photonmap_image_bytes(scene.h, which, C.byref(n), C.byref(nb)))
            img = torch.empty(nb.value, dtype=torch.uint8, device="cuda")
            box = (C.c_float * 6)()
            abi.check(L.rt_photonmap_export_dev(scene.h, which, img.data_ptr(), nb.value, box, sp))
            torch.cuda.current_stream().synchronize()
            return img, [n.value] + [int(np.float32(b).view(np.int32)) for b in box]
        img, hdr = share_from_rank0(make, rank, world, dist, "cuda", torch)
        if rank != 0:
            box = (C.c_float * 6)(*[float(np.int32(b).view(np.float32)) for b in hdr[1:7]])
            abi.check(L.rt_photonmap_import_dev(scene.h, which, img.data_ptr(), hdr[0], img.numel(), box, sp))
        st["image_bytes"] = int(img.numel()); st["build_and_share_s"] = time.perf_counter() - t0
        del img
        out.append(dict(heap=heap, emissions=used, **st))
    return out
