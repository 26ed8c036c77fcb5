"""Multi-GPU host logic (one process per GPU, torch.distributed: backend "nccl" = RCCL over xGMI on the
GPUs, "gloo" in the CPU tests).  SURVEY 8e: the scene and the photon maps are replicated, rays follow their
image tiles, and the path has ONE exchange step -- photon emission is sharded over the ranks by emission
index and the ranks' photon arrays are all-gathered into the same map everywhere.

Nothing here computes photons or pixels: `emit_fn` is rt_photonmap_emit_dev on a GPU rank (the CPU tests pass
the oracle's emitter to exercise the same exchange over gloo).
"""
import numpy as np

from . import abi


def emit_photon_map(emit_fn, target, rank=0, world=1, batch=None, dist=None, device="cpu", stats=None):
    """The `while (stored < max)` emission loops of final.cpp:1057 / :1090, sharded.

    emit_fn(first, count) -> uint8 tensor [n, 40] on `device`: the photons stored by emission indices
    [first, first+count) in emission order.  Every emission index owns its random stream, so the result does
    not depend on `world`: round k hands rank r the indices [(k*world + r)*batch, (k*world + r + 1)*batch),
    the counts and then the (padded) payloads are all-gathered, and every rank appends the blocks in index
    order until `target` photons are there; the first `target` are kept (AddPhoton's capacity test,
    photon.cpp:837).  Returns (numpy photon_dtype[target], emission indices consumed).
    """
    import torch
    batch = int(batch or max(4096, -(-target // (4 * world))))
    parts, have, rnd = [], 0, 0
    while have < target:
        first = (rnd * world + rank) * batch
        mine = emit_fn(first, batch)
        assert mine.dtype == torch.uint8 and mine.dim() == 2 and mine.shape[1] == 40
        if world > 1:
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
            parts.append(b.cpu().numpy())
            have += b.shape[0]
        rnd += 1
        if rnd * world * batch > 1000 * max(target, 1) + 10 * world * batch:
            raise RuntimeError("photon emission stores (almost) nothing: target unreachable")
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


def build_photon_maps(scene, torch, targets, seed, rank=0, world=1, dist=None, batch=None):
    """Both maps of PhotonMappingImageSynthesis (final.cpp:1047-1129): sharded emission, all-gather, then every
    rank balances + scales the identical array and uploads it (rt_photonmap_build).  targets = (global, caustic)."""
    from . import host
    out = []
    for which, target in ((abi.RT_PMAP_GLOBAL, targets[0]), (abi.RT_PMAP_CAUSTIC, targets[1])):
        if target <= 0:
            out.append(None)
            continue
        e = host.final_emit_params(which, seed=seed)
        st = {}
        ph, used = emit_photon_map(gpu_emitter(scene, e, torch), target, rank, world, batch, dist, "cuda", st)
        a = np.zeros(target + 1, abi.photon_dtype); a[1:] = ph
        heap = scene.photonmap_build(which, a, power_scale=float(np.float32(1.0) / np.float32(max(target, 1))))   # final.cpp:1126
        out.append(dict(heap=heap, emissions=used, **st))
    return out
