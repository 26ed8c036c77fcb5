"""Worker of tests/test_gpu_dist_rccl.py: the collectives of the N > 1 path over RCCL ("nccl" backend) with the ranks the box
has -- ONE on the one-GPU box, so this checks that RCCL initialises there and takes the calls, tensor types and shapes that
raytracer_amd/dist.py and bench.py make (uint8 / int64 all-gather, broadcast, async gather to rank 0, all-reduce), not that
they scale.  The GPU emitter is the real one (rt_photonmap_emit_dev)."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench                                              # noqa: E402
from raytracer_amd import abi, host, scene                # noqa: E402
from raytracer_amd.dist import emit_photon_map, gpu_emitter, share_from_rank0   # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def main():
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend="nccl")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", torch.cuda.current_device())
    out = dict(world=world, backend=dist.get_backend())
    # bench.py's helpers (Coll) and its gather of the ranks' packed tiles to rank 0
    coll = bench.Coll(dist, torch, "nccl")
    out["all_ok"] = coll.all_ok(True, dev) and not coll.all_ok(rank != 0, dev)
    out["max_float"] = coll.max_float(1.5 + rank, dev); out["sum_float"] = coll.sum_float(2.0, dev)
    send = torch.arange(3 * 5000, dtype=torch.int64, device=dev).remainder(251).to(torch.uint8).view(-1, 3) + rank
    recv = [torch.zeros_like(send) for _ in range(world)] if rank == 0 else None
    work = dist.gather(send, recv, dst=0, async_op=True)
    work.wait()
    torch.cuda.synchronize()
    out["gather_ok"] = bool(rank != 0 or all(torch.equal(recv[r], (send - rank + r)) for r in range(world)))
    coll.gather0(send, recv if rank == 0 else [], rank)
    # the photon-map exchange: sharded emission + all-gather, then the build's image broadcast from rank 0
    sd = scene.parse_scene_file(os.path.join(GOLDEN, "final_input.txt"), flavour="final")
    gs = host.Scene(sd, device=torch.cuda.current_device())
    res = {}
    for which, target, batch in ((abi.RT_PMAP_GLOBAL, 5000, 700), (abi.RT_PMAP_CAUSTIC, 300, 2048)):
        e = host.final_emit_params(which, seed=77)
        st = {}
        got, used = emit_photon_map(gpu_emitter(gs, e, torch), target, rank, world, batch, dist, "cuda", st, keep_on_device=True,
                                    exchange_always=True)
        alone, used1 = emit_photon_map(gpu_emitter(gs, e, torch), target, 0, 1, batch * world, None, "cuda", None, keep_on_device=True)
        payload, hdr = share_from_rank0(lambda: (got.reshape(-1).clone(), [target, 7]), rank, world, dist, "cuda", torch, exchange_always=True)
        res[str(which)] = dict(n=int(got.shape[0]), equals_single_process=bool(torch.equal(got, alone)) and used == used1,
                               allgather_bytes=int(st.get("allgather_bytes", 0)),
                               broadcast_ok=bool(torch.equal(payload.view(-1, 40), got)) and hdr == [target, 7])
    out["maps"] = res
    gs.close()
    dist.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
