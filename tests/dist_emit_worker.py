"""Worker of tests/test_dist_gloo.py::test_sharded_emission_allgather_over_gloo: one rank of the photon-map
exchange (raytracer_amd/dist.py:emit_photon_map) over gloo, with the CPU oracle standing in for the GPU emitter."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol                                   # noqa: E402
from raytracer_amd import abi, host, scene                # noqa: E402
from raytracer_amd.dist import emit_photon_map, balance_on_rank0   # noqa: E402


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    sd = scene.parse_scene_file(os.path.join(ol.GOLDEN, "final_input.txt"), flavour="final")
    o = ol.OracleScene(sd)
    results = {}
    for which, target, batch in ((abi.RT_PMAP_GLOBAL, 5000, 700), (abi.RT_PMAP_CAUSTIC, 300, 2048)):
        e = host.final_emit_params(which, seed=77)

        def emit(first, count):
            cap = 2 * count + 64
            out = np.zeros(cap, abi.photon_dtype)
            n = ol.oracle().orc_emit_photons(o.h, C.byref(e), first, count, out.ctypes.data, cap)
            return torch.from_numpy(out[:n].view(np.uint8).reshape(-1, 40).copy())
        st = {}
        got, used = emit_photon_map(emit, target, rank, world, batch, dist, "cpu", st)
        # every rank must hold the same array, and it must be what one process emits alone
        digest = torch.tensor([int(np.frombuffer(got.tobytes(), np.uint8).astype(np.int64).sum())], dtype=torch.int64)
        all_d = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(all_d, digest)
        same_everywhere = all(int(d.item()) == int(digest.item()) for d in all_d)
        alone, used1 = emit_photon_map(emit, target, 0, 1, batch * world)
        # one build per node: rank 0 balances + scales, everybody receives the heap (dist.py: balance_on_rank0)
        a = np.zeros(target + 1, abi.photon_dtype); a[1:] = got
        heap = balance_on_rank0(a, 1.0 / target, rank, world, dist, torch)
        b = np.zeros(target + 1, abi.photon_dtype); b[1:] = alone
        abi.check(abi.load().rt_host_photon_balance(b.ctypes.data, target))
        b["power"][:target] = b["power"][:target] * np.float32(1.0 / target)
        hd = torch.tensor([int(np.frombuffer(heap.tobytes(), np.uint8).astype(np.int64).sum())], dtype=torch.int64)
        all_h = [torch.zeros_like(hd) for _ in range(world)]
        dist.all_gather(all_h, hd)
        heap_ok = all(int(d.item()) == int(hd.item()) for d in all_h) and heap.tobytes() == b.tobytes()
        results[str(which)] = dict(n=int(got.size), same_everywhere=same_everywhere, heap_shared_equals_own_balance=bool(heap_ok),
                                   equals_single_process=bool(np.array_equal(got.view(np.uint8), alone.view(np.uint8))),
                                   emissions=int(used), allgather_bytes=int(st.get("allgather_bytes", 0)))
    if rank == 0:
        print(json.dumps(dict(world=world, results=results)), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
