"""Time rt_photonmap_build_dev against the host build.  usage: time_photon_build.py N [--host] [--check]
Photons are made on the device (uniform in a box, a third of them on a wall with the few distinct coordinates a hit
point's rounding leaves), so N = 200 M needs no 8 GB host array unless --host / --check ask for the host build."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import torch
from raytracer_amd import abi, host, scene


def main():
    n = int(float(sys.argv[1]))
    want_host = "--host" in sys.argv or "--check" in sys.argv
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    sd = scene.parse_scene_file(os.path.join(root, "tests", "golden", "final_input.txt"), flavour="final")
    sc = host.Scene(sd)
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    ph = torch.zeros((n + 1, 10), dtype=torch.float32, device="cuda")          # rt_photon = 40 bytes
    ph[1:, 0:3] = torch.rand((n, 3), generator=g, device="cuda") - 0.5
    third = n // 3
    ph[1:1 + third, 1] = -0.5 + (torch.randint(-1, 2, (third,), generator=g, device="cuda").float() * 5.9604645e-08)
    ph[1:, 6:9] = torch.rand((n, 3), generator=g, device="cuda") * 1e-4
    src = ph.clone() if want_host else None
    torch.cuda.synchronize()
    out = {"photons": n}
    for rep in range(2):
        if rep: ph.copy_(src) if src is not None else None
        if rep and src is None: break
        t0 = time.perf_counter()
        sc.photonmap_build_dev(0, ph.data_ptr(), n, 1.0)
        torch.cuda.synchronize()
        out["device_build_s" if rep == 0 else "device_build_again_s"] = round(time.perf_counter() - t0, 4)
    if want_host:
        a = src.cpu().numpy().view(np.uint8).reshape(-1).view(abi.photon_dtype).copy()
        sc2 = host.Scene(sd)
        os.environ["RT_PHOTON_BUILD"] = "host"
        t0 = time.perf_counter()
        h = sc2.photonmap_build(0, a, 1.0)
        out["host_build_s"] = round(time.perf_counter() - t0, 4)
        out["host_threads"] = int(os.environ.get("RT_HOST_THREADS", "0") or 0) or min(16, os.cpu_count() or 1)
        if "--check" in sys.argv:
            d = ph.cpu().numpy().view(np.uint8).reshape(-1).view(abi.photon_dtype)
            out["heap_equal"] = bool(h[1:].tobytes() == d[1:].tobytes())
        sc2.close()
    sc.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
