"""The device half of rt_scene_create (kernels_build.hip): the seven octant layouts of the reference KD-trees (derived from
the reference's own order, layout 7; KDNode::HitTest's pop order, hw4.cpp:831-836) and of the build's BVH (from layout 0)
must be byte for byte the arrays the host emits.  RT_CHECK_LAYOUTS=1 makes rt_scene_create build both, download the
device's and compare; it fails the call on any difference.  Scenes: the bundled monkey (one object), the photon scene
(loose primitives only + a few objects), several objects of different sizes in one file, a one-triangle object and a
300 k-triangle mesh (deep trees)."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from raytracer_amd import abi, host, scene, synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def check_layouts():
    os.environ["RT_CHECK_LAYOUTS"] = "1"
    yield
    os.environ.pop("RT_CHECK_LAYOUTS", None)


def _several_objects():
    """four objects in one scene: 2, 1, 5 000 and 300 triangles (the forest of reference trees lies back to back)"""
    a = synth.mesh_scene(50, 50, seed=3)
    b = synth.mesh_scene(15, 10, seed=4)
    pa = np.asarray(a.prims).reshape(-1, 12); pb = np.asarray(b.prims).reshape(-1, 12).copy()
    pb[:, [1, 4, 7]] += 0.5                                      # lift the second mesh
    one = pa[:1].copy(); one[:, [0, 3, 6]] += 3.0
    floor, mesh_a = pa[:2], pa[2:]
    mesh_b = pb[2:]
    prims = np.concatenate([floor, one, mesh_a, mesh_b]).astype(np.float32)
    first = np.array([0, 2, 3, 3 + len(mesh_a)], np.int64); count = np.array([2, 1, len(mesh_a), len(mesh_b)], np.int64)
    import dataclasses
    return dataclasses.replace(a, prims=prims, prim_type=np.full(len(prims), a.prim_type[0], a.prim_type.dtype),
                               prim_material=np.zeros(len(prims), np.int32), obj_first=first, obj_count=count)


@pytest.mark.parametrize("name", ["monkey", "final", "objects", "mesh_300k"])
def test_device_layouts_equal_the_hosts(check_layouts, name):
    if name == "monkey":
        sd = scene.parse_scene_file(os.path.join(ol.GOLDEN, "monkey.txt"))
    elif name == "final":
        sd = scene.parse_scene_file(os.path.join(ol.GOLDEN, "final_input.txt"), flavour="final")
    elif name == "objects":
        sd = _several_objects()
    else:
        sd = synth.mesh_scene(500, 300)
    g = host.Scene(sd)                 # raises RtError("RT_CHECK_LAYOUTS: layout ... differs") on any difference
    try:
        assert (g.info().flat_nodes >= 1) == (name != "final")       # the photon scene has loose primitives only
        # and the layouts are the ones the kernels walk: a few rays per octant against the strict (layout 7) traversal
        rng = np.random.default_rng(7)
        lo = np.asarray(sd.prims).reshape(-1, 12)[:, :9].reshape(-1, 3).min(0); hi = np.asarray(sd.prims).reshape(-1, 12)[:, :9].reshape(-1, 3).max(0)
        lo = np.nan_to_num(lo, nan=-1.0, posinf=1.0, neginf=-1.0); hi = np.nan_to_num(hi, nan=1.0, posinf=1.0, neginf=-1.0)
        c = (lo + hi) / 2; r = max(1e-3, float(np.linalg.norm(hi - lo)))
        d = rng.normal(size=(4000, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays = np.concatenate([c - d * r + rng.normal(scale=0.05 * r, size=(4000, 3)), d], axis=1).astype(np.float32)
        out = {}
        for mode in ("strict", "fast", "lane", "bvh"):
            g.set_traversal(mode)
            out[mode] = g.trace_closest(rays)["prim"].copy()
        for mode in ("fast", "lane", "bvh"):
            assert np.array_equal(out[mode], out["strict"]), mode
        assert (out["strict"] >= 0).sum() > 100
    finally:
        g.close()


# ---------------------------------------------------------------------------------------------------------------------
# KDNode::BuildKD on the device (kernels_kdbuild.hip; hw4.cpp:330-490): RT_CHECK_KD=1 makes rt_scene_create build every
# object's tree on the device AND on the host and fails the call unless node count, depth, every node's centre / extent /
# axis and every leaf's list (order included) are the same; the arrays the kernels read are made from that tree by the
# same host code either way.  The default takes the device from RT_KD_DEVICE_MIN triangles up.

@pytest.fixture()
def check_kd():
    os.environ["RT_CHECK_KD"] = "1"
    yield
    os.environ.pop("RT_CHECK_KD", None)


def _duplicates_and_ties():
    """a mesh with every triangle twice, a block of identical triangles and coordinates that tie on the split planes"""
    a = synth.mesh_scene(40, 30, seed=5)
    p = np.asarray(a.prims).reshape(-1, 12)
    q = np.round(p[2:] * 8) / 8                                      # vertices on a coarse lattice: many equal coordinates
    prims = np.concatenate([p[:2], p[2:], p[2:], q, np.repeat(p[5:6], 300, axis=0)]).astype(np.float32)
    import dataclasses
    return dataclasses.replace(a, prims=prims, prim_type=np.full(len(prims), a.prim_type[0], a.prim_type.dtype),
                               prim_material=np.zeros(len(prims), np.int32), obj_first=np.array([0, 2], np.int64),
                               obj_count=np.array([2, len(prims) - 2], np.int64))


@pytest.mark.parametrize("name", ["monkey", "objects", "ties", "mesh_300k"])
def test_device_kd_build_equals_the_host_build(check_kd, name):
    if name == "monkey":
        sd = scene.parse_scene_file(os.path.join(ol.GOLDEN, "monkey.txt"))
    elif name == "objects":
        sd = _several_objects()
    elif name == "ties":
        sd = _duplicates_and_ties()
    else:
        sd = synth.mesh_scene(500, 300)
    g = host.Scene(sd)                 # raises RtError("RT_CHECK_KD: ...") on any difference
    try:
        inf = g.info()
        if name == "monkey":           # the reference's own log: 138 nodes, depth 9 (HW4/hw4/executionTime_monkey.txt:4-5)
            assert inf.kd_nodes_built == 138 and inf.kd_max_depth == 9
        assert inf.flat_nodes >= 1
    finally:
        g.close()


def test_device_built_tree_gives_the_reference_hits():
    """the monkey's tree built on the device alone (RT_KD_BUILD=device): the reference's 65 536 primary rays find the reference's triangles"""
    os.environ["RT_KD_BUILD"] = "device"
    try:
        sd = scene.parse_scene_file(os.path.join(ol.GOLDEN, "monkey.txt"))
        g = host.Scene(sd)
    finally:
        os.environ.pop("RT_KD_BUILD", None)
    try:
        gold = np.load(os.path.join(ol.GOLDEN, "monkey256_hits.npz"))
        for mode in ("strict", "fast", "bvh"):          # the reference's own primary rays and the triangles it found for them
            g.set_traversal(mode)
            assert np.array_equal(g.trace_closest(np.ascontiguousarray(gold["rays"]))["prim"], gold["prim"]), mode
    finally:
        g.close()
