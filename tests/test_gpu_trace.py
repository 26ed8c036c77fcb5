"""GPU parity: rt_trace_closest / rt_trace_any (HIP, through the C ABI) against the
CPU oracle on the same rays.  Bar: bit-exact prim ids, len and hit point bits,
occlusion flags (integer / IEEE-exact float work)."""
import numpy as np
import pytest

import oracle_lib as ol
from raytracer_amd import abi, host, scene

pytestmark = pytest.mark.gpu


def _primary_rays(sd, resx, resy, step=1):
    cam = ol.oracle_camera(sd.eye, sd.view, sd.fov, resx, resy)
    xs, ys = np.meshgrid(np.arange(0, resx, step), np.arange(0, resy, step), indexing="ij")
    return ol.oracle_primary_rays(cam, xs.ravel(), ys.ravel())


def _random_rays(n, seed, box=1.5):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-box, box, (n, 3)).astype(np.float32)
    o[:, 2] = rng.uniform(-1.5, 2.0, n)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    return np.concatenate([o, d.astype(np.float32)], axis=1)


def _edge_rays():
    """axis-parallel rays (zero direction components -> inf/NaN slabs, hw4.cpp:287-324 has no guard),
    rays starting on geometry, rays along the floor diagonal (shared triangle edge)."""
    r = []
    for o in ([0, 0, -1], [0.1, 0.2, -1], [0, -0.5, 0.5], [0.25, -0.5, 0.25], [0, 0.1, 0.9]):
        for d in ([0, 0, 1], [0, 1, 0], [1, 0, 0], [0, -1, 0], [0, 0, -1], [0, 0.6, 0.8], [0.6, 0, 0.8], [0.6, 0.8, 0]):
            r.append(o + d)
    t = np.linspace(-0.5, 0.5, 41)
    for a in t:                       # straight down onto the floor diagonal x == z - 0.5 ... and nearby
        r.append([a, 1.0, a + 0.5, 0, -1, 0])
        r.append([a, 1.0, a + 0.5, 1e-3, -1, 0])
    return np.array(r, np.float32)


def _assert_hits_equal(g, o):
    assert np.array_equal(g["prim"], o["prim"]), f"prim ids differ at {np.flatnonzero(g['prim'] != o['prim'])[:10]}"
    hit = o["prim"] >= 0
    assert np.array_equal(g["len"][hit].view(np.uint32), o["len"][hit].view(np.uint32))
    assert np.array_equal(g["p"][hit].view(np.uint32), o["p"][hit].view(np.uint32))
    assert np.all(g["len"][~hit] == np.float32(1e9))


@pytest.fixture(scope="module", params=["fast", "strict", "lane", "bvh"])
def monkey(monkey_sd, request):
    g = host.Scene(monkey_sd)
    g.set_traversal(request.param)
    o = ol.OracleScene(monkey_sd)
    yield g, o
    g.close(); o.close()


def test_scene_info_matches_reference_log(monkey):
    g, o = monkey
    gi, oi = g.info(), o.info()
    assert (gi.kd_nodes_built, gi.kd_max_depth) == (138, 9)          # executionTime_monkey.txt:4-5
    assert (gi.flat_nodes, gi.leaf_refs) == (oi.flat_nodes, oi.leaf_refs)


def test_closest_primary_256(monkey, monkey_sd):
    g, o = monkey
    rays = _primary_rays(monkey_sd, 256, 256)
    _assert_hits_equal(g.trace_closest(rays), o.trace_closest(rays))


def test_closest_random_and_edge_rays(monkey):
    g, o = monkey
    rays = np.concatenate([_random_rays(20000, 7), _edge_rays()])
    _assert_hits_equal(g.trace_closest(rays), o.trace_closest(rays))


def test_closest_ragged_sizes(monkey):
    g, o = monkey
    base = _random_rays(300, 11)
    for n in (0, 1, 63, 64, 65, 255, 257):
        gh = g.trace_closest(base[:n]); oh = o.trace_closest(base[:n])
        assert gh.shape[0] == n
        if n:
            _assert_hits_equal(gh, oh)


def test_any_hit_hw4_semantics(monkey, monkey_sd):
    """IsInShadow, hw4.cpp:957-1051: shadow rays from primary hit points to the light."""
    g, o = monkey
    rays = _primary_rays(monkey_sd, 128, 128)
    hits = o.trace_closest(rays)
    ok = hits["prim"] >= 0
    p = hits["p"][ok]
    light = np.array(monkey_sd.lights[0]["pos"], np.float32)
    d = light - p
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    srays = np.concatenate([p + np.float32(1e-4) * d, d], axis=1).astype(np.float32)
    srays = np.concatenate([srays, _random_rays(5000, 3)])
    go, oo = g.trace_any(srays), o.trace_any(srays)
    assert np.array_equal(go, oo)
    assert 0 < oo.sum() < oo.size


def test_any_hit_with_tmax(monkey):
    """final.cpp:997-1012 semantics on the KD scene: len < tLight."""
    g, o = monkey
    rays = _random_rays(8000, 5)
    tmax = np.random.default_rng(1).uniform(0.05, 2.0, rays.shape[0]).astype(np.float32)
    assert np.array_equal(g.trace_any(rays, tmax), o.trace_any(rays, tmax))


def test_final_scene_brute_force_with_spheres_and_culling():
    """Final/photon scene: no KD-tree, spheres, culling on, TOLERANCE_DEPTH 1e-3 (final.cpp:802-816)."""
    import os
    sd = scene.parse_scene_file(os.path.join(ol.GOLDEN, "final_input.txt"), flavour="final")
    g = host.Scene(sd); o = ol.OracleScene(sd)
    assert g.info().nloose == 14 and g.info().flat_nodes == 0
    rays = np.concatenate([_primary_rays(sd, 96, 96), _random_rays(6000, 9, box=0.45)])
    _assert_hits_equal(g.trace_closest(rays), o.trace_closest(rays))
    tmax = np.random.default_rng(2).uniform(0.05, 1.5, rays.shape[0]).astype(np.float32)
    assert np.array_equal(g.trace_any(rays, tmax), o.trace_any(rays, tmax))
    g.close(); o.close()


def _soup_scene(n, seed, nobj=3, big_floor=True, spread=0.45, tri=0.03):
    """random triangle soup in several objects, plus two large floor triangles (large |e1||e2| slack)."""
    from raytracer_amd.scene import SceneData
    rng = np.random.default_rng(seed)
    c = rng.uniform(-spread, spread, (n, 1, 3)); v = (c + rng.normal(scale=tri, size=(n, 3, 3))).astype(np.float32)
    v[..., 2] += 0.6
    if big_floor:
        v[0] = [[-2, -0.5, -1], [-2, -0.5, 3], [2, -0.5, 3]]; v[1] = [[-2, -0.5, -1], [2, -0.5, 3], [2, -0.5, -1]]
    nrm = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]).astype(np.float32)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-20)
    prims = np.concatenate([v.reshape(n, 9), nrm], axis=1).astype(np.float32)
    cuts = np.linspace(0, n, nobj + 1).astype(np.int64)
    return SceneData(prims=prims, prim_type=np.zeros(n, np.uint8), prim_material=np.zeros(n, np.int32),
                     materials=np.array([[1, 1, 1, 1, 0, 0, 1, 0, 0, 1]], np.float32), obj_first=cuts[:-1],
                     obj_count=np.diff(cuts), lights=[dict(pos=(1, 2, -1))], eye=np.array([0, 0, -1], np.float32),
                     view=np.array([0, 0, 1], np.float32), fov=45.0, resx=64, resy=64)


@pytest.mark.parametrize("mode", ["fast", "strict", "lane", "bvh"])
def test_triangle_soup_all_octants(mode):
    """3 000 overlapping triangles in 3 objects: rays from every octant, inside and outside the geometry."""
    sd = _soup_scene(3000, 31)
    g = host.Scene(sd); g.set_traversal(mode)
    o = ol.OracleScene(sd)
    rays = np.concatenate([_random_rays(30000, 41, box=0.7), _primary_rays(sd, 128, 128), _edge_rays()])
    # unnormalised and tiny/huge directions take the guarded path
    extra = _random_rays(2000, 43); extra[:, 3:6] *= np.float32(7.5); extra[::3, 3:6] *= np.float32(1e-3)
    rays = np.concatenate([rays, extra])
    _assert_hits_equal(g.trace_closest(rays), o.trace_closest(rays))
    assert np.array_equal(g.trace_any(rays), o.trace_any(rays))
    tmax = np.random.default_rng(3).uniform(0.02, 1.5, rays.shape[0]).astype(np.float32)
    assert np.array_equal(g.trace_any(rays, tmax), o.trace_any(rays, tmax))
    g.close(); o.close()


def test_exact_ties_resolve_like_the_reference_order():
    """coplanar duplicated triangles in different objects: equal len, the reference keeps the first visited."""
    from raytracer_amd.scene import SceneData
    base = _soup_scene(400, 77, nobj=1, big_floor=False)
    prims = np.concatenate([base.prims, base.prims[::-1]])           # second object: same triangles, reversed order
    n = prims.shape[0]
    sd = SceneData(prims=prims, prim_type=np.zeros(n, np.uint8), prim_material=np.zeros(n, np.int32),
                   materials=base.materials, obj_first=np.array([0, 400], np.int64), obj_count=np.array([400, 400], np.int64),
                   lights=base.lights, eye=base.eye, view=base.view, fov=45.0, resx=64, resy=64)
    o = ol.OracleScene(sd)
    rays = np.concatenate([_primary_rays(sd, 160, 160), _random_rays(20000, 5, box=0.6)])
    want = o.trace_closest(rays)
    assert (want["prim"] >= 0).sum() > 5000 and (want["prim"][want["prim"] >= 0] < 400).all()   # first object wins every tie
    for mode in ("fast", "strict", "lane", "bvh"):
        g = host.Scene(sd); g.set_traversal(mode)
        _assert_hits_equal(g.trace_closest(rays), want)
        g.close()
    o.close()


def _grazing_rays(sd, n, seed):
    """rays that lie almost IN the plane of a triangle and pass near it: where the reference's rounded u / v / t accept
    hits well outside the triangle (|det| just above the 1e-6 cut-off), i.e. where the culling slack of the fast modes
    (lateral_slack) has to be wide enough."""
    rng = np.random.default_rng(seed)
    tri = sd.prims[:, :9].reshape(-1, 3, 3).astype(np.float64)
    t = tri[rng.integers(0, tri.shape[0], n)]
    e1, e2 = t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]
    nrm = np.cross(e1, e2); nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-300)
    uv = rng.uniform(-0.6, 1.6, (n, 2))
    p = t[:, 0] + uv[:, :1] * e1 + uv[:, 1:] * e2                      # a point in the plane, in or near the triangle
    a = rng.uniform(0, 2 * np.pi, n)[:, None]
    b1 = e1 / np.maximum(np.linalg.norm(e1, axis=1, keepdims=True), 1e-300)
    b2 = np.cross(nrm, b1)
    inplane = np.cos(a) * b1 + np.sin(a) * b2
    tilt = (10.0 ** rng.uniform(-7, -1.5, n) * rng.choice([-1.0, 1.0], n))[:, None]
    d = inplane + tilt * nrm
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dist = (10.0 ** rng.uniform(-1, 1.5, n))[:, None]
    o = p - dist * d
    return np.concatenate([o, d], axis=1).astype(np.float32)


def _adversarial_rays(sd, n, seed, first_tri=0):
    """Rays built to need the culling slack, not merely to graze: for a triangle, the vertex V that is extreme
    along a coordinate axis a (so V defines a face of the triangle's bounding box), a plane point p just BEYOND V along a
    (10^-4 .. 5 % of the triangle's size outside), and a direction d that lies in the triangle's plane at right angles to a,
    tilted out of the plane by exactly so much that |det| = |d . (e1 x e2)| falls in [1.05e-6, 4e-6] -- just above the
    reference's cut-off (hw4.cpp:209), where 1/det amplifies the rounding of u, v by up to 10^6 and the reference accepts
    plane points outside the triangle.  The whole ray stays outside the triangle's box along a, so a traversal that
    inflates its boxes by less than the reference's rounding can reach skips a hit the reference finds.  Solved in
    float64, rounded to float32 at the end."""
    rng = np.random.default_rng(seed)
    tri = sd.prims[first_tri:, :9].reshape(-1, 3, 3).astype(np.float64)
    e1a, e2a = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
    ok = np.linalg.norm(np.cross(e1a, e2a), axis=1) > 2e-5            # a tilt <= 0.2 rad must reach |det| = 4e-6
    tri = tri[ok]
    t = tri[rng.integers(0, tri.shape[0], n)]
    k = np.arange(n)
    ax = rng.integers(0, 3, n); sg = rng.choice([-1.0, 1.0], n)
    a = np.zeros((n, 3)); a[k, ax] = sg
    V = t[k, (t[k, :, ax] * sg[:, None]).argmax(1)]
    e1, e2 = t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]
    N = np.cross(e1, e2); area2 = np.linalg.norm(N, axis=1, keepdims=True); nrm = N / area2
    out = a - (a * nrm).sum(1, keepdims=True) * nrm                   # a projected into the plane: points out of the box
    lo = np.linalg.norm(out, axis=1, keepdims=True)
    w = np.cross(nrm, a); lw = np.linalg.norm(w, axis=1, keepdims=True)
    good = (lo[:, 0] > 0.2) & (lw[:, 0] > 0.2)                        # planes (almost) at right angles to a have no such direction
    out /= np.maximum(lo, 1e-30); w /= np.maximum(lw, 1e-30)
    size = np.sqrt(area2)
    p = V + (10.0 ** rng.uniform(-4, -1.3, (n, 1))) * size * out + rng.uniform(-0.3, 0.3, (n, 1)) * size * w
    tilt = rng.uniform(1.05e-6, 4e-6, (n, 1)) / area2 * rng.choice([-1.0, 1.0], (n, 1))
    d = w + tilt * nrm
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = p - (10.0 ** rng.uniform(-0.3, 1.3, (n, 1))) * d
    return np.concatenate([o, d], axis=1).astype(np.float32)[good]


@pytest.mark.parametrize("which", ["monkey", "soup"])
def test_grazing_rays_agree_across_traversals(which, monkey_sd):
    """1.5 M rays a fraction of a degree off a triangle's plane, from 0.1 to 30 units away: every traversal returns the
    strict one's hit, bit for bit (the strict one is the reference's loop, pinned by the tests above).  On a sample the
    oracle is asked as well."""
    sd = monkey_sd if which == "monkey" else _soup_scene(3000, 31)
    rays = _grazing_rays(sd, 1_500_000, 11)
    g = host.Scene(sd); g.set_traversal("strict")
    want = g.trace_closest(rays); want_any = g.trace_any(rays)
    assert (want["prim"] >= 0).sum() > 100_000
    o = ol.OracleScene(sd)
    sel = np.arange(0, rays.shape[0], 97)
    _assert_hits_equal(want[sel], o.trace_closest(rays[sel]))
    o.close()
    for mode in ("fast", "lane", "bvh"):
        g.set_traversal(mode)
        _assert_hits_equal(g.trace_closest(rays), want)
        assert np.array_equal(g.trace_any(rays), want_any), mode
    g.close()


def _adversarial_scene(which, monkey_sd):
    if which == "monkey":
        return monkey_sd, 2                                            # the floor's two triangles are an object of their own
    if which == "soup":
        return _soup_scene(3000, 31), 2
    # triangles of ~0.3 units in a 4-unit box, seen from up to 20 units away: |e1||e2| / |det| is large enough that the
    # reference accepts plane points a few % of a triangle outside it -- far more than the slab-rounding term covers
    return _soup_scene(600, 9, nobj=2, big_floor=False, spread=2.0, tri=0.3), 0


@pytest.mark.parametrize("which", ["monkey", "soup", "coarse"])
def test_adversarial_grazing_rays_agree_across_traversals(which, monkey_sd):
    """1 M rays built so that |det| lies in [1.05e-6, 4e-6] and the true plane point lies just beyond a box-defining vertex
    of the triangle (_adversarial_rays): every culled traversal must still return the strict one's hit (= the reference's
    loop; the oracle is asked on a sample), bit for bit."""
    sd, first = _adversarial_scene(which, monkey_sd)
    rays = _adversarial_rays(sd, 1_000_000, 5, first_tri=first)
    g = host.Scene(sd); g.set_traversal("strict")
    want = g.trace_closest(rays); want_any = g.trace_any(rays)
    assert (want["prim"] >= 0).sum() > 500_000
    o = ol.OracleScene(sd)
    sel = np.arange(0, rays.shape[0], 61)
    _assert_hits_equal(want[sel], o.trace_closest(rays[sel]))
    o.close()
    for mode in ("fast", "lane", "bvh"):
        g.set_traversal(mode)
        _assert_hits_equal(g.trace_closest(rays), want)
        assert np.array_equal(g.trace_any(rays), want_any), mode
    g.close()


def test_adversarial_rays_have_teeth(monkey_sd, monkeypatch):
    """The same rays against a library whose LATERAL box inflation is switched off (RT_DEBUG_SLACK_SCALE=0, read when a
    scene is built; the slab-rounding term stays): now the culled traversals lose hits the reference finds -- more than a
    thousand on the coarse soup, some on monkey -- so the test above does exercise the slack of DESIGN 4.1e rather than
    pass by luck.  (Measured on MI355X: coarse 1 888 closest-hit / 741 any-hit mismatches per million rays in bvh mode,
    monkey 145 / 89; with the slack at a tenth, none: the bound is the analysis', not the campaign's.)"""
    counts = {}
    for which in ("coarse", "monkey"):
        sd, first = _adversarial_scene(which, monkey_sd)
        rays = _adversarial_rays(sd, 1_000_000, 5, first_tri=first)
        monkeypatch.delenv("RT_DEBUG_SLACK_SCALE", raising=False)
        g = host.Scene(sd); g.set_traversal("strict")
        want = g.trace_closest(rays); want_any = g.trace_any(rays)
        g.close()
        monkeypatch.setenv("RT_DEBUG_SLACK_SCALE", "0")
        g = host.Scene(sd)
        for mode in ("bvh", "fast", "lane"):
            g.set_traversal(mode)
            got = g.trace_closest(rays)
            counts[which, mode] = (int(((got["prim"] != want["prim"]) | (got["len"].view(np.uint32) != want["len"].view(np.uint32))).sum()),
                                   int((g.trace_any(rays) != want_any).sum()))
        g.close()
        monkeypatch.delenv("RT_DEBUG_SLACK_SCALE")
    print("mismatches without the lateral slack:", counts)
    assert counts["coarse", "bvh"][0] >= 1000 and counts["coarse", "lane"][0] >= 1000 and counts["coarse", "fast"][0] >= 100
    assert counts["coarse", "bvh"][1] >= 200
    assert counts["monkey", "bvh"][0] >= 20


@pytest.fixture(scope="module")
def mesh100k():
    """BASELINE configuration 3's generator (raytracer_amd/synth.py) at 102 400 triangles + floor.  No reference scene
    has this many triangles ("parity unpinned": the checker is the oracle, whose KD builder and traversal are pinned on
    monkey and on random soups against the reference)."""
    from raytracer_amd import synth
    sd = synth.mesh_scene(320, 160)
    g = host.Scene(sd); o = ol.OracleScene(sd)
    yield sd, g, o
    g.close(); o.close()


@pytest.mark.parametrize("mode", ["fast", "strict", "lane", "bvh"])
def test_synthetic_mesh_100k_vs_oracle(mesh100k, mode):
    sd, g, o = mesh100k
    g.set_traversal(mode)
    gi, oi = g.info(), o.info()
    assert (gi.kd_nodes_built, gi.kd_max_depth, gi.flat_nodes, gi.leaf_refs) == (oi.kd_nodes_built, oi.kd_max_depth, oi.flat_nodes, oi.leaf_refs)
    rays = np.concatenate([_primary_rays(sd, 320, 180), _random_rays(20000, 71, box=0.8), _edge_rays()])
    want = o.trace_closest(rays)
    assert (want["prim"] >= 2).sum() > 10_000
    _assert_hits_equal(g.trace_closest(rays), want)
    # shadow rays of the primary hits towards the light + the random rays, both any-hit semantics
    ok = want["prim"] >= 0
    pts = want["p"][ok][:40000]
    light = np.array(sd.lights[0]["pos"], np.float32)
    d = light - pts; d /= np.linalg.norm(d, axis=1, keepdims=True)
    srays = np.concatenate([np.concatenate([pts + np.float32(1e-4) * d, d], axis=1).astype(np.float32), rays[-20000:]])
    oo = o.trace_any(srays)
    assert 0 < oo.sum() < oo.size and np.array_equal(g.trace_any(srays), oo)
    tmax = np.random.default_rng(6).uniform(0.02, 1.5, srays.shape[0]).astype(np.float32)
    assert np.array_equal(g.trace_any(srays, tmax), o.trace_any(srays, tmax))


@pytest.mark.parametrize("mode", ["fast", "strict", "lane", "bvh"])
def test_tree_triangles_and_loose_spheres_together(mode, monkey_sd):
    """an `O` object (KD-tree, and the build's BVH over it) next to spheres, which are tested one by one after the trees
    (hw4.cpp:808-871 visits objects first, then the sphere list): closest hit with the strict-< rule between the two
    kinds, any hit; the refilling per-lane kernels test the spheres when a lane takes its ray (a tree hit of equal len
    still wins)."""
    from raytracer_amd.scene import SceneData
    base = _soup_scene(600, 5, nobj=2)
    n = base.prims.shape[0]
    sph = np.zeros((3, base.prims.shape[1]), np.float32)
    sph[:, :4] = [[0.1, 0.0, 0.6, 0.25], [-0.3, 0.2, 0.9, 0.15], [0.0, -0.45, 0.5, 0.1]]
    prims = np.concatenate([base.prims, sph])
    ptype = np.concatenate([np.zeros(n, np.uint8), np.full(3, abi.RT_PRIM_SPHERE, np.uint8)])
    sd = SceneData(prims=prims, prim_type=ptype, prim_material=np.zeros(n + 3, np.int32), materials=base.materials,
                   obj_first=base.obj_first, obj_count=base.obj_count, lights=base.lights, eye=base.eye, view=base.view,
                   fov=45.0, resx=64, resy=64)
    g = host.Scene(sd); g.set_traversal(mode)
    assert g.info().nloose == 3 and g.info().flat_nodes > 0
    o = ol.OracleScene(sd)
    rays = np.concatenate([_random_rays(20000, 3, box=0.7), _primary_rays(sd, 96, 96), _edge_rays()])
    want = o.trace_closest(rays)
    assert (want["prim"] >= n).sum() > 500 and ((want["prim"] >= 0) & (want["prim"] < n)).sum() > 500
    _assert_hits_equal(g.trace_closest(rays), want)
    assert np.array_equal(g.trace_any(rays), o.trace_any(rays))
    tmax = np.random.default_rng(8).uniform(0.02, 1.5, rays.shape[0]).astype(np.float32)
    assert np.array_equal(g.trace_any(rays, tmax), o.trace_any(rays, tmax))
    g.close(); o.close()
