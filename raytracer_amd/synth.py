"""Deterministic synthetic scenes for BASELINE configurations 3-5.

The reference ships no scene larger than ~70 k triangles (and that one, rabbit.txt, is missing
from the reference checkout), so the large-scene workloads are generated: a displaced sphere
tessellated as a latitude/longitude grid (one `O` object) over the monkey scene's two-triangle
floor (another `O` object), lit like monkey.txt (`L 5 5 -5`) and seen by the same camera.
Everything is float32 and a pure function of (nu, nv, seed): the same arrays on every rank.
"""
import numpy as np

from . import abi
from .scene import SceneData

F32 = np.float32


def _hash01(i, j, seed):
    """cheap integer hash -> [0,1), vectorised; defines the per-vertex displacement noise"""
    h = (i.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) + j.astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F)
         + np.uint64(seed) * np.uint64(0x165667B19E3779F9))
    h ^= h >> np.uint64(29); h *= np.uint64(0xBF58476D1CE4E5B9); h ^= h >> np.uint64(32)
    return (h >> np.uint64(40)).astype(np.float64) / float(1 << 24)


def displaced_sphere(nu, nv, seed=1234, center=(0.0, 0.0, 0.55), radius=0.3):
    """nu x nv quads (2 triangles each) on a sphere with low-frequency bumps and per-vertex noise.
    Returns prims [2*nu*nv][12] float32: 9 vertex floats + the flat unit normal (HW4 `T` line layout)."""
    with np.errstate(over="ignore"):
        iu = np.arange(nu + 1); iv = np.arange(nv + 1)
        phi = (iu % nu)[None, :] * (2.0 * np.pi / nu)                 # wraps: column nu == column 0
        theta = (0.02 + 0.96 * iv / nv)[:, None] * np.pi              # open caps: no degenerate pole triangles
        jj, ii = np.meshgrid(iv, iu % nu, indexing="ij")
        noise = _hash01(ii, jj, seed)
    r = radius * (1.0 + 0.12 * np.sin(7 * phi) * np.sin(5 * theta) + 0.05 * np.cos(13 * phi + 3 * theta)
                  + 0.004 * (noise - 0.5))
    x = center[0] + r * np.sin(theta) * np.cos(phi)
    y = center[1] + r * np.cos(theta)
    z = center[2] + r * np.sin(theta) * np.sin(phi)
    p = np.stack([x, y, z], axis=-1).astype(F32)                      # [nv+1][nu+1][3]
    a = p[:-1, :-1]; b = p[:-1, 1:]; c = p[1:, 1:]; d = p[1:, :-1]
    t1 = np.stack([a, b, c], axis=2); t2 = np.stack([a, c, d], axis=2)   # [nv][nu][3 verts][3]
    tris = np.stack([t1, t2], axis=2).reshape(-1, 3, 3)
    n = np.cross((tris[:, 1] - tris[:, 0]).astype(np.float64), (tris[:, 2] - tris[:, 0]).astype(np.float64))
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)
    return np.concatenate([tris.reshape(-1, 9), n.astype(F32)], axis=1).astype(F32)


def mesh_scene(nu=1000, nv=500, seed=1234, reflect=0.0, res=(3840, 2160)):
    """Floor object (2 triangles, monkey.txt's) + displaced-sphere object (2*nu*nv triangles)."""
    floor = np.array([[-0.5, -0.5, 0, -0.5, -0.5, 1, 0.5, -0.5, 1, 0, 1, 0],
                      [-0.5, -0.5, 0, 0.5, -0.5, 1, 0.5, -0.5, 0, 0, 1, 0]], F32)
    mesh = displaced_sphere(nu, nv, seed)
    prims = np.concatenate([floor, mesh])
    n = prims.shape[0]
    mats = np.array([[1.0, 1.0, 0.5, 0.1, 0.9, 0.3, 20, 0.5 if reflect > 0 else 0.0, 0.0, 0.0],
                     [0.5, 0.5, 1.0, 0.1, 0.9, 0.3, 20.0, reflect, 0.0, 1.6]], F32)
    pmat = np.ones(n, np.int32); pmat[:2] = 0
    return SceneData(prims=prims, prim_type=np.zeros(n, np.uint8), prim_material=pmat, materials=mats,
                     obj_first=np.array([0, 2], np.int64), obj_count=np.array([2, n - 2], np.int64),
                     lights=[dict(type=abi.RT_LIGHT_POINT, pos=(5, 5, -5), color=(1, 1, 1), intensity=1.0)],
                     eye=np.array([0, 0, -1], F32), view=np.array([0, 0, 1], F32), fov=45.0,
                     resx=res[0], resy=res[1], flavour="hw4")
