"""Scene-file front end: a Python mirror of the reference's token-loop parsers
(HW4/hw4/hw4.cpp:538-765 and Final/photon/final.cpp:509-735) that produces the
flat arrays of `rt_scene_desc`.  Host-side plumbing only: no intersection or
shading arithmetic happens here.
"""
import ctypes as C
import math
from dataclasses import dataclass, field

import numpy as np

from . import abi

F32 = np.float32


def _atof(tok):
    """atof(): longest numeric prefix, 0.0 when there is none."""
    try:
        return float(tok)
    except ValueError:
        import re
        m = re.match(r"\s*[-+]?(\d+\.?\d*([eE][-+]?\d+)?|\.\d+([eE][-+]?\d+)?)", tok)
        return float(m.group(0)) if m else 0.0


def _atoi(tok):
    import re
    m = re.match(r"\s*[-+]?\d+", tok)
    return int(m.group(0)) if m else 0


def normalize_f32(v):
    """vec3::normalize (algebra3.cpp:280-282,295-302): len in float, d_inv = float(1./len)."""
    v = np.asarray(v, dtype=F32)
    l2 = F32(F32(v[0] * v[0]) + F32(v[1] * v[1])) + F32(v[2] * v[2])
    length = np.sqrt(F32(l2), dtype=F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        d_inv = F32(np.float64(1.0) / np.float64(length))
        return np.array([v[0] * d_inv, v[1] * d_inv, v[2] * d_inv], dtype=F32)


def cross_f32(a, b):
    a = np.asarray(a, F32); b = np.asarray(b, F32)
    return np.array([F32(a[1] * b[2]) - F32(a[2] * b[1]), F32(a[2] * b[0]) - F32(a[0] * b[2]),
                     F32(a[0] * b[1]) - F32(a[1] * b[0])], dtype=F32)


@dataclass
class SceneData:
    """What the parser leaves behind (reference globals: simpleMeshes, simpleLights,
    simpleObjects, camera)."""
    prims: np.ndarray                      # [n][12] float32
    prim_type: np.ndarray                  # [n] uint8
    prim_material: np.ndarray              # [n] int32
    materials: np.ndarray                  # [m][10] float32
    obj_first: np.ndarray                  # [o] int64
    obj_count: np.ndarray                  # [o] int64
    lights: list = field(default_factory=list)   # list of dicts
    eye: np.ndarray = None
    view: np.ndarray = None                # raw V line (normalised by the camera set-up)
    fov: float = 45.0
    resx: int = 256
    resy: int = 256
    flavour: str = "hw4"                   # "hw4" | "final"

    @property
    def nprim(self):
        return int(self.prims.shape[0])

    def light_array(self):
        arr = (abi.rt_light * max(1, len(self.lights)))()
        for i, l in enumerate(self.lights):
            a = arr[i]
            a.type = l.get("type", abi.RT_LIGHT_POINT)
            a.pos[:] = [float(x) for x in l["pos"]]
            a.color[:] = [float(x) for x in l.get("color", (1, 1, 1))]
            a.intensity = float(l.get("intensity", 1.0))
            a.dir1[:] = [float(x) for x in l.get("dir1", (1, 0, 0))]
            a.dir2[:] = [float(x) for x in l.get("dir2", (0, 0, 1))]
            a.size1 = float(l.get("size1", 0.0)); a.size2 = float(l.get("size2", 0.0))
            a.n1 = int(l.get("n1", 1)); a.n2 = int(l.get("n2", 1))
        return arr

    def make_desc(self, device=-1):
        """Returns (rt_scene_desc, keepalive) -- keepalive holds the arrays the desc points into."""
        prims = np.ascontiguousarray(self.prims, dtype=F32).reshape(-1, 12)
        ptype = np.ascontiguousarray(self.prim_type, dtype=np.uint8)
        pmat = np.ascontiguousarray(self.prim_material, dtype=np.int32)
        mats = np.ascontiguousarray(self.materials, dtype=F32).reshape(-1, 10)
        of = np.ascontiguousarray(self.obj_first, dtype=np.int64)
        oc = np.ascontiguousarray(self.obj_count, dtype=np.int64)
        lights = self.light_array()
        d = abi.rt_scene_desc()
        d.abi_version = abi.RT_ABI_VERSION
        final = self.flavour == "final"
        d.flags = abi.RT_SCENE_KD_REFERENCE | (abi.RT_SCENE_CULLING if final else 0)
        d.nprim = prims.shape[0]
        d.prims = prims.ctypes.data; d.prim_type = ptype.ctypes.data; d.prim_material = pmat.ctypes.data
        d.nmaterial = mats.shape[0]; d.materials = mats.ctypes.data
        d.nobject = of.shape[0]; d.obj_first = of.ctypes.data; d.obj_count = oc.ctypes.data
        d.nlight = len(self.lights); d.lights = C.addressof(lights)
        d.tolerance_depth = 0.001 if final else 0.00001      # parameter.h:9 / hw4.cpp:19
        d.shadow_bias = 0.0001; d.refract_bias = 0.01         # hw4.cpp:34-35
        d.device = device
        return d, (prims, ptype, pmat, mats, of, oc, lights)


def parse_scene_text(text, flavour="hw4", resolution=None):
    """Token loop of hw4.cpp:538-748 (flavour "hw4": `T` carries a normal, `O`
    closes an object) or final.cpp:509-735 (flavour "final": `T` has 9 numbers,
    normal = (v2-v0)^(v1-v0); `RectL`; no objects)."""
    toks = text.split()
    pos = 0

    def nxt():
        nonlocal pos
        t = toks[pos] if pos < len(toks) else ""
        pos += 1
        return t

    prims, ptype, pmat, mats, lights = [], [], [], [], []
    cur_mat = None                       # material in force; default of hw4.cpp:521-526
    default_mat = [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 6.0, 0.9, 0.0, 1.0]   # surf, Ka, Kd, Ks, spec, refl, refr, Nr
    mat_index = {}

    def mat_id():
        m = tuple(default_mat if cur_mat is None else cur_mat)
        if m not in mat_index:
            mat_index[m] = len(mats); mats.append(list(m))
        return mat_index[m]

    eye = np.zeros(3, F32); view = np.array([0, 0, 1], F32); fov = 0.0; resx = resy = 0
    obj_first, obj_count = [], []
    cur_start = 0
    has_o = False
    while pos < len(toks):
        t = nxt()
        if t == "E":
            eye = np.array([_atof(nxt()) for _ in range(3)], F32)
        elif t == "V":
            view = np.array([_atof(nxt()) for _ in range(3)], F32)
        elif t == "F":
            fov = float(F32(_atof(nxt())))
        elif t == "R":
            resx = _atoi(nxt()); resy = _atoi(nxt())
        elif t == "S":
            v = [_atof(nxt()) for _ in range(4)]
            prims.append(v + [0.0] * 8); ptype.append(abi.RT_PRIM_SPHERE); pmat.append(mat_id())
        elif t == "T":
            v = np.array([_atof(nxt()) for _ in range(9)], F32)
            if flavour == "hw4":
                nrm = normalize_f32(np.array([_atof(nxt()) for _ in range(3)], F32))
            else:
                nrm = normalize_f32(cross_f32(v[6:9] - v[0:3], v[3:6] - v[0:3]))
            prims.append(list(v) + list(nrm)); ptype.append(abi.RT_PRIM_TRIANGLE); pmat.append(mat_id())
        elif t == "L":
            p = [_atof(nxt()) for _ in range(3)]
            lights.append(dict(type=abi.RT_LIGHT_POINT, pos=p, color=(1, 1, 1), intensity=1.0))
        elif t == "ML":
            v = [_atof(nxt()) for _ in range(7)]
            lights.append(dict(type=abi.RT_LIGHT_POINT, pos=v[0:3], color=v[3:6], intensity=v[6]))
        elif t == "M":
            cur_mat = [float(F32(_atof(nxt()))) for _ in range(10)]
        elif t == "O" and flavour == "hw4":
            n = len(prims) - cur_start
            if n == 0:
                continue
            has_o = True
            obj_first.append(cur_start); obj_count.append(n); cur_start = len(prims)
        elif t == "RectL" and flavour == "final":
            p = [_atof(nxt()) for _ in range(3)]
            s1 = _atof(nxt()); s2 = _atof(nxt()); n1 = _atoi(nxt()); n2 = _atoi(nxt())
            lights.append(dict(type=abi.RT_LIGHT_RECT, pos=p, color=(1, 1, 1), intensity=1.0,
                               dir1=(1, 0, 0), dir2=(0, 0, 1), size1=s1, size2=s2, n1=n1, n2=n2))
    ptype = np.array(ptype, np.uint8) if prims else np.zeros(0, np.uint8)
    if flavour == "hw4":
        if not has_o:                                   # hw4.cpp:754-765
            if len(prims):
                obj_first.append(0); obj_count.append(len(prims))
        elif cur_start < len(prims):
            # meshes after the last `O` are parsed but never put in a tree (hw4.cpp:768):
            # the reference cannot hit them
            ptype[cur_start:] = abi.RT_PRIM_DISABLED
    if resolution is not None:
        resx, resy = resolution
    if not mats:
        mats.append(default_mat)
    return SceneData(prims=np.array(prims, F32).reshape(-1, 12), prim_type=ptype,
                     prim_material=np.array(pmat, np.int32), materials=np.array(mats, F32).reshape(-1, 10),
                     obj_first=np.array(obj_first, np.int64), obj_count=np.array(obj_count, np.int64),
                     lights=lights, eye=eye, view=view, fov=fov, resx=resx, resy=resy, flavour=flavour)


def parse_scene_file(path, flavour="hw4", resolution=None):
    with open(path, "r") as f:
        return parse_scene_text(f.read(), flavour, resolution)


def write_scene_text(sd, with_objects=True):
    """Emit a SceneData in the reference's text format (used to hand synthetic scenes to the
    compiled reference binary).  Floats are written with repr precision of float32."""
    out = [f"E {sd.eye[0]:.9g} {sd.eye[1]:.9g} {sd.eye[2]:.9g}", f"V {sd.view[0]:.9g} {sd.view[1]:.9g} {sd.view[2]:.9g}",
           f"F {sd.fov:.9g}", f"R {sd.resx} {sd.resy}"]
    for l in sd.lights:
        if l.get("type", 0) == abi.RT_LIGHT_POINT:
            c = l.get("color", (1, 1, 1))
            out.append("ML " + " ".join(f"{float(x):.9g}" for x in list(l["pos"]) + list(c) + [l.get("intensity", 1.0)]))
    ends = {int(f + c): True for f, c in zip(sd.obj_first, sd.obj_count)} if with_objects else {}
    last_mat = None
    for i in range(sd.nprim):
        m = int(sd.prim_material[i])
        if m != last_mat:
            out.append("M " + " ".join(f"{float(x):.9g}" for x in sd.materials[m])); last_mat = m
        f = sd.prims[i]
        if sd.prim_type[i] == abi.RT_PRIM_SPHERE:
            out.append("S " + " ".join(f"{float(x):.9g}" for x in f[:4]))
        else:
            n = 12 if sd.flavour == "hw4" else 9
            out.append("T " + " ".join(f"{float(x):.9g}" for x in f[:n]))
        if (i + 1) in ends:
            out.append("O")
    return "\n".join(out) + "\n"
