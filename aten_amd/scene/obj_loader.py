"""Minimal Wavefront OBJ / MTL reader producing what aten's ObjLoader consumes from tinyobjloader.

Mirrors the registration rules of src/libatenscene/ObjLoader.cpp:95-461:
  * a shape starts at every `o` / `g` statement; faces are triangulated;
  * one vertex is emitted per face corner, in file order (no de-duplication, :140-156);
  * per-face material id from `usemtl`.

tinyobjloader itself (3rdparty/tinyobjloader, pinned commit unknown) is absent from the
reference snapshot, so polygon triangulation is stated here: plain fan (0,1,2),(0,2,3)...
All fixtures in this repo record that rule.
"""
import os

import numpy as np


class ObjMaterial:
    def __init__(self, name):
        self.name = name
        self.diffuse = (1.0, 1.0, 1.0)
        self.emission = (0.0, 0.0, 0.0)
        self.diffuse_texname = ""
        self.bump_texname = ""


class ObjShape:
    def __init__(self, name):
        self.name = name
        self.corners = []       # list of (v, vt, vn) 0-based, -1 = absent; 3 per triangle
        self.material_ids = []  # per triangle


def load_mtl(path):
    mtls, cur = [], None
    if not os.path.exists(path):
        return mtls
    with open(path, "r", errors="replace") as f:
        for line in f:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            k = t[0]
            if k == "newmtl":
                cur = ObjMaterial(" ".join(t[1:]))
                mtls.append(cur)
            elif cur is None:
                continue
            elif k == "Kd":
                cur.diffuse = tuple(float(x) for x in t[1:4])
            elif k == "Ke":
                cur.emission = tuple(float(x) for x in t[1:4])
            elif k == "map_Kd":
                cur.diffuse_texname = t[-1]
            elif k in ("map_bump", "map_Bump", "bump"):
                cur.bump_texname = t[-1]
    return mtls


def _fix(i, n):
    """OBJ index (1-based, negative = relative) -> 0-based."""
    return i - 1 if i > 0 else n + i


def load_obj(path):
    """Returns (positions[N,3] f32, texcoords[M,2] f32, normals[K,3] f32, shapes, materials)."""
    pos, tex, nml = [], [], []
    shapes, mtls = [], []
    mtl_index = {}
    cur, cur_mtl = None, -1
    base = os.path.dirname(path)
    with open(path, "r", errors="replace") as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            k = t[0]
            if k == "v":
                pos.append((float(t[1]), float(t[2]), float(t[3])))
            elif k == "vt":
                tex.append((float(t[1]), float(t[2]) if len(t) > 2 else 0.0))
            elif k == "vn":
                nml.append((float(t[1]), float(t[2]), float(t[3])))
            elif k in ("o", "g"):
                cur = ObjShape(" ".join(t[1:]))
                shapes.append(cur)
            elif k == "mtllib":
                mtls = mtls + load_mtl(os.path.join(base, t[1]))       # a second library appends (tinyobj): earlier ids stay valid
                mtl_index = {m.name: i for i, m in enumerate(mtls)}
            elif k == "usemtl":
                cur_mtl = mtl_index.get(" ".join(t[1:]), -1)
            elif k == "f":
                if cur is None:
                    cur = ObjShape("")
                    shapes.append(cur)
                cs = []
                for w in t[1:]:
                    p = w.split("/")
                    v = _fix(int(p[0]), len(pos))
                    vt = _fix(int(p[1]), len(tex)) if len(p) > 1 and p[1] else -1
                    vn = _fix(int(p[2]), len(nml)) if len(p) > 2 and p[2] else -1
                    cs.append((v, vt, vn))
                for j in range(1, len(cs) - 1):     # fan triangulation
                    cur.corners.extend([cs[0], cs[j], cs[j + 1]])
                    cur.material_ids.append(cur_mtl)
    shapes = [s for s in shapes if s.material_ids]   # tinyobj drops empty shapes
    return (np.asarray(pos, np.float32).reshape(-1, 3), np.asarray(tex, np.float32).reshape(-1, 2),
            np.asarray(nml, np.float32).reshape(-1, 3), shapes, mtls)
