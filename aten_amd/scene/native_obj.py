"""ctypes binding of the native scene ingestion in libaten_amd_scene.so (include/aten_amd_scene.h, csrc/host/obj_ingest.cpp):
OBJ / MTL with aten::ObjLoader's registration rules, and the XML aten::MaterialLoader reads."""
import ctypes as C

import numpy as np

from .._hostlib import hostlib

OBJ_TRIANGLE = np.dtype([("idx", np.uint32, 3), ("need_normal", np.int32), ("mesh", np.int32)])
OBJ_MESH = np.dtype([("mtl", np.int32), ("mesh_id", np.uint32), ("first_triangle", np.uint32), ("n_triangles", np.uint32),
                     ("object", np.int32), ("shape", np.int32)])
OBJ_OBJECT = np.dtype([("first_mesh", np.uint32), ("n_meshes", np.uint32), ("shape", np.int32), ("is_emissive_split", np.int32),
                       ("return_order", np.int32)])
assert OBJ_TRIANGLE.itemsize == 20 and OBJ_MESH.itemsize == 24 and OBJ_OBJECT.itemsize == 20


class _MaterialInfo(C.Structure):
    _fields_ = [("name", C.c_char_p), ("diffuse_texname", C.c_char_p), ("bump_texname", C.c_char_p),
                ("diffuse", C.c_float * 3), ("emission", C.c_float * 3)]


class _ParamInfo(C.Structure):
    _fields_ = [("name", C.c_char_p), ("text", C.c_char_p), ("kind", C.c_int32), ("value", C.c_float * 3)]


_bound = False


def _lib():
    global _bound
    l = hostlib()
    if not _bound:
        vp = C.c_void_p
        l.atns_obj_open.argtypes = [C.c_char_p, C.POINTER(vp)]
        l.atns_obj_close.argtypes = [vp]; l.atns_obj_close.restype = None
        for n in ("material_count", "shape_count", "vertex_count", "triangle_count", "mesh_count", "object_count"):
            f = getattr(l, "atns_obj_" + n); f.argtypes = [vp]; f.restype = C.c_uint32
        l.atns_obj_material.argtypes = [vp, C.c_uint32, C.POINTER(_MaterialInfo)]
        l.atns_obj_shape_name.argtypes = [vp, C.c_uint32]; l.atns_obj_shape_name.restype = C.c_char_p
        l.atns_obj_register.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, vp, C.c_uint32, C.c_uint8]
        l.atns_obj_copy.argtypes = [vp, vp, vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32]
        l.atns_mtrlxml_open.argtypes = [C.c_char_p, C.POINTER(vp)]
        l.atns_mtrlxml_close.argtypes = [vp]; l.atns_mtrlxml_close.restype = None
        l.atns_mtrlxml_count.argtypes = [vp]; l.atns_mtrlxml_count.restype = C.c_uint32
        l.atns_mtrlxml_name.argtypes = [vp, C.c_uint32]; l.atns_mtrlxml_name.restype = C.c_char_p
        l.atns_mtrlxml_type.argtypes = [vp, C.c_uint32]; l.atns_mtrlxml_type.restype = C.c_char_p
        l.atns_mtrlxml_param_count.argtypes = [vp, C.c_uint32]; l.atns_mtrlxml_param_count.restype = C.c_uint32
        l.atns_mtrlxml_param.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(_ParamInfo)]
        _bound = True
    return l


class ObjFile:
    """One parsed OBJ (+ MTL).  materials: list of dicts; register() -> vertices / triangles / meshes / objects."""

    def __init__(self, path):
        self._l = _lib()
        self._h = C.c_void_p()
        rc = self._l.atns_obj_open(path.encode(), C.byref(self._h))
        if rc != 0:
            raise IOError("atns_obj_open(%s) failed: %d" % (path, rc))
        self.materials = []
        for i in range(self._l.atns_obj_material_count(self._h)):
            m = _MaterialInfo()
            self._l.atns_obj_material(self._h, i, C.byref(m))
            self.materials.append(dict(name=m.name.decode(), diffuse=tuple(m.diffuse), emission=tuple(m.emission),
                                       diffuse_texname=m.diffuse_texname.decode(), bump_texname=m.bump_texname.decode()))
        self.shape_names = [self._l.atns_obj_shape_name(self._h, i).decode() for i in range(self._l.atns_obj_shape_count(self._h))]

    def close(self):
        if self._h:
            self._l.atns_obj_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def register(self, first_vertex=0, first_mesh_id=0, separate_objs=False, normal_on_the_fly=False, mtl_is_emissive=None,
                 default_is_emissive=False):
        em = np.zeros(max(1, len(self.materials)), np.uint8)
        if mtl_is_emissive is not None:
            em[:len(self.materials)] = np.asarray(mtl_is_emissive, np.uint8)[:len(self.materials)]
        rc = self._l.atns_obj_register(self._h, first_vertex, first_mesh_id, int(separate_objs), int(normal_on_the_fly),
                                       em.ctypes.data, len(em), int(default_is_emissive))
        if rc != 0:
            raise RuntimeError("atns_obj_register failed: %d" % rc)
        nv, nt = self._l.atns_obj_vertex_count(self._h), self._l.atns_obj_triangle_count(self._h)
        nm, no = self._l.atns_obj_mesh_count(self._h), self._l.atns_obj_object_count(self._h)
        pos = np.zeros((nv, 4), np.float32); nml = np.zeros((nv, 4), np.float32)
        tris = np.zeros(nt, OBJ_TRIANGLE); meshes = np.zeros(nm, OBJ_MESH); objs = np.zeros(no, OBJ_OBJECT)
        rc = self._l.atns_obj_copy(self._h, pos.ctypes.data, nml.ctypes.data, nv, tris.ctypes.data, nt, meshes.ctypes.data, nm, objs.ctypes.data, no)
        if rc != 0:
            raise RuntimeError("atns_obj_copy failed: %d" % rc)
        return pos, nml, tris, meshes, objs


def load_material_xml(path):
    """[(name, type, [(param, kind, value, text)])] in file order; kind 0 = vec3, 1 = texture file name, 2 = float, -1 = unknown."""
    l = _lib()
    h = C.c_void_p()
    rc = l.atns_mtrlxml_open(path.encode(), C.byref(h))
    if rc != 0:
        raise IOError("atns_mtrlxml_open(%s) failed: %d" % (path, rc))
    out = []
    try:
        for i in range(l.atns_mtrlxml_count(h)):
            params = []
            for k in range(l.atns_mtrlxml_param_count(h, i)):
                p = _ParamInfo()
                l.atns_mtrlxml_param(h, i, k, C.byref(p))
                params.append((p.name.decode(), int(p.kind), tuple(p.value), p.text.decode()))
            out.append((l.atns_mtrlxml_name(h, i).decode(), l.atns_mtrlxml_type(h, i).decode(), params))
    finally:
        l.atns_mtrlxml_close(h)
    return out
