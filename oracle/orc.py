"""ctypes binding of oracle/liboracle.so.  TEST INFRASTRUCTURE ONLY: import from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never from aten_amd/."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_lib = None


class Destination(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("maxDepth", C.c_int32),
                ("russianRouletteDepth", C.c_int32), ("sample", C.c_int32), ("frame", C.c_uint32),
                ("progressive", C.c_int32), ("nthreads", C.c_int32)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            subprocess.check_call(["make", "-C", _HERE])
        l = C.CDLL(_LIB)
        l.orc_pixel_width_at_distance.restype = C.c_float
        l.orc_pixel_width_at_distance.argtypes = [C.c_void_p, C.c_float]
        l.orc_num_procs.restype = C.c_int
        _lib = l
    return _lib


def init_sampler(w, h, seed=0):
    out = np.zeros(w * h, np.uint32)
    lib().orc_init_sampler(C.c_void_p(out.ctypes.data), w, h, seed)
    return out


def cmj_samples(index, dimension, scramble, n):
    out = np.zeros(n, np.float32)
    lib().orc_cmj_samples(C.c_uint32(index), C.c_uint32(dimension), C.c_uint32(scramble), n, C.c_void_p(out.ctypes.data))
    return out


def create_camera(pos, at, vfov, width, height, up=(0, 1, 0), znear=0.1, zfar=10000.0):
    from aten_amd import layout as L
    cam = np.zeros((), L.CAMERA_PARAM)
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    lib().orc_create_camera(C.c_void_p(cam.ctypes.data), f3(pos), f3(at), f3(up), C.c_float(vfov),
                            C.c_float(znear), C.c_float(zfar), C.c_int32(width), C.c_int32(height))
    return cam


def pixel_width_at_distance(cam, dist):
    return float(lib().orc_pixel_width_at_distance(C.c_void_p(cam.ctypes.data), C.c_float(dist)))


def ray_offset(o, n):
    o = np.ascontiguousarray(o, np.float32); n = np.ascontiguousarray(n, np.float32)
    out = np.zeros_like(o)
    lib().orc_ray_offset(C.c_void_p(o.ctypes.data), C.c_void_p(n.ctypes.data), len(o), C.c_void_p(out.ctypes.data))
    return out


def generate_paths(cam, seeds, w, h, sample, frame):
    from aten_amd import layout as L
    rays = np.zeros(w * h, L.RAY)
    lib().orc_generate_paths(C.c_void_p(cam.ctypes.data), C.c_void_p(seeds.ctypes.data), C.c_uint32(len(seeds)),
                             w, h, sample, C.c_uint32(frame), C.c_void_p(rays.ctypes.data))
    return rays


def trace_closest(scene, rays, t_min=1e-9, t_max=np.finfo(np.float32).max):
    from aten_amd import layout as L
    out = np.zeros(len(rays), L.INTERSECTION)
    stats = np.zeros(2, np.uint64)
    lib().orc_trace_closest(scene.ref(), C.c_void_p(rays.ctypes.data), C.c_uint32(len(rays)),
                            C.c_float(t_min), C.c_float(t_max), C.c_void_p(out.ctypes.data), C.c_void_p(stats.ctypes.data))
    return out, stats


def evaluate_hits(scene, rays, isects):
    out = np.zeros((len(rays), 9), np.float32)
    lib().orc_evaluate_hits(scene.ref(), C.c_void_p(rays.ctypes.data), C.c_void_p(isects.ctypes.data),
                            C.c_uint32(len(rays)), C.c_void_p(out.ctypes.data))
    return out


def material_table(scene, mtrl_id, nrm, wi, index, scramble, uv):
    n = len(nrm)
    nrm = np.ascontiguousarray(nrm, np.float32); wi = np.ascontiguousarray(wi, np.float32)
    index = np.ascontiguousarray(index, np.uint32); scramble = np.ascontiguousarray(scramble, np.uint32)
    uv = np.ascontiguousarray(uv, np.float32)
    s = np.zeros((n, 7), np.float32); e = np.zeros((n, 5), np.float32)
    lib().orc_material_table(scene.ref(), C.c_int32(mtrl_id), C.c_uint32(n), C.c_void_p(nrm.ctypes.data),
                             C.c_void_p(wi.ctypes.data), C.c_void_p(index.ctypes.data), C.c_void_p(scramble.ctypes.data),
                             C.c_void_p(uv.ctypes.data), C.c_void_p(s.ctypes.data), C.c_void_p(e.ctypes.data))
    return s, e


def render(scene, cam, seeds, width, height, max_depth=5, rr_depth=3, spp=1, frame=0,
           film=None, progressive=True, nthreads=0, counters=False):
    """aten::PathTracing::render on the CPU oracle.  Returns film [h, w, 4] (row 0 = bottom)."""
    if film is None:
        film = np.zeros((height, width, 4), np.float32)
    d = Destination(width, height, max_depth, rr_depth, spp, frame, 1 if progressive else 0, nthreads)
    cnt = np.zeros(5, np.uint64)
    lib().orc_render(scene.ref(), C.c_void_p(cam.ctypes.data), C.c_void_p(seeds.ctypes.data), C.c_uint32(len(seeds)),
                     C.byref(d), C.c_void_p(film.ctypes.data), C.c_void_p(cnt.ctypes.data) if counters else None)
    return (film, cnt) if counters else film
