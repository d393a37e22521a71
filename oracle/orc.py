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


def cmj_samples2d(index, dimension, scramble, n):
    out = np.zeros((n, 2), np.float32)
    lib().orc_cmj_samples2d(C.c_uint32(index), C.c_uint32(dimension), C.c_uint32(scramble), n, C.c_void_p(out.ctypes.data))
    return out


def cmj_batch(index, dimension, scramble, draws=1):
    index = np.ascontiguousarray(index, np.uint32); dimension = np.ascontiguousarray(dimension, np.uint32)
    scramble = np.ascontiguousarray(scramble, np.uint32)
    out = np.zeros((len(index), draws), np.float32)
    lib().orc_cmj_batch(len(index), C.c_void_p(index.ctypes.data), C.c_void_p(dimension.ctypes.data),
                        C.c_void_p(scramble.ctypes.data), draws, C.c_void_p(out.ctypes.data))
    return out


def math_kat(kind, a, b=None, c=None):
    a = np.ascontiguousarray(a, np.float32)
    b = np.zeros_like(a) if b is None else np.ascontiguousarray(b, np.float32)
    c = np.zeros_like(a) if c is None else np.ascontiguousarray(c, np.float32)
    out = np.zeros_like(a)
    lib().orc_math_kat(kind, len(a), C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), C.c_void_p(c.ctypes.data),
                       C.c_void_p(out.ctypes.data))
    return out


LIBM_KINDS = ["sinf", "cosf", "atanf", "acosf", "atan2f", "logf", "expf", "powf", "sqrtf", "div", "inversesqrt"]


def libm_probe(kind, a, b=None):
    """The oracle build's libm on arrays (kind: a name from LIBM_KINDS)."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ones_like(a) if b is None else np.ascontiguousarray(b, np.float32)
    out = np.zeros_like(a)
    lib().orc_libm_probe(LIBM_KINDS.index(kind), len(a), C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), C.c_void_p(out.ctypes.data))
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


def render_cost(scene, cam, seeds, width, height, max_depth=5, rr_depth=3, spp=1, frame=0):
    """Per-pixel {BVH node visits, triangle tests} of one frame: uint32 [h, w, 2] (and the film)."""
    film = np.zeros((height, width, 4), np.float32)
    cost = np.zeros((height, width, 2), np.uint32)
    d = Destination(width, height, max_depth, rr_depth, spp, frame, 1, 0)
    lib().orc_render_cost(scene.ref(), C.c_void_p(cam.ctypes.data), C.c_void_p(seeds.ctypes.data), C.c_uint32(len(seeds)),
                          C.byref(d), C.c_void_p(film.ctypes.data), None, C.c_void_p(cost.ctypes.data))
    return cost, film


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


def set_sampling_options(ibl_importance=False, tex_bilinear=False):
    """The product's optional samplers (atn_set_sampling_options) on the CPU twin.  Global: reset after use."""
    lib().orc_set_sampling_options(C.c_int32(int(ibl_importance)), C.c_int32(int(tex_bilinear)))


def sample_texture(scene, texid, uv):
    uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
    out = np.zeros((len(uv), 4), np.float32)
    lib().orc_sample_texture(scene.ref(), C.c_int32(texid), C.c_uint32(len(uv)), C.c_void_p(uv.ctypes.data), C.c_void_p(out.ctypes.data))
    return out


def lbvh_build(tris, bbox_min, bbox_max, vtx_pos, tri_id_offset=0, vtx_offset=0, with_keys=False):
    """idaten::LBVHBuilder::build on the CPU oracle (oracle/orc_lbvh.h): ThreadedBvhNode[2 n - 1] in the reference's
    own order (inner nodes 0 .. n-2, leaves n-1 .. 2n-2)."""
    from aten_amd import layout as L
    tris = np.ascontiguousarray(tris, L.TRIANGLE_PARAM)
    vtx_pos = np.ascontiguousarray(vtx_pos, np.float32).reshape(-1, 4)
    n = len(tris)
    out = np.zeros(2 * n - 1, L.BVH_NODE)
    codes = np.zeros(n, np.uint32); idx = np.zeros(n, np.uint32)
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    rc = lib().orc_lbvh_build(C.c_void_p(tris.ctypes.data), C.c_uint32(n), C.c_int32(tri_id_offset), f3(bbox_min), f3(bbox_max),
                              C.c_void_p(vtx_pos.ctypes.data), C.c_int32(vtx_offset), C.c_void_p(out.ctypes.data),
                              C.c_void_p(codes.ctypes.data), C.c_void_p(idx.ctypes.data))
    if rc != 0:
        raise ValueError("orc_lbvh_build: needs at least two triangles")
    return (out, codes, idx) if with_keys else out


def lbvh_hierarchy(sorted_keys):
    """buildTree (LBVHBuilder.cu:299-350) on sorted keys: (left, right, parent) of the 2 n - 1 nodes."""
    keys = np.ascontiguousarray(sorted_keys, np.uint32)
    n = len(keys)
    l = np.zeros(2 * n - 1, np.int32); r = np.zeros_like(l); p = np.zeros_like(l)
    rc = lib().orc_lbvh_hierarchy(C.c_void_p(keys.ctypes.data), C.c_uint32(n), C.c_void_p(l.ctypes.data),
                                  C.c_void_p(r.ctypes.data), C.c_void_p(p.ctypes.data))
    if rc != 0:
        raise ValueError("orc_lbvh_hierarchy: needs at least two keys")
    return l, r, p


def fill_basic_aovs(normal, p, w2c, albedo, meshid):
    """aten::FillBasicAOVs (renderer/aov.h:158-181) -> (normal_depth[4], albedo_meshid[4])."""
    f = lambda a, n: np.ascontiguousarray(a, np.float32).reshape(n)
    nd, am = np.zeros(4, np.float32), np.zeros(4, np.float32)
    a, b, c, d = f(normal, 3), f(p, 3), f(w2c, 16), f(albedo, 4)
    lib().orc_fill_basic_aovs(C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), C.c_void_p(c.ctypes.data), C.c_void_p(d.ctypes.data),
                              C.c_int32(meshid), C.c_void_p(nd.ctypes.data), C.c_void_p(am.ctypes.data))
    return nd, am


def fill_basic_aovs_if_hit_miss(bg):
    """aten::FillBasicAOVsIfHitMiss (renderer/aov.h:183-198)."""
    nd, am = np.zeros(4, np.float32), np.zeros(4, np.float32)
    b = np.ascontiguousarray(bg, np.float32).reshape(4)
    lib().orc_fill_basic_aovs_if_hit_miss(C.c_void_p(b.ctypes.data), C.c_void_p(nd.ctypes.data), C.c_void_p(am.ctypes.data))
    return nd, am


class Svgf:
    """aten::SVGFRenderer on the CPU oracle (oracle/orc_svgf.h): frame-persistent AOV / moment buffers."""
    BUFFERS = dict(normal_depth=0, albedo_meshid=1, color_variance=2, moment_temporalweight=3,
                   prev_normal_depth=4, prev_albedo_meshid=5, prev_color_variance=6, prev_moment_temporalweight=7,
                   temporary_color=8, motion_depth=9, primary_position=10, atrous0=11, atrous1=12)

    def __init__(self):
        l = lib()
        l.orc_svgf_create.restype = C.c_void_p
        l.orc_svgf_get_buffer.restype = C.c_int
        self._h = C.c_void_p(l.orc_svgf_create())
        self.n = 0

    def close(self):
        if self._h:
            lib().orc_svgf_destroy(self._h)
            self._h = None

    def set_atrous_iterations(self, n):
        lib().orc_svgf_set_atrous_iterations(self._h, C.c_int32(n))

    def set_dilate_temporal_weight(self, on):
        lib().orc_svgf_set_dilate_temporal_weight(self._h, C.c_int32(int(on)))

    def set_motion_depth(self, md):
        md = np.ascontiguousarray(md, np.float32).reshape(-1, 4)
        lib().orc_svgf_set_motion_depth(self._h, C.c_void_p(md.ctypes.data), C.c_uint32(len(md)))

    def render(self, scene, cam, seeds, width, height, max_depth=5, rr_depth=3, spp=1, frame=0, compute_motion=False,
               stages=False, nthreads=0):
        """Returns the final film [h, w, 4] (and, with stages=True, the three intermediate puts [3, h, w, 4])."""
        film = np.zeros((height, width, 4), np.float32)
        st = np.zeros((3, height, width, 4), np.float32) if stages else None
        d = Destination(width, height, max_depth, rr_depth, spp, frame, 0, nthreads)
        lib().orc_svgf_render(self._h, scene.ref(), C.c_void_p(cam.ctypes.data), C.c_void_p(seeds.ctypes.data),
                              C.c_uint32(len(seeds)), C.byref(d), C.c_int32(1 if compute_motion else 0),
                              C.c_void_p(film.ctypes.data), C.c_void_p(st.ctypes.data) if stages else None)
        self.n = width * height
        self.shape = (height, width, 4)
        return (film, st) if stages else film

    def buffer(self, name):
        out = np.zeros(self.shape, np.float32)
        r = lib().orc_svgf_get_buffer(self._h, C.c_int32(self.BUFFERS[name]), C.c_void_p(out.ctypes.data))
        if r < 0:
            raise ValueError(name)
        return out
