"""Host-side handle over the C-ABI, named after the reference's GPU seam.

`PathTracing` exposes the calls an aten application makes on idaten::PathTracing
(src/libidaten/kernel/renderer.h:35-111, src/libidaten/kernel/pathtracing.cpp:23-153):
UpdateSceneData, updateCamera, render, reset -- every one a thin forward into libaten_amd.so.
"""
import ctypes as C

import numpy as np

from . import layout as L
from ._lib import Destination, K_NAMES, lib


class AtenAmdError(RuntimeError):
    pass


class PathTracing:
    def __init__(self, device=0):
        self._l = lib()
        self._ctx = C.c_void_p()
        rc = self._l.atn_create(C.byref(self._ctx), device)
        if rc != 0:
            raise AtenAmdError("atn_create failed (%d): no usable HIP device; libaten_amd has no CPU fallback" % rc)
        if self._l.atn_sizeof_scene_desc() != C.sizeof(L.SceneDesc):
            raise AtenAmdError("atn_scene_desc ABI mismatch")
        if self._l.atn_sizeof_destination() != C.sizeof(Destination):
            raise AtenAmdError("atn_destination ABI mismatch")
        self.width = self.height = 0

    def close(self):
        if self._ctx:
            self._l.atn_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise AtenAmdError("%s (status %d)" % (self._l.atn_last_error(self._ctx).decode(), rc))

    # ---- the reference's renderer surface
    def UpdateSceneData(self, scene):
        self._check(self._l.atn_upload_scene(self._ctx, C.cast(scene.ref(), C.c_void_p)))

    def updateBVH(self, scene, with_matrices=True):
        """idaten::Renderer::updateBVH (renderer.cpp:133-153): objects, matrices and the top layer of `scene`
        (a FlatScene whose bottom-level lists are the ones already uploaded).  with_matrices=False: the reference's
        `mtxs.empty()` form -- new objects and top layer, the uploaded matrices stay."""
        a = scene.arrays
        objs = np.ascontiguousarray(a["objects"])
        mtx = np.ascontiguousarray(a["matrices"]) if with_matrices else np.zeros((0, 4, 4), np.float32)
        top = np.ascontiguousarray(a["bvh_lists"][0])
        self._check(self._l.atn_update_tlas(self._ctx, objs.ctypes.data, len(objs), mtx.ctypes.data if len(mtx) else None,
                                            len(mtx), top.ctypes.data, len(top)))

    # ---- dynamic geometry (the reference's deformation renderer, src/deformation_renderer/main.cpp:636-710)
    def updateGeometry(self, vtx_pos=None, vtx_nml=None, vtx_offset=0, triangles=None, tri_offset=0):
        """idaten::Renderer::updateGeometry (renderer.cpp:155-215): overwrite a vertex / triangle range of the scene."""
        from . import layout as L
        pos = None if vtx_pos is None else np.ascontiguousarray(vtx_pos, np.float32).reshape(-1, 4)
        nml = None if vtx_nml is None else np.ascontiguousarray(vtx_nml, np.float32).reshape(-1, 4)
        if pos is not None and nml is not None and len(pos) != len(nml):
            raise ValueError("updateGeometry: %d positions but %d normals (the C ABI takes ONE vertex count for both arrays)" % (len(pos), len(nml)))
        nv = len(pos) if pos is not None else (len(nml) if nml is not None else 0)
        tr = None if triangles is None else np.ascontiguousarray(triangles, L.TRIANGLE_PARAM)
        self._check(self._l.atn_update_geometry(self._ctx, pos.ctypes.data if pos is not None else None, nml.ctypes.data if nml is not None else None,
                               nv, vtx_offset, tr.ctypes.data if tr is not None else None, len(tr) if tr is not None else 0, tri_offset))

    def lbvh_rebuild_list(self, list_index, tri_offset, n_triangles, bbox_min, bbox_max):
        """idaten::LBVHBuilder::build into the renderer's node list (LBVHBuilder.cu:700-810), on the device."""
        f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
        self._check(self._l.atn_lbvh_rebuild_list(self._ctx, list_index, tri_offset, n_triangles, f3(bbox_min), f3(bbox_max)))

    def lbvh_build(self, triangles, bbox_min, bbox_max, vtx_pos, tri_id_offset=0, vtx_offset=0, with_keys=False):
        """LBVHBuilder::build(..., threadedBvhNodes) (LBVHBuilder.cu:812-833): ThreadedBvhNode[2 n - 1] in the reference's order."""
        from . import layout as L
        tr = np.ascontiguousarray(triangles, L.TRIANGLE_PARAM)
        pos = np.ascontiguousarray(vtx_pos, np.float32).reshape(-1, 4)
        n = len(tr)
        out = np.zeros(max(2 * n - 1, 1), L.BVH_NODE)
        codes = np.zeros(max(n, 1), np.uint32); idx = np.zeros(max(n, 1), np.uint32)
        f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
        self._check(self._l.atn_lbvh_build(self._ctx, tr.ctypes.data, n, tri_id_offset, f3(bbox_min), f3(bbox_max), pos.ctypes.data, len(pos),
                                           vtx_offset, out.ctypes.data, codes.ctypes.data, idx.ctypes.data))
        return (out, codes, idx) if with_keys else out

    def scene_device_arrays(self):
        """(vtx_pos, vtx_nml, triangles) device addresses of the uploaded scene."""
        p = [C.c_void_p() for _ in range(3)]
        self._check(self._l.atn_scene_device_arrays(self._ctx, C.byref(p[0]), C.byref(p[1]), C.byref(p[2])))
        return tuple(x.value for x in p)

    def updateCamera(self, cam):
        self._check(self._l.atn_update_camera(self._ctx, cam.ctypes.data))

    def initSampler(self, width, height, seed=0):
        self._check(self._l.atn_init_sampler(self._ctx, width, height, seed))

    def setRandom(self, seeds):
        seeds = np.ascontiguousarray(seeds, np.uint32)
        self._check(self._l.atn_set_random(self._ctx, seeds.ctypes.data, len(seeds)))

    def getRandom(self):
        out = np.zeros(self._l.atn_random_count(self._ctx), np.uint32)
        if len(out):
            self._check(self._l.atn_get_random(self._ctx, out.ctypes.data, len(out)))
        return out

    def setScreenShard(self, rank, world):
        self._check(self._l.atn_set_screen_shard(self._ctx, rank, world))

    def render(self, width, height, max_depth=5, rr_depth=3, spp=1, frame=0, progressive=True,
               break_on_terminate=True, download=True, count_stats=False, profile=False):
        d = Destination(width, height, max_depth, rr_depth, spp, frame, int(progressive),
                        int(break_on_terminate), int(count_stats), int(profile))
        out = np.empty((height, width, 4), np.float32) if download else None
        self._check(self._l.atn_render(self._ctx, C.byref(d), out.ctypes.data if download else None))
        self.width, self.height = width, height
        return out

    def set_upload_options(self, anyhit_twin=None, anyhit_twin_dirs=None, node_layout=None, planar_lights=None):
        """How the next UpdateSceneData lays the scene out (atn_set_upload_options); None leaves an option as it is."""
        v = lambda x: -1 if x is None else int(x)
        self._check(self._l.atn_set_upload_options(self._ctx, v(anyhit_twin), v(anyhit_twin_dirs), v(node_layout), v(planar_lights)))

    def set_shade_math(self, relaxed):
        """False (default): the parity path; True: the shade kernel under the reference GPU build's --use_fast_math rules (opt-in)."""
        self._check(self._l.atn_set_shade_math(self._ctx, int(relaxed)))

    def set_regeneration(self, on):
        """Path regeneration (include/aten_amd.h): the samples of a frame / the frames of a burst share one pool of path slots.
        Off by default.  Measured guidance: switch it on for render_burst of >= 2 multi-sample frames in the
        break-on-terminate sample loop (1080p 8 spp: 1.14-1.23 x over four serial frames in flight, 1.7-2.0 x for a caller with one
        frame in flight) and for one-frame-in-flight shards of scenes whose paths differ in length; leave it off at 1 spp with
        frames in flight and when every sample is traced (0.72-0.92 x)."""
        self._check(self._l.atn_set_regeneration(self._ctx, int(on)))

    def render_burst(self, width, height, n_frames, max_depth=5, rr_depth=3, spp=1, frame=0, progressive=True,
                     break_on_terminate=True, download=True, profile=False):
        """n_frames consecutive frames (frame, frame + 1, ...) in one call; the film after the last one."""
        d = Destination(width, height, max_depth, rr_depth, spp, frame, int(progressive), int(break_on_terminate), 0, int(profile))
        out = np.empty((height, width, 4), np.float32) if download else None
        self._check(self._l.atn_render_burst(self._ctx, C.byref(d), n_frames, out.ctypes.data if download else None))
        self.width, self.height = width, height
        return out

    def regen_stage_counts(self):
        """(closest-hit rays, shadow rays) per launch of the last regenerated burst."""
        n = C.c_uint32(0)
        self._check(self._l.atn_regen_stage_counts(self._ctx, None, None, 0, C.byref(n)))
        q = np.zeros(n.value, np.uint32); sh = np.zeros(n.value, np.uint32)
        if n.value:
            self._check(self._l.atn_regen_stage_counts(self._ctx, q.ctypes.data, sh.ctypes.data, n.value, C.byref(n)))
        return q, sh

    def set_path_batches(self, n):
        self._check(self._l.atn_set_path_batches(self._ctx, n))

    def set_frames_in_flight(self, n):
        self._check(self._l.atn_set_frames_in_flight(self._ctx, n))

    def side_stream_ptr(self):
        """hipStream_t for the caller's own work beside the frames in flight (atn_side_stream)"""
        p = self._l.atn_side_stream(self._ctx)
        if not p:
            raise AtenAmdError("atn_side_stream failed")
        return p

    def bank_streams(self):
        """(streams replaced by the queue probe so far, every pair of bank streams measured to run side by side)"""
        sw, cc = C.c_int32(0), C.c_int32(0)
        self._check(self._l.atn_bank_streams(self._ctx, C.byref(sw), C.byref(cc)))
        self.side_stream_concurrent = bool(cc.value & 2)    # the side stream (if handed out) runs beside every bank stream
        return sw.value, bool(cc.value & 1)

    def set_sampling_options(self, ibl_importance=False, tex_bilinear=False):
        self._check(self._l.atn_set_sampling_options(self._ctx, int(ibl_importance), int(tex_bilinear)))

    def sample_texture(self, texid, uv):
        uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        out = np.zeros((len(uv), 4), np.float32)
        self._check(self._l.atn_sample_texture(self._ctx, texid, len(uv), uv.ctypes.data, out.ctypes.data))
        return out

    def reset(self):
        self._check(self._l.atn_reset(self._ctx))

    # ---- results / instrumentation
    def synchronize(self):
        self._check(self._l.atn_synchronize(self._ctx))

    def film_device_ptr(self):
        return self._l.atn_film_device(self._ctx)

    def tile_device_ptr(self):
        return self._l.atn_tile_device(self._ctx)

    def tile_slots(self):
        return int(self._l.atn_tile_slots(self._ctx))

    def stream_ptr(self):
        return self._l.atn_stream(self._ctx)

    def assemble_tiles(self, gathered_dev_ptr, world, out_dev_ptr=None, stream_ptr=None):
        self._check(self._l.atn_assemble_tiles_on(self._ctx, gathered_dev_ptr, world, out_dev_ptr, stream_ptr))

    def download_film(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(self._l.atn_download_film(self._ctx, out.ctypes.data))
        return out

    def upload_film(self, film):
        """Resume from a film returned by download_film / render (running mean + sample count)."""
        film = np.ascontiguousarray(film, np.float32)
        h, w = film.shape[:2]
        self._check(self._l.atn_upload_film(self._ctx, w, h, film.ctypes.data))
        self.width, self.height = w, h

    def path_cost(self):
        """uint32 [h, w, 2]: BVH node visits and triangle tests per pixel of the last frame rendered with count_stats=True."""
        out = np.zeros((self.height, self.width, 2), np.uint32)
        self._check(self._l.atn_download_path_cost(self._ctx, out.ctypes.data))
        return out

    def anyhit_twins(self):
        """atn_anyhit_twins: bottom-level lists that have an any-hit twin right now."""
        return int(self._l.atn_anyhit_twins(self._ctx))

    def planar_area_lights(self):
        """atn_planar_area_lights: area lights whose shadow rays may stop at the first hit nearer than the light."""
        return int(self._l.atn_planar_area_lights(self._ctx))

    def stats(self):
        s = np.zeros(8, np.uint64)
        self._check(self._l.atn_get_stats(self._ctx, s.ctypes.data))
        return dict(closest_rays=int(s[0]), shadow_rays=int(s[1]), hits=int(s[2]),
                    closest_nodes=int(s[3]), closest_tris=int(s[4]), shadow_nodes=int(s[5]), shadow_tris=int(s[6]))

    # ---- SVGF (aten::SVGFRenderer)
    SVGF_BUFFERS = dict(normal_depth=0, albedo_meshid=1, color_variance=2, moment_temporalweight=3,
                        prev_normal_depth=4, prev_albedo_meshid=5, prev_color_variance=6, prev_moment_temporalweight=7,
                        temporary_color=8, motion_depth=9, primary_position=10, atrous0=11, atrous1=12, output=13,
                        contribs=14)

    def svgf_render(self, width, height, max_depth=5, rr_depth=3, spp=1, frame=0, compute_motion=False, stages=False,
                    download=True, profile=False):
        """SVGFRenderer::OnRender.  Returns the filtered frame [h, w, 4] (and the three intermediate puts)."""
        d = Destination(width, height, max_depth, rr_depth, spp, frame, 0, 1, 0, int(profile))
        out = np.empty((height, width, 4), np.float32) if download else None
        st = np.empty((3, height, width, 4), np.float32) if stages else None
        self._check(self._l.atn_svgf_render(self._ctx, C.byref(d), 1 if compute_motion else 0,
                                            out.ctypes.data if download else None, st.ctypes.data if stages else None))
        self.width, self.height = width, height
        return (out, st) if stages else out

    def svgf_denoise(self, width, height, frame=0, compute_motion=False, stages=False, download=True, profile=False):
        """The filter passes of OnRender on the buffers as they stand (svgf_upload / a previous path pass)."""
        d = Destination(width, height, 1, 1, 1, frame, 0, 1, 0, int(profile))
        out = np.empty((height, width, 4), np.float32) if download else None
        st = np.empty((3, height, width, 4), np.float32) if stages else None
        self._check(self._l.atn_svgf_denoise(self._ctx, C.byref(d), 1 if compute_motion else 0,
                                             out.ctypes.data if download else None, st.ctypes.data if stages else None))
        self.width, self.height = width, height
        return (out, st) if stages else out

    def svgf_upload(self, name, data):
        data = np.ascontiguousarray(data, np.float32)
        h, w = data.shape[:2]
        self._check(self._l.atn_svgf_upload(self._ctx, self.SVGF_BUFFERS[name], w, h, data.ctypes.data))
        self.width, self.height = w, h

    def svgf_set_motion_depth(self, md):
        md = np.ascontiguousarray(md, np.float32).reshape(-1, 4)
        self._check(self._l.atn_svgf_set_motion_depth(self._ctx, md.ctypes.data, len(md)))

    def svgf_reset(self):
        self._check(self._l.atn_svgf_reset(self._ctx))

    def svgf_set_dilate_temporal_weight(self, on):
        self._check(self._l.atn_svgf_set_dilate_temporal_weight(self._ctx, int(on)))

    def svgf_set_atrous_iterations(self, n):
        self._check(self._l.atn_svgf_set_atrous_iterations(self._ctx, n))

    def svgf_buffer(self, name):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(self._l.atn_svgf_download(self._ctx, self.SVGF_BUFFERS[name], out.ctypes.data))
        return out

    def kernel_times(self):
        ms = np.zeros(len(K_NAMES), np.float32); n = np.zeros(len(K_NAMES), np.uint32)
        self._check(self._l.atn_get_kernel_times(self._ctx, ms.ctypes.data, n.ctypes.data))
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(K_NAMES)}

    def reset_kernel_times(self):
        self._check(self._l.atn_reset_kernel_times(self._ctx))

    # ---- stage entry points (parity tests)
    def generate_paths(self, width, height, sample=0, frame=0):
        rays = np.zeros(width * height, L.RAY)
        self._check(self._l.atn_generate_paths(self._ctx, width, height, sample, frame, rays.ctypes.data))
        return rays

    def trace_closest(self, rays, t_min=1e-9, t_max=float(np.finfo(np.float32).max), stats=False):
        rays = np.ascontiguousarray(rays)
        out = np.zeros(len(rays), L.INTERSECTION)
        st = np.zeros(2, np.uint64)
        self._check(self._l.atn_trace_closest(self._ctx, rays.ctypes.data, len(rays), t_min, t_max,
                                              out.ctypes.data, st.ctypes.data if stats else None))
        return (out, st) if stats else out

    def cmj_samples(self, index, dimension, scramble, n):
        out = np.zeros(n, np.float32)
        self._check(self._l.atn_cmj_samples(self._ctx, index, dimension, scramble, n, out.ctypes.data))
        return out

    def cmj_batch(self, index, dimension, scramble, draws=1):
        index = np.ascontiguousarray(index, np.uint32); dimension = np.ascontiguousarray(dimension, np.uint32)
        scramble = np.ascontiguousarray(scramble, np.uint32)
        if not (len(index) == len(dimension) == len(scramble)):
            raise ValueError("index, dimension and scramble must have one entry per triple")
        out = np.zeros((len(index), draws), np.float32)
        self._check(self._l.atn_cmj_batch(self._ctx, len(index), index.ctypes.data, dimension.ctypes.data,
                                          scramble.ctypes.data, draws, out.ctypes.data))
        return out

    LIBM_KINDS = ["sinf", "cosf", "atanf", "acosf", "atan2f", "logf", "expf", "powf", "sqrtf", "div", "inversesqrt"]

    def libm_probe(self, kind, a, b=None):
        """The device build's math library on arrays (atn_libm_probe; kind: a name from LIBM_KINDS)."""
        a = np.ascontiguousarray(a, np.float32)
        b = np.ones_like(a) if b is None else np.ascontiguousarray(b, np.float32)
        out = np.zeros_like(a)
        self._check(self._l.atn_libm_probe(self._ctx, self.LIBM_KINDS.index(kind), len(a), a.ctypes.data, b.ctypes.data, out.ctypes.data))
        return out

    def ray_offset(self, origins, normals):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3); n = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
        out = np.zeros_like(o)
        self._check(self._l.atn_ray_offset(self._ctx, len(o), o.ctypes.data, n.ctypes.data, out.ctypes.data))
        return out

    def material_table(self, mtrl_id, nrm, wi, index, scramble, uv, dimension=None):
        n = len(nrm)
        dim = np.ascontiguousarray(np.broadcast_to(np.asarray(dimension, np.uint32), (n,))) if dimension is not None else None
        nrm = np.ascontiguousarray(nrm, np.float32); wi = np.ascontiguousarray(wi, np.float32)
        index = np.ascontiguousarray(index, np.uint32); scramble = np.ascontiguousarray(scramble, np.uint32)
        uv = np.ascontiguousarray(uv, np.float32)
        s = np.zeros((n, 7), np.float32); e = np.zeros((n, 5), np.float32)
        self._check(self._l.atn_material_table(self._ctx, mtrl_id, n, nrm.ctypes.data, wi.ctypes.data,
                                               index.ctypes.data, dim.ctypes.data if dim is not None else None, scramble.ctypes.data, uv.ctypes.data,
                                               s.ctypes.data, e.ctypes.data))
        return s, e

    def material_eval(self, mtrl_id, nrm, wi, wo, uv=None):
        """samplePDF / sampleBSDF at given outgoing directions -> [n, 5] {pdf, bsdf.xyz, bsdf's own pdf}."""
        n = len(wo)
        b = lambda a, k: np.ascontiguousarray(np.broadcast_to(np.asarray(a, np.float32), (n, k)))
        nrm, wi, wo = b(nrm, 3), b(wi, 3), b(wo, 3)
        uv = b(uv if uv is not None else (0.5, 0.5), 2)
        e = np.zeros((n, 5), np.float32)
        self._check(self._l.atn_material_eval(self._ctx, mtrl_id, n, nrm.ctypes.data, wi.ctypes.data, wo.ctypes.data, uv.ctypes.data, e.ctypes.data))
        return e

    def compact(self, flags):
        flags = np.ascontiguousarray(flags, np.int32)
        out = np.zeros(max(1, len(flags)), np.int32)
        cnt = C.c_uint32()
        self._check(self._l.atn_compact(self._ctx, flags.ctypes.data, len(flags), out.ctypes.data, C.byref(cnt)))
        return out[:cnt.value]

    def compact2(self, flags_a, flags_b=None, grid_blocks=0):
        """The renderer's two-queue block append over caller flags; returns the queues in device order."""
        fa = np.ascontiguousarray(flags_a, np.int32)
        fb = np.ascontiguousarray(flags_b, np.int32) if flags_b is not None else None
        n = len(fa)
        oa = np.zeros(max(1, n), np.int32); ob = np.zeros(max(1, n), np.int32)
        ca = C.c_uint32(); cb = C.c_uint32()
        self._check(self._l.atn_compact2(self._ctx, fa.ctypes.data, fb.ctypes.data if fb is not None else None, n, grid_blocks,
                                         oa.ctypes.data, C.byref(ca), ob.ctypes.data if fb is not None else None,
                                         C.byref(cb) if fb is not None else None))
        return oa[:ca.value], (ob[:cb.value] if fb is not None else None)


class MultiGpuPathTracing:
    """The node-wide renderer (atn_mgpu_*, include/aten_amd.h): same calls as PathTracing, every visible GPU
    (or the given shard list; an ordinal may repeat) behind them."""

    def __init__(self, devices=None):
        self._l = lib()
        self._mg = C.c_void_p()
        if devices is None:
            rc = self._l.atn_mgpu_create(C.byref(self._mg), None, 0)
        elif isinstance(devices, int):
            rc = self._l.atn_mgpu_create(C.byref(self._mg), None, devices)
        else:
            arr = (C.c_int32 * len(devices))(*devices)
            rc = self._l.atn_mgpu_create(C.byref(self._mg), arr, len(devices))
        if rc != 0:
            raise AtenAmdError("atn_mgpu_create failed (%d)" % rc)
        self.width = self.height = 0

    def close(self):
        if self._mg:
            self._l.atn_mgpu_destroy(self._mg)
            self._mg = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise AtenAmdError("%s (status %d)" % (self._l.atn_mgpu_last_error(self._mg).decode(), rc))

    def shard_count(self):
        return int(self._l.atn_mgpu_shard_count(self._mg))

    def shard_devices(self):
        return [int(self._l.atn_mgpu_shard_device(self._mg, i)) for i in range(self.shard_count())]

    def UpdateSceneData(self, scene):
        self._check(self._l.atn_mgpu_upload_scene(self._mg, C.cast(scene.ref(), C.c_void_p)))

    def updateBVH(self, scene):
        a = scene.arrays
        objs = np.ascontiguousarray(a["objects"]); mtx = np.ascontiguousarray(a["matrices"]); top = np.ascontiguousarray(a["bvh_lists"][0])
        self._check(self._l.atn_mgpu_update_tlas(self._mg, objs.ctypes.data, len(objs), mtx.ctypes.data if len(mtx) else None,
                                                 len(mtx), top.ctypes.data, len(top)))

    # ---- dynamic geometry (the reference's deformation renderer, src/deformation_renderer/main.cpp:636-710)
    def updateGeometry(self, vtx_pos=None, vtx_nml=None, vtx_offset=0, triangles=None, tri_offset=0):
        """idaten::Renderer::updateGeometry (renderer.cpp:155-215): overwrite a vertex / triangle range of the scene."""
        from . import layout as L
        pos = None if vtx_pos is None else np.ascontiguousarray(vtx_pos, np.float32).reshape(-1, 4)
        nml = None if vtx_nml is None else np.ascontiguousarray(vtx_nml, np.float32).reshape(-1, 4)
        if pos is not None and nml is not None and len(pos) != len(nml):
            raise ValueError("updateGeometry: %d positions but %d normals (the C ABI takes ONE vertex count for both arrays)" % (len(pos), len(nml)))
        nv = len(pos) if pos is not None else (len(nml) if nml is not None else 0)
        tr = None if triangles is None else np.ascontiguousarray(triangles, L.TRIANGLE_PARAM)
        self._check(self._l.atn_mgpu_update_geometry(self._mg, pos.ctypes.data if pos is not None else None, nml.ctypes.data if nml is not None else None,
                               nv, vtx_offset, tr.ctypes.data if tr is not None else None, len(tr) if tr is not None else 0, tri_offset))

    def lbvh_rebuild_list(self, list_index, tri_offset, n_triangles, bbox_min, bbox_max):
        """idaten::LBVHBuilder::build into the renderer's node list (LBVHBuilder.cu:700-810), on the device."""
        f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
        self._check(self._l.atn_mgpu_lbvh_rebuild_list(self._mg, list_index, tri_offset, n_triangles, f3(bbox_min), f3(bbox_max)))

    def updateCamera(self, cam):
        self._check(self._l.atn_mgpu_update_camera(self._mg, cam.ctypes.data))

    def initSampler(self, width, height, seed=0):
        self._check(self._l.atn_mgpu_init_sampler(self._mg, width, height, seed))

    def render(self, width, height, max_depth=5, rr_depth=3, spp=1, frame=0, progressive=True,
               break_on_terminate=True, download=True):
        d = Destination(width, height, max_depth, rr_depth, spp, frame, int(progressive), int(break_on_terminate), 0, 0)
        out = np.empty((height, width, 4), np.float32) if download else None
        self._check(self._l.atn_mgpu_render(self._mg, C.byref(d), out.ctypes.data if download else None))
        self.width, self.height = width, height
        return out

    def set_regeneration(self, on):
        self._check(self._l.atn_mgpu_set_regeneration(self._mg, int(on)))

    def render_burst(self, width, height, n_frames, max_depth=5, rr_depth=3, spp=1, frame=0, progressive=True,
                     break_on_terminate=True, download=True):
        """n_frames consecutive frames on every shard, one exchange of the tiles at the end."""
        d = Destination(width, height, max_depth, rr_depth, spp, frame, int(progressive), int(break_on_terminate), 0, 0)
        out = np.empty((height, width, 4), np.float32) if download else None
        self._check(self._l.atn_mgpu_render_burst(self._mg, C.byref(d), n_frames, out.ctypes.data if download else None))
        self.width, self.height = width, height
        return out

    def reset(self):
        self._check(self._l.atn_mgpu_reset(self._mg))

    def synchronize(self):
        self._check(self._l.atn_mgpu_synchronize(self._mg))

    def set_frames_in_flight(self, n):
        self._check(self._l.atn_mgpu_set_frames_in_flight(self._mg, n))

    def film_device_ptr(self):
        return self._l.atn_mgpu_film_device(self._mg)

    def download_film(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(self._l.atn_mgpu_download_film(self._mg, out.ctypes.data))
        return out
