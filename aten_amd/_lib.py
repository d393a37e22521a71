"""ctypes binding of libaten_amd.so (include/aten_amd.h).

The HIP library is the product: there is no CPU fallback and no CPU checker is reachable from here.
If the shared object is missing or no GPU is present the calls fail loudly.
"""
import ctypes as C
import os

from . import build

_lib = None

K_NAMES = ["gen_path", "trace_closest", "shade", "trace_shadow", "accumulate_sample", "gather",
           "svgf_prepare", "svgf_temporal", "svgf_variance", "svgf_atrous", "trace_fused"]

SYMBOLS = [
    "atn_create", "atn_destroy", "atn_last_error", "atn_upload_scene", "atn_update_camera", "atn_update_tlas",
    "atn_init_sampler", "atn_set_random", "atn_set_screen_shard", "atn_render", "atn_reset", "atn_set_path_batches",
    "atn_set_frames_in_flight", "atn_bank_streams", "atn_side_stream", "atn_mgpu_set_frames_in_flight", "atn_set_sampling_options", "atn_sample_texture",
    "atn_svgf_render", "atn_svgf_set_motion_depth", "atn_svgf_reset", "atn_svgf_set_atrous_iterations",
    "atn_svgf_download", "atn_svgf_output_device", "atn_svgf_set_dilate_temporal_weight", "atn_svgf_denoise", "atn_svgf_upload",
    "atn_film_device", "atn_tile_device", "atn_tile_slots", "atn_anyhit_twins", "atn_planar_area_lights", "atn_stream", "atn_synchronize",
    "atn_assemble_tiles", "atn_assemble_tiles_on", "atn_download_film", "atn_upload_film", "atn_get_stats", "atn_get_kernel_times",
    "atn_reset_kernel_times", "atn_generate_paths", "atn_trace_closest", "atn_cmj_samples", "atn_cmj_batch", "atn_ray_offset", "atn_get_random", "atn_random_count",
    "atn_material_table", "atn_material_eval", "atn_compact", "atn_compact2", "atn_sizeof_scene_desc", "atn_sizeof_destination",
    "atn_abi_version", "atn_build_id",
    "atn_download_path_cost", "atn_update_geometry", "atn_scene_device_arrays", "atn_lbvh_rebuild_list", "atn_lbvh_build",
    "atn_mgpu_update_geometry", "atn_mgpu_lbvh_rebuild_list",
    "atn_mgpu_create", "atn_mgpu_destroy", "atn_mgpu_last_error", "atn_mgpu_shard_count", "atn_mgpu_shard_device",
    "atn_mgpu_upload_scene", "atn_mgpu_update_tlas", "atn_mgpu_update_camera", "atn_mgpu_init_sampler",
    "atn_mgpu_set_random", "atn_mgpu_render", "atn_mgpu_reset", "atn_mgpu_synchronize", "atn_mgpu_film_device",
    "atn_mgpu_download_film",
    "atn_set_regeneration", "atn_get_regeneration", "atn_render_burst", "atn_regen_stage_counts",
    "atn_mgpu_set_regeneration", "atn_mgpu_render_burst", "atn_set_upload_options", "atn_libm_probe", "atn_set_shade_math",
]


class Destination(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("maxDepth", C.c_int32),
                ("russianRouletteDepth", C.c_int32), ("sample", C.c_int32), ("frame", C.c_uint32),
                ("progressive", C.c_int32), ("break_on_terminate", C.c_int32),
                ("count_stats", C.c_int32), ("profile", C.c_int32)]


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("ATEN_AMD_LIB", build.HIP_LIB)    # override: kernel-variant experiments (tools/)
        if not os.path.exists(path):
            raise RuntimeError(
                "aten_amd: %s is missing. Build it with `python -m aten_amd.build hip` "
                "(or __graft_entry__.build()); there is no CPU fallback." % path)
        l = C.CDLL(path)
        vp = C.c_void_p
        l.atn_create.argtypes = [C.POINTER(vp), C.c_int]
        l.atn_destroy.argtypes = [vp]; l.atn_destroy.restype = None
        l.atn_last_error.argtypes = [vp]; l.atn_last_error.restype = C.c_char_p
        l.atn_upload_scene.argtypes = [vp, vp]
        l.atn_update_camera.argtypes = [vp, vp]
        l.atn_update_tlas.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32]
        l.atn_download_path_cost.argtypes = [vp, vp]
        l.atn_update_geometry.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, vp, C.c_uint32, C.c_uint32]
        l.atn_scene_device_arrays.argtypes = [vp, vp, vp, vp]
        l.atn_lbvh_rebuild_list.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]
        l.atn_lbvh_build.argtypes = [vp, vp, C.c_uint32, C.c_int32, vp, vp, vp, C.c_uint32, C.c_int32, vp, vp, vp]
        l.atn_mgpu_update_geometry.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, vp, C.c_uint32, C.c_uint32]
        l.atn_mgpu_lbvh_rebuild_list.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]
        l.atn_init_sampler.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32]
        l.atn_set_random.argtypes = [vp, vp, C.c_uint32]
        l.atn_get_random.argtypes = [vp, vp, C.c_uint32]
        l.atn_random_count.argtypes = [vp]; l.atn_random_count.restype = C.c_uint32
        l.atn_set_screen_shard.argtypes = [vp, C.c_int32, C.c_int32]
        l.atn_render.argtypes = [vp, C.POINTER(Destination), vp]
        l.atn_reset.argtypes = [vp]
        l.atn_set_upload_options.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        l.atn_set_regeneration.argtypes = [vp, C.c_int32]
        l.atn_set_shade_math.argtypes = [vp, C.c_int32]
        l.atn_get_regeneration.argtypes = [vp]; l.atn_get_regeneration.restype = C.c_int32
        l.atn_render_burst.argtypes = [vp, C.POINTER(Destination), C.c_int32, vp]
        l.atn_regen_stage_counts.argtypes = [vp, vp, vp, C.c_uint32, C.POINTER(C.c_uint32)]
        l.atn_set_path_batches.argtypes = [vp, C.c_int32]
        l.atn_set_frames_in_flight.argtypes = [vp, C.c_int32]
        l.atn_bank_streams.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        l.atn_set_sampling_options.argtypes = [vp, C.c_int32, C.c_int32]
        l.atn_sample_texture.argtypes = [vp, C.c_int32, C.c_uint32, vp, vp]
        l.atn_mgpu_set_frames_in_flight.argtypes = [vp, C.c_int32]
        l.atn_svgf_render.argtypes = [vp, vp, C.c_int32, vp, vp]
        l.atn_svgf_set_motion_depth.argtypes = [vp, vp, C.c_uint32]
        l.atn_svgf_reset.argtypes = [vp]
        l.atn_svgf_set_atrous_iterations.argtypes = [vp, C.c_int32]
        l.atn_svgf_download.argtypes = [vp, C.c_int32, vp]
        l.atn_svgf_set_dilate_temporal_weight.argtypes = [vp, C.c_int32]
        l.atn_svgf_denoise.argtypes = [vp, vp, C.c_int32, vp, vp]
        l.atn_svgf_upload.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, vp]
        l.atn_svgf_output_device.argtypes = [vp]
        l.atn_svgf_output_device.restype = vp
        l.atn_film_device.argtypes = [vp]; l.atn_film_device.restype = vp
        l.atn_tile_device.argtypes = [vp]; l.atn_tile_device.restype = vp
        l.atn_tile_slots.argtypes = [vp]; l.atn_tile_slots.restype = C.c_uint32
        l.atn_anyhit_twins.argtypes = [vp]; l.atn_anyhit_twins.restype = C.c_uint32
        l.atn_planar_area_lights.argtypes = [vp]; l.atn_planar_area_lights.restype = C.c_uint32
        l.atn_stream.argtypes = [vp]; l.atn_stream.restype = vp
        l.atn_side_stream.argtypes = [vp]; l.atn_side_stream.restype = vp
        l.atn_synchronize.argtypes = [vp]
        l.atn_assemble_tiles.argtypes = [vp, vp, C.c_int32, vp]
        l.atn_assemble_tiles_on.argtypes = [vp, vp, C.c_int32, vp, vp]
        l.atn_download_film.argtypes = [vp, vp]
        l.atn_upload_film.argtypes = [vp, C.c_int32, C.c_int32, vp]
        l.atn_get_stats.argtypes = [vp, vp]
        l.atn_get_kernel_times.argtypes = [vp, vp, vp]
        l.atn_reset_kernel_times.argtypes = [vp]
        l.atn_generate_paths.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, vp]
        l.atn_trace_closest.argtypes = [vp, vp, C.c_uint32, C.c_float, C.c_float, vp, vp]
        l.atn_cmj_samples.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, vp]
        l.atn_cmj_batch.argtypes = [vp, C.c_uint32, vp, vp, vp, C.c_int32, vp]
        l.atn_ray_offset.argtypes = [vp, C.c_uint32, vp, vp, vp]
        l.atn_libm_probe.argtypes = [vp, C.c_int32, C.c_uint32, vp, vp, vp]
        l.atn_material_table.argtypes = [vp, C.c_int32, C.c_uint32, vp, vp, vp, vp, vp, vp, vp, vp]
        l.atn_material_eval.argtypes = [vp, C.c_int32, C.c_uint32, vp, vp, vp, vp, vp]
        l.atn_compact.argtypes = [vp, vp, C.c_uint32, vp, C.POINTER(C.c_uint32)]
        l.atn_compact2.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, vp, C.POINTER(C.c_uint32), vp, C.POINTER(C.c_uint32)]
        l.atn_mgpu_create.argtypes = [C.POINTER(vp), vp, C.c_int32]
        l.atn_mgpu_destroy.argtypes = [vp]; l.atn_mgpu_destroy.restype = None
        l.atn_mgpu_last_error.argtypes = [vp]; l.atn_mgpu_last_error.restype = C.c_char_p
        l.atn_mgpu_shard_count.argtypes = [vp]; l.atn_mgpu_shard_count.restype = C.c_int32
        l.atn_mgpu_shard_device.argtypes = [vp, C.c_int32]; l.atn_mgpu_shard_device.restype = C.c_int32
        l.atn_mgpu_upload_scene.argtypes = [vp, vp]
        l.atn_mgpu_update_tlas.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32]
        l.atn_mgpu_update_camera.argtypes = [vp, vp]
        l.atn_mgpu_init_sampler.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32]
        l.atn_mgpu_set_random.argtypes = [vp, vp, C.c_uint32]
        l.atn_mgpu_render.argtypes = [vp, C.POINTER(Destination), vp]
        l.atn_mgpu_reset.argtypes = [vp]
        l.atn_mgpu_set_regeneration.argtypes = [vp, C.c_int32]
        l.atn_mgpu_render_burst.argtypes = [vp, C.POINTER(Destination), C.c_int32, vp]
        l.atn_mgpu_synchronize.argtypes = [vp]
        l.atn_mgpu_film_device.argtypes = [vp]; l.atn_mgpu_film_device.restype = vp
        l.atn_mgpu_download_film.argtypes = [vp, vp]
        l.atn_build_id.argtypes = []; l.atn_build_id.restype = C.c_char_p
        for n in ("atn_sizeof_scene_desc", "atn_sizeof_destination", "atn_abi_version"):
            getattr(l, n).restype = C.c_uint32
        _lib = l
    return _lib
