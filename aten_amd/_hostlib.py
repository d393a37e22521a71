"""ctypes binding of libaten_amd_scene.so (include/aten_amd_scene.h): host-only BVH builder."""
import ctypes as C
import os

from . import build

_lib = None

ATNS_ABI_VERSION = 2        # include/aten_amd_scene.h

ORDER_AS_SPLIT, ORDER_AREA, ORDER_AREA_SMALL, ORDER_COUNT, ORDER_COUNT_SMALL, ORDER_NEAR_POINT = range(6)


class BvhOptions(C.Structure):          # atns_bvh_options
    _fields_ = [("spatial_splits", C.c_int32), ("spatial_alpha", C.c_float), ("object_bins", C.c_int32),
                ("spatial_bins", C.c_int32), ("sweep_below", C.c_int32), ("child_order", C.c_int32),
                ("max_refs_factor", C.c_float), ("order_point", C.c_float * 3),
                ("order_point_given", C.c_int32), ("reinsert_iterations", C.c_int32), ("reinsert_batch", C.c_float)]


class BvhStats(C.Structure):            # atns_bvh_stats
    _fields_ = [("n_nodes", C.c_uint32), ("n_leaves", C.c_uint32), ("n_spatial_splits", C.c_uint32), ("n_reinsertions", C.c_uint32),
                ("sah_cost", C.c_float)]


def default_bvh_options(**kw):
    o = BvhOptions()
    hostlib().atns_bvh_default_options(C.byref(o))
    for k, v in kw.items():
        if k == "order_point":
            o.order_point[:] = [float(x) for x in v]
            o.order_point_given = 1
        else:
            setattr(o, k, v)
    return o


def hostlib():
    global _lib
    if _lib is None:
        path = build.HOST_LIB
        if not os.path.exists(path):
            build.build_host()
        lib = C.CDLL(path)
        try:
            lib.atns_abi_version.restype = C.c_uint32
            ver = lib.atns_abi_version()
        except AttributeError:
            ver = 1
        if ver != ATNS_ABI_VERSION:
            raise RuntimeError("libaten_amd_scene.so speaks ABI %d, this binding %d: rebuild (python -m aten_amd.build host)" % (ver, ATNS_ABI_VERSION))
        lib.atns_bvh_default_options.argtypes = [C.POINTER(BvhOptions)]
        lib.atns_bvh_default_options.restype = None
        lib.atns_build_blas_opt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(BvhOptions),
                                            C.POINTER(C.c_void_p), C.POINTER(C.c_uint32),
                                            C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(BvhStats)]
        lib.atns_build_blas_opt.restype = C.c_int
        lib.atns_optimize_nodes.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(BvhOptions), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32),
                                            C.POINTER(BvhStats)]
        lib.atns_optimize_nodes.restype = C.c_int
        lib.atns_anyhit_twin.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        lib.atns_anyhit_twin.restype = C.c_int
        lib.atns_build_blas.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_uint32),
                                        C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.atns_build_blas.restype = C.c_int
        lib.atns_build_tlas.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
        lib.atns_build_tlas.restype = C.c_int
        lib.atns_free.argtypes = [C.c_void_p]
        lib.atns_free.restype = None
        lib.atns_validate_nodes.argtypes = [C.c_void_p, C.c_uint32]
        lib.atns_validate_nodes.restype = C.c_int64
        lib.atns_create_camera.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                           C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32]
        lib.atns_create_camera.restype = C.c_int
        _lib = lib
    return _lib
