"""ctypes binding of libaten_amd_scene.so (include/aten_amd_scene.h): host-only BVH builder."""
import ctypes as C
import os

from . import build

_lib = None


def hostlib():
    global _lib
    if _lib is None:
        path = build.HOST_LIB
        if not os.path.exists(path):
            build.build_host()
        lib = C.CDLL(path)
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
