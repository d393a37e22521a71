"""ctypes binding of oracle/_ref/libatenref.so: the reference's OWN sampler / math sources
(sampler/cmj.h, sampler/sampler.cpp, math/math.h) compiled where they lie under /root/reference by
`make -C oracle _ref`.  TEST INFRASTRUCTURE ONLY, and only in the build container: /root/reference
does not exist on the GPU box, where tests use the fixtures minted from this library
(tests/golden/ref_golden.npz, script tests/golden/make_ref_golden.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_ref", "libatenref.so")
REFERENCE = os.environ.get("ATEN_REFERENCE", "/root/reference")
_lib = None

MATH_KINDS = {"max": 0, "min": 1, "clamp": 2, "saturate": 3, "sign": 4, "mix": 5, "lerp": 6, "isclose_2500ulps": 7,
              "isinvalid": 8, "sqr": 9, "rsqrt": 10, "deg2rad": 11}


def available():
    """True when the library exists or can be built (the reference sources are present)."""
    return os.path.exists(_LIB) or os.path.isdir(os.path.join(REFERENCE, "src", "libaten", "sampler"))


def lib():
    global _lib
    if _lib is None:
        if os.path.isdir(os.path.join(REFERENCE, "src", "libaten", "sampler")):
            subprocess.check_call(["make", "-s", "-C", _HERE, "_ref", "REF=" + REFERENCE])
        l = C.CDLL(_LIB)
        l.ref_get_random.restype = C.c_uint32
        _lib = l
    return _lib


def init_sampler(w, h, seed=0):
    out = np.zeros(w * h, np.uint32)
    lib().ref_init_sampler(C.c_void_p(out.ctypes.data), w, h, seed)
    return out


def cmj_samples(index, dimension, scramble, n):
    out = np.zeros(n, np.float32)
    lib().ref_cmj_samples(C.c_uint32(index), C.c_uint32(dimension), C.c_uint32(scramble), n, C.c_void_p(out.ctypes.data))
    return out


def cmj_samples2d(index, dimension, scramble, n):
    out = np.zeros((n, 2), np.float32)
    lib().ref_cmj_samples2d(C.c_uint32(index), C.c_uint32(dimension), C.c_uint32(scramble), n, C.c_void_p(out.ctypes.data))
    return out


def cmj_batch(index, dimension, scramble, draws=1):
    index = np.ascontiguousarray(index, np.uint32); dimension = np.ascontiguousarray(dimension, np.uint32)
    scramble = np.ascontiguousarray(scramble, np.uint32)
    out = np.zeros((len(index), draws), np.float32)
    lib().ref_cmj_batch(len(index), C.c_void_p(index.ctypes.data), C.c_void_p(dimension.ctypes.data),
                        C.c_void_p(scramble.ctypes.data), draws, C.c_void_p(out.ctypes.data))
    return out


def math_kat(kind, a, b=None, c=None):
    a = np.ascontiguousarray(a, np.float32)
    b = np.zeros_like(a) if b is None else np.ascontiguousarray(b, np.float32)
    c = np.zeros_like(a) if c is None else np.ascontiguousarray(c, np.float32)
    out = np.zeros_like(a)
    lib().ref_math_kat(kind, len(a), C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), C.c_void_p(c.ctypes.data),
                       C.c_void_p(out.ctypes.data))
    return out
