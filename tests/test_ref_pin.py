"""Pins the INTEGER rows of the hot path (SURVEY 8(a) a1 seeds, a2 CMJ) and the scalar helpers of
math/math.h against the reference ITSELF: oracle/_ref/libatenref.so is the reference's untouched
sampler/cmj.h + sampler/sampler.cpp + math/math.h compiled in the build container
(`make -C oracle _ref`), tests/golden/ref_golden.npz its outputs (tests/golden/make_ref_golden.py).

  * everywhere (fixtures travel):        oracle == ref_golden.npz, bit for bit;
  * where /root/reference exists (here): oracle == _ref live, on fresh random inputs, and the
                                         committed fixture is what _ref produces today;
  * on the GPU (-m gpu):                 the HIP CMJ and the product's seeds == ref_golden.npz.
"""
import hashlib
import importlib.util
import os

import numpy as np
import pytest

from conftest import GOLDEN

_spec = importlib.util.spec_from_file_location("make_ref_golden", os.path.join(GOLDEN, "make_ref_golden.py"))
mrg = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(mrg)


@pytest.fixture(scope="module")
def ref_golden():
    return dict(np.load(os.path.join(GOLDEN, "ref_golden.npz")))


@pytest.fixture(scope="module")
def ref():
    from oracle import ref as r
    if not r.available():
        pytest.skip("the reference sources are not on this machine (GPU box): the fixtures stand in")
    r.lib()
    return r


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_oracle_equals_reference_fixture(orc, ref_golden):
    got = mrg.mint(orc)
    assert sorted(got) == sorted(ref_golden)
    for k in sorted(got):
        assert got[k].dtype == ref_golden[k].dtype and got[k].shape == ref_golden[k].shape, k
        assert got[k].tobytes() == ref_golden[k].tobytes(), "oracle differs from the reference's own output: %s" % k


def test_fixture_is_what_the_reference_produces(ref, ref_golden):
    got = mrg.mint(ref)
    for k in sorted(ref_golden):
        assert got[k].tobytes() == ref_golden[k].tobytes(), "stale fixture: %s (re-run make_ref_golden.py)" % k


def test_cmj_oracle_equals_reference_live(orc, ref):
    rng = np.random.default_rng()          # fresh inputs every run: the claim is for ALL inputs
    seed = int(rng.integers(0, 1 << 31))
    rng = np.random.default_rng(seed)
    n = 200_000
    idx = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    dim = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    dim[::2] %= 48
    scr = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    a, b = orc.cmj_batch(idx, dim, scr, 2), ref.cmj_batch(idx, dim, scr, 2)
    assert np.array_equal(_bits(a), _bits(b)), "seed %d" % seed
    assert np.all((b >= 0) & (b < 1))
    # the integrator's shape: index = (frame + sample) % 256, dimension 0, a path's worth of draws
    idx = rng.integers(0, 256, 20_000, dtype=np.uint64).astype(np.uint32)
    scr = rng.integers(0, 1 << 32, 20_000, dtype=np.uint64).astype(np.uint32)
    z = np.zeros(20_000, np.uint32)
    assert np.array_equal(_bits(orc.cmj_batch(idx, z, scr, 40)), _bits(ref.cmj_batch(idx, z, scr, 40))), "seed %d" % seed
    for (i, d, s) in [(0, 0, 0), (255, 0, 0xffffffff), (3, 0xffffffff, 0xffffffff), (256, 1, 1)]:
        assert np.array_equal(_bits(orc.cmj_samples2d(i, d, s, 300)), _bits(ref.cmj_samples2d(i, d, s, 300)))


@pytest.mark.parametrize("w,h,seed", [(512, 512, 0), (1920, 1080, 0), (1920, 1080, 12345), (3840, 2160, 0), (1, 1, 0), (7, 3, 2)])
def test_init_sampler_oracle_equals_reference_live(orc, ref, w, h, seed):
    assert orc.init_sampler(w, h, seed).tobytes() == ref.init_sampler(w, h, seed).tobytes()


def test_math_helpers_oracle_equals_reference_live(orc, ref):
    rng = np.random.default_rng(7)
    a = rng.standard_normal(50_000).astype(np.float32) * np.float32(10)
    b = rng.standard_normal(50_000).astype(np.float32) * np.float32(10)
    t = rng.uniform(-1, 2, 50_000).astype(np.float32)
    a[::97] = np.nan; b[::89] = np.nan; a[::101] = np.inf; b[::103] = -np.inf; a[::107] = 0.0; b[::109] = -0.0
    K = ref.MATH_KINDS
    for name in ("max", "min"):
        assert np.array_equal(_bits(orc.math_kat(K[name], a, b)), _bits(ref.math_kat(K[name], a, b))), name
    for name in ("saturate", "sign", "isinvalid", "sqr", "rsqrt", "deg2rad"):
        assert np.array_equal(_bits(orc.math_kat(K[name], a)), _bits(ref.math_kat(K[name], a))), name
    fa, fb = np.nan_to_num(a, nan=1.0, posinf=2.0, neginf=-2.0), np.nan_to_num(b, nan=1.0, posinf=2.0, neginf=-2.0)
    for name in ("mix", "lerp"):
        assert np.array_equal(_bits(orc.math_kat(K[name], fa, fb, t)), _bits(ref.math_kat(K[name], fa, fb, t))), name
    lo, hi = np.minimum(fa, fb), np.maximum(fa, fb)
    assert np.array_equal(_bits(orc.math_kat(K["clamp"], a, lo, hi)), _bits(ref.math_kat(K["clamp"], a, lo, hi)))
    base = rng.uniform(1e-3, 1e3, 50_000).astype(np.float32)
    near = (base.view(np.int32) + rng.integers(-4000, 4000, 50_000).astype(np.int32)).view(np.float32)
    for (x, y) in [(base, near), (-base, -near), (base, -near), (a, b)]:
        assert np.array_equal(orc.math_kat(K["isclose_2500ulps"], x, y), ref.math_kat(K["isclose_2500ulps"], x, y))


def test_mt19937_known_answers(ref_golden):
    # std::mt19937(0)'s published first outputs: what sampler.cpp:14-17 fills the table with
    assert list(ref_golden["seeds_512x512_s0_head"][:3]) == [2357136044, 2546248239, 3071714933]
    assert np.allclose(ref_golden["cmj_1"][:4], [0.0802232176, 0.578303516, 0.591444075, 0.456700593], rtol=0, atol=1e-9)


# ---- the product against the reference's outputs (GPU box: fixtures only) -------------------

@pytest.mark.gpu
def test_gpu_cmj_equals_reference_fixture(gpu, ref_golden):
    inp = mrg.inputs()
    assert np.array_equal(_bits(gpu.cmj_batch(*inp["wide"], draws=1)), _bits(ref_golden["cmj_wide"]))
    assert np.array_equal(_bits(gpu.cmj_batch(*inp["path"], draws=16)), _bits(ref_golden["cmj_path"]))
    for i, (idx, dim, scr) in enumerate(mrg.CMJ_CASES):
        assert np.array_equal(_bits(gpu.cmj_samples(idx, dim, scr, 1024)), _bits(ref_golden["cmj_%d" % i]))


@pytest.mark.gpu
def test_gpu_seeds_equal_reference_fixture(gpu, ref_golden):
    for (w, h, seed) in mrg.SEED_SIZES:
        gpu.initSampler(w, h, seed)
        s = gpu.getRandom()
        tag = "seeds_%dx%d_s%d" % (w, h, seed)
        assert s.dtype == np.uint32 and len(s) == w * h
        assert np.array_equal(s[:64], ref_golden[tag + "_head"])
        assert np.array_equal(s[::max(1, len(s) // 4096)], ref_golden[tag + "_stride"])
        assert hashlib.sha256(s.tobytes()).digest() == ref_golden[tag + "_sha256"].tobytes()
