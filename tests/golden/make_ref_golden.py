"""Mints tests/golden/ref_golden.npz from oracle/_ref/libatenref.so = the reference's own, untouched
sampler/cmj.h, sampler/sampler.cpp and math/math.h compiled in the build container
(`make -C oracle _ref`).  These ARE reference outputs (unlike oracle_golden.npz, which holds oracle
outputs for the float path): they pin the integer rows a1 / a2 of SURVEY 8(a) and the scalar helpers.

    python tests/golden/make_ref_golden.py        # needs /root/reference

Inputs are regenerated from seeds by `inputs()` below, so the fixture stores outputs (+ the small
special-value tables) only."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_golden.npz")

CMJ_CASES = [(0, 0, 0x12345678), (17, 4, 0x9e3779b9), (255, 0, 1), (100, 7, 0xdeadbeef)]
SEED_SIZES = [(512, 512, 0), (1920, 1080, 0), (100, 52, 7)]
N_WIDE = 1 << 17        # arbitrary 32-bit triples, 1 draw each
N_PATH = 1 << 14        # triples shaped like the integrator's (index < 256, dimension 0), 16 draws each


def inputs():
    rng = np.random.default_rng(20260929)
    wide = (rng.integers(0, 1 << 32, N_WIDE, dtype=np.uint64).astype(np.uint32),
            rng.integers(0, 1 << 32, N_WIDE, dtype=np.uint64).astype(np.uint32),
            rng.integers(0, 1 << 32, N_WIDE, dtype=np.uint64).astype(np.uint32))
    wide[1][::2] %= 64          # half of them with the small dimensions a path really reaches
    path = (rng.integers(0, 256, N_PATH, dtype=np.uint64).astype(np.uint32),
            np.zeros(N_PATH, np.uint32),
            rng.integers(0, 1 << 32, N_PATH, dtype=np.uint64).astype(np.uint32))
    special = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 1.5, -0.5, 1e-45, -1e-45, 1e-38, 3.4028235e38, -3.4028235e38,
                        np.inf, -np.inf, np.nan, 1e-9, 1.0000001, 0.99999994, 2.0, 180.0, 45.0, 60.0], np.float32)
    A, B = np.meshgrid(special, special, indexing="ij")
    fa = rng.standard_normal(4096).astype(np.float32) * np.float32(3.0)
    fb = rng.standard_normal(4096).astype(np.float32) * np.float32(3.0)
    ft = rng.uniform(-0.5, 1.5, 4096).astype(np.float32)
    # isClose in ULPs: pairs a few hundred to a few thousand ulps apart, around the 2500 threshold
    base = rng.uniform(0.01, 100.0, 4096).astype(np.float32)
    off = rng.integers(2300, 2700, 4096).astype(np.int32)
    near = (base.view(np.int32) + off).view(np.float32)
    return {"wide": wide, "path": path,
            "pair_a": np.concatenate([A.ravel(), fa]), "pair_b": np.concatenate([B.ravel(), fb]),
            "unary": np.concatenate([special, fa]), "t": ft, "fa": fa, "fb": fb,
            "close_a": np.concatenate([base, -base, A.ravel()]), "close_b": np.concatenate([near, -near, B.ravel()])}


def mint(src):
    """src: oracle.ref (the reference) -- or oracle.orc, which must give the same bytes."""
    inp = inputs()
    g = {}
    for (w, h, seed) in SEED_SIZES:
        s = src.init_sampler(w, h, seed)
        tag = "seeds_%dx%d_s%d" % (w, h, seed)
        g[tag + "_head"] = s[:64].copy()
        g[tag + "_stride"] = s[::max(1, len(s) // 4096)].copy()
        g[tag + "_sha256"] = np.frombuffer(hashlib.sha256(s.tobytes()).digest(), np.uint8).copy()
    for i, (idx, dim, scr) in enumerate(CMJ_CASES):
        g["cmj_%d" % i] = src.cmj_samples(idx, dim, scr, 1024)
        g["cmj2d_%d" % i] = src.cmj_samples2d(idx, dim, scr, 256)
    g["cmj_wide"] = src.cmj_batch(*inp["wide"], draws=1)
    g["cmj_path"] = src.cmj_batch(*inp["path"], draws=16)
    K = {"max": 0, "min": 1, "clamp": 2, "saturate": 3, "sign": 4, "mix": 5, "lerp": 6, "isclose_2500ulps": 7,
         "isinvalid": 8, "sqr": 9, "rsqrt": 10, "deg2rad": 11}
    g["math_max"] = src.math_kat(K["max"], inp["pair_a"], inp["pair_b"])
    g["math_min"] = src.math_kat(K["min"], inp["pair_a"], inp["pair_b"])
    lo = np.minimum(inp["fa"], inp["fb"]); hi = np.maximum(inp["fa"], inp["fb"])
    g["math_clamp"] = src.math_kat(K["clamp"], inp["t"] * 4 - 2, lo, hi)
    for name in ("saturate", "sign", "isinvalid", "sqr", "rsqrt", "deg2rad"):
        g["math_" + name] = src.math_kat(K[name], inp["unary"])
    g["math_mix"] = src.math_kat(K["mix"], inp["fa"], inp["fb"], inp["t"])
    g["math_lerp"] = src.math_kat(K["lerp"], inp["fa"], inp["fb"], inp["t"])
    g["math_isclose_2500ulps"] = src.math_kat(K["isclose_2500ulps"], inp["close_a"], inp["close_b"])
    return g


def main():
    from oracle import ref
    g = mint(ref)
    np.savez_compressed(OUT, **g)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(g), "arrays")


if __name__ == "__main__":
    main()
