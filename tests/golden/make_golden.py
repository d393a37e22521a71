"""Mints the golden vectors under tests/golden/oracle_golden.npz.

`seeds_*` and `cmj_*` (the integer rows) come from the REFERENCE itself: oracle/_ref/libatenref.so =
the untouched sampler/cmj.h + sampler/sampler.cpp compiled here (`make -C oracle _ref`).  The float
path of the reference cannot be built in this image (empty glm / tinyobjloader / stb / nanovdb
submodules), so everything else is an ORACLE output, not a reference output: it pins the oracle
against regressions and gives the GPU tests fixed expected values.  Scene inputs: assets/cornellbox/orig.obj
(fan triangulation) and assets/sponza/sponza_lod.{obj,sbvh}; our BVH builder for Cornell.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aten_amd.scene import scenedefs  # noqa: E402
from oracle import orc  # noqa: E402
from oracle import ref  # noqa: E402  (the reference's own sampler sources; build container only)

OUT = os.path.dirname(os.path.abspath(__file__))

CMJ_CASES = [(0, 0, 0x12345678), (17, 4, 0x9e3779b9), (255, 0, 1), (100, 7, 0xdeadbeef)]


def main():
    g = {}
    g["seeds_512"] = ref.init_sampler(512, 512, 0)[:64]
    for i, (idx, dim, scr) in enumerate(CMJ_CASES):
        g["cmj_%d" % i] = ref.cmj_samples(idx, dim, scr, 1024)

    fs, cam = scenedefs.cornell_box()
    W = H = 64
    c = orc.create_camera(cam["pos"], cam["at"], cam["vfov"], W, H)
    seeds = orc.init_sampler(W, H, 0)
    for frame in (0, 1, 7):
        g["rays_cornell64_f%d" % frame] = orc.generate_paths(c, seeds, W, H, 0, frame)
    rays = orc.generate_paths(c, seeds, W, H, 0, 0)
    isect, stats = orc.trace_closest(fs, rays)
    g["isect_cornell64"] = isect
    g["isect_cornell64_stats"] = stats
    for depth in (3, 5):
        film = np.zeros((H, W, 4), np.float32)
        for frame in range(4):
            orc.render(fs, c, seeds, W, H, depth, 3, frame=frame, film=film)
        g["film_cornell64_d%d_f0to3" % depth] = film

    rng = np.random.default_rng(1234)
    o = rng.uniform(-1.5, 1.5, (256, 3)).astype(np.float32)
    o[::7] *= 1e-3
    n = rng.normal(size=(256, 3)).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    g["offset_o"], g["offset_n"] = o, n
    g["offset_out"] = orc.ray_offset(o, n)

    fs2, cam2 = scenedefs.sponza_lod()
    W2, H2 = 128, 72
    c2 = orc.create_camera(cam2["pos"], cam2["at"], cam2["vfov"], W2, H2)
    seeds2 = orc.init_sampler(W2, H2, 0)
    rays2 = orc.generate_paths(c2, seeds2, W2, H2, 0, 0)
    isect2, stats2 = orc.trace_closest(fs2, rays2)
    g["isect_sponza128x72"] = isect2
    g["isect_sponza128x72_stats"] = stats2
    g["film_sponza128x72_d5_f0"] = orc.render(fs2, c2, seeds2, W2, H2, 5, 3, frame=0)

    np.savez_compressed(os.path.join(OUT, "oracle_golden.npz"), **g)
    print("wrote", os.path.join(OUT, "oracle_golden.npz"), {k: getattr(v, "shape", None) for k, v in g.items()})


if __name__ == "__main__":
    main()
