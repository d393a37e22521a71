#!/usr/bin/env python
"""Mints tests/golden/svgf_golden.npz from the CPU oracle (oracle/orc_svgf.h): the reference holds no fixture for
its SVGF path, so these vectors only pin the oracle against silent change (parity itself is "unpinned").
    python tests/golden/make_golden_svgf.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aten_amd.scene import scenedefs  # noqa: E402
from oracle import orc  # noqa: E402

W, H = 48, 32


def run():
    fs, cam = scenedefs.cornell_box()
    c = orc.create_camera(cam["pos"], cam["at"], cam["vfov"], W, H)
    seeds = orc.init_sampler(W, H, 0)
    sv = orc.Svgf()
    out = {}
    for frame in range(3):
        film, st = sv.render(fs, c, seeds, W, H, 3, 3, frame=frame, compute_motion=True, stages=True)
        out["film%d" % frame] = film
        out["stages%d" % frame] = st
    for name in ("prev_normal_depth", "prev_albedo_meshid", "prev_color_variance", "prev_moment_temporalweight", "motion_depth"):
        out[name] = sv.buffer(name)
    sv.close()
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "svgf_golden.npz"), **run())
    print("written")
