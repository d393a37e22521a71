"""Tree quality of the host BVH builder, measured with the CPU oracle's visit counters (no GPU).

One 5-bounce frame of a scene is rendered by the oracle (test infrastructure) twice -- once per tree --
and the node visits / triangle tests of the SAME rays are compared (rays, hits and shadow rays must be equal:
the tree is an input, closest hits do not depend on it).

  python tools/tree_quality.py [sponza|atrium|cornell] [W H]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from aten_amd.scene import scenedefs                     # noqa: E402
from oracle import orc                                   # noqa: E402


def frame_counters(scene, cam, w, h, depth=5, primary_only=False):
    c = orc.create_camera(cam["pos"], cam["at"], cam["vfov"], w, h)
    seeds = orc.init_sampler(w, h, 0)
    t = time.time()
    film, cnt = orc.render(scene, c, seeds, w, h, max_depth=1 if primary_only else depth, rr_depth=3, counters=True)
    return dict(closest=int(cnt[0]), shadow=int(cnt[1]), hits=int(cnt[2]), nodes=int(cnt[3]), tris=int(cnt[4]),
                sec=time.time() - t)


def report(name, scene, cam, w, h):
    n = sum(len(x) for x in scene.arrays["bvh_lists"][1:])
    full = frame_counters(scene, cam, w, h)
    prim = frame_counters(scene, cam, w, h, primary_only=True)
    print("%-28s nodes %7d  visits %8.2f M  tri tests %6.2f M  primary visits %6.2f M   (rays %d hits %d shadow %d; %.1f s)" % (
        name, n, full["nodes"] / 1e6, full["tris"] / 1e6, prim["nodes"] / 1e6, full["closest"], full["hits"], full["shadow"],
        full["sec"]), flush=True)
    return full, prim


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "sponza"
    w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (480, 270)
    if which == "sponza":
        s_ref, cam = scenedefs.sponza_lod(use_sbvh=True)
        s_own, _ = scenedefs.sponza_lod(use_sbvh=False)
        a, ap = report("reference sponza_lod.sbvh", s_ref, cam, w, h)
        b, bp = report("own builder", s_own, cam, w, h)
        print("ratio own/ref: visits %.3f  primary %.3f  tri tests %.3f" % (b["nodes"] / a["nodes"], bp["nodes"] / ap["nodes"],
                                                                         b["tris"] / a["tris"]))
    elif which == "atrium":
        s, cam = scenedefs.atrium()
        report("atrium own builder", s, cam, w, h)
    else:
        s, cam = scenedefs.cornell_box()
        report("cornell own builder", s, cam, w, h)
