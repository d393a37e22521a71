"""Tree quality of the host BVH builder, measured with the CPU oracle's visit counters (no GPU).

One 5-bounce frame of a scene is rendered by the oracle (test infrastructure) twice -- once per tree --
and the node visits / triangle tests of the SAME rays are compared (rays, hits and shadow rays must be equal:
the tree is an input, closest hits do not depend on it).

  python tools/tree_quality.py [sponza|atrium|cornell|lbvh] [W H]
"""
import os
import sys
import time


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


def lbvh_vs_sah(name, scene, obj_id, cam, w, h):
    """The tree idaten::LBVHBuilder builds (Morton-order median splits; oracle/orc_lbvh.h restates it, the device builder
    atn_lbvh_build is byte-equal to that) against the host split-BVH tree of the same triangles."""
    k = scene.blas_index[obj_id]
    o = scene.arrays["objects"][obj_id]
    t0, n = int(o["triangle_id"]), int(o["triangle_num"])
    tris = scene.arrays["triangles"][t0:t0 + n]
    vp = scene.arrays["vtx_pos"]
    used = vp[tris["idx"].min():tris["idx"].max() + 1, :3]
    a = frame_counters(scene, cam, w, h)
    ap = frame_counters(scene, cam, w, h, primary_only=True)
    sah = scene.arrays["bvh_lists"][k]
    nodes = orc.lbvh_build(tris, used.min(0), used.max(0), vp, tri_id_offset=t0)
    scene.replace_bvh_list(k, nodes)
    b = frame_counters(scene, cam, w, h)
    bp = frame_counters(scene, cam, w, h, primary_only=True)
    scene.replace_bvh_list(k, sah)
    print("%-24s %7d triangles: host tree %7d nodes %8.2f M visits (primary %6.2f M, tri tests %5.2f M) | LBVH %7d nodes %8.2f M visits "
          "(primary %6.2f M, tri tests %5.2f M)  LBVH / host: visits %.3f primary %.3f tri tests %.3f" % (
              name, n, len(sah), a["nodes"] / 1e6, ap["nodes"] / 1e6, a["tris"] / 1e6, len(nodes), b["nodes"] / 1e6, bp["nodes"] / 1e6,
              b["tris"] / 1e6, b["nodes"] / a["nodes"], bp["nodes"] / ap["nodes"], b["tris"] / a["tris"]), flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "sponza"
    w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (480, 270)
    if which == "sponza":
        s_ref, cam = scenedefs.sponza_lod(use_sbvh=True)
        s_own, _ = scenedefs.sponza_lod(use_sbvh=False)
        a, ap = report("reference sponza_lod.sbvh", s_ref, cam, w, h)
        b, bp = report("own builder", s_own, cam, w, h)
        s_opt, _ = scenedefs.sponza_lod(use_sbvh=True, optimize_sbvh=True)
        c, cp = report("reference tree, post passes", s_opt, cam, w, h)
        print("ratio own/ref: visits %.3f  primary %.3f  tri tests %.3f" % (b["nodes"] / a["nodes"], bp["nodes"] / ap["nodes"],
                                                                         b["tris"] / a["tris"]))
        print("ratio (reference tree through atns_optimize_nodes)/ref: visits %.3f  primary %.3f  tri tests %.3f" % (
            c["nodes"] / a["nodes"], cp["nodes"] / ap["nodes"], c["tris"] / a["tris"]))
    elif which == "lbvh":
        b, oid, cam = scenedefs.deformable_room(0.7)
        lbvh_vs_sah("deformable room blob", b.build(), oid, cam, w, h)
        s_own, cam = scenedefs.sponza_lod(use_sbvh=False)
        poly = [i for i, o in enumerate(s_own.arrays["objects"]) if i in s_own.blas_index][0]
        lbvh_vs_sah("sponza_lod.obj", s_own, poly, cam, w, h)
    elif which == "atrium":
        s, cam = scenedefs.atrium()
        report("atrium own builder", s, cam, w, h)
    else:
        s, cam = scenedefs.cornell_box()
        report("cornell own builder", s, cam, w, h)
