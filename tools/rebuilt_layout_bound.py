"""What the top-levels-first node layout would buy for a list the device LBVH rebuilds every tick.

A rebuilt list stays in walk order (the level-by-level order needs a sort of the 2n - 1 nodes per tick).  This
measures the other side of that trade on the GPU: the deforming-mesh room rendered with the blob's list holding the SAME LBVH tree
(built on the CPU by the oracle, which the device builder equals node for node) uploaded
  A  with the layout        (atn_set_upload_options node_layout = 1)
  B  in walk order          (node_layout = 0)
  C  rebuilt on the device  (atn_lbvh_rebuild_list: walk order by construction)
and the time of a rebuild; with --sky (an environment light: shadow rays that may walk any-hit twins) C also with the twins the
rebuild re-threads (k_lbvh_twin_*), with what they cost the rebuild.  A - B is the most a device-side layout pass could recover per
frame; it would pay rebuild-sized cost per tick.

Usage (GPU box): python tools/rebuilt_layout_bound.py [--sky] [--out FILE] [nu,nv ...]       one JSON line per mesh size."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                   # noqa: E402,F401  (the HIP runtime torch ships comes first)
from aten_amd.renderer import PathTracing                      # noqa: E402
from aten_amd.scene import scenedefs                           # noqa: E402
from aten_amd.scene.camera import create_camera                # noqa: E402
from oracle import orc                                         # noqa: E402

W, H, FRAMES = 1920, 1080, 48


def tick(b, oid, t, nu, nv):
    pos, nml, idx = scenedefs.blob_mesh(t, nu=nu, nv=nv)
    b.set_mesh_vertices(oid, pos, idx, nml)
    fs = b.build()
    o = fs.arrays["objects"][oid]
    t0, n = int(o["triangle_id"]), int(o["triangle_num"])
    tris = fs.arrays["triangles"][t0:t0 + n]
    v0, v1 = int(tris["idx"].min()), int(tris["idx"].max()) + 1
    used = fs.arrays["vtx_pos"][v0:v1, :3]
    return fs, dict(list=fs.blas_index[oid], t0=t0, n=n, v0=v0, v1=v1, bmin=used.min(0), bmax=used.max(0))


def timed(r):
    r.set_frames_in_flight(4)
    for f in range(8):
        r.render(W, H, frame=f, download=False)
    r.synchronize()
    best = None
    for _ in range(3):
        r.reset()
        t0 = time.perf_counter()
        for f in range(FRAMES):
            r.render(W, H, frame=f, download=False)
        r.synchronize()
        dt = (time.perf_counter() - t0) / FRAMES * 1e3
        best = dt if best is None else min(best, dt)
    r.reset()
    r.set_frames_in_flight(1)
    film = r.render(W, H, frame=0, count_stats=True).copy()
    return best, film, r.stats()


def main():
    args = sys.argv[1:]
    out = None
    sky = "--sky" in args                                  # an environment light on top: shadow rays that may use any-hit twins
    if sky:
        args.remove("--sky")
    if "--out" in args:
        i = args.index("--out")
        out = args[i + 1]
        del args[i:i + 2]
    sizes = [tuple(int(x) for x in a.split(",")) for a in args] or [(48, 24), (160, 80), (448, 224)]
    lines = []
    for nu, nv in sizes:
        b, oid, cam = scenedefs.deformable_room(0.0, nu=nu, nv=nv)
        if sky:
            env = scenedefs.synthetic_envmap(256, 128)
            b.add_ibl(b.add_texture("sky", env), avg_illum=scenedefs.envmap_avg_illum(env))
        c = create_camera(cam["pos"], cam["at"], cam["vfov"], W, H)
        fs, d = tick(b, oid, 1.3, nu, nv)
        tris = fs.arrays["triangles"][d["t0"]:d["t0"] + d["n"]]
        nodes = orc.lbvh_build(tris, d["bmin"], d["bmax"], fs.arrays["vtx_pos"], tri_id_offset=d["t0"])
        fs.replace_bvh_list(d["list"], nodes)
        rec = dict(scene="deformable_room%s, blob %d x %d" % (" + sky" if sky else "", nu, nv), blob_triangles=d["n"], blob_nodes=len(nodes), width=W, height=H,
                   frames_in_flight=4)
        films = {}
        # (B_no_planar_rule: every atn_update_tlas drops the planar-light flags, DESIGN.md section 5, so C runs without that rule)
        for name, layout, planar in (("A_layout", 1, 1), ("B_walk_order", 0, 1), ("B_no_planar_rule", 0, 0)):
            r = PathTracing(0)
            try:
                r.set_upload_options(anyhit_twin=0, node_layout=layout, planar_lights=planar)
                r.UpdateSceneData(fs); r.updateCamera(c); r.initSampler(W, H, 0)
                ms, films[name], st = timed(r)
                rec[name + "_ms_per_frame"] = round(ms, 4)
                rec[name + "_closest_nodes"] = int(st["closest_nodes"])
                rec[name + "_shadow_nodes"] = int(st["shadow_nodes"])
            finally:
                r.close()
        for twin in ((0, 2) if sky else (0,)):
            tag = "C_device_rebuilt" + ("_twins" if twin else "")
            r = PathTracing(0)
            try:
                r.set_upload_options(anyhit_twin=twin, node_layout=1)
                fs0, d0 = tick(b, oid, 0.0, nu, nv)
                r.UpdateSceneData(fs0); r.updateCamera(c); r.initSampler(W, H, 0)
                fs1, d1 = tick(b, oid, 1.3, nu, nv)
                a = fs1.arrays
                r.updateGeometry(vtx_pos=a["vtx_pos"][d1["v0"]:d1["v1"]], vtx_nml=a["vtx_nml"][d1["v0"]:d1["v1"]], vtx_offset=d1["v0"],
                                 triangles=a["triangles"][d1["t0"]:d1["t0"] + d1["n"]], tri_offset=d1["t0"])
                r.lbvh_rebuild_list(d1["list"], d1["t0"], d1["n"], d1["bmin"], d1["bmax"])
                r.updateBVH(fs1)
                rec[tag + "_anyhit_twins"] = int(r.anyhit_twins())
                ms, films[tag], st = timed(r)
                rec[tag + "_ms_per_frame"] = round(ms, 4)
                rec[tag + "_closest_nodes"] = int(st["closest_nodes"])
                rec[tag + "_shadow_nodes"] = int(st["shadow_nodes"])
                r.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    r.lbvh_rebuild_list(d1["list"], d1["t0"], d1["n"], d1["bmin"], d1["bmax"])
                r.synchronize()
                rec[tag + "_rebuild_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
            finally:
                r.close()
        rec["films_equal"] = bool(all(f.tobytes() == films["A_layout"].tobytes() for f in films.values()))
        rec["layout_gain_ms_per_frame"] = round(rec["B_walk_order_ms_per_frame"] - rec["A_layout_ms_per_frame"], 4)
        rec["layout_gain_frac"] = round(rec["layout_gain_ms_per_frame"] / rec["B_walk_order_ms_per_frame"], 4)
        print(json.dumps(rec), flush=True)
        lines.append(rec)
    if out:
        os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
        with open(out, "w") as f:
            for rec in lines:
                f.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
