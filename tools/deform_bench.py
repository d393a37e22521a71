"""The reference's deformation-renderer loop on the GPU: per tick new vertices (host arrays, as a CPU skinning step would
hand them over) -> atn_update_geometry -> atn_lbvh_rebuild_list -> atn_update_tlas -> atn_render, at 1080p.
Usage (GPU box): python tools/deform_bench.py [nu nv]     prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aten_amd.renderer import PathTracing             # noqa: E402
from aten_amd.scene import scenedefs                  # noqa: E402
from aten_amd.scene.camera import create_camera       # noqa: E402


def main():
    nu, nv = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 128)
    W, H = 1920, 1080
    b, oid, cam = scenedefs.deformable_room(0.0, nu=nu, nv=nv)
    ticks = []
    for k in range(6):
        pos, nml, idx = scenedefs.blob_mesh(0.4 * k, nu=nu, nv=nv)
        b.set_mesh_vertices(oid, pos, idx, nml)
        fs = b.build()
        o = fs.arrays["objects"][oid]
        t0, n = int(o["triangle_id"]), int(o["triangle_num"])
        tris = fs.arrays["triangles"][t0:t0 + n]
        v0, v1 = int(tris["idx"].min()), int(tris["idx"].max()) + 1
        used = fs.arrays["vtx_pos"][v0:v1, :3]
        ticks.append((fs, dict(list=fs.blas_index[oid], t0=t0, n=n, v0=v0, v1=v1, bmin=used.min(0), bmax=used.max(0))))
    r = PathTracing(0)
    r.UpdateSceneData(ticks[0][0])
    r.updateCamera(create_camera(cam["pos"], cam["at"], cam["vfov"], W, H))
    r.initSampler(W, H, 0)
    r.set_frames_in_flight(3)

    def tick(i, render=True, only=None):
        fs, d = ticks[i % len(ticks)]
        a = fs.arrays
        t = time.perf_counter()
        if only in (None, "geom"):
            r.updateGeometry(vtx_pos=a["vtx_pos"][d["v0"]:d["v1"]], vtx_nml=a["vtx_nml"][d["v0"]:d["v1"]], vtx_offset=d["v0"],
                             triangles=a["triangles"][d["t0"]:d["t0"] + d["n"]], tri_offset=d["t0"])
        if only in (None, "lbvh"):
            r.lbvh_rebuild_list(d["list"], d["t0"], d["n"], d["bmin"], d["bmax"])
        if only in (None, "tlas"):
            r.updateBVH(fs)
        if render and only is None:
            r.render(W, H, 5, 3, frame=i, download=False)
        return time.perf_counter() - t

    for i in range(4):
        tick(i)
    r.synchronize()
    n_ticks = 30
    t0 = time.perf_counter()
    for i in range(n_ticks):
        tick(i)
    r.synchronize()
    whole = (time.perf_counter() - t0) / n_ticks
    parts = {}
    for what in ("geom", "lbvh", "tlas"):
        r.synchronize()
        parts[what] = float(np.median([tick(i, only=what) for i in range(12)]))
    t0 = time.perf_counter()
    for i in range(n_ticks):
        r.render(W, H, 5, 3, frame=i, download=False)
    r.synchronize()
    static = (time.perf_counter() - t0) / n_ticks
    r.set_frames_in_flight(1)
    for i in range(3):
        r.render(W, H, 5, 3, frame=i, download=False)
    r.synchronize()
    t0 = time.perf_counter()
    for i in range(n_ticks):
        r.render(W, H, 5, 3, frame=i, download=False)
    r.synchronize()
    serial = (time.perf_counter() - t0) / n_ticks
    print(json.dumps(dict(workload="Cornell box + deforming blob, %d triangles, %dx%d 1spp 5-bounce" % (ticks[0][1]["n"], W, H),
                          ms_per_tick_and_frame=round(whole * 1e3, 3), ms_per_frame_static=round(static * 1e3, 3), ms_per_frame_static_one_in_flight=round(serial * 1e3, 3),
                          ms_update_geometry=round(parts["geom"] * 1e3, 3), ms_lbvh_rebuild=round(parts["lbvh"] * 1e3, 3),
                          ms_update_tlas=round(parts["tlas"] * 1e3, 3))), flush=True)
    r.close()


if __name__ == "__main__":
    main()
