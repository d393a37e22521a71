#!/usr/bin/env python
"""Where a regenerated burst spends its time, next to the serial loop: per-kernel HIP-event times (atn_get_kernel_times) and the
per-launch populations of the pool (atn_regen_stage_counts).
    python tools/regen_diag.py [--scene sponza] [--shard 1] [--burst 8] [--spp 1]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="sponza")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--burst", type=int, default=8)
    ap.add_argument("--shard", type=int, default=1)
    ap.add_argument("--all-samples", action="store_true")
    args = ap.parse_args()
    from aten_amd.renderer import PathTracing
    from aten_amd.scene import scenedefs
    from aten_amd.scene.camera import create_camera
    fs, cam = {"sponza": scenedefs.sponza_lod, "cornell": scenedefs.cornell_box, "atrium": scenedefs.atrium}[args.scene]()
    W, H, K = args.width, args.height, args.burst
    r = PathTracing(0)
    r.UpdateSceneData(fs)
    r.updateCamera(create_camera(cam["pos"], cam["at"], cam["vfov"], W, H))
    r.initSampler(W, H, 0)
    r.setScreenShard(0, args.shard)
    brk = not args.all_samples
    out = {"scene": args.scene, "size": [W, H], "spp": args.spp, "depth": args.depth, "burst": K, "shard": args.shard, "break_on_terminate": brk}
    for mode in ("serial", "regen"):
        r.set_regeneration(mode == "regen")
        r.reset()

        def run(profile):
            if mode == "regen":
                r.render_burst(W, H, K, args.depth, 3, spp=args.spp, frame=0, break_on_terminate=brk, download=False, profile=profile)
            else:
                for f in range(K):
                    r.render(W, H, args.depth, 3, spp=args.spp, frame=f, break_on_terminate=brk, download=False, profile=profile)
        run(False); r.synchronize()
        t0 = time.perf_counter(); run(False); r.synchronize()
        wall = 1e3 * (time.perf_counter() - t0) / K
        r.reset_kernel_times()
        run(True); r.synchronize()
        kt = r.kernel_times()
        out[mode] = {"ms_per_frame": round(wall, 4),
                     "kernels_ms_per_frame": {k: [round(v[0] / K, 4), v[1]] for k, v in kt.items() if v[1]}}
        if mode == "regen":
            q, sh = r.regen_stage_counts()
            out[mode]["slots"] = int(r.tile_slots())
            out[mode]["closest_per_stage"] = [int(x) for x in q]
            out[mode]["shadow_per_stage"] = [int(x) for x in sh]
            out[mode]["nonempty_stages"] = int((q > 0).sum())
    print(json.dumps(out))
    r.close()


if __name__ == "__main__":
    main()
