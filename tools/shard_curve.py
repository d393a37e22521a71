#!/usr/bin/env python
"""Strong-scaling limit measured on ONE GPU: render only rank 0's tiles of an N-way screen shard (tile t -> rank t % N)
and time the frame.  With N GPUs every rank does this much work concurrently, so ms(N) is the compute part of an N-GPU
frame (the exchange, 33 MB / N per rank at 1080p, overlaps with the next frame -- DESIGN.md section 8).
    python tools/shard_curve.py [--scene sponza] [--steps 30]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="sponza")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--frames-in-flight", type=int, default=0)
    ap.add_argument("--regen", type=int, default=0, help="K > 0: path regeneration, bursts of K progressive frames (atn_render_burst)")
    ap.add_argument("--shards", default="1,2,4,8")
    ap.add_argument("--all-samples", action="store_true")
    ap.add_argument("--occupancy", action="store_true", help="with --regen: print the per-launch populations of one burst")
    args = ap.parse_args()
    from aten_amd.renderer import PathTracing
    from aten_amd.scene import scenedefs
    from aten_amd.scene.camera import create_camera
    fs, cam = {"sponza": scenedefs.sponza_lod, "cornell": scenedefs.cornell_box, "atrium": scenedefs.atrium}[args.scene]()
    W, H = args.width, args.height
    r = PathTracing(0)
    r.UpdateSceneData(fs)
    r.updateCamera(create_camera(cam["pos"], cam["at"], cam["vfov"], W, H))
    r.initSampler(W, H, 0)
    if args.frames_in_flight:
        r.set_frames_in_flight(args.frames_in_flight)
    out, occ = {}, {}
    brk = not args.all_samples
    r.set_regeneration(args.regen > 0)
    K = max(args.regen, 1)
    steps = (args.steps + K - 1) // K * K

    def run(first, n_frames):
        for i in range(first, first + n_frames, K):
            if args.regen:
                r.render_burst(W, H, K, args.depth, 3, spp=args.spp, frame=i, break_on_terminate=brk, download=False)
            else:
                r.render(W, H, args.depth, 3, spp=args.spp, frame=i, break_on_terminate=brk, download=False)

    for n in [int(x) for x in args.shards.split(",")]:
        r.setScreenShard(0, n)
        run(0, K if args.regen else 5)
        r.synchronize()
        t0 = time.perf_counter()
        run(0, steps)
        r.synchronize()
        out[n] = round(1e3 * (time.perf_counter() - t0) / steps, 4)
        if args.regen and args.occupancy:
            q, sh = r.regen_stage_counts()
            occ[n] = {"closest": [int(x) for x in q], "shadow": [int(x) for x in sh], "slots": int(r.tile_slots())}
    base = out.get(1, next(iter(out.values())))
    rec = {"scene": args.scene, "ms_per_frame_rank0_of_N": out, "speedup_bound": {n: round(base / v, 2) for n, v in out.items()},
           "frames_in_flight": args.frames_in_flight or 1, "regen_burst": args.regen, "spp": args.spp, "depth": args.depth,
           "break_on_terminate": brk, "size": [W, H]}
    if occ:
        rec["occupancy"] = occ
    print(json.dumps(rec))
    r.close()


if __name__ == "__main__":
    main()
