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
    out = {}
    for n in (1, 2, 4, 8):
        r.setScreenShard(0, n)
        for i in range(5):
            r.render(W, H, args.depth, 3, spp=args.spp, frame=i, download=False)
        r.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            r.render(W, H, args.depth, 3, spp=args.spp, frame=i, download=False)
        r.synchronize()
        out[n] = round(1e3 * (time.perf_counter() - t0) / args.steps, 4)
    base = out[1]
    print(json.dumps({"scene": args.scene, "ms_per_frame_rank0_of_N": out,
                      "speedup_bound": {n: round(base / v, 2) for n, v in out.items()},
                      "frames_in_flight": args.frames_in_flight or 1}))
    r.close()


if __name__ == "__main__":
    main()
