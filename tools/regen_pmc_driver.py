#!/usr/bin/env python
"""8 serial frames, then one regenerated burst of 8 (sponza_lod 1080p): the workload tools/regen_pmc.sh collects counters on."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aten_amd.renderer import PathTracing
from aten_amd.scene import scenedefs
from aten_amd.scene.camera import create_camera

mode = sys.argv[1] if len(sys.argv) > 1 else "both"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
fs, cam = scenedefs.sponza_lod()
W, H = 1920, 1080
r = PathTracing(0)
r.UpdateSceneData(fs)
r.updateCamera(create_camera(cam["pos"], cam["at"], cam["vfov"], W, H))
r.initSampler(W, H, 0)
if mode in ("serial", "both"):
    r.set_regeneration(False)
    for f in range(K):
        r.render(W, H, 5, 3, frame=f, download=False)
    r.synchronize()
if mode in ("regen", "both"):
    r.reset()
    r.set_regeneration(True)
    r.render_burst(W, H, K, 5, 3, frame=0, download=False)
    r.synchronize()
r.close()
