#!/usr/bin/env python
"""Experiment: do frames in flight run in lockstep, and would a phase offset between the banks help?
Renders K frames with 3 in flight; variant 'stagger' delays the second / third bank's first frame by 1/3 and 2/3 of a frame
(torch.cuda._sleep on that bank's stream) and prints throughput for both."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aten_amd.renderer import PathTracing
from aten_amd.scene import scenedefs
from aten_amd.scene.camera import create_camera

scene = sys.argv[1] if len(sys.argv) > 1 else "sponza"
fs, cam = {"sponza": scenedefs.sponza_lod, "cornell": scenedefs.cornell_box, "atrium": scenedefs.atrium}[scene]()
W, H, K = 1920, 1080, 300
r = PathTracing(0)
r.UpdateSceneData(fs); r.updateCamera(create_camera(cam["pos"], cam["at"], cam["vfov"], W, H)); r.initSampler(W, H, 0)
r.set_frames_in_flight(3)
streams = []
for i in range(6):
    r.render(W, H, 5, 3, frame=i, download=False)
    streams.append(r.stream_ptr())
r.synchronize(); torch.cuda.synchronize()
banks = streams[3:6]
clock_hz = torch.cuda.get_device_properties(0).clock_rate * 1e3 if hasattr(torch.cuda.get_device_properties(0), "clock_rate") else 2.4e9

def run(offsets_ms):
    r.reset(); r.synchronize(); torch.cuda.synchronize()
    # the next frame goes to the bank after the last one used: learn the order by rendering, then delay
    t0 = time.perf_counter()
    for i in range(K):
        if i < 3 and offsets_ms[i] > 0:
            # the bank this frame will use is the one used 3 frames ago
            s = torch.cuda.ExternalStream(banks[i % 3])
            with torch.cuda.stream(s):
                torch.cuda._sleep(int(offsets_ms[i] * 1e-3 * 100e6 * 24))   # ~ cycles at 2.4 GHz
        r.render(W, H, 5, 3, frame=i, download=False)
    r.synchronize(); torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / K

for rep in range(2):
    print(scene, "lockstep  %.4f ms/frame" % run([0, 0, 0]))
    print(scene, "stagger   %.4f ms/frame" % run([0, 1.4, 2.8]))
    print(scene, "stagger2  %.4f ms/frame" % run([0, 2.1, 0.0]))
r.close()
