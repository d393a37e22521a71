"""Times atn_lbvh_rebuild_list (the per-tick LBVH rebuild of a deforming mesh) on the GPU for a few mesh sizes.
Usage (GPU box): python tools/lbvh_bench.py [n_triangles ...]      prints one JSON line per size."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aten_amd import layout as L                      # noqa: E402
from aten_amd.renderer import PathTracing             # noqa: E402
from aten_amd.scene.builder import SceneBuilder       # noqa: E402


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [12852, 100_000, 1_000_000]
    for n in sizes:
        rng = np.random.default_rng(n)
        c = rng.random((n, 1, 3), dtype=np.float32) * 10
        p = (c + (rng.random((n, 3, 3), dtype=np.float32) - 0.5) * 0.1).reshape(-1, 3)
        b = SceneBuilder()
        m = b.add_material("m", L.MTRL_DIFFUSE, (0.5, 0.5, 0.5))
        oid = b.add_mesh("soup", p, np.arange(3 * n).reshape(n, 3), m)
        b.create_instance(oid)
        fs = b.build()
        r = PathTracing(0)
        r.UpdateSceneData(fs)
        k = fs.blas_index[oid]
        bmin, bmax = p.min(0), p.max(0)
        for _ in range(3):
            r.lbvh_rebuild_list(k, 0, n, bmin, bmax)
        r.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            r.lbvh_rebuild_list(k, 0, n, bmin, bmax)        # returns when enqueued; rebuilds run back to back on one stream
        r.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(json.dumps(dict(n_triangles=n, rebuild_ms=round(dt * 1e3, 4), mtris_per_s=round(n / dt / 1e6, 2))), flush=True)
        r.close()


if __name__ == "__main__":
    main()
