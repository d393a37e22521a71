import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _torch_runtime_first():
    """torch (device tensors for the tile / gather tests) bundles its own HIP runtime: when it initialises AFTER libaten_amd.so has
    brought up the system one it can report "No HIP GPUs are available" -- so on a GPU box it comes up before any test creates a
    context, whatever subset of the suite runs in whatever order (bench.py does the same)."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def cornell():
    from aten_amd.scene import scenedefs
    return scenedefs.cornell_box()


@pytest.fixture(scope="session")
def sponza():
    from aten_amd.scene import scenedefs
    return scenedefs.sponza_lod()


@pytest.fixture(scope="session")
def sponza_disney():
    from aten_amd import layout as L
    from aten_amd.scene import scenedefs
    return scenedefs.sponza_lod(mtype=L.MTRL_DISNEY)


@pytest.fixture(scope="session")
def gpu():
    """One PathTracing context on cuda:0; fails (does not skip) when the HIP library or GPU is missing."""
    # torch (device tensors for the tile / gather tests) bundles its own HIP runtime: when it initialises AFTER
    # libaten_amd.so has brought up the system one it reports "No HIP GPUs are available" -- so bring it up first, as
    # bench.py does
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    from aten_amd.renderer import PathTracing
    r = PathTracing(0)
    yield r
    r.close()


def make_camera(orc, cam, w, h):
    return orc.create_camera(cam["pos"], cam["at"], cam["vfov"], w, h)


def ulp_diff(a, b):
    """Distance in units in the last place between two float32 arrays (same sign assumed mostly)."""
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    ia = a.view(np.int32).astype(np.int64); ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, np.int64(-2**31) - ia, ia)
    ib = np.where(ib < 0, np.int64(-2**31) - ib, ib)
    return np.abs(ia - ib)


# ---- the parity report: what the full-size GPU-vs-oracle comparisons MEASURED, not only that they passed ---------------
PARITY_REPORT = os.environ.get("ATEN_PARITY_REPORT", os.path.join(ROOT, "gpurun_out", "parity", "parity_report.jsonl"))


def parity_metrics(got, want, tol=1e-3):
    """The numbers behind a frame comparison.  Tolerance of DESIGN.md section 4: |got - want| <= tol * max(1, |want|) per
    channel.  Pixels outside it are (expected to be) paths that took another branch after an ulp of difference in a
    transcendental: valid samples of the same estimator (tests/test_gpu_convergence.py checks that they are)."""
    a = np.asarray(got)[..., :3].astype(np.float64)
    b = np.asarray(want)[..., :3].astype(np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    d = np.where(both_nan, 0.0, np.abs(a - b))
    scale = np.maximum(1.0, np.abs(b))
    rel = d / scale
    inside = np.all((rel <= tol) | both_nan, axis=-1)
    relpix = np.nanmax(rel, axis=-1)
    finite = np.isfinite(relpix)
    ma, mb = np.nanmean(a), np.nanmean(b)
    q = lambda x: float(np.quantile(relpix[finite], x)) if finite.any() else None
    return {
        "pixels": int(inside.size),
        "frac_within_%g" % tol: float(inside.mean()),
        "pixels_outside": int((~inside).sum()),
        "frac_bit_equal": float(np.all((a == b) | both_nan, axis=-1).mean()),
        "max_abs_err": float(np.nanmax(d)),
        "max_rel_err": float(np.nanmax(relpix[finite])) if finite.any() else None,
        "p50_rel_err": q(0.5), "p99_rel_err": q(0.99), "p99.9_rel_err": q(0.999), "p99.99_rel_err": q(0.9999),
        "median_rel_err_of_pixels_inside": float(np.median(relpix[inside])) if inside.any() else None,
        "image_mean_relerr": float(abs(ma - mb) / max(abs(mb), 1e-12)),
        "nonfinite_pixels_got": int((~np.isfinite(a)).any(-1).sum()), "nonfinite_pixels_want": int((~np.isfinite(b)).any(-1).sum()),
    }


def parity_record(config, got, want, tol=1e-3, gpu_stats=None, oracle_counters=None, **extra):
    """Appends one JSON line to the parity report (gpurun_out/parity/parity_report.jsonl on the GPU box; the round's copy is
    profiles/parity_rNN.json) and returns the metrics.  gpu_stats / oracle_counters: the work counters of the SAME frame on
    both sides (atn_get_stats after a counted render; orc.render(..., counters=True)) -- their deltas count diverged paths."""
    import json
    m = parity_metrics(got, want, tol)
    rec = {"config": config, "tolerance": "|got - want| <= %g * max(1, |want|) per channel" % tol}
    rec.update(m)
    if gpu_stats is not None and oracle_counters is not None:
        oc = dict(closest_rays=int(oracle_counters[0]), shadow_rays=int(oracle_counters[1]), hits=int(oracle_counters[2]),
                  node_visits=int(oracle_counters[3]), triangle_tests=int(oracle_counters[4]))
        gc = dict(closest_rays=int(gpu_stats["closest_rays"]), shadow_rays=int(gpu_stats["shadow_rays"]), hits=int(gpu_stats["hits"]),
                  node_visits=int(gpu_stats["closest_nodes"] + gpu_stats["shadow_nodes"]),
                  triangle_tests=int(gpu_stats["closest_tris"] + gpu_stats["shadow_tris"]))
        rec["counters_oracle"] = oc
        rec["counters_gpu"] = gc
        rec["n_paths_diverged"] = {k: gc[k] - oc[k] for k in ("closest_rays", "shadow_rays", "hits")}
        rec["counter_relative_delta"] = {k: (gc[k] - oc[k]) / max(oc[k], 1) for k in oc}
    rec.update(extra)
    try:
        os.makedirs(os.path.dirname(PARITY_REPORT), exist_ok=True)
        with open(PARITY_REPORT, "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass
    return m
