"""Direction lists on the GPU (atn_bvh_list.direction_axes, include/aten_layout.h): a bottom-level list stored as 2^popcount(axes)
segments -- the same tree threaded in different child orders -- of which a ray walks the one the signs of its direction inside the
instance select.  The CPU oracle applies the same rule (ATN_DIRECTION_SEGMENT) to the same bits, so `Intersection` records AND
visit counters must equal the oracle's on every walk flavour and on every way into a nested tree (direct start with and without a
matrix, TLAS leaves, identity instances over an LDS copy), and frames stay inside the stated tolerance."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import make_camera
from test_gpu_parity import frame_tolerance_report

pytestmark = pytest.mark.gpu


def _incoherent(rays, scale, rng):
    extra = rays.copy()
    d = rng.normal(size=(len(extra), 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[::97, 0] = 0.0; d[5::89, 1] = -0.0; d[7::83, 2] = 0.0           # zero components: bit = (dir > 0) is 0 for both zeros
    extra["dir"][:, :3] = d
    extra["org"][:, :3] = (rays["org"][:, :3] + rng.uniform(-scale, scale, size=(len(extra), 3))).astype(np.float32)
    return extra


def _flavours(fs, c, w, h, rays_list, want_list, lds_modes=("0",), flavours=("r", "s"), frame_args=(4, 3)):
    from aten_amd.renderer import PathTracing
    films = {}
    old = {k: os.environ.get(k) for k in ("ATEN_AMD_TRACE", "ATEN_AMD_LDS_NODES")}
    try:
        for flavour in flavours:
            for lds in lds_modes:
                if flavour == "r" and lds == "1":
                    continue
                os.environ["ATEN_AMD_TRACE"] = flavour; os.environ["ATEN_AMD_LDS_NODES"] = lds
                r = PathTracing(0)
                try:
                    r.UpdateSceneData(fs); r.updateCamera(c); r.initSampler(w, h, 0)
                    for rays, (want, wst) in zip(rays_list, want_list):
                        got, st = r.trace_closest(rays, stats=True)
                        assert got.tobytes() == want.tobytes(), (flavour, lds)
                        assert np.array_equal(st, wst), (flavour, lds, st, wst)
                    films[(flavour, lds)] = r.render(w, h, *frame_args, frame=1).copy()
                finally:
                    r.close()
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    return films


@pytest.mark.parametrize("axes", [1, 5, 7])
def test_one_instance_direct_start(orc, axes):
    """sponza_lod through the own builder with direction lists (the top layer is one leaf: walks start inside the nested tree and
    pick their segment in walk_start): primary and incoherent rays, refill and plain walk; the frame inside the oracle's tolerance,
    byte-equal between the flavours, and fewer node visits than the single list of the same tree."""
    from aten_amd.scene import scenedefs
    fs, cam = scenedefs.sponza_lod(use_sbvh=False, bvh_options=dict(order_point=(0.0, 1.0, 3.0), direction_axes=axes))
    one, _ = scenedefs.sponza_lod(use_sbvh=False)
    w, h = 128, 72
    c = make_camera(orc, cam, w, h)
    seeds = orc.init_sampler(w, h, 0)
    rays = orc.generate_paths(c, seeds, w, h, 0, 0)
    extra = _incoherent(rays, 0.5, np.random.default_rng(axes))
    want = [orc.trace_closest(fs, rays), orc.trace_closest(fs, extra)]
    assert (want[1][0]["objid"] >= 0).mean() > 0.5
    assert want[1][1][0] < 0.99 * orc.trace_closest(one, extra)[1][0]
    films = _flavours(fs, c, w, h, [rays, extra], want, frame_args=(5, 3))
    assert films[("r", "0")].tobytes() == films[("s", "0")].tobytes()
    inside, mean_err = frame_tolerance_report(films[("r", "0")], orc.render(fs, c, seeds, w, h, 5, 3, frame=1))
    assert inside >= 0.995 and mean_err <= 5e-3, (inside, mean_err)


def test_instances_behind_a_top_layer(orc, monkeypatch):
    """The Cornell box variant -- eight instances, two of them rotated and moved, six with the identity matrix -- with a direction
    list (x and z) under every instance: the segment is picked at the TLAS leaf from the ray INSIDE the instance (for identity
    instances over an LDS copy: the re-normalised local ray computed once per ray)."""
    from aten_amd.scene import builder, scenedefs
    monkeypatch.setattr(builder, "DEFAULT_BVH_OPTIONS", dict(direction_axes=5))
    fs, cam = scenedefs.cornell_box_variant(lights="mixed")
    assert all(a == 5 for a in fs.arrays["bvh_list_axes"][1:]) and fs.arrays["bvh_list_axes"][0] == 0
    w, h = 96, 96
    c = make_camera(orc, cam, w, h)
    seeds = orc.init_sampler(w, h, 0)
    rays = orc.generate_paths(c, seeds, w, h, 0, 0)
    extra = _incoherent(rays, 0.8, np.random.default_rng(9))
    want = [orc.trace_closest(fs, rays), orc.trace_closest(fs, extra)]
    films = _flavours(fs, c, w, h, [rays, extra], want, lds_modes=("0", "1"))
    assert films[("r", "0")].tobytes() == films[("s", "0")].tobytes() == films[("s", "1")].tobytes()
    inside, mean_err = frame_tolerance_report(films[("s", "1")], orc.render(fs, c, seeds, w, h, 4, 3, frame=1))
    assert inside >= 0.995 and mean_err <= 5e-3, (inside, mean_err)


def test_scaled_instances_of_one_tree(gpu, orc):
    """The atrium stand-in at reduced tessellation (several scaled and rotated instances of one bottom-level tree, Disney + maps +
    IBL + area light) with eight segments per list: counters of a whole 5-bounce frame (closest rays, shadow rays, hits) equal
    the oracle's, the film is inside its tolerance."""
    from aten_amd.scene import scenedefs
    fs, cam = scenedefs.atrium(detail=0.25, bvh_options=dict(order_point=(-7.0, 1.7, 0.6), direction_axes=7))
    w, h = 160, 90
    c = make_camera(orc, cam, w, h)
    gpu.UpdateSceneData(fs); gpu.updateCamera(c); gpu.initSampler(w, h, 0); gpu.setScreenShard(0, 1); gpu.reset()
    seeds = orc.init_sampler(w, h, 0)
    rays = orc.generate_paths(c, seeds, w, h, 0, 0)
    want_i, wst = orc.trace_closest(fs, rays)
    got_i, gst = gpu.trace_closest(rays, stats=True)
    assert got_i.tobytes() == want_i.tobytes() and np.array_equal(gst, wst)
    got = gpu.render(w, h, 3, 3, frame=0)
    want = orc.render(fs, c, seeds, w, h, 3, 3, frame=0)
    inside, mean_err = frame_tolerance_report(got, want)
    assert inside >= 0.998 and mean_err <= 1e-2, (inside, mean_err)


def test_upload_refuses_lists_that_are_not_segments_of_one_tree(gpu):
    """direction_axes on the top layer or beyond three axes, a node count that is not a multiple of the segment count: atn_upload_scene
    returns an error instead of walking them; and a direction list is not a target for the LBVH rebuild."""
    from aten_amd.scene import scenedefs
    fs, cam = scenedefs.sponza_lod(use_sbvh=False, bvh_options=dict(direction_axes=1))
    nodes = fs.arrays["bvh_lists"][1].copy()
    gpu.UpdateSceneData(fs)                                                     # the well-formed list uploads
    with pytest.raises(Exception):
        gpu.lbvh_rebuild_list(1, 0, int(fs.arrays["objects"][0]["triangle_num"]), (-1, -1, -1), (1, 1, 1))
    fs.replace_bvh_list(1, nodes[:-1], direction_axes=1)
    with pytest.raises(Exception, match="multiple"):
        gpu.UpdateSceneData(fs)
    fs.replace_bvh_list(1, nodes, direction_axes=9)
    with pytest.raises(Exception, match="direction_axes"):
        gpu.UpdateSceneData(fs)
    fs.replace_bvh_list(1, nodes, direction_axes=1)
    fs.lists[0].direction_axes = 1
    with pytest.raises(Exception, match="direction_axes"):
        gpu.UpdateSceneData(fs)
    fs.lists[0].direction_axes = 0
    gpu.UpdateSceneData(fs)
