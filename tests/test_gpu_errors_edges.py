"""Error behaviour of the C-ABI (negative status + message, nothing thrown across the boundary, context stays usable)
and edge-case inputs: 1x1 and ragged frames, rays that all miss, a scene without lights, corrupted BVH links."""
import numpy as np
import pytest

from conftest import make_camera

pytestmark = pytest.mark.gpu


def _fresh():
    from aten_amd.renderer import PathTracing
    return PathTracing(0)


def test_calls_out_of_order_report_status(orc, cornell):
    from aten_amd.renderer import AtenAmdError
    fs, cam = cornell
    g = _fresh()
    try:
        with pytest.raises(AtenAmdError, match=r"atn_upload_scene.*status -4"):
            g.render(16, 16)
        g.UpdateSceneData(fs)
        with pytest.raises(AtenAmdError, match="atn_update_camera"):
            g.render(16, 16)
        g.updateCamera(make_camera(orc, cam, 16, 16))
        with pytest.raises(AtenAmdError, match="atn_init_sampler"):
            g.render(16, 16)
        g.initSampler(16, 16, 0)
        with pytest.raises(AtenAmdError, match="bad destination"):
            g.render(16, 0)
        with pytest.raises(AtenAmdError, match="bad destination"):
            g.render(16, 16, max_depth=0)
        g.setScreenShard(1, 2)
        with pytest.raises(AtenAmdError, match="one GPU"):
            g.svgf_render(16, 16, compute_motion=True)
        with pytest.raises(AtenAmdError):
            g.setScreenShard(2, 2)
        g.setScreenShard(0, 1)
        with pytest.raises(AtenAmdError, match="out of range"):
            g.set_path_batches(0)
        img = g.render(16, 16)          # the context survived all of that
        assert np.isfinite(img[..., :3]).all()
    finally:
        g.close()


def test_corrupted_bvh_is_rejected(orc, cornell):
    from aten_amd.renderer import AtenAmdError
    fs, cam = cornell
    g = _fresh()
    try:
        def broken(edit):
            import ctypes as C
            from aten_amd import layout as L
            from aten_amd.scene.builder import FlatScene
            lists = [a.copy() for a in fs.arrays["bvh_lists"]]
            edit(lists)
            arr = (L.BvhList * len(lists))()
            for i, n in enumerate(lists):
                arr[i].nodes = n.ctypes.data
                arr[i].count = len(n)
            b = FlatScene()
            C.memmove(C.byref(b.desc), C.byref(fs.desc), C.sizeof(fs.desc))
            b.desc.bvh_lists = C.addressof(arr)
            b.arrays = dict(fs.arrays)
            b.arrays["bvh_lists"] = lists
            b.keep = [fs, arr, lists]
            return b

        def cycle(lists):
            lists[1]["hit"][1] = 0.0        # second node links back to the root
        with pytest.raises(AtenAmdError, match="cycle|walk order"):
            g.UpdateSceneData(broken(cycle))

        def out_of_range(lists):
            lists[1]["miss"][0] = 1e6
        with pytest.raises(AtenAmdError, match="unreachable|out of range"):
            g.UpdateSceneData(broken(out_of_range))

        def bad_tri(lists):
            leaf = np.nonzero(lists[1]["f1"] >= 0)[0][0]
            lists[1]["f1"][leaf] = 1e6
        with pytest.raises(AtenAmdError, match="triangle id out of range"):
            g.UpdateSceneData(broken(bad_tri))

        def bad_blas(lists):
            leaf = np.nonzero(lists[0]["f2"] >= 0)[0][0]
            lists[0]["f2"][leaf] = np.int32(99).view(np.float32)
        with pytest.raises(AtenAmdError, match="missing BLAS"):
            g.UpdateSceneData(broken(bad_blas))

        big = max(range(1, len(fs.arrays["bvh_lists"])), key=lambda k: len(fs.arrays["bvh_lists"][k]))

        def backward_miss(lists):
            inner = np.nonzero((lists[big]["f0"] < 0) & (lists[big]["f1"] < 0))[0]
            lists[big]["miss"][inner[inner > 0][0]] = 0.0   # in range, reachable -- and an endless walk for every ray that misses it
        with pytest.raises(AtenAmdError, match="backward"):
            g.UpdateSceneData(broken(backward_miss))

        def backward_leaf(lists):
            leaf = np.nonzero(lists[big]["f1"] >= 0)[0][-1]
            lists[big]["hit"][leaf] = lists[big]["miss"][leaf] = 0.0
        with pytest.raises(AtenAmdError, match="backward|cycle"):
            g.UpdateSceneData(broken(backward_leaf))

        def backward_tlas(lists):
            leaf = np.nonzero(lists[0]["f2"] >= 0)[0][-1]
            lists[0]["miss"][leaf] = 0.0
        with pytest.raises(AtenAmdError, match="backward"):
            g.UpdateSceneData(broken(backward_tlas))

        def broken_array(name, edit):
            b = broken(lambda lists: None)
            arr = fs.arrays[name].copy()
            edit(arr)
            b.arrays[name] = arr
            b.keep.append(arr)
            setattr(b.desc, name, arr.ctypes.data)
            return b
        def bad_vertex(t): t["idx"][3][1] = len(fs.arrays["vtx_pos"]) + 7
        with pytest.raises(AtenAmdError, match="vertex index out of range"):
            g.UpdateSceneData(broken_array("triangles", bad_vertex))
        def bad_mtrl(t): t["mtrlid"][0] = 1000
        with pytest.raises(AtenAmdError, match="material id out of range"):
            g.UpdateSceneData(broken_array("triangles", bad_mtrl))
        def bad_light_obj(l): l["arealight_objid"][0] = 12345
        with pytest.raises(AtenAmdError, match="light refers to an object"):
            g.UpdateSceneData(broken_array("lights", bad_light_obj))
        def bad_obj_light(o): o["light_id"][0] = 77
        with pytest.raises(AtenAmdError, match="light id out of range"):
            g.UpdateSceneData(broken_array("objects", bad_obj_light))
        g.UpdateSceneData(fs)               # a good scene still uploads afterwards
    finally:
        g.close()


@pytest.mark.parametrize("size", [(1, 1), (7, 3), (9, 65), (130, 1)])
def test_tiny_and_ragged_frames(gpu, orc, cornell, size):
    fs, cam = cornell
    w, h = size
    c = make_camera(orc, cam, w, h)
    gpu.UpdateSceneData(fs)
    gpu.updateCamera(c)
    gpu.initSampler(w, h, 0)
    gpu.setScreenShard(0, 1)
    gpu.reset()
    seeds = orc.init_sampler(w, h, 0)
    got = gpu.render(w, h, 5, 3, frame=1)
    want = orc.render(fs, c, seeds, w, h, 5, 3, frame=1)
    assert got.shape == (h, w, 4)
    d = np.abs(got[..., :3] - want[..., :3])
    assert np.all((d <= 1e-3 * np.maximum(1.0, np.abs(want[..., :3]))) | (np.isnan(got[..., :3]) & np.isnan(want[..., :3])))
    rays = orc.generate_paths(c, seeds, w, h, 0, 1)
    assert gpu.generate_paths(w, h, 0, 1).tobytes() == rays.tobytes()


def test_all_rays_miss_and_no_lights(gpu, orc, cornell):
    """Camera looking away from the scene: every path ends in ShadeMiss at bounce 0 (background colour); and a scene
    without any light: NEE is skipped (lightnum <= 0, pathtracing_impl.h:199-203), only emissive hits contribute."""
    from aten_amd.scene import scenedefs
    fs, cam = cornell
    away = dict(cam)
    away["pos"], away["at"] = (0.0, 1.0, 8.0), (0.0, 1.0, 20.0)
    w, h = 48, 32
    c = make_camera(orc, away, w, h)
    gpu.UpdateSceneData(fs)
    gpu.updateCamera(c)
    gpu.initSampler(w, h, 0)
    gpu.setScreenShard(0, 1)
    gpu.reset()
    seeds = orc.init_sampler(w, h, 0)
    got = gpu.render(w, h, 5, 3, count_stats=True)
    st = gpu.stats()
    want = orc.render(fs, c, seeds, w, h, 5, 3)
    assert got.tobytes() == want.tobytes()
    assert st["hits"] == 0 and st["shadow_rays"] == 0 and st["closest_rays"] == w * h

    nolight = scenedefs.cornell_box_variant(lights="none", move_boxes=False)
    fs2, cam2 = nolight
    assert len(fs2.arrays["lights"]) == 0
    c2 = make_camera(orc, cam2, w, h)
    gpu.UpdateSceneData(fs2)
    gpu.updateCamera(c2)
    gpu.reset()
    got = gpu.render(w, h, 5, 3, count_stats=True)
    assert gpu.stats()["shadow_rays"] == 0
    want = orc.render(fs2, c2, seeds, w, h, 5, 3)
    d = np.abs(got[..., :3] - want[..., :3])
    assert np.all(d <= 1e-3 * np.maximum(1.0, np.abs(want[..., :3])))


def test_film_checkpoint_and_resume(orc, cornell):
    """Download the progressive film after 3 frames, restore it into a FRESH context and render frames 3..5 there: the
    result equals 6 uninterrupted frames byte for byte (FilmProgressive state = running mean + count, film.cpp:61-71)."""
    fs, cam = cornell
    w, h = 80, 48
    c = make_camera(orc, cam, w, h)

    def fresh():
        g = _fresh()
        g.UpdateSceneData(fs); g.updateCamera(c); g.initSampler(w, h, 0)
        return g
    a = fresh()
    try:
        for f in range(6):
            want = a.render(w, h, 5, 3, frame=f)
    finally:
        a.close()
    b = fresh()
    try:
        for f in range(3):
            ckpt = b.render(w, h, 5, 3, frame=f)
    finally:
        b.close()
    assert np.all(ckpt[..., 3] == 3.0)
    r = fresh()
    try:
        r.upload_film(ckpt)
        for f in range(3, 6):
            got = r.render(w, h, 5, 3, frame=f)
        assert got.tobytes() == want.tobytes()
        from aten_amd.renderer import AtenAmdError
        import ctypes as C
        assert r._l.atn_upload_film(r._ctx, 0, 4, ckpt.ctypes.data) != 0
    finally:
        r.close()


def test_bank_streams_run_side_by_side():
    """Frames in flight need one hardware queue per bank stream; the runtime multiplexes streams onto a few queues and two
    streams on one queue serialise.  atn_set_frames_in_flight measures the pairs and replaces the streams that clash."""
    from aten_amd.renderer import PathTracing
    r = PathTracing(0)
    try:
        # crowd the runtime's queue assignment the way torch.distributed does (a few more live streams) -- whatever it
        # hands out, the banks must end up pairwise concurrent
        r.set_frames_in_flight(3)
        # the probe is a wall-clock measurement (two 300 us spin kernels side by side or one after the other): on a GPU that
        # other processes use at the same moment a round can read "clash" spuriously -- ask up to three times
        for attempt in range(3):
            swaps, concurrent = r.bank_streams()
            if concurrent:
                break
        assert concurrent, "three bank streams on fewer than three hardware queues (%d swaps)" % swaps
        side = r.side_stream_ptr()
        assert side and side != r.stream_ptr() and side == r.side_stream_ptr()      # owned by the context, handed out again
        r.set_frames_in_flight(1)
        r.set_frames_in_flight(3)
        # three banks + the side stream = the four hardware queues there are: the side stream, handed out AFTER the banks were
        # separated, was probed against them (atn_side_stream) -- it runs beside every one of them
        ok = False
        for attempt in range(3):
            _, concurrent = r.bank_streams()
            ok = concurrent and r.side_stream_concurrent
            if ok:
                break
        assert ok, "banks pairwise concurrent: %s, side stream beside all of them: %s" % (concurrent, r.side_stream_concurrent)
        # a fourth bank created AFTER the side stream was handed out is probed against it and the other banks: five streams on
        # four queues cannot all overlap, the FIXED ones (main stream, side stream: the caller holds their handles) must not
        # have been touched, and the first three banks must still be concurrent with the side stream
        r.set_frames_in_flight(4)
        assert side == r.side_stream_ptr()
        r.set_frames_in_flight(3)
        assert any(r.bank_streams()[1] and r.side_stream_concurrent for _ in range(3))
    finally:
        r.close()
