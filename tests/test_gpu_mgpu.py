"""The node-wide renderer behind the C-ABI (atn_mgpu_*, aten_amd/csrc/host/mgpu.hpp): however many GPUs are visible
(one on the test box), and N > 1 shards sharing device 0 so that the worker threads, screen shards, tile pushes,
double-buffered gather and assembly all run.  Contract: the assembled frame equals the single-context frame byte for
byte, whatever N is (tiles are interleaved, seeds and pixel indices global)."""
import numpy as np
import pytest

from aten_amd.scene.camera import create_camera

pytestmark = pytest.mark.gpu


def _single(fs, cam, w, h, frames, depth=5, spp=1, brk=True):
    from aten_amd.renderer import PathTracing
    r = PathTracing(0)
    try:
        r.UpdateSceneData(fs)
        r.updateCamera(create_camera(cam["pos"], cam["at"], cam["vfov"], w, h))
        r.initSampler(w, h, 0)
        out = [r.render(w, h, depth, 3, spp=spp, frame=f, break_on_terminate=brk).copy() for f in range(frames)]
    finally:
        r.close()
    return out


def _multi(devices, fs, cam, w, h, frames, depth=5, spp=1, brk=True, download_each=True):
    from aten_amd.renderer import MultiGpuPathTracing
    m = MultiGpuPathTracing(devices)
    try:
        m.UpdateSceneData(fs)
        m.updateCamera(create_camera(cam["pos"], cam["at"], cam["vfov"], w, h))
        m.initSampler(w, h, 0)
        out = []
        for f in range(frames):
            img = m.render(w, h, depth, 3, spp=spp, frame=f, break_on_terminate=brk, download=download_each)
            if download_each:
                out.append(img.copy())
        if not download_each:
            m.synchronize()
            out.append(m.download_film())
        n = m.shard_count()
    finally:
        m.close()
    return out, n


def test_every_visible_gpu_equals_one_gpu(cornell):
    import torch
    fs, cam = cornell
    w, h = 200, 120
    want = _single(fs, cam, w, h, 3)
    got, n = _multi(None, fs, cam, w, h, 3)
    assert n == torch.cuda.device_count() >= 1
    for a, b in zip(got, want):
        assert a.tobytes() == b.tobytes()


@pytest.mark.parametrize("shards", [1, 2, 3, 8])
def test_shards_sharing_one_gpu_equal_unsharded(cornell, shards):
    fs, cam = cornell
    for (w, h) in ((96, 64), (100, 52)):            # second one: ragged tiles, tile count not a multiple of the shard count
        want = _single(fs, cam, w, h, 4)
        got, n = _multi([0] * shards, fs, cam, w, h, 4)
        assert n == shards
        for f, (a, b) in enumerate(zip(got, want)):
            assert a.tobytes() == b.tobytes(), (shards, w, h, f)


def test_frames_in_flight_then_one_download(sponza):
    """No host synchronisation between frames (download only at the end): the double-buffered gather and the
    progressive film on each shard must still produce the single-context running mean."""
    fs, cam = sponza
    w, h = 320, 180
    want = _single(fs, cam, w, h, 6)[-1]
    got, _ = _multi([0, 0, 0, 0], fs, cam, w, h, 6, download_each=False)
    assert got[0].tobytes() == want.tobytes()


def test_multi_sample_frames_and_all_samples(cornell):
    fs, cam = cornell
    w, h = 64, 64
    for brk in (True, False):
        want = _single(fs, cam, w, h, 2, depth=4, spp=3, brk=brk)
        got, _ = _multi([0, 0], fs, cam, w, h, 2, depth=4, spp=3, brk=brk)
        for a, b in zip(got, want):
            assert a.tobytes() == b.tobytes()


def test_errors(cornell):
    from aten_amd._lib import lib
    from aten_amd.renderer import AtenAmdError, MultiGpuPathTracing
    with pytest.raises(AtenAmdError):
        MultiGpuPathTracing(10 ** 6)                # more devices than exist
    with pytest.raises(AtenAmdError):
        MultiGpuPathTracing([0, 12345])             # bad ordinal
    m = MultiGpuPathTracing([0, 0])
    try:
        with pytest.raises(AtenAmdError, match="shard 0.*atn_upload_scene"):
            m.render(16, 16)
        fs, cam = cornell
        m.UpdateSceneData(fs)
        m.updateCamera(create_camera(cam["pos"], cam["at"], cam["vfov"], 16, 16))
        m.initSampler(16, 16, 0)
        img = m.render(16, 16)
        assert np.isfinite(img[..., :3]).all()
    finally:
        m.close()
    assert lib().atn_mgpu_render(None, None, None) == -1


@pytest.mark.parametrize("in_flight", [2, 3, 4])
def test_frames_in_flight_equal_serial_frames(sponza, in_flight):
    """atn_set_frames_in_flight: consecutive frames on rotating banks of path state and streams, ordered only by the
    film.  The progressive film after 7 frames, every downloaded intermediate frame, a counted frame in the middle and
    a shard change must all equal the one-frame-at-a-time result, byte for byte."""
    from aten_amd.renderer import PathTracing
    fs, cam = sponza
    w, h = 320, 180
    want = _single(fs, cam, w, h, 7)
    r = PathTracing(0)
    try:
        r.UpdateSceneData(fs)
        r.updateCamera(create_camera(cam["pos"], cam["at"], cam["vfov"], w, h))
        r.initSampler(w, h, 0)
        r.set_frames_in_flight(in_flight)
        for f in range(7):
            r.render(w, h, 5, 3, frame=f, download=False)
        r.synchronize()
        assert r.download_film().tobytes() == want[6].tobytes()
        r.reset()
        for f in range(7):
            img = r.render(w, h, 5, 3, frame=f, download=(f in (2, 5)), count_stats=(f == 3))
            if img is not None:
                assert img.tobytes() == want[f].tobytes(), f
            if f == 3:
                assert r.stats()["closest_rays"] >= w * h
        assert r.download_film().tobytes() == want[6].tobytes()
        # back to one frame at a time on the same context
        r.set_frames_in_flight(1)
        r.reset()
        for f in range(3):
            img = r.render(w, h, 5, 3, frame=f)
        assert img.tobytes() == want[2].tobytes()
    finally:
        r.close()


def test_frames_in_flight_on_shards(cornell):
    from aten_amd.renderer import MultiGpuPathTracing
    fs, cam = cornell
    w, h = 160, 96
    want = _single(fs, cam, w, h, 6)
    m = MultiGpuPathTracing([0, 0, 0])
    try:
        m.UpdateSceneData(fs)
        m.updateCamera(create_camera(cam["pos"], cam["at"], cam["vfov"], w, h))
        m.initSampler(w, h, 0)
        m.set_frames_in_flight(2)
        for f in range(6):
            m.render(w, h, 5, 3, frame=f, download=False)
        m.synchronize()
        assert m.download_film().tobytes() == want[5].tobytes()
    finally:
        m.close()
