"""BASELINE config 4 at size: "Crytek Sponza 4K 8spp 8-bounce, Disney BRDF + textures".  The Crytek blob is missing from
the reference snapshot; `scenedefs.atrium()` (full tessellation: 250 882 triangles, a 460 799-node bottom-level tree
~17 MB that does NOT fit a 4 MiB per-XCD L2, six instanced statues, Disney + Sponza textures, IBL + area light) is the
stand-in of the same size class (shape of the workload: src/common/scenedefs.cpp:940-983).

  * 4K primary rays: rays, `Intersection` records and visit counters byte-equal to the oracle (8.3 M rays);
  * 4K / 8 spp / 8 bounces, with the CPU renderer's break after a terminated sample and with every sample traced:
    run-to-run determinism, film sample counts, 2-way screen shard + assembly == unsharded, byte for byte;
  * the same two sample-loop modes against the oracle at 640x360 within the stated tolerance.
"""
import numpy as np
import pytest

from aten_amd.scene.camera import create_camera
from conftest import parity_record
from test_gpu_parity import frame_tolerance_report

pytestmark = pytest.mark.gpu

W4K, H4K = 3840, 2160


@pytest.fixture(scope="module")
def atrium():
    from aten_amd.scene import scenedefs
    fs, cam = scenedefs.atrium()
    assert len(fs.arrays["triangles"]) > 250000
    return fs, cam


@pytest.fixture(scope="module")
def ctx(atrium):
    from aten_amd.renderer import PathTracing
    fs, cam = atrium
    r = PathTracing(0)
    r.UpdateSceneData(fs)
    yield r
    r.close()


def test_primary_rays_4k_bit_exact(ctx, orc, atrium):
    fs, cam = atrium
    c = create_camera(cam["pos"], cam["at"], cam["vfov"], W4K, H4K)
    ctx.updateCamera(c)
    ctx.initSampler(W4K, H4K, 0)
    seeds = orc.init_sampler(W4K, H4K, 0)
    rays = orc.generate_paths(c, seeds, W4K, H4K, 0, 0)
    assert ctx.generate_paths(W4K, H4K, 0, 0).tobytes() == rays.tobytes()
    want, wst = orc.trace_closest(fs, rays)
    got, gst = ctx.trace_closest(rays, stats=True)
    assert got.tobytes() == want.tobytes()
    assert np.array_equal(gst, wst)                 # node visits and triangle tests of 8.3 M walks
    assert (want["objid"] >= 0).mean() > 0.5        # the frame is mostly geometry, not sky


@pytest.mark.parametrize("brk", [True, False], ids=["break_on_terminate", "all_samples"])
def test_4k_8spp_8bounce_properties(ctx, atrium, brk):
    from aten_amd.renderer import MultiGpuPathTracing
    fs, cam = atrium
    c = create_camera(cam["pos"], cam["at"], cam["vfov"], W4K, H4K)
    ctx.updateCamera(c)
    ctx.initSampler(W4K, H4K, 0)
    ctx.setScreenShard(0, 1)
    frames = []
    for run in range(2):
        ctx.reset()
        a = ctx.render(W4K, H4K, 8, 3, spp=8, frame=0, break_on_terminate=brk, count_stats=(run == 0)).copy()
        if run == 0:
            st = ctx.stats()
        b = ctx.render(W4K, H4K, 8, 3, spp=8, frame=1, break_on_terminate=brk).copy()
        frames.append((a, b))
    # determinism: queue order is racy, pixels are not
    assert frames[0][0].tobytes() == frames[1][0].tobytes()
    assert frames[0][1].tobytes() == frames[1][1].tobytes()
    a, b = frames[0]
    # FilmProgressive: w counts the frames put so far (film.cpp:61-71)
    assert np.all(a[..., 3] == 1.0) and np.all(b[..., 3] == 2.0)
    assert np.isfinite(a[..., :3]).all() and (a[..., :3] >= 0).all()
    px = W4K * H4K
    if brk:
        # a pixel stops sampling after its first terminated path: far fewer than 8 camera paths per pixel
        assert px <= st["closest_rays"] and st["closest_rays"] < 8 * px * 8
    else:
        assert st["closest_rays"] >= 8 * px         # every sample's primary ray at least
    assert st["shadow_rays"] > 0 and st["hits"] > 0
    # 2-way screen shard (tiles t % 2) on one GPU + assembly == the unsharded frames
    m = MultiGpuPathTracing([0, 0])
    try:
        m.UpdateSceneData(fs)
        m.updateCamera(c)
        m.initSampler(W4K, H4K, 0)
        sa = m.render(W4K, H4K, 8, 3, spp=8, frame=0, break_on_terminate=brk)
        sb = m.render(W4K, H4K, 8, 3, spp=8, frame=1, break_on_terminate=brk)
    finally:
        m.close()
    assert sa.tobytes() == a.tobytes()
    assert sb.tobytes() == b.tobytes()


@pytest.mark.parametrize("brk", [True, False], ids=["break_on_terminate", "all_samples"])
def test_8spp_8bounce_vs_oracle_640x360(ctx, orc, atrium, brk):
    """Both sample-loop modes against the CPU oracle at a size it renders in seconds.  Depth 8 on displaced, smooth-shaded
    geometry decorrelates paths whose sinf/cosf differ by an ulp (see test_atrium_instanced_disney_textured), so the
    per-pixel band is wide and the mean is what is tight."""
    fs, cam = atrium
    w, h = 640, 360
    c = create_camera(cam["pos"], cam["at"], cam["vfov"], w, h)
    ctx.updateCamera(c)
    ctx.initSampler(w, h, 0)
    ctx.setScreenShard(0, 1)
    ctx.reset()
    seeds = orc.init_sampler(w, h, 0)
    got = ctx.render(w, h, 8, 3, spp=8, frame=0, break_on_terminate=brk, progressive=False)
    if brk:
        want = orc.render(fs, c, seeds, w, h, 8, 3, spp=8, frame=0)
    else:
        # every sample traced = the mean of 8 one-sample frames with frame + sample as the sampler's frame index;
        # invalid samples are skipped in both (pathtracing.cpp:339-347)
        acc = np.zeros((h, w, 3), np.float64); cnt = np.zeros((h, w), np.float64)
        for s in range(8):
            f1 = orc.render(fs, c, seeds, w, h, 8, 3, spp=1, frame=s)
            v = f1[..., :3].astype(np.float64)
            ok = np.isfinite(v).all(-1) & (v >= 0).all(-1)
            acc[ok] += v[ok]; cnt += ok
        want = np.zeros((h, w, 4), np.float32)
        want[..., :3] = (acc / np.maximum(cnt, 1)[..., None]).astype(np.float32)
    frac, mean_err = frame_tolerance_report(got, want)
    parity_record("C4 stand-in at oracle size: atrium 640x360 8spp 8-bounce Disney, %s" % ("break on terminate (CPU renderer's sample loop)" if brk else "all samples traced"),
                  got, want)
    # one depth-8 sample is inside the band for ~95-97 % of the pixels; a pixel that averages 8 independent samples is
    # inside only if all eight are (0.95^8 = 0.66), while with the break most pixels stop after their first sample
    # measured (profiles/parity_r05.json, unchanged in r06): 0.9325 / 0.6701 -- the floors are those minus a margin for another box's ocml
    assert frac >= (0.92 if brk else 0.65), (brk, frac)
    assert mean_err <= 1e-2, (brk, mean_err)


def test_companion_workload_1080p_vs_oracle(ctx, orc, atrium):
    """The workload bench.py reports under `companion` (atrium, 1920x1080, 1 spp, 5 bounces), at its full size against the
    oracle: primary rays, `Intersection` records and visit counters byte for byte (2 M walks through the 17 MB tree), the
    per-launch work counters of a frame equal to the oracle's within the path-flip band, two frames within the frame
    tolerance of the 5-bounce Disney scene."""
    fs, cam = atrium
    w, h = 1920, 1080
    c = create_camera(cam["pos"], cam["at"], cam["vfov"], w, h)
    ctx.updateCamera(c)
    ctx.initSampler(w, h, 0)
    ctx.setScreenShard(0, 1)
    seeds = orc.init_sampler(w, h, 0)
    rays = orc.generate_paths(c, seeds, w, h, 0, 0)
    assert ctx.generate_paths(w, h, 0, 0).tobytes() == rays.tobytes()
    want_i, wst = orc.trace_closest(fs, rays)
    got_i, gst = ctx.trace_closest(rays, stats=True)
    assert got_i.tobytes() == want_i.tobytes()
    assert np.array_equal(gst, wst)
    for frame in (0, 5):
        ctx.reset()
        got = ctx.render(w, h, 5, 3, frame=frame)
        want, cnt = orc.render(fs, c, seeds, w, h, 5, 3, frame=frame, counters=True)
        frac, mean_err = frame_tolerance_report(got, want)
        assert frac >= 0.975, (frame, frac)          # measured 0.9812 / 0.9813 (profiles/parity_r05.json)
        assert mean_err <= 5e-3, (frame, mean_err)
        ctx.reset()
        ctx.render(w, h, 5, 3, frame=frame, count_stats=True, download=False)
        m = parity_record("companion: atrium 1920x1080 1spp 5-bounce Disney + textures + IBL, frame %d" % frame, got, want,
                          gpu_stats=ctx.stats(), oracle_counters=cnt)
        # (the Disney lobes produce NaN samples on both sides -- 0.3 % of this frame's pixels; a path that diverged may be one on
        #  one side only)
        assert abs(m["nonfinite_pixels_got"] - m["nonfinite_pixels_want"]) <= 64

