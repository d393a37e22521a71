"""NPR materials of the path tracer (SURVEY 8(f)4 remainder): aten::Toon / aten::StylizedBrdf (material/toon.cpp), "terminated"
materials that the integrator treats as a light at the first hit (HitTeminatedMaterial, pathtracing_impl.h:482-503) -- one
NEE sample towards a designated NPR target light with an inline visibility test (HitTestToTargetLight), remapped through
a 1-D texture, times a screen-space shadow texture, plus a rim light -- and as their base material deeper in the path
(pathtracing.cpp:160-184).  HIP path (k_shade<., 3>, device/toon.hpp) against the oracle's restatement."""
import numpy as np
import pytest

from aten_amd import layout as L
from aten_amd.renderer import AtenAmdError, PathTracing
from aten_amd.scene import scenedefs
from aten_amd.scene.camera import create_camera
from test_gpu_parity import frame_tolerance_report

pytestmark = pytest.mark.gpu

W = H = 112


def render_pair(orc, scene, frames=(0, 3), depth=5):
    fs, cam = scene
    c = create_camera(cam["pos"], cam["at"], cam["vfov"], W, H)
    seeds = orc.init_sampler(W, H, 0)
    r = PathTracing(0)
    out = []
    try:
        r.UpdateSceneData(fs); r.updateCamera(c); r.initSampler(W, H, 0)
        for f in frames:
            r.reset()
            out.append((r.render(W, H, depth, 3, frame=f), orc.render(fs, c, seeds, W, H, depth, 3, frame=f)))
        rays = orc.generate_paths(c, seeds, W, H, 0, frames[-1])
        isect, _ = orc.trace_closest(fs, rays)
    finally:
        r.close()
    mt = fs.arrays["materials"]["type"][isect["mtrlid"].clip(0)]
    hit = isect["objid"] >= 0
    return out, (hit & (mt == L.MTRL_TOON)).reshape(H, W), (hit & (mt == L.MTRL_STYLIZED)).reshape(H, W)


@pytest.mark.parametrize("target", ["point", "area"])
def test_toon_room_frames(orc, target):
    out, toon_px, sty_px = render_pair(orc, scenedefs.toon_room(target=target))
    assert toon_px.sum() > 1500 and sty_px.sum() > 300
    for got, want in out:
        frac, mean_err = frame_tolerance_report(got, want)
        assert frac >= 0.995 and mean_err <= 5e-3, (frac, mean_err)
    got, want = out[-1]
    # a primary toon hit ends the path with throughput (1) x remap(band) x albedo: the tall box (albedo 0.9 in red) shows
    # exactly the ramp's four bands, on both sides, and the band is the same one except where the luminance sits on an edge
    tall = toon_px & (np.abs(want[..., 1] / np.maximum(want[..., 0], 1e-9) - 0.5 / 0.9) < 1e-3)
    assert tall.sum() > 300
    bands = np.float32([0.15, 0.45, 0.8, 1.0]) * np.float32(0.9)
    for img in (got, want):
        d = np.abs(img[tall][:, 0:1] - bands[None, :]).min(axis=1)
        assert (d < 1e-6).all()
    assert (np.abs(got[tall][:, 0] - want[tall][:, 0]) < 1e-6).mean() > 0.99
    assert len(np.unique(np.round(want[tall][:, 0], 5))) >= 2              # more than one band is visible
    # the stylized box: weight x remap x pdf with a rim light on top, a float path: within the frame tolerance
    assert np.allclose(got[sty_px], want[sty_px], rtol=2e-3, atol=2e-3)


def test_toon_screen_space_shadow_and_lookups(orc):
    """The screen-space shadow texture darkens the bands below the threshold (Toon::bsdf, toon.cpp:148-160); with alpha
    blending on, a half-transparent pane between the surfaces and the target light is looked through by the inline
    visibility test (up to 10 lookups) instead of blocking it."""
    sh = np.full((H, W), 0.25, np.float32)
    sh[:, : W // 2] = 0.6
    plain, toon_px, _ = render_pair(orc, scenedefs.toon_room(target="point"), frames=(1,))
    shaded, toon_px2, _ = render_pair(orc, scenedefs.toon_room(target="point", screen_shadow=sh), frames=(1,))
    for got, want in shaded:
        frac, mean_err = frame_tolerance_report(got, want)
        assert frac >= 0.995 and mean_err <= 5e-3, (frac, mean_err)
    assert np.array_equal(toon_px, toon_px2)
    assert shaded[0][0][toon_px].sum() < 0.97 * plain[0][0][toon_px].sum()     # it did darken something
    pane, _, _ = render_pair(orc, scenedefs.toon_room(target="point", alpha_blocker=True), frames=(1,))
    for got, want in pane:
        frac, mean_err = frame_tolerance_report(got, want)
        assert frac >= 0.995 and mean_err <= 5e-3, (frac, mean_err)


def test_toon_shadow_really_tests_visibility(orc):
    """will_receive_shadow: a toon surface behind a blocker falls into the ramp's darkest band (radiance 0 -> remap(0))."""
    fs, cam = scenedefs.toon_room(target="point")
    mats = fs.arrays["materials"]
    tall = [i for i, n in enumerate(fs.names["materials"]) if n == "tallBox"][0]
    with_shadow, toon_px, _ = render_pair(orc, (fs, cam), frames=(0,))
    mats["toon"]["will_receive_shadow"][tall] = 0
    without, _, _ = render_pair(orc, (fs, cam), frames=(0,))
    g1, w1 = with_shadow[0]; g0, w0 = without[0]
    assert frame_tolerance_report(g0, w0)[0] >= 0.995
    lit_more = (w0[..., 0] > w1[..., 0] + 1e-3) & toon_px
    assert lit_more.sum() > 10                          # pixels the short box (or the box itself) shadows
    assert (g0[lit_more][:, 0] > g1[lit_more][:, 0]).mean() > 0.9


def test_toon_upload_errors(orc):
    fs, cam = scenedefs.toon_room(target="point")
    tall = [i for i, n in enumerate(fs.names["materials"]) if n == "tallBox"][0]
    fs.arrays["materials"]["toon"]["target_light_idx"][tall] = 3
    r = PathTracing(0)
    try:
        with pytest.raises(AtenAmdError, match="target light index"):
            r.UpdateSceneData(fs)
    finally:
        r.close()
