"""GPU parity of the SVGF passes (next tier, BASELINE config 5) against oracle/orc_svgf.h through the C-ABI.

Tolerances: the temporal pass has no transcendental and must agree to the last bit GIVEN EQUAL INPUTS; its inputs
(path-traced colours, AOV depth) carry the path tracer's own tolerance, so the end-to-end comparison uses the frame
tolerance of DESIGN.md (|gpu - cpu| <= 1e-3 * max(1, |cpu|) for >= 99.5 % of pixels); integer-valued planes
(material id, frame count) are compared exactly where the path agrees.
"""
import numpy as np
import pytest

from conftest import make_camera, parity_record

pytestmark = pytest.mark.gpu


def frac_within(a, b, rel=1e-3):
    d = np.abs(a - b)
    tol = rel * np.maximum(1.0, np.abs(b))
    ok = np.all((d <= tol) | (np.isnan(a) & np.isnan(b)), axis=-1)
    return ok.mean()


def _setup(gpu, orc, scene, cam, w, h):
    c = make_camera(orc, cam, w, h)
    gpu.UpdateSceneData(scene)
    gpu.updateCamera(c)
    gpu.initSampler(w, h, 0)
    gpu.setScreenShard(0, 1)
    gpu.svgf_reset()
    return c, orc.init_sampler(w, h, 0)


@pytest.mark.parametrize("which", ["sponza", "cornell"])
def test_svgf_filter_passes_on_equal_inputs(gpu, orc, sponza, cornell, which):
    """The filter passes proper: both sides get the SAME noisy frame and G-buffer (the oracle's path pass, uploaded
    through atn_svgf_upload), frame after frame, each side keeping its own history.  What differs is then only
    powf/expf/sqrt rounding inside the passes."""
    fs, cam = sponza if which == "sponza" else cornell
    w, h = 192, 108
    c, seeds = _setup(gpu, orc, fs, cam, w, h)
    sv = orc.Svgf()
    try:
        for frame in range(5):
            want, wst = sv.render(fs, c, seeds, w, h, 5, 3, frame=frame, compute_motion=True, stages=True)
            contribs = wst[0].copy()
            contribs[..., 3] = 1.0          # Path.contrib.samples with 1 spp
            gpu.svgf_upload("contribs", contribs)
            gpu.svgf_upload("normal_depth", sv.buffer("prev_normal_depth"))     # the set the oracle just wrote
            gpu.svgf_upload("albedo_meshid", sv.buffer("prev_albedo_meshid"))
            gpu.svgf_upload("primary_position", sv.buffer("primary_position"))
            got, gst = gpu.svgf_denoise(w, h, frame=frame, compute_motion=True, stages=True)
            assert frac_within(gpu.svgf_buffer("motion_depth"), sv.buffer("motion_depth"), 1e-5) == 1.0
            assert np.array_equal(gst[0].view(np.uint32), wst[0].view(np.uint32))
            # the variance is E[l^2] - E[l]^2 of radiances up to ~200 (the Cornell light): its cancellation noise alone
            # is ~1e-3 absolute on a few pixels per thousand
            for s_, name, need in ((1, "temporal", 0.9995), (2, "variance", 0.998)):
                f = frac_within(gst[s_][..., :3], wst[s_][..., :3])
                assert f >= need, (frame, name, f)
            for name, need in (("prev_color_variance", 0.998), ("prev_moment_temporalweight", 0.9995), ("temporary_color", 0.999)):
                f = frac_within(gpu.svgf_buffer(name), sv.buffer(name))
                assert f >= need, (frame, name, f)
            f = frac_within(got, want)
            assert f >= 0.999, (frame, "filtered", f)
            a, b = gpu.svgf_buffer("prev_moment_temporalweight"), sv.buffer("prev_moment_temporalweight")
            assert (a[..., 2] == b[..., 2]).mean() >= 0.9995, frame        # accumulated frame counts
        assert np.nanmax(got[..., :3]) > 0
    finally:
        sv.close()


def test_svgf_end_to_end_vs_oracle(gpu, orc, sponza):
    """Path pass + filter passes on the GPU against the oracle.  The path pass agrees for >= 99.5 % of the pixels
    (DESIGN.md tolerance: sinf/cosf ulps occasionally send a path elsewhere); the filters then spread every such
    pixel over their footprint (7x7, then 5 a-trous levels up to +-32 px), so the filtered frames are compared by
    their statistics, and the AOVs / motion vectors, which are first-hit geometry, pixel by pixel."""
    fs, cam = sponza
    w, h = 192, 108
    c, seeds = _setup(gpu, orc, fs, cam, w, h)
    sv = orc.Svgf()
    try:
        for frame in range(3):
            want, wst = sv.render(fs, c, seeds, w, h, 5, 3, frame=frame, compute_motion=True, stages=True)
            got, gst = gpu.svgf_render(w, h, 5, 3, frame=frame, compute_motion=True, stages=True)
            for name in ("prev_normal_depth", "prev_albedo_meshid"):
                a, b = gpu.svgf_buffer(name), sv.buffer(name)
                assert frac_within(a, b, 1e-4) >= 0.9995, (frame, name)
            assert frac_within(gpu.svgf_buffer("primary_position"), sv.buffer("primary_position"), 1e-5) >= 0.9995
            assert frac_within(gpu.svgf_buffer("motion_depth"), sv.buffer("motion_depth"), 1e-4) >= 0.9995
            assert frac_within(gst[0][..., :3], wst[0][..., :3]) >= 0.995, frame
            parity_record("C5 at oracle size: sponza_lod 192x108 SVGF, frame %d, path pass (stage 0)" % frame, gst[0], wst[0])
            parity_record("C5 at oracle size: sponza_lod 192x108 SVGF, frame %d, filtered output" % frame, got, want, tol=5e-2)
            ma, mb = np.nanmean(got[..., :3]), np.nanmean(want[..., :3])
            assert abs(ma - mb) <= 3e-2 * max(abs(mb), 1e-6), (frame, ma, mb)
            assert frac_within(got[..., :3], want[..., :3], rel=5e-2) >= 0.9, frame
    finally:
        sv.close()


def test_svgf_temporal_pass_bit_exact(gpu, orc, cornell):
    """TemporalReprojection + AccumulateMoments contain no transcendental: with the same inputs AND the same history
    (the oracle's previous AOV set uploaded as the GPU's) every pixel must agree to the last bit."""
    fs, cam = cornell
    w, h = 96, 64
    c, seeds = _setup(gpu, orc, fs, cam, w, h)
    sv = orc.Svgf()
    try:
        def feed(wst):
            contribs = wst[0].copy()
            contribs[..., 3] = 1.0
            gpu.svgf_upload("contribs", contribs)
            gpu.svgf_upload("normal_depth", sv.buffer("prev_normal_depth"))
            gpu.svgf_upload("albedo_meshid", sv.buffer("prev_albedo_meshid"))
            gpu.svgf_upload("primary_position", sv.buffer("primary_position"))
        _, wst = sv.render(fs, c, seeds, w, h, 3, 3, frame=0, compute_motion=True, stages=True)
        feed(wst)
        gpu.svgf_denoise(w, h, frame=0, compute_motion=True)
        for frame in (1, 2):
            # make the GPU's history the oracle's
            for name in ("prev_normal_depth", "prev_albedo_meshid", "prev_color_variance", "prev_moment_temporalweight"):
                gpu.svgf_upload(name, sv.buffer(name))
            _, wst = sv.render(fs, c, seeds, w, h, 3, 3, frame=frame, compute_motion=True, stages=True)
            feed(wst)
            _, gst = gpu.svgf_denoise(w, h, frame=frame, compute_motion=True, stages=True)
            assert np.array_equal(gst[1].view(np.uint32), wst[1].view(np.uint32)), frame
            a, b = gpu.svgf_buffer("prev_moment_temporalweight"), sv.buffer("prev_moment_temporalweight")
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), frame
    finally:
        sv.close()


def test_svgf_optional_temporal_weight_dilation_bit_exact(gpu, orc, cornell):
    """RecomputeTemporalWeightFromSurroundingPixels (svgf_impl.h:386-423; run by the CUDA twin only, svgf_tp.cu:150-216)
    behind atn_svgf_set_dilate_temporal_weight: integer compares and a 3x3 minimum, so with equal inputs and equal history
    the moment / temporal-weight plane must agree with the oracle's twin to the last bit -- and differ from the plane
    without the pass."""
    fs, cam = cornell
    w, h = 96, 64
    c, seeds = _setup(gpu, orc, fs, cam, w, h)
    sv = orc.Svgf()
    planes = {}
    try:
        for dilate in (False, True):
            gpu.svgf_reset()
            gpu.svgf_set_dilate_temporal_weight(dilate)
            sv.close(); sv = orc.Svgf()
            sv.set_dilate_temporal_weight(dilate)

            def feed(wst):
                contribs = wst[0].copy()
                contribs[..., 3] = 1.0
                gpu.svgf_upload("contribs", contribs)
                gpu.svgf_upload("normal_depth", sv.buffer("prev_normal_depth"))
                gpu.svgf_upload("albedo_meshid", sv.buffer("prev_albedo_meshid"))
                gpu.svgf_upload("primary_position", sv.buffer("primary_position"))
            _, wst = sv.render(fs, c, seeds, w, h, 3, 3, frame=0, compute_motion=True, stages=True)
            feed(wst)
            gpu.svgf_denoise(w, h, frame=0, compute_motion=True)
            for frame in (1, 2, 3):
                for name in ("prev_normal_depth", "prev_albedo_meshid", "prev_color_variance", "prev_moment_temporalweight"):
                    gpu.svgf_upload(name, sv.buffer(name))
                _, wst = sv.render(fs, c, seeds, w, h, 3, 3, frame=frame, compute_motion=True, stages=True)
                feed(wst)
                gpu.svgf_denoise(w, h, frame=frame, compute_motion=True)
                a, b = gpu.svgf_buffer("prev_moment_temporalweight"), sv.buffer("prev_moment_temporalweight")
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (dilate, frame)
            planes[dilate] = a.copy()
        assert not np.array_equal(planes[False][..., 3], planes[True][..., 3])
    finally:
        gpu.svgf_set_dilate_temporal_weight(False)
        sv.close()


def test_svgf_external_motion_buffer_and_camera_move(gpu, orc, sponza):
    """SetMotionDepthBuffer path (the reference's own interface) and a moving camera through the compute pass."""
    fs, cam = sponza
    w, h = 160, 90
    c, seeds = _setup(gpu, orc, fs, cam, w, h)
    sv = orc.Svgf()
    try:
        md = np.zeros((h, w, 4), np.float32)
        md[..., 0] = 1.5 / w      # a uniform 1.5-pixel shift: exercises the int cast and the clamp
        md[..., 3] = 1.0
        sv.set_motion_depth(md)
        gpu.svgf_set_motion_depth(md)
        for frame in range(3):
            want = sv.render(fs, c, seeds, w, h, 5, 3, frame=frame)
            got = gpu.svgf_render(w, h, 5, 3, frame=frame)
            assert frac_within(got[..., :3], want[..., :3], rel=5e-2) >= 0.9, frame
        # camera move, motion from the compute pass
        cam2 = dict(cam)
        cam2["pos"] = (cam["pos"][0] + 0.05, cam["pos"][1], cam["pos"][2])
        c2 = make_camera(orc, cam2, w, h)
        gpu.updateCamera(c2)
        want = sv.render(fs, c2, seeds, w, h, 5, 3, frame=3, compute_motion=True)
        got = gpu.svgf_render(w, h, 5, 3, frame=3, compute_motion=True)
        a, b = gpu.svgf_buffer("motion_depth"), sv.buffer("motion_depth")
        assert np.abs(b[..., :2]).max() > 0
        assert frac_within(a, b, 1e-3) >= 0.999
        assert frac_within(got[..., :3], want[..., :3], rel=5e-2) >= 0.9
    finally:
        sv.close()


def test_svgf_needs_motion_buffer_or_compute_pass(gpu, orc, cornell):
    from aten_amd.renderer import AtenAmdError
    fs, cam = cornell
    _setup(gpu, orc, fs, cam, 64, 64)
    gpu.svgf_reset()
    gpu2_err = None
    try:
        # fresh history and no motion buffer of this size
        gpu.svgf_render(72, 40, 3, 3, frame=0)
    except AtenAmdError as e:
        gpu2_err = str(e)
    # a motion buffer set by an earlier test may satisfy the size; either outcome must be explicit
    assert gpu2_err is None or "motion" in gpu2_err


@pytest.mark.parametrize("iters", [1, 3])
def test_svgf_atrous_iteration_count(gpu, orc, cornell, iters):
    """SVGFParams::atrous_iter_cnt other than 5: first == final iteration (1) and an odd ping-pong length (3)."""
    fs, cam = cornell
    w, h = 96, 64
    c, seeds = _setup(gpu, orc, fs, cam, w, h)
    sv = orc.Svgf()
    sv.set_atrous_iterations(iters)
    gpu.svgf_set_atrous_iterations(iters)
    try:
        for frame in range(3):
            want, wst = sv.render(fs, c, seeds, w, h, 3, 3, frame=frame, compute_motion=True, stages=True)
            contribs = wst[0].copy()
            contribs[..., 3] = 1.0
            gpu.svgf_upload("contribs", contribs)
            gpu.svgf_upload("normal_depth", sv.buffer("prev_normal_depth"))
            gpu.svgf_upload("albedo_meshid", sv.buffer("prev_albedo_meshid"))
            gpu.svgf_upload("primary_position", sv.buffer("primary_position"))
            got = gpu.svgf_denoise(w, h, frame=frame, compute_motion=True)
            assert frac_within(got, want) >= 0.998, (iters, frame)
    finally:
        gpu.svgf_set_atrous_iterations(5)
        sv.close()


def test_svgf_full_size_properties_1080p(gpu, orc, sponza):
    """Config-5 size, no oracle: a static camera gives zero motion, so every pixel that stays on its surface
    accumulates one frame per render (moments.z == frames rendered) and the temporal weight saturates; the output is
    finite wherever the path-traced input is, and it is smoother than its input."""
    fs, cam = sponza
    w, h = 1920, 1080
    _setup(gpu, orc, fs, cam, w, h)
    n = 4
    for frame in range(n):
        out, st = gpu.svgf_render(w, h, 5, 3, frame=frame, compute_motion=True, stages=True)
    md = gpu.svgf_buffer("motion_depth")
    assert np.all(md[..., :2] == 0.0)
    mt = gpu.svgf_buffer("prev_moment_temporalweight")
    ids = gpu.svgf_buffer("prev_albedo_meshid")[..., 3]
    surf = ids >= 0
    assert mt[..., 2].max() == n
    assert (mt[..., 2][surf] == n).mean() > 0.9            # the rest was disoccluded by sub-pixel jitter at silhouettes
    assert np.all(mt[..., 2][~surf] == 1.0)
    assert (mt[..., 3][surf] > 0.5).mean() > 0.9           # temporal weight ~1 where reprojection succeeded
    raw = st[0][..., :3] * gpu.svgf_buffer("prev_albedo_meshid")[..., :3]
    ok = np.isfinite(raw).all(-1)
    assert np.isfinite(out[..., :3][ok]).mean() > 0.999

    def rough(a):
        a = np.nan_to_num(a)
        return np.abs(4 * a[1:-1, 1:-1] - a[:-2, 1:-1] - a[2:, 1:-1] - a[1:-1, :-2] - a[1:-1, 2:]).mean()
    assert rough(out[..., :3]) < 0.5 * rough(raw)


def test_svgf_frames_in_flight_equal_serial(sponza):
    """Pipelined SVGF frames (atn_set_frames_in_flight > 1: path pass of frame f + 1 on the next bank while the filters of
    frame f run on the filter stream, ordered by the temporal pass) give the same filtered frames and the same history."""
    from aten_amd.renderer import PathTracing
    from aten_amd.scene.camera import create_camera
    fs, cam = sponza
    w, h = 320, 180

    def run(in_flight):
        r = PathTracing(0)
        try:
            r.UpdateSceneData(fs)
            r.updateCamera(create_camera(cam["pos"], cam["at"], cam["vfov"], w, h))
            r.initSampler(w, h, 0)
            r.set_frames_in_flight(in_flight)
            outs = []
            for f in range(6):
                img = r.svgf_render(w, h, 4, 3, frame=f, compute_motion=True, download=(f in (2, 5)))
                if img is not None:
                    outs.append(img.copy())
            hist = r.svgf_buffer("prev_moment_temporalweight").copy()
            return outs, hist
        finally:
            r.close()
    want, want_hist = run(1)
    for n in (2, 3):
        got, hist = run(n)
        for a, b in zip(got, want):
            assert a.tobytes() == b.tobytes(), n
        assert hist.tobytes() == want_hist.tobytes(), n


def test_film_order_survives_svgf_frames_between_renders(sponza):
    """Advisor finding of round 2: an SVGF frame between two render() calls rotates the banks without writing the film, so
    "the previous bank's last event" was not the film's last writer and two k_gather passes (running mean, read-modify-write)
    could overlap or swap.  The film's writers are now ordered by their own event.  Deep frames followed by depth-1 frames
    (which reach their gather first), SVGF frames never downloaded in between, 3 and 4 frames in flight == serial."""
    from aten_amd.renderer import PathTracing
    from aten_amd.scene.camera import create_camera
    fs, cam = sponza
    w, h = 320, 180

    def run(in_flight):
        r = PathTracing(0)
        try:
            r.UpdateSceneData(fs)
            r.updateCamera(create_camera(cam["pos"], cam["at"], cam["vfov"], w, h))
            r.initSampler(w, h, 0)
            r.set_frames_in_flight(in_flight)
            films = []
            f = 0
            for rep in range(6):
                r.render(w, h, 8, 3, frame=f, download=False); f += 1
                r.svgf_render(w, h, 4, 3, frame=rep, compute_motion=True, download=False)
                r.render(w, h, 1, 3, frame=f, download=False); f += 1
                r.render(w, h, 1, 3, frame=f, download=False); f += 1
                if rep == 2:
                    r.reset()               # the clear is a film writer too (ordered before the next frame's gather)
                    f = 0
                if rep in (1, 5):
                    films.append(r.download_film().copy())
            return films
        finally:
            r.close()
    want = run(1)
    assert (want[1][..., 3] == 9).all()
    for n in (3, 4):
        for a, b in zip(run(n), want):
            assert a.tobytes() == b.tobytes(), n


def test_atrous_four_pixel_kernel_against_the_one_pixel_kernel(sponza):
    """the default a-trous kernel (k_svgf_atrous4, four pixels per thread) sums a pixel's 24 weighted taps in LATTICE
    order, the one-pixel kernel (k_svgf_atrous, ATEN_AMD_SVGF_ATROUS4=0) in the reference's ring order (svgf_impl.h:693-726):
    same taps, same weights, another association of the float sums.  Both kernels on IDENTICAL planes, frame after frame (each
    context keeps its own history): the filtered colour differs by float rounding only -- the bound is explicit here instead of
    hiding inside the end-to-end tolerance."""
    import os
    from aten_amd.renderer import PathTracing
    from aten_amd.scene.camera import create_camera
    fs, cam = sponza
    w, h = 192, 108
    ctx = {}
    old = os.environ.get("ATEN_AMD_SVGF_ATROUS4")
    try:
        for flag in ("0", "1"):
            os.environ["ATEN_AMD_SVGF_ATROUS4"] = flag      # read when the context is created
            r = PathTracing(0)
            r.UpdateSceneData(fs)
            r.updateCamera(create_camera(cam["pos"], cam["at"], cam["vfov"], w, h))
            r.initSampler(w, h, 0)
            r.setScreenShard(0, 1)
            r.svgf_reset()
            ctx[flag] = r
        # a third context renders the noisy frames and G-buffers; the two filter contexts get identical uploads every frame and
        # keep their own history
        os.environ["ATEN_AMD_SVGF_ATROUS4"] = "1"
        src = PathTracing(0)
        ctx["src"] = src
        src.UpdateSceneData(fs)
        src.updateCamera(create_camera(cam["pos"], cam["at"], cam["vfov"], w, h))
        src.initSampler(w, h, 0)
        src.setScreenShard(0, 1)
        src.svgf_reset()
        worst = 0.0
        for frame in range(4):
            src.svgf_render(w, h, 5, 3, frame=frame, compute_motion=True)
            planes = {n: src.svgf_buffer(n) for n in ("contribs", "prev_normal_depth", "prev_albedo_meshid", "primary_position")}
            outs = {}
            for flag in ("0", "1"):
                r = ctx[flag]
                r.svgf_upload("contribs", planes["contribs"])
                r.svgf_upload("normal_depth", planes["prev_normal_depth"])
                r.svgf_upload("albedo_meshid", planes["prev_albedo_meshid"])
                r.svgf_upload("primary_position", planes["primary_position"])
                outs[flag] = r.svgf_denoise(w, h, frame=frame, compute_motion=True)
            a, b = outs["0"][..., :3].astype(np.float64), outs["1"][..., :3].astype(np.float64)
            assert np.all(np.isfinite(a)) and np.all(np.isfinite(b)) and a.max() > 0
            rel = np.abs(a - b) / np.maximum(1.0, np.abs(a))
            worst = max(worst, float(rel.max()))
            # rounding of 24-term sums of O(1) weights: a few 1e-7 per level, five levels, fed back through the history
            assert rel.max() <= 2e-5, (frame, rel.max())
            assert (rel <= 2e-6).mean() >= 0.999, (frame, (rel <= 2e-6).mean())
        print("atrous4 vs atrous: worst relative difference %.3g" % worst)
    finally:
        for r in ctx.values():
            r.close()
        if old is None:
            os.environ.pop("ATEN_AMD_SVGF_ATROUS4", None)
        else:
            os.environ["ATEN_AMD_SVGF_ATROUS4"] = old
