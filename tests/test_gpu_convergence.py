"""A pixel whose path took another branch on the GPU than on the CPU (one ulp of difference in a sinf / cosf / logf decides a
comparison) is still a sample of the same unbiased estimator (pathtracing.cpp:269-366): its value differs, its EXPECTATION
does not.  This test follows the pixels that are outside the frame tolerance at frame 0 through 64 frames and holds their means
against each other within their own Monte-Carlo error -- and writes what it measured into the parity report."""
import numpy as np
import pytest

from conftest import make_camera, parity_record, parity_metrics

pytestmark = pytest.mark.gpu

N_FRAMES = 64


def _frames(render, n):
    out = []
    for f in range(n):
        out.append(render(f)[..., :3].astype(np.float64))
    return np.stack(out)                                   # [n, h, w, 3]


@pytest.mark.parametrize("which", ["cornell", "sponza"])
def test_out_of_tolerance_pixels_converge_to_the_oracle(gpu, orc, cornell, sponza, which):
    fs, cam = cornell if which == "cornell" else sponza
    w, h = (256, 256) if which == "cornell" else (256, 144)
    c = make_camera(orc, cam, w, h)
    gpu.UpdateSceneData(fs)
    gpu.updateCamera(c)
    gpu.initSampler(w, h, 0)
    gpu.setScreenShard(0, 1)
    seeds = orc.init_sampler(w, h, 0)

    def g(f):
        gpu.reset()
        return gpu.render(w, h, 5, 3, frame=f)

    G = _frames(g, N_FRAMES)
    O = _frames(lambda f: orc.render(fs, c, seeds, w, h, 5, 3, frame=f), N_FRAMES)
    per_frame = [parity_metrics(G[f], O[f]) for f in range(N_FRAMES)]
    outside0 = ~np.all(np.abs(G[0] - O[0]) <= 1e-3 * np.maximum(1.0, np.abs(O[0])), axis=-1)
    # the rate of diverged pixels does not grow with the frame index (nothing accumulates between frames)
    rates = np.array([1.0 - m["frac_within_0.001"] for m in per_frame])
    assert rates.max() <= 5e-3 and rates[N_FRAMES // 2:].mean() <= 2.0 * max(rates[:N_FRAMES // 2].mean(), 1e-4)

    # pixels that were outside the tolerance in ANY of the frames: their 64-frame means agree within the Monte-Carlo error
    ever = np.zeros((h, w), bool)
    for f in range(N_FRAMES):
        ever |= ~np.all(np.abs(G[f] - O[f]) <= 1e-3 * np.maximum(1.0, np.abs(O[f])), axis=-1)
    assert ever.sum() >= 1, "no diverged pixel in 64 frames: nothing to follow (tighten the tolerance?)"
    lum = np.array([0.212639, 0.71517, 0.0721926])
    g_l, o_l = (G[:, ever] @ lum), (O[:, ever] @ lum)           # [n, k]
    mg, mo = g_l.mean(0), o_l.mean(0)
    # the two sample sets share most of their samples (same seeds): the difference of the means is carried by the frames that
    # differ, so its standard error is that of the per-frame DIFFERENCES
    diff = g_l - o_l
    se = diff.std(0, ddof=1) / np.sqrt(N_FRAMES)
    z = (mg - mo) / np.maximum(se, 1e-9 + 1e-4 * np.maximum(mo, 1e-3))
    # (the literal bound first: within the Monte-Carlo error of two INDEPENDENT 64-sample means -- loose, most samples are shared)
    se_mc = np.sqrt((g_l.var(0, ddof=1) + o_l.var(0, ddof=1)) / N_FRAMES)
    assert np.all(np.abs(mg - mo) <= 4.0 * se_mc + 1e-3 * np.maximum(1.0, mo)), float(np.max(np.abs(mg - mo) / np.maximum(se_mc, 1e-12)))
    assert np.abs(z).max() <= 6.0, (np.abs(z).max(), int(ever.sum()))
    assert np.abs(z).mean() <= 1.6
    # and over the whole image the 64-frame means are closer than any single frame's
    mean_g, mean_o = G.mean(0), O.mean(0)
    m64 = parity_record("convergence: %s %dx%d 1spp 5-bounce, mean of %d frames" % (which, w, h, N_FRAMES), mean_g, mean_o,
                        pixels_outside_in_frame_0=int(outside0.sum()), pixels_outside_in_any_frame=int(ever.sum()),
                        diverged_pixel_rate_per_frame={"min": float(rates.min()), "mean": float(rates.mean()), "max": float(rates.max())},
                        z_of_followed_pixels={"max_abs": float(np.abs(z).max()), "mean_abs": float(np.abs(z).mean()),
                                              "note": "z = (mean_gpu - mean_oracle) / standard error of the per-frame differences, luminance"})
    assert m64["image_mean_relerr"] <= 1e-3
    assert m64["image_mean_relerr"] <= max(np.median([m["image_mean_relerr"] for m in per_frame]), 1e-6) * 1.5 + 1e-5
