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


# (scene, bounces, bound on the rate of out-of-tolerance pixels per frame).  The Disney atrium -- textures, normal maps, IBL and a lamp,
# light of comparable size arriving at every bounce -- shows a last-bit difference in the pixel where sponza_lod's sum absorbs it
# (tools/ulp_study.py, profiles/r06_ulp_study.json: the per-vertex rate of last-bit differences is the same ~30 % for both): 1 % of
# its pixels leave the band at 5 bounces, 3 % at 8.  Followed here like the others.
CASES = {"cornell": ("cornell", 5, 5e-3), "sponza": ("sponza", 5, 5e-3), "atrium-5": ("atrium", 5, 2.5e-2), "atrium-8": ("atrium", 8, 6e-2)}


@pytest.mark.parametrize("which", list(CASES))
def test_out_of_tolerance_pixels_converge_to_the_oracle(gpu, orc, cornell, sponza, which):
    scene, depth, rate_bound = CASES[which]
    if scene == "atrium":
        from aten_amd.scene import scenedefs
        fs, cam = scenedefs.atrium(detail=0.25)
    else:
        fs, cam = cornell if scene == "cornell" else sponza
    w, h = (256, 256) if scene == "cornell" else (256, 144)
    c = make_camera(orc, cam, w, h)
    gpu.UpdateSceneData(fs)
    gpu.updateCamera(c)
    gpu.initSampler(w, h, 0)
    gpu.setScreenShard(0, 1)
    seeds = orc.init_sampler(w, h, 0)

    def g(f):
        gpu.reset()
        return gpu.render(w, h, depth, 3, frame=f)

    G = _frames(g, N_FRAMES)
    O = _frames(lambda f: orc.render(fs, c, seeds, w, h, depth, 3, frame=f), N_FRAMES)
    # a sample the reference counts as invalid (NaN / negative: the Disney lobes produce some, on both sides) is no sample: the
    # frame's pixel is 0 / 0.  Such pixel-frames are left out of both sides' statistics.
    bad = ~(np.isfinite(G).all(-1) & np.isfinite(O).all(-1))          # [n, h, w]
    G = np.where(bad[..., None], 0.0, G); O = np.where(bad[..., None], 0.0, O)
    n_ok = np.maximum((~bad).sum(0), 1)                                # [h, w]
    per_frame = [parity_metrics(G[f], O[f]) for f in range(N_FRAMES)]
    outside0 = ~np.all(np.abs(G[0] - O[0]) <= 1e-3 * np.maximum(1.0, np.abs(O[0])), axis=-1)
    # the rate of diverged pixels does not grow with the frame index (nothing accumulates between frames)
    rates = np.array([1.0 - m["frac_within_0.001"] for m in per_frame])
    assert rates.max() <= rate_bound and rates[N_FRAMES // 2:].mean() <= 2.0 * max(rates[:N_FRAMES // 2].mean(), 1e-4)

    # pixels that were outside the tolerance in ANY of the frames: their 64-frame means agree within the Monte-Carlo error
    ever = np.zeros((h, w), bool)
    for f in range(N_FRAMES):
        ever |= ~np.all(np.abs(G[f] - O[f]) <= 1e-3 * np.maximum(1.0, np.abs(O[f])), axis=-1)
    assert ever.sum() >= 1, "no diverged pixel in 64 frames: nothing to follow (tighten the tolerance?)"
    lum = np.array([0.212639, 0.71517, 0.0721926])
    g_l, o_l = (G[:, ever] @ lum), (O[:, ever] @ lum)           # [n, k]
    nk = n_ok[ever].astype(np.float64)
    mg, mo = g_l.sum(0) / nk, o_l.sum(0) / nk
    # the two sample sets share most of their samples (same seeds): the difference of the means is carried by the frames that
    # differ, so its standard error is that of the per-frame DIFFERENCES
    diff = g_l - o_l
    se = diff.std(0, ddof=1) * np.sqrt(N_FRAMES) / nk
    z = (mg - mo) / np.maximum(se, 1e-9 + 1e-4 * np.maximum(mo, 1e-3))
    # (the literal bound first: within the Monte-Carlo error of two INDEPENDENT 64-sample means -- loose, most samples are shared)
    se_mc = np.sqrt((g_l.var(0, ddof=1) + o_l.var(0, ddof=1)) / N_FRAMES)
    assert np.all(np.abs(mg - mo) <= 4.0 * se_mc + 1e-3 * np.maximum(1.0, mo)), float(np.max(np.abs(mg - mo) / np.maximum(se_mc, 1e-12)))
    assert np.abs(z).max() <= 6.0, (np.abs(z).max(), int(ever.sum()))
    assert np.abs(z).mean() <= 1.6
    # and over the whole image the 64-frame means are closer than any single frame's
    mean_g, mean_o = G.sum(0) / n_ok[..., None], O.sum(0) / n_ok[..., None]
    m64 = parity_record("convergence: %s %dx%d 1spp %d-bounce, mean of %d frames" % (scene, w, h, depth, N_FRAMES), mean_g, mean_o,
                        pixels_outside_in_frame_0=int(outside0.sum()), pixels_outside_in_any_frame=int(ever.sum()),
                        diverged_pixel_rate_per_frame={"min": float(rates.min()), "mean": float(rates.mean()), "max": float(rates.max())},
                        z_of_followed_pixels={"max_abs": float(np.abs(z).max()), "mean_abs": float(np.abs(z).mean()),
                                              "note": "z = (mean_gpu - mean_oracle) / standard error of the per-frame differences, luminance"})
    assert m64["image_mean_relerr"] <= 1e-3
    # (the atrium's frames have fireflies -- one diverged path under the lamp carries a frame's whole error --, so its 64-frame mean is
    # held against the MEAN of the single frames' errors: errors of opposite sign cancel, they never add up to more)
    per = [m["image_mean_relerr"] for m in per_frame]
    typical = np.mean(per) if scene == "atrium" else np.median(per)
    assert m64["image_mean_relerr"] <= max(typical, 1e-6) * 1.5 + 1e-5
