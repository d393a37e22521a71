"""CPU tests of the SVGF restatement (oracle/orc_svgf.h): self-consistency and the committed vectors."""
import importlib.util
import os

import numpy as np

from conftest import GOLDEN, make_camera


def _golden_module():
    spec = importlib.util.spec_from_file_location("make_golden_svgf", os.path.join(GOLDEN, "make_golden_svgf.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_aov_packing_known_answers_from_reference_unittest(orc):
    """src/aten_unittest/aov_host_buffer.cpp:73-108 `FillBasicAOVsTest` and :110-132 `FillBasicAOVsIfHitMissTest`, restated:
    normal (1,2,3), rec.p = vec4(1), identity mtx_W2C, albedo (4,5,6,7), isect.meshid = 2
      -> normal_depth == (normal.xyz, rec.p.z), albedo_meshid == (albedo.xyz, meshid)      (ASSERT_EQ: exact)
    bg (4,5,6,7) -> normal_depth == (0,0,0,-1), albedo_meshid == (bg.xyz, -1).
    The reference's own known answer for the SVGF AOV packing; the GPU planes are held to the same numbers
    in tests/test_gpu_invariants.py::test_svgf_aov_planes_known_answers."""
    normal, p, albedo = (1.0, 2.0, 3.0), (1.0, 1.0, 1.0), (4.0, 5.0, 6.0, 7.0)
    nd, am = orc.fill_basic_aovs(normal, p, np.eye(4, dtype=np.float32), albedo, 2)
    assert nd.tolist() == [1.0, 2.0, 3.0, 1.0]          # .w == rec.p.z (== clip w of (1,1,1,1) under identity)
    assert am.tolist() == [4.0, 5.0, 6.0, 2.0]
    # beyond the unit test's identity: depth is the CLIP-space w, i.e. row 3 of W2C applied to (p, 1) (aov.h:166-168)
    m = np.arange(16, dtype=np.float32).reshape(4, 4)
    nd, _ = orc.fill_basic_aovs(normal, (2.0, 3.0, 5.0), m, albedo, 2)
    assert nd[3] == np.float32(12 * 2 + 13 * 3 + 14 * 5 + 15)
    nd, am = orc.fill_basic_aovs_if_hit_miss((4.0, 5.0, 6.0, 7.0))
    assert nd.tolist() == [0.0, 0.0, 0.0, -1.0]
    assert am.tolist() == [4.0, 5.0, 6.0, -1.0]


def test_svgf_oracle_matches_committed_vectors():
    want = np.load(os.path.join(GOLDEN, "svgf_golden.npz"))
    got = _golden_module().run()
    assert set(got) == set(want.files)
    for k in want.files:
        assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), k


def test_svgf_oracle_behaviour(orc, cornell):
    fs, cam = cornell
    w, h = 64, 40
    c = make_camera(orc, cam, w, h)
    seeds = orc.init_sampler(w, h, 0)
    sv = orc.Svgf()
    try:
        films = []
        for frame in range(5):
            film, st = sv.render(fs, c, seeds, w, h, 3, 3, frame=frame, compute_motion=True, stages=True)
            films.append(film)
            md = sv.buffer("motion_depth")
            hit = sv.buffer("primary_position")[..., 3] == 1.0
            if frame > 0:
                assert np.all(md[..., :2] == 0.0)              # static camera: prev and cur NDC are the same floats
            # AOV depth = clip w = distance along the view direction (float64 check)
            p = sv.buffer("primary_position")[..., :3].astype(np.float64)
            d = np.asarray(cam["at"], np.float64) - np.asarray(cam["pos"], np.float64)
            d /= np.linalg.norm(d)
            depth = (p - np.asarray(cam["pos"], np.float64)) @ d
            nd = sv.buffer("prev_normal_depth")
            # (pixels that see the Specular box carry the AOV of the reflected hit, svgf.cpp:142-157)
            close = np.isclose(nd[..., 3][hit], depth[hit], rtol=1e-5, atol=1e-5)
            assert close.mean() > 0.9
            assert np.all(nd[..., 3][~hit] == -1.0)
            mt = sv.buffer("prev_moment_temporalweight")
            assert mt[..., 2].max() == frame + 1                # accumulated frame count of undisturbed pixels
            assert mt[..., 2].min() >= 1
        # the filtered frame is smoother than the path-traced one: mean absolute Laplacian drops
        def rough(a):
            a = np.nan_to_num(a[..., :3])
            return np.abs(4 * a[1:-1, 1:-1] - a[:-2, 1:-1] - a[2:, 1:-1] - a[1:-1, :-2] - a[1:-1, 2:]).mean()
        albedo = sv.buffer("prev_albedo_meshid")
        assert rough(films[-1]) < 0.7 * rough(st[0] * np.concatenate([albedo[..., :3], np.ones_like(albedo[..., :1])], -1))
    finally:
        sv.close()


def test_svgf_oracle_thread_count_independent(orc, cornell):
    fs, cam = cornell
    w, h = 48, 32
    c = make_camera(orc, cam, w, h)
    seeds = orc.init_sampler(w, h, 0)
    outs = []
    for nt in (1, 4):
        sv = orc.Svgf()
        frames = [sv.render(fs, c, seeds, w, h, 3, 3, frame=f, compute_motion=True, nthreads=nt) for f in range(3)]
        sv.close()
        outs.append(np.stack(frames))
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
