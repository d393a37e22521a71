"""Path regeneration (atn_set_regeneration / atn_render_burst, include/aten_amd.h): the samples of a frame and the frames of a
progressive burst share one pool of path slots -- a path that ends is replaced, in the same launch, by the same pixel's next
primary ray.  Per pixel nothing but the launch a piece of work rides in changes, so every film must equal the serial loop's
(the reference's `for sample { for bounce { ... } }`, src/libidaten/kernel/pathtracing.cpp:105-138, with the CPU renderer's
per-pixel order: pathtracing.cpp:296-366, film.cpp:61-71) BYTE for byte: 1 and 8 spp, both sample-loop modes
(pathtracing.cpp:350-352), bursts of progressive frames, screen shards 1 / 2 / 8, every material set and trace flavour."""
import numpy as np
import pytest

from aten_amd.scene.camera import create_camera

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _torch_first():
    # torch's bundled HIP runtime has to come up before libaten_amd.so's (conftest.py, `gpu` fixture): the shard test reads
    # the tile buffer through a torch tensor
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()


def _ctx(fs, cam, w, h, shard=None):
    from aten_amd.renderer import PathTracing
    r = PathTracing(0)
    r.UpdateSceneData(fs)
    r.updateCamera(create_camera(cam["pos"], cam["at"], cam["vfov"], w, h))
    r.initSampler(w, h, 0)
    if shard:
        r.setScreenShard(*shard)
    return r


def _serial_frames(r, w, h, frames, depth, spp, brk, first=0, count=False):
    """The serial loop: one atn_render per frame; returns the films and the summed ray counters."""
    r.set_regeneration(False)
    out, rays, shadows = [], 0, 0
    for f in range(first, first + frames):
        out.append(r.render(w, h, depth, 3, spp=spp, frame=f, break_on_terminate=brk, count_stats=count).copy())
        if count:
            s = r.stats()
            rays += s["closest_rays"]; shadows += s["shadow_rays"]
    return out, rays, shadows


@pytest.mark.parametrize("brk", [True, False])
@pytest.mark.parametrize("spp", [1, 2, 8])
@pytest.mark.parametrize("scene", ["cornell", "sponza"])
def test_regenerated_samples_equal_the_serial_sample_loop(cornell, sponza, scene, spp, brk):
    fs, cam = cornell if scene == "cornell" else sponza
    w, h = (100, 52) if scene == "cornell" else (160, 90)       # (ragged: neither is a multiple of the 8 x 8 tile)
    r = _ctx(fs, cam, w, h)
    try:
        want, rays, shadows = _serial_frames(r, w, h, 2, 5, spp, brk, count=True)
        r.reset()
        r.set_regeneration(True)
        got = []
        for f in range(2):
            got.append(r.render_burst(w, h, 1, 5, 3, spp=spp, frame=f, break_on_terminate=brk).copy())
            q, sh = r.regen_stage_counts()
            assert len(q) == 2 * spp * 5 + 1            # (n_frames + 1) * spp * maxDepth stages + the last trace launch
            if f == 0:
                assert q[0] == w * h and sh[-1] == 0 and q[-1] == 0
        assert got[0].tobytes() == want[0].tobytes()
        assert got[1].tobytes() == want[1].tobytes()
        # atn_render itself takes the pool when a frame has more than one sample
        r.reset()
        a = r.render(w, h, 5, 3, spp=spp, frame=0, break_on_terminate=brk)
        assert a.tobytes() == want[0].tobytes()
    finally:
        r.close()


@pytest.mark.parametrize("spp,brk", [(1, True), (3, True), (3, False)])
def test_a_burst_of_progressive_frames_equals_frame_after_frame(sponza, spp, brk):
    """K frames in one pool: per pixel frame f's FilmProgressive::put precedes frame f + 1's, the running mean is the serial one.
    Also: a burst that continues a film other frames have written, ray totals equal to the serial loop's counters, and the
    pool's occupancy -- every stage but the tail of the burst carries (almost) a full population."""
    fs, cam = sponza
    w, h, K = 192, 108, 6
    r = _ctx(fs, cam, w, h)
    try:
        want, rays, shadows = _serial_frames(r, w, h, K, 5, spp, brk, count=True)
        r.reset()
        r.set_regeneration(True)
        got = r.render_burst(w, h, K, 5, 3, spp=spp, frame=0, break_on_terminate=brk)
        assert got.tobytes() == want[-1].tobytes()
        q, sh = r.regen_stage_counts()
        assert int(q.sum()) == rays and int(sh.sum()) == shadows
        # the pool is FULL until the burst's last item has been handed out: a slot whose item is finished takes the next one in the same
        # launch, whichever pixel and frame it is (the slot = pixel form of r06's first version ran dry from stage K on)
        full = int((q == w * h).sum())
        assert (q[:full] == w * h).all() and full >= K and q[:full].sum() >= 0.8 * q.sum()
        # serial frames 0, 1 then a burst of the rest onto the same film
        r.reset()
        r.set_regeneration(False)
        for f in range(2):
            r.render(w, h, 5, 3, spp=spp, frame=f, break_on_terminate=brk, download=False)
        r.set_regeneration(True)
        got = r.render_burst(w, h, K - 2, 5, 3, spp=spp, frame=2, break_on_terminate=brk)
        assert got.tobytes() == want[-1].tobytes()
        # not progressive: a burst is its last frame; regeneration does not apply and the serial loop answers
        one = r.render_burst(w, h, 3, 5, 3, spp=spp, frame=0, progressive=False, break_on_terminate=brk)
        r.set_regeneration(False)
        ref = r.render(w, h, 5, 3, spp=spp, frame=2, progressive=False, break_on_terminate=brk)
        assert one.tobytes() == ref.tobytes()
    finally:
        r.close()


@pytest.mark.parametrize("world", [2, 8])
def test_regeneration_on_screen_shards(sponza, world):
    """Every rank of a sharded screen regenerates inside its own tiles: its tile buffer (what the ranks all-gather) and its
    pixels of the film equal the serial loop's on the same shard."""
    from aten_amd.interop import tensor_from_ptr
    fs, cam = sponza
    w, h = 200, 120
    for rank in sorted({0, world - 1, world // 2}):
        r = _ctx(fs, cam, w, h, shard=(rank, world))
        try:
            def tiles():
                r.synchronize()
                return tensor_from_ptr(r.tile_device_ptr(), (r.tile_slots(), 4)).cpu().numpy().copy()
            want_film = []
            r.set_regeneration(False)
            for f in range(3):
                want_film.append(r.render(w, h, 5, 3, spp=2, frame=f, break_on_terminate=False).copy())
            want_tiles = tiles()
            r.reset()
            r.set_regeneration(True)
            got = r.render_burst(w, h, 3, 5, 3, spp=2, frame=0, break_on_terminate=False)
            assert got.tobytes() == want_film[-1].tobytes()
            assert tiles().tobytes() == want_tiles.tobytes()
        finally:
            r.close()


def test_regeneration_with_frames_in_flight_and_other_calls(sponza):
    """Bursts on rotating banks (atn_set_frames_in_flight), interleaved with serial frames: only the film orders them."""
    fs, cam = sponza
    w, h = 160, 90
    r = _ctx(fs, cam, w, h)
    try:
        want, _, _ = _serial_frames(r, w, h, 9, 5, 1, True)
        r.reset()
        r.set_frames_in_flight(3)
        r.set_regeneration(True)
        r.render_burst(w, h, 3, 5, 3, frame=0, download=False)
        r.render(w, h, 5, 3, frame=3, download=False)              # one sample: the serial loop, on the next bank
        r.render_burst(w, h, 4, 5, 3, frame=4, download=False)
        img = r.render_burst(w, h, 1, 5, 3, frame=8)
        assert img.tobytes() == want[8].tobytes()
        r.set_frames_in_flight(1)
    finally:
        r.close()


def test_regeneration_on_the_node_wide_renderer(sponza):
    """atn_mgpu_set_regeneration / atn_mgpu_render_burst: a regenerated pool per shard, one exchange per burst; the assembled film
    equals the unsharded serial loop's."""
    from aten_amd.renderer import MultiGpuPathTracing
    fs, cam = sponza
    w, h = 176, 100
    r = _ctx(fs, cam, w, h)
    try:
        want, _, _ = _serial_frames(r, w, h, 5, 5, 2, False)
    finally:
        r.close()
    m = MultiGpuPathTracing([0, 0, 0])
    try:
        m.UpdateSceneData(fs)
        m.updateCamera(create_camera(cam["pos"], cam["at"], cam["vfov"], w, h))
        m.initSampler(w, h, 0)
        m.set_regeneration(True)
        got = m.render_burst(w, h, 3, 5, 3, spp=2, frame=0, break_on_terminate=False)
        assert got.tobytes() == want[2].tobytes()
        got = m.render_burst(w, h, 2, 5, 3, spp=2, frame=3, break_on_terminate=False)
        assert got.tobytes() == want[4].tobytes()
    finally:
        m.close()


def _variant(name):
    from aten_amd import layout as L
    from aten_amd.scene import scenedefs
    if name == "disney":
        return scenedefs.sponza_lod(mtype=L.MTRL_DISNEY)
    if name == "area+extra":
        return scenedefs.cornell_box_variant(lights="area", extra_materials=True)
    if name == "mixed":
        return scenedefs.cornell_box_variant(lights="mixed", extra_materials="carpaint")
    if name == "sphere":
        return scenedefs.cornell_box_variant(lights="sphere")
    if name == "toon":
        return scenedefs.toon_room(target="area", alpha_blocker=True)
    if name == "atrium":
        return scenedefs.atrium(detail=0.25)
    raise KeyError(name)


@pytest.mark.parametrize("name", ["disney", "area+extra", "mixed", "sphere", "toon", "atrium"])
def test_every_material_set_and_light_kind_regenerates(name):
    """The other k_regen_shade instantiations (Disney, the analytic set, Toon with its inline visibility walk), punctual and
    sphere lights, alpha / stencil lookups of shadow rays (the ALPHA trace flavour), the LDS walk of small scenes."""
    fs, cam = _variant(name)
    w, h = 120, 72
    r = _ctx(fs, cam, w, h)
    try:
        want, rays, shadows = _serial_frames(r, w, h, 2, 6, 3, False, count=True)
        r.reset()
        r.set_regeneration(True)
        got = r.render_burst(w, h, 2, 6, 3, spp=3, frame=0, break_on_terminate=False)
        assert got.tobytes() == want[-1].tobytes()
        q, sh = r.regen_stage_counts()
        assert int(q.sum()) == rays and int(sh.sum()) == shadows
    finally:
        r.close()


def test_an_invalid_sample_does_not_break_the_sample_loop():
    """pathtracing.cpp:339-352: `if (isInvalidColor(c)) continue;` comes BEFORE `if (is_terminated) break;` -- a terminated path whose
    colour is NaN / negative does not stop the pixel.  The Disney atrium produces such samples (0.3 % of its pixels); r06's first
    regenerated epilogue broke on them (found by bench.py's own film check, config.regeneration.film_equals_serial).  Break mode, 8 spp,
    bursts on rotating banks, at a size where hundreds of pixels have an invalid sample."""
    from aten_amd.scene import scenedefs
    fs, cam = scenedefs.atrium(detail=0.25)
    w, h = 320, 180
    r = _ctx(fs, cam, w, h)
    try:
        want, _, _ = _serial_frames(r, w, h, 4, 8, 8, True)
        # the premise: frames of this scene do contain invalid samples (a frame with one sample per pixel shows them as NaN pixels)
        r.reset()
        one = r.render(w, h, 8, 3, spp=1, frame=0)
        assert np.isnan(one[..., :3]).any(axis=-1).sum() >= 20
        r.reset()
        r.set_regeneration(True)
        r.set_frames_in_flight(3)
        r.render_burst(w, h, 2, 8, 3, spp=8, frame=0, break_on_terminate=True, download=False)
        got = r.render_burst(w, h, 2, 8, 3, spp=8, frame=2, break_on_terminate=True)
        assert got.tobytes() == want[3].tobytes()
        r.set_frames_in_flight(1)
    finally:
        r.close()


def test_regeneration_edge_cases(cornell):
    from aten_amd.renderer import AtenAmdError
    fs, cam = cornell
    r = _ctx(fs, cam, 1, 1)
    try:
        r.set_regeneration(False)
        a = r.render(1, 1, 5, 3, spp=4, frame=0)
        r.reset()
        r.set_regeneration(True)
        b = r.render(1, 1, 5, 3, spp=4, frame=0)
        assert a.tobytes() == b.tobytes()
        with pytest.raises(AtenAmdError, match="burst"):
            r.render_burst(1, 1, 0)
        with pytest.raises(AtenAmdError, match="out of range"):
            r.set_regeneration(2)
        # depth 1: every path ends in its first shade, possibly with a shadow ray in flight (the F_PENDING hand-over at every stage)
        r.updateCamera(create_camera(cam["pos"], cam["at"], cam["vfov"], 64, 40))
        r.initSampler(64, 40, 0)
        r.set_regeneration(False)
        r.reset()
        want = [r.render(64, 40, 1, 3, spp=5, frame=f, break_on_terminate=False).copy() for f in range(3)]
        r.reset()
        r.set_regeneration(True)
        got = r.render_burst(64, 40, 3, 1, 3, spp=5, frame=0, break_on_terminate=False)
        assert got.tobytes() == want[-1].tobytes()
    finally:
        r.close()
