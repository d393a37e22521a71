"""The any-hit twins (csrc/host/anyhit_twin.hpp, scene_upload.hpp): at upload a bottom-level list may get further threadings of the
same tree, in child orders an any-hit walk is expected to finish sooner in (one per octant of the ray's direction), and the rays that only ask "is anything in the way
of an infinite light" (stop_t = +inf: IBL, directional) walk it.  The answer of such a walk does not depend on the order -- until the
first accepted hit t_max is the constant the ray came with -- so every film must be BYTE-EQUAL to the one rendered without twins
(ATEN_AMD_ANYHIT_TWIN=0), whatever the scene, walk flavour or update path; only the shadow rays' visit counters go down."""
import os

import numpy as np
import pytest

from conftest import make_camera

pytestmark = pytest.mark.gpu


class _Env:
    def __init__(self, **kw):
        self.kw = kw
    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        for k, v in self.kw.items():
            os.environ[k] = str(v)
    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


def _render(fs, c, w, h, twin, frames=(0, 3), depth=5, stats=False, flavour=None, lds=None, dirs=8):
    from aten_amd.renderer import PathTracing
    env = dict(ATEN_AMD_ANYHIT_TWIN=twin, ATEN_AMD_ANYHIT_TWIN_DIRS=dirs)
    if flavour is not None: env["ATEN_AMD_TRACE"] = flavour
    if lds is not None: env["ATEN_AMD_LDS_NODES"] = lds
    with _Env(**env):
        r = PathTracing(0)
        try:
            r.UpdateSceneData(fs); r.updateCamera(c); r.initSampler(w, h, 0)
            n_lists = len(fs.arrays["bvh_lists"]) - 1
            assert r.anyhit_twins() <= n_lists and (twin != 0 or r.anyhit_twins() == 0) and (twin != 2 or r.anyhit_twins() >= 1)
            films = [r.render(w, h, depth, 3, frame=f, count_stats=stats).copy() for f in frames]
            st = r.stats() if stats else None
            if st is not None: st["twins"] = r.anyhit_twins()
            return films, st
        finally:
            r.close()


def test_headline_scene_films_are_byte_equal_and_shadow_walks_shorter(orc, sponza):
    """The reference-built sponza_lod.sbvh with IBL (every shadow ray is an any-hit ray): films with and without the twin are the
    same bytes; closest-hit visits, rays and hits are the same numbers; shadow-ray node visits fall by more than 15 % with one twin,
    by more than 35 % with one per direction octant."""
    fs, cam = sponza
    w, h = 256, 144
    c = make_camera(orc, cam, w, h)
    a, sa = _render(fs, c, w, h, 0, stats=True)
    for dirs, bound in ((1, 0.85), (8, 0.65)):       # one direction-free twin; eight, one per octant of the ray's direction (the default)
        b, sb = _render(fs, c, w, h, 1, stats=True, dirs=dirs)
        for x, y in zip(a, b):
            assert x.tobytes() == y.tobytes()
        for k in ("closest_rays", "shadow_rays", "hits", "closest_nodes", "closest_tris"):
            assert sa[k] == sb[k], (k, sa[k], sb[k])
        assert sb["shadow_nodes"] < bound * sa["shadow_nodes"], (dirs, sb["shadow_nodes"], sa["shadow_nodes"])
        assert sa["twins"] == 0 and sb["twins"] == 1
    # the plain walk takes the same turn at the same place
    p, _ = _render(fs, c, w, h, 1, flavour="s")
    assert p[0].tobytes() == a[0].tobytes() and p[1].tobytes() == a[1].tobytes()


def test_every_light_kind_and_every_way_into_a_nested_tree(orc):
    """The Cornell variant with area + point + spot + directional lights (only the directional light's shadow rays are any-hit rays:
    area lights need the closest hit's object, punctual ones its distance), twins forced onto every list (the model would give these
    small trees none): identity and transformed instances behind a top layer, refill walk / plain walk / plain walk over an LDS copy."""
    from aten_amd.scene import scenedefs
    fs, cam = scenedefs.cornell_box_variant(lights="mixed")
    w, h = 96, 96
    c = make_camera(orc, cam, w, h)
    want, s0 = _render(fs, c, w, h, 0, stats=True, flavour="s", lds="0")
    for dirs in (1, 8):     # (with eight twins the node image outgrows the LDS copy: ("s", "1") then walks global memory like ("s", "0"))
        for flavour, lds in (("r", "0"), ("s", "0"), ("s", "1")):
            got, s2 = _render(fs, c, w, h, 2, stats=True, flavour=flavour, lds=lds, dirs=dirs)
            assert got[0].tobytes() == want[0].tobytes() and got[1].tobytes() == want[1].tobytes(), (dirs, flavour, lds)
            assert s2["closest_nodes"] == s0["closest_nodes"] and s2["shadow_rays"] == s0["shadow_rays"]


def test_a_rebuilt_list_keeps_its_twins(orc):
    """atn_lbvh_rebuild_list replaces the tree the twins were threadings of: they are re-threaded on the device from the new tree
    (lbvh.hpp, k_lbvh_twin_*), also with frames in flight -- the list still has its twins afterwards, and the frames after the rebuild
    equal those of a context that never had any, byte for byte."""
    from aten_amd.renderer import PathTracing
    from aten_amd.scene import scenedefs
    from aten_amd.scene.camera import create_camera
    from test_gpu_lbvh import tick_data, push_tick
    W, H = 160, 120
    b, oid, cam = scenedefs.deformable_room(0.0)
    env = scenedefs.synthetic_envmap(256, 128)
    tid = b.add_texture("sky", env)
    b.add_ibl(tid, avg_illum=scenedefs.envmap_avg_illum(env))                       # an infinite light: any-hit shadow rays
    c = create_camera(cam["pos"], cam["at"], cam["vfov"], W, H)
    films = {}
    for twin in (0, 2):
        with _Env(ATEN_AMD_ANYHIT_TWIN=twin):
            r = PathTracing(0)
            try:
                fs0, d0 = tick_data(b, oid, 0.0)
                r.UpdateSceneData(fs0); r.updateCamera(c); r.initSampler(W, H, 0)
                r.set_frames_in_flight(3)
                out = [r.render(W, H, frame=0).copy()]
                for f in range(1, 4):
                    r.render(W, H, frame=f, download=False)                          # frames of the old tree still in flight
                fs, d = tick_data(b, oid, 1.3)
                n_before = r.anyhit_twins()
                push_tick(r, fs, d)
                assert r.anyhit_twins() == n_before and (n_before >= 1 if twin else n_before == 0)     # the rebuilt list has its twins still
                r.reset()
                out.append(r.render(W, H, frame=7).copy())
                fs, d = tick_data(b, oid, 2.2)
                push_tick(r, fs, d)
                r.reset()
                out.append(r.render(W, H, frame=8).copy())
                r.set_frames_in_flight(1)
                r.reset()
                out.append(r.render(W, H, frame=9, count_stats=True).copy())
                films[(twin, "stats")] = r.stats()
                films[twin] = out
            finally:
                r.close()
    for x, y in zip(films[0], films[2]):
        assert x.tobytes() == y.tobytes()
    assert films[0][0].tobytes() != films[0][1].tobytes()
    s0, s2 = films[(0, "stats")], films[(2, "stats")]
    for k in ("closest_rays", "shadow_rays", "hits", "closest_nodes", "closest_tris"):
        assert s0[k] == s2[k]
    assert s2["shadow_nodes"] != s0["shadow_nodes"]          # (the any-hit rays do walk other threadings after the rebuild)


def test_top_layer_update_keeps_the_twins(orc):
    """atn_update_tlas re-emits the TLAS leaves with their twin words: moving the instanced boxes gives the same bytes with and
    without twins, and the same as a full upload of the moved scene."""
    from aten_amd.renderer import PathTracing
    from aten_amd.scene import scenedefs
    still, cam = scenedefs.cornell_box_variant(lights="mixed", move_boxes=False)
    moved, _ = scenedefs.cornell_box_variant(lights="mixed", move_boxes=True)
    w, h = 80, 80
    c = make_camera(orc, cam, w, h)
    want, _ = _render(moved, c, w, h, 0, frames=(2,))
    for twin in (0, 2):
        with _Env(ATEN_AMD_ANYHIT_TWIN=twin):
            r = PathTracing(0)
            try:
                r.UpdateSceneData(still); r.updateCamera(c); r.initSampler(w, h, 0)
                base = r.render(w, h, 5, 3, frame=2).copy()
                r.updateBVH(moved); r.reset()
                assert r.render(w, h, 5, 3, frame=2).tobytes() == want[0].tobytes()
                r.updateBVH(still); r.reset()
                assert r.render(w, h, 5, 3, frame=2).tobytes() == base.tobytes()
            finally:
                r.close()


def test_screen_shards_with_twins_equal_one_context_without(orc, sponza):
    """atn_mgpu_*: every shard uploads its own replica -- with twins -- and the assembled film is the film of one context without."""
    from aten_amd.renderer import MultiGpuPathTracing
    fs, cam = sponza
    w, h = 192, 108
    c = make_camera(orc, cam, w, h)
    want, _ = _render(fs, c, w, h, 0, frames=(4,))
    with _Env(ATEN_AMD_ANYHIT_TWIN=1, ATEN_AMD_ANYHIT_TWIN_DIRS=8):
        m = MultiGpuPathTracing([0, 0, 0])
        try:
            m.UpdateSceneData(fs); m.updateCamera(c); m.initSampler(w, h, 0)
            assert m.render(w, h, 5, 3, frame=4).tobytes() == want[0].tobytes()
        finally:
            m.close()


def test_node_layout_changes_no_film(orc, sponza):
    """Where the records lie in the node image (ATEN_AMD_NODE_LAYOUT: walk order, or the top levels first -- the default) decides
    nothing: links are explicit.  Films byte-equal on the headline scene (one deep list + eight twins) and behind a top layer."""
    from aten_amd.scene import scenedefs
    for (fs, cam), w, h in ((sponza, 192, 108), (scenedefs.cornell_box_variant(lights="mixed"), 96, 96)):
        c = make_camera(orc, cam, w, h)
        films = {}
        for layout in (0, 1):
            with _Env(ATEN_AMD_NODE_LAYOUT=layout):
                films[layout], _ = _render(fs, c, w, h, 2, frames=(1,))
        assert films[0][0].tobytes() == films[1][0].tobytes()


def test_shadow_rays_towards_planar_area_lights_stop_early(orc):
    """An area light's shadow ray needs the closest hit's OBJECT -- but when the light's object is planar and rigidly placed the ray meets
    it at distToLight and nowhere else, so a hit nearer than that is on another object and settles "blocked" (scene_upload.hpp,
    planar_area_light).  Films byte-equal with the rule off (ATEN_AMD_PLANAR_LIGHTS=0); shadow walks shorter where lights are occluded;
    a sphere light is not treated so; boxes moving through atn_update_tlas leave the lamp's flag alone."""
    from aten_amd.renderer import PathTracing
    from aten_amd.scene import scenedefs

    def run(fs, cam, w, h, on, depth=5, move=None):
        with _Env(ATEN_AMD_PLANAR_LIGHTS=on):
            r = PathTracing(0)
            try:
                c = make_camera(orc, cam, w, h)
                r.UpdateSceneData(fs); r.updateCamera(c); r.initSampler(w, h, 0)
                n0 = r.planar_area_lights()
                if move is not None:
                    r.updateBVH(move); r.reset()
                films = [r.render(w, h, depth, 3, frame=f, count_stats=True).copy() for f in (0, 2)]
                return films, r.stats(), n0, r.planar_area_lights()
            finally:
                r.close()
    fs, cam = scenedefs.atrium(detail=0.25)
    a, sa, na, _ = run(fs, cam, 160, 90, 0)
    b, sb, nb, _ = run(fs, cam, 160, 90, 1)
    assert na == 0 and nb == 1
    assert all(x.tobytes() == y.tobytes() for x, y in zip(a, b))
    for k in ("closest_rays", "shadow_rays", "hits", "closest_nodes"):
        assert sa[k] == sb[k]
    assert sb["shadow_nodes"] < 0.97 * sa["shadow_nodes"], (sb["shadow_nodes"], sa["shadow_nodes"])
    # the Cornell box: the lamp is a planar quad under the identity
    fs, cam = scenedefs.cornell_box()
    a, sa, _, _ = run(fs, cam, 128, 128, 0)
    b, sb, nb, _ = run(fs, cam, 128, 128, 1)
    assert nb == 1 and all(x.tobytes() == y.tobytes() for x, y in zip(a, b)) and sb["shadow_nodes"] <= sa["shadow_nodes"]
    # a sphere light is no planar polygon object; every light kind beside the lamp
    sp, cam = scenedefs.cornell_box_variant(lights="sphere")
    _, _, n_sp, _ = run(sp, cam, 64, 64, 1, depth=2)
    assert n_sp == 0
    mixed, cam = scenedefs.cornell_box_variant(lights="mixed")
    a, _, _, _ = run(mixed, cam, 96, 96, 0)
    b, _, nb, _ = run(mixed, cam, 96, 96, 1)
    assert nb == 1 and all(x.tobytes() == y.tobytes() for x, y in zip(a, b))
    # new instance matrices through atn_update_tlas that move the BOXES: the lamp's records and matrices come back byte for byte, the
    # flags hold (r06; until then any update dropped them), results as ever
    still, cam = scenedefs.cornell_box_variant(lights="area", move_boxes=False)
    moved, _ = scenedefs.cornell_box_variant(lights="area", move_boxes=True)
    a, sa, _, _ = run(still, cam, 80, 80, 0, move=moved)
    b, sb, n_before, n_after = run(still, cam, 80, 80, 1, move=moved)
    assert n_before == 1 and n_after == 1 and all(x.tobytes() == y.tobytes() for x, y in zip(a, b))
    assert sb["shadow_nodes"] <= sa["shadow_nodes"]


def _hostile_lamp_scene(offset=(0.0, 0.0, 0.0)):
    """A lamp quad on a ROTATED rigid instance (25 degrees about z), three thin blocker strips parallel to it a hair in front of it --
    gaps of 4.5e-3 / 2.4e-3 / 3e-4, i.e. hits at ~0.9985 / 0.9992 / 0.9999 of the light distance for a point 3 units away: below, inside
    and at the far end of the rule's 0.999 band --, a floor, and a wall that crosses the lamp's plane (points near the crossing see the
    lamp at cosines around 0.005 - 0.02).  `offset` translates every vertex and the camera (1e4: the origin-offset bound of the rule is
    then the binding term and the thinnest gaps are below the coordinates' resolution)."""
    from aten_amd import layout as L
    from aten_amd.scene.builder import SceneBuilder
    off = np.asarray(offset, np.float32)
    b = SceneBuilder()
    white = b.add_material("white", L.MTRL_DIFFUSE, (0.7, 0.7, 0.7))
    red = b.add_material("red", L.MTRL_DIFFUSE, (0.7, 0.2, 0.2))
    emit = b.add_material("lamp", L.MTRL_EMISSIVE, (1.0, 1.0, 1.0))

    def quad(name, p, mtrl, mtx=None, flip=False):
        idx = [[0, 2, 1], [0, 3, 2]] if flip else [[0, 1, 2], [0, 2, 3]]
        o = b.add_mesh(name, np.asarray(p, np.float32), idx, mtrl)
        return b.create_instance(o, mtx)
    a = np.deg2rad(25.0)
    R = np.array([[np.cos(a), -np.sin(a), 0, 0], [np.sin(a), np.cos(a), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float64)
    T = np.eye(4); T[:3, 3] = np.array([0.0, 3.0, 0.0]) + off.astype(np.float64)
    M = (T @ R).astype(np.float32)
    # (local frame: the lamp faces -y; the strips lie at local y = -gap, side by side along x)
    lamp = quad("lamp", [[-1.0, 0, -0.6], [1.0, 0, -0.6], [1.0, 0, 0.6], [-1.0, 0, 0.6]], emit, M)
    b.add_area_light(lamp, (1.0, 1.0, 1.0), 40.0)
    for k, gap in enumerate((4.5e-3, 2.4e-3, 3e-4)):
        x0 = -0.9 + 0.6 * k
        quad("strip%d" % k, [[x0, -gap, -0.5], [x0 + 0.4, -gap, -0.5], [x0 + 0.4, -gap, 0.5], [x0, -gap, 0.5]], red, M)
    fl = np.array([[-4, 0, -3], [4, 0, -3], [4, 0, 3], [-4, 0, 3]], np.float32) + off
    quad("floor", fl, white, None, flip=True)
    wl = np.array([[-1.6, 0, -3], [-1.6, 0, 3], [-1.6, 5, 3], [-1.6, 5, -3]], np.float32) + off
    quad("wall", wl, white, None)
    quad("wall_back", wl[[0, 3, 2, 1]], white, None)
    b.set_background((0.0, 0.0, 0.0))
    cam = dict(pos=tuple((np.array([1.0, 1.6, 6.0]) + off).tolist()), at=tuple((np.array([-0.3, 1.8, 0.0]) + off).tolist()), vfov=45.0)
    return b.build(), cam


@pytest.mark.parametrize("offset", [(0.0, 0.0, 0.0), (1e4, 1e4, 1e4)])
def test_planar_light_rule_on_hostile_geometry(orc, offset):
    """The three constants of the early stop towards planar lights (kernels.hpp ShadowJob::fetch_slot: 0.999 of the distance, cosine >=
    0.01, the 300-ulp origin bound) against geometry chosen to sit on them: films byte-equal with the rule off, and inside the
    tolerance of the oracle -- which walks every shadow ray to its closest hit."""
    from aten_amd.renderer import PathTracing
    fs, cam = _hostile_lamp_scene(offset)
    w, h = 160, 120
    c = make_camera(orc, cam, w, h)
    films = {}
    for on in (0, 1):
        r = PathTracing(0)
        try:
            r.set_upload_options(planar_lights=on)
            r.UpdateSceneData(fs); r.updateCamera(c); r.initSampler(w, h, 0)
            assert r.planar_area_lights() == on          # the rotated lamp IS recognised (rigid matrix, planar quad)
            films[on] = [r.render(w, h, 3, 3, frame=f, count_stats=True).copy() for f in (0, 1, 5)]
            films[(on, "stats")] = r.stats()
        finally:
            r.close()
    for a, b in zip(films[0], films[1]):
        assert a.tobytes() == b.tobytes()
    for k in ("closest_rays", "shadow_rays", "hits"):
        assert films[(0, "stats")][k] == films[(1, "stats")][k]
    seeds = orc.init_sampler(w, h, 0)
    want = orc.render(fs, c, seeds, w, h, 3, 3, frame=0)
    got = films[1][0]
    d = np.abs(got[..., :3] - want[..., :3])
    inside = np.all(d <= 1e-3 * np.maximum(1.0, np.abs(want[..., :3])), axis=-1)
    assert inside.mean() >= 0.99, inside.mean()
    lit = want[..., :3].sum(-1) > 0
    assert lit.mean() > 0.2             # (the lamp does light the scene: the comparison is not about black pixels)


def test_an_update_keeps_the_planar_light_certificate_only_if_the_lamp_comes_back_unchanged(orc, cornell):
    """atn_update_tlas (with or without matrices) and atn_update_geometry keep the planar / rigid flags found at upload exactly when
    every flagged light's object records, matrices, triangles and vertices come back byte for byte; an update that re-points the
    lamp's instance, gives it another matrix or writes one of its vertices drops them.  The film is what a fresh upload renders
    either way."""
    import copy
    from aten_amd import layout as L
    from aten_amd.renderer import PathTracing
    fs, cam = cornell
    w, h = 96, 96
    c = make_camera(orc, cam, w, h)
    a = fs.arrays
    lamp = int(a["lights"][0]["arealight_objid"])
    o = a["objects"][lamp]
    mesh = int(o["object_id"]) if int(o["type"]) == L.OBJ_INSTANCE else lamp
    t0, tn = int(a["objects"][mesh]["triangle_id"]), int(a["objects"][mesh]["triangle_num"])
    v = int(a["triangles"][t0]["idx"][0])

    def fresh():
        r = PathTracing(0)
        r.UpdateSceneData(fs); r.updateCamera(c); r.initSampler(w, h, 0)
        assert r.planar_area_lights() == 1
        return r

    r = fresh()
    try:
        want = r.render(w, h, 5, 3, frame=0).copy()
        r.updateBVH(fs, with_matrices=False)                        # the same objects: the flags hold
        assert r.planar_area_lights() == 1
        r.updateBVH(fs)                                             # the same objects and matrices
        assert r.planar_area_lights() == 1
        lo = min(int(a["triangles"][t]["idx"].min()) for t in range(t0, t0 + tn))
        hi = max(int(a["triangles"][t]["idx"].max()) for t in range(t0, t0 + tn))
        other = hi + 1 if hi + 1 < len(a["vtx_pos"]) else lo - 1     # a vertex that is not the lamp's
        assert 0 <= other < len(a["vtx_pos"]) and not (lo <= other <= hi) and lo <= v <= hi
        r.updateGeometry(vtx_pos=a["vtx_pos"][other:other + 1], vtx_nml=a["vtx_nml"][other:other + 1], vtx_offset=other)
        assert r.planar_area_lights() == 1
        r.reset()
        assert r.render(w, h, 5, 3, frame=0).tobytes() == want.tobytes()
        r.updateGeometry(vtx_pos=a["vtx_pos"][v:v + 1], vtx_nml=a["vtx_nml"][v:v + 1], vtx_offset=v)      # one of the lamp's vertices (same value)
        assert r.planar_area_lights() == 0
        r.updateBVH(fs)                                             # never re-derived
        assert r.planar_area_lights() == 0
        r.reset()
        assert r.render(w, h, 5, 3, frame=0).tobytes() == want.tobytes()
    finally:
        r.close()
    r = fresh()
    try:
        r.updateGeometry(triangles=a["triangles"][t0:t0 + 1], tri_offset=t0)                               # one of the lamp's triangles
        assert r.planar_area_lights() == 0
    finally:
        r.close()
    assert int(o["type"]) == L.OBJ_INSTANCE and int(o["mtx_id"]) >= 0        # (the Cornell lamp is an instance with a matrix pair)
    moved = copy.copy(fs)
    moved.arrays = dict(a)
    m = a["matrices"].copy()
    m[int(o["mtx_id"])][0][3] += np.float32(0.25)                   # the lamp moved (both matrices, consistently)
    m[int(o["mtx_id"]) + 1][0][3] -= np.float32(0.25)
    moved.arrays["matrices"] = m
    r = fresh()
    try:
        r.updateBVH(moved)
        assert r.planar_area_lights() == 0
    finally:
        r.close()
    repointed = copy.copy(fs)
    repointed.arrays = dict(a)
    objs = a["objects"].copy()
    objs[lamp]["mtx_id"] = 2 if int(o["mtx_id"]) != 2 else 4          # another instance's matrix pair
    assert int(objs[lamp]["mtx_id"]) != int(o["mtx_id"])
    repointed.arrays["objects"] = objs
    r = fresh()
    try:
        r.updateBVH(repointed, with_matrices=False)
        assert r.planar_area_lights() == 0
    finally:
        r.close()


def test_a_deformation_tick_keeps_the_planar_light_rule(orc):
    """The deforming-mesh room: a tick writes the blob's vertices, rebuilds its list and hands the objects back -- the lamp is not
    touched, so its shadow rays keep stopping early (r06: they used not to, +4 % per frame after the first tick); the frames equal
    those of a context that never had the rule, byte for byte, and shadow rays visit fewer nodes."""
    from aten_amd.renderer import PathTracing
    from aten_amd.scene import scenedefs
    from aten_amd.scene.camera import create_camera
    from test_gpu_lbvh import tick_data, push_tick
    W, H = 160, 120
    b, oid, cam = scenedefs.deformable_room(0.0)
    c = create_camera(cam["pos"], cam["at"], cam["vfov"], W, H)
    out = {}
    for rule in (0, 1):
        r = PathTracing(0)
        try:
            r.set_upload_options(planar_lights=rule)
            fs0, d0 = tick_data(b, oid, 0.0)
            r.UpdateSceneData(fs0); r.updateCamera(c); r.initSampler(W, H, 0)
            assert r.planar_area_lights() == rule
            r.set_frames_in_flight(3)
            films = []
            for k, t in enumerate((1.3, 2.2)):
                for f in range(3):
                    r.render(W, H, frame=f, download=False)                      # frames of the old tree still in flight
                fs, d = tick_data(b, oid, t)
                push_tick(r, fs, d)
                assert r.planar_area_lights() == rule
                r.reset()
                films.append(r.render(W, H, frame=7 + k).copy())
            r.set_frames_in_flight(1)
            r.reset()
            films.append(r.render(W, H, frame=9, count_stats=True).copy())
            out[rule] = (films, r.stats())
        finally:
            r.close()
    for x, y in zip(out[0][0], out[1][0]):
        assert x.tobytes() == y.tobytes()
    s0, s1 = out[0][1], out[1][1]
    for k in ("closest_rays", "shadow_rays", "hits", "closest_nodes", "closest_tris"):
        assert s0[k] == s1[k]
    assert s1["shadow_nodes"] < s0["shadow_nodes"]
