"""The host BVH builder (aten_amd/csrc/host/bvh_builder.cpp, SURVEY 8(f)2: sbvh::onBuild / convert): structure of the
split-BVH trees it emits, and their QUALITY -- node visits of the same rays against the reference-built sponza_lod.sbvh,
counted by the oracle (test infrastructure).  No GPU."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, make_camera


def _build(pos, tris, ids, **kw):
    from aten_amd import layout as L
    from aten_amd._hostlib import hostlib, default_bvh_options, BvhStats
    lib = hostlib()
    out = C.c_void_p(); cnt = C.c_uint32(); st = BvhStats()
    bmin = (C.c_float * 3)(); bmax = (C.c_float * 3)()
    opt = default_bvh_options(**kw)
    rc = lib.atns_build_blas_opt(L.ptr(pos), L.ptr(tris), L.ptr(ids), len(ids), C.byref(opt), C.byref(out), C.byref(cnt), bmin, bmax, C.byref(st))
    assert rc == 0
    nodes = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(cnt.value * 48,)).view(L.BVH_NODE).copy()
    lib.atns_free(out)
    return nodes, st, np.array(list(bmin)), np.array(list(bmax))


@pytest.fixture(scope="module")
def sponza_mesh():
    from aten_amd.scene import scenedefs
    fs, cam = scenedefs.sponza_lod(use_sbvh=True, textures=False, ibl=False)
    A = fs.arrays
    return fs, cam, np.ascontiguousarray(A["vtx_pos"]), np.ascontiguousarray(A["triangles"]), np.arange(len(A["triangles"]), dtype=np.uint32)


def test_scene_library_reports_its_abi_version():
    from aten_amd._hostlib import hostlib, ATNS_ABI_VERSION
    assert hostlib().atns_abi_version() == ATNS_ABI_VERSION == 2
    hdr = open(os.path.join(ROOT, "include", "aten_amd_scene.h")).read()
    assert "#define ATNS_ABI_VERSION 2u" in hdr


def test_split_tree_structure(sponza_mesh):
    """Pre-order threading, one triangle per leaf (sbvh.cpp:880-899), every triangle referenced, every leaf box inside its
    triangle's box and inside every ancestor's, duplication inside the budget; object splits alone give 2n - 1 nodes."""
    from aten_amd._hostlib import hostlib
    _, _, pos, tris, ids = sponza_mesh
    n = len(ids)
    nodes, st, bmin, bmax = _build(pos, tris, ids)
    cnt = len(nodes)
    leaf = nodes["f0"] >= 0
    assert st.n_nodes == cnt and st.n_leaves == leaf.sum() and cnt == 2 * leaf.sum() - 1
    assert st.n_spatial_splits > 500 and n < leaf.sum() <= 4 * n
    assert hostlib().atns_validate_nodes(nodes.ctypes.data, cnt) == leaf.sum()
    idx = np.arange(cnt)
    assert np.all(nodes["hit"][~leaf] == idx[~leaf] + 1)
    assert np.all(nodes["hit"][leaf] == nodes["miss"][leaf])
    assert np.all(nodes["hit"][leaf][:-1] == idx[leaf][:-1] + 1) and nodes["hit"][-1] == -1
    assert np.all(nodes["f2"][leaf] == -1.0) and np.all(nodes["f1"][~leaf] == -1.0)
    tid = nodes["f1"][leaf].astype(np.int64)
    assert set(tid.tolist()) == set(range(n))
    # leaf boxes: inside the triangle's own box
    P = pos[:, :3][tris["idx"][tid]]                                    # [leaves, 3, 3]
    assert np.all(nodes["boxmin"][leaf] >= P.min(1) - 1e-6) and np.all(nodes["boxmax"][leaf] <= P.max(1) + 1e-6)
    # a child's box is inside its parent's: walk with an explicit stack of (index, parent)
    miss = nodes["miss"].astype(np.int64)
    end = np.where(miss < 0, cnt, miss)                                 # inner node: one past its subtree
    for i in np.flatnonzero(~leaf):
        a = i + 1
        b = int(end[a]) if not leaf[a] else a + 1
        for ch in (a, b):
            assert np.all(nodes["boxmin"][ch] >= nodes["boxmin"][i]) and np.all(nodes["boxmax"][ch] <= nodes["boxmax"][i])
    assert np.allclose(bmin, pos[:, :3][tris["idx"]].reshape(-1, 3).min(0)) and np.allclose(bmax, pos[:, :3][tris["idx"]].reshape(-1, 3).max(0))

    # the insertion-based optimisation after the build only ever lowers the sum of node areas, and keeps all of the above
    raw, st_raw, _, _ = _build(pos, tris, ids, reinsert_iterations=0)
    assert st_raw.n_reinsertions == 0 and st.n_reinsertions > 1000 and len(raw) == cnt
    assert st.sah_cost < 0.97 * st_raw.sah_cost
    plain, st0, _, _ = _build(pos, tris, ids, spatial_splits=0, reinsert_iterations=0)
    assert len(plain) == 2 * n - 1 and st0.n_spatial_splits == 0
    assert st_raw.sah_cost < 0.9 * st0.sah_cost                         # what the spatial splits are for (both without re-insertion)
    capped, st1, _, _ = _build(pos, tris, ids, max_refs_factor=1.1)
    assert (capped["f0"] >= 0).sum() <= 1.1 * n + 64


def test_trees_of_deforming_meshes_have_one_leaf_per_triangle(sponza_mesh):
    """atn_lbvh_rebuild_list rebuilds a list in place only when it has one leaf per triangle and n - 1 inner nodes.  The entry
    without options (atns_build_blas) builds exactly that -- on a mesh where the default options DO duplicate references -- and
    so does SceneBuilder for a mesh added with deformable=True, whatever the builder's bvh_options say."""
    from aten_amd import layout as L
    from aten_amd._hostlib import hostlib
    from aten_amd.scene.builder import SceneBuilder
    _, _, pos, tris, ids = sponza_mesh
    n = len(ids)
    split, st, _, _ = _build(pos, tris, ids)
    assert (split["f0"] >= 0).sum() > n and st.n_spatial_splits > 0          # the premise: this mesh gets spatial splits
    lib = hostlib()
    out = C.c_void_p(); cnt = C.c_uint32()
    bmin = (C.c_float * 3)(); bmax = (C.c_float * 3)()
    assert lib.atns_build_blas(L.ptr(pos), L.ptr(tris), L.ptr(ids), n, C.byref(out), C.byref(cnt), bmin, bmax) == 0
    nodes = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(cnt.value * 48,)).view(L.BVH_NODE).copy()
    lib.atns_free(out)
    assert len(nodes) == 2 * n - 1 and (nodes["f0"] >= 0).sum() == n
    assert set(nodes["f1"][nodes["f0"] >= 0].astype(np.int64).tolist()) == set(range(n))

    # a deforming mesh whose rest pose invites spatial splits: long thin slats crossing a pile of small triangles
    rng = np.random.default_rng(3)
    P, I = [], []
    for k in range(60):
        y = 0.02 * k
        P += [(-2.0, y, -0.01), (2.0, y, 0.01), (2.0, y + 0.005, 0.0)]
        I.append((3 * k, 3 * k + 1, 3 * k + 2))
    for k in range(400):
        c = rng.uniform(-1.8, 1.8, 3) * (1, 0.3, 0.02)
        P += [tuple(c), tuple(c + (0.03, 0, 0)), tuple(c + (0, 0.03, 0))]
        I.append((180 + 3 * k, 180 + 3 * k + 1, 180 + 3 * k + 2))
    for deformable in (False, True):
        b = SceneBuilder()
        m = b.add_material("m", L.MTRL_DIFFUSE, (0.7, 0.7, 0.7))
        oid = b.add_mesh("slats", np.array(P, np.float32), np.array(I), m, deformable=deformable)
        b.create_instance(oid)
        fs = b.build()
        lst = fs.arrays["bvh_lists"][1]
        leaves = int((lst["f0"] >= 0).sum())
        if deformable:
            assert leaves == len(I) and len(lst) == 2 * len(I) - 1
        else:
            assert leaves > len(I)          # (the same mesh as a static one is split: the flag is what keeps it rebuildable)


def test_split_tree_is_watertight(orc, sponza_mesh):
    """Clipped references must not lose a sliver of their triangle: 200 k incoherent rays (origins inside the building,
    random directions) find the same closest distance in the reference-built tree, in the split tree, and in the tree built
    from object splits alone."""
    fs_ref, _, _, _, _ = sponza_mesh
    from aten_amd.scene import scenedefs
    from aten_amd import layout as L
    own, _ = scenedefs.sponza_lod(use_sbvh=False, textures=False, ibl=False)
    plain, _ = scenedefs.sponza_lod(use_sbvh=False, textures=False, ibl=False, bvh_options=dict(spatial_splits=0))
    rng = np.random.default_rng(11)
    n = 200_000
    rays = np.zeros(n, L.RAY)
    rays["org"] = (rng.uniform(-1, 1, (n, 3)) * np.array([14.0, 6.0, 6.0]) + np.array([0.0, 6.5, 0.0])).astype(np.float32)
    d = rng.normal(size=(n, 3))
    rays["dir"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    ia, _ = orc.trace_closest(fs_ref, rays)
    ib, _ = orc.trace_closest(own, rays)
    ic, _ = orc.trace_closest(plain, rays)
    assert (ia["objid"] >= 0).mean() > 0.9
    # (the mesh holds coincident triangles: where two of them are hit one ulp apart, which one a walk keeps depends on
    # the order it meets their zero-thickness leaf boxes -- in any pair of trees, the reference's own included)
    for ix in (ib, ic):
        assert np.array_equal(ia["objid"] >= 0, ix["objid"] >= 0)
        assert np.allclose(ia["t"], ix["t"], rtol=1e-5, atol=0)
        assert (ia["t"] == ix["t"]).mean() > 0.9995 and (ia["tri_id"] == ix["tri_id"]).mean() > 0.999


def _visits(orc, scene, cam, w, h):
    c = make_camera(orc, cam, w, h)
    _, cnt = orc.render(scene, c, orc.init_sampler(w, h, 0), w, h, 5, 3, counters=True)
    return int(cnt[3]), int(cnt[4]), (int(cnt[0]), int(cnt[2]))


def test_own_tree_costs_no_more_than_the_reference_tree(orc):
    """on sponza_lod.obj the own tree's node visits per 5-bounce frame <= 1.05 x those of the
    reference-built sponza_lod.sbvh (it was 1.25 x with object splits alone).  Same rays, same hits; only the tree differs."""
    from aten_amd.scene import scenedefs
    w, h = 160, 90
    ref, cam = scenedefs.sponza_lod(use_sbvh=True)
    own, _ = scenedefs.sponza_lod(use_sbvh=False)
    vr, tr, rays_r = _visits(orc, ref, cam, w, h)
    vo, to, rays_o = _visits(orc, own, cam, w, h)
    assert rays_r == rays_o
    assert vo <= 0.95 * vr and to <= 0.95 * tr, (vo / vr, to / tr)
    # without the viewer hint (children ordered towards the mesh's centroid) and from another place in the building
    nohint, _ = scenedefs.sponza_lod(use_sbvh=False, bvh_options={})
    for scene_cam in (cam, dict(pos=(-4.0, 0.6, -0.5), at=(1.0, 2.0, 0.2), vfov=60.0)):
        v1, _, _ = _visits(orc, nohint, scene_cam, w, h)
        v0, _, _ = _visits(orc, ref, scene_cam, w, h)
        assert v1 <= 1.06 * v0, v1 / v0


def _brute_force(pos, idx, o, d):
    """Closest Moeller-Trumbore distance over all triangles (fp32, the traverser's acceptance rules), numpy."""
    f = np.float32
    best = np.full(len(o), np.finfo(f).max, f)
    for t in idx:
        v0, v1, v2 = pos[t[0]], pos[t[1]], pos[t[2]]
        e1, e2 = (v1 - v0).astype(f), (v2 - v0).astype(f)
        r = (o - v0).astype(f)
        u = np.cross(d, e2).astype(f); v = np.cross(r, e1).astype(f)
        with np.errstate(all="ignore"):
            inv = f(1) / (u @ e1).astype(f)
            tt = ((v @ e2) * inv).astype(f); be = (np.einsum("ij,ij->i", u, r) * inv).astype(f); ga = (np.einsum("ij,ij->i", v, d) * inv).astype(f)
        ok = (be >= 0) & (be <= 1) & (ga >= 0) & (ga <= 1) & (be + ga <= 1) & (tt >= 0) & (tt > 1e-9) & np.isfinite(tt)
        best = np.where(ok & (tt < best), tt, best)
    return best


def _soups():
    rng = np.random.default_rng(5)
    out = {}
    # long slivers through a cloud of small triangles: what spatial splits are for
    c = rng.uniform(-4, 4, (220, 1, 3)); small = c + rng.normal(0, 0.08, (220, 3, 3))
    a = rng.uniform(-5, 5, (40, 3)); b = a + rng.normal(0, 6, (40, 3)); sl = np.stack([a, b, a + rng.normal(0, 0.05, (40, 3))], 1)
    out["slivers_in_a_cloud"] = np.concatenate([small, sl])
    # twenty copies of one triangle + zero-area triangles + a few normal ones (object splits cannot separate them)
    one = rng.uniform(-1, 1, (1, 3, 3))
    deg = np.repeat(rng.uniform(-1, 1, (10, 1, 3)), 3, axis=1)               # three equal vertices
    col = rng.uniform(-1, 1, (10, 1, 3)) + np.array([[[0, 0, 0]], ] ) * 0
    col = np.concatenate([col, col + 0.3, col + 0.6], 1)                       # collinear
    out["copies_and_degenerates"] = np.concatenate([np.repeat(one, 20, 0), deg, col, rng.uniform(-1, 1, (30, 3, 3))])
    # everything in the plane y = 2 (zero-thickness boxes on one axis), with large triangles overlapping small ones
    flat = rng.uniform(-3, 3, (150, 3, 3)); flat[:, :, 1] = 2.0
    flat[:20] *= 3.0; flat[:20, :, 1] = 2.0
    out["one_plane"] = flat
    # far from the origin: the clip padding is relative
    out["offset_1e5"] = out["slivers_in_a_cloud"] * 10.0 + np.array([1.0e5, -2.0e5, 3.0e5])
    out["n1"] = rng.uniform(-1, 1, (1, 3, 3)); out["n2"] = rng.uniform(-1, 1, (2, 3, 3)); out["n3"] = rng.uniform(-1, 1, (3, 3, 3))
    return {k: v.astype(np.float32) for k, v in out.items()}


@pytest.mark.parametrize("name", ["slivers_in_a_cloud", "copies_and_degenerates", "one_plane", "offset_1e5", "n1", "n2", "n3"])
def test_builder_on_hostile_meshes(orc, name):
    """Degenerate, coincident, coplanar, far-away and tiny inputs: the builder ends, references every triangle, stays inside its
    duplication budget, and the tree finds what testing every triangle finds."""
    from aten_amd import layout as L
    from aten_amd._hostlib import hostlib
    from aten_amd.scene.builder import SceneBuilder
    tri = _soups()[name]
    n = len(tri)
    sb = SceneBuilder()
    m = sb.add_material("m", L.MTRL_DIFFUSE, (0.5, 0.5, 0.5))
    obj = sb.add_mesh("soup", tri.reshape(-1, 3), np.arange(3 * n).reshape(n, 3), m)
    sb.create_instance(obj)
    fs = sb.build()
    nodes = fs.arrays["bvh_lists"][1]
    leaf = nodes["f0"] >= 0
    assert hostlib().atns_validate_nodes(nodes.ctypes.data, len(nodes)) == leaf.sum()
    assert set(nodes["f1"][leaf].astype(int).tolist()) == set(range(n)) and n <= leaf.sum() <= 4 * n + 1
    assert len(nodes) == 2 * leaf.sum() - 1
    # rays: from around the mesh towards points on its triangles (every triangle is aimed at), plus random directions
    rng = np.random.default_rng(17)
    lo, hi = tri.reshape(-1, 3).min(0), tri.reshape(-1, 3).max(0)
    ext = np.maximum(hi - lo, 0.25 * float((hi - lo).max()))      # (a flat mesh is looked at from beside its plane, not from inside it)
    k = 6000
    w = rng.dirichlet((1, 1, 1), k).astype(np.float32)
    target = (tri[rng.integers(0, n, k)] * w[:, :, None]).sum(1)
    org = (lo + hi) / 2 + rng.normal(0, 1, (k, 3)) * ext * 1.5
    d = target - org
    d[k // 2:] = rng.normal(size=(k - k // 2, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros(k, L.RAY); rays["org"] = org.astype(np.float32); rays["dir"] = d.astype(np.float32)
    got, _ = orc.trace_closest(fs, rays)
    want = _brute_force(fs.arrays["vtx_pos"][:, :3], fs.arrays["triangles"]["idx"], rays["org"], rays["dir"])
    hit = got["objid"] >= 0
    t = np.where(hit, got["t"], np.finfo(np.float32).max)
    # (numpy's dot / cross round differently in the last place than the walk's op order: a grazing ray may flip; the clipped
    #  boxes must not lose hits beyond that)
    agree = np.isclose(t, want, rtol=2e-5, atol=0)
    # (collinear triangles have a determinant that is zero only in exact arithmetic: what Moeller-Trumbore returns for them is
    #  rounding noise, different in numpy and in the walk -- 0.5 % of that mesh's rays, with or without spatial splits)
    floor = 0.99 if name == "copies_and_degenerates" else 0.999
    assert agree.mean() >= floor, (name, agree.mean())
    assert (hit == (want < np.finfo(np.float32).max)).mean() >= floor
    if n >= 20:
        assert hit.mean() > 0.3


def test_post_passes_on_an_imported_tree(orc):
    """atns_optimize_nodes on the reference-written sponza_lod.sbvh: the same leaves (payload floats carried over) and the same
    hits, a valid pre-order threaded list, fewer node visits than the file's own arrangement; lists that are not threaded binary
    trees are refused."""
    from aten_amd.scene import scenedefs
    from aten_amd.scene.builder import read_sbvh, optimize_nodes
    from aten_amd._hostlib import hostlib, default_bvh_options
    hdr, _, nodes = read_sbvh(os.path.join(ROOT, "assets", "sponza", "sponza_lod.sbvh"))
    out = optimize_nodes(nodes, C.byref(default_bvh_options(order_point=(0.0, 1.0, 3.0))))
    n = len(nodes)
    assert len(out) == n and hostlib().atns_validate_nodes(out.ctypes.data, n) == 19000
    leaf_in, leaf_out = nodes["f0"] >= 0, out["f0"] >= 0
    key = lambda a, m: sorted(map(tuple, np.stack([a["f1"][m], a["boxmin"][m][:, 0], a["boxmax"][m][:, 2]], 1).tolist()))
    assert key(nodes, leaf_in) == key(out, leaf_out)                       # the same references: triangle id and box
    idx = np.arange(n)
    assert np.all(out["hit"][~leaf_out] == idx[~leaf_out] + 1)             # pre-order
    assert np.all(out["f2"][~leaf_out] == -1.0)                            # voxel payload of inner nodes dropped
    assert np.allclose(out["boxmin"][0], nodes["boxmin"][0]) and np.allclose(out["boxmax"][0], nodes["boxmax"][0])
    ref, cam = scenedefs.sponza_lod(use_sbvh=True)
    opt, _ = scenedefs.sponza_lod(use_sbvh=True, optimize_sbvh=True)
    vr, tr, rays_r = _visits(orc, ref, cam, 160, 90)
    vo, to, rays_o = _visits(orc, opt, cam, 160, 90)
    assert rays_r == rays_o and vo <= 0.96 * vr and to <= tr, (vo / vr, to / tr)
    # not a tree: a link that points backwards / a cycle / an inner node without a second child
    lib = hostlib()
    bad = nodes.copy(); bad["hit"][5] = 2.0
    o = C.c_void_p(); c = C.c_uint32()
    assert lib.atns_optimize_nodes(bad.ctypes.data, n, None, C.byref(o), C.byref(c), None) == -4
    bad = nodes.copy(); bad["miss"][0] = 1.0
    assert lib.atns_optimize_nodes(bad.ctypes.data, n, None, C.byref(o), C.byref(c), None) == -4
    assert lib.atns_optimize_nodes(None, n, None, C.byref(o), C.byref(c), None) == -1


def _twin(nodes):
    from aten_amd import layout as L
    from aten_amd._hostlib import hostlib
    lib = hostlib()
    nodes = np.ascontiguousarray(nodes)
    out = C.c_void_p(); a = C.c_double(); b = C.c_double()
    rc = lib.atns_anyhit_twin(nodes.ctypes.data, len(nodes), C.byref(out), C.byref(a), C.byref(b))
    if rc:
        return rc, None, None, None
    res = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(len(nodes) * 48,)).view(L.BVH_NODE).copy()
    lib.atns_free(out)
    return rc, res, a.value, b.value


def test_anyhit_twin_is_the_same_tree_in_another_child_order():
    """atns_anyhit_twin (csrc/host/anyhit_twin.hpp; what the upload gives a list for its any-hit rays): a valid threaded list in
    pre-order with the SAME nodes -- every box and payload once -- and the same parent-child relations (an inner node keeps its two
    children, only their order may change); the model's expected any-hit cost does not go up.  The reference-built sponza_lod.sbvh
    is expected to gain (0.88), the atrium's regular grids and the Cornell box are not (> 0.95: they get no twin at upload); lists
    that are not binary trees in pre-order are refused."""
    from aten_amd.scene import scenedefs
    from aten_amd.scene.builder import read_sbvh
    from aten_amd._hostlib import hostlib
    _, _, nodes = read_sbvh(os.path.join(ROOT, "assets", "sponza", "sponza_lod.sbvh"))
    n = len(nodes)
    rc, twin, given, cost = _twin(nodes)
    assert rc == 0 and len(twin) == n and hostlib().atns_validate_nodes(twin.ctypes.data, n) == 19000
    assert cost <= 0.9 * given and cost > 0.5 * given
    leaf = twin["f0"] >= 0
    idx = np.arange(n)
    assert np.all(twin["hit"][~leaf] == idx[~leaf] + 1) and np.all(twin["hit"][leaf] == twin["miss"][leaf])
    key = lambda a: sorted(map(tuple, np.concatenate([a["boxmin"], a["boxmax"], a["f0"][:, None], a["f1"][:, None], a["f2"][:, None], a["f3"][:, None]], 1).tolist()))
    assert key(nodes) == key(twin)

    def shape(a):
        """The tree up to the order of siblings: hash(node) = hash(box, payload, {hash(child), hash(child)}), from the root"""
        is_leaf = (a["f0"] >= 0) | (a["f1"] >= 0)
        hit = a["hit"].astype(np.int64); miss = a["miss"].astype(np.int64)
        raw = [a["boxmin"][j].tobytes() + a["boxmax"][j].tobytes() + np.float32([a["f0"][j], a["f1"][j], a["f2"][j], a["f3"][j]]).tobytes() for j in range(len(a))]
        memo = {}
        order = []
        i = 0
        while i >= 0:
            order.append(i); i = int(hit[i])
        for j in reversed(order):                                       # children come after their parent along the hit links
            if is_leaf[j]:
                memo[j] = hash(raw[j])
            else:
                first = int(hit[j])
                second = int(hit[first]) if is_leaf[first] else int(miss[first])
                memo[j] = hash((raw[j], frozenset([memo[first], memo[second]])))
        return memo[0]
    assert shape(nodes) == shape(twin)
    assert not np.array_equal(nodes["f1"], twin["f1"])                  # ... and it IS another order
    # the scenes the model says gain nothing
    for fs in (scenedefs.atrium(detail=0.25)[0], scenedefs.cornell_box()[0]):
        for lst in fs.arrays["bvh_lists"][1:]:
            rc, _, given, cost = _twin(lst)
            if rc == 0:
                assert 0.95 * given < cost <= given * (1 + 1e-6)
    # not a binary tree in pre-order: a backward link, an inner node whose first child's subtree ends where its own does
    bad = nodes.copy(); bad["miss"][0] = 1.0
    assert _twin(bad)[0] == -4
    bad = nodes.copy(); bad["hit"][5] = 2.0
    assert _twin(bad)[0] == -4
    assert _twin(nodes[:1])[0] == -4


def test_upload_host_logic(tmp_path):
    """tests/cxx/upload_host_test.cpp: csrc/host/scene_upload.hpp (pure host C++) compiled with hipcc for its headers and run here
    without a GPU -- the node image with any-hit twins (one / eight) and both layouts can be walked list by list along its typed
    links, every twin holds its list's triangles, the TLAS leaf and the direct-start copy carry the twin word; planar_area_light
    accepts a flat lamp under a rigid matrix (rotated: the normal rotates with it) and refuses a bent, a scaled and a point one."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "upload_host_test")
    lib_dir = os.path.join(ROOT, "aten_amd")
    from aten_amd._hostlib import hostlib
    hostlib()                                                           # (builds libaten_amd_scene.so if it is missing)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-std=c++17", "-O1", "-w", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cxx", "upload_host_test.cpp"), "-L", lib_dir, "-laten_amd_scene",
                           "-Wl,-rpath," + lib_dir, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
