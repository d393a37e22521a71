"""CPU tests (-m "not gpu"): the oracle against the reference's own known answers and against the
committed golden vectors; host-side scene logic; the C-ABI library's export table."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, make_camera, ulp_diff


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(GOLDEN, "oracle_golden.npz"))


# ---- the reference's own known answers --------------------------------------------------------
def test_camera_kat_from_reference_unittest(orc):
    """src/aten_unittest/pinhole_camera.cpp:6-16: ComputePixelWidthAtDistance(vfov 60, 1280x720, 1)
    EXPECT_FLOAT_EQ 0.000473979220 (gtest FLOAT_EQ = within 4 ulp)."""
    from aten_amd import layout as L
    cam = np.zeros((), L.CAMERA_PARAM)
    cam["vfov"], cam["width"], cam["height"] = 60, 1280, 720
    got = np.float32(orc.pixel_width_at_distance(cam, 1.0))
    assert ulp_diff(got, np.float32(0.000473979220)) <= 4


def test_mt19937_seeds(orc, golden):
    """aten::initSampler draws std::mt19937(0) in pixel order (sampler.cpp:8-18); the first outputs of
    mt19937 seeded with 0 are standard constants."""
    s = orc.init_sampler(512, 512, 0)
    assert s[0] == 2357136044 and s[1] == 2546248239 and s[2] == 3071714933
    assert np.array_equal(s[:64], golden["seeds_512"])
    # seeds depend on W*H only through the count
    assert np.array_equal(orc.init_sampler(64, 64, 0), s[:4096])


def test_reference_sbvh_pins_the_ingestion_triangle_numbering():
    """asset/sponza/sponza_lod.sbvh was written by the reference (sbvh::exportTree, accelerator/sbvh.cpp:1237-1338) from the
    triangles ITS ObjLoader (libatenscene/ObjLoader.cpp:95-461) produced for sponza_lod.obj, and every leaf carries the
    object-local id of its triangle.  So the file pins our ingestion (native obj_ingest.cpp through SceneBuilder.load_obj)
    from outside: leaf k's box must lie inside the bounding box of OUR triangle number triid(k) -- spatial-split leaves are
    clipped to the split planes, the others are the triangle's own box bit for bit -- every triangle must be named by some
    leaf, and the root box must be the vertex bounds.  A different triangulation order, corner order or shape order in the
    loader would break all of these."""
    from aten_amd.scene import scenedefs
    from aten_amd.scene.builder import read_sbvh
    fs, _ = scenedefs.sponza_lod(textures=False, ibl=False)
    tris, pos = fs.arrays["triangles"], fs.arrays["vtx_pos"][:, :3]
    obj = fs.arrays["objects"][0]
    assert obj["triangle_id"] == 0 and obj["triangle_num"] == len(tris) == 12852
    hdr, _, nodes = read_sbvh(os.path.join(ROOT, "assets", "sponza", "sponza_lod.sbvh"))
    leaf = nodes["f1"] >= 0
    tid = nodes["f1"][leaf].astype(np.int64)
    assert leaf.sum() == 19000 and tid.min() == 0 and tid.max() == len(tris) - 1
    assert len(np.unique(tid)) == len(tris)                  # every triangle of the object is referenced
    corners = pos[tris["idx"][tid]]                          # [leaf, corner, xyz]
    tmin, tmax = corners.min(axis=1), corners.max(axis=1)
    bmin, bmax = nodes["boxmin"][leaf], nodes["boxmax"][leaf]
    # inside the triangle's box (exact compare: a clip plane lies between the triangle's extremes, an unclipped side IS the extreme)
    assert np.all(bmin >= tmin) and np.all(bmax <= tmax)
    assert np.all(bmin <= bmax)
    same = np.all(bmin == tmin, axis=1) & np.all(bmax == tmax, axis=1)
    assert same.mean() >= 0.40, same.mean()                  # measured 0.467: the leaves spatial splits did not touch
    # a triangle referenced once was never split: its single leaf box is its own box
    cnt = np.bincount(tid, minlength=len(tris))
    once = cnt[tid] == 1
    assert once.sum() > 5000 and np.all(same[once])
    # the root box and the file header are the vertex bounds of the object, bit for bit
    vmin, vmax = pos[np.unique(tris["idx"])].min(axis=0), pos[np.unique(tris["idx"])].max(axis=0)
    assert np.array_equal(nodes["boxmin"][0], vmin) and np.array_equal(nodes["boxmax"][0], vmax)
    assert np.array_equal(np.float32(hdr["boxmin"]), vmin) and np.array_equal(np.float32(hdr["boxmax"]), vmax)


def test_sbvh_fixture_from_reference_asset():
    """asset/sponza/sponza_lod.sbvh is a reference-written file (format accelerator/sbvh.cpp:1220-1338)."""
    from aten_amd.scene.builder import read_sbvh
    from aten_amd._hostlib import hostlib
    hdr, names, nodes = read_sbvh(os.path.join(ROOT, "assets", "sponza", "sponza_lod.sbvh"))
    assert hdr["nodeNum"] == 37999 and hdr["maxDepth"] == 23
    leaf = nodes["f0"] >= 0
    assert leaf.sum() == 19000                      # 19,000 leaves
    assert (nodes["f2"] < -1).sum() == 6238         # voxel-LOD nodes
    assert nodes["f1"][leaf].min() == 0 and nodes["f1"][leaf].max() == 12851
    assert len(names) == 20
    # every link valid, hit-walk reaches every leaf
    assert hostlib().atns_validate_nodes(nodes.ctypes.data, len(nodes)) == 19000
    assert np.allclose(hdr["boxmin"], nodes["boxmin"][0]) and np.allclose(hdr["boxmax"], nodes["boxmax"][0])


def test_sbvh_export_round_trips(tmp_path):
    """write_sbvh is the inverse of read_sbvh: the reference-written asset comes back byte for byte
    (sbvh::exportTree's layout, accelerator/sbvh.cpp:1237-1338), and threaded_depth reproduces the header's maxDepth."""
    from aten_amd.scene.builder import read_sbvh, write_sbvh, threaded_depth
    src = os.path.join(ROOT, "assets", "sponza", "sponza_lod.sbvh")
    hdr, names, nodes = read_sbvh(src)
    assert threaded_depth(nodes) == hdr["maxDepth"]
    dst = str(tmp_path / "copy.sbvh")
    write_sbvh(dst, nodes, hdr["boxmin"], hdr["boxmax"], None, names, hdr["version"])
    assert open(src, "rb").read() == open(dst, "rb").read()


def test_sbvh_export_of_built_tree_imports_identically(tmp_path):
    """A tree built by atns_build_blas, exported with object-local triangle ids and imported again through
    PolygonObject::importInternalAccelTree's path gives the same flattened scene as building it in place."""
    from aten_amd.scene.builder import SceneBuilder, read_sbvh
    from aten_amd import layout as L
    sb = SceneBuilder()
    m = sb.add_material("m", L.MTRL_DIFFUSE, (0.5, 0.5, 0.5))
    rng = np.random.default_rng(3)
    # a first small object so that the exported object's global triangle ids do not start at 0
    sb.create_instance(sb.add_mesh("pad", rng.random((12, 3)).astype(np.float32), np.arange(12).reshape(4, 3), m))
    c = rng.random((300, 1, 3)).astype(np.float32) * 4
    p = (c + rng.random((300, 3, 3)).astype(np.float32) * 0.3).reshape(-1, 3)
    big = sb.add_mesh("soup", p, np.arange(900).reshape(300, 3), m)
    sb.create_instance(big)
    built = sb.build()
    path = str(tmp_path / "obj.sbvh")
    sb.export_sbvh(big, path)
    hdr, names, nodes = read_sbvh(path)
    assert names == {} and hdr["version"] == 0x01000000
    leaf = nodes["f0"] >= 0
    assert nodes["f1"][leaf].min() == 0
    sb.import_sbvh(big, path)
    again = sb.build()
    for a, b in zip(built.arrays["bvh_lists"], again.arrays["bvh_lists"]):
        assert a.tobytes() == b.tobytes()


def test_lbvh_hierarchy_matches_karras_figure(orc):
    """The keys of the reference's disabled builder self-test (src/libidaten/kernel/LBVHBuilder.cu:877-880),
    {1, 19, 24, 25, 30, 2, 4, 5}, are -- sorted -- the 5-bit keys of Figure 3 of Karras 2012 (the paper the file cites,
    :11-12): 00001 00010 00100 00101 10011 11000 11001 11110.  The figure's tree: node 0 = [0,7] splits after 3 into
    nodes 3 and 4; 3 = [0,3] -> nodes 1 and 2; 1 -> leaves 0, 1; 2 -> leaves 2, 3; 4 = [4,7] -> leaf 4 and node 5;
    5 = [5,7] -> node 6 and leaf 7; 6 -> leaves 5, 6.  Leaves are numbered n - 1 + i = 7 + i here."""
    keys = sorted([1, 19, 24, 25, 30, 2, 4, 5])
    left, right, parent = orc.lbvh_hierarchy(keys)
    assert left[:7].tolist() == [3, 7, 9, 1, 11, 6, 12]
    assert right[:7].tolist() == [4, 8, 10, 2, 5, 14, 13]
    assert parent.tolist() == [-1, 3, 3, 0, 0, 4, 5, 1, 1, 2, 2, 4, 6, 6, 5]
    assert (left[7:] == -1).all() and (right[7:] == -1).all()


def test_lbvh_oracle_tree_is_a_threaded_bvh_over_the_same_triangles(orc):
    """Structure of LBVHBuilder's output (LBVHBuilder.cu:353-489, 533-680): following hit links from node 0 visits all
    2 n - 1 nodes once (inner node -> left child, leaf -> the next subtree), every triangle sits in exactly one leaf, a
    parent's box is the union of its children's, equal Morton codes keep their input order, and rays find the same
    hits through this tree as through the SAH tree of the same mesh."""
    from aten_amd.scene import scenedefs
    b, oid, cam = scenedefs.deformable_room(0.7)
    fs = b.build()
    k = fs.blas_index[oid]
    o = fs.arrays["objects"][oid]
    t0, n = int(o["triangle_id"]), int(o["triangle_num"])
    tris = fs.arrays["triangles"][t0:t0 + n]
    vp = fs.arrays["vtx_pos"]
    used = vp[tris["idx"].min():tris["idx"].max() + 1, :3]
    nodes, codes, idx = orc.lbvh_build(tris, used.min(0), used.max(0), vp, tri_id_offset=t0, with_keys=True)
    assert len(nodes) == 2 * n - 1
    assert (np.diff(codes.astype(np.int64)) >= 0).all()
    same = np.flatnonzero(np.diff(codes.astype(np.int64)) == 0)
    assert len(same) > 0 and (idx[same + 1] > idx[same]).all()          # the pole triangles share codes: stable
    assert sorted(idx.tolist()) == list(range(n))
    leaf = nodes["f0"] >= 0
    assert leaf[n - 1:].all() and not leaf[:n - 1].any()
    assert np.array_equal(np.sort(nodes["f1"][leaf].astype(np.int64)), np.arange(t0, t0 + n))
    assert (nodes["hit"][leaf] == nodes["miss"][leaf]).all()
    seen, cur = [], 0
    while cur >= 0:
        seen.append(cur)
        cur = int(nodes["hit"][cur])
    assert sorted(seen) == list(range(2 * n - 1))
    for i in range(n - 1):                  # inner node i: left child = hit link, right child = the left child's miss link
        l = int(nodes["hit"][i]); r = int(nodes["miss"][l])
        assert np.array_equal(nodes["boxmin"][i], np.minimum(nodes["boxmin"][l], nodes["boxmin"][r]))
        assert np.array_equal(nodes["boxmax"][i], np.maximum(nodes["boxmax"][l], nodes["boxmax"][r]))
    c = make_camera(orc, cam, 96, 96)
    rays = orc.generate_paths(c, orc.init_sampler(96, 96, 0), 96, 96, 0, 0)
    a, _ = orc.trace_closest(fs, rays)
    fs.replace_bvh_list(k, nodes)
    bb, _ = orc.trace_closest(fs, rays)
    assert (a["objid"] == len(fs.arrays["objects"]) - 1).sum() > 300    # the blob is in view
    assert a.tobytes() == bb.tobytes()


def test_oracle_toon_materials(orc):
    """Toon / StylizedBrdf in the oracle (material/toon.cpp; HitTeminatedMaterial, pathtracing_impl.h:482-503): a primary hit
    on a Toon surface ends the path with remap(band) x albedo -- exactly the remap texture's bands --, the visibility test
    towards the target light changes bands in shadow, and the screen-space shadow texture scales bands below its threshold."""
    from aten_amd import layout as L
    from aten_amd.scene import scenedefs
    W = H = 64
    fs, cam = scenedefs.toon_room(target="point")
    c = make_camera(orc, cam, W, H)
    seeds = orc.init_sampler(W, H, 0)
    film = orc.render(fs, c, seeds, W, H, 5, 3, frame=0)
    assert np.isfinite(film).all()
    rays = orc.generate_paths(c, seeds, W, H, 0, 0)
    isect, _ = orc.trace_closest(fs, rays)
    names = fs.names["materials"]
    tall = (isect["objid"] >= 0) & (np.array(names)[isect["mtrlid"].clip(0)] == "tallBox")
    assert tall.sum() > 100
    px = film.reshape(-1, 4)[tall]
    bands = np.float32([0.15, 0.45, 0.8, 1.0])
    for ch, alb in enumerate((0.9, 0.5, 0.4)):
        d = np.abs(px[:, ch:ch + 1] - (bands * np.float32(alb))[None, :]).min(axis=1)
        assert (d < 1e-6).all()
    # deeper in the path the same surfaces are plain diffuse / mirror materials: the red wall picks up no banding
    wall = (isect["objid"] >= 0) & (np.array(names)[isect["mtrlid"].clip(0)] == "leftWall")
    assert len(np.unique(np.round(film.reshape(-1, 4)[wall][:, 0], 4))) > 50
    # screen-space shadow: bands under the threshold are scaled by max(s * (v + offset) * scale, s)
    sh = np.full((H, W), 0.25, np.float32)
    fs2, _ = scenedefs.toon_room(target="point", screen_shadow=sh)
    film2 = orc.render(fs2, c, seeds, W, H, 5, 3, frame=0)
    px2 = film2.reshape(-1, 4)[tall]
    assert (px2[:, 0] <= px[:, 0] + 1e-6).all() and (px2[:, 0] < px[:, 0] - 1e-3).mean() > 0.3
    assert np.allclose(film2.reshape(-1, 4)[wall], film.reshape(-1, 4)[wall])


def test_compaction_kat_data():
    """The commented self-test in src/libidaten/kernel/StreamCompaction.cu:318-400 scans
    f = {3,1,7,0,4,1,6,3,...}; the compaction contract on flags>0 is ascending indices."""
    f = np.array([3, 1, 7, 0, 4, 1, 6, 3], np.int32)
    assert np.array_equal(np.flatnonzero(f > 0), [0, 1, 2, 4, 5, 6, 7])
    assert np.array_equal(np.cumsum(f) - f, [0, 3, 4, 11, 11, 15, 16, 22])   # exclusive scan in that file


# ---- oracle vs committed golden vectors ---------------------------------------------------------
def test_cmj_golden(orc, golden):
    from golden.make_golden import CMJ_CASES
    for i, (idx, dim, scr) in enumerate(CMJ_CASES):
        s = orc.cmj_samples(idx, dim, scr, 1024)
        assert np.array_equal(s, golden["cmj_%d" % i])
        assert s.min() >= 0.0 and s.max() < 1.0


def test_cmj_stratification(orc):
    """CMJ property: over the 256 indices of one dimension the x samples hit every 1/256 stratum once."""
    xs = np.array([orc.cmj_samples(i, 3, 0xabcdef01, 1)[0] for i in range(256)])
    assert len(np.unique(np.floor(xs * 256).astype(int))) == 256


def test_ray_offset_golden(orc, golden):
    out = orc.ray_offset(golden["offset_o"], golden["offset_n"])
    assert np.array_equal(out, golden["offset_out"])


def test_generate_paths_golden(orc, cornell, golden):
    fs, cam = cornell
    c = make_camera(orc, cam, 64, 64)
    seeds = orc.init_sampler(64, 64, 0)
    for frame in (0, 1, 7):
        rays = orc.generate_paths(c, seeds, 64, 64, 0, frame)
        assert rays.tobytes() == golden["rays_cornell64_f%d" % frame].tobytes()
    # sample i of frame f == sample i+1 of frame f-1 (stream depends on frame+sample only)
    a = orc.generate_paths(c, seeds, 64, 64, 1, 6)
    assert a.tobytes() == golden["rays_cornell64_f7"].tobytes()


def test_trace_golden(orc, cornell, sponza, golden):
    fs, cam = cornell
    isect, st = orc.trace_closest(fs, golden["rays_cornell64_f0"])
    assert isect.tobytes() == golden["isect_cornell64"].tobytes()
    assert np.array_equal(st, golden["isect_cornell64_stats"])
    fs2, cam2 = sponza
    c2 = make_camera(orc, cam2, 128, 72)
    rays2 = orc.generate_paths(c2, orc.init_sampler(128, 72, 0), 128, 72, 0, 0)
    isect2, st2 = orc.trace_closest(fs2, rays2)
    assert isect2.tobytes() == golden["isect_sponza128x72"].tobytes()


def test_render_golden(orc, cornell, golden):
    fs, cam = cornell
    c = make_camera(orc, cam, 64, 64)
    seeds = orc.init_sampler(64, 64, 0)
    for depth in (3, 5):
        film = np.zeros((64, 64, 4), np.float32)
        for frame in range(4):
            orc.render(fs, c, seeds, 64, 64, depth, 3, frame=frame, film=film)
        assert np.array_equal(film, golden["film_cornell64_d%d_f0to3" % depth])
        assert np.all(film[..., 3] == 4.0)


def test_render_thread_independence(orc, cornell):
    """Pixel results do not depend on the OpenMP thread count (per-pixel state only)."""
    fs, cam = cornell
    c = make_camera(orc, cam, 48, 32)
    seeds = orc.init_sampler(48, 32, 0)
    a = orc.render(fs, c, seeds, 48, 32, 5, 3, nthreads=1)
    b = orc.render(fs, c, seeds, 48, 32, 5, 3, nthreads=8)
    assert np.array_equal(a, b)


def test_closest_hit_is_topology_independent(orc):
    """The same triangles under our SAH tree and under the reference-built .sbvh tree give the same
    closest hits (ids may differ only on exact-t ties, which SBVH reference duplication makes harmless)."""
    from aten_amd.scene import scenedefs
    a, cam = scenedefs.sponza_lod(use_sbvh=True, textures=False, ibl=False)
    b, _ = scenedefs.sponza_lod(use_sbvh=False, textures=False, ibl=False)
    c = make_camera(orc, cam, 96, 54)
    rays = orc.generate_paths(c, orc.init_sampler(96, 54, 0), 96, 54, 0, 0)
    ia, sa = orc.trace_closest(a, rays)
    ib, sb = orc.trace_closest(b, rays)
    assert np.array_equal(ia["t"], ib["t"])
    same = ia["tri_id"] == ib["tri_id"]
    assert same.mean() > 0.999


def test_brute_force_agrees_with_traversal(orc, cornell):
    """Every Cornell triangle tested directly (numpy Moeller-Trumbore in fp32, same op order) gives the
    traversal's closest t: culling never loses the nearest hit."""
    fs, cam = cornell
    c = make_camera(orc, cam, 32, 32)
    rays = orc.generate_paths(c, orc.init_sampler(32, 32, 0), 32, 32, 0, 0)
    isect, _ = orc.trace_closest(fs, rays)
    pos = fs.arrays["vtx_pos"][:, :3]; tris = fs.arrays["triangles"]["idx"]
    o = rays["org"].astype(np.float32); d = rays["dir"].astype(np.float32)
    best = np.full(len(rays), np.finfo(np.float32).max, np.float32)
    f = np.float32
    for t in tris:
        v0, v1, v2 = pos[t[0]], pos[t[1]], pos[t[2]]
        e1, e2 = (v1 - v0).astype(f), (v2 - v0).astype(f)
        r = (o - v0).astype(f)
        u = np.cross(d, e2).astype(f); v = np.cross(r, e1).astype(f)
        with np.errstate(all="ignore"):
            inv = f(1) / (u @ e1).astype(f)
            tt = ((v @ e2) * inv).astype(f); be = (np.einsum("ij,ij->i", u, r) * inv).astype(f); ga = (np.einsum("ij,ij->i", v, d) * inv).astype(f)
        ok = (be >= 0) & (be <= 1) & (ga >= 0) & (ga <= 1) & (be + ga <= 1) & (tt >= 0) & (tt > 1e-9)
        best = np.where(ok & (tt < best), tt, best)
    hit = isect["objid"] >= 0
    # numpy's dot/cross may associate differently in the last ulp; the decision (which t) must agree closely
    assert np.allclose(isect["t"][hit], best[hit], rtol=1e-5)
    assert np.all(best[~hit] == np.finfo(np.float32).max)


# ---- host logic -------------------------------------------------------------------------------
def test_layout_sizes_match_c_header():
    from aten_amd import layout as L
    from aten_amd._lib import Destination, lib
    l = lib()
    assert l.atn_sizeof_scene_desc() == C.sizeof(L.SceneDesc)
    assert l.atn_sizeof_destination() == C.sizeof(Destination)


def test_headers_compile_as_c99_and_cxx17(tmp_path):
    """include/*.h is the drop-in boundary: plain C (a cgo / ctypes / C caller) and C++ (the aten application) both
    compile it, warnings as errors, and the static layout checks hold."""
    import subprocess
    src = ('#include "aten_amd.h"\n#include "aten_amd_scene.h"\n'
           'int main(void) { atn_scene_desc d; atn_destination t; (void)d; (void)t; return (int)sizeof(atn_toon_param) - 100; }\n')
    for name, cmd in (("h.c", ["gcc", "-std=c99", "-pedantic"]), ("h.cpp", ["g++", "-std=c++17"])):
        f = tmp_path / name
        f.write_text(src)
        subprocess.check_call(cmd + ["-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(f), "-o", str(f) + ".o"])


def test_cxx_application_builds_and_fails_loudly_without_gpu(tmp_path):
    """tests/cxx/aten_app.cpp (a C++ program on the C-ABI only) compiles warning-free against include/ and the in-tree
    libraries; without a GPU atn_create must return an error code and a message -- there is no CPU fallback to fall to."""
    import subprocess
    import torch
    exe = str(tmp_path / "aten_app")
    lib = os.path.join(ROOT, "aten_amd")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cxx", "aten_app.cpp"), "-L", lib, "-laten_amd", "-laten_amd_scene",
                           "-Wl,-rpath," + lib, "-o", exe])
    if torch.cuda.is_available():
        pytest.skip("GPU present: tests/test_gpu_cxx_app.py runs it")
    res = subprocess.run([exe, str(tmp_path), "32", "32", "1"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 1
    assert "atn_create" in res.stderr and "-2" in res.stderr          # ATN_ERR_NO_DEVICE


def test_c_abi_exports_every_declared_symbol():
    """Every function declared in include/aten_amd.h / aten_amd_scene.h is exported (no compute calls)."""
    import re
    from aten_amd._hostlib import hostlib
    from aten_amd._lib import lib
    for header, l in (("aten_amd.h", lib()), ("aten_amd_scene.h", hostlib())):
        src = open(os.path.join(ROOT, "include", header)).read()
        names = set(re.findall(r"\b(atns?_[a-z_0-9]+)\s*\(", src))
        names = {n for n in names if not n.startswith("atn_status")}
        assert names, header
        for n in sorted(names):
            assert hasattr(l, n), "%s: %s not exported" % (header, n)


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from aten_amd.renderer import AtenAmdError, PathTracing
    with pytest.raises(AtenAmdError):
        PathTracing(0)


def test_product_path_never_imports_oracle():
    """Nothing under aten_amd/ may load, link or call the oracle; build.py only holds its build recipe."""
    for r, _, fs in os.walk(os.path.join(ROOT, "aten_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) and f != "build.py":
                s = open(os.path.join(r, f), errors="replace").read().lower()
                assert "oracle" not in s and "orc_" not in s, os.path.join(r, f)
    b = open(os.path.join(ROOT, "aten_amd", "build.py")).read()
    assert "CDLL" not in b and "import orc" not in b


def test_cornell_scene_invariants(cornell):
    """Flat-array invariants an aten app guarantees (SURVEY 8(b) worked example for ObjCornellBoxScene)."""
    fs, _ = cornell
    a = fs.arrays
    objs = a["objects"]
    assert len(objs) == 16 and len(a["triangles"]) == 32 and len(a["vtx_pos"]) == 96
    assert fs.names["materials"][0] == "light"
    assert list(objs["type"][:8]) == [0] * 8 and list(objs["type"][8:]) == [1] * 8
    assert objs["light_id"][0] == 0 and objs["light_id"][8] == 0 and objs["object_id"][8] == 0
    assert a["lights"]["arealight_objid"][0] == 8
    assert np.all(a["triangles"]["needNormal"] == 1)
    tl = a["bvh_lists"][0]
    leaf = tl["f0"] >= 0
    assert sorted(tl["f0"][leaf].astype(int)) == list(range(8, 16))
    assert sorted((tl["f2"][leaf].view(np.uint32) & 0x7fff).tolist()) == list(range(1, 9))
    # light area: 0.47 x 0.38 quad
    assert abs(float(objs["area"][0]) - 0.47 * 0.38) < 1e-6


def test_bvh_builder_contract(cornell):
    from aten_amd._hostlib import hostlib
    fs, _ = cornell
    for nodes in fs.arrays["bvh_lists"][1:]:
        n = len(nodes)
        leaf = nodes["f0"] >= 0
        assert hostlib().atns_validate_nodes(nodes.ctypes.data, n) == leaf.sum() == (n + 1) // 2
        inner = ~leaf
        assert np.all(nodes["hit"][inner] == np.arange(n)[inner] + 1)       # pre-order
        assert np.all(nodes["hit"][leaf] == nodes["miss"][leaf])
        assert np.all(nodes["f2"][leaf] == -1.0)


def test_product_camera_block_equals_oracle(orc):
    """aten_amd.scene.camera.create_camera (host library, csrc/host/camera.cpp) against the oracle's restatement of
    PinholeCamera::CreateCameraParam (camera/pinhole.cpp:34-75): byte-equal blocks, so bench / smoke / scene code
    need nothing from oracle/.  The oracle side is pinned by the reference's camera_test known answer."""
    from aten_amd.scene import scenedefs
    from aten_amd.scene.camera import create_camera
    cams = [scenedefs.cornell_box()[1], scenedefs.sponza_lod()[1],
            dict(pos=(1.5, -2.25, 7.0), at=(-0.3, 0.8, 0.1), vfov=33.3),
            dict(pos=(0, 10, 0.001), at=(0, 0, 0), vfov=90.0)]
    for cam in cams:
        for w, h in ((1280, 720), (1920, 1080), (3840, 2160), (100, 52), (1, 1)):
            a = create_camera(cam["pos"], cam["at"], cam["vfov"], w, h)
            b = orc.create_camera(cam["pos"], cam["at"], cam["vfov"], w, h)
            assert a.tobytes() == b.tobytes(), (cam, w, h)
    a = create_camera((0, 0, 1), (0, 0, 0), 60.0, 640, 480, up=(0, 0, 1) if False else (0, 1, 0), znear=5.0, zfar=0.5)
    assert float(a["znear"]) == 0.5 and float(a["zfar"]) == 5.0         # min / max swap of pinhole.cpp:70-71
    with pytest.raises(ValueError):
        create_camera((0, 0, 1), (0, 0, 0), 60.0, 0, 480)
