"""Native scene ingestion (csrc/host/obj_ingest.cpp behind include/aten_amd_scene.h) against its Python twin
(aten_amd/scene/obj_loader.py + SceneBuilder.load_obj): aten::ObjLoader's registration rules
(src/libatenscene/ObjLoader.cpp:95-461) on the reference's own OBJ assets and on a synthetic file that exercises every
rule; the material XML of aten::MaterialLoader (src/libatenscene/MaterialLoader.cpp:82-218)."""
import os

import numpy as np
import pytest

from aten_amd import layout as L
from aten_amd.scene.builder import SceneBuilder

OBJ = """# every ObjLoader rule in one file
mtllib t.mtl
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0 0 1
v 1 0 1
v 1 1 1
v 0 1 1
vt 0 0
vt 1 0
vt 1 1
vt 0.25
vn 0 0 1
vn nan 0 0
vn 0 1 0
f 1 2 3
o first
usemtl red
f 1/1/1 2/2/1 3/3/1 4/4/1
f -4//3 -3//3 -2//3
usemtl lamp
f 5/1 6/2 7/3
usemtl red
f 1/1/2 2/2/2 3/3/2
o empty_shape
g second group
usemtl lamp
f 5 6 7 8 4
usemtl blue
f 1/1/1 6/2/1 7/3/1
o third
f 2/1/1 3/2/1 8/3/1
usemtl lamp
f 2 3 8
"""
MTL = """newmtl red
Kd 0.8 0.1 0.1
map_Kd missing_red.png
newmtl lamp
Kd 1 1 1
Ke 5 5 5
newmtl blue
Kd 0.1 0.1 0.9
map_bump missing_bump.png
"""


def _create(b):
    def create(name, mtype, color, albedo, nmap):
        if name == "lamp":
            return b.add_material(name, L.MTRL_EMISSIVE, color)
        return b.add_material(name, mtype, color)
    return create


def _snapshot(b, objs):
    return dict(pos=np.asarray(b.pos, np.float32), nml=np.asarray(b.nml, np.float32),
                tris=[(t["idx"], t["needNormal"], t["mtrlid"], t["mesh_id"]) for t in b.tris],
                objects=[(o["name"], [(m["mtrl"], m["mesh_id"], tuple(m["tris"])) for m in o["meshes"]]) for o in b.objects],
                materials=[m[0] for m in b.materials], objs=list(objs), mesh_counter=b.mesh_counter)


@pytest.mark.parametrize("separate", [False, True])
@pytest.mark.parametrize("on_the_fly", [False, True])
def test_native_obj_registration_equals_python_twin(tmp_path, separate, on_the_fly):
    (tmp_path / "t.obj").write_text(OBJ)
    (tmp_path / "t.mtl").write_text(MTL)
    snaps = []
    for native in (False, True):
        b = SceneBuilder()
        b.add_material("preexisting", L.MTRL_DIFFUSE, (1, 1, 1))
        b.add_mesh("quad", np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), np.array([[0, 1, 2]]), 0)   # offsets are not 0
        objs = b.load_obj(str(tmp_path / "t.obj"), create_mtrl=_create(b), separate_objs=separate,
                          normal_on_the_fly=on_the_fly, native=native)
        snaps.append(_snapshot(b, objs))
    py, nat = snaps
    assert np.array_equal(py["pos"], nat["pos"], equal_nan=True) and py["pos"].tobytes() == nat["pos"].tobytes()
    assert py["nml"].tobytes() == nat["nml"].tobytes()
    for k in ("tris", "objects", "materials", "objs", "mesh_counter"):
        assert py[k] == nat[k], k
    # and the rules themselves, on the native result
    names = [o[0] for o in nat["objects"]]
    assert "empty_shape" not in names                                   # shapes without faces are dropped
    assert nat["materials"][:1] == ["preexisting"] and "" in nat["materials"]       # faces before any usemtl: the callback's "" material
    n_tri = 1 + 2 + 1 + 1 + 1 + 3 + 1 + 1 + 1
    assert len(nat["tris"]) == 1 + n_tri and len(nat["pos"]) == 3 + 3 * n_tri        # one vertex per face corner
    need = [t[1] for t in nat["tris"][1:]]
    if on_the_fly:
        assert all(need)
    else:
        # v v v: the "no texcoord" flag -1 overwrites the "no normal" flag 1 (the reference's own TODO, ObjLoader.cpp:186-188);
        # v/vt/vn: 0; v//vn: 0; v/vt (no normal, has uv): 1
        assert need[0] == 0 and need[1] == 0 and need[3] == 0 and need[4] == 1
    assert (nat["nml"][3 + 3 * 5: 3 + 3 * 6, :3] == [0, 1, 0]).all()    # NaN normal -> (0, 1, 0)
    if not separate:
        lamp = nat["materials"].index("lamp")
        split = [o for o in nat["objects"] if len(o[1]) == 1 and o[1][0][0] == lamp]
        assert len(split) == 3                                          # every emissive group became its own object
        assert nat["objs"][-1] == min(i for i, o in enumerate(nat["objects"]) if any(m[0] != lamp for m in o[1]) and o[0] != "quad")


def test_native_obj_on_reference_assets():
    from aten_amd.scene import scenedefs
    for make in (scenedefs.cornell_box, scenedefs.sponza_lod):
        os.environ["ATEN_AMD_PY_OBJ"] = "1"
        try:
            a, _ = make()
        finally:
            os.environ["ATEN_AMD_PY_OBJ"] = "0"
        b, _ = make()
        for k in a.arrays:
            x, y = a.arrays[k], b.arrays[k]
            if isinstance(x, list):
                assert len(x) == len(y) and all(np.asarray(p).tobytes() == np.asarray(q).tobytes() for p, q in zip(x, y)), k
            else:
                assert np.asarray(x).tobytes() == np.asarray(y).tobytes(), k


def test_native_obj_errors(tmp_path):
    from aten_amd.scene import native_obj
    with pytest.raises(IOError):
        native_obj.ObjFile(str(tmp_path / "nope.obj"))
    (tmp_path / "bad.obj").write_text("v 0 0 0\nf 1 2 3\n")
    with pytest.raises(IOError):
        native_obj.ObjFile(str(tmp_path / "bad.obj"))               # index out of range is refused, not read


def test_second_mtllib_appends_and_array_sizes_are_checked(tmp_path):
    """a second `mtllib` APPENDS to the material list (tinyobj's reader does; clearing it left earlier faces with
    ids into the old list and atns_obj_register reading mtl_is_emissive out of bounds); atns_obj_register / atns_obj_copy
    take the sizes of the caller's arrays and refuse short ones."""
    from aten_amd.scene import native_obj
    (tmp_path / "a.mtl").write_text("newmtl red\nKd 1 0 0\nnewmtl lamp\nKe 5 5 5\n")
    (tmp_path / "b.mtl").write_text("newmtl blue\nKd 0 0 1\n")
    (tmp_path / "two.obj").write_text("mtllib a.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nusemtl lamp\nf 1 2 3\nmtllib b.mtl\n"
                                      "usemtl blue\nf 1 3 4\nusemtl red\nf 1 2 4\n")
    o = native_obj.ObjFile(str(tmp_path / "two.obj"))
    assert [m["name"] for m in o.materials] == ["red", "lamp", "blue"]
    pos, nml, tris, meshes, objs = o.register(0, 0, False, False, mtl_is_emissive=[0, 1, 0])
    assert [int(m["mtl"]) for m in meshes] == [1, 2, 0]             # lamp (id from the first library), blue, red
    assert int(objs[meshes[0]["object"]]["is_emissive_split"]) == 1 and len(tris) == 3
    # short arrays are an error code, not an out-of-bounds access
    l, h = o._l, o._h
    em = np.zeros(2, np.uint8)
    assert l.atns_obj_register(h, 0, 0, 0, 0, em.ctypes.data, 2, 0) == -4
    assert l.atns_obj_register(h, 0, 0, 0, 0, None, 0, 0) == 0       # no array: nothing is emissive
    small = np.zeros((1, 4), np.float32)
    assert l.atns_obj_copy(h, small.ctypes.data, None, 1, None, 0, None, 0, None, 0) == -4
    assert np.all(small == 0)


XML = """<?xml version="1.0" encoding="UTF-8"?>
<!-- the format of asset/converted_unitychan/unitychan_mtrl.xml -->
<root>
    <material>
        <name>face &amp; hair</name>
        <type>diffuse</type>
        <albedoMap>face_00.tga</albedoMap>
        <baseColor>1.0 0.5 0.25</baseColor>
        <somethingElse>ignored</somethingElse>
    </material>
    <material>
        <name>metal</name>
        <type>ggx</type>
        <baseColor>0.7 0.6</baseColor>
        <ior>1.5</ior>
        <roughness>2.5e-1</roughness>
        <normalMap>n.png</normalMap>
        <roughnessMap>r.png</roughnessMap>
        <empty/>
    </material>
    <material>
        <name>metal</name>
        <type>specular</type>
    </material>
    <material>
        <name>untyped</name>
        <clearcoatGloss>0.9</clearcoatGloss>
    </material>
</root>
"""


def test_material_xml(tmp_path):
    from aten_amd.scene import native_obj
    p = tmp_path / "m.xml"
    p.write_text(XML)
    mats = native_obj.load_material_xml(str(p))
    assert [(m[0], m[1]) for m in mats] == [("face & hair", "diffuse"), ("metal", "ggx"), ("untyped", "Diffuse")]   # duplicate dropped
    p0 = {n: (k, v, t) for n, k, v, t in mats[0][2]}
    assert p0["albedoMap"][0] == 1 and p0["albedoMap"][2] == "face_00.tga"
    assert p0["baseColor"][0] == 0 and p0["baseColor"][1] == (1.0, 0.5, 0.25)
    assert p0["somethingElse"][0] == -1                                 # not in MaterialLoader's table: the caller skips it
    p1 = {n: (k, v, t) for n, k, v, t in mats[1][2]}
    assert p1["baseColor"][1] == (np.float32(0.7), np.float32(0.6), 0.0)        # fewer than three values: the rest stay 0
    assert p1["ior"] == (2, (1.5, 0.0, 0.0), "1.5") and p1["roughness"][1][0] == np.float32(0.25)
    assert p1["normalMap"][0] == 1 and p1["roughnessMap"][2] == "r.png" and "empty" not in p1
    assert mats[2][2][0][0] == "clearcoatGloss" and mats[2][2][0][1] == 2
    (tmp_path / "noroot.xml").write_text("<materials><material><name>x</name></material></materials>")
    with pytest.raises(IOError):
        native_obj.load_material_xml(str(tmp_path / "noroot.xml"))
    with pytest.raises(IOError):
        native_obj.load_material_xml(str(tmp_path / "missing.xml"))
    ref = "/root/reference/asset/converted_unitychan/unitychan_mtrl.xml"
    if os.path.exists(ref):                                             # the one XML the reference ships (build container only)
        mats = native_obj.load_material_xml(ref)
        assert len(mats) == 9 and mats[0][0] == "face" and mats[0][1] == "diffuse"
        assert all(k in (0, 1, 2) for m in mats for _, k, _, _ in m[2])
