"""Host scene assembly: what aten::context + ObjLoader + AcceleratedScene<sbvh>::build do,
producing the flat arrays of include/aten_layout.h.

Reference behaviour followed (paths relative to /root/reference/src):
  * vertex / triangle registration order ........ libatenscene/ObjLoader.cpp:95-461
  * triangle area, object area, ids ............. libaten/geometry/triangle.cpp:108-134,
                                                  TriangleGroupMesh.cpp:10-35, PolygonObject.cpp:14-55
  * object array = creation order, instances are entries too
                                                  libaten/scene/host_scene_context.cpp:258-263
  * every instance owns an (L2W, W2L) matrix pair  libaten/geometry/instance.h:253-269
  * light_id on both instance and real object .... host_scene_context.cpp:398-427
  * texture storage (flipped, *1/255, a=1) ....... libaten/image/image.cpp:40-88, image/texture.h:93-102
  * .sbvh import ................................. libaten/accelerator/sbvh.cpp:1345-1436,109-128

BVH topology comes from our own builder (aten_amd/csrc/host/bvh_builder.cpp) unless a
reference-built .sbvh is imported.
"""
import ctypes as C
import os
import struct

import numpy as np

from .. import layout as L
from .._hostlib import hostlib, default_bvh_options, BvhStats
from . import obj_loader

F32 = np.float32

# atns_bvh_options fields every SceneBuilder without its own `bvh_options` passes to atns_build_blas_opt ({} = the library's
# defaults); tools/tree_quality.py sweeps through this.
DEFAULT_BVH_OPTIONS = {}


def _length3(x, y, z):
    return np.sqrt((x * x + y * y) + z * z, dtype=F32)


class FlatScene:
    """Owns the numpy arrays behind an atn_scene_desc."""

    def __init__(self):
        self.desc = L.SceneDesc()
        self.keep = []
        self.names = {}

    def ref(self):
        return C.byref(self.desc)

    @staticmethod
    def from_arrays(objects, matrices, materials, lights, triangles, vtx_pos, vtx_nml, bvh_lists, config=None,
                    scene_bbox=None, textures=()):
        """Wrap flat arrays somebody else produced (e.g. a C++ application's dump) in an atn_scene_desc."""
        fs = FlatScene()
        objs = np.ascontiguousarray(objects, L.OBJECT_PARAM); mats = np.ascontiguousarray(materials, L.MATERIAL_PARAM)
        lts = np.ascontiguousarray(lights, L.LIGHT_PARAM); tris = np.ascontiguousarray(triangles, L.TRIANGLE_PARAM)
        mtx = np.ascontiguousarray(matrices, F32).reshape(-1, 4, 4)
        pos = np.ascontiguousarray(vtx_pos, F32).reshape(-1, 4); nml = np.ascontiguousarray(vtx_nml, F32).reshape(-1, 4)
        nodes = [np.ascontiguousarray(n, L.BVH_NODE) for n in bvh_lists]
        lists = (L.BvhList * len(nodes))()
        for i, n in enumerate(nodes):
            lists[i].nodes = n.ctypes.data
            lists[i].count = len(n)
        tex = [np.ascontiguousarray(t, F32) for t in textures]
        texd = (L.TextureDesc * max(1, len(tex)))()
        for i, t in enumerate(tex):
            texd[i].texels = t.ctypes.data
            texd[i].height, texd[i].width = t.shape[0], t.shape[1]
        d = fs.desc
        d.objects, d.n_objects = L.ptr(objs), len(objs)
        d.matrices, d.n_matrices = L.ptr(mtx), len(mtx)
        d.materials, d.n_materials = L.ptr(mats), len(mats)
        d.lights, d.n_lights = L.ptr(lts), len(lts)
        d.triangles, d.n_triangles = L.ptr(tris), len(tris)
        d.vtx_pos, d.vtx_nml, d.n_vertices = L.ptr(pos), L.ptr(nml), len(pos)
        d.bvh_lists, d.n_bvh_lists = C.addressof(lists), len(nodes)
        d.textures, d.n_textures = C.addressof(texd), len(tex)
        if config is not None:
            d.config = config
        if scene_bbox is not None:
            d.scene_bbox_min[:] = [float(x) for x in scene_bbox[0]]
            d.scene_bbox_max[:] = [float(x) for x in scene_bbox[1]]
        d.enable_shadowray_base_stylized_shadow = 1
        fs.keep = [objs, mtx, mats, lts, tris, pos, nml, nodes, lists, texd, tex]
        fs.lists = lists
        fs.arrays = dict(objects=objs, matrices=mtx, materials=mats, lights=lts, triangles=tris, vtx_pos=pos, vtx_nml=nml,
                         bvh_lists=nodes, textures=tex)
        return fs

    def replace_bvh_list(self, k, nodes):
        """Swap node list k (e.g. for a tree built elsewhere over the same triangles)."""
        nodes = np.ascontiguousarray(nodes, L.BVH_NODE)
        self.arrays["bvh_lists"][k] = nodes
        self.lists[k].nodes = nodes.ctypes.data
        self.lists[k].count = len(nodes)


class SceneBuilder:
    def __init__(self):
        self.pos = []           # (x,y,z,u)
        self.nml = []           # (x,y,z,v)
        self.tris = []          # dict rows
        self.materials = []     # (name, np record)
        self.textures = []      # (name, array[h,w,4])
        self.objects = []       # dict: type, ...
        self.matrices = []      # 4x4 f32
        self.lights = []
        self.npr_lights = []
        self.screen_space_texture = None    # float32 [h, w, 4]: context::screen_space_texture
        self.enable_shadowray_base_stylized_shadow = True
        self.blas = {}          # polygon object id -> node array (or None = build)
        self.bvh_options = None # dict of atns_bvh_options fields for atns_build_blas_opt (None = DEFAULT_BVH_OPTIONS / the library's)
        self.mesh_counter = 0
        self.config = L.SceneRenderingConfig()
        self.config.bvh_hit_min = -1.0
        self.config.epsilon_bias = 1e-3
        self.config.bg.envmap_tex_idx = -1
        self.config.bg.avgIllum = 1.0
        self.config.bg.multiplyer = 1.0
        self.config.bg.enable_env_map = 1

    # ---------------------------------------------------------------- materials / textures
    def add_material(self, name, mtype, base_color, albedo_map=-1, normal_map=-1, roughness_map=-1, **std):
        m = np.zeros((), L.MATERIAL_PARAM)
        bc = list(base_color)
        m["baseColor"] = (bc + [0.0])[:4] if len(bc) == 3 else bc      # vec4 = vec3 -> w = 0 (vec4.h:135-141 keeps w)
        if len(bc) == 3:
            m["baseColor"][3] = 1.0     # MaterialParameter() sets (0,0,0,1); operator=(vec3) keeps w
        m["type"] = mtype
        m["attrib"] = L.MTRL_ATTRIB.get(mtype, 0)
        m["id"] = len(self.materials)
        m["albedoMap"], m["normalMap"], m["roughnessMap"] = albedo_map, normal_map, roughness_map
        s = dict(ior=1.0, roughness=0.5, shininess=1.0, subsurface=0.5, metallic=0.5, specular=0.5,
                 specularTint=0.5, anisotropic=0.5, sheen=0.5, sheenTint=0.5, clearcoat=0.5, clearcoatGloss=0.5)
        s.update(std)
        m["standard"] = [s[k] for k in L.STANDARD_FIELDS]
        m["medium"][3] = np.int32(-1).view(F32)     # MediumParameter.grid_idx = -1
        m["medium"][4] = -1.0                       # majorant
        self.materials.append((name, m))
        return len(self.materials) - 1

    def add_carpaint_material(self, name, base_color=(1.0, 1.0, 1.0), albedo_map=-1, **cp):
        """aten::CarPaint: CarPaintMaterialParameter (material.h:163-198, defaults of Init()) in the union next to `standard`."""
        mid = self.add_material(name, L.MTRL_CARPAINT, base_color, albedo_map=albedo_map)
        p = dict(clearcoat_color=(1.0, 1.0, 1.0), clearcoat_ior=3.0, flakes_color=(1.0, 1.0, 0.0), clearcoat_roughness=0.25,
                 diffuse_color=(1.0, 0.0, 1.0), flake_scale=400.0, flake_size=0.25, flake_size_variance=0.7,
                 flake_normal_orientation=0.5, flake_color_multiplier=1.0)
        p.update(cp)
        v = (list(p["clearcoat_color"]) + [p["clearcoat_ior"]] + list(p["flakes_color"]) + [p["clearcoat_roughness"]]
             + list(p["diffuse_color"]) + [p["flake_scale"], p["flake_size"], p["flake_size_variance"],
                                           p["flake_normal_orientation"], p["flake_color_multiplier"]])
        m = self.materials[mid][1]
        m["standard"] = v[:12]
        m["_union_tail"] = v[12:]
        return mid

    def add_toon_material(self, name, base_color, stylized=False, toon_type=None, target_light_idx=-1, remap_texture=-1,
                          albedo_map=-1, normal_map=-1, **p):
        """aten::Toon / aten::StylizedBrdf (material/toon.h:20-34,80-90): ToonParameter with the defaults of material.h:124-161,
        attrib = Diffuse's or Microfacet's by toon_type.  `p`: roughness / ior (standard part) and any TOON_PARAM field."""
        toon_type = L.MTRL_DIFFUSE if toon_type is None else toon_type
        std = {k: p.pop(k) for k in list(p) if k in L.STANDARD_FIELDS}
        mid = self.add_material(name, L.MTRL_STYLIZED if stylized else L.MTRL_TOON, base_color, albedo_map=albedo_map,
                                normal_map=normal_map, **std)
        m = self.materials[mid][1]
        m["attrib"] = 0 if toon_type == L.MTRL_DIFFUSE else L.ATTR_GLOSSY
        t = m["toon"]
        t["target_light_idx"], t["remap_texture"] = target_light_idx, remap_texture
        t["stylized_y_min"], t["stylized_y_max"] = 0.0, 1.0
        t["toon_type"], t["will_receive_shadow"] = toon_type, 1
        t["shadow_threshold"], t["shadow_offset"], t["shadow_scale"] = 0.5, 0.05, 10.0
        for k, v in p.items():
            t[k] = v
        return mid

    def add_npr_target_light(self, light):
        """context::AddNprTargetLight: a LightParameter toon materials aim at (ToonParameter::target_light_idx)."""
        self.npr_lights.append(np.array(light, L.LIGHT_PARAM))
        return len(self.npr_lights) - 1

    def find_material(self, name):
        for i, (n, _) in enumerate(self.materials):
            if n == name:
                return i
        return -1

    def add_texture(self, name, rgba):
        """rgba: float32 [h, w, 4] already in aten's storage order (row 0 = image bottom)."""
        for i, (n, _) in enumerate(self.textures):
            if n == name:
                return i
        self.textures.append((name, np.ascontiguousarray(rgba, F32)))
        return len(self.textures) - 1

    def load_image(self, path):
        """aten::Image::Load for LDR images: bytes * (1/255), vertical flip, missing alpha = 1."""
        from PIL import Image
        real = _resolve_case(path)
        if real is None:
            return -1
        tag = os.path.basename(path)
        for i, (n, _) in enumerate(self.textures):
            if n == tag:
                return i
        img = Image.open(real)
        if img.mode not in ("RGB", "RGBA", "L"):
            img = img.convert("RGB")
        a = np.asarray(img)
        if a.ndim == 2:
            a = a[:, :, None]
        h, w, ch = a.shape
        out = np.zeros((h, w, 4), F32)
        out[:, :, 3] = 1.0
        norm = F32(1.0) / F32(255)
        out[:, :, :min(ch, 4)] = a[:, :, :4].astype(F32) * norm
        return self.add_texture(tag, out[::-1])

    # ---------------------------------------------------------------- geometry
    def load_obj(self, path, create_mtrl=None, separate_objs=False, normal_on_the_fly=False, native=None):
        """ObjLoader::Load.  Returns the list of created PolygonObject ids.  The parsing and aten::ObjLoader's registration
        rules run natively (csrc/host/obj_ingest.cpp through include/aten_amd_scene.h); `native=False` (or
        ATEN_AMD_PY_OBJ=1) takes the Python twin below, kept as the cross-check (tests compare the two byte for byte)."""
        if native is None:
            native = os.environ.get("ATEN_AMD_PY_OBJ", "0") != "1"
        if native:
            return self._load_obj_native(path, create_mtrl, separate_objs, normal_on_the_fly)
        P, T, N, shapes, mtls = obj_loader.load_obj(path)
        base = os.path.dirname(path)
        objs = []
        cur_obj = None

        def new_obj(name):
            self.objects.append(dict(type=L.OBJ_POLYGONS, name=name, meshes=[], first_tri=None))
            return len(self.objects) - 1

        for si, sh in enumerate(shapes):
            ntri = len(sh.material_ids)
            base_v = len(self.pos)
            flags = []
            for (v, vt, vn) in sh.corners:
                px, py, pz = P[v]
                uvz = 0.0
                if vn < 0:
                    nx, ny, nz = 0.0, 1.0, 0.0      # reference leaves nml unset; needNormal is set
                    uvz = 1.0
                else:
                    nx, ny, nz = N[vn]
                    uvz = 1.0 if normal_on_the_fly else 0.0
                if np.isnan(nx) or np.isnan(ny) or np.isnan(nz):
                    nx, ny, nz = 0.0, 1.0, 0.0
                if vt >= 0:
                    u, w = T[vt]
                else:
                    u, w = 0.0, 0.0
                    uvz = -1.0
                self.pos.append((px, py, pz, u))
                self.nml.append((nx, ny, nz, w))
                flags.append(uvz)

            mesh = None
            prev = None
            for i in range(ntri):
                mid = sh.material_ids[i]
                if mesh is None or prev != mid:
                    if mesh is not None:
                        cur_obj = self._register_mesh(mesh, cur_obj, sh.name, objs, new_obj, False)
                    mesh = dict(tris=[], mtrl=self._resolve_material(mid, mtls, base, create_mtrl),
                                mesh_id=self._next_mesh_id())
                    prev = mid
                i0, i1, i2 = base_v + 3 * i, base_v + 3 * i + 1, base_v + 3 * i + 2
                need = 1 if (flags[3 * i] == 1.0 or flags[3 * i + 1] == 1.0 or flags[3 * i + 2] == 1.0
                             or normal_on_the_fly) else 0
                self.tris.append(dict(idx=(i0, i1, i2), needNormal=need, mtrlid=mesh["mtrl"], mesh_id=mesh["mesh_id"]))
                mesh["tris"].append(len(self.tris) - 1)

            if separate_objs:
                if cur_obj is None:
                    cur_obj = new_obj(sh.name)
                self.objects[cur_obj]["meshes"].append(mesh)
                self.objects[cur_obj]["name"] = sh.name
                objs.append(cur_obj)
                cur_obj = new_obj("") if si + 1 < len(shapes) else None
            else:
                cur_obj = self._register_mesh(mesh, cur_obj, sh.name, objs, new_obj, True)

        if not separate_objs and cur_obj is not None:
            objs.append(cur_obj)
        for o in objs:
            self.blas.setdefault(o, None)
        return objs

    def _load_obj_native(self, path, create_mtrl, separate_objs, normal_on_the_fly):
        from . import native_obj
        f = native_obj.ObjFile(path)
        try:
            base = os.path.dirname(path)
            mtls = [obj_loader.ObjMaterial(m["name"]) for m in f.materials]
            for o, m in zip(mtls, f.materials):
                o.diffuse, o.emission = m["diffuse"], m["emission"]
                o.diffuse_texname, o.bump_texname = m["diffuse_texname"], m["bump_texname"]
            # pass 1: the groups in creation order (they do not depend on the object partition): materials are created /
            # found in exactly that order, like ObjLoader's callbacks
            _, _, _, meshes, _ = f.register(len(self.pos), self.mesh_counter, separate_objs, normal_on_the_fly)
            mtrl_of = {}
            for m in meshes:
                mid = int(m["mtl"])
                if mid not in mtrl_of:
                    mtrl_of[mid] = self._resolve_material(mid, mtls, base, create_mtrl)
            is_em = lambda mid: int(self.materials[mtrl_of[mid]][1]["type"]) == L.MTRL_EMISSIVE
            em = [1 if (i in mtrl_of and is_em(i)) else 0 for i in range(len(mtls))]
            # pass 2: the partition into PolygonObjects needs to know which of them are Emissive
            pos, nml, tris, meshes, objects = f.register(len(self.pos), self.mesh_counter, separate_objs, normal_on_the_fly, em,
                                                         (-1 in mtrl_of) and is_em(-1))
            shape_names = f.shape_names
        finally:
            f.close()
        self.pos.extend(map(tuple, pos.tolist()))
        self.nml.extend(map(tuple, nml.tolist()))
        self.mesh_counter += len(meshes)
        first_obj = len(self.objects)
        for ob in objects:
            name = shape_names[ob["shape"]] if ob["shape"] >= 0 else ""
            self.objects.append(dict(type=L.OBJ_POLYGONS, name=name, meshes=[], first_tri=None))
        first_tri = len(self.tris)
        mesh_dicts = []
        for m in meshes:
            d = dict(tris=list(range(first_tri + int(m["first_triangle"]), first_tri + int(m["first_triangle"]) + int(m["n_triangles"]))),
                     mtrl=mtrl_of[int(m["mtl"])], mesh_id=int(m["mesh_id"]))
            mesh_dicts.append(d)
            if m["object"] >= 0:
                self.objects[first_obj + int(m["object"])]["meshes"].append(d)
        for t in tris:
            d = mesh_dicts[int(t["mesh"])]
            self.tris.append(dict(idx=(int(t["idx"][0]), int(t["idx"][1]), int(t["idx"][2])), needNormal=int(t["need_normal"]),
                                  mtrlid=d["mtrl"], mesh_id=d["mesh_id"]))
        order = sorted((int(ob["return_order"]), first_obj + i) for i, ob in enumerate(objects) if ob["return_order"] >= 0)
        objs = [i for _, i in order]
        for o in objs:
            self.blas.setdefault(o, None)
        return objs

    def _register_mesh(self, mesh, cur_obj, shape_name, objs, new_obj, at_shape_end):
        mtype = int(self.materials[mesh["mtrl"]][1]["type"])
        if mtype == L.MTRL_EMISSIVE:
            e = new_obj(shape_name)
            self.objects[e]["meshes"].append(mesh)
            objs.append(e)
            return cur_obj
        if cur_obj is None:
            cur_obj = new_obj(shape_name)
        self.objects[cur_obj]["meshes"].append(mesh)
        return cur_obj

    def _next_mesh_id(self):
        self.mesh_counter += 1
        return self.mesh_counter - 1

    def _resolve_material(self, mid, mtls, base, create_mtrl):
        if mid < 0:
            i = self.find_material("")
            if i < 0:
                i = (create_mtrl("", L.MTRL_DIFFUSE, (1, 1, 1), "", "") if create_mtrl
                     else self.add_material("", L.MTRL_DIFFUSE, (1, 1, 1)))
            return i
        m = mtls[mid]
        i = self.find_material(m.name)
        if i >= 0:
            return i
        if create_mtrl:
            return create_mtrl(m.name, L.MTRL_DIFFUSE, m.diffuse, m.diffuse_texname, m.bump_texname)
        alb = self.load_image(os.path.join(base, m.diffuse_texname)) if m.diffuse_texname else -1
        nm = self.load_image(os.path.join(base, m.bump_texname)) if m.bump_texname else -1
        return self.add_material(m.name, L.MTRL_DIFFUSE, m.diffuse, albedo_map=alb, normal_map=nm)

    def add_mesh(self, name, positions, indices, mtrl, normals=None, uvs=None, need_normal=True, into=None, deformable=False):
        """Programmatic PolygonObject (one TriangleGroupMesh).  positions [N,3], indices [M,3].  `into` appends
        the mesh to an existing polygon object (several materials in one object, like a loaded OBJ).
        deformable: the object's tree will be rebuilt in place on the device (atn_lbvh_rebuild_list), which needs one leaf per
        triangle: it is built with object splits only, whatever `bvh_options` say about spatial splits."""
        positions = np.asarray(positions, F32)
        indices = np.asarray(indices, np.int64)
        if into is None:
            self.objects.append(dict(type=L.OBJ_POLYGONS, name=name, meshes=[], first_tri=None))
            oid = len(self.objects) - 1
        else:
            oid = into
        mesh = dict(tris=[], mtrl=mtrl, mesh_id=self._next_mesh_id())
        flat = indices.reshape(-1)
        P = positions[flat]
        Nn = np.asarray(normals, F32)[flat] if normals is not None else np.tile(np.array([0, 1, 0], F32), (len(flat), 1))
        UV = np.asarray(uvs, F32)[flat] if uvs is not None else np.zeros((len(flat), 2), F32)
        base_v = len(self.pos)
        self.pos.extend(map(tuple, np.concatenate([P, UV[:, :1]], axis=1).tolist()))
        self.nml.extend(map(tuple, np.concatenate([Nn, UV[:, 1:2]], axis=1).tolist()))
        first = len(self.tris)
        nn = 1 if need_normal else 0
        mid = mesh["mesh_id"]
        self.tris.extend(dict(idx=(base_v + 3 * i, base_v + 3 * i + 1, base_v + 3 * i + 2), needNormal=nn,
                              mtrlid=mtrl, mesh_id=mid) for i in range(len(indices)))
        mesh["tris"] = list(range(first, first + len(indices)))
        self.objects[oid]["meshes"].append(mesh)
        self.blas[oid] = None
        if deformable:
            self.objects[oid]["deformable"] = True
        return oid

    def set_mesh_vertices(self, obj_id, positions, indices, normals=None):
        """Move the vertices of a polygon object made by ONE add_mesh call (same index list): a deformation tick."""
        mesh, = self.objects[obj_id]["meshes"]
        flat = np.asarray(indices, np.int64).reshape(-1)
        assert len(flat) == 3 * len(mesh["tris"])
        base_v = self.tris[mesh["tris"][0]]["idx"][0]
        P = np.asarray(positions, F32)[flat]
        for j in range(len(flat)):
            self.pos[base_v + j] = (float(P[j, 0]), float(P[j, 1]), float(P[j, 2]), self.pos[base_v + j][3])
        if normals is not None:
            Nn = np.asarray(normals, F32)[flat]
            for j in range(len(flat)):
                self.nml[base_v + j] = (float(Nn[j, 0]), float(Nn[j, 1]), float(Nn[j, 2]), self.nml[base_v + j][3])
        if self.blas.get(obj_id) is not None:
            self.blas[obj_id] = None
        return base_v, len(flat)

    def add_sphere(self, center, radius, mtrl):
        """TransformableFactory::createSphere (geometry/sphere.h:16-28): a transformable of type Sphere.  The
        BVH traverser never tests spheres, so on this path a sphere only matters as an area light's shape."""
        self.objects.append(dict(type=L.OBJ_SPHERE, center=tuple(float(x) for x in center), radius=float(radius),
                                 mtrl=mtrl, light_id=-1))
        return len(self.objects) - 1

    def create_instance(self, obj_id, mtx_L2W=None):
        """TransformableFactory::createInstance: adds an (L2W, W2L) matrix pair and an Instance entry."""
        M = np.eye(4, dtype=F32) if mtx_L2W is None else np.asarray(mtx_L2W, F32).reshape(4, 4)
        Minv = np.linalg.inv(M.astype(np.float64)).astype(F32) if mtx_L2W is not None else np.eye(4, dtype=F32)
        mid = len(self.matrices)
        self.matrices.append(M)
        self.matrices.append(Minv)
        self.objects.append(dict(type=L.OBJ_INSTANCE, object_id=obj_id, mtx_id=mid, light_id=-1))
        return len(self.objects) - 1

    def _bvh_options(self, deformable=False):
        kw = self.bvh_options if self.bvh_options is not None else DEFAULT_BVH_OPTIONS
        if deformable:
            kw = dict(kw or {}, spatial_splits=0)
        return C.byref(default_bvh_options(**kw)) if kw else None

    def import_sbvh(self, obj_id, path, optimize=False):
        """PolygonObject::importInternalAccelTree + sbvh::buildAsNestedTree's triangle offset.  optimize: the imported tree goes
        through the builder's post passes first (atns_optimize_nodes: same boxes and leaves, re-arranged; `bvh_options` apply)."""
        hdr, mtrl_names, nodes = read_sbvh(path)
        if optimize:
            nodes = optimize_nodes(nodes, self._bvh_options())
        self.blas[obj_id] = ("imported", nodes, hdr)

    def export_sbvh(self, obj_id, path, lib=None):
        """sbvh::exportTree for one polygon object (accelerator/sbvh.cpp:1237-1338): the object's BLAS (the imported one,
        or the SAH tree atns_build_blas builds) with object-local triangle ids, in the reference's .sbvh format."""
        lib = lib or hostlib()
        spec = self.blas.get(obj_id)
        if spec is not None and spec[0] == "imported":
            hdr = spec[2]
            return write_sbvh(path, spec[1], hdr["boxmin"], hdr["boxmax"], hdr["maxDepth"], None, hdr["version"])
        o = self.objects[obj_id]
        if o["type"] != L.OBJ_POLYGONS:
            raise ValueError("only polygon objects carry a bottom-level tree")
        tri_ids = [t for m in o["meshes"] for t in m["tris"]]
        pos = np.asarray(self.pos, F32).reshape(-1, 4)
        tris = np.zeros(len(self.tris), L.TRIANGLE_PARAM)
        tris["idx"] = np.asarray([t["idx"] for t in self.tris], np.int32)
        ids = np.asarray(tri_ids, np.uint32)
        out = C.c_void_p(); cnt = C.c_uint32()
        bmin = (C.c_float * 3)(); bmax = (C.c_float * 3)()
        rc = lib.atns_build_blas_opt(L.ptr(pos), L.ptr(tris), L.ptr(ids), len(ids), self._bvh_options(), C.byref(out), C.byref(cnt),
                                     bmin, bmax, None)
        if rc != 0:
            raise RuntimeError("atns_build_blas_opt failed: %d" % rc)
        nodes = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(cnt.value * 48,)).view(L.BVH_NODE).copy()
        lib.atns_free(out)
        first = min(tri_ids)
        leaf = nodes["f0"] >= 0
        nodes["f1"][leaf] -= F32(first)         # the file holds object-local ids; import adds the offset back
        return write_sbvh(path, nodes, list(bmin), list(bmax))

    # ---------------------------------------------------------------- lights
    def add_area_light(self, instance_id, color, intensity, scale=1.0):
        l = np.zeros((), L.LIGHT_PARAM)
        l["type"] = L.LIGHT_AREA
        l["attrib"] = 0
        l["light_color"] = color
        l["innerAngle"] = l["outerAngle"] = np.pi
        l["scale"], l["intensity"] = scale, intensity
        l["arealight_objid"] = instance_id
        l["envmapidx"] = -1
        self.lights.append(l)
        lid = len(self.lights) - 1
        o = self.objects[instance_id]
        o["light_id"] = lid
        if o["type"] == L.OBJ_INSTANCE:
            self.objects[o["object_id"]]["light_id"] = lid
        return lid

    def add_ibl(self, envmap_tex, scale=1.0, avg_illum=None, multiplyer=1.0):
        l = np.zeros((), L.LIGHT_PARAM)
        l["type"] = L.LIGHT_IBL
        l["attrib"] = L.LATTR_INFINITE | L.LATTR_IBL
        l["innerAngle"] = l["outerAngle"] = np.pi
        l["scale"], l["intensity"] = scale, 1.0
        l["arealight_objid"] = -1
        l["envmapidx"] = envmap_tex
        self.lights.append(l)
        self.config.bg.envmap_tex_idx = envmap_tex
        self.config.bg.multiplyer = multiplyer
        if avg_illum is not None:
            self.config.bg.avgIllum = avg_illum
        return len(self.lights) - 1

    def add_point_light(self, pos, color, intensity, scale=1.0):
        l = np.zeros((), L.LIGHT_PARAM)
        l["type"] = L.LIGHT_POINT
        l["attrib"] = L.LATTR_SINGULAR
        l["pos"] = list(pos) + [1.0]
        l["light_color"] = color
        l["innerAngle"] = l["outerAngle"] = np.pi
        l["scale"], l["intensity"] = scale, intensity
        l["arealight_objid"] = -1
        l["envmapidx"] = -1
        self.lights.append(l)
        return len(self.lights) - 1

    def set_background(self, color):
        self.config.bg.bg_color[:] = color

    # ---------------------------------------------------------------- build
    def build(self):
        fs = FlatScene()
        lib = hostlib()
        pos = np.asarray(self.pos, F32).reshape(-1, 4)
        nml = np.asarray(self.nml, F32).reshape(-1, 4)
        nt = len(self.tris)
        tris = np.zeros(nt, L.TRIANGLE_PARAM)
        if nt:
            idx = np.asarray([t["idx"] for t in self.tris], np.int32)
            tris["idx"] = idx
            tris["needNormal"] = [t["needNormal"] for t in self.tris]
            tris["mtrlid"] = [t["mtrlid"] for t in self.tris]
            tris["mesh_id"] = [t["mesh_id"] for t in self.tris]
            # triangle::BuildTriangle: area = 0.5 * |cross(e0, e1)| in fp32 (triangle.cpp:125-130)
            p0, p1, p2 = pos[idx[:, 0], :3], pos[idx[:, 1], :3], pos[idx[:, 2], :3]
            e0, e1 = (p1 - p0).astype(F32), (p2 - p0).astype(F32)
            cx = e0[:, 1] * e1[:, 2] - e0[:, 2] * e1[:, 1]
            cy = e0[:, 2] * e1[:, 0] - e0[:, 0] * e1[:, 2]
            cz = e0[:, 0] * e1[:, 1] - e0[:, 1] * e1[:, 0]
            tris["area"] = F32(0.5) * _length3(cx, cy, cz)

        objs = np.zeros(len(self.objects), L.OBJECT_PARAM)
        objs["object_id"] = -1; objs["mtx_id"] = -1; objs["triangle_id"] = -1; objs["light_id"] = -1
        objs["sphere_mtrl_id"] = -1
        obj_bbox = {}
        bvh_lists = [None]          # [0] = TLAS
        blas_index = {}
        bvh_stats = {}
        for oid, o in enumerate(self.objects):
            if o["type"] != L.OBJ_POLYGONS:
                continue
            tri_ids = [t for m in o["meshes"] for t in m["tris"]]
            if not tri_ids:
                continue
            first, num = tri_ids[0], len(tri_ids)
            area = F32(0)
            for m in o["meshes"]:
                ma = F32(0)
                for t in m["tris"]:
                    ma = F32(ma + tris["area"][t])
                area = F32(area + ma)
            objs[oid]["type"] = L.OBJ_POLYGONS
            objs[oid]["area"] = area
            objs[oid]["triangle_id"] = first
            objs[oid]["triangle_num"] = num
            objs[oid]["light_id"] = o.get("light_id", -1)
            spec = self.blas.get(oid)
            if spec is not None and spec[0] == "imported":
                nodes = spec[1].copy()
                leaf = nodes["f0"] >= 0
                nodes["f1"][leaf] += F32(first)     # sbvh::buildAsNestedTree, sbvh.cpp:109-128
                hdr = spec[2]
                bmin, bmax = np.asarray(hdr["boxmin"], F32), np.asarray(hdr["boxmax"], F32)
            else:
                out = C.c_void_p(); cnt = C.c_uint32()
                bmin = (C.c_float * 3)(); bmax = (C.c_float * 3)()
                ids = np.asarray(tri_ids, np.uint32)
                st = BvhStats()
                rc = lib.atns_build_blas_opt(L.ptr(pos), L.ptr(tris), L.ptr(ids), num, self._bvh_options(bool(o.get("deformable"))),
                                             C.byref(out), C.byref(cnt), bmin, bmax, C.byref(st))
                if rc != 0:
                    raise RuntimeError("atns_build_blas_opt failed: %d" % rc)
                bvh_stats[oid] = dict(nodes=st.n_nodes, leaves=st.n_leaves, spatial_splits=st.n_spatial_splits, reinsertions=st.n_reinsertions, sah=st.sah_cost)
                nodes = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(cnt.value * 48,)).view(L.BVH_NODE).copy()
                lib.atns_free(out)
                bmin, bmax = np.asarray(list(bmin), F32), np.asarray(list(bmax), F32)
            obj_bbox[oid] = (bmin, bmax)
            bvh_lists.append(nodes)
            blas_index[oid] = len(bvh_lists) - 1

        inst = []
        for oid, o in enumerate(self.objects):
            if o["type"] == L.OBJ_SPHERE:
                objs[oid]["type"] = L.OBJ_SPHERE
                r = F32(o["radius"])
                objs[oid]["area"] = F32(0)      # transformable default; sphere area is computed on the fly
                objs[oid]["sphere_center"] = o["center"]
                objs[oid]["sphere_radius"] = r
                objs[oid]["sphere_mtrl_id"] = o["mtrl"]
                objs[oid]["light_id"] = o.get("light_id", -1)
                c = np.asarray(o["center"], F32)
                # scene->add(sphere): a top-layer leaf without nested tree (exid = -1), never tested
                inst.append((oid, -1, c - r, c + r))
                continue
            if o["type"] != L.OBJ_INSTANCE:
                continue
            objs[oid]["type"] = L.OBJ_INSTANCE
            objs[oid]["object_id"] = o["object_id"]
            objs[oid]["mtx_id"] = o["mtx_id"]
            objs[oid]["light_id"] = o.get("light_id", -1)
            bmin, bmax = obj_bbox[o["object_id"]]
            M = self.matrices[o["mtx_id"]]
            corners = np.array([[x, y, z, 1.0] for x in (bmin[0], bmax[0]) for y in (bmin[1], bmax[1]) for z in (bmin[2], bmax[2])], F32)
            w = (corners @ M.T)[:, :3]
            inst.append((oid, blas_index[o["object_id"]], w.min(0), w.max(0)))

        if inst:
            boxes = np.asarray([np.concatenate([a, b]) for (_, _, a, b) in inst], F32)
            oids = np.asarray([i[0] for i in inst], np.int32)
            exids = np.asarray([i[1] for i in inst], np.int32)
            mesh_ids = np.full(len(inst), -1, np.int32)
            out = C.c_void_p(); cnt = C.c_uint32()
            rc = lib.atns_build_tlas(L.ptr(boxes), L.ptr(oids), L.ptr(exids), L.ptr(mesh_ids), len(inst), C.byref(out), C.byref(cnt))
            if rc != 0:
                raise RuntimeError("atns_build_tlas failed: %d" % rc)
            tlas = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(cnt.value * 48,)).view(L.BVH_NODE).copy()
            lib.atns_free(out)
            bvh_lists[0] = tlas
            smin, smax = boxes[:, :3].min(0), boxes[:, 3:].max(0)
        else:
            raise RuntimeError("scene has no instances")

        mats = np.zeros(len(self.materials), L.MATERIAL_PARAM)
        for i, (_, m) in enumerate(self.materials):
            mats[i] = m
        lights = np.zeros(len(self.lights), L.LIGHT_PARAM)
        for i, l in enumerate(self.lights):
            lights[i] = l
        mtx = np.asarray(self.matrices, F32).reshape(-1, 4, 4) if self.matrices else np.zeros((0, 4, 4), F32)

        lists = (L.BvhList * len(bvh_lists))()
        for i, n in enumerate(bvh_lists):
            lists[i].nodes = n.ctypes.data
            lists[i].count = len(n)
        texd = (L.TextureDesc * max(1, len(self.textures)))()
        for i, (_, t) in enumerate(self.textures):
            texd[i].texels = t.ctypes.data
            texd[i].height, texd[i].width = t.shape[0], t.shape[1]

        d = fs.desc
        d.objects, d.n_objects = L.ptr(objs), len(objs)
        d.matrices, d.n_matrices = L.ptr(mtx), len(mtx)
        d.materials, d.n_materials = L.ptr(mats), len(mats)
        d.lights, d.n_lights = L.ptr(lights), len(lights)
        d.triangles, d.n_triangles = L.ptr(tris), len(tris)
        d.vtx_pos, d.vtx_nml, d.n_vertices = L.ptr(pos), L.ptr(nml), len(pos)
        d.bvh_lists, d.n_bvh_lists = C.addressof(lists), len(bvh_lists)
        d.textures, d.n_textures = C.addressof(texd), len(self.textures)
        d.config = self.config
        d.scene_bbox_min[:] = [float(x) for x in smin]
        d.scene_bbox_max[:] = [float(x) for x in smax]
        npr = (np.stack(self.npr_lights) if self.npr_lights else np.zeros(0, L.LIGHT_PARAM)).astype(L.LIGHT_PARAM)
        d.npr_target_lights, d.n_npr_target_lights = L.ptr(npr), len(npr)
        d.enable_shadowray_base_stylized_shadow = 1 if self.enable_shadowray_base_stylized_shadow else 0
        sst = None
        if self.screen_space_texture is not None:
            sst = np.ascontiguousarray(self.screen_space_texture, F32)
            d.screen_space_texture.texels = sst.ctypes.data
            d.screen_space_texture.height, d.screen_space_texture.width = sst.shape[0], sst.shape[1]
        fs.keep = [objs, mtx, mats, lights, tris, pos, nml, bvh_lists, lists, texd, [t for _, t in self.textures], npr, sst]
        fs.lists = lists
        fs.blas_index = dict(blas_index)      # polygon object id -> its node list
        fs.bvh_stats = bvh_stats              # polygon object id -> atns_bvh_stats of the tree built here
        fs.arrays = dict(objects=objs, matrices=mtx, materials=mats, lights=lights, triangles=tris,
                         vtx_pos=pos, vtx_nml=nml, bvh_lists=bvh_lists, textures=[t for _, t in self.textures])
        fs.names = dict(materials=[n for n, _ in self.materials], textures=[n for n, _ in self.textures])
        return fs


def optimize_nodes(nodes, options=None):
    """atns_optimize_nodes on a numpy node array; options: None or ctypes.byref(BvhOptions)."""
    lib = hostlib()
    nodes = np.ascontiguousarray(nodes)
    out = C.c_void_p(); cnt = C.c_uint32(); st = BvhStats()
    rc = lib.atns_optimize_nodes(nodes.ctypes.data, len(nodes), options, C.byref(out), C.byref(cnt), C.byref(st))
    if rc != 0:
        raise RuntimeError("atns_optimize_nodes failed: %d" % rc)
    res = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(cnt.value * 48,)).view(L.BVH_NODE).copy()
    lib.atns_free(out)
    return res


def _resolve_case(path):
    """Case-insensitive file lookup (sponza.mtl names SP_LUK.JPG, the file is sp_luk.JPG)."""
    if os.path.exists(path):
        return path
    d, f = os.path.dirname(path) or ".", os.path.basename(path)
    if not os.path.isdir(d):
        return None
    for e in os.listdir(d):
        if e.lower() == f.lower():
            return os.path.join(d, e)
    return None


def read_sbvh(path):
    """SbvhFileHeader + voxel-material table + ThreadedSbvhNode[] (accelerator/sbvh.cpp:1220-1235,1345-1400)."""
    with open(path, "rb") as f:
        buf = f.read()
    magic, ver, node_num, max_depth, cnt_mtrl = struct.unpack_from("<4sIIII", buf, 0)
    box = struct.unpack_from("<6f", buf, 20)
    off = 44
    if magic != b"SBVH":
        raise ValueError("not an SBVH file: %r" % magic)
    names = {}
    for _ in range(cnt_mtrl):
        mid, ln = struct.unpack_from("<ii", buf, off)
        off += 8
        names[mid] = buf[off:off + ln].split(b"\0")[0].decode()
        off += ln
    nodes = np.frombuffer(buf, L.BVH_NODE, count=node_num, offset=off).copy()
    assert off + node_num * 48 == len(buf), "trailing bytes in sbvh file"
    hdr = dict(version=ver, nodeNum=node_num, maxDepth=max_depth, boxmin=box[:3], boxmax=box[3:])
    return hdr, names, nodes


def threaded_depth(nodes):
    """Depth of a threaded tree, root = depth 0 as sbvh's m_maxDepth counts it: an inner node's left child is its hit link, the right child is the
    left child's miss link (sbvh::convert / registerThreadedBvh order, accelerator/sbvh.cpp:1085-1180)."""
    n = len(nodes)
    if n == 0:
        return 0
    depth = np.zeros(n, np.int32)
    depth[0] = 1                            # stored +1 so that 0 means 'not reached'
    hit = nodes["hit"].astype(np.int64)
    miss = nodes["miss"].astype(np.int64)
    leaf = nodes["f0"] >= 0
    for i in range(n):                      # children always follow their parent in the array
        if leaf[i] or depth[i] == 0:
            continue
        l = hit[i]
        if 0 <= l < n:
            depth[l] = depth[i] + 1
            r = miss[l]
            if 0 <= r < n and r != miss[i]:
                depth[r] = depth[i] + 1
    return int(depth.max()) - 1


def write_sbvh(path, nodes, boxmin, boxmax, max_depth=None, mtrl_names=None, version=0x01000000):
    """Inverse of read_sbvh: the file sbvh::exportTree writes (accelerator/sbvh.cpp:1220-1338) -- SbvhFileHeader,
    the voxel-material table (id, 4-byte-aligned length, zero-padded name) and ThreadedSbvhNode[].  A file read with
    read_sbvh and written back is byte-identical; a tree built by atns_build_blas exported this way loads in the
    reference through PolygonObject::importInternalAccelTree."""
    nodes = np.ascontiguousarray(nodes, L.BVH_NODE)
    if max_depth is None:
        max_depth = threaded_depth(nodes)
    mtrl_names = mtrl_names or {}
    out = bytearray()
    out += struct.pack("<4sIIII", b"SBVH", version, len(nodes), int(max_depth), len(mtrl_names))
    out += struct.pack("<6f", *[float(x) for x in boxmin], *[float(x) for x in boxmax])
    for mid in sorted(mtrl_names):          # std::map iteration order
        name = mtrl_names[mid].encode()
        aligned = (len(name) + 3) // 4 * 4
        out += struct.pack("<ii", mid, aligned) + name + b"\0" * (aligned - len(name))
    out += nodes.tobytes()
    with open(path, "wb") as f:
        f.write(out)
    return len(out)
