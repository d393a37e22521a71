"""Scene definitions for the BASELINE configs, following src/common/scenedefs.cpp of the reference.

  cornell_box()  <- ObjCornellBoxScene::makeScene / getCameraPosAndAt (scenedefs.cpp:732-802)
  sponza_lod()   <- SponzaScene (scenedefs.cpp:806-860) restricted to the blobs that exist:
                    asset/sponza/sponza_lod.obj + sponza_lod.sbvh (+ textures).  The full
                    sponza.obj/.sbvh are missing large blobs in the reference snapshot.
Data files are committed under assets/ (copied byte-for-byte from /root/reference/asset).
"""
import os

import numpy as np

from .. import layout as L
from .builder import SceneBuilder

ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "assets")


def cornell_box(asset_dir=None):
    """Returns (FlatScene, camera dict(pos, at, vfov))."""
    asset_dir = asset_dir or os.path.join(ASSETS, "cornellbox")
    b = SceneBuilder()
    emit = b.add_material("light", L.MTRL_EMISSIVE, (1.0, 1.0, 1.0))      # scenedefs.cpp:735

    def create_mtrl(name, mtype, clr, albedo, nml):                        # scenedefs.cpp:738-771
        if name == "shortBox":
            return b.add_material(name, L.MTRL_SPECULAR, (0.7, 0.6, 0.5), roughness=0.1, ior=0.01)
        if name == "floor":
            return b.add_material(name, L.MTRL_GGX, (0.7, 0.6, 0.5), roughness=0.1, ior=0.01)
        return b.add_material(name, mtype, clr)

    objs = b.load_obj(os.path.join(asset_dir, "orig.obj"), create_mtrl=create_mtrl,
                      separate_objs=True, normal_on_the_fly=True)
    # createInstance(ctxt, objs[0], trans 0, rot 0, scale 1): identity matrix pair (:775-781)
    light = b.create_instance(objs[0])
    b.add_area_light(light, b.materials[emit][1]["baseColor"][:3], 200.0)  # :783-784
    for o in objs[1:]:
        b.create_instance(o)                                               # :786-789
    b.set_background((0.0, 0.0, 0.0))
    cam = dict(pos=(0.0, 1.0, 3.0), at=(0.0, 1.0, 0.0), vfov=45.0)         # :794-802
    return b.build(), cam


def synthetic_envmap(w=2048, h=1024, seed=0):
    """Stand-in for the missing asset/envmap/studio015.hdr: smooth analytic sky + sun lobe.

    value(u, v) = sky(v) + sun, with
      sky  = mix((0.35,0.30,0.25), (0.45,0.65,1.0), smoothstep(0.45,0.75,v)) * 1.2
      sun  = (60,55,45) * exp(-((u-0.3)^2 + (v-0.8)^2) / 0.0008)
    Stored in aten's texture order (row 0 = v near 0 = bottom).  `seed` is unused (kept so the
    fixture name records determinism).
    """
    v = (np.arange(h, dtype=np.float32) + 0.5) / h
    u = (np.arange(w, dtype=np.float32) + 0.5) / w
    t = np.clip((v - 0.45) / 0.3, 0, 1)
    t = t * t * (3 - 2 * t)
    ground = np.array([0.35, 0.30, 0.25], np.float32)
    sky = np.array([0.45, 0.65, 1.0], np.float32)
    col = (ground[None, :] * (1 - t[:, None]) + sky[None, :] * t[:, None]) * np.float32(1.2)
    img = np.repeat(col[:, None, :], w, axis=1)
    d2 = (u[None, :] - 0.3) ** 2 + (v[:, None] - 0.8) ** 2
    sun = np.exp(-d2 / 0.0008).astype(np.float32)
    img = img + sun[:, :, None] * np.array([60, 55, 45], np.float32)[None, None, :]
    out = np.ones((h, w, 4), np.float32)
    out[:, :, :3] = img
    return out


def envmap_avg_illum(tex):
    """ImageBasedLight::preCompute's sin(theta)-weighted mean luminance (light/ibl.cpp:10-75)."""
    h = tex.shape[0]
    lum = 0.212639 * tex[:, :, 0] + 0.71517 * tex[:, :, 1] + 0.0721926 * tex[:, :, 2]
    theta = np.pi * (np.arange(h) + 0.5) / h
    s = np.sin(theta)[:, None]
    return float((lum * s).sum() / (s.sum() * tex.shape[1]))


def sponza_lod(asset_dir=None, mtype=L.MTRL_GGX, ibl=True, use_sbvh=True, textures=True, bvh_options=None, optimize_sbvh=False):
    """BASELINE config 3 stand-in: sponza_lod.obj (12,852 tris) with the reference-built
    sponza_lod.sbvh tree, GGX materials, synthetic IBL."""
    asset_dir = asset_dir or os.path.join(ASSETS, "sponza")
    b = SceneBuilder()

    def create_mtrl(name, mt, clr, albedo, nml):
        alb = b.load_image(os.path.join(asset_dir, albedo)) if (albedo and textures) else -1
        nm = b.load_image(os.path.join(asset_dir, nml)) if (nml and textures) else -1
        if mtype == L.MTRL_GGX:
            return b.add_material(name, L.MTRL_GGX, clr, albedo_map=alb, normal_map=nm, roughness=0.3, ior=0.01)
        if mtype == L.MTRL_DISNEY:
            return b.add_material(name, L.MTRL_DISNEY, clr, albedo_map=alb, normal_map=nm,
                                  roughness=0.4, metallic=0.1, specular=0.5, clearcoat=0.2)
        return b.add_material(name, mt, clr, albedo_map=alb, normal_map=nm)

    cam = dict(pos=(0.0, 1.0, 3.0), at=(0.0, 1.0, 0.0), vfov=45.0)         # scenedefs.cpp:847-860
    # own tree (use_sbvh=False): children nearer the viewer are threaded first
    b.bvh_options = dict(order_point=cam["pos"]) if bvh_options is None else bvh_options
    objs = b.load_obj(os.path.join(asset_dir, "sponza_lod.obj"), create_mtrl=create_mtrl)
    if use_sbvh:
        b.import_sbvh(objs[0], os.path.join(asset_dir, "sponza_lod.sbvh"), optimize=optimize_sbvh)
    b.create_instance(objs[0])
    if ibl:
        env = synthetic_envmap()
        tid = b.add_texture("synthetic_sky_2048x1024", env)
        b.add_ibl(tid, avg_illum=envmap_avg_illum(env))
    else:
        b.set_background((1.0, 1.0, 1.0))
    return b.build(), cam


def cornell_box_variant(lights="area", move_boxes=True, asset_dir=None, extra_materials=False):
    """Parity-test variant of the Cornell box (not a reference scene): the two boxes are instanced with
    non-identity matrices (translation / rotation about y, so the W2L ray transform, the L2W hit transform
    and the instance area ratio are exercised) and the light set is selectable:
      "area"  the reference's polygon light            "point" / "spot" / "directional"  punctual lights
      "mixed" area + point + spot + directional (uniform light pick among four)
      "sphere" a sphere area light (AreaLight over a sphere object; the sphere itself is never hit by rays)
    """
    asset_dir = asset_dir or os.path.join(ASSETS, "cornellbox")
    b = SceneBuilder()
    emit = b.add_material("light", L.MTRL_EMISSIVE, (1.0, 1.0, 1.0))

    def create_mtrl(name, mtype, clr, albedo, nml):
        if name == "shortBox":
            return b.add_material(name, L.MTRL_DISNEY, (0.7, 0.6, 0.5), roughness=0.35, metallic=0.3,
                                  specular=0.6, clearcoat=0.4, clearcoatGloss=0.7, sheen=0.3)
        if name == "floor":
            return b.add_material(name, L.MTRL_GGX, (0.7, 0.6, 0.5), roughness=0.1, ior=0.01)
        if extra_materials is True and name == "tallBox":       # glass (refraction.cpp), like the reference's glass spheres
            return b.add_material(name, L.MTRL_REFRACTION, (0.9, 0.9, 0.9), ior=1.5)
        if extra_materials and name == "leftWall":
            return b.add_material(name, L.MTRL_OREN_NAYAR, clr, roughness=0.6)
        if extra_materials == "rough" and name == "tallBox":     # frosted glass (microfacet_refraction.cpp)
            return b.add_material(name, L.MTRL_MICROFACET_REFRACTION, (0.9, 0.9, 0.9), ior=1.5, roughness=0.2)
        if extra_materials == "rough" and name == "rightWall":
            return b.add_material(name, L.MTRL_VELVET, clr, roughness=0.4)
        if extra_materials == "carpaint" and name == "shortBox":   # clearcoat over flakes over diffuse (car_paint.cpp)
            return b.add_carpaint_material(name, (1.0, 1.0, 1.0))
        if extra_materials == "carpaint" and name == "backWall":
            return b.add_carpaint_material(name, (0.9, 0.9, 0.9), diffuse_color=(0.1, 0.3, 0.8), flakes_color=(0.9, 0.9, 1.0),
                                           flake_size=0.4, clearcoat_ior=1.6, flake_scale=60.0)
        if extra_materials == "retro" and name == "rightWall":   # prismatic-sheet retroreflector (retroreflective.cpp)
            return b.add_material(name, L.MTRL_RETROREFLECTIVE, clr, roughness=0.3, ior=1.5)
        if extra_materials == "retro" and name == "tallBox":
            return b.add_material(name, L.MTRL_RETROREFLECTIVE, (0.9, 0.9, 0.9), roughness=0.1, ior=1.33)
        if extra_materials and name == "backWall":
            return b.add_material(name, L.MTRL_BECKMAN, (0.7, 0.7, 0.7), roughness=0.25, ior=0.2)
        return b.add_material(name, mtype, clr)

    objs = b.load_obj(os.path.join(asset_dir, "orig.obj"), create_mtrl=create_mtrl,
                      separate_objs=True, normal_on_the_fly=True)
    names = [b.objects[o]["name"] for o in objs]

    def rot_y_trans(deg, t):
        c, s_ = np.cos(np.radians(deg)), np.sin(np.radians(deg))
        return np.array([[c, 0, s_, t[0]], [0, 1, 0, t[1]], [-s_, 0, c, t[2]], [0, 0, 0, 1]], np.float32)

    inst = {}
    for o, n in zip(objs, names):
        if n == "light" and lights not in ("area", "mixed"):
            continue        # an emissive object without a registered light has light_id = -1 (the reference would index lights[-1])
        M = None
        if move_boxes and n == "tallBox":
            M = rot_y_trans(17.0, (0.15, 0.0, 0.1))
        if move_boxes and n == "shortBox":
            M = rot_y_trans(-23.0, (-0.2, 0.25, 0.05))
        inst[n] = b.create_instance(o, M)
    if lights in ("area", "mixed"):
        b.add_area_light(inst["light"], (1.0, 1.0, 1.0), 200.0)
    if lights == "sphere":
        sm = b.add_material("sphere_emit", L.MTRL_EMISSIVE, (1.0, 0.9, 0.8))
        sp = b.add_sphere((0.1, 1.55, 0.2), 0.18, sm)
        b.add_area_light(sp, (1.0, 0.9, 0.8), 30.0)
    if lights in ("point", "mixed"):
        b.add_point_light((0.3, 1.6, 0.4), (1.0, 0.9, 0.8), 40.0)
    if lights in ("spot", "mixed"):
        l = np.zeros((), L.LIGHT_PARAM)
        l["type"] = L.LIGHT_SPOT
        l["attrib"] = L.LATTR_SINGULAR
        l["pos"] = (-0.5, 1.8, 0.6, 1.0)
        d = np.array([0.3, -1.0, -0.4], np.float32)
        d /= np.linalg.norm(d)
        l["dir"] = (d[0], d[1], d[2], 0.0)
        l["light_color"] = (0.8, 0.9, 1.0)
        l["innerAngle"], l["outerAngle"] = np.radians(25.0), np.radians(60.0)
        l["scale"], l["intensity"] = 1.0, 60.0
        l["arealight_objid"], l["envmapidx"] = -1, -1
        b.lights.append(l)
    if lights in ("directional", "mixed"):
        l = np.zeros((), L.LIGHT_PARAM)
        l["type"] = L.LIGHT_DIRECTION
        l["attrib"] = L.LATTR_SINGULAR | L.LATTR_INFINITE
        d = np.array([-0.2, -0.6, -1.0], np.float32)
        d /= np.linalg.norm(d)
        l["dir"] = (d[0], d[1], d[2], 0.0)
        l["light_color"] = (1.0, 0.95, 0.9)
        l["innerAngle"] = l["outerAngle"] = np.pi
        l["scale"], l["intensity"] = 1.0, 3.0
        l["arealight_objid"], l["envmapidx"] = -1, -1
        b.lights.append(l)
    b.set_background((0.02, 0.03, 0.05))
    cam = dict(pos=(0.0, 1.0, 3.0), at=(0.0, 1.0, 0.0), vfov=45.0)
    return b.build(), cam


# ---------------------------------------------------------------------------------------------------
# Procedural stand-in for BASELINE config 4 (Crytek Sponza: sponza.obj/.sbvh are missing blobs)
# ---------------------------------------------------------------------------------------------------
def _grid_mesh(fn, nu, nv, uv_scale=(1.0, 1.0), flip=False):
    """Tessellated parametric surface: fn(u, v) -> xyz for u, v in [0,1] (arrays).  Smooth vertex normals
    from central differences; two triangles per cell."""
    u = np.linspace(0.0, 1.0, nu + 1)
    v = np.linspace(0.0, 1.0, nv + 1)
    U, V = np.meshgrid(u, v, indexing="ij")
    P = fn(U, V).astype(np.float64)
    e = 1e-4
    du = fn(np.clip(U + e, 0, 1), V) - fn(np.clip(U - e, 0, 1), V)
    dv = fn(U, np.clip(V + e, 0, 1)) - fn(U, np.clip(V - e, 0, 1))
    N = np.cross(du.reshape(-1, 3), dv.reshape(-1, 3))
    N /= np.maximum(np.linalg.norm(N, axis=1, keepdims=True), 1e-20)
    if flip:
        N = -N
    idx = np.arange((nu + 1) * (nv + 1)).reshape(nu + 1, nv + 1)
    a, b_, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    tri = np.concatenate([np.stack([a, b_, c], 1), np.stack([a, c, d], 1)]) if not flip else \
        np.concatenate([np.stack([a, c, b_], 1), np.stack([a, d, c], 1)])
    uv = np.stack([U.ravel() * uv_scale[0], V.ravel() * uv_scale[1]], 1)
    return P.reshape(-1, 3).astype(np.float32), N.astype(np.float32), uv.astype(np.float32), tri


def _icosphere(level):
    t = (1.0 + 5 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10),
         (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(level):
        cache, nf = {}, []

        def mid(i, j):
            k = (min(i, j), max(i, j))
            if k not in cache:
                m = v[i] + v[j]
                v.append(m / np.linalg.norm(m))
                cache[k] = len(v) - 1
            return cache[k]
        for (a, b_, c) in f:
            ab, bc, ca = mid(a, b_), mid(b_, c), mid(c, a)
            nf += [(a, ab, ca), (b_, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.asarray(v, np.float64), np.asarray(f, np.int64)


def atrium(asset_dir=None, mtype=L.MTRL_DISNEY, detail=1.0, bvh_options=None):
    """Procedural colonnaded hall, ~250 k unique triangles (+ 6 instances of a 20 k-triangle statue), Disney
    materials with the Sponza albedo / normal-map textures, IBL + one polygon area light.  Stand-in for
    BASELINE config 4 "Crytek Sponza 4K 8spp 8-bounce Disney + textures" (≈262 k triangles), whose
    geometry blob is absent; deterministic (no RNG).  `detail` scales the tessellation."""
    asset_dir = asset_dir or os.path.join(ASSETS, "sponza")
    b = SceneBuilder()
    cam = dict(pos=(-7.0, 1.7, 0.6), at=(0.0, 1.5, 0.0), vfov=45.0)
    # the trees built here: children nearer the viewer are threaded first
    b.bvh_options = dict(order_point=cam["pos"]) if bvh_options is None else bvh_options

    def tex(name):
        return b.load_image(os.path.join(asset_dir, name))

    def mtrl(name, clr, alb, nm, **kw):
        if mtype == L.MTRL_DISNEY:
            std = dict(roughness=0.5, metallic=0.0, specular=0.5, clearcoat=0.0, clearcoatGloss=0.5, sheen=0.0)
            std.update(kw)
            return b.add_material(name, L.MTRL_DISNEY, clr, albedo_map=tex(alb) if alb else -1,
                                  normal_map=tex(nm) if nm else -1, **std)
        return b.add_material(name, mtype, clr, albedo_map=tex(alb) if alb else -1, normal_map=tex(nm) if nm else -1,
                              roughness=kw.get("roughness", 0.3), ior=0.01)

    m_floor = mtrl("floor", (0.8, 0.8, 0.8), "KAMEN.JPG", "KAMEN-nml.png", roughness=0.25, specular=0.7, clearcoat=0.3)
    m_wall = mtrl("wall", (0.8, 0.8, 0.8), "01_STUB.JPG", "01_STUB-nml.png", roughness=0.7)
    m_col = mtrl("column", (0.8, 0.8, 0.8), "sp_luk.JPG", "sp_luk-nml.png", roughness=0.55, sheen=0.2)
    m_gold = mtrl("statue", (0.9, 0.7, 0.3), None, None, roughness=0.3, metallic=0.9, specular=0.8)
    m_emit = b.add_material("lamp", L.MTRL_EMISSIVE, (1.0, 0.95, 0.9))

    def n(x):
        return max(2, int(round(x * detail)))

    X, Z, H = 8.0, 4.0, 6.0
    # floor: gentle relief
    P, N, UV, T = _grid_mesh(lambda u, v: np.stack([(2 * u - 1) * X, 0.02 * np.sin(37.0 * u) * np.cos(23.0 * v)
                                                    + 0.01 * np.sin(211.0 * u + 89.0 * v), (1 - 2 * v) * Z], -1),
                             n(320), n(160), uv_scale=(8.0, 4.0))
    hall = b.add_mesh("hall", P, T, m_floor, normals=N, uvs=UV, need_normal=False)
    # walls with brick-like relief; inward-facing
    def wall(axis, sign):
        def fn(u, v):
            relief = 0.03 * np.sin(60.0 * u) * np.sin(40.0 * v) + 0.015 * np.sin(170.0 * u + 130.0 * v)
            if axis == "z":
                return np.stack([(2 * u - 1) * X * sign, v * H, np.full_like(u, sign * Z) - sign * relief], -1)
            return np.stack([np.full_like(u, sign * X) - sign * relief, v * H, (1 - 2 * u) * Z * sign], -1)
        return _grid_mesh(fn, n(160), n(64), uv_scale=(6.0, 3.0))
    for axis, sign in (("z", 1.0), ("z", -1.0), ("x", 1.0), ("x", -1.0)):
        P, N, UV, T = wall(axis, sign)
        b.add_mesh("wall", P, T, m_wall, normals=N, uvs=UV, need_normal=False, into=hall)
    # fluted columns
    for side in (-1.0, 1.0):
        for k in range(6):
            cx, cz = -6.25 + 2.5 * k, side * 2.3

            def col(u, v, cx=cx, cz=cz):
                ang = 2 * np.pi * u
                r = 0.28 + 0.02 * np.cos(16 * ang) + 0.06 * np.exp(-40.0 * v) + 0.06 * np.exp(-40.0 * (1 - v))
                return np.stack([cx + r * np.cos(ang), v * 5.0, cz - r * np.sin(ang)], -1)
            P, N, UV, T = _grid_mesh(col, n(48), n(40), uv_scale=(2.0, 5.0))
            b.add_mesh("column", P, T, m_col, normals=N, uvs=UV, need_normal=False, into=hall)
    # lamp: one quad under the open roof
    lamp_p = np.array([[-1.0, 5.6, -0.6], [1.0, 5.6, -0.6], [1.0, 5.6, 0.6], [-1.0, 5.6, 0.6]], np.float32)
    lamp = b.add_mesh("lamp", lamp_p, [[0, 1, 2], [0, 2, 3]], m_emit)
    # statue: displaced icosphere, instanced
    V, F = _icosphere(5 if detail >= 1.0 else 3)
    disp = 1.0 + 0.12 * np.sin(7.0 * V[:, 0]) * np.sin(9.0 * V[:, 1]) * np.sin(5.0 * V[:, 2]) + 0.05 * np.sin(23.0 * V[:, 1])
    SP = (V * disp[:, None]).astype(np.float32)
    statue = b.add_mesh("statue", SP, F, m_gold, need_normal=True)

    def trs(scale, deg, t):
        c, s_ = np.cos(np.radians(deg)), np.sin(np.radians(deg))
        return np.array([[c * scale, 0, s_ * scale, t[0]], [0, scale, 0, t[1]], [-s_ * scale, 0, c * scale, t[2]],
                         [0, 0, 0, 1]], np.float32)

    b.create_instance(hall)
    li = b.create_instance(lamp)
    b.add_area_light(li, (1.0, 0.95, 0.9), 60.0)
    for k in range(6):
        b.create_instance(statue, trs(0.45 + 0.05 * (k % 3), 30.0 * k, (-5.0 + 2.0 * k, 0.62 + 0.05 * (k % 3), (-1) ** k * 0.9)))
    env = synthetic_envmap()
    tid = b.add_texture("synthetic_sky_2048x1024", env)
    b.add_ibl(tid, avg_illum=envmap_avg_illum(env))
    return b.build(), cam


def blob_mesh(t, nu=48, nv=24, centre=(0.33, 1.05, 0.35), radius=0.3):
    """A UV sphere whose radius is modulated by two travelling waves of phase t: the stand-in for the reference's
    skinned character (asset/converted_unitychan, absent from the snapshot) in the deformation-renderer sequence.
    Returns (positions [V,3], normals [V,3], indices [2 nu nv, 3])."""
    f = np.float32
    th = (np.arange(nv + 1, dtype=f) / f(nv)) * f(np.pi)
    ph = (np.arange(nu, dtype=f) / f(nu)) * f(2 * np.pi)
    TH, PH = np.meshgrid(th, ph, indexing="ij")
    r = f(radius) * (f(1) + f(0.25) * np.sin(f(3) * PH + f(t)) * np.sin(f(2) * TH) + f(0.15) * np.sin(f(5) * TH - f(2 * t)))
    d = np.stack([np.sin(TH) * np.cos(PH), np.cos(TH), np.sin(TH) * np.sin(PH)], axis=-1).astype(f)
    pos = (np.asarray(centre, f)[None, None, :] + r[..., None] * d).reshape(-1, 3).astype(f)
    nml = d.reshape(-1, 3)
    idx = []
    for i in range(nv):
        for j in range(nu):
            a, b = i * nu + j, i * nu + (j + 1) % nu
            c, e = a + nu, b + nu
            idx.append((a, c, b)); idx.append((b, c, e))
    return pos, nml, np.asarray(idx, np.int64)


def deformable_room(t=0.0, asset_dir=None, **blob):
    """The Cornell box with a deforming mesh in it (the reference's deformation renderer puts a skinned model in a
    room: src/deformation_renderer/main.cpp:262-340, scenedefs.cpp DeformScene).  Returns (builder, blob object id,
    camera): call builder.build() for the scene at phase t, builder.set_mesh_vertices(...) + build() for the next."""
    asset_dir = asset_dir or os.path.join(ASSETS, "cornellbox")
    b = SceneBuilder()
    emit = b.add_material("light", L.MTRL_EMISSIVE, (1.0, 1.0, 1.0))
    objs = b.load_obj(os.path.join(asset_dir, "orig.obj"), create_mtrl=lambda name, mt, clr, a, n: b.add_material(name, mt, clr),
                      separate_objs=True, normal_on_the_fly=True)
    light = b.create_instance(objs[0])
    b.add_area_light(light, b.materials[emit][1]["baseColor"][:3], 200.0)
    for o in objs[1:]:
        b.create_instance(o)
    skin = b.add_material("skin", L.MTRL_GGX, (0.8, 0.45, 0.3), roughness=0.35, ior=1.4)
    pos, nml, idx = blob_mesh(t, **blob)
    oid = b.add_mesh("blob", pos, idx, skin, normals=nml, deformable=True)
    b.create_instance(oid)
    b.set_background((0.0, 0.0, 0.0))
    cam = dict(pos=(0.0, 1.0, 3.0), at=(0.0, 1.0, 0.0), vfov=45.0)
    return b, oid, cam


def toon_ramp(steps=(0.15, 0.45, 0.8, 1.0), width=64, tint=(1.0, 1.0, 1.0)):
    """A 1-D remap texture (ToonParameter::remap_texture): `width` texels in `len(steps)` flat bands."""
    t = np.ones((1, width, 4), np.float32)
    for x in range(width):
        b = steps[min(x * len(steps) // width, len(steps) - 1)]
        t[0, x, :3] = [b * c for c in tint]
    return t


def toon_room(asset_dir=None, target="point", screen_shadow=None, alpha_blocker=False):
    """The Cornell box with NPR materials (material/toon.cpp): the tall box is a Toon material of diffuse type, the short
    box a StylizedBrdf of specular type with a stylised highlight and a rim light, the floor a Toon material of specular type;
    all aim at ONE NPR target light (context::AddNprTargetLight) -- a point light, or the room's area light -- and the
    area light still lights the rest of the room.  `screen_shadow` [h, w]: the screen-space shadow texture of the
    stylized-shadow feature (enabled on the tall box when given)."""
    asset_dir = asset_dir or os.path.join(ASSETS, "cornellbox")
    b = SceneBuilder()
    emit = b.add_material("light", L.MTRL_EMISSIVE, (1.0, 1.0, 1.0))
    ramp = b.add_texture("ramp4", toon_ramp())
    ramp_warm = b.add_texture("ramp3warm", toon_ramp((0.2, 0.6, 1.0), tint=(1.0, 0.85, 0.6)))

    def create_mtrl(name, mtype, clr, albedo, nml):
        if name == "tallBox":
            extra = dict(shadow_enable=1, shadow_threshold=0.9, shadow_offset=0.05, shadow_scale=1.0) if screen_shadow is not None else {}
            return b.add_toon_material(name, (0.9, 0.5, 0.4), target_light_idx=0, remap_texture=ramp, **extra)
        if name == "shortBox":
            return b.add_toon_material(name, (0.4, 0.6, 0.9), stylized=True, toon_type=L.MTRL_SPECULAR, target_light_idx=0,
                                       remap_texture=ramp_warm, roughness=0.3, ior=1.5, stylized_y_min=0.05, stylized_y_max=0.6,
                                       translation_dt=0.1, translation_db=-0.05, scale_t=0.2, scale_b=0.1, split_t=0.05, split_b=0.02,
                                       square_sharp=0.7, square_magnitude=0.3,
                                       rim_enable=1, rim_width=0.4, rim_softness=0.3, rim_color=(0.3, 0.5, 1.0), rim_spread=1.0)
        if name == "floor":
            return b.add_toon_material(name, (0.7, 0.7, 0.6), toon_type=L.MTRL_SPECULAR, target_light_idx=0, remap_texture=-1,
                                       roughness=0.4, ior=1.4, will_receive_shadow=1)
        return b.add_material(name, mtype, clr)

    objs = b.load_obj(os.path.join(asset_dir, "orig.obj"), create_mtrl=create_mtrl, separate_objs=True, normal_on_the_fly=True)
    light = b.create_instance(objs[0])
    lid = b.add_area_light(light, b.materials[emit][1]["baseColor"][:3], 200.0)
    for o in objs[1:]:
        b.create_instance(o)
    if alpha_blocker:
        glass = b.add_material("pane", L.MTRL_DIFFUSE, (0.9, 0.9, 0.9, 0.5))
        b.config.enable_alpha_blending = 1
        q = np.array([[-0.6, 1.5, -0.4], [0.6, 1.5, -0.4], [0.6, 1.5, 0.6], [-0.6, 1.5, 0.6]], np.float32)
        b.create_instance(b.add_mesh("pane", q, [[0, 1, 2], [0, 2, 3]], glass))
    if target == "point":
        l = np.zeros((), L.LIGHT_PARAM)
        l["type"] = L.LIGHT_POINT; l["attrib"] = L.LATTR_SINGULAR
        l["pos"] = [0.3, 1.7, 1.2, 1.0]; l["light_color"] = (1.0, 1.0, 1.0)
        l["innerAngle"] = l["outerAngle"] = np.pi
        l["scale"], l["intensity"] = 1.0, 4.0
        l["arealight_objid"] = -1; l["envmapidx"] = -1
        b.add_npr_target_light(l)
    else:
        b.add_npr_target_light(b.lights[lid])
    if screen_shadow is not None:
        sst = np.zeros(screen_shadow.shape + (4,), np.float32)
        sst[..., 0] = screen_shadow
        b.screen_space_texture = sst
    b.set_background((0.0, 0.0, 0.0))
    cam = dict(pos=(0.0, 1.0, 3.0), at=(0.0, 1.0, 0.0), vfov=45.0)
    return b.build(), cam
