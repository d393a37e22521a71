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


def sponza_lod(asset_dir=None, mtype=L.MTRL_GGX, ibl=True, use_sbvh=True, textures=True):
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

    objs = b.load_obj(os.path.join(asset_dir, "sponza_lod.obj"), create_mtrl=create_mtrl)
    if use_sbvh:
        b.import_sbvh(objs[0], os.path.join(asset_dir, "sponza_lod.sbvh"))
    b.create_instance(objs[0])
    if ibl:
        env = synthetic_envmap()
        tid = b.add_texture("synthetic_sky_2048x1024", env)
        b.add_ibl(tid, avg_illum=envmap_avg_illum(env))
    else:
        b.set_background((1.0, 1.0, 1.0))
    cam = dict(pos=(0.0, 1.0, 3.0), at=(0.0, 1.0, 0.0), vfov=45.0)         # scenedefs.cpp:847-860
    return b.build(), cam


def cornell_box_variant(lights="area", move_boxes=True, asset_dir=None):
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
