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
