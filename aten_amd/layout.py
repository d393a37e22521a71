"""numpy / ctypes mirrors of include/aten_layout.h.

Every dtype's itemsize is asserted against the byte sizes of the reference structs
(SURVEY.md section 8(a)/(b)); field names follow the reference's own
(src/libaten/geometry/geomparam.h, material/material.h, light/light_parameter.h,
camera/camera.h, accelerator/threaded_bvh.h, scene/hit_parameter.h).
"""
import ctypes as C

import numpy as np

f4, i4, u4 = np.float32, np.int32, np.uint32

VEC4 = np.dtype([("x", f4), ("y", f4), ("z", f4), ("w", f4)])
MAT4 = np.dtype((f4, (4, 4)))

BVH_NODE = np.dtype([
    ("boxmin", f4, 3), ("hit", f4),
    ("boxmax", f4, 3), ("miss", f4),
    ("f0", f4), ("f1", f4), ("f2", f4), ("f3", f4),
])

OBJECT_PARAM = np.dtype([
    ("type", i4), ("area", f4), ("object_id", i4), ("mtx_id", i4),
    ("triangle_id", i4), ("triangle_num", i4), ("light_id", i4), ("_pad0", i4),
    ("sphere_center", f4, 3), ("sphere_radius", f4), ("sphere_mtrl_id", i4), ("_pad1", i4, 3),
])

TRIANGLE_PARAM = np.dtype([
    ("idx", i4, 3), ("_pad", f4),
    ("area", f4), ("needNormal", i4), ("mtrlid", i4), ("mesh_id", i4),
])

STANDARD_FIELDS = ["ior", "roughness", "shininess", "subsurface", "metallic", "specular",
                   "specularTint", "anisotropic", "sheen", "sheenTint", "clearcoat", "clearcoatGloss"]

# aten::ToonParameter, 100 B (material.h:124-161)
TOON_PARAM = np.dtype([
    ("target_light_idx", i4), ("remap_texture", i4), ("stylized_y_min", f4), ("stylized_y_max", f4),
    ("toon_type", i4), ("will_receive_shadow", np.uint8), ("_pad0", np.uint8, 3),
    ("translation_dt", f4), ("translation_db", f4), ("scale_t", f4), ("scale_b", f4),
    ("split_t", f4), ("split_b", f4), ("square_sharp", f4), ("square_magnitude", f4),
    ("rim_width", f4), ("rim_softness", f4), ("rim_enable", np.uint8), ("_pad1", np.uint8, 3),
    ("rim_color", f4, 3), ("rim_spread", f4),
    ("shadow_threshold", f4), ("shadow_offset", f4), ("shadow_scale", f4), ("shadow_enable", np.uint8), ("_pad2", np.uint8, 3),
])
assert TOON_PARAM.itemsize == 100

MATERIAL_PARAM = np.dtype([
    ("baseColor", f4, 4), ("type", i4), ("attrib", u4),
    ("id", np.uint16), ("isIdealRefraction", np.uint8), ("is_medium", np.uint8),
    ("albedoMap", i4), ("normalMap", i4), ("roughnessMap", i4), ("stencil_type", i4),
    ("standard", f4, 12), ("_union_tail", f4, 4),
    ("medium", f4, 8), ("toon", TOON_PARAM), ("feature_line", np.uint8, 8),
])

LIGHT_PARAM = np.dtype([
    ("pos", f4, 4), ("dir", f4, 4), ("type", i4), ("light_color", f4, 3),
    ("innerAngle", f4), ("outerAngle", f4), ("attrib", u4), ("scale", f4),
    ("intensity", f4), ("arealight_objid", i4), ("envmapidx", i4), ("_pad", i4),
])

CAMERA_PARAM = np.dtype([
    ("origin", f4, 3), ("lookat", f4, 3), ("aspect", f4), ("center", f4, 3),
    ("u", f4, 3), ("v", f4, 3), ("dir", f4, 3), ("right", f4, 3), ("up", f4, 3),
    ("dist", f4), ("vfov", f4), ("width", i4), ("height", i4), ("znear", f4), ("zfar", f4),
])

INTERSECTION = np.dtype([
    ("t", f4), ("objid", i4), ("mtrlid", i4), ("meshid", i4),
    ("tri_id", i4), ("a", f4), ("b", f4), ("isVoxel", i4),
])

RAY = np.dtype([("org", f4, 3), ("dir", f4, 3)])

assert VEC4.itemsize == 16 and MAT4.itemsize == 64
assert BVH_NODE.itemsize == 48
assert OBJECT_PARAM.itemsize == 64
assert TRIANGLE_PARAM.itemsize == 32
assert MATERIAL_PARAM.itemsize == 248 and MATERIAL_PARAM.fields["standard"][1] == 44
assert MATERIAL_PARAM.fields["medium"][1] == 108 and MATERIAL_PARAM.fields["toon"][1] == 140
assert LIGHT_PARAM.itemsize == 80 and LIGHT_PARAM.fields["attrib"][1] == 56
assert CAMERA_PARAM.itemsize == 124
assert INTERSECTION.itemsize == 32 and RAY.itemsize == 24

# enums (include/aten_layout.h)
OBJ_POLYGONS, OBJ_INSTANCE, OBJ_SPHERE = 0, 1, 2
(MTRL_EMISSIVE, MTRL_DIFFUSE, MTRL_OREN_NAYAR, MTRL_SPECULAR, MTRL_REFRACTION, MTRL_GGX,
 MTRL_BECKMAN, MTRL_VELVET, MTRL_MICROFACET_REFRACTION, MTRL_RETROREFLECTIVE, MTRL_CARPAINT,
 MTRL_DISNEY, MTRL_TOON, MTRL_STYLIZED) = range(14)
MTRL_TOON_SPECULAR = 16
ATTR_EMISSIVE, ATTR_SINGULAR, ATTR_TRANSLUCENT, ATTR_GLOSSY = 1, 2, 4, 8
# aten::MaterialAttribute* constants, src/libaten/material/material.h:34-39
MTRL_ATTRIB = {
    MTRL_EMISSIVE: ATTR_EMISSIVE,
    MTRL_DIFFUSE: 0,
    MTRL_SPECULAR: ATTR_SINGULAR | ATTR_GLOSSY,
    MTRL_GGX: ATTR_GLOSSY,
    MTRL_BECKMAN: ATTR_GLOSSY,
    MTRL_OREN_NAYAR: 0,
    MTRL_VELVET: ATTR_GLOSSY,
    MTRL_MICROFACET_REFRACTION: ATTR_SINGULAR | ATTR_TRANSLUCENT | ATTR_GLOSSY,
    MTRL_REFRACTION: ATTR_SINGULAR | ATTR_TRANSLUCENT | ATTR_GLOSSY,
    MTRL_DISNEY: ATTR_GLOSSY,
    MTRL_RETROREFLECTIVE: ATTR_GLOSSY,      # MaterialAttributeMicrofacet (retroreflective.h:26-31)
    MTRL_CARPAINT: ATTR_GLOSSY,             # MaterialAttributeMicrofacet (car_paint.h:16-44)
}
LIGHT_AREA, LIGHT_IBL, LIGHT_DIRECTION, LIGHT_POINT, LIGHT_SPOT = range(5)
LATTR_SINGULAR, LATTR_INFINITE, LATTR_IBL = 1, 2, 4


class Background(C.Structure):
    _fields_ = [("bg_color", C.c_float * 3), ("envmap_tex_idx", C.c_int32), ("avgIllum", C.c_float),
                ("multiplyer", C.c_float), ("enable_env_map", C.c_uint8), ("_pad", C.c_uint8 * 3)]


class SceneRenderingConfig(C.Structure):
    _fields_ = [("enable_alpha_blending", C.c_uint8), ("_pad0", C.c_uint8 * 3),
                ("feature_line", C.c_uint8 * 28), ("bvh_hit_min", C.c_float),
                ("epsilon_bias", C.c_float), ("bg", Background)]


class TextureDesc(C.Structure):
    _fields_ = [("texels", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32)]


class BvhList(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("count", C.c_uint32), ("_pad", C.c_uint32)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("objects", C.c_void_p), ("n_objects", C.c_uint32), ("_p0", C.c_uint32),
        ("matrices", C.c_void_p), ("n_matrices", C.c_uint32), ("_p1", C.c_uint32),
        ("materials", C.c_void_p), ("n_materials", C.c_uint32), ("_p2", C.c_uint32),
        ("lights", C.c_void_p), ("n_lights", C.c_uint32), ("_p3", C.c_uint32),
        ("triangles", C.c_void_p), ("n_triangles", C.c_uint32), ("_p4", C.c_uint32),
        ("vtx_pos", C.c_void_p), ("vtx_nml", C.c_void_p), ("n_vertices", C.c_uint32), ("_p5", C.c_uint32),
        ("bvh_lists", C.c_void_p), ("n_bvh_lists", C.c_uint32), ("_p6", C.c_uint32),
        ("textures", C.c_void_p), ("n_textures", C.c_uint32), ("_p7", C.c_uint32),
        ("config", SceneRenderingConfig),
        ("scene_bbox_min", C.c_float * 3), ("scene_bbox_max", C.c_float * 3),
        ("npr_target_lights", C.c_void_p), ("n_npr_target_lights", C.c_uint32),
        ("enable_shadowray_base_stylized_shadow", C.c_int32), ("screen_space_texture", TextureDesc),
    ]


assert C.sizeof(SceneRenderingConfig) == 68
assert C.sizeof(Background) == 28


def ptr(a):
    """void* of a C-contiguous numpy array (None for empty)."""
    if a is None or a.size == 0:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data
