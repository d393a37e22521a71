/*
 * aten_layout.h -- byte-exact plain-old-data mirrors of the flat scene arrays that
 * nackdai/aten hands to its GPU backend (idaten::Renderer::UpdateSceneData,
 * src/libidaten/kernel/renderer.cpp:12-131).
 *
 * Nothing here depends on glm or on aten's class hierarchy: every struct is a C POD
 * whose size/offsets are pinned with static asserts against the reference layout
 * (sizes measured on the reference headers).
 *
 * An aten application can memcpy its own vectors straight into these types.
 */
#ifndef ATEN_LAYOUT_H_
#define ATEN_LAYOUT_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* aten::vec3 == glm::highp_vec3 : 12 B, align 4 (src/libaten/math/vec3.h:213). */
typedef struct atn_vec3 { float x, y, z; } atn_vec3;
/* aten::vec4 : 16 B (src/libaten/math/vec4.h:9-20). */
typedef struct atn_vec4 { float x, y, z, w; } atn_vec4;
/* aten::mat4 : row-major m[row][col], applied as M*p (src/libaten/math/mat4.h:10-25). */
typedef struct atn_mat4 { float m[4][4]; } atn_mat4;

/* ---- BVH node: aten::ThreadedBvhNode / aten::ThreadedSbvhNode, 48 B -------------
 * src/libaten/accelerator/threaded_bvh.h:13-56, src/libaten/accelerator/sbvh.h:19-62.
 * All links/ids are stored as float; -1 = none.
 *   TLAS node : f0=object_id f1=primid f2=exid(bitfield punned to float) f3=meshid
 *   BLAS node : f0=isleaf    f1=triid  f2=voxeldepth                     f3=mtrlid */
typedef struct atn_bvh_node {
    float boxmin[3];
    float hit;
    float boxmax[3];
    float miss;
    float f0, f1, f2, f3;
} atn_bvh_node;

/* exid bitfield {mainExid:15, lodExid:15, hasLod:1, noExternal:1} (threaded_bvh.h:29-37). */
#define ATN_EXID_MAIN(bits)       ((int32_t)((bits) & 0x7fffu))
#define ATN_EXID_LOD(bits)        ((int32_t)(((bits) >> 15) & 0x7fffu))
#define ATN_EXID_HAS_LOD(bits)    ((int32_t)(((bits) >> 30) & 1u))
#define ATN_EXID_NO_EXTERNAL(bits) ((int32_t)(((bits) >> 31) & 1u))

/* ---- aten::ObjectParameter, 64 B, align 16 (src/libaten/geometry/geomparam.h:23-64) */
enum { ATN_OBJ_POLYGONS = 0, ATN_OBJ_INSTANCE = 1, ATN_OBJ_SPHERE = 2, ATN_OBJ_TYPE_MAX = 3 };
typedef struct atn_object_param {
    int32_t type;
    float area;
    int32_t object_id;
    int32_t mtx_id;
    int32_t triangle_id;
    int32_t triangle_num;
    int32_t light_id;
    int32_t _pad0;
    struct {
        float center[3];
        float radius;
        int32_t mtrl_id;
        int32_t _pad[3];
    } sphere;
} atn_object_param;

/* ---- aten::TriangleParameter, 32 B, align 16 (geomparam.h:66-98) */
typedef struct atn_triangle_param {
    int32_t idx[3];
    float _pad;
    float area;
    int32_t needNormal;
    int32_t mtrlid;
    int32_t mesh_id;
} atn_triangle_param;

/* ---- aten::MaterialParameter, 248 B, align 4 (src/libaten/material/material.h:234-317) */
enum {
    ATN_MTRL_EMISSIVE = 0, ATN_MTRL_DIFFUSE = 1, ATN_MTRL_OREN_NAYAR = 2, ATN_MTRL_SPECULAR = 3,
    ATN_MTRL_REFRACTION = 4, ATN_MTRL_GGX = 5, ATN_MTRL_BECKMAN = 6, ATN_MTRL_VELVET = 7,
    ATN_MTRL_MICROFACET_REFRACTION = 8, ATN_MTRL_RETROREFLECTIVE = 9, ATN_MTRL_CARPAINT = 10,
    ATN_MTRL_DISNEY = 11, ATN_MTRL_TOON = 12, ATN_MTRL_STYLIZED_BRDF = 13, ATN_MTRL_TYPE_MAX = 14,
    ATN_MTRL_VOLUME = 15, ATN_MTRL_TOON_SPECULAR = 16   /* "specialized" types after MaterialTypeMax (material.h:59-63) */
};
/* MaterialAttribute bit-field (material.h:27-32), LSB first. */
#define ATN_MTRL_ATTR_EMISSIVE    0x1u
#define ATN_MTRL_ATTR_SINGULAR    0x2u
#define ATN_MTRL_ATTR_TRANSLUCENT 0x4u
#define ATN_MTRL_ATTR_GLOSSY      0x8u

typedef struct atn_standard_mtrl {   /* StandardMaterialParameter, 48 B (material.h:66-127) */
    float ior, roughness, shininess, subsurface, metallic, specular;
    float specularTint, anisotropic, sheen, sheenTint, clearcoat, clearcoatGloss;
} atn_standard_mtrl;

/* aten::ToonParameter, 100 B (material.h:124-161); bools are one byte followed by padding */
typedef struct atn_toon_param {
    int32_t target_light_idx;      /*   0: index into the scene's NPR target lights (context::GetNprTargetLight), -1 = none */
    int32_t remap_texture;         /*   4 */
    float stylized_y_min;          /*   8 */
    float stylized_y_max;          /*  12 */
    int32_t toon_type;             /*  16: ATN_MTRL_DIFFUSE or ATN_MTRL_SPECULAR */
    uint8_t will_receive_shadow;   /*  20 */
    uint8_t _pad0[3];
    struct {                       /*  24: Hightlight */
        float translation_dt, translation_db, scale_t, scale_b, split_t, split_b, square_sharp, square_magnitude;
    } highlight;
    struct {                       /*  56: RimLight */
        float width, softness;
        uint8_t enable; uint8_t _pad[3];
        float color[3];
        float spread;
    } rim_light;
    struct {                       /*  84: StylizedShadow */
        float threshold, offset, scale;
        uint8_t enable; uint8_t _pad[3];
    } stylized_shadow;
} atn_toon_param;

typedef struct atn_material_param {
    atn_vec4 baseColor;            /*   0 */
    int32_t type;                  /*  16 */
    uint32_t attrib;               /*  20 */
    uint16_t id;                   /*  24 */
    uint8_t isIdealRefraction;     /*  26 */
    uint8_t is_medium;             /*  27 */
    int32_t albedoMap;             /*  28 */
    int32_t normalMap;             /*  32 */
    int32_t roughnessMap;          /*  36 */
    int32_t stencil_type;          /*  40 */
    union {                        /*  44, 64 B */
        atn_standard_mtrl standard;
        float carpaint[16];
    } u;
    float medium[8];               /* 108, MediumParameter 32 B */
    atn_toon_param toon;           /* 140, ToonParameter 100 B */
    uint8_t feature_line[8];       /* 240 */
} atn_material_param;

/* ---- aten::LightParameter, 80 B (src/libaten/light/light_parameter.h:53-94) */
enum { ATN_LIGHT_AREA = 0, ATN_LIGHT_IBL = 1, ATN_LIGHT_DIRECTION = 2, ATN_LIGHT_POINT = 3, ATN_LIGHT_SPOT = 4 };
#define ATN_LIGHT_ATTR_SINGULAR 0x1u
#define ATN_LIGHT_ATTR_INFINITE 0x2u
#define ATN_LIGHT_ATTR_IBL      0x4u
typedef struct atn_light_param {
    atn_vec4 pos;
    atn_vec4 dir;
    int32_t type;
    float light_color[3];
    float innerAngle;
    float outerAngle;
    uint32_t attrib;
    float scale;
    float intensity;
    int32_t arealight_objid;
    int32_t envmapidx;
    int32_t _pad;
} atn_light_param;

/* ---- aten::CameraParameter, 124 B (src/libaten/camera/camera.h:15-36) */
typedef struct atn_camera_param {
    float origin[3];
    float lookat[3];
    float aspect;
    float center[3];
    float u[3];
    float v[3];
    float dir[3];
    float right[3];
    float up[3];
    float dist;
    float vfov;
    int32_t width;
    int32_t height;
    float znear;
    float zfar;
} atn_camera_param;

/* ---- aten::SceneRenderingConfig, 68 B (src/libaten/renderer/scene_rendering_config.h) */
typedef struct atn_background {
    float bg_color[3];
    int32_t envmap_tex_idx;
    float avgIllum;
    float multiplyer;
    uint8_t enable_env_map;
    uint8_t _pad[3];
} atn_background;

typedef struct atn_scene_rendering_config {
    uint8_t enable_alpha_blending;        /*  0 */
    uint8_t _pad0[3];
    uint8_t feature_line[28];             /*  4, FeatureLineConfig (not on this path) */
    float bvh_hit_min;                    /* 32 */
    float epsilon_bias_for_traversing_shadow_ray_in_medium; /* 36 */
    atn_background bg;                    /* 40 */
} atn_scene_rendering_config;

/* ---- aten::Intersection, 32 B (src/libaten/scene/hit_parameter.h:28-64) */
typedef struct atn_intersection {
    float t;
    int32_t objid;
    int32_t mtrlid;
    int32_t meshid;
    int32_t tri_id;     /* union: voxel nml_x */
    float a, b;         /* union: voxel nml_y, nml_z */
    int32_t isVoxel;
} atn_intersection;

/* ---- aten::ray, 24 B (src/libaten/math/ray.h) */
typedef struct atn_ray { float org[3]; float dir[3]; } atn_ray;

/* ---- one texture: aten::texture::colors() is vec4[w*h] (src/libaten/image/texture.h:104-122) */
typedef struct atn_texture_desc {
    const atn_vec4* texels;
    int32_t width;
    int32_t height;
} atn_texture_desc;

/* ---- one BVH node list: scene.getAccel()->getNodes()[k] (src/libaten/accelerator/sbvh.h:167-170) */
typedef struct atn_bvh_list {
    const atn_bvh_node* nodes;
    uint32_t count;
    uint32_t _pad;
} atn_bvh_list;

/*
 * The whole scene as idaten::Renderer::UpdateSceneData receives it
 * (src/libidaten/kernel/renderer.cpp:12-131). All pointers are host memory owned by
 * the caller; atn_upload_scene copies.
 *
 * mtx_id convention: ObjectParameter.mtx_id is the ELEMENT index of the
 * object's local-to-world matrix in `matrices` (world-to-local is at mtx_id+1), -1 = none.
 * This is the host convention of src/libaten/geometry/instance.h:253-269, which is what the
 * CPU traverser reads (src/libaten/accelerator/threaded_bvh_traverser.h:149-155).
 */
typedef struct atn_scene_desc {
    const atn_object_param* objects;     uint32_t n_objects;   uint32_t _p0;
    const atn_mat4* matrices;            uint32_t n_matrices;  uint32_t _p1;
    const atn_material_param* materials; uint32_t n_materials; uint32_t _p2;
    const atn_light_param* lights;       uint32_t n_lights;    uint32_t _p3;
    const atn_triangle_param* triangles; uint32_t n_triangles; uint32_t _p4;
    const atn_vec4* vtx_pos;   /* (pos.xyz, u) : GetExtractedPosAndNmlInVertices, host_scene_context.cpp:88-102 */
    const atn_vec4* vtx_nml;   /* (nml.xyz, v) */
    uint32_t n_vertices;       uint32_t _p5;
    const atn_bvh_list* bvh_lists;       uint32_t n_bvh_lists; uint32_t _p6;  /* [0] = TLAS */
    const atn_texture_desc* textures;    uint32_t n_textures;  uint32_t _p7;
    atn_scene_rendering_config config;
    float scene_bbox_min[3];   /* ctxt.GetSceneBoundingBox() (host_scene_context.h:586-590) */
    float scene_bbox_max[3];
    /* ---- NPR (Toon / StylizedBrdf materials, material/toon.cpp); all may be left zero */
    const atn_light_param* npr_target_lights;   /* context::GetNprTargetLightParameters (host_scene_context.cpp:76-86) */
    uint32_t n_npr_target_lights;
    int32_t enable_shadowray_base_stylized_shadow;   /* context member, host_scene_context.h:50 (default true there) */
    atn_texture_desc screen_space_texture;      /* context::GetScreenSpaceTextureAt reads .x of texel (x, y); texels NULL = 1.0 */
} atn_scene_desc;

#ifdef __cplusplus
} /* extern "C" */

static_assert(sizeof(atn_vec3) == 12, "vec3");
static_assert(sizeof(atn_vec4) == 16, "vec4");
static_assert(sizeof(atn_mat4) == 64, "mat4");
static_assert(sizeof(atn_bvh_node) == 48, "ThreadedBvhNode (sbvh.h:66 static_assert)");
static_assert(sizeof(atn_object_param) == 64, "ObjectParameter");
static_assert(offsetof(atn_object_param, light_id) == 24, "ObjectParameter.light_id");
static_assert(offsetof(atn_object_param, sphere) == 32, "ObjectParameter.sphere");
static_assert(sizeof(atn_triangle_param) == 32, "TriangleParameter");
static_assert(sizeof(atn_standard_mtrl) == 48, "StandardMaterialParameter");
static_assert(sizeof(atn_material_param) == 248, "MaterialParameter");
static_assert(offsetof(atn_material_param, type) == 16, "MaterialParameter.type");
static_assert(offsetof(atn_material_param, id) == 24, "MaterialParameter.id");
static_assert(offsetof(atn_material_param, albedoMap) == 28, "MaterialParameter.albedoMap");
static_assert(offsetof(atn_material_param, stencil_type) == 40, "MaterialParameter.stencil_type");
static_assert(offsetof(atn_material_param, u) == 44, "MaterialParameter.standard");
static_assert(offsetof(atn_material_param, medium) == 108, "MaterialParameter.medium");
static_assert(offsetof(atn_material_param, toon) == 140, "MaterialParameter.toon");
static_assert(sizeof(atn_toon_param) == 100, "ToonParameter");
static_assert(offsetof(atn_toon_param, highlight) == 24 && offsetof(atn_toon_param, rim_light) == 56 && offsetof(atn_toon_param, stylized_shadow) == 84, "ToonParameter members");
static_assert(offsetof(atn_material_param, feature_line) == 240, "MaterialParameter.feature_line");
static_assert(sizeof(atn_light_param) == 80, "LightParameter");
static_assert(offsetof(atn_light_param, type) == 32, "LightParameter.type");
static_assert(offsetof(atn_light_param, attrib) == 56, "LightParameter.attrib");
static_assert(offsetof(atn_light_param, arealight_objid) == 68, "LightParameter.arealight_objid");
static_assert(sizeof(atn_camera_param) == 124, "CameraParameter");
static_assert(offsetof(atn_camera_param, center) == 28, "CameraParameter.center");
static_assert(offsetof(atn_camera_param, dist) == 100, "CameraParameter.dist");
static_assert(sizeof(atn_scene_rendering_config) == 68, "SceneRenderingConfig");
static_assert(offsetof(atn_scene_rendering_config, bvh_hit_min) == 32, "SceneRenderingConfig.bvh_hit_min");
static_assert(offsetof(atn_scene_rendering_config, bg) == 40, "SceneRenderingConfig.bg");
static_assert(sizeof(atn_intersection) == 32, "Intersection");
static_assert(sizeof(atn_ray) == 24, "ray");
#endif

#endif /* ATEN_LAYOUT_H_ */
