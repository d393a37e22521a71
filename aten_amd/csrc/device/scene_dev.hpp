// Device-resident scene: what the kernels read.  Built once by Renderer::UpdateSceneData from
// the caller's flat arrays (include/aten_layout.h); see DESIGN.md "data layout in HBM".
#pragma once
#include "vec.hpp"
#include "../../../include/aten_layout.h"

namespace atn {

// One 48-byte record per BVH node, all node lists concatenated into a single array with ABSOLUTE
// indices, each list re-laid-out in walk (pre-)order so that an inner node's hit link is always
// index + 1.  The walk order -- and therefore every hit/miss decision -- is exactly the
// reference's (threaded_bvh_traverser.h:98-304); only the storage differs.
//
// Links are int32 bit patterns: kLinkEnd (-1) = leave this list; otherwise the BYTE offset of the
// target record in `nodes` (a multiple of 16) with the target's type in the low bits:
// kLinkLeafBit = triangle leaf, kLinkTlasBit = TLAS leaf with a nested tree, 0 = inner node.
//
//   inner    : q0 = {boxmin.xyz, tag}       q1 = {boxmax.xyz, miss link}
//              tag (int bits) = type bits of the NEXT record (offset + 48), i.e. of the implicit hit link
//   tri leaf : q0 = {v0.xyz, triangle id}   q1 = {e1.xyz, next link}   q2 = {e2.xyz, 0}
//              (v0, e1 = v1 - v0, e2 = v2 - v0 of the leaf's triangle: the three dependent
//               gathers node -> TriangleParameter -> 3 vertices become one 48-byte read;
//               e1/e2 are the same IEEE subtractions intersectTriangle performs, done at upload)
//   TLAS leaf: q0 = {objid, w2l_row (index of W2L's first row in `matrices`, or -1), BLAS root link, 0}
//              q1 = {meshid, top hit link, top miss link, 0}
//   dead leaf: a leaf with neither triangle nor nested tree (sphere instance: never tested on this
//              path); typed as an inner record whose tag is kTagDead: always "miss".
//              q0 = {0,0,0, kTagDead}      q1 = {0,0,0, miss link}
constexpr int32_t kLinkEnd = -1;
constexpr int32_t kLinkLeafBit = 1;
constexpr int32_t kLinkTlasBit = 2;
constexpr int32_t kLinkTypeMask = 3;
constexpr uint32_t kLinkOffsetMask = ~15u;
constexpr uint32_t kNodeBytes = 48;
constexpr int32_t kTagDead = 8;

// MaterialParameter reduced to what this path reads (96 B instead of 248 B AoS).
constexpr uint32_t kAttrIdealRefraction = 0x10000u;   // MaterialParameter::isIdealRefraction, folded into attrib at upload
constexpr uint32_t kAttrMaybeAlpha = 0x20000u;        // baseColor.a < 1 or an albedo texel with a < 1 exists: material::isTranslucentByAlpha
                                                      // can be true, so shadow-ray hits on it must evaluate it (set at upload)

struct DevMaterial {
    float4 baseColor;
    int32_t type;
    uint32_t attrib;
    int32_t id;
    int32_t albedoMap;
    int32_t normalMap;
    int32_t roughnessMap;
    float ior, roughness;
    float subsurface, metallic, specular, specularTint;
    float sheen, sheenTint, clearcoat, clearcoatGloss;
};
static_assert(sizeof(DevMaterial) == 80, "DevMaterial");

struct DevTexture {
    uint32_t offset;    // first texel in `texels`
    int32_t width, height;
    int32_t _pad;
};

struct DevScene {
    const float4* nodes;                // DevNodes records, 3 float4 per node
    const atn_triangle_param* tris;     // 32 B each (ids / needNormal / mtrlid / mesh_id)
    const float4* vtx_pos;              // (pos.xyz, u)
    const float4* vtx_nml;              // (nml.xyz, v)
    const atn_object_param* objects;
    const float4* matrices;             // 4 rows per mat4
    const DevMaterial* materials;
    const atn_light_param* lights;
    const float4* texels;
    const DevTexture* textures;
    int32_t n_lights;
    int32_t n_textures;
    int32_t n_materials;
    float bvh_hit_min;
    // background (scene_rendering_config.bg)
    float bg_color[3];
    int32_t envmap_tex_idx;
    float avgIllum;
    float multiplyer;
    int32_t enable_env_map;
    int32_t any_alpha;          // some material carries kAttrMaybeAlpha
    int32_t root_link;                  // typed link of TLAS node 0
    float ibl_scene_radius;             // ImageBasedLight::sample's scene_radius (ibl.h:106-111), precomputed on host
};

} // namespace atn
