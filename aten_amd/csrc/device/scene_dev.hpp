// Device-resident scene: what the kernels read.  Built once by Renderer::UpdateSceneData from
// the caller's flat arrays (include/aten_layout.h); see DESIGN.md "data layout in HBM".
#pragma once
#include "vec.hpp"
#include "../../../include/aten_layout.h"

namespace atn {

// BVH records.  All node lists live in ONE byte image (DevScene::nodes); a link is the BYTE offset of the target
// record (a multiple of 16) with the target's type in the low bits: kLinkLeafBit = triangle leaf, kLinkTlasBit = TLAS
// leaf with a nested tree, 0 = inner node; kLinkEnd (-1, both type bits set) = leave this list.  Every link that does NOT
// name an inner record also carries the sign bit (kLinkNotInner; offsets are below 2^31): the hot loop's "is this lane on
// an inner node" is ONE signed compare (`link >= 0`) instead of a mask and a compare.  Links are explicit,
// so the walk order -- and therefore every hit/miss decision -- is exactly the reference's
// (threaded_bvh_traverser.h:98-304) whatever the storage order is; a bottom-level list's records lie top levels first, walk (pre-)order below
// (host/scene_upload.hpp: assign_offsets), the top layer's in walk order.
//
//   inner    (32 B): q0 = {boxmin.xyz, hit link}    q1 = {boxmax.xyz, miss link}
//   tri leaf (48 B): q0 = {v0.xyz, triangle id}     q1 = {e1.xyz, next link}   q2 = {e2.xyz, 0}
//              (v0, e1 = v1 - v0, e2 = v2 - v0 of the leaf's triangle: the three dependent
//               gathers node -> TriangleParameter -> 3 vertices become one 48-byte read;
//               e1/e2 are the same IEEE subtractions intersectTriangle performs, done at upload)
//   TLAS leaf (32 B): q0 = {objid, w2l_row (index of W2L's first row in `matrices`, or -1), BLAS root link, flags}
//                     flags bit 0 (kTlasIdentity): W2L is bit for bit the identity matrix, so the ray inside this instance is the SAME
//                     for every such instance -- mat4::applyRay(I, ray), which is not the world ray: the direction is re-normalised --
//                     and the plain walk over an LDS copy computes it once per ray (DevScene::ident_row, traverse.hpp)
//                     q1 = {meshid, top hit link, top miss link, twin}
//                     twin (0 = none): byte distance from the BLAS root record to the root of the list's first ANY-HIT TWIN -- the
//                     same tree threaded in a child order an any-hit walk is expected to finish sooner in (host/anyhit_twin.hpp; an
//                     any-hit walk's answer does not depend on the order); bit 0: eight twins, one per octant of the ray's direction,
//                     each as long as the list (traverse.hpp: anyhit_root).  Rays with stop_t = +inf enter there, no other ray does.
//   dead leaf (32 B): a leaf with neither triangle nor nested tree (sphere instance: never tested on this
//              path); an inner record whose hit link IS its miss link.
constexpr int32_t kLinkEnd = -1;
constexpr int32_t kLinkLeafBit = 1;
constexpr int32_t kLinkTlasBit = 2;
constexpr int32_t kLinkTypeMask = 3;
constexpr int32_t kLinkNotInner = (int32_t)0x80000000u;   // set on leaf / TLAS-leaf links and on kLinkEnd
constexpr int32_t kLinkToLeaf = kLinkNotInner | kLinkLeafBit;
constexpr int32_t kLinkToTlas = kLinkNotInner | kLinkTlasBit;
constexpr uint32_t kLinkOffsetMask = 0x7ffffff0u;
constexpr int32_t kTlasIdentity = 1;
constexpr uint32_t kInnerBytes = 32;
constexpr uint32_t kTriLeafBytes = 48;
constexpr uint32_t kShadeTriQuads = 8;
// Material sets: which BSDFs a k_shade instantiation contains (DevScene::material_set picks the smallest that covers
// the uploaded materials).  Every BSDF in the type switch costs registers in a kernel that runs at 3 waves per SIMD:
// without Disney compiled in, the GGX-only headline scene and the Cornell box gain 2 %.
constexpr int kMsCore = 0;        // Emissive, Diffuse, Specular, GGX
constexpr int kMsDisney = 1;      // + Disney
constexpr int kMsAnalytic = 2;    // + Refraction, Beckman, Oren-Nayar, Velvet, MicrofacetRefraction, Retroreflective
constexpr int kMsCarPaint = 3;    // + CarPaint (flake normals, shared random number)
constexpr int kMsToon = 4;        // + Toon / StylizedBrdf (inline visibility walk)
// A node image of at most this many bytes (Cornell box: 71 nodes = 2.9 KB; instanced props) is copied into LDS by every
// block of the plain walk and ALL its records are read from there (one source, no per-lane selection between sources; an
// LDS copy of only the top of a deep tree -- a "treelet" -- lost to the L1 in every form tried).
constexpr uint32_t kLdsNodesMaxBytes = 32u * 1024u;

// MaterialParameter reduced to what this path reads (96 B instead of 248 B AoS).
constexpr uint32_t kAttrIdealRefraction = 0x10000u;   // MaterialParameter::isIdealRefraction, folded into attrib at upload
constexpr uint32_t kAttrStencilAlways = 0x40000u;     // MaterialParameter::stencil_type == StencilType::ALWAYS (material.h:224-228)
constexpr uint32_t kAttrStencilStencil = 0x80000u;    // ... == StencilType::STENCIL
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

// A texture whose every channel value is EXACTLY k / 255.0f -- or exactly k * (1.0f / 255), aten::Image::Load's form
// (image/image.cpp:76-80) -- for an integer k in 0..255, i.e. an 8-bit image the caller converted with one IEEE operation,
// is stored as packed RGBA8 and converted back with the same operation on fetch -- bit-identical values at a quarter of the bytes (sponza_lod: 50 MB of float4 texels -> 12.5 MB, 32 texels
// per 128-byte line instead of 8).  Anything else (HDR environment maps, filtered images) stays float4.
// Texels are row-major (64-byte sector tiling was measured in r03: k_shade FETCH_SIZE -0.4 %, the lookups are incoherent).
struct DevTexture {
    uint32_t offset;    // first texel in `texels` (format 0) or `texels8` (format 1)
    int32_t width, height;
    int32_t format;     // 0 = float4, 1 = RGBA8 decoded as k / 255.0f, 2 = RGBA8 decoded as k * (1.0f / 255)
};

struct DevScene {
    const float4* nodes;                // byte image of the BVH records (see above)
    const atn_triangle_param* tris;     // 32 B each (ids / needNormal / mtrlid / mesh_id)
    const float4* shade_tris;           // kShadeTriQuads float4 per triangle: everything a HIT needs in one 128-byte line
                                        //   {p0,u0} {p1,u1} {p2,u2} {n0,v0} {n1,v1} {n2,v2} {area, needNormal, mtrlid, mesh_id} {idx0..2, 0}
                                        // (k_pack_shade_tris): one aligned line instead of the dependent gathers
                                        // TriangleParameter -> 3 x position -> 3 x normal, seven lines in the worst case
    const float4* vtx_pos;              // (pos.xyz, u)
    const float4* vtx_nml;              // (nml.xyz, v)
    const atn_object_param* objects;
    const float4* matrices;             // 4 rows per mat4
    const DevMaterial* materials;
    const atn_toon_param* toon;         // one per material (+ the fallback): ToonParameter (material.h:124-161), material set 3
    const atn_light_param* npr_lights;  // context::GetNprTargetLight
    const float* screen_shadow;         // context::screen_space_texture, x channel, [ss_h][ss_w]; null = 1.0
    int32_t n_npr_lights, ss_w, ss_h;
    int32_t enable_shadowray_base_stylized_shadow;
    const float4* carpaint;             // 4 per material: CarPaintMaterialParameter (the union member of MaterialParameter, material.h:163-176)
    const atn_light_param* lights;
    const float4* texels;
    const uint32_t* texels8;            // packed r | g << 8 | b << 16 | a << 24
    const DevTexture* textures;
    int32_t n_lights;
    int32_t planar_lights;              // 1 = light_plane[i].w says "a planar area light on a rigid instance" (host/scene_upload.hpp: planar_area_light);
                                        // cleared by every update that can move a light's vertices or its instance
    const float4* light_plane;          // per light: {the plane's unit normal in world space, 1} or zeros
    float inv_n_lights;                 // 1.0f / (float)n_lights: the light pick's pdf, divided once (0 without lights)
    int32_t n_textures;
    int32_t n_materials;
    float bvh_hit_min;
    // background (scene_rendering_config.bg)
    float bg_color[3];
    int32_t envmap_tex_idx;
    float avgIllum;
    float multiplyer;
    int32_t enable_env_map;
    int32_t any_alpha;          // some material carries kAttrMaybeAlpha or kAttrStencilStencil: a shadow-ray hit may be "ignored"
    int32_t enable_alpha_blending;      // scene_rendering_config.enable_alpha_blending
    int32_t material_set;               // kMsCore .. kMsToon: which k_shade is launched
    int32_t root_link;                  // typed link of TLAS node 0
    float ibl_scene_radius;             // ImageBasedLight::sample's scene_radius (ibl.h:106-111), precomputed on host
    uint32_t node_bytes;                // size of the whole node image (a tree of a few KB is walked from an LDS copy: trace_simple<., ., true>)
    uint32_t mtx_quads;                 // float4 rows in `matrices` (the LDS copy holds them behind the node image)
    int32_t ident_row;                  // w2l_row of one TLAS leaf flagged kTlasIdentity, -1 = none
    // The top layer is ONE leaf (a scene = one instance: sponza, the atrium): its record, so that a walk can start INSIDE the nested
    // tree (walk_start) instead of standing on the leaf through its first burst.  root_direct = 0: walks start at root_link.
    int32_t root_direct, root_objid, root_meshid, root_w2l, root_blas, root_flags;
    int32_t root_twin;                  // that leaf's twin word (TLAS leaf record, above)
    float root_m[12];                   // rows 0..2 of that instance's W2L (root_w2l >= 0): kernel arguments, i.e. scalar registers -- no loads at a refill
    // optional samplers, off by default (they leave the parity path of aten::PathTracing; atn_set_sampling_options):
    const float* ibl_cdf_v;             // ImageBasedLight::preCompute's tables of the environment map (light/ibl.cpp:10-118)
    const float* ibl_cdf_u;             // [ibl_h][ibl_w]
    int32_t ibl_w, ibl_h;
    int32_t ibl_importance;             // 1 = sample the IBL light from those tables (ImageBasedLight::sample, ibl.cpp:180-230)
    int32_t tex_bilinear;               // 1 = texture::AtWithBilinear (image/texture.cpp:77-125) instead of texture::at
};

} // namespace atn
