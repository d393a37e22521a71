/*
 * aten_amd_scene.h -- host-only (no GPU) C-ABI for producing the BVH arrays the integrator
 * consumes.  This replaces, for callers that do not link libaten, what
 *   aten::sbvh::onBuild / sbvh::convert       (src/libaten/accelerator/sbvh.cpp:190-421,827-950)
 *   aten::ThreadedBVH::build / setOrder       (src/libaten/accelerator/threaded_bvh.cpp:178-357)
 * produce: vectors of 48-byte threaded nodes (hit/miss links, one triangle per leaf).
 *
 * The tree TOPOLOGY is ours (a split BVH: SAH object splits against spatial splits with triangle clipping and reference
 * unsplitting, csrc/host/bvh_builder.cpp), the node FORMAT is the reference's; closest-hit results do not depend on
 * topology except for exact-t ties.
 */
#ifndef ATEN_AMD_SCENE_H_
#define ATEN_AMD_SCENE_H_

#include "aten_layout.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Version of THIS header's entry points (libaten_amd_scene.so; atn_abi_version covers libaten_amd.so only).  Bumped when an
 * existing signature or struct changes; a binding checks it once after loading.  2 = atns_obj_register / atns_obj_copy with
 * counts and capacities, atns_build_blas_opt. */
#define ATNS_ABI_VERSION 2u
uint32_t atns_abi_version(void);

/* Bottom-level tree over the triangles tri_ids[0..n_tris) of `tris` (global ids are written
 * into the leaves, like sbvh::convert's `ref.triid + m_offsetTriIdx`, sbvh.cpp:897).
 * Nodes come out in depth-first pre-order: an inner node's hit link is always index+1.
 * *out_nodes is malloc'ed; release with atns_free.  Returns 0 or a negative error code.
 * This entry builds with OBJECT splits only -- exactly one leaf per triangle and n_tris - 1 inner nodes -- which is the shape
 * atn_lbvh_rebuild_list requires of a list it rebuilds in place (deforming meshes).  Spatial splits (duplicated references, better
 * trees for static meshes) come with atns_build_blas_opt, whose default options switch them on; a tree with duplicated references
 * cannot be rebuilt in place (ATN_ERR_UNSUPPORTED): pass spatial_splits = 0 for meshes that deform. */
int atns_build_blas(const atn_vec4* vtx_pos, const atn_triangle_param* tris,
                    const uint32_t* tri_ids, uint32_t n_tris,
                    atn_bvh_node** out_nodes, uint32_t* out_count,
                    float out_bbox_min[3], float out_bbox_max[3]);

/* The knobs of the bottom-level builder.  atns_bvh_default_options fills in what atns_build_blas_opt(options = NULL) uses. */
enum {
    ATNS_ORDER_AS_SPLIT = 0,      /* lower side of the split plane first (what sbvh::onBuild does, sbvh.cpp:386-404) */
    ATNS_ORDER_AREA = 1,          /* child with the larger surface area first */
    ATNS_ORDER_AREA_SMALL = 2,    /* ... smaller ... */
    ATNS_ORDER_COUNT = 3,         /* child with more references first */
    ATNS_ORDER_COUNT_SMALL = 4,   /* ... fewer ... */
    ATNS_ORDER_NEAR_POINT = 5     /* child whose box is nearer to order_point first: where most rays start (a viewer
                                     position, the middle of a room); default point = area-weighted centroid of the mesh */
};
typedef struct atns_bvh_options {
    int32_t spatial_splits;       /* 0 = object splits only */
    float spatial_alpha;          /* look at a spatial split when overlap(children) / area(root) >= this (sbvh.cpp:232: 1e-5) */
    int32_t object_bins;          /* bins of the binned object split (nodes of sweep_below references and more) */
    int32_t spatial_bins;         /* bins of the chopped-reference spatial split */
    int32_t sweep_below;          /* exact sweep over sorted centroids for nodes with fewer references than this */
    int32_t child_order;          /* ATNS_ORDER_*: which child the fixed-order threaded walk enters first */
    float max_refs_factor;        /* duplication budget: references <= this * triangles (then object splits only) */
    float order_point[3];
    int32_t order_point_given;    /* 0 = ignore order_point, use the centroid */
    int32_t reinsert_iterations;  /* rounds of the insertion-based optimisation after the build (0 = off): nodes whose box is
                                     large for what their children need are taken out and their subtrees re-inserted where
                                     they add the least area */
    float reinsert_batch;         /* share of the inner nodes a round works on */
} atns_bvh_options;
typedef struct atns_bvh_stats {
    uint32_t n_nodes, n_leaves, n_spatial_splits, n_reinsertions;
    float sah_cost;               /* sum over nodes of area(node) / area(root): expected box tests of a random long ray */
} atns_bvh_stats;
void atns_bvh_default_options(atns_bvh_options* out);
/* atns_build_blas with explicit options (NULL = defaults) and optional statistics. */
int atns_build_blas_opt(const atn_vec4* vtx_pos, const atn_triangle_param* tris,
                        const uint32_t* tri_ids, uint32_t n_tris, const atns_bvh_options* options,
                        atn_bvh_node** out_nodes, uint32_t* out_count,
                        float out_bbox_min[3], float out_bbox_max[3], atns_bvh_stats* out_stats);

/* The two post passes of the builder -- insertion-based optimisation, then the child order -- on a threaded list somebody else
 * built (a bottom-level tree imported from a reference-written .sbvh): the same boxes and the same leaves (their four payload
 * floats are carried over; the voxel-LOD payload of inner nodes, never read on this path, is dropped), re-arranged and
 * re-threaded in depth-first pre-order.  options: child_order / order_point / reinsert_* are used (NULL = defaults; without
 * order_point the middle of the root box).  Returns 0, -1 (null / empty), -3 (memory), -4 (`nodes` is not a threaded binary tree). */
int atns_optimize_nodes(const atn_bvh_node* nodes, uint32_t count, const atns_bvh_options* options,
                        atn_bvh_node** out_nodes, uint32_t* out_count, atns_bvh_stats* out_stats);

/* The second threading the upload gives a bottom-level list for its shadow rays (csrc/host/anyhit_twin.hpp): the same tree -- boxes,
 * leaves, parent-child relations -- with the two children of an inner node in the order an ANY-hit walk is expected to finish
 * sooner in (surface-area model); out_cost_*: the model's expected cost of such a walk in the list as given and in the twin.  The
 * answer of an any-hit walk does not depend on the order, so the integrator may use it without changing a result; exported for
 * tools and tests.  *out_nodes: `count` nodes, malloc'ed (atns_free).  Returns 0, -1 (null), -3 (memory), -4 (`nodes` is not a binary
 * tree in pre-order along its hit links: such a list gets no twin). */
int atns_anyhit_twin(const atn_bvh_node* nodes, uint32_t count, atn_bvh_node** out_nodes, double* out_cost_as_given, double* out_cost_twin);

/* Top-level tree over instances.  boxes: n * {min.xyz, max.xyz} (world space, already
 * transformed like aabb::transform in threaded_bvh.cpp:203-204); object_ids: transformable index
 * of each instance; blas_list_ids: index of the instance's node list (>= 1), or -1 for "no
 * external tree" (sphere leaves, never hit on this path); mesh_ids: TLAS meshid field. */
int atns_build_tlas(const float* boxes, const int32_t* object_ids, const int32_t* blas_list_ids,
                    const int32_t* mesh_ids, uint32_t n,
                    atn_bvh_node** out_nodes, uint32_t* out_count);

void atns_free(void* p);

/* The camera block aten::PinholeCamera::CreateCameraParam computes (src/libaten/camera/pinhole.cpp:34-75),
 * for callers that do not link libaten.  Returns 0, or -1 on a null / non-positive argument. */
int atns_create_camera(atn_camera_param* out, const float origin[3], const float lookat[3], const float up[3],
                       float vfov, float z_near, float z_far, int32_t width, int32_t height);

/* Sanity walk of a node list: every link in range or -1, pre-order reachability of all leaves.
 * Returns number of leaves reached by following hit links only, or negative on a bad link. */
int64_t atns_validate_nodes(const atn_bvh_node* nodes, uint32_t count);

/* ---- scene ingestion: what libatenscene does between a file and aten::context -----------------------
 * Wavefront OBJ / MTL with the REGISTRATION rules of aten::ObjLoader::Load (src/libatenscene/ObjLoader.cpp:95-461): one
 * vertex per face corner in file order, uv.z flags, a TriangleGroupMesh per run of equal material ids inside a shape,
 * needNormal per triangle, PolygonObject partition (one per shape, or one per file with emissive groups split out).
 * tinyobjloader is absent from the reference snapshot: polygons are triangulated as a plain fan (0,1,2), (0,2,3), ...
 * Two steps, because ObjLoader's object partition depends on the TYPE of the material its callback created:
 *   atns_obj_open -> materials (name, Kd, Ke, map_Kd, map_bump) -> the caller creates / finds its materials
 *   atns_obj_register(..., which of them are emissive) -> vertices, triangles, meshes, objects */
typedef struct atns_obj atns_obj;
typedef struct atns_obj_material_info {
    const char* name;               /* owned by the handle */
    const char* diffuse_texname;    /* map_Kd, "" if none */
    const char* bump_texname;       /* map_bump / map_Bump / bump, "" if none */
    float diffuse[3];               /* Kd (default 1 1 1) */
    float emission[3];              /* Ke (default 0 0 0) */
} atns_obj_material_info;
typedef struct atns_obj_triangle {
    uint32_t idx[3];                /* TriangleParameter::idx: vertex indices, first_vertex-based */
    int32_t need_normal;            /* TriangleParameter::needNormal (ObjLoader.cpp:387-393) */
    int32_t mesh;                   /* index into the mesh array */
} atns_obj_triangle;
typedef struct atns_obj_mesh {      /* one aten::TriangleGroupMesh */
    int32_t mtl;                    /* index into the OBJ's material list, -1 = no usemtl */
    uint32_t mesh_id;               /* first_mesh_id + creation order */
    uint32_t first_triangle, n_triangles;
    int32_t object;                 /* index into the object array */
    int32_t shape;
} atns_obj_mesh;
typedef struct atns_obj_object {    /* one aten::PolygonObject, in creation order */
    uint32_t first_mesh;            /* unused (an object's meshes are those whose `object` names it, in mesh order) */
    uint32_t n_meshes;
    int32_t shape;                  /* the shape that named it (ObjLoader's create_obj_functor) */
    int32_t is_emissive_split;      /* 1 = split out because its material is emissive (ObjLoader.cpp:262-283,425-437) */
    int32_t return_order;           /* position in the vector ObjLoader::Load returns (emissive objects as they appear, the
                                       file's own object last), -1 = created but never returned */
} atns_obj_object;
int atns_obj_open(const char* path, atns_obj** out);                    /* 0, or negative: -3 = unreadable / bad indices */
void atns_obj_close(atns_obj* h);
uint32_t atns_obj_material_count(const atns_obj* h);
int atns_obj_material(const atns_obj* h, uint32_t i, atns_obj_material_info* out);
uint32_t atns_obj_shape_count(const atns_obj* h);
const char* atns_obj_shape_name(const atns_obj* h, uint32_t i);
/* first_vertex = ctxt.GetVertexNum() before the load; separate_objs = will_register_shape_as_separate_obj;
 * normal_on_the_fly = need_compute_normal_on_the_fly; mtl_is_emissive[i] != 0 <=> OBJ material i resolved to an Emissive
 * material (n_mtl_is_emissive entries, at least atns_obj_material_count; default_is_emissive: the same for faces without
 * usemtl).  Returns 0, -1 (null handle), -2 (out of memory: nothing throws across this boundary), -4 (array too short). */
int atns_obj_register(atns_obj* h, uint32_t first_vertex, uint32_t first_mesh_id, int32_t separate_objs, int32_t normal_on_the_fly,
                      const uint8_t* mtl_is_emissive, uint32_t n_mtl_is_emissive, uint8_t default_is_emissive);
uint32_t atns_obj_vertex_count(const atns_obj* h);
uint32_t atns_obj_triangle_count(const atns_obj* h);
uint32_t atns_obj_mesh_count(const atns_obj* h);
uint32_t atns_obj_object_count(const atns_obj* h);
/* copies what atns_obj_register produced into caller arrays of the given CAPACITIES (elements; -4 if one is smaller than
 * the count above, nothing is written then); any pointer may be NULL */
int atns_obj_copy(const atns_obj* h, atn_vec4* vtx_pos, atn_vec4* vtx_nml, uint32_t cap_vertices, atns_obj_triangle* tris, uint32_t cap_triangles,
                  atns_obj_mesh* meshes, uint32_t cap_meshes, atns_obj_object* objects, uint32_t cap_objects);

/* The material XML aten::MaterialLoader reads (src/libatenscene/MaterialLoader.cpp:82-218):
 *   <root><material><name>..</name><type>..</type><baseColor>r g b</baseColor><ior>..</ior><albedoMap>file</albedoMap>..
 * kind: 0 = vec3 (value[0..2]), 1 = texture file name (text), 2 = float (value[0]), -1 = not in MaterialLoader's table
 * (skipped there).  A repeated name drops the later material; a missing type is "Diffuse". */
typedef struct atns_mtrlxml atns_mtrlxml;
typedef struct atns_mtrlxml_param_info { const char* name; const char* text; int32_t kind; float value[3]; } atns_mtrlxml_param_info;
int atns_mtrlxml_open(const char* path, atns_mtrlxml** out);            /* -3 unreadable, -4 not that format (no <root>, ...) */
void atns_mtrlxml_close(atns_mtrlxml* h);
uint32_t atns_mtrlxml_count(const atns_mtrlxml* h);
const char* atns_mtrlxml_name(const atns_mtrlxml* h, uint32_t i);
const char* atns_mtrlxml_type(const atns_mtrlxml* h, uint32_t i);
uint32_t atns_mtrlxml_param_count(const atns_mtrlxml* h, uint32_t i);
int atns_mtrlxml_param(const atns_mtrlxml* h, uint32_t i, uint32_t k, atns_mtrlxml_param_info* out);

#ifdef __cplusplus
}
#endif
#endif
