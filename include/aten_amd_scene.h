/*
 * aten_amd_scene.h -- host-only (no GPU) C-ABI for producing the BVH arrays the integrator
 * consumes.  This replaces, for callers that do not link libaten, what
 *   aten::sbvh::onBuild / sbvh::convert       (src/libaten/accelerator/sbvh.cpp:190-421,827-950)
 *   aten::ThreadedBVH::build / setOrder       (src/libaten/accelerator/threaded_bvh.cpp:178-357)
 * produce: vectors of 48-byte threaded nodes (hit/miss links, one triangle per leaf).
 *
 * The tree TOPOLOGY is ours (binned SAH, optional spatial splits), the node FORMAT is the
 * reference's; closest-hit results do not depend on topology except for exact-t ties.
 */
#ifndef ATEN_AMD_SCENE_H_
#define ATEN_AMD_SCENE_H_

#include "aten_layout.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Bottom-level tree over the triangles tri_ids[0..n_tris) of `tris` (global ids are written
 * into the leaves, like sbvh::convert's `ref.triid + m_offsetTriIdx`, sbvh.cpp:897).
 * Nodes come out in depth-first pre-order: an inner node's hit link is always index+1.
 * *out_nodes is malloc'ed; release with atns_free.  Returns 0 or a negative error code. */
int atns_build_blas(const atn_vec4* vtx_pos, const atn_triangle_param* tris,
                    const uint32_t* tri_ids, uint32_t n_tris,
                    atn_bvh_node** out_nodes, uint32_t* out_count,
                    float out_bbox_min[3], float out_bbox_max[3]);

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

#ifdef __cplusplus
}
#endif
#endif
