// Host-side launchers of the path-regeneration kernels.  The kernels are compiled in a translation unit of their own
// (csrc/regen.hip -- in parallel with aten_amd.hip, whose kernels they leave byte for byte as they were); aten_amd.hip
// (PathTracing::run_regen) reaches them through these functions.
#pragma once
#include "kernels.hpp"

namespace atn {

struct RegenTraceLaunch {
    bool refill;        // the persistent lane-refilling walk (deep trees) / the plain walk
    bool alpha;         // DevScene::any_alpha: shadow rays may have to look behind ignored surfaces
    bool lds_nodes;     // the walk over an LDS copy of the node image (small scenes)
    uint32_t grid, block, lds_bytes;
};

void regen_launch_begin(uint32_t grid, hipStream_t st, const PathBuffers& pb, const FrameParams& fp, const atn_camera_param& cam);
// shadow rays of stage bs (< 0: none) + closest-hit rays of stage bc (< 0: none); `launch` indexes the job-fetch cursor
void regen_launch_trace(const RegenTraceLaunch& cfg, hipStream_t st, const PathBuffers& pb, const DevScene& sc, int32_t bs, int32_t bc, int32_t launch);
// waves: 0 = the compiler's allocation, 4 / 5 = held to that many waves per SIMD (the three smaller material sets only)
void regen_launch_shade(int material_set, int waves, uint32_t grid, hipStream_t st, const PathBuffers& pb, const DevScene& sc, const FrameParams& fp,
                        const atn_camera_param& cam, int32_t stage, const RegenOut& ro);
// the stable compaction in front of trace(stage): regions written by shade(stage - 1) (by regen_launch_begin for stage 0) -> dense queues
void regen_launch_compact(uint32_t grid, hipStream_t st, const PathBuffers& pb, int32_t stage, uint32_t chunk_size, uint32_t* group_counts_next, uint32_t n_groups);
// the pending epilogues of retired slots (per slot), in front of regen_launch_end (per pixel)
void regen_launch_flush(uint32_t grid, hipStream_t st, const PathBuffers& pb, const FrameParams& fp, const RegenOut& ro);
void regen_launch_end(uint32_t grid, hipStream_t st, const PathBuffers& pb, const FrameParams& fp, const RegenOut& ro);

} // namespace atn
