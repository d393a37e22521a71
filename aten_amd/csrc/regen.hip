// Path regeneration (BASELINE.json north_star "path compaction/regeneration"): the kernels of the pool form of the sample loop
// -- k_regen_begin, k_regen_shade<material set>, k_trace_fused<.., REGEN = true>, k_regen_end (device/kernels.hpp) -- as a translation
// unit of their own, and the launchers aten_amd.hip calls (device/regen_launch.hpp).  What replaces what: the reference's GPU loop
// `for sample { generate; for bounce { hit test; shade; shadow; compact } gather }` (src/libidaten/kernel/pathtracing.cpp:105-138) on
// a ray population that decays bounce by bounce becomes `begin; for stage { trace; shade + epilogue + regenerate }; end` on a pool
// that stays full for a whole burst of samples / progressive frames.
#include <hip/hip_runtime.h>

#define ATN_REGEN_TU 1
#include "../../include/aten_amd.h"
#include "device/regen_launch.hpp"

namespace atn {

void regen_launch_begin(uint32_t grid, hipStream_t st, const PathBuffers& pb, const FrameParams& fp, const atn_camera_param& cam)
{
    hipLaunchKernelGGL(k_regen_begin, dim3(grid), dim3(256), 0, st, pb, fp, cam);
}

void regen_launch_compact(uint32_t grid, hipStream_t st, const PathBuffers& pb, int32_t stage, uint32_t chunk_size, uint32_t* group_counts_next, uint32_t n_groups)
{
    hipLaunchKernelGGL(k_regen_compact, dim3(grid), dim3(256), 0, st, pb, stage, chunk_size, group_counts_next, n_groups);
}

void regen_launch_flush(uint32_t grid, hipStream_t st, const PathBuffers& pb, const FrameParams& fp, const RegenOut& ro)
{
    hipLaunchKernelGGL(k_regen_flush, dim3(grid), dim3(256), 0, st, pb, fp, ro);
}

void regen_launch_end(uint32_t grid, hipStream_t st, const PathBuffers& pb, const FrameParams& fp, const RegenOut& ro)
{
    hipLaunchKernelGGL(k_regen_end, dim3(grid), dim3(256), 0, st, pb, fp, ro);
}

template <bool REFILL, bool LDSN>
static void trace_alpha(const RegenTraceLaunch& c, hipStream_t st, const PathBuffers& pb, const DevScene& sc, int32_t bs, int32_t bc, int32_t launch)
{
    const dim3 g(c.grid), t(c.block);
    if (c.alpha) hipLaunchKernelGGL((k_trace_fused<REFILL, true, LDSN, true>), g, t, c.lds_bytes, st, pb, sc, bs, bc, launch);
    else hipLaunchKernelGGL((k_trace_fused<REFILL, false, LDSN, true>), g, t, c.lds_bytes, st, pb, sc, bs, bc, launch);
}

void regen_launch_trace(const RegenTraceLaunch& c, hipStream_t st, const PathBuffers& pb, const DevScene& sc, int32_t bs, int32_t bc, int32_t launch)
{
    if (c.lds_nodes) { if (c.refill) trace_alpha<true, true>(c, st, pb, sc, bs, bc, launch); else trace_alpha<false, true>(c, st, pb, sc, bs, bc, launch); }
    else { if (c.refill) trace_alpha<true, false>(c, st, pb, sc, bs, bc, launch); else trace_alpha<false, false>(c, st, pb, sc, bs, bc, launch); }
}

template <int MS>
static void shade_waves(int waves, uint32_t grid, hipStream_t st, const PathBuffers& pb, const DevScene& sc, const FrameParams& fp,
                        const atn_camera_param& cam, int32_t stage, const RegenOut& ro)
{
    const dim3 g(grid), t(256);
    if (waves == 5) hipLaunchKernelGGL((k_regen_shade_wn<MS, 5>), g, t, 0, st, pb, sc, fp, cam, stage, ro);
    else if (waves == 4) hipLaunchKernelGGL((k_regen_shade_wn<MS, 4>), g, t, 0, st, pb, sc, fp, cam, stage, ro);
    else hipLaunchKernelGGL((k_regen_shade<MS>), g, t, 0, st, pb, sc, fp, cam, stage, ro);
}

void regen_launch_shade(int material_set, int waves, uint32_t grid, hipStream_t st, const PathBuffers& pb, const DevScene& sc, const FrameParams& fp,
                        const atn_camera_param& cam, int32_t stage, const RegenOut& ro)
{
    const dim3 g(grid), t(256);
    switch (material_set) {
    case kMsCore: shade_waves<kMsCore>(waves, grid, st, pb, sc, fp, cam, stage, ro); break;
    case kMsDisney: shade_waves<kMsDisney>(waves, grid, st, pb, sc, fp, cam, stage, ro); break;
    case kMsAnalytic: shade_waves<kMsAnalytic>(waves, grid, st, pb, sc, fp, cam, stage, ro); break;
    case kMsCarPaint: hipLaunchKernelGGL((k_regen_shade<kMsCarPaint>), g, t, 0, st, pb, sc, fp, cam, stage, ro); break;
    default: hipLaunchKernelGGL((k_regen_shade<kMsToon>), g, t, 0, st, pb, sc, fp, cam, stage, ro); break;
    }
}

} // namespace atn
