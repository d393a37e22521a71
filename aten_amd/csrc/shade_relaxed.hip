// k_shade under RELAXED floating-point rules, opt-in (atn_set_shade_math): what the reference's own GPU build does to the same
// formulas (src/libidaten/CMakeLists.txt:188 `--use_fast_math`: fused multiply-adds, approximate division and square root, the
// hardware's sin / cos / exp2 / log2 behind sinf / cosf / expf / logf / powf, denormals flushed).  NOT the parity path: the CPU
// renderer is an SSE2 build without FMA and the default kernels (aten_amd.hip) round every operation the way it does; this unit
// exists to put a number on what that costs and for callers who want the frames and not the bits.
// Same sources (device/kernels.hpp: shade_body), other flags (build.py, HIP_UNITS), the transcendental calls redirected below.
#include <hip/hip_runtime.h>

#define ATN_TEMPLATES_ONLY 1
#define sinf __sinf
#define cosf __cosf
#define expf __expf
#define logf __logf
#define powf __powf
#include "../../include/aten_amd.h"
#include "device/relaxed_launch.hpp"

namespace atn {

template <int MS>
__global__ void __launch_bounds__(256) k_shade_relaxed(PathBuffers pb, DevScene sc, FrameParams fp, atn_camera_param cam, int32_t bounce)
{
    shade_body<false, MS>(pb, sc, fp, cam, bounce, SvgfShade{});
}
template <int MS, int WAVES>
__global__ void __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) __launch_bounds__(256) k_shade_relaxed_wn(PathBuffers pb, DevScene sc, FrameParams fp, atn_camera_param cam, int32_t bounce)
{
    shade_body<false, MS>(pb, sc, fp, cam, bounce, SvgfShade{});
}

template <int MS>
static void relaxed_waves(int waves, uint32_t grid, hipStream_t st, const PathBuffers& pb, const DevScene& sc, const FrameParams& fp, const atn_camera_param& cam, int32_t bounce)
{
    const dim3 g(grid), t(256);
    if (waves == 5) hipLaunchKernelGGL((k_shade_relaxed_wn<MS, 5>), g, t, 0, st, pb, sc, fp, cam, bounce);
    else hipLaunchKernelGGL((k_shade_relaxed_wn<MS, 4>), g, t, 0, st, pb, sc, fp, cam, bounce);
}

void relaxed_launch_shade(int material_set, int waves, uint32_t grid, hipStream_t st, const PathBuffers& pb, const DevScene& sc, const FrameParams& fp,
                          const atn_camera_param& cam, int32_t bounce)
{
    const dim3 g(grid), t(256);
    switch (material_set) {
    case kMsCore: relaxed_waves<kMsCore>(waves, grid, st, pb, sc, fp, cam, bounce); break;
    case kMsDisney: relaxed_waves<kMsDisney>(waves, grid, st, pb, sc, fp, cam, bounce); break;
    case kMsAnalytic: relaxed_waves<kMsAnalytic>(waves, grid, st, pb, sc, fp, cam, bounce); break;
    case kMsCarPaint: hipLaunchKernelGGL((k_shade_relaxed<kMsCarPaint>), g, t, 0, st, pb, sc, fp, cam, bounce); break;
    default: hipLaunchKernelGGL((k_shade_relaxed<kMsToon>), g, t, 0, st, pb, sc, fp, cam, bounce); break;
    }
}

} // namespace atn
