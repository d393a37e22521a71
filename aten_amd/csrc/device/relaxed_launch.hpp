// Host-side launcher of the relaxed-math k_shade (csrc/shade_relaxed.hip: same sources, other compiler flags; opt-in through
// atn_set_shade_math).
#pragma once
#include "kernels.hpp"

namespace atn {
void relaxed_launch_shade(int material_set, int waves, uint32_t grid, hipStream_t st, const PathBuffers& pb, const DevScene& sc, const FrameParams& fp,
                          const atn_camera_param& cam, int32_t bounce);
} // namespace atn
