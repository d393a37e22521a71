// The SVGF a-trous kernels as a translation unit of their own: compiled WITH the SLP vectoriser (packed fp32), unlike
// aten_amd.hip (device/svgf_atrous.hpp says why).  aten_amd.hip launches them through the prototypes in device/svgf.hpp.
#include "device/svgf_atrous.hpp"
