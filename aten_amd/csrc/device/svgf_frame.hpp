// SVGF: the frame descriptor every filter kernel takes, and the helpers they share (see svgf.hpp for the passes).
// Split out of svgf.hpp so that the a-trous kernels can be compiled as a translation unit of their own (svgf_atrous.hpp).
#pragma once
#include "vec.hpp"

namespace atn {

struct SvgfFrame {
    // AOVs of the current / previous frame (svgf_types.h:36-44): normal+depth, albedo+meshid,
    // colour+variance, moments+temporal weight
    float4* nd; float4* am; float4* cv; float4* mt;
    const float4* pnd; const float4* pam; const float4* pcv; const float4* pmt;
    float4* cv_out;         // EstimateVariance writes here: every tap reads the value the pass started with (SVGF)
    float4* atrous[2];      // atrous_clr_variance ping-pong
    float4* tmp;            // temporary_color_buffer
    float4* motion;         // motion_depth_buffer
    const float4* g_nd;     // G-buffer staging written by the path pass (normal+depth, albedo+id); PrepareForDenoise's
    const float4* g_am;     // kernel moves it into the current AOV set.  null = the caller uploaded the AOVs itself.
    float4* primary;        // world position of the bounce-0 hit, w = 1 (0 on a miss): input of the motion pass
    float4* contribs;       // Path.contrib as vec4: contrib.xyz, samples
    float4* out;            // what dst.buffer holds when OnRender returns
    float4* stages;         // optional 3 x [w*h]: the puts after the path, temporal and variance passes
    float w2c[16], prev_w2c[16];
    int32_t width, height;
    uint32_t frame;
    int32_t atrous_iter_cnt;
    float camera_distance;
    int32_t compute_motion;
};

ATN_DEV int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return (v < lo) ? lo : (hi < v) ? hi : v; }     // std::clamp
ATN_DEV float clampf(float v, float lo, float hi) { return (v < lo) ? lo : (hi < v) ? hi : v; }
ATN_DEV float4 div4(const float4& a, float t) { return make_float4(a.x / t, a.y / t, a.z / t, a.w / t); }
ATN_DEV bool svgf_pixel(const SvgfFrame& sf, int32_t& ix, int32_t& iy)
{
    // XCD-aware block -> tile map: consecutive block ids land on consecutive XCDs (b % 8), so give every XCD its own
    // vertical strip of the frame and walk the strips row by row: the +-2*step rows a filter pass re-reads stay in
    // that XCD's 4 MiB L2 instead of being fetched by all eight.
    // (the host rounds gridDim.x up to a multiple of 8; tiles beyond the frame fail the bounds test below)
    const uint32_t gx = gridDim.x;
    const uint32_t b = blockIdx.x + blockIdx.y * gx;
    const uint32_t strip = gx >> 3;
    const uint32_t xcd = b & 7u, local = b >> 3;
    const uint32_t bx = xcd * strip + local % strip, by = local / strip;
    ix = (int32_t)(bx * 8u + (threadIdx.x & 7u));
    iy = (int32_t)(by * 32u + (threadIdx.x >> 3));
    return ix < sf.width && iy < sf.height;
}

// x^128 by seven squarings (powf(x, 128.0f) of the reference: the same value up to a few dozen ulp, far inside the
// frame tolerance, at 7 instead of ~60 instructions)
ATN_DEV float pow128(float x)
{
    float r = x * x; r = r * r; r = r * r; r = r * r; r = r * r; r = r * r; r = r * r;
    return r;
}
// expf as one v_exp_f32 (2^(x * log2 e)): relative error ~ |x| * 1e-7, again far inside the frame tolerance; the filter
// weights it produces only ever multiply colours
ATN_DEV float svgf_exp(float x) { return __expf(x); }

} // namespace atn
