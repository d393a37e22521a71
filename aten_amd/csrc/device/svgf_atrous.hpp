// SVGF a-trous wavelet filter kernels (one pixel per thread, and four pixels per thread).
//
// A translation unit of their own (csrc/svgf_atrous.hip), because of ONE compiler switch: the rest of the library is built with
// -fno-slp-vectorize -- the SLP vectoriser pairs neighbouring fp32 operations into v_pk_*_f32, which needs even-aligned register
// pairs and v_mov shuffles and costs the branchy path-tracing kernels 15-25 % of their registers for no gain in issue slots
// (k_shade 128 + spills -> 124 VGPRs, the plain walk 84 -> 63, Cornell 1.43 -> 1.27 ms per frame) --
// while these two kernels are straight-line arithmetic over 24 taps, VALU-bound, and 22 % FASTER with the packed forms
// (k_svgf_atrous4 0.508 vs 0.653 ms per frame).  So they keep the vectoriser.
#pragma once
#include "svgf_frame.hpp"

namespace atn {

// SVGFRenderer::AtrousFilter (svgf.cpp:328-410): ExtractCenterPixel<false>, CheckIfBackgroundPixelForAtrous,
// Exec3x3GaussFilter on the variance, ExecAtrousWaveletFilter, PostProcessForAtrousFilter (svgf_impl.h:558-843)
__global__ void __launch_bounds__(256) k_svgf_atrous(SvgfFrame sf, int32_t iter)
{
    int32_t ix, iy;
    if (!svgf_pixel(sf, ix, iy)) return;
    const int32_t width = sf.width, height = sf.height;
    const int32_t idx = ix + iy * width;
    const int32_t curr = iter & 1, next = 1 - curr;
    const bool is_first = iter == 0, is_final = iter == sf.atrous_iter_cnt - 1;
    const float4* __restrict__ src = is_first ? sf.cv : sf.atrous[curr];
    float4* __restrict__ dst = sf.atrous[next];

    const float4 nml_depth = sf.nd[idx];
    const float4 am = sf.am[idx];
    const float center_depth = nml_depth.w;
    const int32_t center_meshid = (int32_t)am.w;
    const float4 contrib = src[idx];
    const float4 center_color = make_float4(contrib.x, contrib.y, contrib.z, 1.0F);
    const f3 center_normal = mk3(nml_depth);

    if (am.w < 0) {
        dst[idx] = make_float4(center_color.x, center_color.y, center_color.z, 0.0F);
        if (is_final) {
            sf.out[idx] = make_float4(am.x * center_color.x, am.y * center_color.y, am.z * center_color.z, am.w * center_color.w);
            return;
        }
        // not final: the reference's outer optional is engaged but empty and the caller filters the pixel anyway
    }

    // 3x3 Gauss filter of the variance
    float gauss = 0.0F;
    {
        const float k3[3] = { 1.0F / 16.0F, 1.0F / 8.0F, 1.0F / 4.0F };
#pragma unroll
        for (int32_t i = 0; i < 9; i++) {
            const int32_t ox = i % 3 - 1, oy = i / 3 - 1;
            const int32_t xx = clampi(ix + ox, 0, width - 1);
            const int32_t yy = clampi(iy + oy, 0, height - 1);
            const float kk = k3[(ox == 0 ? 1 : 0) + (oy == 0 ? 1 : 0)];
            gauss += kk * src[xx + yy * width].w;
        }
    }

    const float sigmaZ = 1.0F, sigmaL = 4.0F;       // sigmaN = 128: pow128
    const int32_t step_scale = 1 << iter;
    const float sqrt_gauss = sqrtf(gauss);
    const float center_luminance = luminance(center_color.x, center_color.y, center_color.z);
    float4 sumC = center_color;
    float sumV = center_color.w;
    float weight = 1.0F;
    const float pixel_distance_ratio = (center_depth / sf.camera_distance) * (float)height;
    // The two per-tap divisions have denominators that do not depend on the tap (luminance) or only on its length
    // class (depth): one reciprocal each per pixel, then a multiply per tap (<= 1 ulp from the quotient) instead of
    // two IEEE divisions (~20 instructions) per tap.
    const float inv_l = 1.0F / (sigmaL * sqrt_gauss + 0.000001F);
    const float fs = (float)step_scale;
    const float inv_z0 = 1.0F / (sigmaZ * (pixel_distance_ratio * (fs * 1.0F)) + 0.000001F);
    const float inv_z1 = 1.0F / (sigmaZ * (pixel_distance_ratio * (fs * 2.0F)) + 0.000001F);
    const float inv_z2 = 1.0F / (sigmaZ * (pixel_distance_ratio * (fs * 1.41421354F)) + 0.000001F);
    const float inv_z3 = 1.0F / (sigmaZ * (pixel_distance_ratio * (fs * 2.23606801F)) + 0.000001F);
    const float inv_z4 = 1.0F / (sigmaZ * (pixel_distance_ratio * (fs * 2.82842708F)) + 0.000001F);
    // tap tables of svgf_impl.h:693-726, generated: six groups of four; sqrt(sx^2 + sy^2) = 2^iter * sqrt(ox^2 + oy^2)
    // exactly (scaling by a power of two commutes with the correctly rounded square root)
#pragma unroll
    for (int32_t i = 0; i < 24; i++) {
        constexpr int8_t ox[24] = { 1, 0, -1, 0, 2, 0, -2, 0, 1, -1, -1, 1, 1, -1, -1, 1, 2, -2, -2, 2, 2, -2, -2, 2 };
        constexpr int8_t oy[24] = { 0, 1, 0, -1, 0, 2, 0, -2, 1, 1, -1, -1, 2, 2, -2, -2, 1, 1, -1, -1, 2, 2, -2, -2 };
        const float hh = i < 4 ? 2.0F / 3.0F : i < 8 ? 1.0F / 6.0F : i < 12 ? 4.0F / 9.0F : i < 20 ? 1.0F / 9.0F : 1.0F / 36.0F;
        const int32_t sx = ox[i] * step_scale, sy = oy[i] * step_scale;
        const int32_t xx = clampi(ix + sx, 0, width - 1);
        const int32_t yy = clampi(iy + sy, 0, height - 1);
        const float inv_z = i < 4 ? inv_z0 : i < 8 ? inv_z1 : i < 12 ? inv_z2 : i < 20 ? inv_z3 : inv_z4;
        const int32_t qidx = xx + yy * width;
        const float4 q_nd = sf.nd[qidx];
        const int32_t meshid = (int32_t)sf.am[qidx].w;
        const float4 color = src[qidx];
        const float variance = color.w;
        const float lum = luminance(color.x, color.y, color.z);
        const float Wz = (3.0F * fabsf(center_depth - q_nd.w)) * inv_z;
        const float dn = dot(center_normal, mk3(q_nd));
        const float Wn = pow128(0.0F < dn ? dn : 0.0F);
        const float el = svgf_exp(-fabsf(center_luminance - lum) * inv_l);
        const float Wl = 1.0F < el ? 1.0F : el;          // std::min(e, 1.0f)
        const float Wm = meshid == center_meshid ? 1.0F : 0.0F;
        const float W = svgf_exp(-Wl * Wl - Wz) * Wn * Wm * hh;
        sumC = add4(sumC, mul4(W, color));
        sumV += W * W * variance;
        weight += W;
#if ATN_ATROUS1_FENCE
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);    // keep one ring group (12 loads) in flight, not all 72
#endif
    }
    sumC = div4(sumC, weight);
    sumV /= (weight * weight);
    const float4 filtered = make_float4(sumC.x, sumC.y, sumC.z, sumV);

    dst[idx] = filtered;
    if (is_first) {
        const float4 t = sf.tmp[idx];
        sf.tmp[idx] = make_float4(filtered.x, filtered.y, filtered.z, t.w);
    }
    if (is_final) sf.out[idx] = make_float4(am.x * filtered.x, am.y * filtered.y, am.z * filtered.z, am.w * filtered.w);
}

// ------------------------------------------------------------------------------------------------------------------
#ifndef ATN_ATROUS1_FENCE
#define ATN_ATROUS1_FENCE 0
#endif
// The same filter, FOUR pixels per thread (r03).  At a-trous level i the 24 taps of a pixel sit on a 5 x 5 lattice of
// pitch s = 2^i around it; the pixels (x0 + a s, y0 + b s), a, b in {0, 1}, share 20 of their 25 lattice points, so a
// thread that owns that 2 x 2 group reads a 6 x 6 lattice (36 points: normal+depth, colour+variance, mesh id) for four
// pixels instead of 100.  Why it matters: a 16-byte wave load costs the per-CU L1 >= 16 clocks (profiles/r03_calibration.json)
// and the one-pixel kernel issues 82 of them per pixel -- it is bound there and by VALU issue at the same time; shared
// points also share their address arithmetic and luminance.  Per (pixel, tap) the arithmetic is the one-pixel kernel's
// to the operation; only the ORDER in which a pixel's 24 weighted taps are summed differs (lattice order instead of
// svgf_impl.h:693-726's ring order), i.e. float rounding of a sum of <= 25 positive terms.  THIS kernel is the default
// (ATEN_AMD_SVGF_ATROUS4=0 selects the one-pixel, ring-order kernel): the default SVGF output therefore differs from a
// reference-order evaluation by that rounding -- measured bound against the one-pixel kernel on identical planes and histories:
// tests/test_gpu_svgf.py::test_atrous_four_pixel_kernel_against_the_one_pixel_kernel.
// Thread (tx, ty) -> x0 = (tx >> i) * 2s + (tx & (s - 1)): every residue class modulo s gets its own lattice.
struct AtrousPixel {
    f3 n;           // centre normal
    float depth, lum, inv_l, inv_z0, inv_z1, inv_z2, inv_z3, inv_z4;
    int32_t meshid;
    float4 sumC;    // .w unused
    float sumV, weight;
};

ATN_DEV void atrous_tap(AtrousPixel& c, const float4& q_nd, int32_t q_meshid, const float4& color, float lum, float inv_z, float hh)
{
    const float Wz = (3.0F * fabsf(c.depth - q_nd.w)) * inv_z;
    const float dn = dot(c.n, mk3(q_nd));
    const float Wn = pow128(0.0F < dn ? dn : 0.0F);
    const float el = svgf_exp(-fabsf(c.lum - lum) * c.inv_l);
    const float Wl = 1.0F < el ? 1.0F : el;          // std::min(e, 1.0f)
    const float Wm = q_meshid == c.meshid ? 1.0F : 0.0F;
    const float W = svgf_exp(-Wl * Wl - Wz) * Wn * Wm * hh;
    c.sumC.x += W * color.x; c.sumC.y += W * color.y; c.sumC.z += W * color.z;
    c.sumV += W * W * color.w;
    c.weight += W;
}

#ifndef ATN_ATROUS4_WAVES
#define ATN_ATROUS4_WAVES 0
#endif
#if ATN_ATROUS4_WAVES
__attribute__((amdgpu_waves_per_eu(ATN_ATROUS4_WAVES, ATN_ATROUS4_WAVES)))
#endif
__global__ void __launch_bounds__(256) k_svgf_atrous4(SvgfFrame sf, int32_t iter)
{
    const int32_t width = sf.width, height = sf.height;
    const int32_t s = 1 << iter;
    // XCD-aware block -> tile map as in svgf_pixel, over the (tx, ty) space of pixel GROUPS
    int32_t tx, ty;
    {
        const uint32_t gx = gridDim.x;
        const uint32_t b = blockIdx.x + blockIdx.y * gx;
        const uint32_t strip = gx >> 3;
        const uint32_t xcd = b & 7u, local = b >> 3;
        const uint32_t bx = xcd * strip + local % strip, by = local / strip;
        tx = (int32_t)(bx * 8u + (threadIdx.x & 7u));
        ty = (int32_t)(by * 32u + (threadIdx.x >> 3));
    }
    const int32_t x0 = ((tx >> iter) << (iter + 1)) + (tx & (s - 1));
    const int32_t y0 = ((ty >> iter) << (iter + 1)) + (ty & (s - 1));
    if (x0 >= width || y0 >= height) return;

    const int32_t curr = iter & 1, next = 1 - curr;
    const bool is_first = iter == 0, is_final = iter == sf.atrous_iter_cnt - 1;
    const float4* __restrict__ src = is_first ? sf.cv : sf.atrous[curr];
    float4* __restrict__ dst = sf.atrous[next];
    const float fs = (float)s;

    AtrousPixel px[4];
    float4 am_c[4];         // albedo + id of the four centres (final level: re-modulation)
    bool valid[4], done[4];
#pragma unroll
    for (int32_t p = 0; p < 4; p++) {
        const int32_t ix = x0 + (p & 1) * s, iy = y0 + (p >> 1) * s;
        valid[p] = ix < width && iy < height;
        done[p] = !valid[p];
        const int32_t cx = valid[p] ? ix : x0, cy = valid[p] ? iy : y0;
        const int32_t idx = cx + cy * width;
        const float4 nml_depth = sf.nd[idx];
        const float4 am = sf.am[idx];
        const float4 contrib = src[idx];
        am_c[p] = am;
        AtrousPixel& c = px[p];
        c.n = mk3(nml_depth); c.depth = nml_depth.w; c.meshid = (int32_t)am.w;
        c.sumC = make_float4(contrib.x, contrib.y, contrib.z, 1.0F);
        c.sumV = 1.0F;                  // centre colour's w is set to 1 by ExtractCenterPixel (as in k_svgf_atrous)
        c.weight = 1.0F;
        c.lum = luminance(contrib.x, contrib.y, contrib.z);
        if (valid[p] && am.w < 0) {
            dst[idx] = make_float4(contrib.x, contrib.y, contrib.z, 0.0F);
            if (is_final) {
                sf.out[idx] = make_float4(am.x * contrib.x, am.y * contrib.y, am.z * contrib.z, am.w * 1.0F);
                done[p] = true;         // (the one-pixel kernel returns here)
            }
        }
        // 3x3 Gauss filter of the variance (pitch ONE pixel at every level)
        float gauss = 0.0F;
        {
            const float k3[3] = { 1.0F / 16.0F, 1.0F / 8.0F, 1.0F / 4.0F };
#pragma unroll
            for (int32_t i = 0; i < 9; i++) {
                const int32_t ox = i % 3 - 1, oy = i / 3 - 1;
                const int32_t xx = clampi(cx + ox, 0, width - 1);
                const int32_t yy = clampi(cy + oy, 0, height - 1);
                const float kk = k3[(ox == 0 ? 1 : 0) + (oy == 0 ? 1 : 0)];
                gauss += kk * src[xx + yy * width].w;
            }
        }
        const float sigmaZ = 1.0F, sigmaL = 4.0F;
        const float sqrt_gauss = sqrtf(gauss);
        const float pixel_distance_ratio = (c.depth / sf.camera_distance) * (float)height;
        c.inv_l = 1.0F / (sigmaL * sqrt_gauss + 0.000001F);
        c.inv_z0 = 1.0F / (sigmaZ * (pixel_distance_ratio * (fs * 1.0F)) + 0.000001F);
        c.inv_z1 = 1.0F / (sigmaZ * (pixel_distance_ratio * (fs * 2.0F)) + 0.000001F);
        c.inv_z2 = 1.0F / (sigmaZ * (pixel_distance_ratio * (fs * 1.41421354F)) + 0.000001F);
        c.inv_z3 = 1.0F / (sigmaZ * (pixel_distance_ratio * (fs * 2.23606801F)) + 0.000001F);
        c.inv_z4 = 1.0F / (sigmaZ * (pixel_distance_ratio * (fs * 2.82842708F)) + 0.000001F);
    }

    // the 6 x 6 lattice: point (u, v) is tap (u - a, v - b) of pixel (a, b) when both offsets are within +-2 and not both 0.
    // Rows in a REAL loop: unrolled, the 96 inlined (pixel, tap) bodies interleave into 256 VGPRs + 256 AGPRs + scratch.  v is
    // wave-uniform, so everything that depends on |v - b| only (is the row a tap row of pixel row b, the ring weights
    // h(|du|, |dv|), which of the five depth reciprocals) is scalar work done once per row; a row's 18 loads are issued together.
#pragma unroll 1
    for (int32_t v = -2; v <= 3; v++) {
        const int32_t yy = clampi(y0 + v * s, 0, height - 1);
        float4 q_nd[6], col[6];
        int32_t mid[6];
        float lum[6];
#pragma unroll
        for (int32_t k = 0; k < 6; k++) {
            const int32_t xx = clampi(x0 + (k - 2) * s, 0, width - 1);
            const int32_t qidx = xx + yy * width;
            q_nd[k] = sf.nd[qidx];
            mid[k] = (int32_t)sf.am[qidx].w;
            col[k] = src[qidx];
        }
#pragma unroll
        for (int32_t k = 0; k < 6; k++) lum[k] = luminance(col[k].x, col[k].y, col[k].z);
#pragma unroll
        for (int32_t b = 0; b < 2; b++) {
            const int32_t dv = v - b;
            const int32_t av = dv < 0 ? -dv : dv;
            if (av > 2) continue;                                   // scalar: the row is outside pixel row b's 5 x 5
            // h of svgf_impl.h:693-726 by ring: (1,0) 2/3, (2,0) 1/6, (1,1) 4/9, (1,2) 1/9, (2,2) 1/36
            const float hh0 = av == 1 ? 2.0F / 3.0F : 1.0F / 6.0F;                               // |du| = 0 (av != 0)
            const float hh1 = av == 0 ? 2.0F / 3.0F : av == 1 ? 4.0F / 9.0F : 1.0F / 9.0F;       // |du| = 1
            const float hh2 = av == 0 ? 1.0F / 6.0F : av == 1 ? 1.0F / 9.0F : 1.0F / 36.0F;      // |du| = 2
#pragma unroll
            for (int32_t a = 0; a < 2; a++) {
                AtrousPixel& c = px[a + 2 * b];
                // tap length classes: 1 -> inv_z0, 2 -> inv_z1, sqrt 2 -> inv_z2, sqrt 5 -> inv_z3, sqrt 8 -> inv_z4
                const float iz0 = av == 1 ? c.inv_z0 : c.inv_z1;
                const float iz1 = av == 0 ? c.inv_z0 : av == 1 ? c.inv_z2 : c.inv_z3;
                const float iz2 = av == 0 ? c.inv_z1 : av == 1 ? c.inv_z3 : c.inv_z4;
#pragma unroll
                for (int32_t k = 0; k < 6; k++) {
                    const int32_t du = (k - 2) - a;
                    const int32_t au = du < 0 ? -du : du;
                    if (au > 2) continue;                           // compile time
                    if (au == 0) { if (av != 0) atrous_tap(c, q_nd[k], mid[k], col[k], lum[k], iz0, hh0); }    // (0, 0) is the centre
                    else atrous_tap(c, q_nd[k], mid[k], col[k], lum[k], au == 1 ? iz1 : iz2, au == 1 ? hh1 : hh2);
                }
            }
        }
    }

#pragma unroll
    for (int32_t p = 0; p < 4; p++) {
        if (done[p]) continue;
        const int32_t ix = x0 + (p & 1) * s, iy = y0 + (p >> 1) * s;
        const int32_t idx = ix + iy * width;
        AtrousPixel& c = px[p];
        const float4 filtered = make_float4(c.sumC.x / c.weight, c.sumC.y / c.weight, c.sumC.z / c.weight, c.sumV / (c.weight * c.weight));
        dst[idx] = filtered;
        if (is_first) {
            const float4 t = sf.tmp[idx];
            sf.tmp[idx] = make_float4(filtered.x, filtered.y, filtered.z, t.w);
        }
        if (is_final) {
            const float4 am = am_c[p];
            sf.out[idx] = make_float4(am.x * filtered.x, am.y * filtered.y, am.z * filtered.z, am.w * filtered.w);
        }
    }
}

} // namespace atn
