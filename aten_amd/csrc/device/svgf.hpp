// SVGF passes (next-tier row of SURVEY 8(f)): the per-pixel functions of
// src/libaten/renderer/svgf/svgf_impl.h as full-frame HIP kernels, driven in the order of
// aten::SVGFRenderer::OnRender (src/libaten/renderer/svgf/svgf.cpp:452-637).
//
// All buffers are full-frame float4[w*h], idx = x + y*w (row 0 = bottom), resident in HBM across
// frames.  Every pass is a gather over a small pixel neighbourhood: HBM/L2-bandwidth work, one thread
// per pixel, 8x8-pixel blocks of four waves so that a wave's taps fall into few cache lines.
#pragma once
#include "kernels.hpp"
#include "svgf_frame.hpp"

namespace atn {


__global__ void __launch_bounds__(256) k_svgf_fill(float4* p, uint32_t n, float4 v)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// Per-sample epilogue of SVGFRenderer::OnRender's sample loop (svgf.cpp:488-513): Path.contrib keeps the LAST
// sample's radiance (it is reset before every sample) and counts the generated samples; invalid colours only
// skip the termination test.
__global__ void __launch_bounds__(256) k_svgf_sample_end(PathBuffers pb, FrameParams fp, SvgfFrame sf)
{
    const uint32_t slot = (uint32_t)fp.slot_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= (uint32_t)fp.slot_end) return;
    int32_t x, y;
    if (!slot_to_pixel(fp, slot, x, y)) return;
    if (fp.sample > 0 && pb.done[slot]) return;
    const float4 c = pb.contrib[slot];
    sf.contribs[y * fp.width + x] = make_float4(c.x, c.y, c.z, (float)(fp.sample + 1));
    const bool invalid = isnan(c.x) || isinf(c.x) || isnan(c.y) || isinf(c.y) || isnan(c.z) || isinf(c.z)
        || c.x < 0 || c.y < 0 || c.z < 0;
    if (invalid) return;
    const uint32_t flags = __float_as_uint(pb.ray_d[slot].w);
    if (flags & F_TERMINATED) pb.done[slot] = 1;
}

// svgf::PrepareForDenoise (svgf_impl.h:119-144) + the motion pass that stands in for the reference's GL raster
// pass (src/shader/ssrt_fs.glsl:31-47; static geometry): motion = prevNDC01 - curNDC01, z = clip w.
__global__ void __launch_bounds__(256) k_svgf_prepare(SvgfFrame sf)
{
    int32_t ix, iy;
    if (!svgf_pixel(sf, ix, iy)) return;
    const int32_t idx = ix + iy * sf.width;
    if (sf.g_nd) {
        // The path pass writes its AOVs (FillBasicAOVs, every pixel every frame) into a staging pair instead of the
        // current AOV set, so that it never touches planes a filter pass of the PREVIOUS frame may still read: with
        // frames in flight it runs while that frame is being filtered.  Same values, one extra 32-byte move per pixel.
        sf.nd[idx] = sf.g_nd[idx];
        sf.am[idx] = sf.g_am[idx];
    }
    const float4 c = sf.contribs[idx];
    const float4 contrib = div4(c, c.w);
    if (sf.frame == 0) {
        const float lum = luminance(contrib.x, contrib.y, contrib.z);
        float4 mt = sf.mt[idx];
        mt.x += lum * lum; mt.y += lum; mt.z += 1;
        sf.mt[idx] = mt;
        sf.cv[idx] = make_float4(contrib.x, contrib.y, contrib.z, sf.cv[idx].w);
    }
    sf.tmp[idx] = c;
    if (sf.stages) sf.stages[idx] = make_float4(c.x, c.y, c.z, 1.0F);
    sf.out[idx] = make_float4(c.x, c.y, c.z, 1.0F);
    if (sf.compute_motion) {
        const float4 wp = sf.primary[idx];
        float4 md = make_float4(0.0F, 0.0F, -1.0F, 1.0F);
        if (wp.w != 0.0F) {
            const float* a = sf.w2c; const float* b = sf.prev_w2c;
            const float cx = a[0] * wp.x + a[1] * wp.y + a[2] * wp.z + a[3] * 1.0F;
            const float cy = a[4] * wp.x + a[5] * wp.y + a[6] * wp.z + a[7] * 1.0F;
            const float cw = a[12] * wp.x + a[13] * wp.y + a[14] * wp.z + a[15] * 1.0F;
            const float px = b[0] * wp.x + b[1] * wp.y + b[2] * wp.z + b[3] * 1.0F;
            const float py = b[4] * wp.x + b[5] * wp.y + b[6] * wp.z + b[7] * 1.0F;
            const float pw = b[12] * wp.x + b[13] * wp.y + b[14] * wp.z + b[15] * 1.0F;
            const float csx = (cx / cw) * 0.5F + 0.5F, csy = (cy / cw) * 0.5F + 0.5F;
            const float psx = (px / pw) * 0.5F + 0.5F, psy = (py / pw) * 0.5F + 0.5F;
            md = make_float4(psx - csx, psy - csy, cw, 1.0F);
        }
        sf.motion[idx] = md;
    }
}

// RecomputeTemporalWeightFromSurroundingPixels (svgf_impl.h:386-423): the temporal weight of a non-background pixel becomes
// the minimum over its 3x3 neighbourhood.  Only the CUDA twin runs it, right after temporal reprojection and IN PLACE
// (src/libidaten/svgf/svgf_tp.cu:150-216), so neighbours race; here every tap reads the weight the pass started with
// (two launches through a scalar plane).  Optional: atn_svgf_set_dilate_temporal_weight.
__global__ void __launch_bounds__(256) k_svgf_dilate_weight(SvgfFrame sf, float* __restrict__ out)
{
    int32_t ix, iy;
    if (!svgf_pixel(sf, ix, iy)) return;
    const int32_t width = sf.width, height = sf.height;
    const int32_t idx = ix + iy * width;
    float w = sf.mt[idx].w;
    if ((int32_t)sf.am[idx].w >= 0) {
        for (int32_t y = -1; y <= 1; y++)
            for (int32_t x = -1; x <= 1; x++) {
                const int32_t xx = ix + x, yy = iy + y;
                if ((0 <= xx) && (xx < width) && (0 <= yy) && (yy < height)) {
                    const float nw = sf.mt[xx + yy * width].w;
                    w = (nw < w) ? nw : w;      // aten::min
                }
            }
    }
    out[idx] = w;
}
__global__ void __launch_bounds__(256) k_svgf_store_weight(SvgfFrame sf, const float* __restrict__ in)
{
    int32_t ix, iy;
    if (!svgf_pixel(sf, ix, iy)) return;
    const int32_t idx = ix + iy * sf.width;
    sf.mt[idx].w = in[idx];
}

// SVGFRenderer::TemporalReprojection (svgf.cpp:231-296) = ExtractCenterPixel + UpdateAOVIfBackgroundPixel +
// svgf::TemporalReprojection + AccumulateMoments (svgf_impl.h:154-380).  No transcendental: bit-exact.
__global__ void __launch_bounds__(256) k_svgf_temporal(SvgfFrame sf, float threshold_normal, float threshold_depth)
{
    int32_t ix, iy;
    if (!svgf_pixel(sf, ix, iy)) return;
    const int32_t width = sf.width, height = sf.height;
    const int32_t idx = ix + iy * width;
    const float4 nd = sf.nd[idx];
    const float center_depth = nd.w;
    const int32_t center_meshid = (int32_t)sf.am[idx].w;
    const float4 cc = sf.contribs[idx];
    float4 curr_color = div4(make_float4(cc.x, cc.y, cc.z, 1.0F), cc.w);
    const f3 center_normal = mk3(nd);

    if (center_meshid < 0) {
        sf.cv[idx] = curr_color;
        sf.mt[idx] = make_float4(1.0F, 1.0F, 1.0F, sf.mt[idx].w);
        if (sf.stages) sf.stages[(size_t)width * height + idx] = curr_color;
        sf.out[idx] = curr_color;
        return;
    }

    float4 sum = make_float4(0, 0, 0, 0);
    float weight = 0.0F;
    const float4 motion_depth = sf.motion[idx];     // the centre pixel's motion vector for all nine taps (:314-320)
    for (int32_t y = -1; y <= 1; y++) {
        for (int32_t x = -1; x <= 1; x++) {
            const int32_t xx = clampi(ix + x, 0, width - 1);
            const int32_t yy = clampi(iy + y, 0, height - 1);
            int32_t prev_x = (int32_t)((float)xx + motion_depth.x * (float)width);
            int32_t prev_y = (int32_t)((float)yy + motion_depth.y * (float)height);
            prev_x = clampi(prev_x, 0, width - 1);
            prev_y = clampi(prev_y, 0, height - 1);
            const int32_t prev_idx = prev_x + prev_y * width;
            const float4 pnd = sf.pnd[prev_idx];
            const float prev_depth = pnd.w;
            const int32_t prev_meshid = (int32_t)sf.pam[prev_idx].w;
            const f3 prev_normal = mk3(pnd);
            const float Wz = clampf((threshold_depth - fabsf(1 - center_depth / prev_depth)) / threshold_depth, 0.0F, 1.0F);
            const float Wn = clampf((dot(center_normal, prev_normal) - threshold_normal) / (1.0F - threshold_normal), 0.0F, 1.0F);
            const float Wm = center_meshid == prev_meshid ? 1.0F : 0.0F;
            const float4 prev_color = sf.pcv[prev_idx];
            const float W = Wz * Wn * Wm;
            sum = add4(sum, mul4(W, prev_color));
            weight += W;
        }
    }
    if (weight > 0.0F) {
        sum = div4(sum, weight);
        weight /= 9;
        curr_color = add4(mul4(0.2F, curr_color), mul4(0.8F, sum));
    }
    const float4 cv_old = sf.cv[idx];
    sf.cv[idx] = make_float4(curr_color.x, curr_color.y, curr_color.z, cv_old.w);

    // AccumulateMoments
    const float lum = luminance(curr_color.x, curr_color.y, curr_color.z);
    f3 center_moment = mk3(lum * lum, lum, 0.0F);
    int32_t frame = 1;
    if (weight > 0.0F) {
        const float4 pm = sf.pmt[idx];
        frame = (int32_t)(pm.z + 1);
        center_moment = center_moment + mk3(pm);
    }
    sf.mt[idx] = make_float4(center_moment.x, center_moment.y, (float)frame, weight);
    if (sf.stages) sf.stages[(size_t)width * height + idx] = curr_color;
    sf.out[idx] = curr_color;
}

// svgf::EstimateVariance (svgf_impl.h:441-545)
__global__ void __launch_bounds__(256) k_svgf_variance(SvgfFrame sf)
{
    int32_t ix, iy;
    if (!svgf_pixel(sf, ix, iy)) return;
    const int32_t width = sf.width, height = sf.height;
    const int32_t idx = ix + iy * width;
    const float4 normal_depth = sf.nd[idx];
    const float4 mtw = sf.mt[idx];
    const float4 center_color = sf.cv[idx];
    const float center_depth = normal_depth.w;
    const int32_t center_meshid = (int32_t)sf.am[idx].w;
    float4* st = sf.stages ? sf.stages + (size_t)2 * width * height : nullptr;
    if (center_meshid < 0) {
        sf.mt[idx] = make_float4(0.0F, 0.0F, 1.0F, mtw.w);
        sf.cv_out[idx] = center_color;
        if (st) st[idx] = make_float4(0, 0, 0, 0);
        sf.out[idx] = make_float4(0, 0, 0, 0);
        return;
    }
    const float pixel_distance_ratio = (center_depth / sf.camera_distance) * (float)height;
    f3 center_moment = mk3(mtw);
    const int32_t frame = (int32_t)center_moment.z;
    center_moment = center_moment / center_moment.z;
    float variance = 0.0F;
    float4 color = center_color;
    if (frame < 4) {
        const f3 center_normal = mk3(normal_depth);
        f3 moment_sum = center_moment;
        float weight = 1.0F;
        const int32_t radius = frame > 1 ? 2 : 3;
        for (int32_t v = -radius; v <= radius; v++) {
            for (int32_t u = -radius; u <= radius; u++) {
                if (u != 0 || v != 0) {
                    const int32_t xx = clampi(ix + u, 0, width - 1);
                    const int32_t yy = clampi(iy + v, 0, height - 1);
                    const int32_t sidx = xx + yy * width;
                    const float4 s_nd = sf.nd[sidx];
                    const float4 s_mt = sf.mt[sidx];
                    const int32_t sample_meshid = (int32_t)sf.am[sidx].w;
                    const float4 sample_color = sf.cv[sidx];
                    f3 moment = mk3(s_mt);
                    moment = moment / moment.z;
                    const float uv_length = sqrtf((float)(u * u + v * v));
                    const float Wz = fabsf(s_nd.w - center_depth) / (pixel_distance_ratio * uv_length + 1e-2F);
                    const float dn = dot(mk3(s_nd), center_normal);
                    const float Wn = pow128(0.0F < dn ? dn : 0.0F);       // std::max(0.0f, d)
                    const float Wm = center_meshid == sample_meshid ? 1.0F : 0.0F;
                    const float W = svgf_exp(-Wz) * Wn * Wm;
                    moment_sum = moment_sum + moment * W;
                    color = add4(color, mul4(W, sample_color));
                    weight += W;
                }
            }
        }
        moment_sum = moment_sum / weight;
        color = div4(color, weight);
        const float var = moment_sum.x - moment_sum.y * moment_sum.y;
        variance = 0.0F < var ? var : 0.0F;
    }
    else {
        const float var = center_moment.x - center_moment.y * center_moment.y;
        variance = 0.0F < var ? var : 0.0F;
    }
    color.w = variance;
    sf.cv_out[idx] = color;
    const float4 o = make_float4(variance, variance, variance, 1.0F);
    if (st) st[idx] = o;
    sf.out[idx] = o;
}

// SVGFRenderer::AtrousFilter: k_svgf_atrous (one pixel per thread, ring order) and k_svgf_atrous4 (four pixels per thread, the
// default) are defined in device/svgf_atrous.hpp and compiled in a translation unit of their own, csrc/svgf_atrous.hip (the one
// place where the SLP vectoriser stays on: see that header).
__global__ void k_svgf_atrous(SvgfFrame sf, int32_t iter);
__global__ void k_svgf_atrous4(SvgfFrame sf, int32_t iter);

// CopyFromTeporaryColorBufferToAov: CopyVectorBuffer<3> (svgf.cpp:402-410)
__global__ void __launch_bounds__(256) k_svgf_copy(SvgfFrame sf)
{
    int32_t ix, iy;
    if (!svgf_pixel(sf, ix, iy)) return;
    const int32_t idx = ix + iy * sf.width;
    const float4 t = sf.tmp[idx];
    const float4 c = sf.cv[idx];
    sf.cv[idx] = make_float4(t.x, t.y, t.z, c.w);
}

} // namespace atn
