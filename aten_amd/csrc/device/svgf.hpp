// SVGF passes (next-tier row of SURVEY 8(f)): the per-pixel functions of
// src/libaten/renderer/svgf/svgf_impl.h as full-frame HIP kernels, driven in the order of
// aten::SVGFRenderer::OnRender (src/libaten/renderer/svgf/svgf.cpp:452-637).
//
// All buffers are full-frame float4[w*h], idx = x + y*w (row 0 = bottom), resident in HBM across
// frames.  Every pass is a gather over a small pixel neighbourhood: HBM/L2-bandwidth work, one thread
// per pixel, 8x8-pixel blocks of four waves so that a wave's taps fall into few cache lines.
#pragma once
#include "kernels.hpp"

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
