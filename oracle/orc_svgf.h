/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h).
 *
 * orc_svgf.h: CPU restatement of aten's SVGF renderer (next-tier row of SURVEY 8(f)):
 *   aten::SVGFRenderer::{OnRender, ExecRendering, Shade, TemporalReprojection, EstimateVariance, AtrousFilter}
 *     src/libaten/renderer/svgf/svgf.cpp:19-639
 *   AT_NAME::svgf::*           src/libaten/renderer/svgf/svgf_impl.h:62-873
 *   SVGFParams / buffers       src/libaten/renderer/svgf/svgf_types.h:54-166
 *   FillBasicAOVs*             src/libaten/renderer/aov.h:158-198
 *   MatricesForRendering       src/libaten/renderer/pathtracing/pt_params.h:150-185, math/mat4.h:140-156,237-285,457-513
 * PARITY STATUS: unpinned (the reference holds no fixture for this path and cannot be built here).
 *
 * One place where the reference has no single answer: svgf::EstimateVariance overwrites
 * aov_color_variance[idx] while other pixels of the same OpenMP pass read it as a filter tap
 * (svgf_impl.h:512,537-538), so its result depends on thread timing.  This restatement reads every tap
 * from the values the pass started with (what every schedule gives when no neighbour has been
 * processed yet, and what a data-parallel device pass gives).
 */
#pragma once
#include "orc_pt.h"

namespace orc {
namespace svgf {

inline v4 operator/(const v4& v, float t) { return v4(v.x / t, v.y / t, v.z / t, v.w / t); }     // vec4.h operator/
inline v4& operator+=(v4& a, const v4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; return a; }
inline v4& operator/=(v4& a, float t) { a.x /= t; a.y /= t; a.z /= t; a.w /= t; return a; }
template <class T> inline T clampv(T f, T a, T b) { return (f < a) ? a : (b < f) ? b : f; }       // std::clamp

enum { NormalDepth = 0, AlbedoMeshId = 1, ColorVariance = 2, MomentTemporalWeight = 3, NumAov = 4 };

// mat4::operator*= (mat4.h:140-156)
inline m4 mul(const m4& a, const m4& b)
{
    m4 tmp;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        tmp.m[i][j] = 0.0f;
        for (int k = 0; k < 4; ++k) tmp.m[i][j] += a.m[i][k] * b.m[k][j];
    }
    return tmp;
}

// mat4::invert, Gauss-Jordan with row pivoting (mat4.h:237-285)
inline m4 invert(const m4& src)
{
    m4 mtx = src, dst = m4::identity();
    for (int i = 0; i < 4; ++i) {
        float f = std::fabs(mtx.m[i][i]);
        for (int j = i + 1; j < 4; ++j) {
            if (f < std::fabs(mtx.m[j][i])) {
                f = std::fabs(mtx.m[j][i]);
                for (int c = 0; c < 4; c++) { std::swap(mtx.m[i][c], mtx.m[j][c]); std::swap(dst.m[i][c], dst.m[j][c]); }
            }
        }
        f = 1.0f / mtx.m[i][i];
        for (int c = 0; c < 4; c++) { mtx.m[i][c] = mtx.m[i][c] * f; dst.m[i][c] = dst.m[i][c] * f; }
        for (int j = 0; j < 4; ++j) {
            if (j != i) {
                float temp = mtx.m[j][i];
                for (int c = 0; c < 4; c++) {
                    const float v1 = mtx.m[i][c] * temp, v2 = dst.m[i][c] * temp;
                    mtx.m[j][c] = mtx.m[j][c] - v1;
                    dst.m[j][c] = dst.m[j][c] - v2;
                }
            }
        }
    }
    return dst;
}

struct Matrices {       // MatricesForRendering
    m4 W2V = m4::identity(), V2C = m4::identity(), C2V = m4::identity(), V2W = m4::identity(), PrevW2V = m4::identity();
    m4 GetW2C() const { return mul(V2C, W2V); }
    void Reset(const atn_camera_param& cam)
    {
        PrevW2V = W2V;
        // mat4::lookat(origin, center, up) (mat4.h:457-477) on the existing matrix
        const v3 eye = ld3(cam.origin), at = ld3(cam.center), up = ld3(cam.up);
        const v3 zaxis = normalize(eye - at);
        const v3 xaxis = normalize(cross(up, zaxis));
        const v3 yaxis = cross(zaxis, xaxis);
        W2V.m[0][0] = xaxis.x; W2V.m[1][0] = yaxis.x; W2V.m[2][0] = zaxis.x;
        W2V.m[0][1] = xaxis.y; W2V.m[1][1] = yaxis.y; W2V.m[2][1] = zaxis.y;
        W2V.m[0][2] = xaxis.z; W2V.m[1][2] = yaxis.z; W2V.m[2][2] = zaxis.z;
        W2V.m[0][3] = -dot(xaxis, eye); W2V.m[1][3] = -dot(yaxis, eye); W2V.m[2][3] = -dot(zaxis, eye);
        W2V.m[3][3] = 1;
        // mat4::perspective(znear, zfar, vfov, aspect) (mat4.h:479-513)
        const float fH = 1 / std::tan(Deg2Rad(cam.vfov) * 0.5f);
        const float fW = fH / cam.aspect;
        V2C.m[0][0] = fW; V2C.m[1][1] = fH;
        V2C.m[2][2] = cam.zfar / (cam.znear - cam.zfar);
        V2C.m[2][3] = cam.znear * cam.zfar / (cam.znear - cam.zfar);
        V2C.m[3][2] = -1.0f; V2C.m[3][3] = 0.0f;
        C2V = invert(V2C);
        V2W = invert(W2V);
    }
};

struct Params {         // SVGFParams<std::vector<vec4>> + the renderer's frame-persistent state
    int32_t width = 0, height = 0;
    int32_t curr_aov_pos = 0;
    int32_t atrous_iter_cnt = 5;
    int32_t dilate_temporal_weight = 0;     // optional pass of the CUDA twin (svgf_tp.cu:150-216), off on the CPU path
    std::vector<v4> aovs[2][NumAov];
    std::vector<v4> atrous_clr_variance[2];
    std::vector<v4> temporary_color_buffer;
    std::vector<v4> motion_depth_buffer;
    std::vector<v4> primary_position;       // not in the reference: world position of the bounce-0 hit (w = 1) or
                                            // w = 0 on a miss; input of the motion compute pass that replaces the GL raster pass
    Matrices mtxs;
    std::vector<v4>& cur(int t) { return aovs[curr_aov_pos][t]; }
    std::vector<v4>& prev(int t) { return aovs[1 - curr_aov_pos][t]; }
    void InitBuffers(int32_t w, int32_t h)
    {
        width = w; height = h;
        const size_t n = (size_t)w * h;
        for (auto& a : aovs) for (auto& b : a) if (b.empty()) b.resize(n);      // vec4() = (0,0,0,1)
        for (auto& b : atrous_clr_variance) b.resize(n);
        temporary_color_buffer.resize(n);
        primary_position.resize(n);
    }
};

// FillBasicAOVs (aov.h:158-181); known answers: aten_unittest/aov_host_buffer.cpp:73-108 (tests/test_oracle_svgf_cpu.py)
inline void FillBasicAOVs(v4& aovNormalDepth, const v3& normal, const v3& rec_p, const m4& mtx_W2C, v4& aovAlbedoMeshId,
    const v4& albedo, int32_t isect_meshid)
{
    // World coordinate to Clip coordinate.
    v4 pos(rec_p, 1);
    pos = mtx_W2C.apply(pos);

    aovNormalDepth.x = normal.x;
    aovNormalDepth.y = normal.y;
    aovNormalDepth.z = normal.z;
    aovNormalDepth.w = pos.w;

    aovAlbedoMeshId.x = albedo.x;
    aovAlbedoMeshId.y = albedo.y;
    aovAlbedoMeshId.z = albedo.z;
    aovAlbedoMeshId.w = static_cast<float>(isect_meshid);
}

// FillBasicAOVsIfHitMiss (aov.h:183-198); known answers: aov_host_buffer.cpp:110-132
inline void FillBasicAOVsIfHitMiss(v4& aovNormalDepth, v4& aovAlbedoMeshId, const v4& bg)
{
    aovNormalDepth.x = 0.0F;
    aovNormalDepth.y = 0.0F;
    aovNormalDepth.z = 0.0F;
    aovNormalDepth.w = -1;

    aovAlbedoMeshId.x = bg.x;
    aovAlbedoMeshId.y = bg.y;
    aovAlbedoMeshId.z = bg.z;
    aovAlbedoMeshId.w = -1;
}

// SVGFRenderer::Shade's use of it: FillBasicAOVs, then "aov_albedo_meshid[idx].w = isect.mtrlid" (svgf.cpp:128-131,144-147)
inline void FillAOVs(v4& nd, v4& am, const v3& normal, const HitRec& rec, const m4& W2C, const v4& texcolor, const Isect& isect)
{
    FillBasicAOVs(nd, normal, rec.p, W2C, am, texcolor, isect.meshid);
    am.w = static_cast<float>(isect.mtrlid);
}

// SVGFRenderer::Shade, svgf.cpp:89-229
inline void Shade(PathState& path, const Scene& ctxt, Ray& ray, ShadowRay& shadow_ray, const Isect& isect,
    int32_t rrDepth, int32_t bounce, const m4& W2C, v4& aov_nd, v4& aov_am, v4& primary_pos, PathCounters* cnt)
{
    if (cnt) cnt->hits++;
    const Ray ray_in = ray;
    const auto& obj = ctxt.GetObject(static_cast<uint32_t>(isect.objid));
    HitRec rec;
    evaluate_hit_result(rec, obj, ctxt, ray_in, isect);
    bool isBackfacing = dot(rec.normal, -ray_in.dir) < 0.0F;
    v3 orienting_normal = rec.normal;

    atn_material_param mtrl;
    FillMaterial(mtrl, ctxt, rec.mtrlid);

    if (bounce == 0) {
        v4 texcolor = sampleTexture(ctxt, mtrl.albedoMap, rec.u, rec.v, v4(1.0f));
        FillAOVs(aov_nd, aov_am, orienting_normal, rec, W2C, texcolor, isect);
        mtrl.albedoMap = -1;        // "for exporting separated albedo"
        primary_pos = v4(rec.p, 1.0f);
    }
    else if (bounce == 1 && path.last_hit_mtrl_idx >= 0) {
        const auto& last = ctxt.GetMaterial(path.last_hit_mtrl_idx);
        if (last.type == ATN_MTRL_SPECULAR) {
            v4 texcolor = sampleTexture(ctxt, mtrl.albedoMap, rec.u, rec.v, v4(1.0f));
            FillAOVs(aov_nd, aov_am, orienting_normal, rec, W2C, texcolor, isect);
            mtrl.albedoMap = -1;
        }
    }

    v4 albedo = sampleTexture(ctxt, mtrl.albedoMap, rec.u, rec.v, v4(1.0F));
    shadow_ray.isActive = false;

    // CheckMaterialTranslucentByAlpha: scene_rendering_config.enable_alpha_blending is off on this path
    // (pathtracing_impl.h:524-526) -> false.

    if (HitImplicitLight(ctxt, isect.objid, isBackfacing, bounce, path, ray_in, rec, mtrl)) return;

    if (!attr_translucent(mtrl) && isBackfacing) orienting_normal = -orienting_normal;
    float pre_sampled_r;
    {
        v3 nn;
        pre_sampled_r = applyNormal(ctxt, mtrl, orienting_normal, nn, rec.u, rec.v, ray_in.dir, &path.sampler);
        orienting_normal = nn;
    }
    FillShadowRay(shadow_ray, ctxt, path, mtrl, ray_in, rec.p, orienting_normal, rec.u, rec.v, albedo, pre_sampled_r);
    const float russianProb = ComputeRussianProbability(bounce, rrDepth, path);
    MaterialSampling sampling;
    sampleMaterial(&sampling, ctxt, &mtrl, orienting_normal, ray_in.dir, &path.sampler, rec.u, rec.v, pre_sampled_r);
    PrepareForNextBounce(rec, russianProb, orienting_normal, mtrl, sampling, albedo.xyz(), path, ray);
}

// ShadeMiss with AOV spans, pathtracing_impl.h:112-176 + FillBasicAOVsIfHitMiss (aov.h:183-198)
inline void ShadeMissAov(int32_t ix, int32_t iy, int32_t width, int32_t height, int32_t bounce,
    const Scene& ctxt, const atn_camera_param& camera, PathState& path, const Ray& ray, v4& aov_nd, v4& aov_am, v4& primary_pos)
{
    if (!path.is_terminated && !path.isHit) {
        v3 dir = ray.dir;
        if (bounce == 0) {
            float s = ix / (float)(width);
            float t = iy / (float)(height);
            dir = PinholeSample(camera, s, t).dir;
            primary_pos = v4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        v4 emit = Background_SampleFromRay(dir, ctxt.cfg().bg, ctxt);
        float misW = 1.0f;
        if (bounce == 0 || (bounce == 1 && path.is_singular)) {
            FillBasicAOVsIfHitMiss(aov_nd, aov_am, emit);
        }
        else {
            float pdfLight = IBL_samplePdf(emit.xyz(), ctxt.cfg().bg.avgIllum);
            misW = path.pdfb / (pdfLight + path.pdfb);
        }
        v3 contrib = 1.0F * (misW * emit).xyz() + v3(0.0F);
        contrib *= path.throughput;
        path.contrib += contrib;
        path.is_terminated = true;
    }
}

// SVGFRenderer::ExecRendering, svgf.cpp:19-87
inline void ExecRendering(PathState& path, Ray& ray, ShadowRay& shadow_ray, int32_t ix, int32_t iy,
    int32_t width, int32_t height, const Scene& ctxt, const atn_camera_param& camera, int32_t maxDepth, int32_t rrDepth,
    const m4& W2C, v4& aov_nd, v4& aov_am, v4& primary_pos, PathCounters* cnt)
{
    int32_t depth = 0;
    while (depth < maxDepth) {
        bool willContinue = true;
        path.isHit = false;
        Isect isect;
        if (cnt) cnt->closest_rays++;
        bool is_hit = TraverseClosest(isect, ctxt, ray, EPS, INF, cnt ? &cnt->trav : nullptr);
        if (is_hit) {
            path.isHit = true;
            Shade(path, ctxt, ray, shadow_ray, isect, rrDepth, depth, W2C, aov_nd, aov_am, primary_pos, cnt);
            HitShadowRay(ctxt, path, shadow_ray, isect.mtrlid >= 0 ? ctxt.GetMaterial(isect.mtrlid).stencil_type : 0, cnt);
            willContinue = !path.is_terminated;
        }
        else {
            ShadeMissAov(ix, iy, width, height, depth, ctxt, camera, path, ray, aov_nd, aov_am, primary_pos);
            willContinue = false;
        }
        if (!willContinue) break;
        depth++;
    }
}

// svgf::TemporalReprojection + wrapper SVGFRenderer::TemporalReprojection (svgf_impl.h:284-380, svgf.cpp:231-296)
inline v4 TemporalReprojection(int32_t ix, int32_t iy, int32_t width, int32_t height, float threshold_normal,
    float threshold_depth, const v4& contrib_c, Params& p)
{
    const int32_t idx = ix + iy * width;
    auto& cur_nd = p.cur(NormalDepth); auto& cur_am = p.cur(AlbedoMeshId);
    auto& cur_cv = p.cur(ColorVariance); auto& cur_mt = p.cur(MomentTemporalWeight);
    auto& prev_nd = p.prev(NormalDepth); auto& prev_am = p.prev(AlbedoMeshId);
    auto& prev_cv = p.prev(ColorVariance); auto& prev_mt = p.prev(MomentTemporalWeight);

    // ExtractCenterPixel<true> (svgf_impl.h:154-181)
    const float center_depth = cur_nd[idx].w;
    const int32_t center_meshid = static_cast<int32_t>(cur_am[idx].w);
    v4 curr_color(contrib_c.x, contrib_c.y, contrib_c.z, 1.0f);
    curr_color /= contrib_c.w;
    const v3 center_normal(cur_nd[idx].x, cur_nd[idx].y, cur_nd[idx].z);

    // UpdateAOVIfBackgroundPixel (:194-213)
    if (center_meshid < 0) {
        cur_cv[idx] = curr_color;
        cur_mt[idx] = v4(1.0F, 1.0F, 1.0F, cur_mt[idx].w);
        return curr_color;
    }

    v4 sum(0, 0, 0, 0);
    float weight = 0.0f;
    for (int32_t y = -1; y <= 1; y++) {
        for (int32_t x = -1; x <= 1; x++) {
            int32_t xx = clampv(ix + x, 0, width - 1);
            int32_t yy = clampv(iy + y, 0, height - 1);
            const v4 motion_depth = p.motion_depth_buffer[idx];       // centre pixel's motion for all 9 taps (quirk)
            int32_t prev_x = static_cast<int32_t>(xx + motion_depth.x * width);
            int32_t prev_y = static_cast<int32_t>(yy + motion_depth.y * height);
            prev_x = clampv(prev_x, 0, width - 1);
            prev_y = clampv(prev_y, 0, height - 1);
            int32_t prev_idx = prev_x + prev_y * width;
            const v4& nml_depth = prev_nd[prev_idx];
            const v4& texclr_meshid = prev_am[prev_idx];
            const float prev_depth = nml_depth.w;
            const int32_t prev_meshid = (int32_t)texclr_meshid.w;
            const v3 prev_normal(nml_depth.x, nml_depth.y, nml_depth.z);
            float Wz = clampv((threshold_depth - std::abs(1 - center_depth / prev_depth)) / threshold_depth, 0.0f, 1.0f);
            float Wn = clampv((dot(center_normal, prev_normal) - threshold_normal) / (1.0f - threshold_normal), 0.0f, 1.0f);
            float Wm = center_meshid == prev_meshid ? 1.0f : 0.0f;
            const v4& prev_color = prev_cv[prev_idx];
            float W = Wz * Wn * Wm;
            sum += prev_color * W;
            weight += W;
        }
    }
    if (weight > 0.0f) {
        sum /= weight;
        weight /= 9;
        curr_color = 0.2f * curr_color + 0.8f * sum;
    }
    cur_mt[idx].w = weight;
    cur_cv[idx].x = curr_color.x; cur_cv[idx].y = curr_color.y; cur_cv[idx].z = curr_color.z;

    // AccumulateMoments (:224-264)
    {
        const v4& color_variance = cur_cv[idx];
        float lum = luminance(color_variance.x, color_variance.y, color_variance.z);
        v3 center_moment(lum * lum, lum, 0);
        int32_t frame = 1;
        if (weight > 0.0f) {
            const v4 mt = prev_mt[idx];
            v3 prev_moment(mt.x, mt.y, mt.z);
            frame = static_cast<int32_t>(prev_moment.z + 1);
            center_moment += prev_moment;
        }
        center_moment.z = static_cast<float>(frame);
        cur_mt[idx].x = center_moment.x; cur_mt[idx].y = center_moment.y; cur_mt[idx].z = center_moment.z;
    }
    return curr_color;
}

// RecomputeTemporalWeightFromSurroundingPixels, svgf_impl.h:386-423; `mt_in` = the moment / temporal-weight plane as it was
// when the pass began.  Returns false for a background pixel (std::nullopt).
inline bool RecomputeTemporalWeightFromSurroundingPixels(int32_t ix, int32_t iy, int32_t width, int32_t height,
    const std::vector<v4>& aov_texclr_meshid, const std::vector<v4>& mt_in, float& out)
{
    const int32_t idx = iy * width + ix;
    const int32_t center_meshId = static_cast<int32_t>(aov_texclr_meshid[idx].w);
    if (center_meshId < 0) return false;
    float temporal_weight = mt_in[idx].w;
    for (int32_t y = -1; y <= 1; y++) {
        for (int32_t x = -1; x <= 1; x++) {
            int32_t xx = ix + x, yy = iy + y;
            if ((0 <= xx) && (xx < width) && (0 <= yy) && (yy < height)) {
                float w = mt_in[yy * width + xx].w;
                temporal_weight = fmin_(temporal_weight, w);
            }
        }
    }
    out = temporal_weight;
    return true;
}

// svgf::EstimateVariance (svgf_impl.h:441-545); `cv_in` = aov_color_variance as it was when the pass began
inline v4 EstimateVariance(int32_t ix, int32_t iy, int32_t width, int32_t height, float camera_distance,
    const std::vector<v4>& cv_in, Params& p)
{
    const int32_t idx = ix + iy * width;
    auto& aov_nd = p.cur(NormalDepth); auto& aov_am = p.cur(AlbedoMeshId);
    auto& aov_cv = p.cur(ColorVariance); auto& aov_mt = p.cur(MomentTemporalWeight);
    const v4 normal_depth = aov_nd[idx];
    const v4 texclr_meshid = aov_am[idx];
    const v4 moment_temporalweight = aov_mt[idx];
    const v4 center_color = cv_in[idx];
    const float center_depth = aov_nd[idx].w;
    const int32_t center_meshid = static_cast<int32_t>(texclr_meshid.w);
    if (center_meshid < 0) {
        aov_mt[idx].x = 0; aov_mt[idx].y = 0; aov_mt[idx].z = 1;
        return v4(0, 0, 0, 0);
    }
    const float pixel_distance_ratio = (center_depth / camera_distance) * height;
    v3 center_moment(moment_temporalweight.x, moment_temporalweight.y, moment_temporalweight.z);
    int32_t frame = static_cast<int32_t>(center_moment.z);
    center_moment /= center_moment.z;
    float variance = 0.0f;
    v4 color = center_color;
    if (frame < 4) {
        const v3 center_normal(normal_depth.x, normal_depth.y, normal_depth.z);
        v3 moment_sum(center_moment.x, center_moment.y, center_moment.z);
        float weight = 1.0f;
        int32_t radius = frame > 1 ? 2 : 3;
        for (int32_t v = -radius; v <= radius; v++) {
            for (int32_t u = -radius; u <= radius; u++) {
                if (u != 0 || v != 0) {
                    int32_t xx = clampv(ix + u, 0, width - 1);
                    int32_t yy = clampv(iy + v, 0, height - 1);
                    int32_t sample_idx = xx + yy * width;
                    const v4& s_nd = aov_nd[sample_idx];
                    const v4& s_am = aov_am[sample_idx];
                    const v4 s_mt = aov_mt[sample_idx];
                    const v3 sample_nml(s_nd.x, s_nd.y, s_nd.z);
                    const float sample_depth = s_nd.w;
                    const int32_t sample_meshid = static_cast<int32_t>(s_am.w);
                    const v4& sample_color = cv_in[sample_idx];
                    v3 moment(s_mt.x, s_mt.y, s_mt.z);
                    moment /= moment.z;
                    const float uv_length = std::sqrt(static_cast<float>(u * u + v * v));
                    const float Wz = std::abs(sample_depth - center_depth) / (pixel_distance_ratio * uv_length + 1e-2f);
                    const float Wn = std::pow(std::max(0.0f, dot(sample_nml, center_normal)), 128.0f);
                    const float Wm = center_meshid == sample_meshid ? 1.0f : 0.0f;
                    const float W = std::exp(-Wz) * Wn * Wm;
                    moment_sum += moment * W;
                    color += sample_color * W;
                    weight += W;
                }
            }
        }
        moment_sum /= weight;
        color /= weight;
        variance = std::max(0.0f, moment_sum.x - moment_sum.y * moment_sum.y);
    }
    else {
        variance = std::max(0.0f, center_moment.x - center_moment.y * center_moment.y);
    }
    color.w = variance;
    aov_cv[idx] = color;
    return v4(variance, variance, variance, 1);
}

// Exec3x3GaussFilter on .w (svgf_impl.h:558-611)
inline float Gauss3x3W(int32_t ix, int32_t iy, int32_t width, int32_t height, const std::vector<v4>& buffer)
{
    static constexpr float kernel[] = { 1.0 / 16.0, 1.0 / 8.0, 1.0 / 16.0, 1.0 / 8.0, 1.0 / 4.0, 1.0 / 8.0, 1.0 / 16.0, 1.0 / 8.0, 1.0 / 16.0 };
    static constexpr int32_t offsetx[] = { -1, 0, 1, -1, 0, 1, -1, 0, 1 };
    static constexpr int32_t offsety[] = { -1, -1, -1, 0, 0, 0, 1, 1, 1 };
    float sum = 0;
    for (int32_t i = 0; i < 9; i++) {
        int32_t xx = clampv(ix + offsetx[i], 0, width - 1);
        int32_t yy = clampv(iy + offsety[i], 0, height - 1);
        float tmp = buffer[xx + yy * width].w;
        sum += kernel[i] * tmp;
    }
    return sum;
}

// SVGFRenderer::AtrousFilter (svgf.cpp:328-410) = ExtractCenterPixel<false> + CheckIfBackgroundPixelForAtrous +
// Exec3x3GaussFilter + ExecAtrousWaveletFilter + PostProcessForAtrousFilter.  Returns true and *out when the
// reference's optional holds a colour (final iteration).
inline bool AtrousFilter(int32_t iter, int32_t ix, int32_t iy, int32_t width, int32_t height, float camera_distance,
    Params& p, v4* out)
{
    const int32_t idx = ix + iy * width;
    const int32_t curr = iter & 0x01, next = 1 - curr;
    const bool isFirstIter = (iter == 0);
    const bool isFinalIter = (iter == p.atrous_iter_cnt - 1);
    auto& aov_nd = p.cur(NormalDepth); auto& aov_am = p.cur(AlbedoMeshId); auto& aov_cv = p.cur(ColorVariance);
    const std::vector<v4>& src = isFirstIter ? aov_cv : p.atrous_clr_variance[curr];
    std::vector<v4>& dst = p.atrous_clr_variance[next];

    const v4 nml_depth = aov_nd[idx];
    const float center_depth = nml_depth.w;
    const int32_t center_meshid = static_cast<int32_t>(aov_am[idx].w);
    const v4 contrib = src[idx];
    v4 center_color(contrib.x, contrib.y, contrib.z, 1.0f);
    const v3 center_normal(nml_depth.x, nml_depth.y, nml_depth.z);

    // CheckIfBackgroundPixelForAtrous (svgf_impl.h:625-657): compares the float, not the int
    if (aov_am[idx].w < 0) {
        dst[idx] = v4(center_color.x, center_color.y, center_color.z, 0.0f);
        if (isFinalIter) {
            v4 r = aov_am[idx];
            r = r * center_color;
            *out = r;
            return true;
        }
        // not the final iteration: the outer optional is engaged but empty, the caller falls through and filters anyway
    }

    const float gauss_filtered_variance = Gauss3x3W(ix, iy, width, height, src);

    // ExecAtrousWaveletFilter (svgf_impl.h:680-800)
    static constexpr float sigmaZ = 1.0f, sigmaN = 128.0f, sigmaL = 4.0f;
    static constexpr float h[] = {
        2.0f / 3.0f,  2.0f / 3.0f,  2.0f / 3.0f,  2.0f / 3.0f,  1.0f / 6.0f,  1.0f / 6.0f,  1.0f / 6.0f,  1.0f / 6.0f,
        4.0f / 9.0f,  4.0f / 9.0f,  4.0f / 9.0f,  4.0f / 9.0f,  1.0f / 9.0f,  1.0f / 9.0f,  1.0f / 9.0f,  1.0f / 9.0f,
        1.0f / 9.0f,  1.0f / 9.0f,  1.0f / 9.0f,  1.0f / 9.0f,  1.0f / 36.0f, 1.0f / 36.0f, 1.0f / 36.0f, 1.0f / 36.0f };
    static constexpr int32_t offsetx[] = { 1, 0, -1, 0, 2, 0, -2, 0, 1, -1, -1, 1, 1, -1, -1, 1, 2, -2, -2, 2, 2, -2, -2, 2 };
    static constexpr int32_t offsety[] = { 0, 1, 0, -1, 0, 2, 0, -2, 1, 1, -1, -1, 2, 2, -2, -2, 1, 1, -1, -1, 2, 2, -2, -2 };
    const int32_t step_scale = 1 << iter;
    float sqrt_gauss_filtered_variance = std::sqrt(gauss_filtered_variance);
    float center_luminance = luminance(center_color.x, center_color.y, center_color.z);
    v4 sumC = center_color;
    float sumV = center_color.w;
    float weight = 1.0f;
    const float pixel_distance_ratio = (center_depth / camera_distance) * height;
    for (int32_t i = 0; i < 24; i++) {
        int32_t sx = offsetx[i] * step_scale, sy = offsety[i] * step_scale;
        int32_t xx = clampv(ix + sx, 0, width - 1);
        int32_t yy = clampv(iy + sy, 0, height - 1);
        const float u_length = std::sqrt(static_cast<float>(sx * sx + sy * sy));
        const int32_t qidx = xx + yy * width;
        const v4& q_nd = aov_nd[qidx];
        const v4& q_am = aov_am[qidx];
        const v3 normal(q_nd.x, q_nd.y, q_nd.z);
        const float depth = q_nd.w;
        const int32_t meshid = static_cast<int32_t>(q_am.w);
        const v4& color = src[qidx];
        const float variance = color.w;
        float lum = luminance(color.x, color.y, color.z);
        float Wz = 3.0f * std::fabs(center_depth - depth) / (sigmaZ * (pixel_distance_ratio * u_length) + 0.000001f);
        float Wn = powf(std::max(0.0f, dot(center_normal, normal)), sigmaN);
        float Wl = std::min(expf(-std::fabs(center_luminance - lum) / (sigmaL * sqrt_gauss_filtered_variance + 0.000001f)), 1.0f);
        float Wm = meshid == center_meshid ? 1.0f : 0.0f;
        float W = expf(-Wl * Wl - Wz) * Wn * Wm * h[i];
        sumC += W * color;
        sumV += W * W * variance;
        weight += W;
    }
    sumC /= weight;
    sumV /= (weight * weight);
    const v4 filtered(sumC.x, sumC.y, sumC.z, sumV);

    // PostProcessForAtrousFilter (:817-843)
    dst[idx] = filtered;
    if (isFirstIter) {
        p.temporary_color_buffer[idx].x = filtered.x;
        p.temporary_color_buffer[idx].y = filtered.y;
        p.temporary_color_buffer[idx].z = filtered.z;
    }
    if (isFinalIter) {
        v4 r = aov_am[idx];
        r = r * filtered;
        *out = r;
        return true;
    }
    return false;
}

// The pass that stands in for the reference's GL raster pass (src/shader/ssrt_fs.glsl:31-47 with
// prev/cur clip positions from W2C matrices; static geometry: mtx_prev_L2W == mtx_L2W):
//   motion = prevNDC01 - curNDC01, z = view depth (clip w), w = 1.
inline v4 ComputeMotionDepth(const v4& world_pos, const m4& W2C, const m4& prevW2C)
{
    if (world_pos.w == 0.0f) return v4(0.0f, 0.0f, -1.0f, 1.0f);
    const v4 cur = W2C.apply(v4(world_pos.x, world_pos.y, world_pos.z, 1.0f));
    const v4 prv = prevW2C.apply(v4(world_pos.x, world_pos.y, world_pos.z, 1.0f));
    const float cx = (cur.x / cur.w) * 0.5f + 0.5f, cy = (cur.y / cur.w) * 0.5f + 0.5f;
    const float px = (prv.x / prv.w) * 0.5f + 0.5f, py = (prv.y / prv.w) * 0.5f + 0.5f;
    return v4(px - cx, py - cy, cur.w, 1.0f);
}

} // namespace svgf
} // namespace orc
