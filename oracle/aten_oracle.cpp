/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * C entry points (ctypes) over the CPU restatement in orc_*.h.  Loaded only by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by aten_amd/.
 * Parity status: see orc_core.h header ("parity unpinned" except the three reference KATs).
 */
#include "orc_pt.h"
#include "orc_svgf.h"
#include "orc_lbvh.h"
#include <cstdio>
#include <omp.h>

using namespace orc;

extern "C" {

// aten::initSampler, sampler/sampler.cpp:8-18
void orc_init_sampler(uint32_t* out, int w, int h, int seed)
{
    std::vector<uint32_t> s;
    init_sampler(s, w, h, seed);
    std::memcpy(out, s.data(), s.size() * 4);
}

// n successive CMJ::nextSample() values after init(index, dimension, scramble)
void orc_cmj_samples(uint32_t index, uint32_t dimension, uint32_t scramble, int n, float* out)
{
    CMJ c; c.init(index, dimension, scramble);
    for (int i = 0; i < n; i++) out[i] = c.nextSample();
}

// n successive CMJ::nextSample2D(): out[2i] = x, out[2i+1] = y (cmj.h:39-44)
void orc_cmj_samples2d(uint32_t index, uint32_t dimension, uint32_t scramble, int n, float* out)
{
    CMJ c; c.init(index, dimension, scramble);
    for (int i = 0; i < n; i++) c.nextSample2D(out[2 * i], out[2 * i + 1]);
}

// per triple k: init(idx[k], dim[k], scr[k]) then `draws` x nextSample()  (twin of oracle/ref_driver.cpp)
void orc_cmj_batch(int32_t n, const uint32_t* idx, const uint32_t* dim, const uint32_t* scr, int32_t draws, float* out)
{
    for (int32_t k = 0; k < n; k++) {
        CMJ c; c.init(idx[k], dim[k], scr[k]);
        for (int32_t d = 0; d < draws; d++) out[(size_t)k * draws + d] = c.nextSample();
    }
}

// The oracle's restatements of math/math.h's scalar helpers, by the kinds of ref_math_kat (oracle/ref_driver.cpp),
// so that tests/test_ref_pin.py can hold them against the reference's own header.
void orc_math_kat(int32_t kind, int32_t n, const float* a, const float* b, const float* c, float* out)
{
    for (int32_t i = 0; i < n; i++) {
        switch (kind) {
        case 0: out[i] = fmax_(a[i], b[i]); break;                       // math.h:146-150
        case 1: out[i] = fmin_(a[i], b[i]); break;                       // math.h:172-176
        case 2: out[i] = clamp_(a[i], b[i], c[i]); break;                // math.h:179-183
        case 3: out[i] = saturate_(a[i]); break;                         // math.h:185-189
        case 4: out[i] = ToonSpecular::sign(a[i]); break;                             // math.h:62-73
        case 5: out[i] = mix(a[i], b[i], c[i]); break;                   // math.h:294-300
        case 6: out[i] = a[i] * (1.0F - c[i]) + b[i] * c[i]; break;      // math.h:191-194 (lerp, as in orc_core.h's bilinear)
        case 7: out[i] = isCloseUlps(a[i], b[i], 2500) ? 1.0F : 0.0F; break;   // math.h:308-340
        case 8: out[i] = (std::isnan(a[i]) || std::isinf(a[i])) ? 1.0F : 0.0F; break;   // math.h:196-204
        case 9: out[i] = sqr(a[i]); break;                               // math.h:42-45
        case 10: out[i] = inversesqrt(a[i]); break;                      // math.h:33-40
        case 11: out[i] = Deg2Rad(a[i]); break;                          // math.h:18-21
        default: out[i] = 0.0F;
        }
    }
}

// The math-library functions the float path calls, as THIS build's libm answers them (the oracle's side of the ulp study,
// tools/ulp_study.py: which function's last bit the GPU's ocml disagrees on).  kind: 0 sinf 1 cosf 2 atanf 3 acosf 4 atan2f(a, b)
// 5 logf 6 expf 7 powf(a, b) 8 sqrtf 9 a / b 10 1 / sqrtf(a) as orc_math.h's inversesqrt writes it
void orc_libm_probe(int32_t kind, int32_t n, const float* a, const float* b, float* out)
{
    for (int32_t i = 0; i < n; i++) {
        switch (kind) {
        case 0: out[i] = std::sin(a[i]); break;
        case 1: out[i] = std::cos(a[i]); break;
        case 2: out[i] = std::atan(a[i]); break;
        case 3: out[i] = std::acos(a[i]); break;
        case 4: out[i] = std::atan2(a[i], b[i]); break;
        case 5: out[i] = std::log(a[i]); break;
        case 6: out[i] = std::exp(a[i]); break;
        case 7: out[i] = std::pow(a[i], b[i]); break;
        case 8: out[i] = std::sqrt(a[i]); break;
        case 9: out[i] = a[i] / b[i]; break;
        case 10: out[i] = inversesqrt(a[i]); break;
        default: out[i] = 0.0F;
        }
    }
}

void orc_create_camera(atn_camera_param* out, const float* origin, const float* lookat, const float* up,
    float vfov, float z_near, float z_far, int32_t width, int32_t height)
{
    *out = CreateCameraParam(ld3(origin), ld3(lookat), ld3(up), vfov, z_near, z_far, width, height);
}

float orc_pixel_width_at_distance(const atn_camera_param* cam, float dist)
{
    return ComputePixelWidthAtDistance(*cam, dist);
}

void orc_ray_offset(const float* o, const float* n, int count, float* out)
{
    for (int i = 0; i < count; i++) {
        v3 r = Ray::Offset(ld3(o + 3 * i), ld3(n + 3 * i));
        out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
    }
}

// GeneratePath for every pixel of a w x h image (sample index `sample`, frame `frame`).
void orc_generate_paths(const atn_camera_param* cam, const uint32_t* seeds, uint32_t n_seeds,
    int w, int h, int sample, uint32_t frame, atn_ray* rays_out)
{
#pragma omp parallel for
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            int idx = y * w + x;
            PathState p; Ray r;
            GeneratePath(r, x, y, sample, frame, p, *cam, seeds[idx % n_seeds]);
            atn_ray& o = rays_out[idx];
            o.org[0] = r.org.x; o.org[1] = r.org.y; o.org[2] = r.org.z;
            o.dir[0] = r.dir.x; o.dir[1] = r.dir.y; o.dir[2] = r.dir.z;
        }
    }
}

// Traverse<Closest> for a batch of rays.  stats_out = {node visits, triangle tests} (may be null)
void orc_trace_closest(const atn_scene_desc* scene, const atn_ray* rays, uint32_t n,
    float t_min, float t_max, atn_intersection* out, uint64_t* stats_out)
{
    Scene ctxt(scene);
    uint64_t nodes = 0, tris = 0;
#pragma omp parallel for reduction(+:nodes,tris)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        Ray r; r.org = ld3(rays[i].org); r.dir = ld3(rays[i].dir);
        Isect is; TraverseStats st;
        TraverseClosest(is, ctxt, r, t_min, t_max, &st);
        nodes += st.nodes; tris += st.tris;
        atn_intersection& o = out[i];
        o.t = is.t; o.objid = is.objid; o.mtrlid = is.mtrlid; o.meshid = is.meshid;
        o.tri_id = is.tri_id; o.a = is.a; o.b = is.b; o.isVoxel = is.isVoxel;
    }
    if (stats_out) { stats_out[0] = nodes; stats_out[1] = tris; }
}

// evaluate_hit_result for a batch of (ray, intersection): out = {p.xyz, area, n.xyz, u, v} (9 floats)
void orc_evaluate_hits(const atn_scene_desc* scene, const atn_ray* rays, const atn_intersection* isects,
    uint32_t n, float* out)
{
    Scene ctxt(scene);
    for (uint32_t i = 0; i < n; i++) {
        Isect is; is.t = isects[i].t; is.objid = isects[i].objid; is.mtrlid = isects[i].mtrlid;
        is.meshid = isects[i].meshid; is.tri_id = isects[i].tri_id; is.a = isects[i].a; is.b = isects[i].b;
        float* o = out + 9 * i;
        if (is.objid < 0) { for (int k = 0; k < 9; k++) o[k] = 0; continue; }
        Ray r; r.org = ld3(rays[i].org); r.dir = ld3(rays[i].dir);
        HitRec rec;
        evaluate_hit_result(rec, ctxt.GetObject(is.objid), ctxt, r, is);
        o[0] = rec.p.x; o[1] = rec.p.y; o[2] = rec.p.z; o[3] = rec.area;
        o[4] = rec.normal.x; o[5] = rec.normal.y; o[6] = rec.normal.z; o[7] = rec.u; o[8] = rec.v;
    }
}

// BSDF tables.  For case i: normal n[i], incoming wi[i], sampler (index[i], dimension 0, scramble[i]),
// uv[i].  out_sample = {dir.xyz, bsdf.xyz, pdf} ; then pdf/bsdf re-evaluated at wo = sampled dir:
// out_eval = {samplePDF, sampleBSDF.bsdf.xyz, sampleBSDF.pdf}.
void orc_material_table(const atn_scene_desc* scene, int32_t mtrl_id, uint32_t n,
    const float* nrm, const float* wi, const uint32_t* index, const uint32_t* scramble, const float* uv,
    float* out_sample, float* out_eval)
{
    Scene ctxt(scene);
    atn_material_param m;
    FillMaterial(m, ctxt, mtrl_id);
    for (uint32_t i = 0; i < n; i++) {
        CMJ s; s.init(index[i], 0, scramble[i]);
        MaterialSampling ms;
        v3 N = ld3(nrm + 3 * i), WI = ld3(wi + 3 * i);
        float pre_r = 0.0F;
        if (m.type == ATN_MTRL_CARPAINT) {      // material::applyNormal runs first, as in shade
            v3 nn;
            pre_r = applyNormal(ctxt, m, N, nn, uv[2 * i], uv[2 * i + 1], WI, &s);
            N = nn;
        }
        sampleMaterial(&ms, ctxt, &m, N, WI, &s, uv[2 * i], uv[2 * i + 1], pre_r);
        float* o = out_sample + 7 * i;
        o[0] = ms.dir.x; o[1] = ms.dir.y; o[2] = ms.dir.z;
        o[3] = ms.bsdf.x; o[4] = ms.bsdf.y; o[5] = ms.bsdf.z; o[6] = ms.pdf;
        float p = samplePDF(ctxt, &m, N, WI, ms.dir, uv[2 * i], uv[2 * i + 1]);
        MaterialSampling ev = sampleBSDF(ctxt, &m, N, WI, ms.dir, uv[2 * i], uv[2 * i + 1], pre_r);
        float* e = out_eval + 5 * i;
        e[0] = p; e[1] = ev.bsdf.x; e[2] = ev.bsdf.y; e[3] = ev.bsdf.z; e[4] = ev.pdf;
    }
}

struct orc_destination {    // aten::Destination (renderer/renderer.h:15-23) + explicit frame
    int32_t width, height, maxDepth, russianRouletteDepth, sample;
    uint32_t frame;
    int32_t progressive;    // 1: FilmProgressive::put (film.cpp:61-71), 0: Film::put (film.cpp:33-45)
    int32_t nthreads;       // OMPUtil::setThreadNum; <=0 = all
};

// aten::PathTracing::OnRender, renderer/pathtracing/pathtracing.cpp:269-366.
// film: vec4[w*h], row 0 = bottom.  counters_out (may be null): {closest rays, shadow rays, hits,
// node visits, triangle tests}.
// Optional samplers of the product (atn_set_sampling_options), so that they too have a CPU twin to be compared with.
void orc_set_sampling_options(int32_t ibl_importance, int32_t tex_bilinear)
{
    sampling_options().ibl_importance = ibl_importance;
    sampling_options().tex_bilinear = tex_bilinear;
}

// texture::at / texture::AtWithBilinear for n lookups of one texture (current texture mode)
void orc_sample_texture(const atn_scene_desc* scene, int32_t texid, uint32_t n, const float* uv, float* out)
{
    Scene ctxt(scene);
    for (uint32_t i = 0; i < n; i++) {
        const v4 c = sampleTexture(ctxt, texid, uv[2 * i], uv[2 * i + 1], v4(0.0F));
        out[4 * i] = c.x; out[4 * i + 1] = c.y; out[4 * i + 2] = c.z; out[4 * i + 3] = c.w;
    }
}

// ImageBasedLight::preCompute's tables (cdfV[h], cdfU[h*w]) for the scene's environment map
int32_t orc_ibl_tables(const atn_scene_desc* scene, float* cdf_v, float* cdf_u)
{
    Scene ctxt(scene);
    if (ctxt.cfg().bg.envmap_tex_idx < 0) return -1;
    SamplingOptions& o = sampling_options();
    IBL_preCompute(o, ctxt);
    for (int32_t y = 0; y < o.h; y++) {
        cdf_v[y] = o.cdfV[y];
        for (int32_t x = 0; x < o.w; x++) cdf_u[(size_t)y * o.w + x] = o.cdfU[y][x];
    }
    return 0;
}

// cost_out (may be null): uint32 {BVH node visits, triangle tests}[h][w] of all the pixel's walks of this frame -- what the
// product's atn_download_path_cost reports (its stand-in for the reference's PathTimeProfiler heat map)
void orc_render_cost(const atn_scene_desc* scene, const atn_camera_param* camera,
    const uint32_t* seeds, uint32_t n_seeds, const orc_destination* dst, atn_vec4* film,
    uint64_t* counters_out, uint32_t* cost_out)
{
    Scene ctxt(scene);
    if (sampling_options().ibl_importance && ctxt.cfg().bg.envmap_tex_idx >= 0 && ctxt.cfg().bg.enable_env_map)
        IBL_preCompute(sampling_options(), ctxt);
    const int32_t width = dst->width, height = dst->height;
    const uint32_t samples = (uint32_t)dst->sample;
    int32_t maxDepth = dst->maxDepth;
    int32_t rrDepth = dst->russianRouletteDepth;
    if (rrDepth > maxDepth) rrDepth = maxDepth - 1;     // pathtracing.cpp:282-284

    uint64_t c_closest = 0, c_shadow = 0, c_hits = 0, c_nodes = 0, c_tris = 0;
    const bool count = counters_out != nullptr || cost_out != nullptr;
    if (dst->nthreads > 0) omp_set_num_threads(dst->nthreads);

#pragma omp parallel for reduction(+:c_closest,c_shadow,c_hits,c_nodes,c_tris)
    for (int32_t y = 0; y < height; y++) {
        for (int32_t x = 0; x < width; x++) {
            v3 col(0); uint32_t cnt = 0;
            int32_t idx = y * width + x;
            PathState path;     // path_host_.Clear() zeroes contrib/attrib each frame (pt_params.h:101-119)
            path.samples = 0;
            Ray ray; ShadowRay shadow_ray;
            PathCounters pc;
            for (uint32_t i = 0; i < samples; i++) {
                const uint32_t rnd = seeds[idx % n_seeds];
                GeneratePath(ray, x, y, (int32_t)i, dst->frame, path, *camera, rnd);
                path.contrib = v3(0);
                radiance(path, ray, shadow_ray, x, y, width, height, ctxt, *camera, maxDepth, rrDepth, count ? &pc : nullptr);
                if (isInvalidColor(path.contrib)) continue;
                col += path.contrib;
                cnt++;
                if (path.is_terminated) break;          // pathtracing.cpp:350-352
            }
            col /= (float)cnt;
            v4 v(col, 1);
            atn_vec4& cur = film[idx];
            if (dst->progressive) {
                float n = static_cast<float>(static_cast<int32_t>(cur.w));
                v4 c(cur.x, cur.y, cur.z, cur.w);
                c = n * c + v;
                float d = n + 1;
                cur.x = c.x / d; cur.y = c.y / d; cur.z = c.z / d;
                cur.w = n + 1;
            }
            else {
                cur.x = v.x; cur.y = v.y; cur.z = v.z; cur.w = v.w;
            }
            c_closest += pc.closest_rays; c_shadow += pc.shadow_rays; c_hits += pc.hits;
            c_nodes += pc.trav.nodes; c_tris += pc.trav.tris;
            if (cost_out) { cost_out[2 * (size_t)idx] = (uint32_t)pc.trav.nodes; cost_out[2 * (size_t)idx + 1] = (uint32_t)pc.trav.tris; }
        }
    }
    if (counters_out) {
        counters_out[0] = c_closest; counters_out[1] = c_shadow; counters_out[2] = c_hits;
        counters_out[3] = c_nodes; counters_out[4] = c_tris;
    }
}

void orc_render(const atn_scene_desc* scene, const atn_camera_param* camera,
    const uint32_t* seeds, uint32_t n_seeds, const orc_destination* dst, atn_vec4* film,
    uint64_t* counters_out)
{
    orc_render_cost(scene, camera, seeds, n_seeds, dst, film, counters_out, nullptr);
}

int orc_num_procs() { return omp_get_num_procs(); }

// ---------------------------------------------------------------------------------------------------
// SVGF (next tier): aten::SVGFRenderer, src/libaten/renderer/svgf/svgf.cpp:412-637
// ---------------------------------------------------------------------------------------------------
void* orc_svgf_create() { return new svgf::Params(); }
void orc_svgf_destroy(void* h) { delete static_cast<svgf::Params*>(h); }
void orc_svgf_set_atrous_iterations(void* h, int32_t n) { static_cast<svgf::Params*>(h)->atrous_iter_cnt = n; }
void orc_svgf_set_dilate_temporal_weight(void* h, int32_t on) { static_cast<svgf::Params*>(h)->dilate_temporal_weight = on; }

// aten::FillBasicAOVs / FillBasicAOVsIfHitMiss (renderer/aov.h:158-198) on their own, for the reference's known answers
// (aten_unittest/aov_host_buffer.cpp:73-132).  w2c: 16 floats, row-major (aten::mat4)
void orc_fill_basic_aovs(const float* normal, const float* p, const float* w2c, const float* albedo, int32_t meshid, float* out_nd, float* out_am)
{
    m4 m;
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) m.m[r][c] = w2c[4 * r + c];
    v4 nd, am;
    svgf::FillBasicAOVs(nd, v3(normal[0], normal[1], normal[2]), v3(p[0], p[1], p[2]), m, am, v4(albedo[0], albedo[1], albedo[2], albedo[3]), meshid);
    out_nd[0] = nd.x; out_nd[1] = nd.y; out_nd[2] = nd.z; out_nd[3] = nd.w;
    out_am[0] = am.x; out_am[1] = am.y; out_am[2] = am.z; out_am[3] = am.w;
}
void orc_fill_basic_aovs_if_hit_miss(const float* bg, float* out_nd, float* out_am)
{
    v4 nd, am;
    svgf::FillBasicAOVsIfHitMiss(nd, am, v4(bg[0], bg[1], bg[2], bg[3]));
    out_nd[0] = nd.x; out_nd[1] = nd.y; out_nd[2] = nd.z; out_nd[3] = nd.w;
    out_am[0] = am.x; out_am[1] = am.y; out_am[2] = am.z; out_am[3] = am.w;
}

// SVGFRenderer::SetMotionDepthBuffer (svgf.cpp:441-450): vec4 {motion.xy in screen fractions, depth, 1} per pixel
void orc_svgf_set_motion_depth(void* h, const atn_vec4* md, uint32_t n)
{
    auto* p = static_cast<svgf::Params*>(h);
    p->motion_depth_buffer.resize(n);
    for (uint32_t i = 0; i < n; i++) p->motion_depth_buffer[i] = v4(md[i].x, md[i].y, md[i].z, md[i].w);
}

// which: 0-3 current AOVs, 4-7 previous AOVs, 8 temporary colour, 9 motion/depth, 10 primary hit position,
// 11/12 the two a-trous ping-pong buffers
int orc_svgf_get_buffer(void* h, int32_t which, atn_vec4* out)
{
    auto* p = static_cast<svgf::Params*>(h);
    const std::vector<v4>* b = nullptr;
    if (which >= 0 && which < 4) b = &p->cur(which);
    else if (which < 8) b = &p->prev(which - 4);
    else if (which == 8) b = &p->temporary_color_buffer;
    else if (which == 9) b = &p->motion_depth_buffer;
    else if (which == 10) b = &p->primary_position;
    else if (which == 11 || which == 12) b = &p->atrous_clr_variance[which - 11];
    if (!b) return -1;
    for (size_t i = 0; i < b->size(); i++) { out[i].x = (*b)[i].x; out[i].y = (*b)[i].y; out[i].z = (*b)[i].z; out[i].w = (*b)[i].w; }
    return (int)b->size();
}

// SVGFRenderer::OnRender (svgf.cpp:452-637).  `film` receives what dst.buffer holds when OnRender returns (Film::put,
// film.cpp:33-45): the last a-trous iteration's albedo-multiplied colour.  `stage_out` (may be null): 3 x vec4[w*h] =
// the values put after the path pass, after the temporal pass and after the variance pass.
// compute_motion != 0: fill motion_depth_buffer from the primary hits (ComputeMotionDepth) instead of an external buffer.
void orc_svgf_render(void* h, const atn_scene_desc* scene, const atn_camera_param* camera,
    const uint32_t* seeds, uint32_t n_seeds, const orc_destination* dst, int32_t compute_motion,
    atn_vec4* film, atn_vec4* stage_out)
{
    auto& P = *static_cast<svgf::Params*>(h);
    Scene ctxt(scene);
    const int32_t width = dst->width, height = dst->height;
    const uint32_t samples = (uint32_t)dst->sample;
    int32_t maxDepth = dst->maxDepth;
    int32_t rrDepth = dst->russianRouletteDepth;
    if (rrDepth > maxDepth) rrDepth = maxDepth - 1;         // svgf.cpp:423-425
    P.InitBuffers(width, height);
    P.mtxs.Reset(*camera);
    const m4 W2C = P.mtxs.GetW2C();
    const m4 prevW2C = svgf::mul(P.mtxs.V2C, P.mtxs.PrevW2V);
    const size_t n = (size_t)width * height;
    std::vector<v4> contribs(n);
    auto put = [&](int stage, int32_t idx, const v4& v) {
        if (stage_out && stage < 3) { atn_vec4& o = stage_out[(size_t)stage * n + idx]; o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w; }
        atn_vec4& o = film[idx]; o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w;
    };
    if (dst->nthreads > 0) omp_set_num_threads(dst->nthreads);

#pragma omp parallel for
    for (int32_t y = 0; y < height; y++) {
        for (int32_t x = 0; x < width; x++) {
            int32_t idx = y * width + x;
            PathState path;             // path_host_.Clear(): contrib/attrib/throughput zeroed every frame
            path.samples = 0;
            Ray ray; ShadowRay shadow_ray;
            for (uint32_t i = 0; i < samples; i++) {
                const uint32_t rnd = seeds[idx % n_seeds];
                GeneratePath(ray, x, y, (int32_t)i, dst->frame, path, *camera, rnd);
                path.contrib = v3(0);
                svgf::ExecRendering(path, ray, shadow_ray, x, y, width, height, ctxt, *camera, maxDepth, rrDepth, W2C,
                    P.cur(svgf::NormalDepth)[idx], P.cur(svgf::AlbedoMeshId)[idx], P.primary_position[idx], nullptr);
                if (isInvalidColor(path.contrib)) continue;
                if (path.is_terminated) break;
            }
            // PrepareForDenoise (svgf_impl.h:119-144)
            const v4 c(path.contrib, path.samples);
            v4 contrib = c;
            svgf::operator/=(contrib, c.w);
            if (dst->frame == 0) {
                float lum = luminance(contrib.x, contrib.y, contrib.z);
                v4& mt = P.cur(svgf::MomentTemporalWeight)[idx];
                mt.x += lum * lum; mt.y += lum; mt.z += 1;
                v4& cv = P.cur(svgf::ColorVariance)[idx];
                cv = v4(contrib.x, contrib.y, contrib.z, cv.w);
            }
            P.temporary_color_buffer[idx] = c;
            contribs[idx] = c;
            put(0, idx, v4(path.contrib, 1));       // Film::put(x, y, vec3) -> vec4(v, 1)
        }
    }

    if (compute_motion) {
        P.motion_depth_buffer.resize(n);
#pragma omp parallel for
        for (int32_t i = 0; i < (int32_t)n; i++) P.motion_depth_buffer[i] = svgf::ComputeMotionDepth(P.primary_position[i], W2C, prevW2C);
    }

#pragma omp parallel for
    for (int32_t y = 0; y < height; y++) {
        for (int32_t x = 0; x < width; x++) {
            int32_t idx = y * width + x;
            v4 temporal_projected_clr;
            if (dst->frame > 0) temporal_projected_clr = svgf::TemporalReprojection(x, y, width, height, 0.98f, 0.05f, contribs[idx], P);
            else temporal_projected_clr = v4(contribs[idx].x, contribs[idx].y, contribs[idx].z, 1.0f);   // vec4() then operator=(vec3): w stays 1
            put(1, idx, temporal_projected_clr);
        }
    }

    if (P.dilate_temporal_weight && dst->frame > 0) {
        const std::vector<v4> mt_in = P.cur(svgf::MomentTemporalWeight);
        std::vector<v4>& mt = P.cur(svgf::MomentTemporalWeight);
#pragma omp parallel for
        for (int32_t y = 0; y < height; y++)
            for (int32_t x = 0; x < width; x++) {
                float w;
                if (svgf::RecomputeTemporalWeightFromSurroundingPixels(x, y, width, height, P.cur(svgf::AlbedoMeshId), mt_in, w))
                    mt[y * width + x].w = w;
            }
    }

    // Camera::ComputeScreenDistance (camera.h:216-221): tan() of half the vertical fov IN DEGREES, as written
    const float camera_distance = height / (2.0f * std::tan(0.5f * camera->vfov));
    {
        const std::vector<v4> cv_in = P.cur(svgf::ColorVariance);
#pragma omp parallel for
        for (int32_t y = 0; y < height; y++) {
            for (int32_t x = 0; x < width; x++) {
                put(2, y * width + x, svgf::EstimateVariance(x, y, width, height, camera_distance, cv_in, P));
            }
        }
    }

    for (int32_t i = 0; i < P.atrous_iter_cnt; i++) {
#pragma omp parallel for
        for (int32_t y = 0; y < height; y++) {
            for (int32_t x = 0; x < width; x++) {
                v4 out;
                if (svgf::AtrousFilter(i, x, y, width, height, camera_distance, P, &out)) put(3, y * width + x, out);
            }
        }
    }

    // CopyFromTeporaryColorBufferToAov: CopyVectorBuffer<3> (svgf.cpp:402-410)
    {
        auto& cv = P.cur(svgf::ColorVariance);
        for (size_t i = 0; i < n; i++) { cv[i].z = P.temporary_color_buffer[i].z; cv[i].y = P.temporary_color_buffer[i].y; cv[i].x = P.temporary_color_buffer[i].x; }
    }
    P.curr_aov_pos = 1 - P.curr_aov_pos;
}

// idaten::LBVHBuilder::build with threadedBvhNodes (LBVHBuilder.cu:812-833): the node array in the reference's order
int orc_lbvh_build(const atn_triangle_param* tris, uint32_t n, int32_t tri_id_offset, const float* bmin, const float* bmax,
                   const atn_vec4* vtx_pos, int32_t vtx_offset, atn_bvh_node* out_nodes, uint32_t* out_codes, uint32_t* out_indices)
{
    return orc::lbvh::build(tris, n, tri_id_offset, bmin, bmax, vtx_pos, vtx_offset, out_nodes, out_codes, out_indices) ? 0 : -1;
}

// buildTree alone on given SORTED keys (LBVHBuilder.cu:299-350): left / right / parent of the 2 n - 1 nodes
int orc_lbvh_hierarchy(const uint32_t* sorted_keys, uint32_t n, int32_t* left, int32_t* right, int32_t* parent)
{
    if (n < 2) return -1;
    std::vector<orc::lbvh::Node> nodes;
    orc::lbvh::build_tree(sorted_keys, n, nodes);
    for (size_t i = 0; i < nodes.size(); i++) { left[i] = nodes[i].left; right[i] = nodes[i].right; parent[i] = nodes[i].parent; }
    return 0;
}

} // extern "C"
