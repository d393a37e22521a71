/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h).
 *
 * orc_core.h: CPU restatement of aten's path-tracing hot path over the flat scene arrays of
 * include/aten_layout.h.  Every function cites the reference file:line it follows
 * (paths relative to /root/reference/src/libaten unless noted).
 *
 * PARITY STATUS: the reference cannot be built in this image (glm, tinyobjloader, stb,
 * nanovdb submodules are empty) and its own tests hold no golden vectors for this path except
 *   - camera_test  ComputePixelWidthAtDistance == 0.000473979220  (src/aten_unittest/pinhole_camera.cpp:6-16)
 *   - the commented scan self-test in src/libidaten/kernel/StreamCompaction.cu:318-400
 *   - the on-disk asset/sponza/sponza_lod.sbvh fixture (format: accelerator/sbvh.cpp:1220-1338)
 * which tests/ check.  Everything else is "parity unpinned": this file IS the definition.
 */
#pragma once
#include "orc_math.h"
#include "../include/aten_layout.h"
#include <vector>
#include <random>

namespace orc {

// ---------------------------------------------------------------------------------------
// Sampler: sampler/sampler.cpp:8-28, sampler/cmj.h:9-124
// ---------------------------------------------------------------------------------------
inline void init_sampler(std::vector<uint32_t>& seeds, int w, int h, int seed)
{
    seeds.resize(size_t(w) * h);
    std::mt19937 src(seed);
    for (auto& s : seeds) s = src();
}

struct CMJ {
    uint32_t m_idx{ 0 }, m_dimension{ 0 }, m_scramble{ 0 };
    enum { CMJ_DIM = 16 };

    void init(uint32_t index, uint32_t dimension, uint32_t scramble)
    {
        m_idx = index; m_dimension = dimension; m_scramble = scramble;
    }
    static uint32_t permute(uint32_t i, uint32_t l, uint32_t p)    // cmj.h:51-85
    {
        uint32_t w = l - 1;
        w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16;
        do {
            i ^= p;             i *= 0xe170893d;
            i ^= p >> 16;       i ^= (i & w) >> 4;
            i ^= p >> 8;        i *= 0x0929eb3f;
            i ^= p >> 23;       i ^= (i & w) >> 1;
            i *= 1 | p >> 27;   i *= 0x6935fa69;
            i ^= (i & w) >> 11; i *= 0x74dcb303;
            i ^= (i & w) >> 2;  i *= 0x9e501cc3;
            i ^= (i & w) >> 2;  i *= 0xc860a3df;
            i &= w;
            i ^= i >> 5;
        } while (i >= l);
        return (i + p) % l;
    }
    static float randfloat(uint32_t i, uint32_t p)                  // cmj.h:87-101
    {
        i ^= p;
        i ^= i >> 17; i ^= i >> 10; i *= 0xb36534e5;
        i ^= i >> 12; i ^= i >> 21; i *= 0x93fc4795;
        i ^= 0xdf6e307f;
        i ^= i >> 17; i *= 1 | p >> 18;
        return i * (1.0f / 4294967808.0f);
    }
    // cmj.h:103-114 (both components; nextSample only uses .x)
    static void cmj(int32_t s, int32_t n, int32_t p, float& ox, float& oy)
    {
        // The reference multiplies the int32_t `p` by literals: 0xa511e9b3 / 0xa399d265 do not fit an int, so those
        // products are unsigned; 0x63d83595 / 0x711ad6a5 DO fit, so those two are int x int and overflow -- undefined
        // behaviour that every compiler of the reference resolves as two's-complement wrap.  Spelled out here as the
        // unsigned product (same bits, no UB; found by running this file under UBSan).
        const uint32_t up = (uint32_t)p;
        int32_t sx = permute(s % n, n, up * 0xa511e9b3u);
        int32_t sy = permute(s / n, n, up * 0x63d83595u);
        float jx = randfloat(s, up * 0xa399d265u);
        float jy = randfloat(s, up * 0x711ad6a5u);
        ox = (s % n + (sy + jx) / n) / n;
        oy = (s / n + (sx + jy) / n) / n;
    }
    float nextSample()                                              // cmj.h:32-37,116-121
    {
        int32_t idx = permute(m_idx, CMJ_DIM * CMJ_DIM, 0xa399d265 * m_dimension * m_scramble);
        float x, y;
        cmj(idx, CMJ_DIM, m_dimension * m_scramble, x, y);
        m_dimension++;
        return x;
    }
    void nextSample2D(float& x, float& y)                           // cmj.h:39-44
    {
        int32_t idx = permute(m_idx, CMJ_DIM * CMJ_DIM, 0xa399d265 * m_dimension * m_scramble);
        cmj(idx, CMJ_DIM, m_dimension * m_scramble, x, y);
        m_dimension++;
    }
};

// ---------------------------------------------------------------------------------------
// Scene view over the flat arrays (what aten::context's accessors return,
// scene/host_scene_context.h:60-119,126,148,192,199,233,586,606)
// ---------------------------------------------------------------------------------------
struct Scene {
    const atn_scene_desc* d;
    explicit Scene(const atn_scene_desc* desc) : d(desc) {}

    const atn_object_param& GetObject(uint32_t i) const { return d->objects[i]; }
    const atn_material_param& GetMaterial(uint32_t i) const { return d->materials[i]; }
    const atn_triangle_param& GetTriangle(uint32_t i) const { return d->triangles[i]; }
    const atn_light_param& GetLight(uint32_t i) const { return d->lights[i]; }
    int32_t GetLightNum() const { return (int32_t)d->n_lights; }
    m4 GetMatrix(uint32_t i) const { m4 r; std::memcpy(&r, &d->matrices[i], sizeof(m4)); return r; }
    v3 GetPositionAsVec3(uint32_t i) const { const auto& p = d->vtx_pos[i]; return v3(p.x, p.y, p.z); }
    v4 GetPositionAsVec4(uint32_t i) const { const auto& p = d->vtx_pos[i]; return v4(p.x, p.y, p.z, p.w); }
    v4 GetNormalAsVec4(uint32_t i) const { const auto& p = d->vtx_nml[i]; return v4(p.x, p.y, p.z, p.w); }
    const atn_bvh_node* GetBvhNodes(uint32_t list) const { return d->bvh_lists[list].nodes; }
    const atn_scene_rendering_config& cfg() const { return d->config; }
    // context::GetNprTargetLight (scene/host_scene_context.cpp:70-74)
    const atn_light_param& GetNprTargetLight(uint32_t i) const { return d->npr_target_lights[i]; }
    // context::GetScreenSpaceTextureAt (host_scene_context.h:611-617) -> texture::AtByXY(x, y).x (image/texture.cpp:36-60)
    float GetScreenSpaceTextureAt(int32_t x, int32_t y) const
    {
        const atn_texture_desc& t = d->screen_space_texture;
        if (t.texels && t.width > 0 && t.height > 0) return t.texels[(uint32_t)(y * t.width + x)].x;
        return 1.0F;
    }
};

// ---------------------------------------------------------------------------------------
// Camera: camera/pinhole.cpp:34-75 (CreateCameraParam), :97-118 (sample),
//         camera/camera.h:182-197 (ComputePixelWidthAtDistance -- the reference's only KAT)
// ---------------------------------------------------------------------------------------
inline float Deg2Rad(float d) { return (PI * (d) / 180.0F); }   // math/math.h:18-21

inline atn_camera_param CreateCameraParam(const v3& origin, const v3& lookat, const v3& up,
    float vfov, float z_near, float z_far, int32_t width, int32_t height)
{
    atn_camera_param p;
    std::memset(&p, 0, sizeof(p));
    float theta = Deg2Rad(vfov);
    p.aspect = width / (float)height;
    float half_height = std::tan(theta / 2);
    float half_width = p.aspect * half_height;
    v3 dir = normalize(lookat - origin);
    v3 right = normalize(cross(dir, up));
    v3 cup = cross(right, dir);
    v3 center = origin + dir;
    v3 u = half_width * right;
    v3 v = half_height * cup;
    auto st = [](float* d, const v3& s) { d[0] = s.x; d[1] = s.y; d[2] = s.z; };
    st(p.origin, origin); st(p.lookat, lookat); st(p.dir, dir); st(p.right, right); st(p.up, cup);
    st(p.center, center); st(p.u, u); st(p.v, v);
    p.dist = height / (2.0F * std::tan(theta / 2));
    p.vfov = vfov; p.width = width; p.height = height;
    p.znear = std::min(z_near, z_far);
    p.zfar = std::max(z_near, z_far);
    return p;
}

inline float ComputePixelWidthAtDistance(const atn_camera_param& param, float distance_from_camera)
{
    distance_from_camera = std::fabs(distance_from_camera);
    float hfov = param.vfov * param.height / float(param.width);
    hfov = Deg2Rad(hfov);
    float half_width = std::tan(hfov / 2) * distance_from_camera;
    float width = half_width * 2;
    return width / float(param.width);
}

inline v3 ld3(const float* p) { return v3(p[0], p[1], p[2]); }

inline Ray PinholeSample(const atn_camera_param& cam, float s, float t)   // pinhole.cpp:97-118
{
    s = 2.0F * s - 1.0F;
    t = 2.0F * t - 1.0F;
    v3 pos_on_lens = s * ld3(cam.u) + t * ld3(cam.v);
    pos_on_lens = pos_on_lens + ld3(cam.center);
    Ray r;
    r.dir = normalize(pos_on_lens - ld3(cam.origin));
    r.org = ld3(cam.origin);
    return r;
}

// ---------------------------------------------------------------------------------------
// Intersection primitives: math/aabb.h:62-86, math/intersect.h:45-90, geometry/triangle.h:40-67
// ---------------------------------------------------------------------------------------
inline bool aabb_hit(const Ray& r, const v3& _min, const v3& _max, float t_min, float t_max)
{
    v3 invdir = 1.0F / (r.dir + v3(1e-6F));
    v3 oxinvdir = -r.org * invdir;
    const v3 f = _max * invdir + oxinvdir;
    const v3 n = _min * invdir + oxinvdir;
    const v3 tmax = vmax(f, n);
    const v3 tmin = vmin(f, n);
    const float t1 = fmin_(min_from_vec3(tmax), t_max);
    const float t0 = fmax_(max_from_vec3(tmin), t_min);
    return t0 <= t1;
}

struct TriHit { bool isIntersect; float a, b, t; };
inline TriHit intersectTriangle(const Ray& ray, const v3& v0, const v3& v1, const v3& v2)
{
    v3 e1 = v1 - v0;
    v3 e2 = v2 - v0;
    v3 r = ray.org - v0;
    v3 d = ray.dir;
    v3 u = cross(d, e2);
    v3 v = cross(r, e1);
    float inv = 1.0F / dot(u, e1);
    float t = dot(v, e2) * inv;
    float beta = dot(u, r) * inv;
    float gamma = dot(v, d) * inv;
    TriHit res;
    res.isIntersect = ((beta >= 0.0F && beta <= 1.0F) && (gamma >= 0.0F && gamma <= 1.0F)
        && (beta + gamma <= 1.0F) && t >= 0.0F);
    res.a = beta; res.b = gamma; res.t = t;
    return res;
}

struct Isect {      // aten::Intersection, scene/hit_parameter.h:28-64
    float t{ INF };
    int32_t objid{ -1 }, mtrlid{ -1 }, meshid{ -1 };
    int32_t tri_id{ -1 };
    float a{ 0 }, b{ 0 };
    int32_t isVoxel{ 0 };
};

inline bool triangle_hit(const atn_triangle_param& prm, const Scene& ctxt, const Ray& r, Isect* isect)
{
    bool isHit = false;
    const v3 v0 = ctxt.GetPositionAsVec3(prm.idx[0]);
    const v3 v1 = ctxt.GetPositionAsVec3(prm.idx[1]);
    const v3 v2 = ctxt.GetPositionAsVec3(prm.idx[2]);
    const auto res = intersectTriangle(r, v0, v1, v2);
    if (res.isIntersect) {
        if (res.t < isect->t) {
            isect->t = res.t; isect->a = res.a; isect->b = res.b;
            isHit = true;
        }
    }
    return isHit;
}

struct TraverseStats { uint64_t nodes{ 0 }, tris{ 0 }; };

// ThreadedBvhTraverser<true>::Traverse<Closest>, accelerator/threaded_bvh_traverser.h:98-304.
// lod_depth is always -1 on this path, so the voxel branch (:221-277) is dead; spheres at TLAS
// leaves fall through untested (:146-219).
inline bool TraverseClosest(Isect& isect, const Scene& ctxt, const Ray r, float t_min, float t_max,
    TraverseStats* stats = nullptr)
{
    t_min = ctxt.cfg().bvh_hit_min > 0 ? ctxt.cfg().bvh_hit_min : t_min;
    isect.isVoxel = false;

    const atn_bvh_node* node_list = ctxt.GetBvhNodes(0);
    Ray transformed_ray = r;
    int32_t nodeid = 0;
    int32_t objid = -1, meshid = -1;
    int32_t top_layer_hit = -1, top_layer_miss = -1;

    while (nodeid >= 0) {
        bool is_hit = false;
        atn_bvh_node node = node_list[nodeid];
        if (stats) stats->nodes++;

        if (node.f0 >= 0 || node.f1 >= 0) {            // ThreadedBvhNode::isLeaf, threaded_bvh.h:41-44
            if (node.f2 >= 0) {                         // node.ex_bvh.exid >= 0 : nested BVH
                const auto& obj = ctxt.GetObject(static_cast<uint32_t>(node.f0));
                if (obj.mtx_id >= 0) {
                    const m4 mtx_W2L = ctxt.GetMatrix(obj.mtx_id + 1);
                    // mat4::applyRay, math/mat4.h:223-235: ray(org,dir) ctor re-normalises dir.
                    transformed_ray = Ray(mtx_W2L.apply(r.org), mtx_W2L.applyXYZ(r.dir));
                }
                else {
                    transformed_ray = r;
                }
                uint32_t bits = (uint32_t)float_as_int(node.f2);
                int32_t exid = ATN_EXID_MAIN(bits);     // enable_lod is false on this path
                node_list = ctxt.GetBvhNodes(exid);
                objid = static_cast<int32_t>(node.f0);
                meshid = static_cast<int32_t>(node.f3);
                top_layer_hit = static_cast<int32_t>(node.hit);
                top_layer_miss = static_cast<int32_t>(node.miss);
                is_hit = true;
                node.hit = 0;
            }
            else if (node.f1 >= 0) {                    // triangle leaf
                const auto& prim = ctxt.GetTriangle(static_cast<uint32_t>(node.f1));
                Isect isect_tmp;
                isect_tmp.t = INF;
                if (stats) stats->tris++;
                is_hit = triangle_hit(prim, ctxt, transformed_ray, &isect_tmp);
                bool is_intersect = t_min < isect_tmp.t && (is_hit && isect_tmp.t < isect.t);
                if (is_intersect) {
                    isect = isect_tmp;
                    isect.objid = objid;
                    isect.tri_id = static_cast<int32_t>(node.f1);
                    isect.mtrlid = static_cast<int32_t>(prim.mtrlid);
                    isect.meshid = static_cast<int32_t>(prim.mesh_id);
                    isect.meshid = (isect.meshid < 0 ? meshid : isect.meshid);
                    t_max = isect.t;
                }
            }
        }
        else {
            is_hit = aabb_hit(transformed_ray, ld3(node.boxmin), ld3(node.boxmax), t_min, t_max);
        }

        nodeid = static_cast<int32_t>(is_hit ? node.hit : node.miss);

        if (nodeid < 0) {
            nodeid = is_hit ? top_layer_hit : top_layer_miss;
            top_layer_hit = -1;
            top_layer_miss = -1;
            node_list = ctxt.GetBvhNodes(0);
            transformed_ray = r;
        }
    }
    return (isect.objid >= 0);
}

// ---------------------------------------------------------------------------------------
// Hit evaluation: geometry/EvaluateHitResult.h:10-72, geometry/PolygonObject.h:37-72,
// geometry/triangle.h:69-120
// ---------------------------------------------------------------------------------------
struct HitRec {     // aten::hitrecord, scene/hit_parameter.h:7-26
    v3 p; float area{ 0 };
    v3 normal; int32_t mtrlid{ -1 };
    float u{ 0 }, v{ 0 };
    int32_t meshid{ -1 };
    bool isVoxel{ false };
};

inline void triangle_EvaluateHitResult(const Scene& ctxt, const atn_triangle_param& tri, HitRec* rec, const Isect* isect)
{
    const v4 p0 = ctxt.GetPositionAsVec4(tri.idx[0]);
    const v4 p1 = ctxt.GetPositionAsVec4(tri.idx[1]);
    const v4 p2 = ctxt.GetPositionAsVec4(tri.idx[2]);
    const v4 n0 = ctxt.GetNormalAsVec4(tri.idx[0]);
    const v4 n1 = ctxt.GetNormalAsVec4(tri.idx[1]);
    const v4 n2 = ctxt.GetNormalAsVec4(tri.idx[2]);
    const float u0 = p0.w, v0 = n0.w, u1 = p1.w, v1 = n1.w, u2 = p2.w, v2 = n2.w;

    float a = isect->a;
    float b = isect->b;
    float c = 1 - a - b;

    rec->p = (c * p0 + a * p1 + b * p2).xyz();
    rec->normal = (c * n0 + a * n1 + b * n2).xyz();
    rec->u = c * u0 + a * u1 + b * u2;
    rec->v = c * v0 + a * v1 + b * v2;

    if (tri.needNormal > 0) {
        v4 e01 = p1 - p0;
        v4 e02 = p2 - p0;
        e01.w = e02.w = 0.0F;
        rec->normal = normalize(cross(e01, e02)).xyz();
    }
    rec->area = tri.area;
}

inline void PolygonObject_evaluate_hit_result(const atn_object_param& obj, const Scene& ctxt,
    const m4& mtx_L2W, HitRec& rec, const Isect& isect)
{
    const auto& faceParam = ctxt.GetTriangle(isect.tri_id);
    triangle_EvaluateHitResult(ctxt, faceParam, &rec, &isect);

    v4 p0 = ctxt.GetPositionAsVec4(faceParam.idx[0]);
    v4 p1 = ctxt.GetPositionAsVec4(faceParam.idx[1]);
    p0.w = p1.w = 1.0F;

    float orignalLen = length(p1.xyz() - p0.xyz());
    float scaledLen = 0;
    {
        v4 _p0 = mtx_L2W.apply(p0);
        v4 _p1 = mtx_L2W.apply(p1);
        scaledLen = length(_p1.xyz() - _p0.xyz());
    }
    float ratio = scaledLen / orignalLen;
    ratio = ratio * ratio;
    rec.area = obj.area * ratio;
    rec.mtrlid = isect.mtrlid;
}

// isClose(A, B, maxUlps), math/math.h:340-372
inline bool isCloseUlps(float A, float B, int32_t maxUlps)
{
    int32_t aInt = float_as_int(A);
    if (aInt < 0) aInt = (int32_t)(0x80000000u - (uint32_t)aInt);
    int32_t bInt = float_as_int(B);
    if (bInt < 0) bInt = (int32_t)(0x80000000u - (uint32_t)bInt);
    // (the reference's `abs(aInt - bInt)` overflows a signed int for operands of opposite huge magnitude -- +-inf against a
    // large value of the other sign; its compilers wrap, and so does this, spelled out: found by tools/sanitize_cpu.sh)
    const int32_t d = (int32_t)((uint32_t)aInt - (uint32_t)bInt);
    const int32_t intDiff = d < 0 ? (int32_t)(0u - (uint32_t)d) : d;
    return intDiff <= maxUlps;
}

// sphere::hit, geometry/sphere.cpp:30-91.  Only reached through AreaLight::sample for sphere lights:
// the BVH traverser never tests spheres.
inline bool sphere_hit(const atn_object_param& prm, const Ray& r, float t_min, float t_max, Isect* isect)
{
    (void)t_min; (void)t_max;
    const v3 p_o = ld3(prm.sphere.center) - r.org;
    const float b = dot(p_o, r.dir);
    const float D4 = b * b - dot(p_o, p_o) + prm.sphere.radius * prm.sphere.radius;
    if (D4 < 0.0F) return false;
    const float sqrt_D4 = std::sqrt(D4);
    const float t1 = b - sqrt_D4;
    const float t2 = b + sqrt_D4;
    const bool close = isCloseUlps(std::fabs(b), sqrt_D4, 2500);
    if (t1 > EPS && !close) isect->t = t1;
    else if (t2 > EPS && !close) isect->t = t2;
    else return false;
    return true;
}

inline void evaluate_hit_result(HitRec& rec, const atn_object_param& obj, const Scene& ctxt, const Ray& r, const Isect& isect)
{
    const atn_object_param& real_obj = obj.type == ATN_OBJ_INSTANCE ? ctxt.GetObject(obj.object_id) : obj;
    const int32_t mtx_id = obj.type == ATN_OBJ_INSTANCE ? obj.mtx_id : -1;
    m4 mtx_L2W = m4::identity();
    if (mtx_id >= 0) mtx_L2W = ctxt.GetMatrix(mtx_id);

    if (real_obj.type == ATN_OBJ_POLYGONS) {
        PolygonObject_evaluate_hit_result(real_obj, ctxt, mtx_L2W, rec, isect);
    }
    else if (real_obj.type == ATN_OBJ_SPHERE) {
        // sphere::EvaluateHitResult (geometry/sphere.cpp:93-107) + getUV (:5-12)
        rec.p = r.org + isect.t * r.dir;
        rec.normal = (rec.p - ld3(real_obj.sphere.center)) / real_obj.sphere.radius;
        rec.mtrlid = isect.mtrlid;
        const float radius = real_obj.sphere.radius;
        rec.area = 4 * PI * radius * radius;
        const float phi = std::asin(rec.normal.y);
        const float theta = std::atan(rec.normal.x / rec.normal.z);
        rec.u = (theta + PI * 0.5F) / PI;
        rec.v = (phi + PI * 0.5F) / PI;
    }

    rec.p = mtx_L2W.apply(rec.p);
    rec.normal = normalize(mtx_L2W.applyXYZ(rec.normal));
    rec.isVoxel = false;
    rec.mtrlid = isect.mtrlid;
    rec.meshid = isect.meshid;
}

// ---------------------------------------------------------------------------------------
// Textures: image/texture.cpp:36-75, image/texture.h:197-208, material/sample_texture.h:42-87
// ---------------------------------------------------------------------------------------
inline int32_t NormalizeToWrapRepeat(int32_t value, int32_t wrap_size)
{
    if (wrap_size <= 0) return 0;       // 1-texel-wide texture: reference divides by zero; guarded here
    if (value > wrap_size) {
        auto n = value / wrap_size;
        value -= n * wrap_size;
    }
    else if (value < 0) {
        auto n = std::abs(value / wrap_size);
        value += (n + 1) * wrap_size;
    }
    return value;
}

inline v4 texture_at(const atn_texture_desc& tex, float u, float v)
{
    int32_t iu = static_cast<int32_t>(u * (tex.width - 1));
    int32_t iv = static_cast<int32_t>(v * (tex.height - 1));
    const auto x = NormalizeToWrapRepeat(iu, tex.width - 1);
    const auto y = NormalizeToWrapRepeat(iv, tex.height - 1);
    const atn_vec4& c = tex.texels[(uint32_t)(y * tex.width + x)];
    return v4(c.x, c.y, c.z, c.w);
}

// Optional samplers (off = the parity path of aten::PathTracing): see include/aten_amd.h, atn_set_sampling_options.
struct SamplingOptions {
    int ibl_importance = 0, tex_bilinear = 0;
    // ImageBasedLight::preCompute tables (light/ibl.cpp:10-118), rebuilt whenever the envmap pointer / size changes
    const atn_vec4* env = nullptr; int32_t w = 0, h = 0; float multiplyer = 0.0F;
    std::vector<float> cdfV; std::vector<std::vector<float>> cdfU;
};
inline SamplingOptions& sampling_options() { static SamplingOptions o; return o; }

// texture::AtWithBilinear, image/texture.cpp:77-125 (texel coordinates clamped into the image)
inline v4 texture_at_bilinear(const atn_texture_desc& tex, float u, float v)
{
    const int32_t width_ = tex.width, height_ = tex.height;
    const float fx = u * (width_ - 1);
    const float fy = v * (height_ - 1);
    float frac_x = fx - 0.5F - static_cast<int32_t>(fx);
    float frac_y = fy - 0.5F - static_cast<int32_t>(fy);
    auto x = static_cast<int32_t>(fx);
    auto y = static_cast<int32_t>(fy);
    auto nearest_x = x;
    if (frac_x >= 0.5F) { nearest_x = x + 1; }
    else { nearest_x = x - 1; frac_x = 1.0F - frac_x; }
    auto nearest_y = y;
    if (frac_y >= 0.5F) { nearest_y = y + 1; }
    else { nearest_y = y - 1; frac_y = 1.0F - frac_y; }
    nearest_x = std::min(std::max(nearest_x, 0), width_ - 1);
    nearest_y = std::min(std::max(nearest_y, 0), height_ - 1);
    x = std::min(std::max(x, 0), width_ - 1);
    y = std::min(std::max(y, 0), height_ - 1);
    auto at = [&](int32_t xx, int32_t yy) { const atn_vec4& c = tex.texels[(uint32_t)(yy * width_ + xx)]; return v4(c.x, c.y, c.z, c.w); };
    const v4 c00 = at(x, y), c10 = at(nearest_x, y), c01 = at(x, nearest_y), c11 = at(nearest_x, nearest_y);
    auto lerp = [](const v4& a, const v4& b, float f) { return a * (1.0F - f) + b * f; };
    const v4 c0 = lerp(c00, c10, frac_x);
    const v4 c1 = lerp(c01, c11, frac_x);
    return lerp(c0, c1, frac_y);
}

inline v4 sampleTexture(const Scene& ctxt, int32_t texid, float u, float v, const v4& defaultValue)
{
    v4 ret = defaultValue;
    if (texid >= 0 && (uint32_t)texid < ctxt.d->n_textures && ctxt.d->textures[texid].texels) {
        if (sampling_options().tex_bilinear) return texture_at_bilinear(ctxt.d->textures[texid], u, v);
        ret = texture_at(ctxt.d->textures[texid], u, v);
    }
    return ret;
}

inline void applyNormalMap(const Scene& ctxt, int32_t normalMapIdx, const v3& orgNml, v3& newNml, float u, float v)
{
    if (normalMapIdx >= 0) {
        v3 nml = sampleTexture(ctxt, normalMapIdx, u, v, v4(0.0F)).xyz();
        nml = 2.0F * nml - v3(1.0F);
        nml = normalize(nml);
        v3 n = normalize(orgNml);
        v3 t, b;
        GetTangentCoordinate(n, t, b);
        newNml = nml.z * n + nml.x * t + nml.y * b;
        newNml = normalize(newNml);
    }
    else {
        newNml = normalize(orgNml);
    }
}

} // namespace orc
