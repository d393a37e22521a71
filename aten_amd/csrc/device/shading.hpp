// Hit evaluation, textures, BSDFs (Lambert / specular / GGX / Disney), lights and NEE: the device
// side of aten's shared "impl" headers.  Each function cites the reference file:line
// (relative to /root/reference/src/libaten) whose arithmetic it reproduces operation by operation.
#pragma once
#include "cmj.hpp"
#include "scene_dev.hpp"
#include "traverse.hpp"

namespace atn {

struct HitRec { f3 p; float area; f3 normal; float u, v; };
struct MtrlSample { f3 dir; f3 bsdf; float pdf; };
// What k_shade works out ONCE per hit for everything that follows at the same (normal, wi, material): the BSDF sample, the NEE
// evaluation of the same BSDF, the light sample around the same normal.  Each value is the result of exactly the operations the
// consumers would run themselves on the same operands (GetTangentCoordinate of the shading normal, the roughness lookup, GGX's
// Smith lambda of the view direction): sharing them changes no bit, it takes ~90 + ~65 lane-instructions out of every NEE vertex.
// A null pointer = compute on the spot (the stage kernels behind atn_material_table, the toon path).
struct HitPre { f3 t, b; float rough, lambda_v; };

ATN_DEV m4 load_m4(const DevScene& sc, int32_t elem) // element index of a mat4
{
    m4 m;
    m.r0 = sc.matrices[4 * elem + 0]; m.r1 = sc.matrices[4 * elem + 1];
    m.r2 = sc.matrices[4 * elem + 2]; m.r3 = sc.matrices[4 * elem + 3];
    return m;
}

ATN_DEV int32_t triangle_mtrlid(const DevScene& sc, int32_t tri_id)
{
    return __float_as_int(sc.shade_tris[(size_t)kShadeTriQuads * (uint32_t)tri_id + 6].z);
}

// evaluate_hit_result (geometry/EvaluateHitResult.h:10-72) -> PolygonObject::evaluate_hit_result
// (geometry/PolygonObject.h:37-72) -> triangle::EvaluateHitResult (geometry/triangle.h:69-120)
ATN_DEV void evaluate_hit(HitRec& rec, const DevScene& sc, int32_t objid, int32_t tri_id, float a, float b)
{
    const atn_object_param* obj = &sc.objects[objid];
    const bool is_inst = obj->type == ATN_OBJ_INSTANCE;
    const atn_object_param* real_obj = is_inst ? &sc.objects[obj->object_id] : obj;
    const int32_t mtx_id = is_inst ? obj->mtx_id : -1;
    m4 L2W = m4_identity();
    if (mtx_id >= 0) L2W = load_m4(sc, mtx_id);

    // the triangle's packed record (scene_dev.hpp): the vertices' own float4s, copied
    const float4* st = sc.shade_tris + (size_t)kShadeTriQuads * (uint32_t)tri_id;
    const float4 p0 = st[0], p1 = st[1], p2 = st[2];
    const float4 n0 = st[3], n1 = st[4], n2 = st[5];
    const int32_t need_normal = __float_as_int(st[6].y);
    const float c = 1 - a - b;

    float4 P = add4(add4(mul4(c, p0), mul4(a, p1)), mul4(b, p2));
    float4 N = add4(add4(mul4(c, n0), mul4(a, n1)), mul4(b, n2));
    rec.p = mk3(P);
    rec.normal = mk3(N);
    rec.u = (c * p0.w + a * p1.w) + b * p2.w;
    rec.v = (c * n0.w + a * n1.w) + b * n2.w;
    if (need_normal > 0) {
        float4 e01 = sub4(p1, p0), e02 = sub4(p2, p0);
        e01.w = 0.0F; e02.w = 0.0F;
        rec.normal = mk3(normalize4(cross4(e01, e02)));
    }
    // PolygonObject: whole-object area scaled by the instance's edge-length ratio
    {
        const f3 q0 = mk3(p0), q1 = mk3(p1);
        const float orignalLen = length(q1 - q0);
        const f3 s0 = m4_apply_w1(L2W, q0), s1 = m4_apply_w1(L2W, q1);
        const float scaledLen = length(s1 - s0);
        float ratio = scaledLen / orignalLen;
        ratio = ratio * ratio;
        rec.area = real_obj->area * ratio;
    }
    rec.p = m4_apply(L2W, rec.p);
    rec.normal = normalize(m4_applyXYZ(L2W, rec.normal));
}

// texture::at (image/texture.cpp:64-75, image/texture.h:197-208): point sample, wrap-repeat
ATN_DEV int32_t wrap_repeat(int32_t value, int32_t wrap_size)
{
    if (wrap_size <= 0) return 0;
    if (value > wrap_size) { int32_t n = value / wrap_size; value -= n * wrap_size; }
    else if (value < 0) { int32_t n = abs(value / wrap_size); value += (n + 1) * wrap_size; }
    return value;
}
ATN_DEV float4 fetch_texel(const DevScene& sc, const DevTexture& t, int32_t x, int32_t y)
{
    const uint32_t idx = t.offset + (uint32_t)(y * t.width + x);
    if (t.format) {
        // the same IEEE operation the caller's 8-bit -> float conversion made (which one: checked per texel at upload)
        const uint32_t p = sc.texels8[idx];
        const float r = (float)(p & 255u), g = (float)((p >> 8) & 255u), b = (float)((p >> 16) & 255u), a = (float)(p >> 24);
        if (t.format == 1) return make_float4(r / 255.0F, g / 255.0F, b / 255.0F, a / 255.0F);
        const float norm = 1.0F / 255;
        return make_float4(r * norm, g * norm, b * norm, a * norm);
    }
    return sc.texels[idx];
}
ATN_DEV float4 lerp4(const float4& a, const float4& b, float f)      // aten::lerp, math/math.h:190-194
{
    return add4(mul4(1.0F - f, a), mul4(f, b));
}
ATN_DEV float4 sample_texture(const DevScene& sc, int32_t texid, float u, float v, const float4& def)
{
    if (texid < 0 || texid >= sc.n_textures) return def;
    const DevTexture t = sc.textures[texid];
    if (sc.tex_bilinear) {
        // texture::AtWithBilinear, image/texture.cpp:77-125, as the reference writes it (its "nearest" neighbour and
        // weight rule included); texel coordinates are clamped into the image, which the reference leaves to the caller
        const float fx = u * (float)(t.width - 1);
        const float fy = v * (float)(t.height - 1);
        float frac_x = (fx - 0.5F) - (float)(int32_t)fx;
        float frac_y = (fy - 0.5F) - (float)(int32_t)fy;
        int32_t x = (int32_t)fx, y = (int32_t)fy;
        int32_t nx, ny;
        if (frac_x >= 0.5F) nx = x + 1; else { nx = x - 1; frac_x = 1.0F - frac_x; }
        if (frac_y >= 0.5F) ny = y + 1; else { ny = y - 1; frac_y = 1.0F - frac_y; }
        nx = nx < 0 ? 0 : (nx > t.width - 1 ? t.width - 1 : nx);
        ny = ny < 0 ? 0 : (ny > t.height - 1 ? t.height - 1 : ny);
        x = x < 0 ? 0 : (x > t.width - 1 ? t.width - 1 : x);
        y = y < 0 ? 0 : (y > t.height - 1 ? t.height - 1 : y);
        const float4 c00 = fetch_texel(sc, t, x, y), c10 = fetch_texel(sc, t, nx, y);
        const float4 c01 = fetch_texel(sc, t, x, ny), c11 = fetch_texel(sc, t, nx, ny);
        return lerp4(lerp4(c00, c10, frac_x), lerp4(c01, c11, frac_x), frac_y);
    }
    const int32_t iu = (int32_t)(u * (float)(t.width - 1));
    const int32_t iv = (int32_t)(v * (float)(t.height - 1));
    const int32_t x = wrap_repeat(iu, t.width - 1);
    const int32_t y = wrap_repeat(iv, t.height - 1);
    return fetch_texel(sc, t, x, y);
}

// applyNormalMap, material/sample_texture.h:62-87
ATN_DEV f3 apply_normal_map(const DevScene& sc, int32_t normalMap, const f3& orgNml, float u, float v)
{
    if (normalMap >= 0) {
        f3 nml = mk3(sample_texture(sc, normalMap, u, v, make_float4(0, 0, 0, 0)));
        nml = 2.0F * nml - mk3(1.0F);
        nml = normalize(nml);
        const f3 n = normalize(orgNml);
        f3 t, b;
        tangent_coordinate(n, t, b);
        f3 r = (nml.z * n + nml.x * t) + nml.y * b;
        return normalize(r);
    }
    return normalize(orgNml);
}

// ------------------------------------------------------------------ material helpers
ATN_DEV float schlick_f0_cos(float f0, float costheta)         // material/material.h:500-509
{
    const float c = sclamp(1 - costheta, 0.0F, 1.0F);
    const float c5 = (((c * c) * c) * c) * c;
    return f0 + (1.0F - f0) * c5;
}
ATN_DEV float schlick_fresnel(float ni, float nt, const f3& w, const f3& n)    // material.h:466-498
{
    float costheta = dot(w, n);
    if (costheta < 0) { float t = ni; ni = nt; nt = t; costheta = -costheta; }
    float f0 = (ni - nt) / (ni + nt);
    f0 = f0 * f0;
    return schlick_f0_cos(f0, costheta);
}
ATN_DEV f3 reflect_vector(const f3& wi, const f3& n)            // material.h:517-524
{
    const f3 wo = wi - (2 * dot(wi, n)) * n;
    return normalize(wo);
}

// Diffuse, material/diffuse.h:86-137
ATN_DEV float diffuse_pdf(const f3& n, const f3& wo) { return fabsf(dot(n, wo)) / kPi; }
ATN_DEV f3 diffuse_dir(const f3& n, float r1, float r2, const HitPre* pre = nullptr)
{
    const float costheta = sqrtf(1 - r1);
    const float sintheta = sqrtf(r1);
    const float phi = kPi2 * r2;
    const float cosphi = cosf(phi);
    const float sinphi = sinf(phi);
    f3 t, b;
    if (pre) { t = pre->t; b = pre->b; } else tangent_coordinate(n, t, b);
    const f3 dir = ((t * sintheta) * cosphi + (b * sintheta) * sinphi) + n * costheta;
    return normalize(dir);
}
ATN_DEV f3 diffuse_brdf() { return mk3(1.0F) / kPi; }

// GGX, material/ggx.cpp:107-274
ATN_DEV float ggx_D(const f3& m, const f3& n, float roughness)
{
    const float a2 = roughness * roughness;
    const float costheta = fabsf(dot(m, n));
    const float cos2 = costheta * costheta;
    const float denom = (a2 - 1) * cos2 + 1.0f;
    const float denom2 = denom * denom;
    return denom > 0 ? a2 / (kPi * denom2) : 0.0F;
}
ATN_DEV float ggx_lambda(float roughness, const f3& w, const f3& n)
{
    const float cos_theta = fabsf(dot(w, n));
    const float cos2 = cos_theta * cos_theta;
    const float sin2 = 1.0f - cos2;
    const float tan2 = sin2 / cos2;
    const float a2 = 1.0f / ((roughness * roughness) * tan2);
    return (-1.0f + sqrtf(1.0f + 1.0f / a2)) / 2.0f;
}
ATN_DEV float ggx_G2(float roughness, const f3& view, const f3& light, const f3& n, const HitPre* pre = nullptr)
{
    const float lwi = pre ? pre->lambda_v : ggx_lambda(roughness, view, n);
    const float lwo = ggx_lambda(roughness, light, n);
    return 1.0f / ((1.0f + lwi) + lwo);
}
ATN_DEV float ggx_pdf_h(float roughness, const f3& n, const f3& m, const f3& wo)
{
    const float D = ggx_D(m, n, roughness);
    const float costheta = fabsf(dot(m, n));
    const float denom = 4 * fabsf(dot(wo, m));
    return denom > 0 ? (D * costheta) / denom : 0.0F;
}
ATN_DEV float ggx_pdf(float roughness, const f3& n, const f3& wi, const f3& wo)
{
    const f3 wh = normalize((-wi) + wo);
    return ggx_pdf_h(roughness, n, wh, wo);
}
ATN_DEV f3 ggx_sample_m(float roughness, const f3& n, float r1, float r2, const HitPre* pre = nullptr)
{
    float theta = atanf(roughness * sqrtf(r1 / (1 - r1)));
    theta = ((theta >= 0) ? theta : (theta + 2 * kPi));
    const float phi = (2 * kPi) * r2;
    const float costheta = cosf(theta);
    const float sintheta = sinf(theta);
    const float cosphi = cosf(phi);
    const float sinphi = sinf(phi);
    f3 t, b;
    if (pre) { t = pre->t; b = pre->b; } else tangent_coordinate(n, t, b);
    const f3 m = ((t * sintheta) * cosphi + (b * sintheta) * sinphi) + n * costheta;
    return normalize(m);
}
ATN_DEV f3 ggx_dir(float r1, float r2, float roughness, const f3& wi, const f3& n, const HitPre* pre = nullptr)
{
    return reflect_vector(wi, ggx_sample_m(roughness, n, r1, r2, pre));
}
// (pre: HitPre::lambda_v is the Smith lambda of V at this roughness and N -- the GGX material only)
ATN_DEV f3 ggx_brdf_h(float roughness, float ior, const f3& N, const f3& V, const f3& L, const f3& H, const HitPre* pre = nullptr)     // ComputeBRDFWithHalfVector
{
    const float NL = fabsf(dot(N, L));
    const float NV = fabsf(dot(N, V));
    const float D = ggx_D(H, N, roughness);
    const float G = ggx_G2(roughness, V, L, N, pre);
    const float F = schlick_fresnel(1.0F, ior, L, H);
    const float denom = (4 * NL) * NV;
    const float bsdf = denom > kEps ? ((F * G) * D) / denom : 0.0f;
    return mk3(bsdf);
}
ATN_DEV f3 ggx_brdf(float roughness, float ior, const f3& N, const f3& wi, const f3& wo, const HitPre* pre = nullptr)
{
    const f3 V = -wi, L = wo;
    return ggx_brdf_h(roughness, ior, N, V, L, normalize(L + V), pre);
}

// ToonSpecular (material/toon.cpp:288-367): GGX with the "stylized highlight" half vector; a material type that only
// Toon::ComputeBRDF creates.  Material set 3.
ATN_DEV float sign_of(float f) { return f == 0.0F ? 0.0F : (f > 0.0F ? 1.0F : -1.0F); }      // aten::sign, math/math.h:62-73
ATN_DEV f3 toon_specular_half(const atn_toon_param& tp, const f3& N, const f3& V, const f3& L)
{
    f3 H = normalize(L + V);
    f3 t, b;
    tangent_coordinate(N, t, b);
    H = (H + tp.highlight.translation_dt * t) + tp.highlight.translation_db * b;
    H = normalize(H);
    H = (H - (tp.highlight.scale_t * dot(H, t)) * t) - (tp.highlight.scale_b * dot(H, b)) * b;
    H = normalize(H);
    H = (H - (tp.highlight.split_t * sign_of(dot(H, t))) * t) - (tp.highlight.split_b * sign_of(dot(H, b))) * b;
    H = normalize(H);
    const float sqrnorm_t = sinf(powf(acosf(dot(H, t)), tp.highlight.square_sharp));
    const float sqrnorm_b = sinf(powf(acosf(dot(H, b)), tp.highlight.square_sharp));
    H = H - tp.highlight.square_magnitude * (((sqrnorm_t * dot(H, t)) * t) + ((sqrnorm_b * dot(H, b)) * b));
    H = normalize(H);
    return H;
}
ATN_DEV float ggx_roughness(const DevScene& sc, const DevMaterial& m, float u, float v)
{
    return sample_texture(sc, m.roughnessMap, u, v, make_float4(m.roughness, m.roughness, m.roughness, m.roughness)).x;
}

// Disney, material/disney_brdf.cpp:38-555
ATN_DEV f3 dis_ctint(const f3& base)
{
    const float Y = dot(base, mk3(0.3F, 0.6F, 0.1F));
    return Y > 0 ? base / Y : mk3(1.0F);
}
ATN_DEV float dis_schlick(float u)
{
    const float m = sclamp(1.0F - u, 0.0F, 1.0F);
    const float m2 = m * m;
    return (m2 * m2) * m;
}
ATN_DEV f3 dis_diffuse_brdf(const f3& base, float roughness, float subsurface, const f3& V, const f3& L, const f3& N)
{
    const f3 H = normalize(V + L);
    const float LdotH = dot(L, H);
    const float NdotV = dot(V, N);
    const float NdotL = dot(L, N);
    const float FV = dis_schlick(dot(V, N));
    const float FL = dis_schlick(dot(L, N));
    const float Fd90 = 0.5F + ((2 * LdotH) * LdotH) * roughness;
    float fd = mixf(1.0F, Fd90, FL) * mixf(1.0F, Fd90, FV);
    const float Fss90 = (LdotH * LdotH) * roughness;
    const float Fss = mixf(1.0F, Fss90, FL) * mixf(1.0F, Fss90, FV);
    const float ss = 1.25F * (Fss * (1.0F / (NdotL + NdotV) - 0.5F) + 0.5F);
    fd = mixf(fd, ss, subsurface);
    return (base / kPi) * fd;
}
ATN_DEV f3 dis_sheen_brdf(const f3& base, float sheen, float sheen_tint, const f3& V, const f3& L)
{
    const f3 H = normalize(V + L);
    const f3 Csheen = mix3(mk3(1.0F), dis_ctint(base), sheen_tint);
    const float FH = dis_schlick(dot(L, H));
    return (sheen * Csheen) * FH;
}
ATN_DEV float dis_gtr1(float a, float NdotH)
{
    if (a >= 1) return 1 / kPi;
    const float a2 = a * a;
    const float t = 1.0F + ((a2 - 1.0F) * NdotH) * NdotH;
    return (a2 - 1) / ((kPi * logf(a2)) * t);
}
ATN_DEV f3 dis_clearcoat_brdf(float clearcoat, const f3& V, const f3& L)
{
    const f3 H = normalize(V + L);
    const float FH = dis_schlick(fabsf(dot(L, H)));
    const float F = mixf(0.04F, 1.0F, FH);
    return mk3((0.25F * clearcoat) * F);
}
ATN_DEV float dis_clearcoat_pdf(float gloss_or_rough, const f3& V, const f3& L, const f3& N)
{
    const f3 H = normalize(V + L);
    const float NdotH = dot(N, H);
    const float a = mixf(0.1F, 0.001F, gloss_or_rough);
    const float D = dis_gtr1(a, NdotH);
    const float costheta = fabsf(dot(H, N));
    const float denom = 4 * fabsf(dot(L, H));
    return denom > 0 ? (D * costheta) / denom : 0.0F;
}
ATN_DEV f3 dis_specular_brdf(const f3& base, float roughness, float metallic, float specular, float specular_tint,
                             const f3& V, const f3& L, const f3& N)
{
    const f3 H = normalize(V + L);
    const f3 Cspec = mix3(mk3(1.0F), dis_ctint(base), specular_tint);
    const f3 F_s0 = mix3((0.08F * specular) * Cspec, base, metallic);
    const f3 F = mix3(F_s0, mk3(1.0F), dot(L, H));
    const float D = ggx_D(H, N, roughness);
    const float G = ggx_G2(roughness, V, L, N);
    const float NdotL = fabsf(dot(N, L));
    const float NdotV = fabsf(dot(N, V));
    const float denom = (4 * NdotV) * NdotL;
    return denom > 0 ? ((F * D) * G) / denom : mk3(0.0F);
}
ATN_DEV float dis_specular_pdf(float roughness, const f3& V, const f3& L, const f3& N)
{
    const f3 H = normalize(V + L);
    return ggx_pdf_h(roughness, N, H, L);
}
struct DisW { float d, sh, sp, cc; };     // lobe weights: diffuse, sheen, specular, clearcoat (disney_brdf.h:105-111)

ATN_DEV DisW dis_weights(const f3& base, float metalic, float sheen, float specular, float clearcoat)   // :311-335
{
    const float lum = luminance(base.x, base.y, base.z);
    DisW w;
    w.d = lum * (1 - metalic);
    w.sh = sheen * (1 - metalic);
    w.sp = mixf(specular, 1.0F, metalic);
    w.cc = 0.25F * clearcoat;
    float norm = 0.0F;
    norm += w.d; norm += w.sh; norm += w.sp; norm += w.cc;
    if (norm > 0) { w.d /= norm; w.sh /= norm; w.sp /= norm; w.cc /= norm; }
    return w;
}
// evaluate every lobe whose weight is > 0 at (V, wo) and add weight * pdf (disney_brdf.cpp:404-433,527-550)
ATN_DEV void dis_eval_rest(const DevMaterial& m, const f3& base, const DisW& w, const f3& V, const f3& wo, const f3& N,
                           f3& d, f3& sh, f3& sp, f3& cc, float& p)
{
    if (w.d > 0.0F) {
        d = dis_diffuse_brdf(base, m.roughness, m.subsurface, V, wo, N);
        p += w.d * diffuse_pdf(N, wo);
    }
    if (w.sh > 0.0F) {
        sh = dis_sheen_brdf(base, m.sheen, m.sheenTint, V, wo);
        p += w.sh * (1 / kPi);
    }
    if (w.sp > 0.0F) {
        sp = dis_specular_brdf(base, m.roughness, m.metallic, m.specular, m.specularTint, V, wo, N);
        p += w.sp * dis_specular_pdf(m.roughness, V, wo, N);
    }
    if (w.cc > 0.0F) {
        cc = dis_clearcoat_brdf(m.clearcoat, V, wo);
        p += w.cc * dis_clearcoat_pdf(m.roughness, V, wo, N);   // reference quirk: roughness, not gloss (:431,517,548)
    }
}
ATN_DEV float disney_pdf(const DevMaterial& m, const f3& n, const f3& wi, const f3& wo)   // :347-378
{
    const f3 base = mk3(m.baseColor);
    const DisW w = dis_weights(base, m.metallic, m.sheen, m.specular, m.clearcoat);
    const f3 V = -wi;
    float p = 0.0F;
    p += w.d * diffuse_pdf(n, wo);
    p += w.sh * (1 / kPi);
    p += w.sp * dis_specular_pdf(m.roughness, V, wo, n);
    p += w.cc * dis_clearcoat_pdf(m.clearcoatGloss, V, wo, n);
    return p;
}
ATN_DEV MtrlSample disney_bsdf(const DevMaterial& m, const f3& n, const f3& wi, const f3& wo)   // :380-441
{
    const f3 base = mk3(m.baseColor);
    const DisW w = dis_weights(base, m.metallic, m.sheen, m.specular, m.clearcoat);
    const f3 V = -wi;
    f3 d = mk3(0.0F), sh = mk3(0.0F), sp = mk3(0.0F), cc = mk3(0.0F);
    float p = 0.0F;
    dis_eval_rest(m, base, w, V, wo, n, d, sh, sp, cc, p);
    MtrlSample r;
    r.bsdf = ((1 - m.metallic) * (d + sh) + sp) + cc;
    r.pdf = p;
    r.dir = wo;
    return r;
}
ATN_DEV void disney_sample(MtrlSample& res, const DevMaterial& m, const f3& n, const f3& wi, Cmj& smp, const HitPre* pre = nullptr)  // :443-555
{
    HitPre frame;       // (the tangent frame only: Disney's lobes have their own roughnesses)
    if (pre) { frame.t = pre->t; frame.b = pre->b; frame.rough = 0.0F; frame.lambda_v = 0.0F; }
    const HitPre* fr = pre ? &frame : nullptr;
    const float r1 = cmj_next(smp), r2 = cmj_next(smp), r3 = cmj_next(smp);
    const f3 base = mk3(m.baseColor);
    DisW w = dis_weights(base, m.metallic, m.sheen, m.specular, m.clearcoat);
    const float c0 = w.d, c1 = c0 + w.sh, c2 = c1 + w.sp;      // GetCDF, :337-345
    const f3 V = -wi, N = n;
    f3 wo; float p = 0;
    f3 d = mk3(0.0F), sh = mk3(0.0F), sp = mk3(0.0F), cc = mk3(0.0F);
    if (r3 < c0) {
        wo = diffuse_dir(N, r1, r2, fr);
        d = dis_diffuse_brdf(base, m.roughness, m.subsurface, V, wo, N);
        p = diffuse_pdf(N, wo);
        p *= w.d; w.d = 0.0F;
    }
    else if (r3 < c1) {
        wo = diffuse_dir(N, r1, r2, fr);
        sh = dis_sheen_brdf(base, m.sheen, m.sheenTint, V, wo);
        p = 1 / kPi;
        p *= w.sh; w.sh = 0.0F;
    }
    else if (r3 < c2) {
        wo = ggx_dir(r1, r2, m.roughness, -V, N, fr);
        sp = dis_specular_brdf(base, m.roughness, m.metallic, m.specular, m.specularTint, V, wo, N);
        p = dis_specular_pdf(m.roughness, V, wo, N);
        p *= w.sp; w.sp = 0.0F;
    }
    else {
        const float a = mixf(0.1F, 0.001F, m.clearcoatGloss);
        wo = reflect_vector(-V, ggx_sample_m(a, N, r1, r2, fr));
        cc = dis_clearcoat_brdf(m.clearcoat, V, wo);
        p = dis_clearcoat_pdf(m.roughness, V, wo, N);
        p *= w.cc; w.cc = 0.0F;
    }
    dis_eval_rest(m, base, w, V, wo, N, d, sh, sp, cc, p);
    res.pdf = p;
    res.bsdf = ((1 - m.metallic) * (d + sh) + sp) + cc;
    res.dir = wo;
}

// Refraction, material/refraction.cpp:62-161 + material::ComputeRefractVector (material.h:549-576)
ATN_DEV f3 refract_vector(float ni, float nt, const f3& wi, const f3& n)
{
    const f3 w = -wi;
    f3 N = n;
    float costheta = dot(w, n);
    if (costheta < 0.0F) { const float t = ni; ni = nt; nt = t; costheta = -costheta; N = -N; }
    const float sintheta_2 = 1.0F - costheta * costheta;
    const float ni_nt = ni / nt;
    const float ni_nt_2 = ni_nt * ni_nt;
    const f3 wo = (ni_nt * costheta - sqrtf(1.0F - ni_nt_2 * sintheta_2)) * N - ni_nt * w;
    return normalize(wo);
}
ATN_DEV f3 refraction_brdf(float ni, float nt, const f3& wo, const f3& n, float transmittance)
{
    const float c = fabsf(dot(wo, n));
    const float nt_ni = nt / ni;
    return mk3(c == 0.0F ? 0.0F : ((nt_ni * nt_ni) * transmittance) / c);
}
ATN_DEV void refraction_sample(MtrlSample& r, const DevMaterial& m, const f3& n, const f3& wi, Cmj& smp)
{
    float ni = 1.0F, nt = m.ior;
    const f3 V = -wi;
    f3 N = n;
    if (!(dot(V, N) >= 0.0F)) { N = -n; const float t = ni; ni = nt; nt = t; }
    const float ni_nt = ni / nt;
    const float cos_i = dot(V, N);
    const float cos_t_2 = 1.0F - ((ni_nt * ni_nt) * (1.0F - cos_i * cos_i));
    if (cos_t_2 < 0.0F) { const float t = ni; ni = nt; nt = t; N = -N; }
    f3 wo = refract_vector(ni, nt, wi, N);
    const float R = schlick_fresnel(ni, nt, wo, N);
    const float T = 1 - R;
    if (m.attrib & kAttrIdealRefraction) {
        r.pdf = 1.0F; r.dir = wo; r.bsdf = refraction_brdf(ni, nt, wo, N, T);
        return;
    }
    const float prob = 0.25F + 0.5F * R;
    const float u = cmj_next(smp);
    if (u < prob) {
        wo = reflect_vector(wi, N);
        const float c = fabsf(dot(wo, N));
        r.pdf = prob; r.dir = wo; r.bsdf = mk3(c == 0.0F ? 0.0F : R / c);
    }
    else {
        r.pdf = 1.0F - prob; r.dir = wo; r.bsdf = refraction_brdf(ni, nt, wo, N, T);
    }
}

// MicrofacetBeckman, material/beckman.cpp:103-255
ATN_DEV float beckman_D(const f3& m, const f3& n, float roughness)
{
    const float costheta = fabsf(dot(m, n));
    if (costheta <= 0) return 0;
    const float cos2 = costheta * costheta;
    const float cos4 = cos2 * cos2;
    const float sintheta = sqrtf(1 - cos2);
    const float tantheta = sintheta / costheta;
    const float tan2 = tantheta * tantheta;
    const float a2 = roughness * roughness;
    float D = 1.0f / ((kPi * a2) * cos4);
    D *= expf(-tan2 / a2);
    return D;
}
ATN_DEV float beckman_pdf(float roughness, const f3& n, const f3& wi, const f3& wo)
{
    const f3 wh = normalize(-wi + wo);
    const float costheta = fabsf(dot(wh, n));
    const float D = beckman_D(wh, n, roughness);
    const float denom = 4 * fabsf(dot(wo, wh));
    return denom > 0 ? (D * costheta) / denom : 0;
}
ATN_DEV f3 beckman_sample_m(float roughness, const f3& n, float r1, float r2)
{
    const float a2 = roughness * roughness;
    const float theta = atanf(sqrtf(-a2 * logf(1.0F - r1 * 0.99F)));
    const float phi = kPi2 * r2;
    const float costheta = cosf(theta), sintheta = sinf(theta);
    const float cosphi = cosf(phi), sinphi = sinf(phi);
    f3 t, b;
    tangent_coordinate(n, t, b);
    const f3 m = ((t * sintheta) * cosphi + (b * sintheta) * sinphi) + n * costheta;
    return normalize(m);
}
ATN_DEV float beckman_G1(float roughness, const f3& v, const f3& n)
{
    const float costheta = sclamp(fabsf(dot(v, n)), 0.0F, 1.0F);
    const float sintheta = sqrtf(1.0F - costheta * costheta);
    const float tantheta = sintheta / costheta;
    const float a = 1.0F / (roughness * tantheta);
    const float a2 = a * a;
    if (a < 1.6F) return (3.535F * a + 2.181F * a2) / ((1.0F + 2.276F * a) + 2.577F * a2);
    return 1.0F;
}
ATN_DEV f3 beckman_brdf(float roughness, float ior, const f3& N, const f3& wi, const f3& wo)
{
    const f3 V = -wi, L = wo;
    const f3 H = normalize(L + V);
    const float NL = fabsf(dot(N, L));
    const float NV = fabsf(dot(N, V));
    const float D = beckman_D(H, N, roughness);
    const float G2 = beckman_G1(roughness, V, N) * beckman_G1(roughness, L, N);
    const float F = schlick_fresnel(1.0F, ior, L, H);
    const float denom = (4 * NL) * NV;
    return mk3(denom > kEps ? ((F * G2) * D) / denom : 0.0f);
}

// OrenNayar, material/oren_nayar.cpp:8-140
ATN_DEV float oren_nayar_pdf(const f3& normal, const f3& wo)
{
    const float NL = dot(normal, wo);
    return NL > 0 ? NL / kPi : 0.0F;
}
ATN_DEV f3 oren_nayar_brdf(float roughness, const f3& normal, const f3& wi, const f3& wo)
{
    const float NL = dot(normal, wo);
    const float NV = dot(normal, -wi);
    const float a2 = roughness * roughness;
    const float A = 1.0F - 0.5F * (a2 / (a2 + 0.33F));
    const float B = 0.45F * (a2 / (a2 + 0.09F));
    const float LV = dot(wo, -wi);
    const float s = LV - NL * NV;
    const float t = s <= 0 ? 1.0F : s / smax(NL, NV);
    return mk3((1.0F / kPi) * (A + B * smax(0.0F, s / t)));
}

// MicrofacetVelvet, material/velvet.cpp:57-212
ATN_DEV float velvet_param(int idx, float f)
{
    const float p0[5] = { 25.3245F, 3.32435F, 0.16801F, -1.27393F, -4.85967F };
    const float p1[5] = { 21.5473F, 3.82987F, 0.19823F, -1.97760F, -4.32054F };
    return (f * p0[idx] + (1 - f)) + p1[idx];          // "+ p1" as the reference writes it (velvet.cpp:82)
}
ATN_DEV float velvet_L(float x, float roughness)
{
    const float f = powf(1.0F - roughness, 2.0F);
    const float a = velvet_param(0, f), b = velvet_param(1, f), c = velvet_param(2, f), d = velvet_param(3, f), e = velvet_param(4, f);
    return (a / (1 + b * powf(x, c)) + d * x) + e;
}
ATN_DEV float velvet_lambda(float roughness, const f3& w, const f3& m)
{
    const float cos_theta = sclamp(fabsf(dot(w, m)), 0.0F, 1.0F);
    if (cos_theta < 0.5F) return expf(velvet_L(cos_theta, roughness));
    return expf(2.0F * velvet_L(0.5F, roughness) - velvet_L(1 - cos_theta, roughness));
}
ATN_DEV f3 velvet_brdf(float roughness, const f3& N, const f3& wi, const f3& wo)
{
    const f3 V = -wi, L = wo;
    const f3 H = normalize(L + V);
    const float NL = fabsf(dot(N, L)), NV = fabsf(dot(N, V));
    // ComputeDistribution
    const float cos_theta = fabsf(dot(H, N));
    const float inv_r = 1.0F / roughness;
    const float sin_theta = sqrtf(sclamp(1 - cos_theta * cos_theta, 0.0F, 1.0F));
    const float D = ((2.0F + inv_r) * powf(sin_theta, inv_r)) / kPi2;
    // ComputeShadowingMaskingFunction
    float lambda_wi = velvet_lambda(roughness, V, N);
    const float lambda_wo = velvet_lambda(roughness, L, N);
    const float cos_theta_wi = sclamp(fabsf(dot(V, N)), 0.0F, 1.0F);
    lambda_wi = powf(lambda_wi, 1.0F + 2.0F * powf(1.0F - cos_theta_wi, 8.0F));
    const float G = 1.0F / ((1.0F + lambda_wi) + lambda_wo);
    const float denom = (4 * NL) * NV;
    return mk3(denom > kEps ? ((1.0F * G) * D) / denom : 0.0F);
}

// MicrofacetRefraction, material/microfacet_refraction.cpp:69-171
ATN_DEV void microfacet_refraction_sample(MtrlSample& r, const DevScene& sc, const DevMaterial& mt, const f3& n, const f3& wi,
                                          Cmj& smp, float tu, float tv)
{
    const float roughness = ggx_roughness(sc, mt, tu, tv);
    const float ior = mt.ior;
    float ni = 1.0F, nt = ior;
    const f3 V = -wi;
    f3 N = n;
    if (!(dot(V, N) >= 0.0F)) { N = -n; const float t = ni; ni = nt; nt = t; }
    const float r1 = cmj_next(smp), r2 = cmj_next(smp);
    const f3 m = ggx_sample_m(roughness, N, r1, r2);
    const float R = schlick_fresnel(ni, nt, wi, m);
    const float T = 1 - R;
    const float prob = R;
    const float u = cmj_next(smp);
    if (u < prob) {
        const f3 wo = reflect_vector(wi, m);
        const float VN = dot(V, N), LN = dot(wo, N);
        if (VN * LN < 0) {
            r.dir = reflect_vector(wi, N); r.pdf = 1.0f; r.bsdf = mk3(0.0F);
            return;
        }
        r.dir = wo;
        r.pdf = ggx_pdf_h(roughness, N, m, wo);
        r.pdf *= prob;
        r.bsdf = ggx_brdf(roughness, ior, N, wi, wo);
    }
    else {
        const f3 wo = refract_vector(ni, nt, wi, m);
        const float D = ggx_D(m, N, roughness);
        const float G = ggx_G2(roughness, V, wo, N);
        const float LH = fabsf(dot(wo, m));
        const float VH = fabsf(dot(V, m));
        const float denom = ni * dot(V, m) + nt * dot(wo, m);
        const float denom2 = denom * denom;
        const float costheta = fabsf(dot(m, n));
        const float nt2 = nt * nt;
        r.pdf = denom2 > 0 ? (D * costheta) * ((nt2 * LH) / denom2) : 1.0F;
        r.pdf *= 1.0F - prob;
        r.dir = wo;
        float VN = dot(V, N), LN = dot(wo, N);
        if (VN * LN > 0) {
            r.dir = refract_vector(ni, nt, wi, N); r.pdf = 1.0f; r.bsdf = mk3(0.0F);
            return;
        }
        VN = fabsf(VN); LN = fabsf(LN);
        r.bsdf = mk3(denom2 > 0 ? ((VH * LH) / (VN * LN)) * ((((nt2 * T) * D) * G) / denom2) : 0.0F);
    }
}

// Retroreflective, material/retroreflective.cpp:17-612 (prismatic-sheet model: Beckman surface reflection + retroreflection
// lobe around -wi + diffuse, mixed by Fresnel F and the effective retroreflective area E of the refracted direction).
// kEraTable: {incident angle inside the sheet in degrees, effective retroreflective area} -- measured data the model
// interpolates (retroreflective.cpp:59-161), angle literals as the reference spells them.
struct EraEntry { float deg, area; };
__device__ const EraEntry kEraTable[101] = {
    { 0.00000F, 0.64754F }, { 0.90000F, 0.65542F }, { 1.80000F, 0.65597F }, { 2.70000F, 0.65809F },
    { 3.60000F, 0.65676F }, { 4.50000F, 0.65617F }, { 5.40000F, 0.65473F }, { 6.30000F, 0.65207F },
    { 7.20000F, 0.64913F }, { 8.10000F, 0.64519F }, { 9.00000F, 0.64118F }, { 9.90000F, 0.63707F },
    { 10.80000F, 0.63161F }, { 11.70000F, 0.62889F }, { 12.60000F, 0.62211F }, { 13.50000F, 0.61503F },
    { 14.40000F, 0.60473F }, { 15.30000F, 0.59359F }, { 16.20000F, 0.58159F }, { 17.10000F, 0.56907F },
    { 18.00000F, 0.55633F }, { 18.90000F, 0.54344F }, { 19.80000F, 0.53001F }, { 20.70000F, 0.51531F },
    { 21.60000F, 0.49711F }, { 22.50000F, 0.47744F }, { 23.40000F, 0.45818F }, { 24.30000F, 0.43884F },
    { 25.20000F, 0.41917F }, { 26.10000F, 0.39954F }, { 27.00000F, 0.37793F }, { 27.90000F, 0.35501F },
    { 28.80000F, 0.33171F }, { 29.70000F, 0.30684F }, { 30.60000F, 0.28187F }, { 31.50000F, 0.25732F },
    { 32.40000F, 0.22999F }, { 33.30000F, 0.20212F }, { 34.20000F, 0.17373F }, { 35.10000F, 0.14399F },
    { 36.00000F, 0.11725F }, { 36.90000F, 0.09801F }, { 37.80000F, 0.08237F }, { 38.70000F, 0.06934F },
    { 39.60000F, 0.05785F }, { 40.50000F, 0.04836F }, { 41.40001F, 0.03978F }, { 42.30000F, 0.03220F },
    { 43.20000F, 0.02613F }, { 44.10000F, 0.02063F }, { 45.00000F, 0.01595F }, { 45.90000F, 0.01213F },
    { 46.80000F, 0.00893F }, { 47.70000F, 0.00630F }, { 48.60000F, 0.00445F }, { 49.50000F, 0.00273F },
    { 50.40000F, 0.00157F }, { 51.30000F, 0.00081F }, { 52.20000F, 0.00036F }, { 53.10000F, 0.00012F },
    { 54.00000F, 0.00001F }, { 54.90000F, 0.00000F }, { 55.80000F, 0.00000F }, { 56.70000F, 0.00000F },
    { 57.60000F, 0.00000F }, { 58.50000F, 0.00000F }, { 59.40000F, 0.00000F }, { 60.30000F, 0.00000F },
    { 61.20000F, 0.00000F }, { 62.10001F, 0.00000F }, { 63.00000F, 0.00000F }, { 63.90001F, 0.00000F },
    { 64.80000F, 0.00000F }, { 65.70000F, 0.00000F }, { 66.60001F, 0.00000F }, { 67.50000F, 0.00000F },
    { 68.39999F, 0.00000F }, { 69.30000F, 0.00000F }, { 70.20000F, 0.00000F }, { 71.10000F, 0.00000F },
    { 72.00000F, 0.00000F }, { 72.90000F, 0.00000F }, { 73.80000F, 0.00000F }, { 74.70000F, 0.00000F },
    { 75.60000F, 0.00000F }, { 76.50000F, 0.00000F }, { 77.40000F, 0.00000F }, { 78.30000F, 0.00000F },
    { 79.20000F, 0.00000F }, { 80.10001F, 0.00000F }, { 81.00001F, 0.00000F }, { 81.90000F, 0.00000F },
    { 82.80001F, 0.00000F }, { 83.70000F, 0.00000F }, { 84.60000F, 0.00000F }, { 85.50001F, 0.00000F },
    { 86.40000F, 0.00000F }, { 87.30000F, 0.00000F }, { 88.20000F, 0.00000F }, { 89.10001F, 0.00000F },
    { 90.00000F, 0.00000F },
};
ATN_DEV float deg2rad(float d) { return (kPi * (d) / 180.0F); }           // aten::Deg2Rad, math/math.h:18-21
ATN_DEV float retro_era(const f3& into_sheet_dir, const f3& surface_normal)    // GetEffectiveRetroreflectiveArea, :167-202
{
    const float c = dot(into_sheet_dir, -surface_normal);
    if (c < 0.0F) return 0.0F;
    const float theta = acosf(c);
    const float step = deg2rad(90.00000F) / (float)(101 - 1);
    const uint32_t idx = (uint32_t)(theta / step);
    if (idx >= 101u) return 0.0F;
    const float d = deg2rad(kEraTable[idx].deg);
    const float t = smin(1.0F, fabsf(d - theta) / step);
    const float a = kEraTable[idx].area;
    const float b = idx < 100u ? kEraTable[idx + 1].area : 0.0F;
    return a * (1 - t) + b * t;
}
ATN_DEV float retro_roughness(float roughness, float ni, float nt, const f3& wi, const f3& wn)    // ComputeRoughness, :204-231
{
    const f3 uo = -wi;
    const f3 ut = refract_vector(ni, nt, wi, wn);
    const float n = nt / ni;
    const float J1_denom = dot(-wi, wn) + n * dot(ut, wn);
    const float J1 = J1_denom > 0 ? fabsf(dot(uo, wn)) / sqr(J1_denom) : 0.0F;
    const float J2_denom = -n * dot(ut, wn) + dot(uo, wn);
    const float J2 = J2_denom > 0 ? fabsf(dot(uo, wn)) / sqr(J2_denom) : 0.0F;
    const float a2 = roughness * roughness;
    const float a0 = (J1 > 0 ? a2 / J1 : 0.0F) + (J2 > 0 ? a2 / J2 : 0.0F);
    return sqrtf(a0);
}
ATN_DEV f3 retro_rr_brdf(float roughness, float ior, const f3& wn, const f3& wi, const f3& wo, float& used_E, float& used_F)   // :233-274
{
    const float ni = 1.0F, nt = ior;
    const f3 uo = -wi;
    const f3 ut = refract_vector(ni, nt, wi, wn);
    const float E = retro_era(ut, wn);
    used_E = E;
    const float a = retro_roughness(roughness, ni, nt, wi, wn);
    const float D = beckman_D(wo, uo, a);
    float F = (1.0F - schlick_fresnel(ni, nt, -wi, wn));
    F *= (1.0F - schlick_fresnel(ni, nt, wo, wn));
    used_F = F;
    float G = beckman_G1(roughness, wi, ut);
    G *= beckman_G1(roughness, ut, wo);
    const float c = fabsf(dot(wo, wn));
    return mk3(c > 0 ? (((E * F) * G) * D) / c : 0.0F);
}
ATN_DEV float retro_rr_pdf(float roughness, float ni, float nt, const f3& wn, const f3& wi, const f3& wo)     // :276-292
{
    const f3 uo = -wi;
    const float a = retro_roughness(roughness, ni, nt, wi, wn);
    const float D = beckman_D(wo, uo, a);
    return D * fabsf(dot(uo, wo));
}
ATN_DEV f3 retro_rr_dir(float r1, float r2, float roughness, float ni, float nt, const f3& wi, const f3& wn)    // :294-327
{
    const f3 uo = -wi;
    const float a = retro_roughness(roughness, ni, nt, wi, wn);
    const float a2 = a * a;
    const float theta = atanf(sqrtf(-a2 * logf(1.0F - r1 * 0.99F)));
    const float phi = kPi2 * r2;
    f3 t, b;
    tangent_coordinate(uo, t, b);
    const float costheta = cosf(theta), sintheta = sinf(theta);
    const float cosphi = cosf(phi), sinphi = sinf(phi);
    const f3 wo = ((t * sintheta) * cosphi + (b * sintheta) * sinphi) + uo * costheta;
    return normalize(wo);
}
ATN_DEV f3 retro_diffuse_brdf(float E, float F, float ni, float nt)      // RetroreflectiveDiffuse::EvalBRDF, :332-358
{
    const float kd = 1.0F;
    const float brdf_0 = ((F * (1.0F - E)) * sqr(ni / nt)) * (kd / kPi);
    float f0 = (ni - nt) / (ni + nt);
    f0 = f0 * f0;
    const float Fd = (1.0F - f0) * (-160.0F / 21.0F);
    return mk3(brdf_0 / (1.0F - kd * Fd));
}
ATN_DEV float retro_diffuse_pdf(const f3& n, const f3& wo) { return 1.0F / (1.0F - diffuse_pdf(n, wo)); }      // :360-369
struct RetroW { float r, rr, d; };
ATN_DEV RetroW retro_weights(float ni, float nt, const f3& wi, const f3& n)      // ComputeWeights, :379-407
{
    RetroW w;
    const float F = schlick_fresnel(ni, nt, -wi, n);
    w.r = F;
    const f3 ut = refract_vector(ni, nt, wi, n);
    const float E = retro_era(ut, n);
    w.rr = (1 - F) * E;
    w.d = (1 - F) * (1 - E);
    float norm = 0.0F;
    norm += w.r; norm += w.rr; norm += w.d;
    w.r /= norm; w.rr /= norm; w.d /= norm;
    return w;
}
// the E / F the diffuse term takes when the retroreflection lobe was not evaluated (:470-478, :559-567)
ATN_DEV void retro_EF(float ni, float nt, const f3& wi, const f3& n, const f3& wo, float& E, float& F)
{
    const f3 ut = refract_vector(ni, nt, wi, n);
    E = retro_era(ut, n);
    F = (1.0F - schlick_fresnel(ni, nt, -wi, n));
    F *= (1.0F - schlick_fresnel(ni, nt, wo, n));
}
ATN_DEV float retro_pdf(const DevMaterial& m, const f3& n, const f3& wi, const f3& wo)      // :419-446
{
    const float roughness = m.roughness, ni = 1.0F, nt = m.ior;
    const RetroW w = retro_weights(ni, nt, wi, n);
    float pdf = 0.0F;
    if (w.r > 0.0F) pdf += w.r * beckman_pdf(roughness, n, wi, wo);
    if (w.rr > 0.0F) pdf += w.rr * retro_rr_pdf(roughness, ni, nt, n, wi, wo);
    if (w.d > 0.0F) pdf += w.d * retro_diffuse_pdf(n, wo);
    return pdf;
}
ATN_DEV MtrlSample retro_bsdf(const DevMaterial& m, const f3& n, const f3& wi, const f3& wo)      // :448-500
{
    const float roughness = m.roughness, ior = m.ior, ni = 1.0F, nt = ior;
    const RetroW w = retro_weights(ni, nt, wi, n);
    f3 f_r = mk3(0.0F), f_rr = mk3(0.0F), f_d = mk3(0.0F);
    float pdf = 0.0F;
    if (w.r > 0.0F) {
        f_r = beckman_brdf(roughness, ior, n, wi, wo);
        pdf += beckman_pdf(roughness, n, wi, wo) * w.r;
    }
    float used_E = 0.0F, used_F = 0.0F;
    if (w.rr > 0.0F) {
        f_rr = retro_rr_brdf(roughness, ior, n, wi, wo, used_E, used_F);
        pdf += retro_rr_pdf(roughness, ni, nt, n, wi, wo) * w.rr;
    }
    else retro_EF(ni, nt, wi, n, wo, used_E, used_F);
    if (w.d > 0.0F) {
        f_d = retro_diffuse_brdf(used_E, used_F, ni, nt);
        pdf += retro_diffuse_pdf(n, wo) * w.d;
    }
    MtrlSample r;
    r.bsdf = (f_r + f_rr) + f_d;
    r.pdf = pdf;
    r.dir = wo;
    return r;
}
ATN_DEV void retro_sample(MtrlSample& res, const DevMaterial& m, const f3& n, const f3& wi, Cmj& smp)      // :502-611
{
    const float r1 = cmj_next(smp), r2 = cmj_next(smp), r3 = cmj_next(smp);
    const float roughness = m.roughness, ior = m.ior, ni = 1.0F, nt = ior;
    RetroW w = retro_weights(ni, nt, wi, n);
    const float c0 = w.r, c1 = c0 + w.rr;         // GetCDF, :409-417
    f3 f_r = mk3(0.0F), f_rr = mk3(0.0F), f_d = mk3(0.0F);
    float pdf = 0.0F;
    f3 wo = mk3(0.0F);
    float used_E = 0.0F, used_F = 0.0F;
    if (r3 < c0) {
        wo = reflect_vector(wi, beckman_sample_m(roughness, n, r1, r2));
        f_r = beckman_brdf(roughness, ior, n, wi, wo);
        pdf += beckman_pdf(roughness, n, wi, wo) * w.r;
        w.r = 0.0F;
    }
    else if (r3 < c1) {
        wo = retro_rr_dir(r1, r2, roughness, ni, nt, wi, n);
        f_rr = retro_rr_brdf(roughness, ior, n, wi, wo, used_E, used_F);
        pdf += retro_rr_pdf(roughness, ni, nt, n, wi, wo) * w.rr;
        w.rr = 0.0F;
    }
    else {
        // the reference evaluates F with `wo` BEFORE it assigns it (:559-565; `aten::vec3 wo;` is still unset there):
        // both sides of the parity test take the unset vector as (0, 0, 0)
        retro_EF(ni, nt, wi, n, wo, used_E, used_F);
        wo = diffuse_dir(n, r1, r2);
        f_d = retro_diffuse_brdf(used_E, used_F, ni, nt);
        pdf += retro_diffuse_pdf(n, wo) * w.d;
        w.d = 0.0F;
    }
    if (w.r > 0.0F) {
        f_r = beckman_brdf(roughness, ior, n, wi, wo);
        pdf += beckman_pdf(roughness, n, wi, wo) * w.r;
    }
    if (w.rr > 0.0F) {
        f_rr = retro_rr_brdf(roughness, ior, n, wi, wo, used_E, used_F);
        pdf += retro_rr_pdf(roughness, ni, nt, n, wi, wo) * w.rr;
    }
    if (w.d > 0.0F) {
        retro_EF(ni, nt, wi, n, wo, used_E, used_F);
        f_d = retro_diffuse_brdf(used_E, used_F, ni, nt);
        pdf += retro_diffuse_pdf(n, wo) * w.d;
    }
    res.pdf = pdf;
    res.bsdf = (f_r + f_rr) + f_d;
    res.dir = wo;
}

// CarPaint, material/car_paint.cpp:14-236 + FlakesNormal (material/FlakesNormal.cpp:5-185, FlakesNormal.h:20-52): clearcoat
// (Beckman) over procedural metal flakes over a diffuse base, chosen by one random number that material::applyNormal
// draws BEFORE next-event estimation ("pre_sampled_r") and that the light evaluation and the direction sampling share.
struct CarPaintP {
    f3 clearcoat_color; float clearcoat_ior;
    f3 flakes_color; float clearcoat_roughness;
    f3 diffuse_color; float flake_scale;
    float flake_size, flake_size_variance, flake_normal_orientation, flake_color_multiplier;
};
ATN_DEV CarPaintP carpaint_params(const DevScene& sc, int32_t mtrl_id)
{
    const float4 a = sc.carpaint[4 * mtrl_id], b = sc.carpaint[4 * mtrl_id + 1], c = sc.carpaint[4 * mtrl_id + 2], d = sc.carpaint[4 * mtrl_id + 3];
    CarPaintP p;
    p.clearcoat_color = mk3(a); p.clearcoat_ior = a.w;
    p.flakes_color = mk3(b); p.clearcoat_roughness = b.w;
    p.diffuse_color = mk3(c); p.flake_scale = c.w;
    p.flake_size = d.x; p.flake_size_variance = d.y; p.flake_normal_orientation = d.z; p.flake_color_multiplier = d.w;
    return p;
}
ATN_DEV float compute_fresnel(float ni, float nt, const f3& wi, const f3& normal)     // material::computeFresnel, material.h:445-467
{
    float cosi = dot(normal, wi);
    if (cosi < 0) { const float t = ni; ni = nt; nt = t; cosi = -cosi; }
    const float nnt = ni / nt;
    const float sini2 = 1.0F - cosi * cosi;
    const float sint2 = (nnt * nnt) * sini2;
    const float cost = sqrtf(smax(0.0F, 1.0F - sint2));
    const float rp = (nt * cosi - ni * cost) / (nt * cosi + ni * cost);
    const float rs = (ni * cosi - nt * cost) / (ni * cosi + nt * cost);
    return (rp * rp + rs * rs) * 0.5F;
}
ATN_DEV float flake_density(float flake_size, float aspect_wh)     // FlakesNormal::computeFlakeDensity, FlakesNormal.h:20-52
{
    const float aspect = 1.0F / aspect_wh;
    const float D = ((kPi * flake_size) * flake_size) * aspect;
    return smin(D, 1.0F);
}
// Bob Jenkins' lookup3 mix / final (public domain), as FlakesNormal.cpp:12-44 uses them
ATN_DEV uint32_t rotl32(uint32_t v, uint32_t h) { return (v << h) | (v >> (32u - h)); }
ATN_DEV uint32_t flake_inthash(const float k[4])
{
    // float -> uint32 of possibly NEGATIVE cell coordinates: the reference's x86-64 build converts through a 64-bit
    // integer and keeps the low 32 bits (two's-complement wrap); v_cvt_u32_f32 would clamp to 0, so spell the wrap out
    const uint32_t len = 4;
    uint32_t a = 0xdeadbeefu + (len << 2) + 13u, b = a, c = a;
    a += (uint32_t)(long long)k[0];
    b += (uint32_t)(long long)k[1];
    c += (uint32_t)(long long)k[2];
    a -= c; a ^= rotl32(c, 4); c += b;
    b -= a; b ^= rotl32(a, 6); a += c;
    c -= b; c ^= rotl32(b, 8); b += a;
    a -= c; a ^= rotl32(c, 16); c += b;
    b -= a; b ^= rotl32(a, 19); a += c;
    c -= b; c ^= rotl32(b, 4); b += a;
    a += (uint32_t)(long long)k[3];
    c ^= b; c -= rotl32(b, 14);
    a ^= c; a -= rotl32(c, 11);
    b ^= a; b -= rotl32(a, 25);
    c ^= b; c -= rotl32(b, 16);
    a ^= c; a -= rotl32(c, 4);
    b ^= a; b -= rotl32(a, 14);
    c ^= b; c -= rotl32(b, 24);
    return c;
}
ATN_DEV f3 flake_cellnoise(const f3& p)      // cellnoise + hash3, FlakesNormal.cpp:93-123
{
    float iv[4] = { floorf(p.x), floorf(p.y), floorf(p.z), 0.0F };
    const float to01 = 1.0F / (float)0xffffffffu;
    f3 r;
    iv[3] = 0.0F; r.x = (float)flake_inthash(iv) * to01;
    iv[3] = 1.0F; r.y = (float)flake_inthash(iv) * to01;
    iv[3] = 2.0F; r.z = (float)flake_inthash(iv) * to01;
    return r;
}
ATN_DEV float4 flakes_normal_gen(float u, float v, float flake_scale, float flake_size, float flake_size_variance, float flake_normal_orientation)
{
    // FlakesNormal::gen, FlakesNormal.cpp:125-183
    const float safe_var = sclamp(flake_size_variance, 0.1F, 1.0F);
    const float cx[9] = { 0.5F, 1.5F, 1.5F, 0.5F, -0.5F, -0.5F, -0.5F, 0.5F, 1.5F };
    const float cy[9] = { 0.5F, 0.5F, 1.5F, 1.5F, 1.5F, 0.5F, -0.5F, -0.5F, -0.5F };
    const f3 position = flake_scale * mk3(u, v, 0.0F);
    const f3 base = mk3(floorf(position.x), floorf(position.y), floorf(position.z));
    f3 nearest = mk3(0.0F, 0.0F, 1.0F);
    int32_t nearest_idx = -1;
#pragma unroll 1
    for (int32_t i = 0; i < 9; ++i) {
        f3 center = base + mk3(cx[i], cy[i], 0.0F);
        f3 off = flake_cellnoise(center) * 2.0F + (-1.0F);
        off.z *= safe_var;
        off = normalize(off);
        center = center + 0.5F * off;
        const float dist = length(position - center);        // glm::distance
        if (dist < flake_size && center.z < nearest.z) { nearest = center; nearest_idx = i; }
    }
    f3 result = mk3(0.5F, 0.5F, 1.0F);
    float alpha = 0.0F;
    if (nearest_idx != -1) {
        f3 rn = flake_cellnoise((base + mk3(cx[nearest_idx], cy[nearest_idx], 0.0F)) + mk3(0.0F, 0.0F, 1.5F));
        rn = 2.0F * rn + (-1.0F);
        // glm::faceforward(N, I, Nref) = dot(Nref, I) < 0 ? N : -N with N = Nref = rn, I = (0, 0, 1)
        rn = dot(rn, mk3(0.0F, 0.0F, 1.0F)) < 0.0F ? rn : -rn;
        rn = normalize(mix3(rn, mk3(0.0F, 0.0F, 1.0F), flake_normal_orientation));
        result = rn;
        alpha = 1.0F;
    }
    return make_float4(result.x, result.y, result.z, alpha);
}
// material::applyNormal (material_impl.h:208-230): CarPaint::applyNormalMap (car_paint.cpp:195-236) draws the shared
// random number and may replace the normal by a flake's; every other material takes its normal map and returns -1.
// MS < kMsCarPaint compiles CarPaint out (see the material sets at sample_material).
template <int MS>
ATN_DEV float apply_normal(const DevScene& sc, const DevMaterial& m, int32_t mtrl_id, f3& nml, float u, float v, const f3& wi, Cmj& smp)
{
    if (MS < kMsCarPaint || m.type != ATN_MTRL_CARPAINT) {
        nml = apply_normal_map(sc, m.normalMap, nml, u, v);
        return -1.0F;
    }
    const CarPaintP p = carpaint_params(sc, mtrl_id);
    const f3 V = -wi;
    const f3 N = normalize(nml);
    const float r0 = cmj_next(smp);
    const float fresnel = compute_fresnel(1.0F, p.clearcoat_ior, V, N);
    f3 out = N;
    if (!(r0 < fresnel)) {
        const float4 fl = flakes_normal_gen(u, v, p.flake_scale, p.flake_size, p.flake_size_variance, p.flake_normal_orientation);
        if (fl.w > 0.0F) {
            // applyTangentSpaceCoord, car_paint.cpp:14-23
            const f3 n = normalize(nml);
            f3 t, b;
            tangent_coordinate(n, t, b);
            out = normalize((fl.z * n + fl.x * t) + fl.y * b);
        }
    }
    nml = out;
    return r0;
}
ATN_DEV float carpaint_pdf(const CarPaintP& p, const f3& normal, const f3& wi, const f3& wo)      // car_paint.cpp:25-57
{
    const f3 V = -wi;
    const float fresnel = compute_fresnel(1.0F, p.clearcoat_ior, V, normal);
    const float bp = beckman_pdf(p.clearcoat_roughness, normal, wi, wo);
    const float fbp = beckman_pdf(1.0F, normal, wi, wo);
    const float dens = flake_density(p.flake_size, 1.0F);
    const float dp = diffuse_pdf(normal, wo);
    const float pdf = fresnel * bp + (1.0F - fresnel) * (dens * fbp + (1 - dens) * dp);
    return sclamp(pdf, 0.0F, 1.0F);
}
ATN_DEV f3 carpaint_dir(const CarPaintP& p, const f3& normal, const f3& wi, Cmj& smp, float pre_r)      // car_paint.cpp:59-109
{
    const f3 V = -wi;
    float r0 = pre_r;
    float r1 = cmj_next(smp);
    const float fresnel = compute_fresnel(1.0F, p.clearcoat_ior, V, normal);
    const float dens = flake_density(p.flake_size, 1.0F);
    if (r0 < fresnel) {
        r0 /= fresnel;
        return reflect_vector(wi, beckman_sample_m(p.clearcoat_roughness, normal, r0, r1));
    }
    r0 -= fresnel;
    r0 /= (1.0F - fresnel);
    if (r1 < dens) {
        r1 /= dens;
        return reflect_vector(wi, beckman_sample_m(1.0F, normal, r0, r1));
    }
    r1 -= dens;
    r1 /= (1.0F - dens);
    return diffuse_dir(normal, r0, r1);
}
ATN_DEV f3 carpaint_bsdf(const DevScene& sc, const DevMaterial& m, const CarPaintP& p, const f3& normal, const f3& wi, const f3& wo,
                         float u, float v, float pre_r)      // car_paint.cpp:111-176
{
    const f3 albedo = mk3(sample_texture(sc, m.albedoMap, u, v, make_float4(1.0F, 1.0F, 1.0F, 1.0F)));
    const f3 V = -wi;
    const float fresnel = compute_fresnel(1.0F, p.clearcoat_ior, V, normal);
    f3 bsdf;
    if (pre_r < fresnel) {
        bsdf = beckman_brdf(p.clearcoat_roughness, p.clearcoat_ior, normal, wi, wo);
        bsdf = bsdf * p.clearcoat_color;
    }
    else {
        const bool on_flakes = flakes_normal_gen(u, v, p.flake_scale, p.flake_size, p.flake_size_variance, p.flake_normal_orientation).w > 0.0F;
        if (on_flakes) {
            bsdf = beckman_brdf(1.0F, 10.0F, normal, wi, wo);
            bsdf = bsdf * (p.flakes_color * p.flake_color_multiplier);
        }
        else {
            bsdf = p.diffuse_color / kPi;
        }
    }
    return albedo * bsdf;
}

// material::sampleMaterial / samplePDF / sampleBSDF, material/material_impl.h:24-206
// Material set MS of a k_shade instantiation: kMsCore .. kMsToon (scene_dev.hpp), chosen by the host from the uploaded
// materials (DevScene::material_set); BSDFs outside the set are compiled out (shade per frame on sponza_lod, measured:
// core + Disney 1.32 ms, + the analytic ones 1.35, + CarPaint 1.65; core without Disney: frame -2 %).
// mtrl_id / pre_r: only CarPaint reads them (its parameter block and the random number material::applyNormal drew).
template <int MS = kMsCarPaint>
ATN_DEV void sample_material(MtrlSample& r, const DevScene& sc, const DevMaterial& m, const f3& normal,
                             const f3& wi, Cmj& smp, float u, float v, int32_t mtrl_id = 0, float pre_r = 0.0F, const HitPre* pre = nullptr)
{
    if (MS >= kMsCarPaint && m.type == ATN_MTRL_CARPAINT) {       // CarPaint::sample, car_paint.cpp:178-193
        const CarPaintP p = carpaint_params(sc, mtrl_id);
        r.dir = carpaint_dir(p, normal, wi, smp, pre_r);
        r.pdf = carpaint_pdf(p, normal, wi, r.dir);
        r.bsdf = carpaint_bsdf(sc, m, p, normal, wi, r.dir, u, v, pre_r);
        return;
    }
    if (MS >= kMsAnalytic) {
        switch (m.type) {
        case ATN_MTRL_REFRACTION:
            refraction_sample(r, m, normal, wi, smp);
            return;
        case ATN_MTRL_BECKMAN: {
            const float rough = pre ? pre->rough : ggx_roughness(sc, m, u, v);
            const float r1 = cmj_next(smp), r2 = cmj_next(smp);
            r.dir = reflect_vector(wi, beckman_sample_m(rough, normal, r1, r2));
            r.pdf = beckman_pdf(rough, normal, wi, r.dir);
            r.bsdf = beckman_brdf(rough, m.ior, normal, wi, r.dir);
            return;
        }
        case ATN_MTRL_VELVET: {
            const float r1 = cmj_next(smp), r2 = cmj_next(smp);
            r.dir = diffuse_dir(normal, r1, r2, pre);
            r.pdf = diffuse_pdf(normal, r.dir);
            r.bsdf = velvet_brdf(pre ? pre->rough : ggx_roughness(sc, m, u, v), normal, wi, r.dir);
            return;
        }
        case ATN_MTRL_MICROFACET_REFRACTION:
            microfacet_refraction_sample(r, sc, m, normal, wi, smp, u, v);
            return;
        case ATN_MTRL_RETROREFLECTIVE:
            retro_sample(r, m, normal, wi, smp);
            return;
        case ATN_MTRL_OREN_NAYAR: {
            const float r1 = cmj_next(smp), r2 = cmj_next(smp);
            r.dir = diffuse_dir(normal, r1, r2, pre);
            r.pdf = oren_nayar_pdf(normal, r.dir);
            r.bsdf = oren_nayar_brdf(pre ? pre->rough : ggx_roughness(sc, m, u, v), normal, wi, r.dir);
            return;
        }
        default: break;
        }
    }
    switch (m.type) {
    case ATN_MTRL_SPECULAR: {
        r.dir = reflect_vector(wi, normal);
        r.pdf = 1.0F;
        const float c = dot(normal, r.dir);
        r.bsdf = mk3(c == 0.0F ? 0.0F : 1.0F / c);
        break;
    }
    case ATN_MTRL_GGX: {
        const float rough = pre ? pre->rough : ggx_roughness(sc, m, u, v);
        const float r1 = cmj_next(smp), r2 = cmj_next(smp);
        r.dir = ggx_dir(r1, r2, rough, wi, normal, pre);
        r.pdf = ggx_pdf(rough, normal, wi, r.dir);
        r.bsdf = ggx_brdf(rough, m.ior, normal, wi, r.dir, pre);
        break;
    }
    case ATN_MTRL_DISNEY:
        if (MS >= kMsDisney) { disney_sample(r, m, normal, wi, smp, pre); break; }
        [[fallthrough]];
    default: {  // Diffuse, Emissive (emissive.h:70-83) and the reference's fallback
        const float r1 = cmj_next(smp), r2 = cmj_next(smp);
        r.dir = diffuse_dir(normal, r1, r2, pre);
        r.pdf = diffuse_pdf(normal, r.dir);
        r.bsdf = diffuse_brdf();
        break;
    }
    }
}
template <int MS = kMsCarPaint>
ATN_DEV float material_pdf(const DevScene& sc, const DevMaterial& m, const f3& normal, const f3& wi, const f3& wo, float u, float v,
                           int32_t mtrl_id = 0, const HitPre* pre = nullptr)
{
    if (MS >= kMsCarPaint && m.type == ATN_MTRL_CARPAINT) return carpaint_pdf(carpaint_params(sc, mtrl_id), normal, wi, wo);
    if (MS >= kMsToon && m.type == ATN_MTRL_TOON_SPECULAR) {       // ToonSpecular::ComputePDF, toon.cpp:288-301
        const f3 V = -wi;
        return ggx_pdf_h(m.roughness, normal, toon_specular_half(sc.toon[mtrl_id], normal, V, wo), wo);
    }
    if (MS >= kMsAnalytic) {
        switch (m.type) {
        case ATN_MTRL_REFRACTION: return 1.0F;
        case ATN_MTRL_BECKMAN: return beckman_pdf(pre ? pre->rough : ggx_roughness(sc, m, u, v), normal, wi, wo);
        case ATN_MTRL_OREN_NAYAR: return oren_nayar_pdf(normal, wo);
        case ATN_MTRL_VELVET: return diffuse_pdf(normal, wo);
        case ATN_MTRL_MICROFACET_REFRACTION: return 1.0F;
        case ATN_MTRL_RETROREFLECTIVE: return retro_pdf(m, normal, wi, wo);
        default: break;
        }
    }
    switch (m.type) {
    case ATN_MTRL_SPECULAR: return 1.0F;
    case ATN_MTRL_GGX: return ggx_pdf(pre ? pre->rough : ggx_roughness(sc, m, u, v), normal, wi, wo);
    case ATN_MTRL_DISNEY: if (MS >= kMsDisney) return disney_pdf(m, normal, wi, wo); [[fallthrough]];
    default: return diffuse_pdf(normal, wo);
    }
}
template <int MS = kMsCarPaint>
ATN_DEV MtrlSample material_bsdf(const DevScene& sc, const DevMaterial& m, const f3& normal, const f3& wi, const f3& wo, float u, float v,
                                 int32_t mtrl_id = 0, float pre_r = 0.0F, const HitPre* pre = nullptr)
{
    MtrlSample r; r.pdf = 0.0F; r.dir = wo; r.bsdf = mk3(0.0F);
    if (MS >= kMsCarPaint && m.type == ATN_MTRL_CARPAINT) { r.bsdf = carpaint_bsdf(sc, m, carpaint_params(sc, mtrl_id), normal, wi, wo, u, v, pre_r); return r; }
    if (MS >= kMsToon && m.type == ATN_MTRL_TOON_SPECULAR) {       // ToonSpecular::ComputeBRDF, toon.cpp:303-322
        const f3 V = -wi;
        r.bsdf = ggx_brdf_h(m.roughness, m.ior, normal, V, wo, toon_specular_half(sc.toon[mtrl_id], normal, V, wo));
        return r;
    }
    if (MS >= kMsAnalytic) {
        switch (m.type) {
        case ATN_MTRL_REFRACTION: r.bsdf = mk3(0.0F); return r;
        case ATN_MTRL_BECKMAN: r.bsdf = beckman_brdf(pre ? pre->rough : ggx_roughness(sc, m, u, v), m.ior, normal, wi, wo); return r;
        case ATN_MTRL_OREN_NAYAR: r.bsdf = oren_nayar_brdf(pre ? pre->rough : ggx_roughness(sc, m, u, v), normal, wi, wo); return r;
        case ATN_MTRL_VELVET: r.bsdf = velvet_brdf(pre ? pre->rough : ggx_roughness(sc, m, u, v), normal, wi, wo); return r;
        case ATN_MTRL_MICROFACET_REFRACTION: r.bsdf = mk3(0.0F); return r;
        case ATN_MTRL_RETROREFLECTIVE: return retro_bsdf(m, normal, wi, wo);
        default: break;
        }
    }
    switch (m.type) {
    case ATN_MTRL_SPECULAR: { const float c = dot(normal, wo); r.bsdf = mk3(c == 0.0F ? 0.0F : 1.0F / c); break; }
    case ATN_MTRL_GGX: r.bsdf = ggx_brdf(pre ? pre->rough : ggx_roughness(sc, m, u, v), m.ior, normal, wi, wo, pre); break;
    case ATN_MTRL_DISNEY: if (MS >= kMsDisney) { r = disney_bsdf(m, normal, wi, wo); break; } [[fallthrough]];
    default: r.bsdf = diffuse_brdf(); break;
    }
    return r;
}

// ------------------------------------------------------------------ background / lights
ATN_DEV void direction_to_uv(const f3& dir, float& u, float& v)    // renderer/background.h:88-127
{
    const float temp = atan2f(dir.x, dir.z);
    const float r = length(dir);
    const float phi = (temp >= 0) ? temp : (temp + 2 * kPi);
    const float theta = acosf(dir.y / r);
    u = phi / (2 * kPi);
    v = 1 - theta / kPi;
}
ATN_DEV float4 background_sample(const DevScene& sc, const f3& dir)   // background.h:34-62
{
    if (sc.envmap_tex_idx < 0 || !sc.enable_env_map) {
        return make_float4(sc.bg_color[0], sc.bg_color[1], sc.bg_color[2], 0.0F);
    }
    float u, v;
    direction_to_uv(dir, u, v);
    const float4 c = sample_texture(sc, sc.envmap_tex_idx, u, v, make_float4(1, 1, 1, 1));
    return mul4(sc.multiplyer, c);
}

struct LightSample { f3 pos, dir, nml, color; float dist, pdf; uint32_t attrib; };

ATN_DEV f3 area_light_color(const atn_light_param& p, float area)   // light/arealight.h:58-63
{
    const float lum = (p.scale * p.intensity) / area;
    return mk3(p.light_color[0], p.light_color[1], p.light_color[2]) * lum;
}

// samplePdfAndCdf, light/ibl.cpp:133-176: binary search of a CDF normalised to [0, 1]; returns the cell and its probability
ATN_DEV int32_t ibl_sample_cdf(float r, const float* __restrict__ cdf, int32_t n, float& out_pdf)
{
    if (n < 2) { out_pdf = n == 1 ? cdf[0] : 0.0F; return 0; }     // (the reference's loop needs two cells)
    int32_t top = 0, tail = n - 1;
    for (;;) {
        const int32_t mid = (top + tail) >> 1;
        if (r < cdf[mid]) tail = mid; else top = mid;
        if (tail - top == 1) {
            const float top_c = cdf[top], tail_c = cdf[tail];
            if (r <= top_c) { out_pdf = top_c; return top; }
            out_pdf = tail_c - top_c;
            return tail;
        }
    }
}
// Background::ConvertUVToDirection, renderer/background.h:64-86
ATN_DEV f3 uv_to_direction(float u, float v)
{
    const float phi = (2 * kPi) * u;
    const float theta = (1 - v) * kPi;
    f3 dir;
    dir.y = cosf(theta);
    const float xz = sqrtf(1 - dir.y * dir.y);
    dir.x = xz * sinf(phi);
    dir.z = xz * cosf(phi);
    return normalize(dir);
}
// Solid-angle density of the table sampler for texel (x, y): P(x, y) * w * h / (2 pi^2 sin(theta_y)).
// (ImageBasedLight::sample writes pi^2 where the texel's solid angle (2 pi / w)(pi / h) sin(theta) calls for 2 pi^2; with
// its constant the estimator is half as bright as the scene.  This is an optional sampler, not a parity path: the
// density used here is the true one, and the SAME function prices both the light sample and the BSDF-sampled miss.)
ATN_DEV float ibl_texel_pdf(const DevScene& sc, float pdf_u, float pdf_v, int32_t y)
{
    // (here(): this optional sampler's scalars are converted where they are used, not hoisted through all of k_shade)
    const int32_t w = here(sc.ibl_w), h = here(sc.ibl_h);
    const float v = (float)((double)y + 0.5) / (float)h;
    const float theta = kPi * v;
    const float pi2 = kPi * kPi;
    return (pdf_u * pdf_v) * ((float)(w * h) / ((2.0F * pi2) * sinf(theta)));
}
ATN_DEV float ibl_direction_pdf(const DevScene& sc, const f3& dir)
{
    float u, v;
    direction_to_uv(dir, u, v);
    const int32_t w = here(sc.ibl_w), h = here(sc.ibl_h);
    int32_t x = (int32_t)(u * (float)w), y = (int32_t)(v * (float)h);
    x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
    y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
    const float* __restrict__ cu = sc.ibl_cdf_u + (size_t)y * w;
    const float pu = x > 0 ? cu[x] - cu[x - 1] : cu[0];
    const float pv = y > 0 ? sc.ibl_cdf_v[y] - sc.ibl_cdf_v[y - 1] : sc.ibl_cdf_v[0];
    return ibl_texel_pdf(sc, pu, pv, y);
}

// Light::sample (light/light_impl.h:12-43) and the per-type samplers it dispatches to
ATN_DEV void sample_light(LightSample& res, const atn_light_param& lp, const DevScene& sc, const f3& org, const f3& nml, Cmj& smp,
                          const HitPre* pre = nullptr)
{
    res.pdf = 0.0F; res.dist = 0.0F; res.color = mk3(0.0F);
    res.pos = org; res.dir = mk3(0.0F, 1.0F, 0.0F); res.nml = mk3(0.0F, 1.0F, 0.0F);
    switch (lp.type) {
    case ATN_LIGHT_AREA: {      // AreaLight::sample, arealight.h:65-143 (polygon lights)
        if (lp.arealight_objid < 0) break;
        const atn_object_param* obj = &sc.objects[lp.arealight_objid];
        const atn_object_param* real_obj = obj->type == ATN_OBJ_INSTANCE ? &sc.objects[obj->object_id] : obj;
        if (real_obj->type == ATN_OBJ_SPHERE) {
            // sphere::SamplePosAndNormal (geometry/sphere.cpp:109-150) -> sphere::hit (:30-91) along the ray to
            // the sampled point -> sphere::EvaluateHitResult (:93-107) -> evaluate_hit_result's L2W step
            const float r1 = cmj_next(smp), r2 = cmj_next(smp);
            const float rad = real_obj->sphere.radius;
            const f3 center = mk3(real_obj->sphere.center[0], real_obj->sphere.center[1], real_obj->sphere.center[2]);
            const float z = 2.0F * r1 - 1.0F;
            const float sin_theta = sqrtf(1 - z * z);
            const float phi = (2 * kPi) * r2;
            const float x = cosf(phi) * sin_theta;
            const float y = sinf(phi) * sin_theta;
            const f3 sdir = normalize(mk3(x, y, z));
            const f3 spos = center + sdir * (rad + kEps);
            const f3 rdir = normalize(spos - org);
            // sphere::hit reads the LIGHT object's own parameters (arealight.h:115 passes &obj)
            const f3 ocenter = mk3(obj->sphere.center[0], obj->sphere.center[1], obj->sphere.center[2]);
            const float orad = obj->sphere.radius;
            const f3 p_o = ocenter - org;
            const float b = dot(p_o, rdir);
            const float D4 = (b * b - dot(p_o, p_o)) + orad * orad;
            if (D4 < 0.0F) break;
            const float sqrt_D4 = sqrtf(D4);
            const float t1 = b - sqrt_D4, t2 = b + sqrt_D4;
            // isClose(|b|, sqrt_D4, 2500 ulps), math/math.h:340-372
            int32_t ai = __float_as_int(fabsf(b)), bi = __float_as_int(sqrt_D4);
            if (ai < 0) ai = (int32_t)(0x80000000u - (uint32_t)ai);
            if (bi < 0) bi = (int32_t)(0x80000000u - (uint32_t)bi);
            const bool close = abs(ai - bi) <= 2500;
            float t;
            if (t1 > kEps && !close) t = t1;
            else if (t2 > kEps && !close) t = t2;
            else break;
            const bool is_inst = obj->type == ATN_OBJ_INSTANCE;
            m4 L2W = m4_identity();
            if (is_inst && obj->mtx_id >= 0) L2W = load_m4(sc, obj->mtx_id);
            f3 p = org + t * rdir;
            f3 n = (p - center) / rad;
            const float area = ((4 * kPi) * rad) * rad;
            p = m4_apply(L2W, p);
            n = normalize(m4_applyXYZ(L2W, n));
            res.pos = p;
            res.pdf = 1 / area;
            res.dir = p - org;
            res.dist = length(res.dir);
            res.dir = normalize(res.dir);
            res.nml = n;
            res.color = area_light_color(lp, area);
            break;
        }
        if (real_obj->type != ATN_OBJ_POLYGONS) break;
        // PolygonObject::SamplePosAndNormal (PolygonObject.h:113-156) + triangle::SamplePosAndNormal (triangle.h:122-162)
        const float r = cmj_next(smp);
        uint32_t tri_idx = (uint32_t)((float)real_obj->triangle_num * r);
        tri_idx += (uint32_t)real_obj->triangle_id;
        const float r0 = cmj_next(smp), r1 = cmj_next(smp);
        const float a = sqrtf(r0) * (1.0F - r1);
        const float b = sqrtf(r0) * r1;
        HitRec rec;
        evaluate_hit(rec, sc, lp.arealight_objid, (int32_t)tri_idx, a, b);
        res.pos = rec.p;
        res.pdf = 1 / rec.area;
        res.dir = rec.p - org;
        res.dist = length(res.dir);
        res.dir = normalize(res.dir);
        res.nml = rec.normal;
        res.color = area_light_color(lp, rec.area);
        break;
    }
    case ATN_LIGHT_IBL: {       // ImageBasedLight::sample, light/ibl.h:71-133
        const float r1 = cmj_next(smp), r2 = cmj_next(smp);
        if (sc.ibl_importance) {
            // the table sampler, ImageBasedLight::sample(ctxt, org, nml, sampler), light/ibl.cpp:180-230
            float pdf_u, pdf_v;
            const int32_t iw = here(sc.ibl_w), ih = here(sc.ibl_h);
            const int32_t y = ibl_sample_cdf(r1, sc.ibl_cdf_v, ih, pdf_v);
            const int32_t x = ibl_sample_cdf(r2, sc.ibl_cdf_u + (size_t)y * iw, iw, pdf_u);
            const float u = (float)((double)x + 0.5) / (float)iw;
            const float v = (float)((double)y + 0.5) / (float)ih;
            res.pdf = ibl_texel_pdf(sc, pdf_u, pdf_v, y);
            res.dir = uv_to_direction(u, v);
            const float4 lum = mul4(sc.multiplyer, sample_texture(sc, lp.envmapidx, u, v, make_float4(1, 1, 1, 1)));
            res.color = mk3(mul4(lp.scale, lum));
            res.pos = org + sc.ibl_scene_radius * res.dir;      // (the reference leaves pos unset there: "currently not used")
            res.nml = -normalize(res.dir);
            res.dist = 1.0F;
            break;
        }
        res.dir = diffuse_dir(nml, r1, r2, pre);
        float u, v;
        direction_to_uv(res.dir, u, v);
        res.pos = org + sc.ibl_scene_radius * res.dir;
        res.nml = -normalize(res.dir);
        res.pdf = 1.0f / (2.0f * kPi);
        res.dist = 1.0F;
        const float4 lum = sample_texture(sc, lp.envmapidx, u, v, make_float4(1, 1, 1, 1));
        res.color = mk3(mul4(lp.scale, lum));
        break;
    }
    case ATN_LIGHT_POINT: {     // light/pointlight.h:40-58
        const f3 lpos = mk3(lp.pos.x, lp.pos.y, lp.pos.z);
        const f3 lcol = mk3(lp.light_color[0], lp.light_color[1], lp.light_color[2]);
        res.pdf = 1.0f;
        res.dir = lpos - org;
        res.dist = length(res.dir);
        res.dir = normalize(res.dir);
        res.pos = lpos;
        res.nml = normalize(-res.dir);
        const float dist2 = sqr(res.dist);
        res.color = ((lcol * lp.scale) * lp.intensity) / dist2;
        break;
    }
    case ATN_LIGHT_SPOT: {      // light/spotlight.h:58-92
        const f3 lpos = mk3(lp.pos.x, lp.pos.y, lp.pos.z);
        const f3 lcol = mk3(lp.light_color[0], lp.light_color[1], lp.light_color[2]);
        const f3 ldir = mk3(lp.dir.x, lp.dir.y, lp.dir.z);
        res.pdf = 1.0f;
        res.pos = lpos;
        res.nml = ldir;
        res.dir = lpos - org;
        res.dist = length(res.dir);
        res.dir = normalize(res.dir);
        const float rho = dot(ldir, -res.dir);
        const float cosHalfInner = cosf(lp.innerAngle * 0.5F);
        const float cosHalfOuter = cosf(lp.outerAngle * 0.5F);
        if (rho > cosHalfOuter) {
            float att = (rho - cosHalfOuter) / (cosHalfInner - cosHalfOuter);
            att = sclamp(att, 0.0f, 1.0f);
            const float dist2 = sqr(res.dist);
            res.color = (((lp.scale * lcol) * att) * lp.intensity) / dist2;
        }
        else {
            res.pdf = 0.0f;
            res.color = mk3(0.0F);
        }
        break;
    }
    case ATN_LIGHT_DIRECTION: { // light/directionallight.h:40-63
        const f3 lcol = mk3(lp.light_color[0], lp.light_color[1], lp.light_color[2]);
        res.pdf = 1.0f;
        const float4 nd = normalize4(make_float4(lp.dir.x, lp.dir.y, lp.dir.z, lp.dir.w));
        res.dir = mk3(-nd.x, -nd.y, -nd.z);
        res.nml = mk3(nd);
        res.pos = org + (100000.0F * 0.5F) * res.dir;
        res.color = (lcol * lp.scale) * lp.intensity;
        res.dist = 1.0F;
        break;
    }
    default: break;
    }
    res.attrib = lp.attrib;
}

// How many sampler dimensions sample_light consumes for this light (k_shade takes the draws at the reference's place in the
// sample stream and evaluates the light later): area light over a polygon object 3 (triangle pick PolygonObject.h:121, then
// r0, r1 triangle.h:137-138), over a sphere 2 (sphere.cpp:117-118), image-based light 2 (ibl.h:98-99), punctual lights 0.
ATN_DEV uint32_t light_sample_draws(const atn_light_param& lp, const DevScene& sc)
{
    if (lp.type == ATN_LIGHT_IBL) return 2u;
    if (lp.type != ATN_LIGHT_AREA || lp.arealight_objid < 0) return 0u;
    const atn_object_param* obj = &sc.objects[lp.arealight_objid];
    const int32_t t = obj->type == ATN_OBJ_INSTANCE ? sc.objects[obj->object_id].type : obj->type;
    return t == ATN_OBJ_SPHERE ? 2u : (t == ATN_OBJ_POLYGONS ? 3u : 0u);
}

// ComputeRadianceNEE, renderer/pathtracing/pathtracing_nee_impl.h:23-95.  `then(radiance)` runs where the reference returns a
// value (k_shade stores the shadow job right there: the three floats never cross the join behind the validity test).
template <int MS = kMsCarPaint, class Then>
ATN_DEV bool radiance_nee_then(const DevScene& sc, const f3& wi, const f3& nml, const DevMaterial& m,
                               float hu, float hv, float light_select_prob, const LightSample& ls, int32_t mtrl_id, float pre_r,
                               float* weight_ptr, Then&& then, const HitPre* pre = nullptr)
{
    if (weight_ptr) *weight_ptr = 0.0F;
    const float cosShadow = dot(nml, ls.dir);
    float path_pdf = material_pdf<MS>(sc, m, nml, wi, ls.dir, hu, hv, mtrl_id, pre);
    const MtrlSample ev = material_bsdf<MS>(sc, m, nml, wi, ls.dir, hu, hv, mtrl_id, pre_r, pre);
    if (ev.pdf > 0) path_pdf = ev.pdf;
    const float cosLight = dot(ls.nml, -ls.dir);
    float dist2 = sqr(ls.dist);
    const bool isInfinite = (ls.attrib & ATN_LIGHT_ATTR_INFINITE) != 0;
    const bool is_singular = (ls.attrib & ATN_LIGHT_ATTR_SINGULAR) != 0;
    dist2 = (isInfinite || is_singular) ? 1.0F : dist2;
    if (cosShadow >= 0 && cosLight >= 0 && dist2 > 0 && path_pdf > 0.0F && ls.pdf > 0.0F) {
        if (!isInfinite) path_pdf = (path_pdf * cosLight) / dist2;
        const float f = ls.pdf * light_select_prob;
        const float misW = is_singular ? 1.0f : f / (f + path_pdf);
        const float G = isInfinite ? cosShadow * cosLight : (cosShadow * cosLight) / dist2;
        if (weight_ptr) *weight_ptr = (misW / ls.pdf) / light_select_prob;      // pathtracing_nee_impl.h:87-89
        then(((((misW * ev.bsdf) * ls.color) * G) / ls.pdf) / light_select_prob);
        return true;
    }
    return false;
}
template <int MS = kMsCarPaint>
ATN_DEV bool radiance_nee(f3& out, const DevScene& sc, const f3& wi, const f3& nml, const DevMaterial& m,
                          float hu, float hv, float light_select_prob, const LightSample& ls, int32_t mtrl_id = 0, float pre_r = 0.0F,
                          float* weight_ptr = nullptr)
{
    return radiance_nee_then<MS>(sc, wi, nml, m, hu, hv, light_select_prob, ls, mtrl_id, pre_r, weight_ptr, [&](const f3& r) { out = r; });
}

} // namespace atn
