/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h / orc_core.h for status).
 *
 * orc_pt.h: BSDFs, lights, NEE and the radiance loop of aten::PathTracing (CPU path).
 */
#pragma once
#include "orc_core.h"

namespace orc {

struct MaterialSampling {   // material/material.h:338-347
    v3 dir; v3 bsdf; float pdf{ 0 };
};

inline bool attr_emissive(const atn_material_param& m) { return (m.attrib & ATN_MTRL_ATTR_EMISSIVE) != 0; }
inline bool attr_singular(const atn_material_param& m) { return (m.attrib & ATN_MTRL_ATTR_SINGULAR) != 0; }
inline bool attr_translucent(const atn_material_param& m) { return (m.attrib & ATN_MTRL_ATTR_TRANSLUCENT) != 0; }

// ---- material/material.h:444-565 helpers -------------------------------------------------
inline float ComputeSchlickFresnelWithF0AndCosTheta(float f0, float costheta)
{
    const float c = saturate_(1 - costheta);
    const float c5 = c * c * c * c * c;
    return f0 + (1.0F - f0) * c5;
}
inline float ComputeSchlickFresnel(float ni, float nt, const v3& w, const v3& n)
{
    float costheta = dot(w, n);
    if (costheta < 0) { std::swap(ni, nt); costheta = -costheta; }
    float f0 = (ni - nt) / (ni + nt);
    f0 = f0 * f0;
    return ComputeSchlickFresnelWithF0AndCosTheta(f0, costheta);
}
inline v3 ComputeReflectVector(const v3& wi, const v3& n)
{
    v3 wo = wi - 2 * dot(wi, n) * n;
    return normalize(wo);
}

// ---- Diffuse: material/diffuse.h:86-137 ----------------------------------------------------
namespace Diffuse {
inline float ComputePDF(const v3& n, const v3& wo) { const float c = std::fabs(dot(n, wo)); return c / PI; }
inline v3 SampleDirection(const v3& n, float r1, float r2)
{
    const float costheta = std::sqrt(1 - r1);
    const float sintheta = std::sqrt(r1);
    const float phi = PI_2 * r2;
    const float cosphi = std::cos(phi);
    const float sinphi = std::sin(phi);
    v3 t, b;
    GetTangentCoordinate(n, t, b);
    v3 dir = t * sintheta * cosphi + b * sintheta * sinphi + n * costheta;
    return normalize(dir);
}
inline v3 ComputeBRDF() { return v3(1.0F) / PI; }
inline void sample(MaterialSampling* res, const v3& normal, CMJ* sampler)
{
    const float r1 = sampler->nextSample();
    const float r2 = sampler->nextSample();
    res->dir = SampleDirection(normal, r1, r2);
    res->pdf = ComputePDF(normal, res->dir);
    res->bsdf = ComputeBRDF();
}
}

// ---- Specular: material/specular.h:65-102, specular.cpp:36-48 -------------------------------
namespace Specular {
inline v3 ComputeBRDF(const v3& wo, const v3& n)
{
    const float c = dot(n, wo);
    const float bsdf = c == 0.0F ? 0.0F : 1.0F / c;
    return v3(bsdf);
}
inline void sample(MaterialSampling* res, const v3& normal, const v3& wi)
{
    res->dir = ComputeReflectVector(wi, normal);
    res->pdf = 1.0F;
    res->bsdf = ComputeBRDF(res->dir, normal);
}
}

// ---- GGX: material/ggx.cpp:74-274 -----------------------------------------------------------
namespace GGX {
inline float ComputeDistribution(const v3& m, const v3& n, float roughness)
{
    const float a = roughness;
    const float a2 = a * a;
    const float costheta = std::fabs(dot(m, n));
    const float cos2 = costheta * costheta;
    const float denom = (a2 - 1) * cos2 + 1.0f;
    const float denom2 = denom * denom;
    return denom > 0 ? a2 / (PI * denom2) : 0;
}
inline float Lambda(float roughness, const v3& w, const v3& n)
{
    const float alpha = roughness;
    const float cos_theta = std::fabs(dot(w, n));
    const float cos2 = cos_theta * cos_theta;
    const float sin2 = 1.0f - cos2;
    const float tan2 = sin2 / cos2;
    const float a2 = 1.0f / (alpha * alpha * tan2);
    return (-1.0f + std::sqrt(1.0f + 1.0f / a2)) / 2.0f;
}
inline float ComputeG2Smith(float roughness, const v3& view, const v3& light, const v3& n)
{
    const float lambda_wi = Lambda(roughness, view, n);
    const float lambda_wo = Lambda(roughness, light, n);
    return 1.0f / (1.0f + lambda_wi + lambda_wo);
}
inline float ComputePDFWithHalfVector(float roughness, const v3& n, const v3& m, const v3& wo)
{
    const float D = ComputeDistribution(m, n, roughness);
    const float costheta = std::fabs(dot(m, n));
    const float denom = 4 * std::fabs(dot(wo, m));
    return denom > 0 ? (D * costheta) / denom : 0;
}
inline float ComputePDF(float roughness, const v3& n, const v3& wi, const v3& wo)
{
    const v3 wh = normalize(-wi + wo);
    return ComputePDFWithHalfVector(roughness, n, wh, wo);
}
inline v3 SampleMicrosurfaceNormal(float roughness, const v3& n, float r1, float r2)
{
    const float a = roughness;
    float theta = std::atan(a * std::sqrt(r1 / (1 - r1)));
    theta = ((theta >= 0) ? theta : (theta + 2 * PI));
    const float phi = 2 * PI * r2;
    const float costheta = std::cos(theta);
    const float sintheta = std::sin(theta);
    const float cosphi = std::cos(phi);
    const float sinphi = std::sin(phi);
    v3 t, b;
    GetTangentCoordinate(n, t, b);
    v3 m = t * sintheta * cosphi + b * sintheta * sinphi + n * costheta;
    return normalize(m);
}
inline v3 SampleDirection(float r1, float r2, float roughness, const v3& wi, const v3& n)
{
    const v3 m = SampleMicrosurfaceNormal(roughness, n, r1, r2);
    return ComputeReflectVector(wi, m);
}
inline v3 ComputeBRDFWithHalfVector(float roughness, float ior, const v3& N, const v3& V, const v3& L, const v3& H)
{
    float NL = std::fabs(dot(N, L));
    float NV = std::fabs(dot(N, V));
    const float ni = 1.0F;
    const float nt = ior;
    const float D = ComputeDistribution(H, N, roughness);
    const float G = ComputeG2Smith(roughness, V, L, N);
    const float F = ComputeSchlickFresnel(ni, nt, L, H);
    const float denom = 4 * NL * NV;
    const float bsdf = denom > EPS ? F * G * D / denom : 0.0f;
    return v3(bsdf);
}
inline v3 ComputeBRDF(float roughness, float ior, const v3& n, const v3& wi, const v3& wo)
{
    const v3 V = -wi;
    const v3 L = wo;
    const v3 H = normalize(L + V);
    return ComputeBRDFWithHalfVector(roughness, ior, n, V, L, H);
}
inline float roughness_of(const Scene& ctxt, const atn_material_param& p, float u, float v)
{
    return sampleTexture(ctxt, p.roughnessMap, u, v, v4(p.u.standard.roughness)).x;
}
inline void sample(MaterialSampling* res, const Scene& ctxt, const atn_material_param& p,
    const v3& normal, const v3& wi, CMJ* sampler, float u, float v)
{
    const float rough = roughness_of(ctxt, p, u, v);
    const float r1 = sampler->nextSample();
    const float r2 = sampler->nextSample();
    res->dir = SampleDirection(r1, r2, rough, wi, normal);
    res->pdf = ComputePDF(rough, normal, wi, res->dir);
    res->bsdf = ComputeBRDF(rough, p.u.standard.ior, normal, wi, res->dir);
}
}

// ---- Disney: material/disney_brdf.cpp:38-555 ------------------------------------------------
namespace Disney {
enum { C_Diffuse = 0, C_Sheen = 1, C_Specular = 2, C_Clearcoat = 3, C_Num = 4 };  // disney_brdf.h:105-111

inline v3 ComputeCtint(const v3& base_color)
{
    const float Y = dot(base_color, v3(0.3F, 0.6F, 0.1F));
    return Y > 0 ? base_color / Y : v3(1.0F);
}
inline float SchlickFresnel(float u)
{
    const float m = clamp_(1.0F - u, 0.0F, 1.0F);
    const float m2 = m * m;
    return m2 * m2 * m;
}
inline v3 Diffuse_EvalBRDF(const v3& base_color, float roughness, float subsurface, const v3& V, const v3& L, const v3& N)
{
    const v3 H = normalize(V + L);
    const float LdotH = dot(L, H);
    const float NdotV = dot(V, N);
    const float NdotL = dot(L, N);
    const float FV = SchlickFresnel(dot(V, N));
    const float FL = SchlickFresnel(dot(L, N));
    const float Fd90 = 0.5F + 2 * LdotH * LdotH * roughness;
    float fd = mix(1.0F, Fd90, FL) * mix(1.0F, Fd90, FV);
    const float Fss90 = LdotH * LdotH * roughness;
    const float Fss = mix(1.0F, Fss90, FL) * mix(1.0F, Fss90, FV);
    const float ss = 1.25F * (Fss * (1.0F / (NdotL + NdotV) - 0.5F) + 0.5F);
    fd = mix(fd, ss, subsurface);
    return base_color / PI * fd;
}
inline v3 Sheen_EvalBRDF(const v3& base_color, float sheen, float sheen_tint, const v3& V, const v3& L)
{
    const v3 H = normalize(V + L);
    const v3 Ctint = ComputeCtint(base_color);
    const v3 Csheen = mix(v3(1.0F), Ctint, sheen_tint);
    const float FH = SchlickFresnel(dot(L, H));
    return sheen * Csheen * FH;
}
inline float Sheen_EvalPDF() { return 1 / (PI); }
inline float D_GTR1(float roughness, float NdotH)
{
    const float a = roughness;
    if (a >= 1) return 1 / PI;
    const float a2 = a * a;
    const float t = 1.0F + (a2 - 1.0F) * NdotH * NdotH;
    return (a2 - 1) / (PI * std::log(a2) * t);
}
inline v3 Clearcoat_EvalBRDF(float clearcoat, const v3& V, const v3& L)
{
    const v3 H = normalize(V + L);
    const float LdotH = dot(L, H);
    const float FH = SchlickFresnel(std::fabs(LdotH));
    const float F = mix(0.04F, 1.0F, FH);
    return v3(0.25F * clearcoat * F);
}
inline float Clearcoat_EvalPDF(float clearcoat_gloss, const v3& V, const v3& L, const v3& N)
{
    const v3 H = normalize(V + L);
    const float NdotH = dot(N, H);
    const float a_clearcoat = mix(0.1F, 0.001F, clearcoat_gloss);
    const float D = D_GTR1(a_clearcoat, NdotH);
    const float costheta = std::fabs(dot(H, N));
    const float denom = 4 * std::fabs(dot(L, H));
    return denom > 0 ? (D * costheta) / denom : 0;
}
inline v3 Clearcoat_SampleDirection(float r1, float r2, float clearcoat_gloss, const v3& V, const v3& N)
{
    const float a_clearcoat = mix(0.1F, 0.001F, clearcoat_gloss);
    const v3 m = GGX::SampleMicrosurfaceNormal(a_clearcoat, N, r1, r2);
    return ComputeReflectVector(-V, m);
}
inline v3 Specular_EvalBRDF(const v3& base_color, float roughness, float metallic, float specular,
    float specular_tint, const v3& V, const v3& L, const v3& N)
{
    const v3 H = normalize(V + L);
    const v3 Ctint = ComputeCtint(base_color);
    const v3 Cspec = mix(v3(1), Ctint, specular_tint);
    const v3 F_s0 = mix(0.08F * specular * Cspec, base_color, metallic);
    const v3 F = mix(F_s0, v3(1.0F), dot(L, H));
    const float D = GGX::ComputeDistribution(H, N, roughness);
    const float G = GGX::ComputeG2Smith(roughness, V, L, N);
    const float NdotL = std::fabs(dot(N, L));
    const float NdotV = std::fabs(dot(N, V));
    const float denom = 4 * NdotV * NdotL;
    return denom > 0 ? F * D * G / denom : v3(0.0F);
}
inline float Specular_EvalPDF(float roughness, const v3& V, const v3& L, const v3& N)
{
    const v3 H = normalize(V + L);
    return GGX::ComputePDFWithHalfVector(roughness, N, H, L);
}
inline void ComputeWeights(float w[C_Num], const v3& base_color, float metalic, float sheen, float specular, float clearcoat)
{
    const float lum = luminance(base_color.x, base_color.y, base_color.z);
    w[C_Diffuse] = lum * (1 - metalic);
    w[C_Sheen] = sheen * (1 - metalic);
    w[C_Specular] = mix(specular, 1.0F, metalic);
    w[C_Clearcoat] = 0.25F * clearcoat;
    float norm = 0.0F;
    for (int i = 0; i < C_Num; i++) norm += w[i];
    if (norm > 0) for (int i = 0; i < C_Num; i++) w[i] /= norm;
}
inline v3 base_of(const atn_material_param& m) { return v3(m.baseColor.x, m.baseColor.y, m.baseColor.z); }

inline float pdf(const atn_material_param& mtrl, const v3& n, const v3& wi, const v3& wo)   // :347-378
{
    const auto& s = mtrl.u.standard;
    float w[C_Num];
    ComputeWeights(w, base_of(mtrl), s.metallic, s.sheen, s.specular, s.clearcoat);
    const v3 V = -wi, L = wo, N = n;
    float p = 0.0F;
    p += w[C_Diffuse] * Diffuse::ComputePDF(N, L);
    p += w[C_Sheen] * Sheen_EvalPDF();
    p += w[C_Specular] * Specular_EvalPDF(s.roughness, V, L, N);
    p += w[C_Clearcoat] * Clearcoat_EvalPDF(s.clearcoatGloss, V, L, N);
    return p;
}
inline MaterialSampling bsdf(const atn_material_param& mtrl, const v3& n, const v3& wi, const v3& wo)  // :380-441
{
    const auto& s = mtrl.u.standard;
    const v3 base = base_of(mtrl);
    float w[C_Num];
    ComputeWeights(w, base, s.metallic, s.sheen, s.specular, s.clearcoat);
    const v3 V = -wi, N = n;
    float p = 0.0F;
    v3 d(0.0F), sh(0.0F), sp(0.0F), cc(0.0F);
    if (w[C_Diffuse] > 0.0F) {
        d = Diffuse_EvalBRDF(base, s.roughness, s.subsurface, V, wo, N);
        p += Diffuse::ComputePDF(N, wo) * w[C_Diffuse];
    }
    if (w[C_Sheen] > 0.0F) {
        sh = Sheen_EvalBRDF(base, s.sheen, s.sheenTint, V, wo);
        p += Sheen_EvalPDF() * w[C_Sheen];
    }
    if (w[C_Specular] > 0.0F) {
        sp = Specular_EvalBRDF(base, s.roughness, s.metallic, s.specular, s.specularTint, V, wo, N);
        p += Specular_EvalPDF(s.roughness, V, wo, N) * w[C_Specular];
    }
    if (w[C_Clearcoat] > 0.0F) {
        cc = Clearcoat_EvalBRDF(s.clearcoat, V, wo);
        p += Clearcoat_EvalPDF(s.roughness, V, wo, N) * w[C_Clearcoat];   // quirk: roughness, :431
    }
    MaterialSampling r;
    r.bsdf = (1 - s.metallic) * (d + sh) + sp + cc;
    r.pdf = p;
    return r;
}
inline void sample(MaterialSampling& result, const atn_material_param& mtrl, const v3& n, const v3& wi, CMJ* sampler)  // :443-555
{
    const float r1 = sampler->nextSample();
    const float r2 = sampler->nextSample();
    const float r3 = sampler->nextSample();
    const auto& s = mtrl.u.standard;
    const v3 base = base_of(mtrl);
    float w[C_Num];
    ComputeWeights(w, base, s.metallic, s.sheen, s.specular, s.clearcoat);
    float cdf[C_Num];
    cdf[C_Diffuse] = w[C_Diffuse];
    cdf[C_Sheen] = cdf[C_Diffuse] + w[C_Sheen];
    cdf[C_Specular] = cdf[C_Sheen] + w[C_Specular];
    cdf[C_Clearcoat] = cdf[C_Specular] + w[C_Clearcoat];
    const v3 V = -wi, N = n;
    v3 wo;
    float p = 0;
    v3 d(0.0F), sh(0.0F), sp(0.0F), cc(0.0F);
    if (r3 < cdf[C_Diffuse]) {
        wo = Diffuse::SampleDirection(N, r1, r2);
        d = Diffuse_EvalBRDF(base, s.roughness, s.subsurface, V, wo, N);
        p = Diffuse::ComputePDF(N, wo);
        p *= w[C_Diffuse];
        w[C_Diffuse] = 0.0F;
    }
    else if (r3 < cdf[C_Sheen]) {
        wo = Diffuse::SampleDirection(N, r1, r2);
        sh = Sheen_EvalBRDF(base, s.sheen, s.sheenTint, V, wo);
        p = Sheen_EvalPDF();
        p *= w[C_Sheen];
        w[C_Sheen] = 0.0F;
    }
    else if (r3 < cdf[C_Specular]) {
        wo = GGX::SampleDirection(r1, r2, s.roughness, -V, N);
        sp = Specular_EvalBRDF(base, s.roughness, s.metallic, s.specular, s.specularTint, V, wo, N);
        p = Specular_EvalPDF(s.roughness, V, wo, N);
        p *= w[C_Specular];
        w[C_Specular] = 0.0F;
    }
    else {
        wo = Clearcoat_SampleDirection(r1, r2, s.clearcoatGloss, V, N);
        cc = Clearcoat_EvalBRDF(s.clearcoat, V, wo);
        p = Clearcoat_EvalPDF(s.roughness, V, wo, N);                      // quirk: roughness, :517
        p *= w[C_Clearcoat];
        w[C_Clearcoat] = 0.0F;
    }
    if (w[C_Diffuse] > 0.0F) {
        d = Diffuse_EvalBRDF(base, s.roughness, s.subsurface, V, wo, N);
        p += w[C_Diffuse] * Diffuse::ComputePDF(N, wo);
    }
    if (w[C_Sheen] > 0.0F) {
        sh = Sheen_EvalBRDF(base, s.sheen, s.sheenTint, V, wo);
        p += w[C_Sheen] * Sheen_EvalPDF();
    }
    if (w[C_Specular] > 0.0F) {
        sp = Specular_EvalBRDF(base, s.roughness, s.metallic, s.specular, s.specularTint, V, wo, N);
        p += w[C_Specular] * Specular_EvalPDF(s.roughness, V, wo, N);
    }
    if (w[C_Clearcoat] > 0.0F) {
        cc = Clearcoat_EvalBRDF(s.clearcoat, V, wo);
        p += w[C_Clearcoat] * Clearcoat_EvalPDF(s.roughness, V, wo, N);    // quirk: roughness, :548
    }
    result.pdf = p;
    result.bsdf = (1 - s.metallic) * (d + sh) + sp + cc;
    result.dir = wo;
}
}

// ---- Refraction: material/refraction.cpp:62-161, material.h:549-576 (ComputeRefractVector) ------
namespace Refraction {
inline v3 ComputeRefractVector(float ni, float nt, const v3& wi, const v3& n)
{
    const v3 w = -wi;
    v3 N = n;
    float costheta = dot(w, n);
    if (costheta < 0.0F) { std::swap(ni, nt); costheta = -costheta; N = -N; }
    const float sintheta_2 = 1.0F - costheta * costheta;
    const float ni_nt = ni / nt;
    const float ni_nt_2 = ni_nt * ni_nt;
    v3 wo = (ni_nt * costheta - std::sqrt(1.0F - ni_nt_2 * sintheta_2)) * N - ni_nt * w;
    return normalize(wo);
}
inline v3 ComputeBRDF(float ni, float nt, const v3& wo, const v3& n, float fresnel_transmittance)
{
    const float c = std::abs(dot(wo, n));
    const float nt_ni = nt / ni;
    const float bsdf = c == 0.0F ? 0.0F : (nt_ni * nt_ni) * fresnel_transmittance / c;
    return v3(bsdf);
}
inline void sample(MaterialSampling& result, CMJ* sampler, const atn_material_param& param, const v3& n, const v3& wi)
{
    float ni = 1.0F;
    float nt = param.u.standard.ior;
    const v3 V = -wi;
    v3 N = n;
    const bool is_enter = dot(V, N) >= 0.0F;
    if (!is_enter) { N = -n; std::swap(ni, nt); }
    const float ni_nt = ni / nt;
    const float cos_i = dot(V, N);
    const float cos_t_2 = 1.0F - (ni_nt * ni_nt * (1.0F - cos_i * cos_i));
    if (cos_t_2 < 0.0F) { std::swap(ni, nt); N = -N; }
    v3 wo = ComputeRefractVector(ni, nt, wi, N);
    const float F = ComputeSchlickFresnel(ni, nt, wo, N);
    const float R = F;
    const float T = 1 - R;
    if (param.isIdealRefraction) {
        result.pdf = 1.0F; result.dir = wo; result.bsdf = ComputeBRDF(ni, nt, wo, N, T);
        return;
    }
    const float prob = 0.25F + 0.5F * R;
    const float u = sampler->nextSample();
    if (u < prob) {
        wo = ComputeReflectVector(wi, N);
        const float c = std::abs(dot(wo, N));
        const float bsdf = c == 0.0F ? 0.0F : R / c;
        result.pdf = prob; result.dir = wo; result.bsdf = v3(bsdf);
    }
    else {
        result.pdf = 1.0F - prob; result.dir = wo; result.bsdf = ComputeBRDF(ni, nt, wo, N, T);
    }
}
} // namespace Refraction

// ---- MicrofacetBeckman: material/beckman.cpp:103-255 ---------------------------------------------
namespace Beckman {
inline float ComputeDistribution(const v3& m, const v3& n, float roughness)
{
    const float costheta = std::abs(dot(m, n));
    if (costheta <= 0) return 0;
    const float cos2 = costheta * costheta;
    const float cos4 = cos2 * cos2;
    const float sintheta = std::sqrt(1 - cos2);
    const float tantheta = sintheta / costheta;
    const float tan2 = tantheta * tantheta;
    const float a = roughness;
    const float a2 = a * a;
    float D = 1.0f / (PI * a2 * cos4);
    D *= std::exp(-tan2 / a2);
    return D;
}
inline float ComputePDF(float roughness, const v3& n, const v3& wi, const v3& wo)
{
    const v3 wh = normalize(-wi + wo);
    const float costheta = std::abs(dot(wh, n));
    const float D = ComputeDistribution(wh, n, roughness);
    const float denom = 4 * std::abs(dot(wo, wh));
    return denom > 0 ? (D * costheta) / denom : 0;
}
inline v3 SampleMicrosurfaceNormal(float roughness, const v3& n, float r1, float r2)
{
    const float a = roughness;
    const float a2 = a * a;
    const float theta = std::atan(std::sqrt(-a2 * std::log(1.0F - r1 * 0.99F)));
    const float phi = PI_2 * r2;
    const float costheta = std::cos(theta);
    const float sintheta = std::sin(theta);
    const float cosphi = std::cos(phi);
    const float sinphi = std::sin(phi);
    v3 t, b;
    GetTangentCoordinate(n, t, b);
    v3 m = t * sintheta * cosphi + b * sintheta * sinphi + n * costheta;
    return normalize(m);
}
inline float ComputeG1(float roughness, const v3& v, const v3& n)
{
    const float costheta = saturate_(std::abs(dot(v, n)));
    const float sintheta = std::sqrt(1.0F - costheta * costheta);
    const float tantheta = sintheta / costheta;
    const float a = 1.0F / (roughness * tantheta);
    const float a2 = a * a;
    if (a < 1.6F) return (3.535F * a + 2.181F * a2) / (1.0F + 2.276F * a + 2.577F * a2);
    return 1.0F;
}
inline v3 ComputeBRDF(float roughness, float ior, const v3& n, const v3& wi, const v3& wo)
{
    const v3 V = -wi, L = wo, N = n;
    const v3 H = normalize(L + V);
    const float NL = std::abs(dot(N, L));
    const float NV = std::abs(dot(N, V));
    const float D = ComputeDistribution(H, N, roughness);
    const float G2 = ComputeG1(roughness, V, N) * ComputeG1(roughness, L, N);
    const float F = ComputeSchlickFresnel(1.0F, ior, L, H);
    const float denom = 4 * NL * NV;
    const float bsdf = denom > EPS ? F * G2 * D / denom : 0.0f;
    return v3(bsdf);
}
inline void sample(MaterialSampling* res, const Scene& ctxt, const atn_material_param& p, const v3& normal, const v3& wi,
    CMJ* sampler, float u, float v)
{
    const float roughness = GGX::roughness_of(ctxt, p, u, v);     // same sampleTexture(roughnessMap, vec4(roughness)).r
    const float r1 = sampler->nextSample();
    const float r2 = sampler->nextSample();
    const v3 m = SampleMicrosurfaceNormal(roughness, normal, r1, r2);
    res->dir = ComputeReflectVector(wi, m);
    res->pdf = ComputePDF(roughness, normal, wi, res->dir);
    res->bsdf = ComputeBRDF(roughness, p.u.standard.ior, normal, wi, res->dir);
}
} // namespace Beckman

// ---- Retroreflective: material/retroreflective.cpp:17-612 -------------------------------------------------
namespace Retroreflective {
struct EraEntry { float deg, area; };
// {angle in degrees, effective retroreflective area}: the measured table the model interpolates (retroreflective.cpp:59-161)
static const EraEntry ERATable[101] = {
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
inline float GetEffectiveRetroreflectiveArea(const v3& into_prismatic_sheet_dir, const v3& surface_normal)
{
    const float c = dot(into_prismatic_sheet_dir, -surface_normal);
    if (c < 0.0F) return 0.0F;
    const float theta = std::acos(c);
    const float Step = Deg2Rad(90.00000F) / (101 - 1);
    const size_t idx = static_cast<size_t>(theta / Step);
    float a = 0.0F, b = 0.0F, t = 0.0F;
    if (idx >= 101) return 0.0F;
    const float d = Deg2Rad(ERATable[idx].deg);
    t = fmin_(1.0F, std::abs(d - theta) / Step);
    a = ERATable[idx].area;
    if (idx < 101 - 1) b = ERATable[idx + 1].area;
    return a * (1 - t) + b * t;
}
inline float ComputeRoughness(float roughness, float ni, float nt, const v3& wi, const v3& wn)
{
    const v3 uo = -wi;
    const v3 ut = Refraction::ComputeRefractVector(ni, nt, wi, wn);
    const float n = nt / ni;
    const float J1_denom = dot(-wi, wn) + n * dot(ut, wn);
    const float J1 = J1_denom > 0 ? std::abs(dot(uo, wn)) / sqr(J1_denom) : 0.0F;
    const float J2_denom = -n * dot(ut, wn) + dot(uo, wn);
    const float J2 = J2_denom > 0 ? std::abs(dot(uo, wn)) / sqr(J2_denom) : 0.0F;
    const float a = roughness;
    const float a2 = a * a;
    float a0 = (J1 > 0 ? a2 / J1 : 0.0F) + (J2 > 0 ? a2 / J2 : 0.0F);
    a0 = std::sqrt(a0);
    return a0;
}
inline v3 RR_EvalBRDF(float roughness, float ior, const v3& wn, const v3& wi, const v3& wo, float* used_E, float* used_F)
{
    const float ni = 1.0F, nt = ior;
    const v3 uo = -wi;
    const v3 ut = Refraction::ComputeRefractVector(ni, nt, wi, wn);
    const float E = GetEffectiveRetroreflectiveArea(ut, wn);
    *used_E = E;
    const float a = ComputeRoughness(roughness, ni, nt, wi, wn);
    const float D = Beckman::ComputeDistribution(wo, uo, a);
    float F = (1.0F - ComputeSchlickFresnel(ni, nt, -wi, wn));
    F *= (1.0F - ComputeSchlickFresnel(ni, nt, wo, wn));
    *used_F = F;
    float G = Beckman::ComputeG1(roughness, wi, ut);
    G *= Beckman::ComputeG1(roughness, ut, wo);
    const float c = std::abs(dot(wo, wn));
    const float brdf = c > 0 ? E * F * G * D / c : 0.0F;
    return v3(brdf);
}
inline float RR_EvalPDF(float roughness, float ni, float nt, const v3& wn, const v3& wi, const v3& wo)
{
    const v3 uo = -wi;
    const float a = ComputeRoughness(roughness, ni, nt, wi, wn);
    const float D = Beckman::ComputeDistribution(wo, uo, a);
    return D * std::abs(dot(uo, wo));
}
inline v3 RR_SampleDirection(float r1, float r2, float roughness, float ni, float nt, const v3& wi, const v3& wn)
{
    const v3 uo = -wi;
    const float a = ComputeRoughness(roughness, ni, nt, wi, wn);
    const float a2 = a * a;
    const float theta = std::atan(std::sqrt(-a2 * std::log(1.0F - r1 * 0.99F)));
    const float phi = PI_2 * r2;
    v3 t, b;
    GetTangentCoordinate(uo, t, b);
    const float costheta = std::cos(theta), sintheta = std::sin(theta);
    const float cosphi = std::cos(phi), sinphi = std::sin(phi);
    v3 wo = t * sintheta * cosphi + b * sintheta * sinphi + uo * costheta;
    return normalize(wo);
}
inline v3 D_EvalBRDF(float E, float F, float ni, float nt)
{
    constexpr float kd = 1.0F;
    const float brdf_0 = F * (1.0F - E) * sqr(ni / nt) * (kd / PI);
    float f0 = (ni - nt) / (ni + nt);
    f0 = f0 * f0;
    const float Fd = (1.0F - f0) * (-160.0F / 21.0F);
    return v3(brdf_0 / (1.0F - kd * Fd));
}
inline float D_EvalPDF(const v3& n, const v3& wo) { return 1.0F / (1.0F - Diffuse::ComputePDF(n, wo)); }
inline void ComputeWeights(float w[3], float ni, float nt, const v3& wi, const v3& n)
{
    const float F = ComputeSchlickFresnel(ni, nt, -wi, n);
    w[0] = F;
    const v3 ut = Refraction::ComputeRefractVector(ni, nt, wi, n);
    const float E = GetEffectiveRetroreflectiveArea(ut, n);
    w[1] = (1 - F) * E;
    w[2] = (1 - F) * (1 - E);
    float norm = 0.0F;
    for (int i = 0; i < 3; i++) norm += w[i];
    for (int i = 0; i < 3; i++) w[i] /= norm;
}
inline void EF_without_lobe(float ni, float nt, const v3& wi, const v3& n, const v3& wo, float& used_E, float& used_F)
{
    const v3 ut = Refraction::ComputeRefractVector(ni, nt, wi, n);
    used_E = GetEffectiveRetroreflectiveArea(ut, n);
    float F = (1.0F - ComputeSchlickFresnel(ni, nt, -wi, n));
    F *= (1.0F - ComputeSchlickFresnel(ni, nt, wo, n));
    used_F = F;
}
inline float pdf(const atn_material_param& param, const v3& n, const v3& wi, const v3& wo)
{
    const float roughness = param.u.standard.roughness, ni = 1.0F, nt = param.u.standard.ior;
    float w[3];
    ComputeWeights(w, ni, nt, wi, n);
    float p = 0.0F;
    if (w[0] > 0.0F) p += w[0] * Beckman::ComputePDF(roughness, n, wi, wo);
    if (w[1] > 0.0F) p += w[1] * RR_EvalPDF(roughness, ni, nt, n, wi, wo);
    if (w[2] > 0.0F) p += w[2] * D_EvalPDF(n, wo);
    return p;
}
inline MaterialSampling bsdf(const atn_material_param& param, const v3& n, const v3& wi, const v3& wo)
{
    const float roughness = param.u.standard.roughness, ior = param.u.standard.ior, ni = 1.0F, nt = ior;
    float w[3];
    ComputeWeights(w, ni, nt, wi, n);
    v3 f_r(0.0F), f_rr(0.0F), f_d(0.0F);
    float p = 0.0F;
    if (w[0] > 0.0F) {
        f_r = Beckman::ComputeBRDF(roughness, ior, n, wi, wo);
        p += Beckman::ComputePDF(roughness, n, wi, wo) * w[0];
    }
    float used_E = 0.0F, used_F = 0.0F;
    if (w[1] > 0.0F) {
        f_rr = RR_EvalBRDF(roughness, ior, n, wi, wo, &used_E, &used_F);
        p += RR_EvalPDF(roughness, ni, nt, n, wi, wo) * w[1];
    }
    else EF_without_lobe(ni, nt, wi, n, wo, used_E, used_F);
    if (w[2] > 0.0F) {
        f_d = D_EvalBRDF(used_E, used_F, ni, nt);
        p += D_EvalPDF(n, wo) * w[2];
    }
    MaterialSampling result;
    result.bsdf = f_r + f_rr + f_d;
    result.pdf = p;
    return result;
}
inline void sample(MaterialSampling& result, const atn_material_param& param, const v3& n, const v3& wi, CMJ* sampler)
{
    const float r1 = sampler->nextSample();
    const float r2 = sampler->nextSample();
    const float r3 = sampler->nextSample();
    const float roughness = param.u.standard.roughness, ior = param.u.standard.ior, ni = 1.0F, nt = ior;
    float w[3];
    ComputeWeights(w, ni, nt, wi, n);
    const float cdf0 = w[0], cdf1 = cdf0 + w[1];
    v3 f_r(0.0F), f_rr(0.0F), f_d(0.0F);
    float p = 0.0F;
    v3 wo;      // the reference reads it unset in the diffuse branch below (retroreflective.cpp:559-565): (0, 0, 0) here
    float used_E = 0.0F, used_F = 0.0F;
    if (r3 < cdf0) {
        wo = ComputeReflectVector(wi, Beckman::SampleMicrosurfaceNormal(roughness, n, r1, r2));
        f_r = Beckman::ComputeBRDF(roughness, ior, n, wi, wo);
        p += Beckman::ComputePDF(roughness, n, wi, wo) * w[0];
        w[0] = 0.0F;
    }
    else if (r3 < cdf1) {
        wo = RR_SampleDirection(r1, r2, roughness, ni, nt, wi, n);
        f_rr = RR_EvalBRDF(roughness, ior, n, wi, wo, &used_E, &used_F);
        p += RR_EvalPDF(roughness, ni, nt, n, wi, wo) * w[1];
        w[1] = 0.0F;
    }
    else {
        EF_without_lobe(ni, nt, wi, n, wo, used_E, used_F);
        wo = Diffuse::SampleDirection(n, r1, r2);
        f_d = D_EvalBRDF(used_E, used_F, ni, nt);
        p += D_EvalPDF(n, wo) * w[2];
        w[2] = 0.0F;
    }
    if (w[0] > 0.0F) {
        f_r = Beckman::ComputeBRDF(roughness, ior, n, wi, wo);
        p += Beckman::ComputePDF(roughness, n, wi, wo) * w[0];
    }
    if (w[1] > 0.0F) {
        f_rr = RR_EvalBRDF(roughness, ior, n, wi, wo, &used_E, &used_F);
        p += RR_EvalPDF(roughness, ni, nt, n, wi, wo) * w[1];
    }
    if (w[2] > 0.0F) {
        EF_without_lobe(ni, nt, wi, n, wo, used_E, used_F);
        f_d = D_EvalBRDF(used_E, used_F, ni, nt);
        p += D_EvalPDF(n, wo) * w[2];
    }
    result.pdf = p;
    result.bsdf = f_r + f_rr + f_d;
    result.dir = wo;
}
} // namespace Retroreflective

// ---- CarPaint: material/car_paint.cpp:14-236, FlakesNormal.cpp:5-185, FlakesNormal.h:20-52, material.h:445-467 ------
namespace CarPaint {
struct Param {      // CarPaintMaterialParameter, material.h:163-176 (the union member next to `standard`)
    v3 clearcoat_color; float clearcoat_ior;
    v3 flakes_color; float clearcoat_roughness;
    v3 diffuse_color; float flake_scale;
    float flake_size, flake_size_variance, flake_normal_orientation, flake_color_multiplier;
};
inline Param param_of(const atn_material_param& m)
{
    const float* c = m.u.carpaint;
    Param p;
    p.clearcoat_color = v3(c[0], c[1], c[2]); p.clearcoat_ior = c[3];
    p.flakes_color = v3(c[4], c[5], c[6]); p.clearcoat_roughness = c[7];
    p.diffuse_color = v3(c[8], c[9], c[10]); p.flake_scale = c[11];
    p.flake_size = c[12]; p.flake_size_variance = c[13]; p.flake_normal_orientation = c[14]; p.flake_color_multiplier = c[15];
    return p;
}
inline float computeFresnel(float ni, float nt, const v3& wi, const v3& normal)
{
    float cosi = dot(normal, wi);
    if (cosi < 0) { std::swap(ni, nt); cosi = -cosi; }
    const float nnt = ni / nt;
    const float sini2 = float(1.0) - cosi * cosi;
    const float sint2 = nnt * nnt * sini2;
    const float cost = std::sqrt(fmax_(float(0.0), float(1.0) - sint2));
    const float rp = (nt * cosi - ni * cost) / (nt * cosi + ni * cost);
    const float rs = (ni * cosi - nt * cost) / (ni * cosi + nt * cost);
    return (rp * rp + rs * rs) * float(0.5);
}
inline float computeFlakeDensity(float flake_size, float flakeMapAspect)
{
    float aspect = float(1) / flakeMapAspect;
    float D = PI * flake_size * flake_size * aspect;
    D = fmin_(D, float(1));
    return D;
}
inline float bits_to_01(uint32_t bits) { uint32_t div = 0xffffffff; return bits * (1.0f / float(div)); }
inline uint32_t rotl32(uint32_t var, uint32_t hops) { return (var << hops) | (var >> (32 - hops)); }
inline void bjmix(uint32_t& a, uint32_t& b, uint32_t& c)
{
    a -= c;  a ^= rotl32(c, 4);  c += b;
    b -= a;  b ^= rotl32(a, 6);  a += c;
    c -= b;  c ^= rotl32(b, 8);  b += a;
    a -= c;  a ^= rotl32(c, 16);  c += b;
    b -= a;  b ^= rotl32(a, 19);  a += c;
    c -= b;  c ^= rotl32(b, 4);  b += a;
}
inline uint32_t bjfinal(uint32_t a, uint32_t b, uint32_t c)
{
    c ^= b; c -= rotl32(b, 14);
    a ^= c; a -= rotl32(c, 11);
    b ^= a; b -= rotl32(a, 25);
    c ^= b; c -= rotl32(b, 16);
    a ^= c; a -= rotl32(c, 4);
    b ^= a; b -= rotl32(a, 14);
    c ^= b; c -= rotl32(b, 24);
    return c;
}
// `(uint32_t)k[i]` of a NEGATIVE float is undefined in C++; the reference's x86-64 build converts through a 64-bit
// integer and keeps the low 32 bits.  That behaviour is spelled out here and on the device.
inline uint32_t f2u_wrap(float f) { return (uint32_t)(int64_t)f; }
inline uint32_t inthash(const float k[4])
{
    uint32_t len = 4;
    uint32_t a = 0xdeadbeef + (len << 2) + 13;
    uint32_t b = 0xdeadbeef + (len << 2) + 13;
    uint32_t c = 0xdeadbeef + (len << 2) + 13;
    a += f2u_wrap(k[0]);
    b += f2u_wrap(k[1]);
    c += f2u_wrap(k[2]);
    bjmix(a, b, c);
    a += f2u_wrap(k[3]);
    c = bjfinal(a, b, c);
    return c;
}
inline v3 cellnoise(const v3& p)
{
    float iv[4] = { std::floor(p.x), std::floor(p.y), std::floor(p.z), 0.0F };
    v3 result;
    iv[3] = 0; result.x = bits_to_01(inthash(iv));
    iv[3] = 1; result.y = bits_to_01(inthash(iv));
    iv[3] = 2; result.z = bits_to_01(inthash(iv));
    return result;
}
inline v4 FlakesNormal_gen(float u, float v, float flake_scale, float flake_size, float flake_size_variance, float flake_normal_orientation)
{
    float safe_flake_size_variance = clamp_(flake_size_variance, float(0.1), float(1.0));
    const v3 cellCenters[9] = {
        v3(0.5, 0.5, 0.0), v3(1.5, 0.5, 0.0), v3(1.5, 1.5, 0.0), v3(0.5, 1.5, 0.0), v3(-0.5, 1.5, 0.0),
        v3(-0.5, 0.5, 0.0), v3(-0.5, -0.5, 0.0), v3(0.5, -0.5, 0.0), v3(1.5, -0.5, 0.0)
    };
    v3 position(u, v, 0.0);
    position = flake_scale * position;
    v3 base(std::floor(position.x), std::floor(position.y), std::floor(position.z));
    v3 nearestCell(0.0, 0.0, 1.0);
    int32_t nearestCellIndex = -1;
    for (int32_t cellIndex = 0; cellIndex < 9; ++cellIndex) {
        v3 cellCenter = base + cellCenters[cellIndex];
        v3 centerOffset = cellnoise(cellCenter) * float(2.0) - v3(float(1.0));
        centerOffset.z *= safe_flake_size_variance;
        centerOffset = normalize(centerOffset);
        cellCenter = cellCenter + float(0.5) * centerOffset;
        float cellDistance = length(position - cellCenter);      // glm::distance
        if (cellDistance < flake_size && cellCenter.z < nearestCell.z) {
            nearestCell = cellCenter;
            nearestCellIndex = cellIndex;
        }
    }
    v3 result(0.5, 0.5, 1.0);
    float alpha = 0.0;
    v3 I(0, 0, 1);
    if (nearestCellIndex != -1) {
        v3 randomNormal = cellnoise(base + cellCenters[nearestCellIndex] + v3(0.0, 0.0, 1.5));
        randomNormal = float(2.0) * randomNormal - v3(float(1.0));
        randomNormal = dot(randomNormal, I) < 0.0F ? randomNormal : -randomNormal;     // glm::faceforward(N, I, Nref = N)
        randomNormal = normalize(mix(randomNormal, v3(0.0, 0.0, 1.0), flake_normal_orientation));
        result = randomNormal;
        alpha = 1.0;
    }
    return v4(result.x, result.y, result.z, alpha);
}
inline float pdf(const atn_material_param& m, const v3& normal, const v3& wi, const v3& wo)
{
    const Param p = param_of(m);
    const v3 V = -wi, N = normal;
    float fresnel = computeFresnel(float(1), p.clearcoat_ior, V, N);
    float beckman_pdf = Beckman::ComputePDF(p.clearcoat_roughness, N, wi, wo);
    float flakes_beckman_pdf = Beckman::ComputePDF(float(1), N, wi, wo);
    float flakes_density = computeFlakeDensity(p.flake_size, float(1));
    float diffuse_pdf = Diffuse::ComputePDF(N, wo);
    float r = fresnel * beckman_pdf + (float(1) - fresnel) * (flakes_density * flakes_beckman_pdf + (1 - flakes_density) * diffuse_pdf);
    return clamp_(r, float(0), float(1));
}
inline v3 sampleDirection(const atn_material_param& m, const v3& normal, const v3& wi, CMJ* sampler, float pre_sampled_r)
{
    const Param p = param_of(m);
    const v3 V = -wi, N = normal;
    float r0 = pre_sampled_r;
    float r1 = sampler->nextSample();
    float fresnel = computeFresnel(float(1), p.clearcoat_ior, V, N);
    float flakes_density = computeFlakeDensity(p.flake_size, float(1));
    v3 dir;
    if (r0 < fresnel) {
        r0 /= fresnel;
        dir = ComputeReflectVector(wi, Beckman::SampleMicrosurfaceNormal(p.clearcoat_roughness, N, r0, r1));
    }
    else {
        r0 -= fresnel;
        r0 /= (float(1) - fresnel);
        if (r1 < flakes_density) {
            r1 /= flakes_density;
            dir = ComputeReflectVector(wi, Beckman::SampleMicrosurfaceNormal(float(1), N, r0, r1));
        }
        else {
            r1 -= flakes_density;
            r1 /= (float(1) - flakes_density);
            dir = Diffuse::SampleDirection(N, r0, r1);
        }
    }
    return dir;
}
inline v3 bsdf(const Scene& ctxt, const atn_material_param& m, const v3& normal, const v3& wi, const v3& wo, float u, float v, float pre_sampled_r)
{
    const Param p = param_of(m);
    const v3 albedo = sampleTexture(ctxt, m.albedoMap, u, v, v4(float(1))).xyz();
    const v3 V = -wi, N = normal;
    float fresnel = computeFresnel(float(1), p.clearcoat_ior, V, N);
    v3 r;
    if (pre_sampled_r < fresnel) {
        r = Beckman::ComputeBRDF(p.clearcoat_roughness, p.clearcoat_ior, N, wi, wo);
        r = r * p.clearcoat_color;
    }
    else {
        const bool is_on_flakes = FlakesNormal_gen(u, v, p.flake_scale, p.flake_size, p.flake_size_variance, p.flake_normal_orientation).w > float(0);
        if (is_on_flakes) {
            r = Beckman::ComputeBRDF(float(1), float(10), N, wi, wo);
            r = r * (p.flakes_color * p.flake_color_multiplier);
        }
        else {
            r = p.diffuse_color / PI;
        }
    }
    return albedo * r;
}
inline float applyNormalMap(const atn_material_param& m, const v3& orgNml, v3& newNml, float u, float v, const v3& wi, CMJ* sampler)
{
    const Param p = param_of(m);
    const v3 V = -wi;
    const v3 N = normalize(orgNml);
    float r0 = sampler->nextSample();
    float fresnel = computeFresnel(float(1), p.clearcoat_ior, V, N);
    if (r0 < fresnel) {
        newNml = N;
    }
    else {
        v4 flakes_nml = FlakesNormal_gen(u, v, p.flake_scale, p.flake_size, p.flake_size_variance, p.flake_normal_orientation);
        if (flakes_nml.w > float(0)) {
            // applyTangentSpaceCoord, car_paint.cpp:14-23
            v3 n = normalize(orgNml);
            v3 t, b;
            GetTangentCoordinate(n, t, b);
            newNml = flakes_nml.z * n + flakes_nml.x * t + flakes_nml.y * b;
            newNml = normalize(newNml);
        }
        else {
            newNml = N;
        }
    }
    return r0;
}
} // namespace CarPaint

// material::applyNormal, material_impl.h:208-230
inline float applyNormal(const Scene& ctxt, const atn_material_param& mtrl, const v3& orgNml, v3& newNml, float u, float v,
    const v3& wi, CMJ* sampler)
{
    if (mtrl.type == ATN_MTRL_CARPAINT) return CarPaint::applyNormalMap(mtrl, orgNml, newNml, u, v, wi, sampler);
    applyNormalMap(ctxt, mtrl.normalMap, orgNml, newNml, u, v);
    return float(-1);
}

// ---- OrenNayar: material/oren_nayar.cpp:8-140 ------------------------------------------------------
namespace OrenNayar {
inline float pdf(const v3& normal, const v3& wo)
{
    const float NL = dot(normal, wo);
    return NL > 0 ? NL / PI : 0.0F;
}
inline v3 computeBsdf(float roughness, const v3& normal, const v3& wi, const v3& wo)
{
    const float NL = dot(normal, wo);
    const float NV = dot(normal, -wi);
    const float a = roughness;
    const float a2 = a * a;
    const float A = float(1) - float(0.5) * (a2 / (a2 + float(0.33)));
    const float B = float(0.45) * (a2 / (a2 + float(0.09)));
    const float LV = dot(wo, -wi);
    const float s = LV - NL * NV;
    const float t = s <= 0 ? float(1) : s / std::max(NL, NV);
    const float bsdf = (1.0F / PI) * (A + B * std::max(float(0), s / t));
    return v3(bsdf);
}
inline void sample(MaterialSampling* res, const Scene& ctxt, const atn_material_param& p, const v3& normal, const v3& wi,
    CMJ* sampler, float u, float v)
{
    const float r1 = sampler->nextSample();
    const float r2 = sampler->nextSample();
    res->dir = Diffuse::SampleDirection(normal, r1, r2);
    res->pdf = pdf(normal, res->dir);
    res->bsdf = computeBsdf(GGX::roughness_of(ctxt, p, u, v), normal, wi, res->dir);
}
} // namespace OrenNayar

// ---- MicrofacetVelvet: material/velvet.cpp:57-212 ----------------------------------------------------
namespace Velvet {
inline float ComputeDistribution(const v3& m, const v3& n, float roughness)
{
    const float cos_theta = std::abs(dot(m, n));
    const float inv_r = 1.0F / roughness;
    const float sin_theta = std::sqrt(saturate_(1 - cos_theta * cos_theta));
    return ((2.0F + inv_r) * std::pow(sin_theta, inv_r)) / (PI_2);
}
inline float InterpolateVelvetParam(int idx, float interp_factor)
{
    constexpr float p0[] = { 25.3245F, 3.32435F, 0.16801F, -1.27393F, -4.85967F };
    constexpr float p1[] = { 21.5473F, 3.82987F, 0.19823F, -1.97760F, -4.32054F };
    // as written in the reference (velvet.cpp:82): "+ p1", not "* p1"
    return interp_factor * p0[idx] + (1 - interp_factor) + p1[idx];
}
inline float ComputeVelvetLForLambda(float x, float roughness)
{
    const float interp_factor = std::pow(float(1) - roughness, float(2));
    const float a = InterpolateVelvetParam(0, interp_factor);
    const float b = InterpolateVelvetParam(1, interp_factor);
    const float c = InterpolateVelvetParam(2, interp_factor);
    const float d = InterpolateVelvetParam(3, interp_factor);
    const float e = InterpolateVelvetParam(4, interp_factor);
    return a / (1 + b * std::pow(x, c)) + d * x + e;
}
inline float ComputeVelvetLambda(float roughness, const v3& w, const v3& m)
{
    const float cos_theta = saturate_(std::abs(dot(w, m)));
    if (cos_theta < 0.5F) return std::exp(ComputeVelvetLForLambda(cos_theta, roughness));
    return std::exp(2.0F * ComputeVelvetLForLambda(0.5F, roughness) - ComputeVelvetLForLambda(1 - cos_theta, roughness));
}
inline float ComputeShadowingMaskingFunction(float roughness, const v3& view, const v3& light, const v3& n)
{
    float lambda_wi = ComputeVelvetLambda(roughness, view, n);
    const float lambda_wo = ComputeVelvetLambda(roughness, light, n);
    const float cos_theta_wi = saturate_(std::abs(dot(view, n)));
    lambda_wi = std::pow(lambda_wi, 1.0F + 2.0F * std::pow(1.0F - cos_theta_wi, 8.0F));
    return 1.0F / (1.0F + lambda_wi + lambda_wo);
}
inline v3 ComputeBRDF(float roughness, const v3& n, const v3& wi, const v3& wo)
{
    const v3 V = -wi, L = wo, N = n;
    const v3 H = normalize(L + V);
    const float NL = std::abs(dot(N, L));
    const float NV = std::abs(dot(N, V));
    const float D = ComputeDistribution(H, N, roughness);
    const float G = ComputeShadowingMaskingFunction(roughness, V, L, N);
    constexpr float F = 1.0F;
    const float denom = 4 * NL * NV;
    const float bsdf = denom > EPS ? F * G * D / denom : 0.0F;
    return v3(bsdf);
}
inline void sample(MaterialSampling* res, const Scene& ctxt, const atn_material_param& p, const v3& normal, const v3& wi,
    CMJ* sampler, float u, float v)
{
    const float r1 = sampler->nextSample();
    const float r2 = sampler->nextSample();
    res->dir = Diffuse::SampleDirection(normal, r1, r2);
    res->pdf = Diffuse::ComputePDF(normal, res->dir);
    res->bsdf = ComputeBRDF(GGX::roughness_of(ctxt, p, u, v), normal, wi, res->dir);
}
} // namespace Velvet

// ---- MicrofacetRefraction: material/microfacet_refraction.cpp:69-171 -----------------------------------
namespace MicrofacetRefraction {
inline void sample(MaterialSampling& result, const Scene& ctxt, const atn_material_param& p, const v3& n, const v3& wi,
    CMJ* sampler, float tu, float tv)
{
    const float roughness = GGX::roughness_of(ctxt, p, tu, tv);
    const float ior = p.u.standard.ior;
    float ni = 1.0F, nt = ior;
    const v3 V = -wi;
    v3 N = n;
    if (!(dot(V, N) >= 0.0F)) { N = -n; std::swap(ni, nt); }
    const float r1 = sampler->nextSample();
    const float r2 = sampler->nextSample();
    const v3 m = GGX::SampleMicrosurfaceNormal(roughness, N, r1, r2);
    const float F = ComputeSchlickFresnel(ni, nt, wi, m);
    const float R = F, T = 1 - R;
    const float prob = R;
    const float u = sampler->nextSample();
    if (u < prob) {
        const v3 wo = ComputeReflectVector(wi, m);
        const float VN = dot(V, N), LN = dot(wo, N);
        if (VN * LN < 0) {
            result.dir = ComputeReflectVector(wi, N); result.pdf = 1.0f; result.bsdf = v3(0);
            return;
        }
        result.dir = wo;
        result.pdf = GGX::ComputePDFWithHalfVector(roughness, N, m, wo);
        result.pdf *= prob;
        result.bsdf = GGX::ComputeBRDF(roughness, ior, N, wi, wo);
    }
    else {
        const v3 wo = Refraction::ComputeRefractVector(ni, nt, wi, m);
        const float D = GGX::ComputeDistribution(m, N, roughness);
        const float G = GGX::ComputeG2Smith(roughness, V, wo, N);
        const float LH = std::abs(dot(wo, m));
        const float VH = std::abs(dot(V, m));
        const float denom = ni * dot(V, m) + nt * dot(wo, m);
        const float denom2 = denom * denom;
        const float costheta = std::abs(dot(m, n));
        const float nt2 = nt * nt;
        result.pdf = denom2 > 0 ? D * costheta * (nt2 * LH / denom2) : 1.0F;
        result.pdf *= 1.0F - prob;
        result.dir = wo;
        float VN = dot(V, N), LN = dot(wo, N);
        if (VN * LN > 0) {
            result.dir = Refraction::ComputeRefractVector(ni, nt, wi, N); result.pdf = 1.0f; result.bsdf = v3(0);
            return;
        }
        VN = std::abs(VN); LN = std::abs(LN);
        const float bsdf = denom2 > 0 ? ((VH * LH) / (VN * LN)) * (nt2 * T * D * G / denom2) : 0.0F;
        result.bsdf = v3(bsdf);
    }
}
} // namespace MicrofacetRefraction

// ---- dispatch: material/material_impl.h:24-206 (types outside the BASELINE configs fall
//      to the reference's own default branch: Diffuse) --------------------------------------
// ToonSpecular (material/toon.cpp:288-367, toon_specular.h): GGX evaluated with a "stylized highlight" half vector
// (Anjyo & Hiramitsu); a material TYPE that only Toon::ComputeBRDF creates, for pdf / bsdf evaluation.
namespace ToonSpecular {
inline float sign(float f) { return f == 0.0F ? 0.0F : (f > 0.0F ? 1.0F : -1.0F); }     // math/math.h:62-73
inline v3 ComputeHalfVector(const atn_material_param& param, const v3& N, const v3& V, const v3& L)
{
    const auto& h = param.toon.highlight;
    v3 H = normalize(L + V);
    v3 t, b;
    GetTangentCoordinate(N, t, b);
    H = H + h.translation_dt * t + h.translation_db * b;
    H = normalize(H);
    H = H - h.scale_t * dot(H, t) * t - h.scale_b * dot(H, b) * b;
    H = normalize(H);
    H = H - h.split_t * sign(dot(H, t)) * t - h.split_b * sign(dot(H, b)) * b;
    H = normalize(H);
    const float sqrnorm_t = std::sin(std::pow(std::acos(dot(H, t)), h.square_sharp));
    const float sqrnorm_b = std::sin(std::pow(std::acos(dot(H, b)), h.square_sharp));
    H = H - h.square_magnitude * (sqrnorm_t * dot(H, t) * t + sqrnorm_b * dot(H, b) * b);
    H = normalize(H);
    return H;
}
inline float ComputePDF(const atn_material_param& param, const v3& normal, const v3& wi, const v3& wo)
{
    const v3 V = -wi, L = wo, N = normal;
    const v3 H = ComputeHalfVector(param, N, V, L);
    return GGX::ComputePDFWithHalfVector(param.u.standard.roughness, N, H, L);
}
inline v3 ComputeBRDF(const atn_material_param& param, const v3& normal, const v3& wi, const v3& wo)
{
    const v3 V = -wi, L = wo, N = normal;
    const v3 H = ComputeHalfVector(param, N, V, L);
    return GGX::ComputeBRDFWithHalfVector(param.u.standard.roughness, param.u.standard.ior, N, V, L, H);
}
} // namespace ToonSpecular

inline void sampleMaterial(MaterialSampling* result, const Scene& ctxt, const atn_material_param* mtrl,
    const v3& normal, const v3& wi, CMJ* sampler, float u, float v, float pre_sampled_r = 0.0F)
{
    switch (mtrl->type) {
    case ATN_MTRL_CARPAINT:     // CarPaint::sample, car_paint.cpp:178-193
        result->dir = CarPaint::sampleDirection(*mtrl, normal, wi, sampler, pre_sampled_r);
        result->pdf = CarPaint::pdf(*mtrl, normal, wi, result->dir);
        result->bsdf = CarPaint::bsdf(ctxt, *mtrl, normal, wi, result->dir, u, v, pre_sampled_r);
        break;
    case ATN_MTRL_SPECULAR: Specular::sample(result, normal, wi); break;
    case ATN_MTRL_REFRACTION: Refraction::sample(*result, sampler, *mtrl, normal, wi); break;
    case ATN_MTRL_BECKMAN: Beckman::sample(result, ctxt, *mtrl, normal, wi, sampler, u, v); break;
    case ATN_MTRL_OREN_NAYAR: OrenNayar::sample(result, ctxt, *mtrl, normal, wi, sampler, u, v); break;
    case ATN_MTRL_VELVET: Velvet::sample(result, ctxt, *mtrl, normal, wi, sampler, u, v); break;
    case ATN_MTRL_MICROFACET_REFRACTION: MicrofacetRefraction::sample(*result, ctxt, *mtrl, normal, wi, sampler, u, v); break;
    case ATN_MTRL_GGX: GGX::sample(result, ctxt, *mtrl, normal, wi, sampler, u, v); break;
    case ATN_MTRL_DISNEY: Disney::sample(*result, *mtrl, normal, wi, sampler); break;
    case ATN_MTRL_RETROREFLECTIVE: Retroreflective::sample(*result, *mtrl, normal, wi, sampler); break;
    case ATN_MTRL_EMISSIVE:     // emissive::sample == Diffuse (material/emissive.h:70-83)
    case ATN_MTRL_DIFFUSE:
    default: Diffuse::sample(result, normal, sampler); break;
    }
}
inline float samplePDF(const Scene& ctxt, const atn_material_param* mtrl, const v3& normal, const v3& wi, const v3& wo, float u, float v)
{
    switch (mtrl->type) {
    case ATN_MTRL_SPECULAR: return 1.0F;
    case ATN_MTRL_REFRACTION: return 1.0F;        // refraction::pdf asserts and returns 1 (refraction.cpp:8-17); never reached: NEE skips singular materials
    case ATN_MTRL_BECKMAN: return Beckman::ComputePDF(GGX::roughness_of(ctxt, *mtrl, u, v), normal, wi, wo);
    case ATN_MTRL_OREN_NAYAR: return OrenNayar::pdf(normal, wo);
    case ATN_MTRL_VELVET: return Diffuse::ComputePDF(normal, wo);
    case ATN_MTRL_MICROFACET_REFRACTION: return 1.0F;     // asserts and returns 1 (microfacet_refraction.cpp:13-22); singular: NEE never asks
    case ATN_MTRL_GGX: return GGX::ComputePDF(GGX::roughness_of(ctxt, *mtrl, u, v), normal, wi, wo);
    case ATN_MTRL_DISNEY: return Disney::pdf(*mtrl, normal, wi, wo);
    case ATN_MTRL_RETROREFLECTIVE: return Retroreflective::pdf(*mtrl, normal, wi, wo);
    case ATN_MTRL_CARPAINT: return CarPaint::pdf(*mtrl, normal, wi, wo);
    case ATN_MTRL_TOON_SPECULAR: return ToonSpecular::ComputePDF(*mtrl, normal, wi, wo);       // material_impl.h:135-137
    default: return Diffuse::ComputePDF(normal, wo);
    }
}
inline MaterialSampling sampleBSDF(const Scene& ctxt, const atn_material_param* mtrl, const v3& normal, const v3& wi, const v3& wo, float u, float v,
    float pre_sampled_r = 0.0F)
{
    MaterialSampling r;     // pdf = 0 unless the BSDF returns its own (Disney)
    switch (mtrl->type) {
    case ATN_MTRL_SPECULAR: r.bsdf = Specular::ComputeBRDF(wo, normal); break;
    case ATN_MTRL_REFRACTION: r.bsdf = v3(0.0F); break;     // refraction::bsdf asserts and returns vec3() (refraction.cpp:30-39)
    case ATN_MTRL_BECKMAN: r.bsdf = Beckman::ComputeBRDF(GGX::roughness_of(ctxt, *mtrl, u, v), mtrl->u.standard.ior, normal, wi, wo); break;
    case ATN_MTRL_OREN_NAYAR: r.bsdf = OrenNayar::computeBsdf(GGX::roughness_of(ctxt, *mtrl, u, v), normal, wi, wo); break;
    case ATN_MTRL_VELVET: r.bsdf = Velvet::ComputeBRDF(GGX::roughness_of(ctxt, *mtrl, u, v), normal, wi, wo); break;
    case ATN_MTRL_MICROFACET_REFRACTION: r.bsdf = v3(0.0F); break;
    case ATN_MTRL_GGX: r.bsdf = GGX::ComputeBRDF(GGX::roughness_of(ctxt, *mtrl, u, v), mtrl->u.standard.ior, normal, wi, wo); break;
    case ATN_MTRL_DISNEY: r = Disney::bsdf(*mtrl, normal, wi, wo); break;
    case ATN_MTRL_RETROREFLECTIVE: r = Retroreflective::bsdf(*mtrl, normal, wi, wo); break;
    case ATN_MTRL_CARPAINT: r.bsdf = CarPaint::bsdf(ctxt, *mtrl, normal, wi, wo, u, v, pre_sampled_r); break;
    case ATN_MTRL_TOON_SPECULAR: r.bsdf = ToonSpecular::ComputeBRDF(*mtrl, normal, wi, wo); break;     // material_impl.h:196-198
    default: r.bsdf = Diffuse::ComputeBRDF(); break;
    }
    return r;
}

// FillMaterial, material_impl.h:232-262 (voxel branch dead on this path)
inline void FillMaterial(atn_material_param& dst, const Scene& ctxt, int32_t mtrl_id)
{
    if (mtrl_id >= 0) {
        dst = ctxt.GetMaterial((uint32_t)mtrl_id);
    }
    else {
        std::memset(&dst, 0, sizeof(dst));
        dst.type = ATN_MTRL_DIFFUSE;
        dst.attrib = 0;
        dst.baseColor = atn_vec4{ 1.0f, 1.0f, 1.0f, 1.0f };    // vec4::operator=(vec3) keeps w = 1 (vec4.h:135-141)
        dst.albedoMap = dst.normalMap = dst.roughnessMap = -1;
        dst.u.standard = atn_standard_mtrl{ 1.0f, 0.5f, 1.0f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f };
    }
}

// ---------------------------------------------------------------------------------------
// Lights: light/light_impl.h:12-43, arealight.h:58-143, ibl.h:46-133, pointlight.h:40-58,
//         spotlight.h:58-92, directionallight.h:40-63; renderer/background.h:34-127
// ---------------------------------------------------------------------------------------
struct LightSampleResult {
    v3 pos, dir; float dist_to_light{ 0 };
    v3 nml; v3 light_color; float pdf{ 0 };
    uint32_t attrib{ 0 };
};

inline v3 ConvertDirectionToUV(const v3& dir)       // background.h:88-127
{
    float temp = std::atan2(dir.x, dir.z);
    float r = length(dir);
    float phi = (float)((temp >= 0) ? temp : (temp + 2 * PI));
    float theta = std::acos(dir.y / r);
    float u = phi / (2 * PI);
    float v = 1 - theta / PI;
    return v3(u, v, 0);
}
inline v4 Background_SampleFromRay(const v3& in_ray, const atn_background& bg, const Scene& ctxt)  // background.h:34-62
{
    if (bg.envmap_tex_idx < 0 || !bg.enable_env_map) {
        return v4(v3(bg.bg_color[0], bg.bg_color[1], bg.bg_color[2]));   // vec3 -> vec4 : w = 0
    }
    v3 uv = ConvertDirectionToUV(in_ray);
    const v4 result = sampleTexture(ctxt, bg.envmap_tex_idx, uv.x, uv.y, v4(1.0F));
    return result * bg.multiplyer;
}
inline float IBL_samplePdf(const v3& clr, float avgIllum)     // ibl.h:46-58
{
    float illum = luminance(clr);
    float pdf = illum / avgIllum;
    pdf /= (2.0f * PI);
    return pdf;
}

inline v3 AreaLight_ComputeLightColor(const atn_light_param& p, float area)   // arealight.h:58-63
{
    float lum = p.scale * p.intensity / area;
    return v3(p.light_color[0], p.light_color[1], p.light_color[2]) * lum;
}

inline void AreaLight_sample(LightSampleResult& result, const atn_light_param& param, const Scene& ctxt, const v3& org, CMJ* sampler)
{
    if (param.arealight_objid < 0) return;
    const atn_object_param& obj = ctxt.GetObject(param.arealight_objid);

    // evaluate SamplePosAndNormal (EvaluateHitResult.h:74-111) -> PolygonObject::SamplePosAndNormal
    // (PolygonObject.h:113-156) -> triangle::SamplePosAndNormal (triangle.h:122-162)
    const atn_object_param& real_obj = obj.type == ATN_OBJ_INSTANCE ? ctxt.GetObject(obj.object_id) : obj;
    if (real_obj.type == ATN_OBJ_SPHERE) {
        // sphere::SamplePosAndNormal (geometry/sphere.cpp:109-150): result.triangle_id stays -1, so
        // AreaLight::sample falls to sphere::hit along the ray towards the sampled point (arealight.h:105-116)
        const float r1 = sampler->nextSample();
        const float r2 = sampler->nextSample();
        const float rad = real_obj.sphere.radius;
        const v3 center = ld3(real_obj.sphere.center);
        const float z = 2.0F * r1 - 1.0F;
        const float sin_theta = std::sqrt(1 - z * z);
        const float phi = 2 * PI * r2;
        const float x = std::cos(phi) * sin_theta;
        const float y = std::sin(phi) * sin_theta;
        v3 sdir = normalize(v3(x, y, z));
        const v3 pos = center + sdir * (rad + EPS);
        Ray ray(org, pos - org);
        Isect isect;
        if (!sphere_hit(obj, ray, EPS, INF, &isect)) return;
        HitRec rec;
        evaluate_hit_result(rec, obj, ctxt, ray, isect);
        result.pos = rec.p;
        result.pdf = 1 / rec.area;
        result.dir = rec.p - org;
        result.dist_to_light = length(result.dir);
        result.dir = normalize(result.dir);
        result.nml = rec.normal;
        result.light_color = AreaLight_ComputeLightColor(param, rec.area);
        return;
    }
    if (real_obj.type != ATN_OBJ_POLYGONS) return;

    float r = sampler->nextSample();
    uint32_t tri_idx = static_cast<uint32_t>(real_obj.triangle_num * r);
    tri_idx += real_obj.triangle_id;
    const auto& tri = ctxt.GetTriangle(tri_idx);
    const v4 p0 = ctxt.GetPositionAsVec4(tri.idx[0]);
    const v4 p1 = ctxt.GetPositionAsVec4(tri.idx[1]);
    const v4 p2 = ctxt.GetPositionAsVec4(tri.idx[2]);
    float r0 = sampler->nextSample();
    float r1 = sampler->nextSample();
    float a = std::sqrt(r0) * (1.0F - r1);
    float b = std::sqrt(r0) * r1;
    v3 pos = ((1 - a - b) * p0 + a * p1 + b * p2).xyz();

    v3 dir = pos - org;
    Ray ray(org, dir);
    Isect isect;
    isect.t = length(dir);
    isect.tri_id = (int32_t)tri_idx;
    isect.a = a;
    isect.b = b;

    HitRec rec;
    evaluate_hit_result(rec, obj, ctxt, ray, isect);

    // AreaLight::sample(hitrecord...), arealight.h:39-56
    result.pos = rec.p;
    result.pdf = 1 / rec.area;
    result.dir = rec.p - org;
    result.dist_to_light = length(result.dir);
    result.dir = normalize(result.dir);
    result.nml = rec.normal;
    result.light_color = AreaLight_ComputeLightColor(param, rec.area);
}

inline float scene_radius_for_ibl(const Scene& ctxt)    // ibl.h:106-111, math/aabb.h:231-234,346-362
{
    float scene_radius = 10000.0F;
    const float* mn = ctxt.d->scene_bbox_min; const float* mx = ctxt.d->scene_bbox_max;
    bool valid = !((mn[0] >= mx[0]) || (mn[1] >= mx[1]) || (mn[2] >= mx[2]));
    if (valid) {
        v3 vmn = ld3(mn), vmx = ld3(mx);
        v3 center = (vmn + vmx) * 0.5F;                 // aabb::getCenter, math/aabb.h
        float radius = length(vmx - center);
        scene_radius = radius / std::tan(Deg2Rad(30.0F) / 2);
    }
    return scene_radius;
}

// ---- optional table sampler: ImageBasedLight::preCompute + sample(ctxt, org, nml, sampler), light/ibl.cpp:10-230
inline void IBL_preCompute(SamplingOptions& o, const Scene& ctxt)
{
    const auto& bg = ctxt.cfg().bg;
    const atn_texture_desc& envmap = ctxt.d->textures[bg.envmap_tex_idx];
    if (o.env == envmap.texels && o.w == envmap.width && o.h == envmap.height && o.multiplyer == bg.multiplyer) return;
    o.env = envmap.texels; o.w = envmap.width; o.h = envmap.height; o.multiplyer = bg.multiplyer;
    const int32_t width = envmap.width, height = envmap.height;
    o.cdfV.clear(); o.cdfU.assign(height, std::vector<float>());
    for (int32_t y = 0; y < height; y++) {
        float scale = std::sin(PI * (float)(y + 0.5) / height);
        float pdfV = 0;
        std::vector<float>& pdfU = o.cdfU[y];
        for (int32_t x = 0; x < width; x++) {
            float u = (float)(x + 0.5) / width;
            float v = (float)(y + 0.5) / height;
            const v4 t = texture_at(envmap, u, v);
            const v3 clr = v3(t.x, t.y, t.z) * bg.multiplyer;      // SampleFromUVWithTexture, ibl.h:147-153
            const float illum = luminance(clr);
            pdfV += illum * scale;
            pdfU.push_back(illum * scale);
        }
        o.cdfV.push_back(pdfV);
    }
    auto normalise = [](std::vector<float>& c) {
        float sum = 0;
        for (size_t i = 0; i < c.size(); i++) { sum += c[i]; if (i > 0) c[i] += c[i - 1]; }
        if (sum > 0) {
            float invSum = 1 / sum;
            for (size_t i = 0; i < c.size(); i++) { c[i] *= invSum; c[i] = std::min(std::max(c[i], 0.0F), 1.0F); }
        }
    };
    normalise(o.cdfV);
    for (auto& row : o.cdfU) normalise(row);
}
inline int32_t IBL_samplePdfAndCdf(float r, const std::vector<float>& cdf, float& outPdf)     // ibl.cpp:133-176
{
    outPdf = 0;
    if (cdf.size() < 2) { outPdf = cdf.empty() ? 0.0F : cdf[0]; return 0; }
    int32_t idxTop = 0, idxTail = (int32_t)cdf.size() - 1;
    for (;;) {
        int32_t idxMid = (idxTop + idxTail) >> 1;
        if (r < cdf[idxMid]) idxTail = idxMid; else idxTop = idxMid;
        if ((idxTail - idxTop) == 1) {
            const float topCdf = cdf[idxTop], tailCdf = cdf[idxTail];
            if (r <= topCdf) { outPdf = topCdf; return idxTop; }
            outPdf = tailCdf - topCdf;
            return idxTail;
        }
    }
}
// the true solid-angle density of the table sampler (the reference writes pi^2 where the texel's solid angle calls for
// 2 pi^2: see csrc/device/shading.hpp, ibl_texel_pdf)
inline float IBL_texel_pdf(const SamplingOptions& o, float pdfU, float pdfV, int32_t y)
{
    const float v = (float)(y + 0.5) / o.h;
    const float theta = PI * v;
    const float pi2 = PI * PI;
    return (pdfU * pdfV) * ((float)(o.w * o.h) / ((2.0F * pi2) * std::sin(theta)));
}
inline v3 ConvertUVToDirection(float u, float v)       // renderer/background.h:64-86
{
    float phi = 2 * PI * u;
    float theta = (1 - v) * PI;
    v3 dir;
    dir.y = std::cos(theta);
    float xz = std::sqrt(1 - dir.y * dir.y);
    dir.x = xz * std::sin(phi);
    dir.z = xz * std::cos(phi);
    return normalize(dir);
}
inline float IBL_direction_pdf(const Scene& ctxt, const v3& dir)
{
    SamplingOptions& o = sampling_options();
    const v3 uv = ConvertDirectionToUV(dir);
    int32_t x = (int32_t)(uv.x * (float)o.w), y = (int32_t)(uv.y * (float)o.h);
    x = std::min(std::max(x, 0), o.w - 1); y = std::min(std::max(y, 0), o.h - 1);
    const std::vector<float>& cu = o.cdfU[y];
    const float pu = x > 0 ? cu[x] - cu[x - 1] : cu[0];
    const float pv = y > 0 ? o.cdfV[y] - o.cdfV[y - 1] : o.cdfV[0];
    return IBL_texel_pdf(o, pu, pv, y);
}

inline void IBL_sample(LightSampleResult& result, const atn_light_param& param, const Scene& ctxt, const v3& org, const v3& nml, CMJ* sampler)
{
    const float r1 = sampler->nextSample();
    const float r2 = sampler->nextSample();
    SamplingOptions& o = sampling_options();
    if (o.ibl_importance && ctxt.cfg().bg.envmap_tex_idx >= 0 && ctxt.cfg().bg.enable_env_map) {
        float pdfU, pdfV;
        const int32_t y = IBL_samplePdfAndCdf(r1, o.cdfV, pdfV);
        const int32_t x = IBL_samplePdfAndCdf(r2, o.cdfU[y], pdfU);
        const float u = (float)(x + 0.5) / o.w;
        const float v = (float)(y + 0.5) / o.h;
        result.pdf = IBL_texel_pdf(o, pdfU, pdfV, y);
        result.dir = ConvertUVToDirection(u, v);
        const v4 lum = sampleTexture(ctxt, param.envmapidx, u, v, v4(1.0F)) * ctxt.cfg().bg.multiplyer;
        result.light_color = (param.scale * lum).xyz();
        result.pos = org + scene_radius_for_ibl(ctxt) * result.dir;
        result.nml = -normalize(result.dir);
        result.dist_to_light = 1.0F;
        return;
    }
    result.dir = Diffuse::SampleDirection(nml, r1, r2);
    const v3 uv = ConvertDirectionToUV(result.dir);
    float scene_radius = scene_radius_for_ibl(ctxt);
    result.pos = org + scene_radius * result.dir;
    result.nml = -normalize(result.dir);
    result.pdf = 1.0f / (2.0f * PI);
    result.dist_to_light = 1.0F;
    const v4 lum = sampleTexture(ctxt, param.envmapidx, uv.x, uv.y, v4(1.0F));
    result.light_color = (param.scale * lum).xyz();
}

inline void Light_sample(LightSampleResult& result, const atn_light_param& param, const Scene& ctxt, const v3& org, const v3& nml, CMJ* sampler)
{
    const v3 lpos(param.pos.x, param.pos.y, param.pos.z);
    const v3 ldir(param.dir.x, param.dir.y, param.dir.z);
    const v3 lcol(param.light_color[0], param.light_color[1], param.light_color[2]);
    switch (param.type) {
    case ATN_LIGHT_AREA: AreaLight_sample(result, param, ctxt, org, sampler); break;
    case ATN_LIGHT_IBL: IBL_sample(result, param, ctxt, org, nml, sampler); break;
    case ATN_LIGHT_POINT: {
        result.pdf = 1.0f;
        result.dir = lpos - org;
        result.dist_to_light = length(result.dir);
        result.dir = normalize(result.dir);
        result.pos = lpos;
        result.nml = normalize(-result.dir);
        const float dist2 = sqr(result.dist_to_light);
        result.light_color = lcol * param.scale * param.intensity / dist2;
        break;
    }
    case ATN_LIGHT_SPOT: {
        result.pdf = 1.0f;
        result.pos = lpos;
        result.nml = ldir;
        result.dir = lpos - org;
        result.dist_to_light = length(result.dir);
        result.dir = normalize(result.dir);
        v3 dir_to_light = -result.dir;
        float rho = dot(ldir, dir_to_light);
        float cosHalfInner = std::cos(param.innerAngle * 0.5F);
        float cosHalfOuter = std::cos(param.outerAngle * 0.5F);
        if (rho > cosHalfOuter) {
            float att = (rho - cosHalfOuter) / (cosHalfInner - cosHalfOuter);
            att = clamp_(att, 0.0f, 1.0f);
            float dist2 = sqr(result.dist_to_light);
            result.light_color = param.scale * lcol * att * param.intensity / dist2;
        }
        else {
            result.pdf = 0.0f;
            result.light_color = v3(0.0F);
        }
        break;
    }
    case ATN_LIGHT_DIRECTION: {
        result.pdf = 1.0f;
        // param.dir is a vec4: normalize(vec4) includes w (vec4.h:293-298)
        v4 nd = normalize(v4(param.dir.x, param.dir.y, param.dir.z, param.dir.w));
        result.dir = -nd.xyz();
        result.nml = nd.xyz();
        result.pos = org + 100000.0F * 0.5F * result.dir;
        result.light_color = lcol * param.scale * param.intensity;
        result.dist_to_light = 1.0F;
        break;
    }
    default: break;
    }
    result.attrib = param.attrib;
}

// ---------------------------------------------------------------------------------------
// Path state: renderer/pathtracing/pt_params.h:25-78,189-200
// ---------------------------------------------------------------------------------------
struct PathState {
    v3 throughput{ 1.0F }; float pdfb{ 1.0F };
    v3 contrib{ 0.0F }; float samples{ 0 };
    bool isHit{ false }, is_terminated{ false }, is_singular{ false };
    int32_t last_hit_mtrl_idx{ -1 };
    int32_t screen_space_x{ 0 }, screen_space_y{ 0 };       // PathAttribute, pt_params.h:69-70
    CMJ sampler;
};
struct ShadowRay {
    v3 rayorg; float distToLight{ 0 };
    v3 raydir; bool isActive{ false };
    v3 lightcontrib; uint32_t targetLightId{ 0 };
};

struct PathCounters {       // not in the reference: work counters for the roofline model
    uint64_t closest_rays{ 0 }, shadow_rays{ 0 }, hits{ 0 };
    TraverseStats trav;
};

// GeneratePath, renderer/pathtracing/pathtracing_impl.h:65-110
inline void GeneratePath(Ray& generated_ray, int32_t ix, int32_t iy, int32_t sample, uint32_t frame,
    PathState& path, const atn_camera_param& camera, const uint32_t rnd)
{
    auto scramble = rnd * 0x1fe3434f * (((frame + sample) + 133 * rnd) / (CMJ::CMJ_DIM * CMJ::CMJ_DIM));
    path.sampler.init((frame + sample) % (CMJ::CMJ_DIM * CMJ::CMJ_DIM), 0, scramble);
    float r1 = path.sampler.nextSample();
    float r2 = path.sampler.nextSample();
    float s = (ix + r1) / (float)(camera.width);
    float t = (iy + r2) / (float)(camera.height);
    generated_ray = PinholeSample(camera, s, t);
    path.throughput = v3(1);
    path.pdfb = 1.0f;
    path.isHit = false; path.is_terminated = false; path.is_singular = false;
    path.last_hit_mtrl_idx = -1;
    path.screen_space_x = ix; path.screen_space_y = iy;     // pathtracing_impl.h:106-107
    path.samples += 1;
}

// ComputeRadianceNEE, renderer/pathtracing/pathtracing_nee_impl.h:23-95
inline bool ComputeRadianceNEE(v3& out, const Scene& ctxt, const v3& wi, const v3& surface_nml,
    const atn_material_param& surface_mtrl, float hit_u, float hit_v, float light_select_prob,
    const LightSampleResult& ls, float pre_sampled_random = 0.0F, float* weight_ptr = nullptr)
{
    if (weight_ptr) *weight_ptr = 0.0F;
    float cosShadow = dot(surface_nml, ls.dir);
    float path_pdf = samplePDF(ctxt, &surface_mtrl, surface_nml, wi, ls.dir, hit_u, hit_v);
    MaterialSampling ev = sampleBSDF(ctxt, &surface_mtrl, surface_nml, wi, ls.dir, hit_u, hit_v, pre_sampled_random);
    if (ev.pdf > 0) path_pdf = ev.pdf;
    const v3& bsdf = ev.bsdf;
    const v3& emit = ls.light_color;
    float cosLight = dot(ls.nml, -ls.dir);
    float dist2 = sqr(ls.dist_to_light);
    const bool isInfinite = (ls.attrib & ATN_LIGHT_ATTR_INFINITE) != 0;
    const bool is_singular = (ls.attrib & ATN_LIGHT_ATTR_SINGULAR) != 0;
    dist2 = (isInfinite || is_singular) ? 1.0F : dist2;

    if (cosShadow >= 0 && cosLight >= 0 && dist2 > 0 && path_pdf > 0.0F && ls.pdf > 0.0F) {
        if (!isInfinite) path_pdf = path_pdf * cosLight / dist2;
        float misW = is_singular ? 1.0f : (ls.pdf * light_select_prob) / ((ls.pdf * light_select_prob) + path_pdf);
        const float G = isInfinite ? cosShadow * cosLight : cosShadow * cosLight / dist2;
        out = (misW * bsdf * emit * G / ls.pdf) / light_select_prob;
        if (weight_ptr) *weight_ptr = misW / ls.pdf / light_select_prob;     // pathtracing_nee_impl.h:87-89
        return true;
    }
    return false;
}

// SampleLight + FillShadowRay, pathtracing_impl.h:178-264
inline void FillShadowRay(ShadowRay& shadow_ray, const Scene& ctxt, PathState& path,
    const atn_material_param& mtrl, const Ray& ray, const v3& hit_pos, const v3& hit_nml,
    float hit_u, float hit_v, const v4& external_albedo, float pre_sampled_r = 0.0F)
{
    shadow_ray.isActive = false;
    const int32_t lightnum = ctxt.GetLightNum();
    bool is_invalid_mtrl = attr_singular(mtrl) || attr_translucent(mtrl);
    if (lightnum <= 0 || is_invalid_mtrl) return;

    int32_t target_light_idx = std::min<int32_t>(static_cast<int32_t>(path.sampler.nextSample() * lightnum), lightnum - 1);
    float lightSelectPdf = 1.0f / lightnum;
    const auto& light = ctxt.GetLight(target_light_idx);
    LightSampleResult sampleres;
    Light_sample(sampleres, light, ctxt, hit_pos, hit_nml, &path.sampler);

    v3 dirToLight = normalize(sampleres.dir);
    float distToLight = length(sampleres.pos - hit_pos);
    shadow_ray.rayorg = Ray::Offset(hit_pos, hit_nml);
    shadow_ray.raydir = dirToLight;
    shadow_ray.targetLightId = target_light_idx;
    shadow_ray.distToLight = distToLight;
    shadow_ray.lightcontrib = v3(0);

    v3 radiance;
    if (ComputeRadianceNEE(radiance, ctxt, ray.dir, hit_nml, mtrl, hit_u, hit_v, lightSelectPdf, sampleres, pre_sampled_r)) {
        // vec3 * vec3 * vec4 (component-wise; .w dropped on store)
        shadow_ray.lightcontrib = path.throughput * radiance * external_albedo.xyz();
        shadow_ray.isActive = true;
    }
}

// HitShadowRay + HitTestToTargetLight + scene::hitLight,
// pathtracing_impl.h:266-393, scene/scene.h:64-134.
// surface_stencil_type: stencil_type of the material at the SHADED point (pathtracing.cpp:59-66 passes
// ctxt.GetMaterial(isect.mtrlid)); ALWAYS raises the lookup budget to 10 and makes STENCIL surfaces transparent to
// the shadow ray; scene_rendering_config.enable_alpha_blending raises it to 10 as well.
// HitTestToTargetLight, pathtracing_impl.h:266-365
inline bool HitTestToTargetLight(const Scene& ctxt, const Ray& original_ray, const atn_light_param& light, float distToLight,
    int32_t surface_stencil_type, PathCounters* cnt)
{
    const bool valid_obj = (light.type == ATN_LIGHT_AREA) && light.arealight_objid >= 0;
    const int32_t lightobj = valid_obj ? light.arealight_objid : -1;
    int32_t hitobj = lightobj;      // kept across lookups, like the reference's pointer (:291)

    Ray r = original_ray;
    size_t max_lookups = ctxt.d->config.enable_alpha_blending ? 10 : 1;
    const bool need_stencil_check = surface_stencil_type == 1;      // StencilType::ALWAYS
    max_lookups = need_stencil_check ? 10 : max_lookups;

    bool is_hit_to_light = false;
    if (cnt) cnt->shadow_rays++;
    for (size_t i = 0; i < max_lookups; i++) {
        Isect isect;
        bool isHit = TraverseClosest(isect, ctxt, r, EPS, distToLight - EPS, cnt ? &cnt->trav : nullptr);
        if (isHit) {
            hitobj = isect.objid;
            const auto& hobj = ctxt.GetObject(static_cast<uint32_t>(isect.objid));
            HitRec rec;
            evaluate_hit_result(rec, hobj, ctxt, r, isect);
            bool is_ignore_hit = false;
            if (isect.mtrlid >= 0) {        // (the reference indexes unconditionally)
                const auto& hm = ctxt.GetMaterial(isect.mtrlid);
                if (need_stencil_check && hm.stencil_type == 2) is_ignore_hit = true;       // StencilType::STENCIL
                // material::isTranslucentByAlpha (material.cpp:193-210)
                v4 albedo = sampleTexture(ctxt, hm.albedoMap, rec.u, rec.v, v4(1.0F));
                const float alpha = albedo.w * hm.baseColor.w;
                if (alpha < 1.0F) is_ignore_hit = true;
            }
            if (is_ignore_hit) {
                // go through the object: offset along the normal that faces the ray's direction (:319-330); with a
                // budget of one lookup the loop ends here and the ray counts as blocked
                v3 orienting_normal = rec.normal;
                const bool is_same_facing = dot(rec.normal, original_ray.dir) > 0.0F;
                if (!is_same_facing) orienting_normal = -orienting_normal;
                r = Ray(rec.p, original_ray.dir, orienting_normal);
                continue;
            }
        }
        if (hitobj == lightobj) is_hit_to_light = true;
        else if (light.attrib & ATN_LIGHT_ATTR_INFINITE) is_hit_to_light = !isHit;
        else if (light.attrib & ATN_LIGHT_ATTR_SINGULAR) is_hit_to_light = isect.t > distToLight;
        else is_hit_to_light = false;
        break;
    }
    return is_hit_to_light;
}

inline bool HitShadowRay(const Scene& ctxt, PathState& path, const ShadowRay& shadow_ray, int32_t surface_stencil_type,
    PathCounters* cnt)
{
    if (path.is_terminated) return false;
    if (!shadow_ray.isActive) return false;
    const auto& light = ctxt.GetLight(shadow_ray.targetLightId);
    const Ray original_ray(shadow_ray.rayorg, shadow_ray.raydir);
    const bool is_hit_to_light = HitTestToTargetLight(ctxt, original_ray, light, shadow_ray.distToLight, surface_stencil_type, cnt);
    if (is_hit_to_light) path.contrib += shadow_ray.lightcontrib;
    return is_hit_to_light;
}

// ---------------------------------------------------------------------------------------
// Toon / StylizedBrdf, material/toon.cpp:88-286,371-445.  "Terminated" materials (material.h:583-588): at bounce 0 the
// path tracer treats them as a light (HitTeminatedMaterial, pathtracing_impl.h:482-503) whose colour is Toon::bsdf --
// a one-sample NEE towards ONE designated target light (with its own visibility test), quantised through a 1-D remap
// texture, times a screen-space shadow texture, plus a rim light.
// ---------------------------------------------------------------------------------------
namespace Toon {
inline float bezier_smoothstep(float edge0, float edge1, float mid, float t, float s)   // toon.cpp:222-239
{
    if (t <= edge0) return 0;
    else if (t >= edge1) return 1;
    t = (t - edge0) / (edge1 - edge0);
    t *= s;
    float B0 = 0.0F, B1 = mid, B2 = 1.0F;
    float P = (B0 - 2 * B1 + B2) * t * t + (-2 * B0 + 2 * B1) * t + B0;
    return P;
}
inline v3 ComputeRimLight(const atn_material_param& param, const v3& normal, const v3& wi)     // toon.cpp:243-286
{
    v3 c(0.0F);
    const v3 V = -wi, N = normal;
    if (param.toon.rim_light.enable) {
        const float NdotV = dot(V, N);
        if (NdotV > 0) {
            const float rim = bezier_smoothstep(1.0F - param.toon.rim_light.width, 1.0F, (1 - param.toon.rim_light.softness) * 0.5F,
                1 - NdotV, param.toon.rim_light.spread);
            c += rim * v3(param.toon.rim_light.color[0], param.toon.rim_light.color[1], param.toon.rim_light.color[2]);
        }
    }
    return c;
}
inline atn_material_param base_material(const atn_material_param& param)        // toon.cpp:184-190,390-396
{
    atn_material_param base_mtrl = param;
    base_mtrl.type = param.toon.toon_type == ATN_MTRL_DIFFUSE ? ATN_MTRL_DIFFUSE : ATN_MTRL_TOON_SPECULAR;
    return base_mtrl;
}
// Toon::ComputeBRDF, toon.cpp:163-218
inline void ComputeBRDF(v3& toon_term, float& remap_v, const Scene& ctxt, const atn_material_param& param,
    const LightSampleResult* sampled_light, const v3& normal, const v3& wi, float u, float v)
{
    v3 radiance(0.0F);
    if (sampled_light) {
        const atn_material_param base_mtrl = base_material(param);
        v3 res;
        if (ComputeRadianceNEE(res, ctxt, wi, normal, base_mtrl, u, v, 1.0F, *sampled_light, 0.0F)) radiance = res;
    }
    float lum_y = clamp_(luminance(radiance), 0.0F, 1.0F);
    lum_y = clamp_(std::pow(lum_y, 1.0F / 2.2F), 0.0F, 1.0F);
    const v4 remap = sampleTexture(ctxt, param.toon.remap_texture, lum_y, 0.5F, v4(1.0F));
    toon_term = remap.xyz();
    remap_v = lum_y;
}
// StylizedBrdf::ComputeBRDF, toon.cpp:373-445
inline void Stylized_ComputeBRDF(v3& toon_term, float& remap_v, const Scene& ctxt, const atn_material_param& param,
    const LightSampleResult* sampled_light, const v3& normal, const v3& wi, float u, float v)
{
    v3 radiance(0.0F);
    float pdf = 1.0F;
    if (sampled_light) {
        const atn_material_param base_mtrl = base_material(param);
        float nee_weight = 0.0F;
        v3 res;
        if (ComputeRadianceNEE(res, ctxt, wi, normal, base_mtrl, u, v, 1.0F, *sampled_light, 0.0F, &nee_weight)) {
            radiance = res;
            pdf = 1.0F / nee_weight;
        }
    }
    constexpr float W_MIN = 0.01F;
    // color::sRGBtoXYZ (misc/color.h:75-88): only Y is used
    const float y = dot(v3(0.212639F, 0.71517F, 0.0721926F), radiance);
    const float weight = fmax_(y, W_MIN);
    const float y_min = fmax_(0.0F, fmin_(param.toon.stylized_y_min, param.toon.stylized_y_max));
    const float y_max = fmax_(param.toon.stylized_y_min, param.toon.stylized_y_max);
    float rv = 0.0F;
    if (y_max <= y) rv = 1.0F;
    else if (y <= y_min) rv = 0.0F;
    else rv = (y - y_min) / (y_max - y_min);
    const v4 remap = sampleTexture(ctxt, param.toon.remap_texture, rv, 0.5F, v4(radiance.x, radiance.y, radiance.z, 0.0F));
    toon_term = weight * remap.xyz() * pdf;
    remap_v = rv;
}
// Toon::bsdf, toon.cpp:88-161
inline v3 bsdf(const Scene& ctxt, const atn_material_param& param, PathState& path, const v3& hit_pos, const v3& normal,
    const v3& wi, float u, float v, PathCounters* cnt)
{
    const atn_light_param* target_light = (param.toon.target_light_idx >= 0 && (uint32_t)param.toon.target_light_idx < ctxt.d->n_npr_target_lights)
        ? &ctxt.GetNprTargetLight((uint32_t)param.toon.target_light_idx) : nullptr;     // (the reference asserts the range)
    v3 toon_term(0.0F);
    float remap_v = 1.0F;
    if (target_light) {
        LightSampleResult light_sample;
        Light_sample(light_sample, *target_light, ctxt, hit_pos, normal, &path.sampler);
        const Ray r(hit_pos, light_sample.dir, normal);
        bool is_hit_to_target_light = true;
        if (param.toon.will_receive_shadow)
            is_hit_to_target_light = HitTestToTargetLight(ctxt, r, *target_light, light_sample.dist_to_light, param.stencil_type, cnt);
        if (param.type == ATN_MTRL_TOON)
            ComputeBRDF(toon_term, remap_v, ctxt, param, is_hit_to_target_light ? &light_sample : nullptr, normal, wi, u, v);
        else if (param.type == ATN_MTRL_STYLIZED_BRDF)
            Stylized_ComputeBRDF(toon_term, remap_v, ctxt, param, is_hit_to_target_light ? &light_sample : nullptr, normal, wi, u, v);
    }
    if (param.toon.stylized_shadow.enable) {
        float shadow = ctxt.GetScreenSpaceTextureAt(path.screen_space_x, path.screen_space_y);
        if (remap_v >= param.toon.stylized_shadow.threshold) shadow = 1.0F;
        else {
            const float offset = param.toon.stylized_shadow.offset, scale = param.toon.stylized_shadow.scale;
            shadow = fmin_(fmax_(shadow * (remap_v + offset) * scale, shadow), 1.0F);
        }
        toon_term = toon_term * shadow;
    }
    return toon_term + ComputeRimLight(param, normal, wi);
}
} // namespace Toon

// HitImplicitLight, pathtracing_impl.h:395-451
inline bool HitImplicitLight(const Scene& ctxt, int32_t hit_obj_id, bool is_back_facing, int32_t bounce,
    PathState& path, const Ray& ray, const HitRec& hrec, const atn_material_param& m)
{
    if (!attr_emissive(m)) return false;
    if (is_back_facing) return false;
    const auto& obj = ctxt.GetObject(hit_obj_id);
    // the reference indexes lights[light_id] unchecked (UB for an emissive surface that is no registered light);
    // both sides of the parity test define that case as "emits nothing"
    const bool is_light = obj.light_id >= 0 && obj.light_id < ctxt.GetLightNum();
    const v3 light_color = is_light ? AreaLight_ComputeLightColor(ctxt.GetLight(obj.light_id), hrec.area) : v3(0.0f);
    float weight = 1.0f;
    if (bounce > 0) {
        float cosLight = dot(hrec.normal, -ray.dir);
        float dist2 = squared_length(hrec.p - ray.org);
        if (cosLight >= 0) {
            float pdfLight = 1 / hrec.area;
            pdfLight = pdfLight * dist2 / cosLight;
            weight = path.pdfb / (path.pdfb + pdfLight);
        }
    }
    v3 contrib = path.throughput * weight * light_color;
    path.contrib += contrib;
    path.is_terminated = true;
    return true;
}

// ComputeRussianProbability, pathtracing_impl.h:680-698
inline float ComputeRussianProbability(int32_t bounce, int32_t rr_bounce, PathState& path)
{
    float russian_prob = 1.0f;
    if (bounce > rr_bounce) {
        if (squared_length(path.throughput) > 0) {
            russian_prob = max_from_vec3(path.throughput);
            float p = path.sampler.nextSample();
            path.is_terminated = (p >= russian_prob);
        }
    }
    return russian_prob;
}

// PrepareForNextBounce, pathtracing_impl.h:700-743
inline void PrepareForNextBounce(const HitRec& rec, float russian_prob, const v3& normal,
    const atn_material_param& mtrl, const MaterialSampling& sampling, const v3& albedo,
    PathState& path, Ray& ray)
{
    const v3 next_dir = normalize(sampling.dir);
    const float pdfb = sampling.pdf;
    const v3 bsdf = sampling.bsdf;
    v3 ray_along_normal = dot(normal, next_dir) >= 0.0f ? normal : -normal;
    float c = dot(ray_along_normal, next_dir);
    if (pdfb > 0 && c > 0) {
        path.throughput *= albedo * bsdf * c / pdfb;
        path.throughput /= russian_prob;
    }
    else {
        path.is_terminated = true;
    }
    if (path.is_terminated) return;
    path.pdfb = pdfb;
    path.is_singular = attr_singular(mtrl);
    path.last_hit_mtrl_idx = mtrl.id;
    ray = Ray(rec.p, next_dir, ray_along_normal);
}

// PathTracing::shade, renderer/pathtracing/pathtracing.cpp:91-236
inline void shade(PathState& path, const Scene& ctxt, Ray& ray, ShadowRay& shadow_ray, const Isect& isect,
    int32_t rrDepth, int32_t bounce, PathCounters* cnt)
{
    if (path.is_terminated) return;
    if (cnt) cnt->hits++;
    const Ray ray_in = ray;
    const auto& obj = ctxt.GetObject(static_cast<uint32_t>(isect.objid));
    HitRec rec;
    evaluate_hit_result(rec, obj, ctxt, ray_in, isect);

    bool isBackfacing = dot(rec.normal, -ray_in.dir) < 0.0F;
    v3 orienting_normal = rec.normal;

    atn_material_param mtrl;
    FillMaterial(mtrl, ctxt, rec.mtrlid);

    v4 albedo = sampleTexture(ctxt, mtrl.albedoMap, rec.u, rec.v,
        v4(mtrl.baseColor.x, mtrl.baseColor.y, mtrl.baseColor.z, mtrl.baseColor.w));
    shadow_ray.isActive = false;

    // alpha_blend.transmission == 1, alpha_blend.throughput == 0 on this path (:145)
    albedo = 1.0F * albedo + v4(v3(0.0F));

    // HitTeminatedMaterial, pathtracing_impl.h:453-509, and what PathTracing::shade does with its answer
    // (pathtracing.cpp:160-184)
    const bool is_toon_mtrl = mtrl.type == ATN_MTRL_TOON || mtrl.type == ATN_MTRL_STYLIZED_BRDF;
    bool is_hit_implicit_light = false;
    if (mtrl.type == ATN_MTRL_EMISSIVE) {
        is_hit_implicit_light = HitImplicitLight(ctxt, isect.objid, isBackfacing, bounce, path, ray_in, rec, mtrl);
    }
    else if (is_toon_mtrl) {
        if (bounce == 0) {
            // "treat toon as a light": the stylised colour is the path's contribution and the path ends
            const v3 toon_bsdf = Toon::bsdf(ctxt, mtrl, path, rec.p, rec.normal, ray_in.dir, rec.u, rec.v, cnt);
            path.contrib += path.throughput * toon_bsdf * albedo.xyz();
            path.is_terminated = true;
        }
        is_hit_implicit_light = true;
    }
    if (is_hit_implicit_light) {
        if (is_toon_mtrl && (bounce > 0 || ctxt.d->enable_shadowray_base_stylized_shadow)) {
            // deeper in the path a toon surface is its plain base material.  toon_type is Diffuse or Specular, so the
            // comparison with ToonSpecular below is always false: an ideal mirror that NEE treats as non-singular
            mtrl.type = mtrl.toon.toon_type == ATN_MTRL_DIFFUSE ? ATN_MTRL_DIFFUSE : ATN_MTRL_SPECULAR;
            const bool sing = mtrl.toon.toon_type == ATN_MTRL_TOON_SPECULAR;
            mtrl.attrib = (mtrl.attrib & ~(uint32_t)ATN_MTRL_ATTR_SINGULAR) | (sing ? ATN_MTRL_ATTR_SINGULAR : 0u);
        }
        else return;
    }

    if (!attr_translucent(mtrl) && isBackfacing) orienting_normal = -orienting_normal;

    // material::applyNormal (pathtracing.cpp:181-188): normal map, or CarPaint's flake normal + shared random number
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

// ShadeMiss, pathtracing_impl.h:112-176
inline void ShadeMiss(int32_t ix, int32_t iy, int32_t width, int32_t height, int32_t bounce,
    const Scene& ctxt, const atn_camera_param& camera, PathState& path, const Ray& ray)
{
    if (!path.is_terminated && !path.isHit) {
        v3 dir = ray.dir;
        if (bounce == 0) {
            float s = ix / (float)(width);
            float t = iy / (float)(height);
            dir = PinholeSample(camera, s, t).dir;
        }
        v4 emit = Background_SampleFromRay(dir, ctxt.cfg().bg, ctxt);
        float misW = 1.0f;
        if (bounce == 0 || (bounce == 1 && path.is_singular)) {
        }
        else {
            float pdfLight = IBL_samplePdf(emit.xyz(), ctxt.cfg().bg.avgIllum);
            if (sampling_options().ibl_importance && ctxt.cfg().bg.envmap_tex_idx >= 0 && ctxt.cfg().bg.enable_env_map)
                pdfLight = IBL_direction_pdf(ctxt, dir);
            misW = path.pdfb / (pdfLight + path.pdfb);
        }
        // ApplyAlphaBlend: transmission(1) * c + throughput(0)
        v3 contrib = 1.0F * (misW * emit).xyz() + v3(0.0F);
        contrib *= path.throughput;
        path.contrib += contrib;
        path.is_terminated = true;
    }
}

// PathTracing::radiance, pathtracing.cpp:22-89
inline void radiance(PathState& path, Ray& ray, ShadowRay& shadow_ray, int32_t ix, int32_t iy,
    int32_t width, int32_t height, const Scene& ctxt, const atn_camera_param& camera,
    int32_t maxDepth, int32_t rrDepth, PathCounters* cnt)
{
    int32_t depth = 0;
    while (depth < maxDepth) {
        bool willContinue = true;
        Isect isect;
        path.isHit = false;
        if (cnt) cnt->closest_rays++;
        bool is_hit = TraverseClosest(isect, ctxt, ray, EPS, INF, cnt ? &cnt->trav : nullptr);
        if (is_hit) {
            path.isHit = true;
            shade(path, ctxt, ray, shadow_ray, isect, rrDepth, depth, cnt);
            HitShadowRay(ctxt, path, shadow_ray, isect.mtrlid >= 0 ? ctxt.GetMaterial(isect.mtrlid).stencil_type : 0, cnt);
            willContinue = !path.is_terminated;
        }
        else {
            ShadeMiss(ix, iy, width, height, depth, ctxt, camera, path, ray);
            willContinue = false;
        }
        if (!willContinue) break;
        depth++;
    }
}

inline bool isInvalidColor(const v3& v)     // renderer/renderer.h:58-68
{
    bool b = std::isnan(v.x) || std::isinf(v.x) || std::isnan(v.y) || std::isinf(v.y) || std::isnan(v.z) || std::isinf(v.z);
    if (!b) b = (v.x < 0 || v.y < 0 || v.z < 0);
    return b;
}

} // namespace orc
