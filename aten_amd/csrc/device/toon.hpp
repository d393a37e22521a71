// Toon / StylizedBrdf materials (material set kMsToon): material/toon.cpp:88-286,371-445 of the reference.
//
// "Terminated" materials (material.h:583-588): at bounce 0 the path tracer treats such a surface as a light
// (HitTeminatedMaterial, pathtracing_impl.h:482-503) whose colour is Toon::bsdf -- a one-sample NEE towards ONE
// designated NPR target light, with its own visibility test INSIDE the shading step, quantised through a 1-D remap
// texture, times a screen-space shadow texture, plus a rim light -- and the path ends.  Deeper in the path the surface is
// its plain base material (k_shade).  The visibility test is HitTestToTargetLight (pathtracing_impl.h:266-350) run
// inline by the shading lane: these materials are rare and only primary hits pay for it, so it does not get a queue of
// its own; the kernel that contains it is only launched for scenes that have such a material (DevScene::material_set).
#pragma once
#include "shading.hpp"

namespace atn {

// HitTestToTargetLight: closest hit toward the light, visible iff the hit object is the light's object (or the rules
// for infinite / singular lights); hits on alpha-translucent (or, with StencilType::ALWAYS on the shaded surface,
// STENCIL) surfaces are looked through, up to 10 times when alpha blending or the stencil check is on.
ATN_DEV bool toon_visible(const DevScene& sc, const f3& org, const f3& dir, float dist_to_light, const atn_light_param& light,
                          bool surface_stencil_always)
{
    const char* __restrict__ nb = reinterpret_cast<const char*>(sc.nodes);
    const float t_min = sc.bvh_hit_min > 0 ? sc.bvh_hit_min : kEps;
    const int32_t lightobj = (light.type == ATN_LIGHT_AREA && light.arealight_objid >= 0) ? light.arealight_objid : -1;
    int32_t hitobj = lightobj;
    const uint32_t max_lookups = (sc.enable_alpha_blending || surface_stencil_always) ? 10u : 1u;
    f3 o = org;
    const f3 d0 = normalize(dir);       // aten::ray's constructor normalises; a restarted ray is built from it again (:330)
    f3 d = d0;
    for (uint32_t i = 0; i < max_lookups; i++) {
        Walk w;
        walk_start(w, sc, make_float4(o.x, o.y, o.z, dist_to_light - kEps), make_float4(d.x, d.y, d.z, 0.0F), -kInf);
        walk_run<false>(w, sc, nb, t_min, nullptr);
        const bool is_hit = w.hit.objid >= 0;
        if (is_hit) {
            hitobj = w.hit.objid;
            const int32_t mid = triangle_mtrlid(sc, w.hit.tri);
            const uint32_t mattr = mid >= 0 ? sc.materials[mid].attrib : 0u;
            bool ignore = surface_stencil_always && (mattr & kAttrStencilStencil);
            if (ignore || (mattr & kAttrMaybeAlpha)) {
                HitRec rec;
                evaluate_hit(rec, sc, w.hit.objid, w.hit.tri, w.hit.a, w.hit.b);
                if (mattr & kAttrMaybeAlpha) {
                    const DevMaterial& hm = sc.materials[mid];
                    const float4 albedo = sample_texture(sc, hm.albedoMap, rec.u, rec.v, make_float4(1.0F, 1.0F, 1.0F, 1.0F));
                    if (albedo.w * hm.baseColor.w < 1.0F) ignore = true;
                }
                if (ignore) {
                    const bool is_same_facing = dot(rec.normal, d0) > 0.0F;
                    o = ray_offset(rec.p, is_same_facing ? rec.normal : -rec.normal);
                    d = normalize(d0);
                    continue;
                }
            }
        }
        if (hitobj == lightobj) return true;
        if (light.attrib & ATN_LIGHT_ATTR_INFINITE) return !is_hit;
        if (light.attrib & ATN_LIGHT_ATTR_SINGULAR) return w.hit.t > dist_to_light;
        return false;
    }
    return false;
}

ATN_DEV float bezier_smoothstep(float edge0, float edge1, float mid, float t, float s)     // toon.cpp:222-239
{
    if (t <= edge0) return 0.0F;
    else if (t >= edge1) return 1.0F;
    t = (t - edge0) / (edge1 - edge0);
    t *= s;
    const float B0 = 0.0F, B1 = mid, B2 = 1.0F;
    return (((B0 - 2 * B1) + B2) * t) * t + ((-2 * B0 + 2 * B1) * t) + B0;
}

// Toon::bsdf.  `m` is the surface's material (type Toon or StylizedBrdf), mtrl_id its slot.
ATN_DEV f3 toon_bsdf(const DevScene& sc, const DevMaterial& m, int32_t mtrl_id, Cmj& smp, const f3& hit_pos, const f3& normal,
                     const f3& wi, float u, float v, int32_t screen_x, int32_t screen_y)
{
    const atn_toon_param& tp = sc.toon[mtrl_id];
    f3 toon_term = mk3(0.0F);
    float remap_v = 1.0F;
    if (tp.target_light_idx >= 0 && tp.target_light_idx < sc.n_npr_lights) {
        const atn_light_param& light = sc.npr_lights[tp.target_light_idx];
        LightSample ls;
        sample_light(ls, light, sc, hit_pos, normal, smp);
        bool lit = true;
        if (tp.will_receive_shadow)
            lit = toon_visible(sc, ray_offset(hit_pos, normal), ls.dir, ls.dist, light, (m.attrib & kAttrStencilAlways) != 0);
        // {Toon, StylizedBrdf}::ComputeBRDF: NEE of the base material (throughput 1, light-selection pdf 1)
        f3 radiance = mk3(0.0F);
        float pdf = 1.0F;
        if (lit) {
            DevMaterial base = m;
            base.type = tp.toon_type == ATN_MTRL_DIFFUSE ? ATN_MTRL_DIFFUSE : ATN_MTRL_TOON_SPECULAR;
            float nee_weight = 0.0F;
            f3 res;
            if (radiance_nee<kMsToon>(res, sc, wi, normal, base, u, v, 1.0F, ls, mtrl_id, 0.0F, &nee_weight)) {
                radiance = res;
                pdf = 1.0F / nee_weight;
            }
        }
        if (m.type == ATN_MTRL_TOON) {
            float lum_y = sclamp(luminance(radiance.x, radiance.y, radiance.z), 0.0F, 1.0F);
            lum_y = sclamp(powf(lum_y, 1.0F / 2.2F), 0.0F, 1.0F);
            toon_term = mk3(sample_texture(sc, tp.remap_texture, lum_y, 0.5F, make_float4(1.0F, 1.0F, 1.0F, 1.0F)));
            remap_v = lum_y;
        }
        else {
            constexpr float W_MIN = 0.01F;
            const float y = dot(mk3(0.212639F, 0.71517F, 0.0721926F), radiance);       // color::sRGBtoXYZ(...).y
            const float weight = smax(y, W_MIN);
            const float y_min = smax(0.0F, smin(tp.stylized_y_min, tp.stylized_y_max));
            const float y_max = smax(tp.stylized_y_min, tp.stylized_y_max);
            float rv;
            if (y_max <= y) rv = 1.0F;
            else if (y <= y_min) rv = 0.0F;
            else rv = (y - y_min) / (y_max - y_min);
            const f3 remap = mk3(sample_texture(sc, tp.remap_texture, rv, 0.5F, make_float4(radiance.x, radiance.y, radiance.z, 0.0F)));
            toon_term = (weight * remap) * pdf;
            remap_v = rv;
        }
    }
    if (tp.stylized_shadow.enable) {
        float shadow = 1.0F;
        if (sc.screen_shadow && sc.ss_w > 0 && sc.ss_h > 0) shadow = sc.screen_shadow[(uint32_t)(screen_y * sc.ss_w + screen_x)];
        if (remap_v >= tp.stylized_shadow.threshold) shadow = 1.0F;
        else shadow = smin(smax((shadow * (remap_v + tp.stylized_shadow.offset)) * tp.stylized_shadow.scale, shadow), 1.0F);
        toon_term = toon_term * shadow;
    }
    // ComputeRimLight, toon.cpp:243-286
    f3 rim = mk3(0.0F);
    if (tp.rim_light.enable) {
        const float NdotV = dot(-wi, normal);
        if (NdotV > 0) {
            const float r = bezier_smoothstep(1.0F - tp.rim_light.width, 1.0F, (1 - tp.rim_light.softness) * 0.5F, 1 - NdotV, tp.rim_light.spread);
            rim = r * mk3(tp.rim_light.color[0], tp.rim_light.color[1], tp.rim_light.color[2]);
        }
    }
    return toon_term + rim;
}

} // namespace atn
