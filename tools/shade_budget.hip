// Static instruction budget of k_shade's ingredients: one probe kernel per ingredient (inputs from memory, result stored), compiled
// with the product's flags; tools/shade_budget.py counts the VALU / SALU / memory instructions of each probe in the ISA and subtracts
// the empty probe.  Not part of the library.
#include <hip/hip_runtime.h>
#include "../include/aten_amd.h"
#define ATN_TEMPLATES_ONLY 1    // (templates and helpers only: none of kernels.hpp's own __global__ functions)
#include "../aten_amd/csrc/device/kernels.hpp"
using namespace atn;

#define PROBE(name, ...) \
    extern "C" __global__ void probe_##name(DevScene sc, const float4* __restrict__ in, float4* __restrict__ out) { \
        const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; \
        const float4 a = in[4 * t], b = in[4 * t + 1], c = in[4 * t + 2], d = in[4 * t + 3]; \
        float4 r = make_float4(0, 0, 0, 0); \
        __VA_ARGS__ \
        out[t] = r; }

PROBE(empty, { r = make_float4(a.x, b.x, c.x, d.x); })
PROBE(cmj_next, { Cmj s; s.idx = __float_as_uint(a.x); s.dim = __float_as_uint(a.y); s.scramble = __float_as_uint(a.z); r.x = cmj_next(s) + b.x + c.x + d.x; })
PROBE(cmj_next2, { Cmj s; s.idx = __float_as_uint(a.x); s.dim = __float_as_uint(a.y); s.scramble = __float_as_uint(a.z); r.x = cmj_next(s); r.y = cmj_next(s) + b.x + c.x + d.x; })
PROBE(normalize, { const f3 n = normalize(mk3(a)); r = make_float4(n.x, n.y, n.z, b.x + c.x + d.x); })
PROBE(divide, { r.x = a.x / a.y + b.x + c.x + d.x; })
PROBE(sqrt, { r.x = sqrtf(a.x) + b.x + c.x + d.x; })
PROBE(sincos, { r.x = sinf(a.x); r.y = cosf(a.x) + b.x + c.x + d.x; })
PROBE(atanf, { r.x = atanf(a.x) + b.x + c.x + d.x; })
PROBE(tangent, { f3 t3, b3; tangent_coordinate(mk3(a), t3, b3); r = make_float4(t3.x + b3.x, t3.y + b3.y, t3.z + b3.z, b.x + c.x + d.x); })
PROBE(ray_offset, { const f3 o = ray_offset(mk3(a), mk3(b)); r = make_float4(o.x, o.y, o.z, c.x + d.x); })
PROBE(ggx_dir, { const f3 o = ggx_dir(c.x, c.y, c.z, mk3(a), mk3(b)); r = make_float4(o.x, o.y, o.z, d.x); })
PROBE(ggx_pdf, { r.x = ggx_pdf(d.x, mk3(a), mk3(b), mk3(c)); })
PROBE(ggx_brdf, { const f3 o = ggx_brdf(d.x, d.y, mk3(a), mk3(b), mk3(c)); r = make_float4(o.x, o.y, o.z, 0); })
PROBE(ggx_pdf_brdf, { r.x = ggx_pdf(d.x, mk3(a), mk3(b), mk3(c)); const f3 o = ggx_brdf(d.x, d.y, mk3(a), mk3(b), mk3(c)); r.y = o.x; })
PROBE(ggx_sample_all, { const f3 o = ggx_dir(d.z, d.w, d.x, mk3(b), mk3(a)); r.x = ggx_pdf(d.x, mk3(a), mk3(b), o); const f3 q = ggx_brdf(d.x, d.y, mk3(a), mk3(b), o); r.y = q.x + o.x + o.y + o.z + c.x; })
PROBE(ggx_lambda, { r.x = ggx_lambda(d.x, mk3(a), mk3(b)) + c.x; })
PROBE(diffuse_dir, { const f3 o = diffuse_dir(mk3(a), c.x, c.y); r = make_float4(o.x, o.y, o.z, b.x + d.x); })
PROBE(sample_texture, { r = sample_texture(sc, __float_as_int(a.x), a.y, a.z, b); r.w += c.x + d.x; })
PROBE(evaluate_hit, { HitRec rec; evaluate_hit(rec, sc, __float_as_int(a.x), __float_as_int(a.y), a.z, a.w); r = make_float4(rec.p.x + rec.normal.x + rec.u, rec.p.y + rec.normal.y + rec.v, rec.p.z + rec.normal.z + rec.area, b.x + c.x + d.x); })
PROBE(apply_normal_map, { const f3 o = apply_normal_map(sc, __float_as_int(a.x), mk3(b), a.y, a.z); r = make_float4(o.x, o.y, o.z, c.x + d.x); })
PROBE(background, { r = background_sample(sc, mk3(a)); r.w += b.x + c.x + d.x; })
PROBE(sample_light, { Cmj s; s.idx = __float_as_uint(a.x); s.dim = __float_as_uint(a.y); s.scramble = __float_as_uint(a.z); LightSample ls; sample_light(ls, sc.lights[__float_as_int(a.w)], sc, mk3(b), mk3(c), s);
    r = make_float4(ls.pos.x + ls.dir.x + ls.nml.x + ls.color.x, ls.pos.y + ls.dir.y + ls.nml.y + ls.color.y, ls.pos.z + ls.dir.z + ls.nml.z + ls.color.z, ls.dist + ls.pdf + __uint_as_float(ls.attrib) + d.x); })
PROBE(radiance_nee_core, { LightSample ls; ls.pos = mk3(a); ls.dir = mk3(b); ls.nml = mk3(c); ls.color = mk3(d); ls.dist = a.w; ls.pdf = b.w; ls.attrib = __float_as_uint(c.w);
    f3 o = mk3(0.0F); radiance_nee<kMsCore>(o, sc, mk3(in[4 * t + 1]), mk3(in[4 * t + 2]), sc.materials[__float_as_int(d.w)], a.w, b.w, 1.0F, ls, 0, 0.0F); r = make_float4(o.x, o.y, o.z, 0); })
PROBE(sample_material_core, { Cmj s; s.idx = __float_as_uint(a.x); s.dim = __float_as_uint(a.y); s.scramble = __float_as_uint(a.z); MtrlSample ms;
    sample_material<kMsCore>(ms, sc, sc.materials[__float_as_int(a.w)], mk3(b), mk3(c), s, d.x, d.y, 0, 0.0F); r = make_float4(ms.dir.x + ms.bsdf.x, ms.dir.y + ms.bsdf.y, ms.dir.z + ms.bsdf.z, ms.pdf); })
PROBE(sample_material_disney, { Cmj s; s.idx = __float_as_uint(a.x); s.dim = __float_as_uint(a.y); s.scramble = __float_as_uint(a.z); MtrlSample ms;
    sample_material<kMsDisney>(ms, sc, sc.materials[__float_as_int(a.w)], mk3(b), mk3(c), s, d.x, d.y, 0, 0.0F); r = make_float4(ms.dir.x + ms.bsdf.x, ms.dir.y + ms.bsdf.y, ms.dir.z + ms.bsdf.z, ms.pdf); })
PROBE(radiance_nee_disney, { LightSample ls; ls.pos = mk3(a); ls.dir = mk3(b); ls.nml = mk3(c); ls.color = mk3(d); ls.dist = a.w; ls.pdf = b.w; ls.attrib = __float_as_uint(c.w);
    f3 o = mk3(0.0F); radiance_nee<kMsDisney>(o, sc, mk3(in[4 * t + 1]), mk3(in[4 * t + 2]), sc.materials[__float_as_int(d.w)], a.w, b.w, 1.0F, ls, 0, 0.0F); r = make_float4(o.x, o.y, o.z, 0); })
