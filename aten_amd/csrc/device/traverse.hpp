// Stackless two-level (TLAS -> BLAS) hit/miss-link walk + Moeller-Trumbore, one ray per lane.
// Decision-for-decision the walk of aten::ThreadedBvhTraverser<true>::Traverse<Closest>
// (src/libaten/accelerator/threaded_bvh_traverser.h:98-304) over the device node records of
// scene_dev.hpp; box test = aabb::hit (src/libaten/math/aabb.h:62-86), triangle test =
// intersectTriangle (src/libaten/math/intersect.h:45-90) + triangle::hit (geometry/triangle.h:40-67).
#pragma once
#include "scene_dev.hpp"

namespace atn {

struct Hit {
    float t;
    int32_t objid;      // instance object id (TLAS leaf), -1 = miss
    int32_t tri;        // global triangle id
    float a, b;         // barycentrics
    int32_t meshid;     // TLAS-leaf mesh id remembered for `prim.mesh_id < 0`
};

struct TravCounters { uint32_t nodes, tris; };

// Per-ray constants of aabb::hit: invdir = 1 / (dir + 1e-6), oxinvdir = -org * invdir.
// The reference recomputes them at every node from the same inputs; hoisting is value-identical.
struct RaySlab { f3 org, dir, invdir, oxinvdir; };

ATN_DEV void slab_setup(RaySlab& s, const f3& org, const f3& dir)
{
    s.org = org; s.dir = dir;
    s.invdir = 1.0F / (dir + 1e-6F);
    s.oxinvdir = (-org) * s.invdir;
}

ATN_DEV bool slab_hit(const RaySlab& s, const f3& bmin, const f3& bmax, float t_min, float t_max)
{
    const f3 f = bmax * s.invdir + s.oxinvdir;
    const f3 n = bmin * s.invdir + s.oxinvdir;
    const f3 tmx = mk3(smax(f.x, n.x), smax(f.y, n.y), smax(f.z, n.z));
    const f3 tmn = mk3(smin(f.x, n.x), smin(f.y, n.y), smin(f.z, n.z));
    const float t1 = smin(min3(tmx), t_max);
    const float t0 = smax(max3(tmn), t_min);
    return t0 <= t1;
}

template <bool COUNT>
ATN_DEV bool traverse_closest(Hit& hit, const DevScene& sc, const f3& org, const f3& dir,
                              float t_min, float t_max, TravCounters* cnt)
{
    t_min = sc.bvh_hit_min > 0 ? sc.bvh_hit_min : t_min;

    hit.t = kInf; hit.objid = -1; hit.tri = -1; hit.a = 0.0F; hit.b = 0.0F; hit.meshid = -1;

    RaySlab ray;
    slab_setup(ray, org, dir);

    int32_t nodeid = 0;
    int32_t objid = -1, meshid = -1;
    int32_t top_hit = -1, top_miss = -1;
    const float4* __restrict__ nodes = sc.nodes;

    while (nodeid >= 0) {
        const float4 q0 = nodes[3 * nodeid + 0];
        const float4 q1 = nodes[3 * nodeid + 1];
        if (COUNT) cnt->nodes++;
        bool is_hit;
        int32_t next_hit, next_miss;

        if (q0.w == kTagInner) {
            is_hit = slab_hit(ray, mk3(q0), mk3(q1), t_min, t_max);
            next_hit = nodeid + 1;
            next_miss = (int32_t)q1.w;
        }
        else if (q0.w >= 0.0F) {
            // triangle leaf: v0, e1, e2 embedded
            const float4 q2 = nodes[3 * nodeid + 2];
            if (COUNT) cnt->tris++;
            const f3 e1 = mk3(q1), e2 = mk3(q2);
            const f3 r = ray.org - mk3(q0);
            const f3 u = cross(ray.dir, e2);
            const f3 v = cross(r, e1);
            const float inv = 1.0F / dot(u, e1);
            const float t = dot(v, e2) * inv;
            const float beta = dot(u, r) * inv;
            const float gamma = dot(v, ray.dir) * inv;
            const bool isect = ((beta >= 0.0F && beta <= 1.0F) && (gamma >= 0.0F && gamma <= 1.0F)
                && (beta + gamma <= 1.0F) && t >= 0.0F);
            is_hit = isect && (t < kInf);                       // triangle::hit against isect_tmp.t = INF
            const bool accept = (t_min < (is_hit ? t : kInf)) && is_hit && (t < hit.t);
            if (accept) {
                hit.t = t; hit.a = beta; hit.b = gamma;
                hit.objid = objid; hit.tri = (int32_t)q0.w; hit.meshid = meshid;
                t_max = t;
            }
            next_hit = next_miss = (int32_t)q1.w;
        }
        else if (q0.w == kTagTlasNested) {
            objid = __float_as_int(q0.x);
            const int32_t w2l = __float_as_int(q0.y);
            meshid = __float_as_int(q1.x);
            top_hit = __float_as_int(q1.y);
            top_miss = __float_as_int(q1.z);
            if (w2l >= 0) {
                // mat4::applyRay (mat4.h:223-235): the ray(org, dir) constructor re-normalises dir
                m4 m;
                m.r0 = sc.matrices[w2l + 0]; m.r1 = sc.matrices[w2l + 1];
                m.r2 = sc.matrices[w2l + 2]; m.r3 = sc.matrices[w2l + 3];
                const f3 o = m4_apply(m, org);
                const f3 d = normalize(m4_applyXYZ(m, dir));
                slab_setup(ray, o, d);
            }
            else {
                slab_setup(ray, org, dir);
            }
            is_hit = true;
            next_hit = __float_as_int(q0.z);    // BLAS root
            next_miss = top_miss;
        }
        else {
            // TLAS leaf without nested tree: nothing is tested (threaded_bvh_traverser.h:146-219)
            is_hit = false;
            next_hit = next_miss = (int32_t)q1.w;
        }

        nodeid = is_hit ? next_hit : next_miss;

        if (nodeid < 0) {
            // leave the bottom layer (or finish the top layer: top_* are -1 there)
            nodeid = is_hit ? top_hit : top_miss;
            top_hit = -1; top_miss = -1;
            slab_setup(ray, org, dir);
        }
    }
    return hit.objid >= 0;
}

} // namespace atn
