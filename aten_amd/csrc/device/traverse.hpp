// Stackless two-level (TLAS -> BLAS) hit/miss-link walk + Moeller-Trumbore.
//
// Per ray this is decision-for-decision the walk of
// aten::ThreadedBvhTraverser<true>::Traverse<Closest>
// (src/libaten/accelerator/threaded_bvh_traverser.h:98-304) over the device node records of
// scene_dev.hpp; box test = aabb::hit (src/libaten/math/aabb.h:62-86), triangle test =
// intersectTriangle (src/libaten/math/intersect.h:45-90) + triangle::hit (geometry/triangle.h:40-67).
//
// The trace kernels are bound by wave-level VALU issue and vector-memory instruction count, not by
// bytes (rocprof), so the loop is written to issue as little as possible:
//   * links are byte offsets with the target's node type in the low bits: no index arithmetic, no
//     float<->int conversion, and the loads use a scalar base + 32-bit vector offset;
//   * the two 16-byte halves every node kind needs are loaded once, before the type branch;
//   * hardware min/max for the slab test whenever the ray's 1/dir is finite (see slab_hit_fast);
//   * the world-space slab constants are kept, so leaving a nested tree costs moves, not divides.
// None of this changes a ray's own operation sequence, so results stay bit-identical.
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

// nodes / tris: the thread's totals (COUNT instantiations only); ray_*: the same for the walk in progress, handed to
// Job::cost when the walk finishes (the per-pixel cost map, atn_download_path_cost)
struct TravCounters { uint32_t nodes, tris, ray_nodes, ray_tris; };

// Per-ray constants of aabb::hit: invdir = 1 / (dir + 1e-6), oxinvdir = -org * invdir.
// The reference recomputes them at every node from the same inputs; hoisting is value-identical.
struct RaySlab { f3 org, dir, invdir, oxinvdir; bool finite; };

ATN_DEV bool is_finite3(const f3& v)
{
    return (fabsf(v.x) <= kInf) && (fabsf(v.y) <= kInf) && (fabsf(v.z) <= kInf);   // false for NaN and inf
}

ATN_DEV void slab_setup(RaySlab& s, const f3& org, const f3& dir)
{
    s.org = org; s.dir = dir;
    s.invdir = 1.0F / (dir + 1e-6F);
    s.oxinvdir = (-org) * s.invdir;
    s.finite = is_finite3(s.invdir) && is_finite3(s.oxinvdir);
}

// aabb::hit with the host's std::max / std::min (select form: NaN-sensitive, math.h:148-180).
ATN_DEV bool slab_hit_exact(const RaySlab& s, const f3& bmin, const f3& bmax, float t_min, float t_max)
{
    const f3 f = bmax * s.invdir + s.oxinvdir;
    const f3 n = bmin * s.invdir + s.oxinvdir;
    const f3 tmx = mk3(smax(f.x, n.x), smax(f.y, n.y), smax(f.z, n.z));
    const f3 tmn = mk3(smin(f.x, n.x), smin(f.y, n.y), smin(f.z, n.z));
    const float t1 = smin(min3(tmx), t_max);
    const float t0 = smax(max3(tmn), t_min);
    return t0 <= t1;
}

// Same test with hardware min/max.  With finite invdir / oxinvdir and finite boxes no NaN can
// arise (finite * finite + finite is finite or +-inf), and for non-NaN operands v_max/v_min equal
// the select form except for the sign of a zero, which `t0 <= t1` cannot see.
// The instructions are spelled out: through fminf/fmaxf the compiler prepends a canonicalising
// `v_max_f32 x, x, x` to every operand (IEEE mode, operands not provably quiet), 8 extra VALU per node.
ATN_DEV float hw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
ATN_DEV float hw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
ATN_DEV float hw_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
ATN_DEV float hw_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

ATN_DEV bool slab_hit_fast(const RaySlab& s, const f3& bmin, const f3& bmax, float t_min, float t_max)
{
    const f3 f = bmax * s.invdir + s.oxinvdir;
    const f3 n = bmin * s.invdir + s.oxinvdir;
    const float t1 = hw_min3(hw_min(hw_max(f.x, n.x), hw_max(f.y, n.y)), hw_max(f.z, n.z), t_max);
    const float t0 = hw_max3(hw_max(hw_min(f.x, n.x), hw_min(f.y, n.y)), hw_min(f.z, n.z), t_min);
    return t0 <= t1;
}

#ifndef ATN_TRACE_BLOCK
#define ATN_TRACE_BLOCK 256
#endif
constexpr int kTraceBlock = ATN_TRACE_BLOCK;        // threads per block of the persistent (refill) trace kernels

ATN_DEV float4 ld16(const char* base, uint32_t byte_off)
{
    return *reinterpret_cast<const float4*>(base + byte_off);
}

// A 16-byte quarter of a record through a BUFFER load (raw buffer over the node image).  Same path through the L1 as a global load,
// but an intrinsic the optimiser cannot split: with plain loads the leaf step's second quarter -- whose four components are used
// by different branches (triangle leaf: e1 and the next link; TLAS leaf: mesh id and the two top links) -- was fetched by TWO
// instructions (dwordx3 before the branch + dword inside it), and a wave load costs the L1 ~16 clocks however narrow it is (DESIGN.md
// section 6): 4 loads per leaf step instead of 3.
typedef int atn_v4i __attribute__((ext_vector_type(4)));
ATN_DEV __amdgpu_buffer_rsrc_t node_rsrc(const DevScene& sc)
{
    return __builtin_amdgcn_make_buffer_rsrc((void*)sc.nodes, 0, (int)sc.node_bytes, 0x00020000);     // raw, 32-bit data format
}
ATN_DEV float4 ld16_buf(__amdgpu_buffer_rsrc_t r, uint32_t byte_off)
{
    const atn_v4i v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}

// The block's dynamic LDS: the WHOLE node image of a small scene (LDSN).
extern __shared__ float4 atn_dyn_lds[];
// a 16-byte quarter of a node record: from global memory, or -- LDSN -- from the block's LDS copy of the node image
// (indexing the __shared__ array keeps the access in the LDS address space: ds_read_b128, not a flat load)
template <bool LDSN>
ATN_DEV float4 ldn(const char* base, uint32_t byte_off)
{
    if constexpr (LDSN) return atn_dyn_lds[byte_off >> 4];
    else return ld16(base, byte_off);
}

// the block's LDS copy of a small scene's walk data: the node image, then the matrix rows (LDSN)
ATN_DEV void lds_scene_copy(const DevScene& sc)
{
    const uint32_t n16 = sc.node_bytes >> 4;
    for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) atn_dyn_lds[i] = sc.nodes[i];
    for (uint32_t i = threadIdx.x; i < sc.mtx_quads; i += blockDim.x) atn_dyn_lds[n16 + i] = sc.matrices[i];
    __syncthreads();
}
template <bool LDSN>
ATN_DEV float4 ldm(const DevScene& sc, int32_t row)
{
    if constexpr (LDSN) return atn_dyn_lds[(sc.node_bytes >> 4) + (uint32_t)row];
    else return sc.matrices[row];
}

// One Moeller-Trumbore test against a triangle-leaf record (q0, q1, q2) -- intersectTriangle (math/intersect.h:45-90) +
// triangle::hit (geometry/triangle.h:40-67) + the traverser's acceptance (threaded_bvh_traverser.h:236-262).
// Returns triangle::hit's result; `accept` = the hit became the ray's closest one.
ATN_DEV bool leaf_test(const RaySlab& ray, const float4& q0, const float4& q1, const float4& q2, float t_min,
                       Hit& hit, float& t_max, int32_t objid, int32_t meshid, bool& accept, float& t_out)
{
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
    const bool is_hit = isect && (t < kInf);                        // triangle::hit against isect_tmp.t = INF
    accept = (t_min < (is_hit ? t : kInf)) && is_hit && (t < hit.t);
    if (accept) {
        hit.t = t; hit.a = beta; hit.b = gamma;
        hit.objid = objid; hit.tri = __float_as_int(q0.w); hit.meshid = meshid;
        t_max = t;
    }
    t_out = t;
    return is_hit;
}

// Job interface (all jobs of a launch share t_min):
//   float t_min
//   void fetch(uint32_t j, float4& a, float4& b, float& stop_t)   a = {org.xyz, t_max}, b = {dir.xyz, payload bits}
//        stop_t: the walk may stop at the first ACCEPTED hit whose t <= stop_t.  -inf = plain closest-hit walk;
//        +inf = "any hit" (only finish()'s is_hit is used); a finite value = the caller only needs to know whether
//        the closest hit is nearer than stop_t (shadow rays toward point / spot lights).  This is exact, not an
//        approximation: up to that hit the closest-hit walk is the same walk, its final hit can only be nearer,
//        and when no such hit is accepted the walk runs to its end and reports the exact closest hit.
//   bool finish(uint32_t payload, const Hit& h, bool is_hit, float4& a, float4& b, float& stop_t)
//        true = the job is not over: the lane walks the ray (a, b, stop_t) next (a shadow ray restarting behind an
//        ignored surface)

// Per-lane state of one walk.  A lane is idle <=> node == kLinkEnd at the top of an iteration.
struct Walk {
    RaySlab wray, ray;      // world-space ray and the ray of the list being walked (transformed inside a nested tree)
    RaySlab lray;           // IDENT walks only: the ray inside an instance whose W2L is the identity matrix (walk_start)
    bool pending;           // DEFER walks only: the walk is over, Job::finish has not run yet (walk_finish)
    Hit hit;
    float t_max, stop_t;
    uint32_t payload;
    int32_t node, objid, meshid, top_hit, top_miss;
};

// The root an any-hit ray enters a nested tree at (scene_dev.hpp, TLAS leaf `twin` word: distance to the list's first any-hit twin,
// which is the size of the list and of every twin; bit 0: eight twins back to back, one per octant of the ray's direction inside the
// instance -- bit a = dir[a] > 0).  Any twin is a threading of the same tree and an any-hit walk's answer is the same in all of
// them: which one a ray takes only decides how soon it meets its occluder.
ATN_DEV int32_t anyhit_root(int32_t root, int32_t twin, const f3& d)
{
    const int32_t delta = twin & ~15;
    const uint32_t octant = (d.x > 0.0F ? 1u : 0u) | (d.y > 0.0F ? 2u : 0u) | (d.z > 0.0F ? 4u : 0u);
    return root + delta + (int32_t)((twin & 1) ? octant : 0u) * delta;
}

// IDENT (the plain walk over an LDS copy of a small scene): instances whose W2L is bit for bit the identity matrix (TLAS-leaf flag
// kTlasIdentity, set at upload) all see the SAME local ray -- mat4::applyRay(I, ray): the origin through the matrix product, the
// direction re-normalised, NOT the world ray -- so it is computed here once per ray, with every lane of the wave taking part, instead
// of once per instance entered by whichever lanes stand on a TLAS leaf (the Cornell box: 8 identity instances, ~3 entered per ray;
// the ~130-instruction TLAS-leaf block was 40 % of that walk's VALU).  Same operations on the same operands: same bits.
// DIRECT START (DevScene::root_direct, wave-uniform): when the top layer is one leaf, every ray's first visit is that leaf, and a
// lane standing on a leaf sits out a whole burst before the leaf step lets it into the nested tree -- 5 of a ray's ~70 lane slots.
// The leaf's record travels in the kernel arguments instead and the walk starts INSIDE the nested tree: the leaf step's
// operations (mat4::applyRay, slab constants of the local ray) are done here, the world-space slab constants -- never used: there
// is no top-layer box to test and the walk ends when the nested list ends -- are not computed at all.  The visit is still counted.
template <bool IDENT = false, bool COUNT = false>
ATN_DEV void walk_start(Walk& w, const DevScene& sc, const float4& a, const float4& b, float stop_t, TravCounters* cnt = nullptr)
{
    w.t_max = a.w;
    w.stop_t = stop_t;
    w.payload = __float_as_uint(b.w);
    w.hit.t = kInf; w.hit.objid = -1; w.hit.tri = -1; w.hit.a = 0.0F; w.hit.b = 0.0F; w.hit.meshid = -1;
    if (sc.root_direct) {
        if (COUNT) { cnt->nodes++; cnt->ray_nodes++; }
        w.wray.org = mk3(a); w.wray.dir = mk3(b); w.wray.invdir = mk3(0.0F); w.wray.oxinvdir = mk3(0.0F); w.wray.finite = true;
        if (sc.root_w2l >= 0) {
            m4 m;       // the instance's W2L rows travel in the kernel arguments (scalar registers): no loads here
            m.r0 = make_float4(sc.root_m[0], sc.root_m[1], sc.root_m[2], sc.root_m[3]);
            m.r1 = make_float4(sc.root_m[4], sc.root_m[5], sc.root_m[6], sc.root_m[7]);
            m.r2 = make_float4(sc.root_m[8], sc.root_m[9], sc.root_m[10], sc.root_m[11]);
            m.r3 = make_float4(0.0F, 0.0F, 0.0F, 1.0F);        // (not read: m4_apply / m4_applyXYZ use rows 0..2)
            const f3 o = m4_apply(m, w.wray.org);
            const f3 d = normalize(m4_applyXYZ(m, w.wray.dir));
            slab_setup(w.ray, o, d);
        }
        else {
            slab_setup(w.ray, w.wray.org, w.wray.dir);
        }
        w.lray = w.ray;
        w.node = sc.root_blas;
        if (sc.root_twin != 0 && stop_t == kInf) w.node = anyhit_root(sc.root_blas, sc.root_twin, w.ray.dir);       // any-hit rays walk a twin (scene_dev.hpp)
        w.objid = sc.root_objid; w.meshid = sc.root_meshid; w.top_hit = kLinkEnd; w.top_miss = kLinkEnd;
        return;
    }
    slab_setup(w.wray, mk3(a), mk3(b));
    w.ray = w.wray;
    w.node = sc.root_link; w.objid = -1; w.meshid = -1; w.top_hit = kLinkEnd; w.top_miss = kLinkEnd;
    if constexpr (IDENT) {
        if (sc.ident_row >= 0) {        // wave-uniform
            m4 m;
            m.r0 = ldm<true>(sc, sc.ident_row + 0); m.r1 = ldm<true>(sc, sc.ident_row + 1);
            m.r2 = ldm<true>(sc, sc.ident_row + 2); m.r3 = ldm<true>(sc, sc.ident_row + 3);
            const f3 o = m4_apply(m, w.wray.org);
            const f3 d = normalize(m4_applyXYZ(m, w.wray.dir));
            slab_setup(w.lray, o, d);
        }
    }
}

#ifndef ATN_INNER_BURST
#define ATN_INNER_BURST 5
#endif
constexpr int kInnerBurst = ATN_INNER_BURST;       // persistent (refill) walk: 3 / 4 / 5 / 6 steps measured, DESIGN.md section 7
#ifndef ATN_SIMPLE_BURST_LDS
#define ATN_SIMPLE_BURST_LDS 2  /* the plain walk over an LDS copy of the scene steps in bursts of this many inner-node steps */
#endif
constexpr int kSimpleBurstLds = ATN_SIMPLE_BURST_LDS;

// both 16-byte halves of a record through ONE address computation (the second load takes an immediate offset)
template <bool LDSN>
ATN_DEV void ld32(const char* base, uint32_t byte_off, float4& a, float4& b)
{
    if constexpr (LDSN) { a = atn_dyn_lds[byte_off >> 4]; b = atn_dyn_lds[(byte_off >> 4) + 1u]; }
    else {
        const float4* p = reinterpret_cast<const float4*>(base + byte_off);
        a = p[0]; b = p[1];
    }
}

// (SPECULATION past triangle leaves -- a lane that reaches a leaf parks it, keeps stepping with the old t_max and is rolled back
// when the leaf step accepts the hit -- was built and measured in r04: byte-equal, lane utilisation 0.37 -> 0.42, and 5-6 %
// SLOWER.  profiles/r04_variants_spec_leaves.txt; the code is in git history, commit "Speculation past
// triangle leaves".)
// A burst of inner-node steps, spelled out (no loop counter, compare and branch per step), with ONE form of the slab test
// (FAST: hardware min/max, valid when every live lane's slab constants are finite; else the select form, valid for all
// inputs).  See walk_iteration.
template <bool COUNT, int BURST, bool FAST, bool LDSN>
ATN_DEV void inner_burst(Walk& w, const char* __restrict__ nb, float t_min, TravCounters* cnt)
{
#pragma unroll
    for (int k = 0; k < BURST; k++) {
        if (w.node >= 0) {      // inner record (scene_dev.hpp: every other link carries the sign bit)
            float4 q0, q1;
            ld32<LDSN>(nb, (uint32_t)w.node, q0, q1);       // no type bits: the link is the byte offset
            if (COUNT) { cnt->nodes++; cnt->ray_nodes++; }
            const bool box = FAST ? slab_hit_fast(w.ray, mk3(q0), mk3(q1), t_min, w.t_max)
                                  : slab_hit_exact(w.ray, mk3(q0), mk3(q1), t_min, w.t_max);
            w.node = __float_as_int(box ? q0.w : q1.w);
        }
    }
}

// One wave iteration of the walk, for every live lane: a BURST of inner-node steps in a tight loop (two 16-byte loads, the
// slab test, the link select -- nothing else), then ONE step for the lanes that stand on a triangle leaf or a TLAS leaf,
// then the lanes whose list ended leave the bottom layer or finish.
// Why: at any moment only ~7 of 64 lanes stand on a leaf (one visit in nine), so a loop that offers every node kind on
// every iteration issues the ~75-instruction triangle block each time for a handful of lanes, and drags the leave /
// finish (and refill) bookkeeping -- ~60 scalar instructions -- through every inner-node step.  Here the triangle block
// runs once per burst with several times the lanes, and the inner-node step is ~30 VALU + ~15 SALU; a lane that reaches a
// leaf waits, masked off, for the end of the burst.
// The slab test takes ONE of its two forms per wave, chosen once per burst: the hardware min/max form when every live
// lane's slab constants are finite (`all_finite`, wave-uniform, refreshed only where rays change), the select form --
// valid for all inputs -- otherwise.
// A ray's own sequence of operations is the reference walk's, so results stay bit-identical.
// The TLAS-leaf step (two matrix products, normalize, slab constants of the local ray: ~130 VALU, three IEEE divides and a square
// root among them) costs the wave the same for one lane as for sixty.  In a scene of several instances (the atrium: 8) some lane
// of a refilled wave stands on a TLAS leaf in practically every iteration, so the block rode along with every burst -- a third
// of the iteration's VALU for ~6 lanes.  The refill walk therefore lets lanes into nested trees only every ATN_TLAS_PERIOD-th
// iteration; they stand for at most PERIOD - 1 iterations.  Which iteration a lane takes a step in never changes what the step
// computes.  (atrium fused trace 5.25 -> 5.06 ms; period 2 / 3 / 4: 5.12 / 5.06 / 5.08; "or as soon as 6 / 12 lanes wait" on top:
// slower -- the ballot is on the critical path of every iteration -- profiles/r04_variants_tlas_period.txt.)  One-instance scenes
// start inside the nested tree (walk_start) and never get here.
#ifndef ATN_TLAS_PERIOD
#define ATN_TLAS_PERIOD 3
#endif
template <bool COUNT, int BURST, class Job, bool LDSN = false, bool DEFER = false>
ATN_DEV void walk_iteration(Walk& w, bool& all_finite, const DevScene& sc, const char* __restrict__ nb, float t_min,
                            const Job& job, TravCounters* cnt, uint32_t iter = 0)
{
    // ---- burst of inner-node steps.  kLinkEnd has the sign bit set like every link to a leaf, so `node >= 0` alone selects
    // the live lanes on inner nodes.  An inner record's hit link is never kLinkEnd (checked at upload; a dead leaf's "hit"
    // link is its miss link): a list that ends here ended on a MISS.
    const bool live = w.node != kLinkEnd;
    if (all_finite) inner_burst<COUNT, BURST, true, LDSN>(w, nb, t_min, cnt);
    else inner_burst<COUNT, BURST, false, LDSN>(w, nb, t_min, cnt);
    bool ended = live && w.node == kLinkEnd;    // this lane's walk left a list in this iteration ...
    bool is_hit = false;                        // ... and this was the result of its last step

    // ---- one step for the lanes on a triangle leaf or a TLAS leaf (both read the record's first two quarters)
    const bool tlas_turn = LDSN || ATN_TLAS_PERIOD <= 1 || (iter % (uint32_t)ATN_TLAS_PERIOD) == 0u;      // wave-uniform
    const bool at_tlas = w.node != kLinkEnd && (w.node & kLinkTypeMask) == kLinkTlasBit && tlas_turn;
    if (w.node != kLinkEnd && w.node < 0 && ((w.node & kLinkLeafBit) || tlas_turn)) {
        const uint32_t off = (uint32_t)w.node & kLinkOffsetMask;
        float4 q0, q1;
        if constexpr (LDSN) { q0 = ldn<true>(nb, off); q1 = ldn<true>(nb, off + 16u); }
        else { const __amdgpu_buffer_rsrc_t rs = node_rsrc(sc); q0 = ld16_buf(rs, off); q1 = ld16_buf(rs, off + 16u); }
        if (COUNT) { cnt->nodes++; cnt->ray_nodes++; }
        if (w.node & kLinkLeafBit) {
            float4 q2;
            if constexpr (LDSN) q2 = ldn<true>(nb, off + 32u);
            else q2 = ld16_buf(node_rsrc(sc), off + 32u);
            if (COUNT) { cnt->tris++; cnt->ray_tris++; }
            bool accept; float t;
            is_hit = leaf_test(w.ray, q0, q1, q2, t_min, w.hit, w.t_max, w.objid, w.meshid, accept, t);
            w.node = __float_as_int(q1.w);      // leaf: hit link == miss link
            if (accept && t <= w.stop_t) { w.node = kLinkEnd; w.top_hit = kLinkEnd; w.top_miss = kLinkEnd; }    // see Job::fetch
            ended = w.node == kLinkEnd;
        }
        else {
            w.objid = __float_as_int(q0.x);
            const int32_t w2l = __float_as_int(q0.y);
            w.meshid = __float_as_int(q1.x);
            w.top_hit = __float_as_int(q1.y);
            w.top_miss = __float_as_int(q1.z);
            if (LDSN && (__float_as_int(q0.w) & kTlasIdentity)) {
                w.ray = w.lray;                 // identity instance: walk_start<true> has the local ray
            }
            else if (w2l >= 0) {
                // mat4::applyRay (mat4.h:223-235): the ray(org, dir) constructor re-normalises dir
                m4 m;
                m.r0 = ldm<LDSN>(sc, w2l + 0); m.r1 = ldm<LDSN>(sc, w2l + 1);
                m.r2 = ldm<LDSN>(sc, w2l + 2); m.r3 = ldm<LDSN>(sc, w2l + 3);
                const f3 o = m4_apply(m, w.wray.org);
                const f3 d = normalize(m4_applyXYZ(m, w.wray.dir));
                slab_setup(w.ray, o, d);
            }
            else {
                w.ray = w.wray;
            }
            is_hit = true;
            // BLAS root link (never kLinkEnd: empty lists are rejected at upload); any-hit rays (stop_t = +inf, Job::fetch): the root of the list's twin
            w.node = __float_as_int(q0.z);
            if (w.stop_t == kInf && __float_as_int(q1.w) != 0) w.node = anyhit_root(w.node, __float_as_int(q1.w), w.ray.dir);
            ended = false;
        }
    }

    // ---- a list ended: leave the bottom layer (top_* are kLinkEnd inside the top layer), or finish
    if (ended) {
        w.node = is_hit ? w.top_hit : w.top_miss;
        w.top_hit = kLinkEnd; w.top_miss = kLinkEnd;
        if (w.node != kLinkEnd) w.ray = w.wray;     // (13 registers: only for the lanes that go on in the top layer)
        if (w.node == kLinkEnd) {
            if constexpr (DEFER) w.pending = true;      // the lane is idle from here on; walk_finish runs with the next refill
            else {
                float4 ra, rb;
                float rstop;
                if (COUNT) { job.cost(w.payload, cnt->ray_nodes, cnt->ray_tris); cnt->ray_nodes = 0; cnt->ray_tris = 0; }
                if (job.finish(w.payload, w.hit, w.hit.objid >= 0, ra, rb, rstop)) walk_start<LDSN, COUNT>(w, sc, ra, rb, rstop, cnt);
            }
        }
    }
    // rays changed in the two blocks above: refresh the wave's slab-form flag (cheap, and only then)
    if (__any(ended || at_tlas) || !all_finite) all_finite = __all(w.node == kLinkEnd || w.ray.finite) != 0;
}

// One ray walked to the end of the top layer, per lane (w set up by walk_start): the walk as a function, for a kernel
// that needs a visibility answer in the middle of something else (device/toon.hpp).
template <bool COUNT>
ATN_DEV void walk_run(Walk& w, const DevScene& sc, const char* __restrict__ nb, float t_min, TravCounters* cnt)
{
    while (w.node != kLinkEnd) {
        const uint32_t off = (uint32_t)w.node & kLinkOffsetMask;
        const float4 q0 = ld16(nb, off);
        const float4 q1 = ld16(nb, off + 16u);
        if (COUNT) { cnt->nodes++; cnt->ray_nodes++; }
        bool is_hit;
        if (w.node >= 0) {
            // inner node, or a dead leaf (both links = its miss link).  An inner record's hit link is never kLinkEnd
            // (checked at upload), so a list can only end here on a miss.
            const bool box = w.ray.finite ? slab_hit_fast(w.ray, mk3(q0), mk3(q1), t_min, w.t_max)
                                          : slab_hit_exact(w.ray, mk3(q0), mk3(q1), t_min, w.t_max);
            w.node = __float_as_int(box ? q0.w : q1.w);
            is_hit = false;
        }
        else if (w.node & kLinkLeafBit) {
            const float4 q2 = ld16(nb, off + 32u);
            if (COUNT) { cnt->tris++; cnt->ray_tris++; }
            bool accept; float t;
            is_hit = leaf_test(w.ray, q0, q1, q2, t_min, w.hit, w.t_max, w.objid, w.meshid, accept, t);
            w.node = __float_as_int(q1.w);      // leaf: hit link == miss link
            if (accept && t <= w.stop_t) { w.node = kLinkEnd; w.top_hit = kLinkEnd; w.top_miss = kLinkEnd; }    // see Job::fetch
        }
        else {
            // TLAS leaf with a nested tree
            w.objid = __float_as_int(q0.x);
            const int32_t w2l = __float_as_int(q0.y);
            w.meshid = __float_as_int(q1.x);
            w.top_hit = __float_as_int(q1.y);
            w.top_miss = __float_as_int(q1.z);
            if (w2l >= 0) {
                // mat4::applyRay (mat4.h:223-235): the ray(org, dir) constructor re-normalises dir
                m4 m;
                m.r0 = sc.matrices[w2l + 0]; m.r1 = sc.matrices[w2l + 1];
                m.r2 = sc.matrices[w2l + 2]; m.r3 = sc.matrices[w2l + 3];
                const f3 o = m4_apply(m, w.wray.org);
                const f3 d = normalize(m4_applyXYZ(m, w.wray.dir));
                slab_setup(w.ray, o, d);
            }
            else {
                w.ray = w.wray;
            }
            is_hit = true;
            w.node = __float_as_int(q0.z);
            if (w.stop_t == kInf && __float_as_int(q1.w) != 0) w.node = anyhit_root(w.node, __float_as_int(q1.w), w.ray.dir);      // BLAS root link (any-hit rays: the twin's)
        }
        if (w.node == kLinkEnd) {
            // leave the bottom layer (top_* are kLinkEnd inside the top layer)
            w.node = is_hit ? w.top_hit : w.top_miss;
            w.top_hit = kLinkEnd; w.top_miss = kLinkEnd;
            w.ray = w.wray;
        }
    }
}

#ifndef ATN_DEFER_FINISH
#define ATN_DEFER_FINISH 1
#endif
// Job::finish for the lanes whose walk ended since the last call (Walk::pending).  A finished lane is idle until the wave
// refills -- 16 idle lanes -- so the refill walk finishes its rays THEN, all of them at once: the finish block (hit record or
// shadow result: ~55 VALU, ~80 SALU, 9 memory instructions) used to ride along with every iteration for the one or two lanes
// whose walk had just ended.  true = some lane's job handed back another ray to walk (a shadow ray behind an ignored surface).
template <bool COUNT, class Job, bool LDSN>
ATN_DEV bool walk_finish(Walk& w, const DevScene& sc, const Job& job, TravCounters* cnt)
{
    bool restarted = false;
    if (w.pending) {
        w.pending = false;
        float4 ra, rb;
        float rstop;
        if (COUNT) { job.cost(w.payload, cnt->ray_nodes, cnt->ray_tris); cnt->ray_nodes = 0; cnt->ray_tris = 0; }
        if (job.finish(w.payload, w.hit, w.hit.objid >= 0, ra, rb, rstop)) { walk_start<LDSN, COUNT>(w, sc, ra, rb, rstop, cnt); restarted = true; }
    }
    return __any(restarted) != 0;
}

// Plain flavour: one ray per lane for the lifetime of its walk, grid-stride over the jobs; every iteration offers every
// node kind.  Without refill a wave lasts as long as its longest ray, so what counts here is the latency of a single
// walk, and making lanes wait at leaves for the end of a burst (walk_iteration) only lengthens it: measured on MI355X
// with the burst form, Cornell 1080p 1.85 -> 2.08 ms per frame and sponza_lod forced onto this flavour 6.24 -> 7.17 ms.
// Used for small trees and small launches, where the refill bookkeeping costs more than the idle lanes it removes.
// LDSN (scenes whose node image is at most kLdsNodesMaxBytes: the Cornell box is 2.9 KB): every block first copies the image
// into its dynamic LDS and the walk reads ALL records from there -- one source, so none of the per-lane selection that
// sank the LDS treelet of the deep trees; a walk step waits for the LDS (~100 clocks) instead of the
// L1 behind a queue of other waves' gathers, and the plain walk is latency-bound by construction.
template <bool COUNT, class Job, bool LDSN = false>
ATN_DEV void trace_simple(const DevScene& sc, uint32_t count, const Job& job, TravCounters* cnt)
{
    const char* __restrict__ nb = reinterpret_cast<const char*>(sc.nodes);
    if constexpr (LDSN) lds_scene_copy(sc);
    const float t_min = sc.bvh_hit_min > 0 ? sc.bvh_hit_min : job.t_min;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < count; j += stride) {
        Walk w;
        float4 a, b;
        float stop_t;
        job.fetch(j, a, b, stop_t);
        if constexpr (LDSN && kSimpleBurstLds > 0) {
            // Over an LDS copy a step waits ~100 clocks, not for the L1 behind other waves' gathers, so what the plain walk pays
            // for is issue: every iteration of the loop below offers every node kind, and the ~75-instruction triangle block
            // and the ~110-instruction TLAS-leaf block (two matrix products, normalize, three IEEE divides) run each time
            // for a handful of lanes.  The burst form of the step (walk_iteration: kSimpleBurstLds inner-node steps, then ONE
            // step for the lanes on a leaf) issues them once per burst: Cornell 1080p trace 1.11 -> 1.00 ms per frame.
            // (From global memory the same form LOSES -- 1.85 -> 2.08 ms, r02 -- there the waiting lanes cost more.)
            // Job::finish runs ONCE per wave, when every lane's walk is over (walk_finish), not in every iteration in which some
            // lane's walk ends -- without refill a finished lane has nothing to do anyway
            walk_start<true, COUNT>(w, sc, a, b, stop_t, cnt);
            w.pending = false;
            bool all_finite = __all(w.ray.finite) != 0;
            for (;;) {
                while (__any(w.node != kLinkEnd))
                    walk_iteration<COUNT, kSimpleBurstLds, Job, LDSN, true>(w, all_finite, sc, nb, t_min, job, cnt);
                if (!walk_finish<COUNT, Job, LDSN>(w, sc, job, cnt)) break;
                all_finite = __all(w.node == kLinkEnd || w.ray.finite) != 0;    // (a shadow ray restarted behind an ignored surface)
            }
            continue;
        }
      restart:
        walk_start<LDSN, COUNT>(w, sc, a, b, stop_t, cnt);
        // (the loop of walk_run, spelled out: as a call the compiler lays the kernel out 4 % slower on Cornell 1080p)
        while (w.node != kLinkEnd) {
            const uint32_t off = (uint32_t)w.node & kLinkOffsetMask;
            const float4 q0 = ldn<LDSN>(nb, off);
            const float4 q1 = ldn<LDSN>(nb, off + 16u);
            if (COUNT) { cnt->nodes++; cnt->ray_nodes++; }
            bool is_hit;
            if (w.node >= 0) {
                // inner node, or a dead leaf (both links = its miss link).  An inner record's hit link is never kLinkEnd
                // (checked at upload), so a list can only end here on a miss.
                const bool box = w.ray.finite ? slab_hit_fast(w.ray, mk3(q0), mk3(q1), t_min, w.t_max)
                                              : slab_hit_exact(w.ray, mk3(q0), mk3(q1), t_min, w.t_max);
                w.node = __float_as_int(box ? q0.w : q1.w);
                is_hit = false;
            }
            else if (w.node & kLinkLeafBit) {
                const float4 q2 = ldn<LDSN>(nb, off + 32u);
                if (COUNT) { cnt->tris++; cnt->ray_tris++; }
                bool accept; float t;
                is_hit = leaf_test(w.ray, q0, q1, q2, t_min, w.hit, w.t_max, w.objid, w.meshid, accept, t);
                w.node = __float_as_int(q1.w);      // leaf: hit link == miss link
                if (accept && t <= w.stop_t) { w.node = kLinkEnd; w.top_hit = kLinkEnd; w.top_miss = kLinkEnd; }    // see Job::fetch
            }
            else {
                // TLAS leaf with a nested tree
                w.objid = __float_as_int(q0.x);
                const int32_t w2l = __float_as_int(q0.y);
                w.meshid = __float_as_int(q1.x);
                w.top_hit = __float_as_int(q1.y);
                w.top_miss = __float_as_int(q1.z);
                if (LDSN && (__float_as_int(q0.w) & kTlasIdentity)) {
                    w.ray = w.lray;             // identity instance: walk_start<true> has the local ray
                }
                else if (w2l >= 0) {
                    // mat4::applyRay (mat4.h:223-235): the ray(org, dir) constructor re-normalises dir
                    m4 m;
                    m.r0 = ldm<LDSN>(sc, w2l + 0); m.r1 = ldm<LDSN>(sc, w2l + 1);
                    m.r2 = ldm<LDSN>(sc, w2l + 2); m.r3 = ldm<LDSN>(sc, w2l + 3);
                    const f3 o = m4_apply(m, w.wray.org);
                    const f3 d = normalize(m4_applyXYZ(m, w.wray.dir));
                    slab_setup(w.ray, o, d);
                }
                else {
                    w.ray = w.wray;
                }
                is_hit = true;
                w.node = __float_as_int(q0.z);
            if (w.stop_t == kInf && __float_as_int(q1.w) != 0) w.node = anyhit_root(w.node, __float_as_int(q1.w), w.ray.dir);      // BLAS root link (any-hit rays: the twin's)
            }
            if (w.node == kLinkEnd) {
                // leave the bottom layer (top_* are kLinkEnd inside the top layer)
                w.node = is_hit ? w.top_hit : w.top_miss;
                w.top_hit = kLinkEnd; w.top_miss = kLinkEnd;
                w.ray = w.wray;
            }
        }
        if (COUNT) { job.cost(w.payload, cnt->ray_nodes, cnt->ray_tris); cnt->ray_nodes = 0; cnt->ray_tris = 0; }
        if (job.finish(w.payload, w.hit, w.hit.objid >= 0, a, b, stop_t)) goto restart;
    }
}

// ---------------------------------------------------------------------------------------------
// Same per-ray walk, but the wave is persistent: it reserves chunks of kFetchChunk jobs with one
// atomicAdd, stages the chunk's rays in LDS with one coalesced burst, and whenever kRefillLanes
// lanes have finished their rays it hands them new ones (ballot + popcount prefix).  Rays visit
// very different numbers of nodes (sponza_lod: mean 56, long tail); without refill a wave idles
// ~60 % of its lane-iterations waiting for its longest ray.
#ifndef ATN_REFILL_LANES
#define ATN_REFILL_LANES 16
#endif
constexpr uint32_t kRefillLanes = ATN_REFILL_LANES;
#ifndef ATN_FETCH_CHUNK
#define ATN_FETCH_CHUNK 128
#endif
constexpr uint32_t kFetchChunk = ATN_FETCH_CHUNK;
constexpr int kTraceWavesPerBlock = kTraceBlock / 64;
// (Merging the launch tail's half-empty waves -- within a block through LDS, or through a continuation launch -- was
// built and measured in r03: correct, 12 % / 20 % slower.  DESIGN.md section 7, profiles/r03_variants_tail_merge.txt.)
struct TraceShared {
    float4 stage[kTraceWavesPerBlock][kFetchChunk][2]; float stop[kTraceWavesPerBlock][kFetchChunk];    // 18 KB
};

// (XCD-affine job lists -- rays filed by the cell of their origin, a block draining its own XCD's list first -- were built and
// measured in r04: L2 hit 0.79 -> 0.85 and fabric reads -41 % on the 250 K-triangle atrium, and the launch no faster.
// profiles/r04_variants_ray_cells.txt; code in git history, commit "Ray cells".)
template <bool COUNT, class Job, bool LDSN = false>
ATN_DEV void trace_refill(const DevScene& sc, TraceShared& sh, uint32_t count, uint32_t* fetch_counter,
                          const Job& job, TravCounters* cnt)
{
    const char* __restrict__ nb = reinterpret_cast<const char*>(sc.nodes);
    const float t_min = sc.bvh_hit_min > 0 ? sc.bvh_hit_min : job.t_min;
    const uint32_t lane = __lane_id();
    float4 (*stage)[2] = sh.stage[threadIdx.x >> 6];
    float* stage_stop = sh.stop[threadIdx.x >> 6];

    uint32_t c_count = 0, c_next = 0;   // wave-uniform: staged chunk size / next unassigned entry
    bool drained = false;               // wave-uniform: the global queue is empty
    const uint32_t wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    bool first_chunk = true;            // wave-uniform
    bool all_finite = true;             // wave-uniform: every live lane's current slab constants are finite
    uint32_t iter = 0;                  // wave-uniform
    Walk w;
    slab_setup(w.wray, mk3(0.0F), mk3(0.0F, 0.0F, 1.0F));
    w.ray = w.wray;
    w.node = kLinkEnd;
    w.t_max = 0.0F; w.stop_t = -kInf; w.payload = 0;
    w.hit.t = kInf; w.hit.objid = -1; w.hit.tri = -1; w.hit.a = 0.0F; w.hit.b = 0.0F; w.hit.meshid = -1;
    w.objid = -1; w.meshid = -1; w.top_hit = kLinkEnd; w.top_miss = kLinkEnd;
    w.pending = false;

    for (;;) {
        // ---- refill
        unsigned long long m_idle = __ballot(w.node == kLinkEnd);
        uint32_t n_idle = (uint32_t)__popcll(m_idle);
        if (n_idle >= kRefillLanes) {
            // the rays that ended since the last refill are finished here, together (walk_finish)
            if (walk_finish<COUNT, Job, LDSN>(w, sc, job, cnt)) {
                m_idle = __ballot(w.node == kLinkEnd);
                n_idle = (uint32_t)__popcll(m_idle);
                all_finite = __all(w.node == kLinkEnd || w.ray.finite) != 0;
            }
            if (c_next >= c_count && !drained) {
                // The first chunk of every wave is pre-assigned (chunk index = global wave id) and the shared
                // cursor starts after those: a same-address atomic retires only every ~11 ns, so a launch that
                // opens with one atomic per wave (5 K waves) would stall for tens of microseconds.
                uint32_t base = 0;
                if (first_chunk) {
                    base = wave_id * kFetchChunk;
                    first_chunk = false;
                }
                else {
                    if (lane == 0) base = (atomicAdd(fetch_counter, 1u) + n_waves) * kFetchChunk;
                }
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (base >= count) { drained = true; c_count = 0; c_next = 0; }
                else {
                    c_count = count - base < kFetchChunk ? count - base : kFetchChunk;
                    c_next = 0;
#pragma unroll
                    for (uint32_t e = 0; e < kFetchChunk; e += 64) {
                        if (e + lane < c_count) {
                            float4 a, b;
                            float st;
                            job.fetch(base + e + lane, a, b, st);
                            stage[e + lane][0] = a;
                            stage[e + lane][1] = b;
                            stage_stop[e + lane] = st;
                        }
                    }
                }
            }
            if (c_next < c_count) {
                const uint32_t avail = c_count - c_next;
                if (w.node == kLinkEnd) {
                    const uint32_t k = bits_below_lane(m_idle);
                    if (k < avail) walk_start<LDSN, COUNT>(w, sc, stage[c_next + k][0], stage[c_next + k][1], stage_stop[c_next + k], cnt);
                }
                c_next += n_idle < avail ? n_idle : avail;
                all_finite = __all(w.node == kLinkEnd || w.ray.finite) != 0;
            }
            else if (n_idle == 64u) {
                break;          // drained, chunk empty, nothing in flight
            }
        }
        walk_iteration<COUNT, kInnerBurst, Job, LDSN, ATN_DEFER_FINISH != 0>(w, all_finite, sc, nb, t_min, job, cnt, iter++);
    }
}

} // namespace atn
