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

struct TravCounters { uint32_t nodes, tris; };

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

constexpr int kTraceBlock = 256;

ATN_DEV float4 ld16(const char* base, uint32_t byte_off)
{
    return *reinterpret_cast<const float4*>(base + byte_off);
}

// Job interface (all jobs of a launch share t_min):
//   float t_min
//   void fetch(uint32_t j, float4& a, float4& b, float& stop_t)   a = {org.xyz, t_max}, b = {dir.xyz, payload bits}
//        stop_t: the walk may stop at the first ACCEPTED hit whose t <= stop_t.  -inf = plain closest-hit walk;
//        +inf = "any hit" (only finish()'s is_hit is used); a finite value = the caller only needs to know whether
//        the closest hit is nearer than stop_t (shadow rays toward point / spot lights).  This is exact, not an
//        approximation: up to that hit the closest-hit walk is the same walk, its final hit can only be nearer,
//        and when no such hit is accepted the walk runs to its end and reports the exact closest hit.
//   void finish(uint32_t payload, const Hit& h, bool is_hit)
// One ray per lane for the lifetime of its walk; grid-stride over the jobs.
template <bool COUNT, class Job>
ATN_DEV void trace_simple(const DevScene& sc, uint32_t count, const Job& job, TravCounters* cnt)
{
    const char* __restrict__ nb = reinterpret_cast<const char*>(sc.nodes);
    const float t_min = sc.bvh_hit_min > 0 ? sc.bvh_hit_min : job.t_min;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < count; j += stride) {
        float4 a, b;
        float stop_t;
        job.fetch(j, a, b, stop_t);
        float t_max = a.w;
        const uint32_t payload = __float_as_uint(b.w);
        RaySlab wray, ray;
        slab_setup(wray, mk3(a), mk3(b));
        ray = wray;
        Hit hit; hit.t = kInf; hit.objid = -1; hit.tri = -1; hit.a = 0.0F; hit.b = 0.0F; hit.meshid = -1;
        int32_t node = sc.root_link, objid = -1, meshid = -1, top_hit = kLinkEnd, top_miss = kLinkEnd;

        while (node != kLinkEnd) {
            const uint32_t off = (uint32_t)node & kLinkOffsetMask;
            const float4 q0 = ld16(nb, off);
            const float4 q1 = ld16(nb, off + 16u);
            if (COUNT) cnt->nodes++;
            bool is_hit;
            if (!(node & kLinkTypeMask)) {
                // inner node (or a leaf with nothing to test: its tag makes the slab result irrelevant)
                is_hit = ray.finite ? slab_hit_fast(ray, mk3(q0), mk3(q1), t_min, t_max)
                                    : slab_hit_exact(ray, mk3(q0), mk3(q1), t_min, t_max);
                const int32_t tag = __float_as_int(q0.w);
                const int32_t hit_link = (int32_t)(off + kNodeBytes) | tag;       // tag = type bits of the next node
                node = (is_hit && tag != kTagDead) ? hit_link : __float_as_int(q1.w);
                is_hit = is_hit && tag != kTagDead;
            }
            else if (node & kLinkLeafBit) {
                const float4 q2 = ld16(nb, off + 32u);
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
                    hit.objid = objid; hit.tri = __float_as_int(q0.w); hit.meshid = meshid;
                    t_max = t;
                }
                node = __float_as_int(q1.w);        // leaf: hit link == miss link
                if (accept && t <= stop_t) { node = kLinkEnd; top_hit = kLinkEnd; top_miss = kLinkEnd; }    // see Job::fetch
            }
            else {
                // TLAS leaf with a nested tree
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
                    const f3 o = m4_apply(m, wray.org);
                    const f3 d = normalize(m4_applyXYZ(m, wray.dir));
                    slab_setup(ray, o, d);
                }
                else {
                    ray = wray;
                }
                is_hit = true;
                node = __float_as_int(q0.z);        // BLAS root link
            }
            if (node == kLinkEnd) {
                // leave the bottom layer (top_* are kLinkEnd inside the top layer)
                node = is_hit ? top_hit : top_miss;
                top_hit = kLinkEnd; top_miss = kLinkEnd;
                ray = wray;
            }
        }
        job.finish(payload, hit, hit.objid >= 0);
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
#ifndef ATN_LEAF_CADENCE
#define ATN_LEAF_CADENCE 3
#endif
constexpr uint32_t kLeafCadence = ATN_LEAF_CADENCE;
struct TraceShared { float4 stage[kTraceWavesPerBlock][kFetchChunk][2]; float stop[kTraceWavesPerBlock][kFetchChunk]; };   // 18 KB

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
    const bool is_hit = isect && (t < kInf);
    accept = (t_min < (is_hit ? t : kInf)) && is_hit && (t < hit.t);
    if (accept) {
        hit.t = t; hit.a = beta; hit.b = gamma;
        hit.objid = objid; hit.tri = __float_as_int(q0.w); hit.meshid = meshid;
        t_max = t;
    }
    t_out = t;
    return is_hit;
}

// The persistent walk.  One wave iteration = refill, then a BURST of kInnerBurst inner-node steps in a tight loop (two
// 16-byte loads, the slab test, the link select -- nothing else), then ONE step for the lanes that stand on a
// triangle leaf or a TLAS leaf, then the lanes whose list ended leave the bottom layer or finish.
// Why: at any moment only ~7 of 64 lanes stand on a leaf (one visit in nine), so a loop that offers every node kind
// on every iteration issues the ~75-instruction triangle block each time for a handful of lanes, and drags the
// refill / leave / finish bookkeeping (~60 scalar instructions) through every inner-node step.  Here a lane that
// reaches a leaf waits, masked off, for the end of the burst (<= kInnerBurst - 1 steps), the triangle block runs once
// per burst with several times the lanes, and the inner-node step is ~30 VALU + ~10 SALU.
// The slab test takes ONE of its two forms per wave: the hardware min/max form when every live lane's slab constants
// are finite (a wave-uniform flag, refreshed only where rays change), the select form -- valid for all inputs --
// otherwise.  A ray's own sequence of operations is the reference walk's, so results stay bit-identical.
#ifndef ATN_INNER_BURST
#define ATN_INNER_BURST 4
#endif
constexpr int kInnerBurst = ATN_INNER_BURST;

template <bool COUNT, class Job>
ATN_DEV void trace_refill(const DevScene& sc, TraceShared& sh, uint32_t count, uint32_t* fetch_counter,
                          const Job& job, TravCounters* cnt)
{
    const char* __restrict__ nb = reinterpret_cast<const char*>(sc.nodes);
    const float t_min = sc.bvh_hit_min > 0 ? sc.bvh_hit_min : job.t_min;
    const uint32_t lane = __lane_id();
    const unsigned long long lt = (1ull << lane) - 1ull;
    float4 (*stage)[2] = sh.stage[threadIdx.x >> 6];
    float* stage_stop = sh.stop[threadIdx.x >> 6];

    uint32_t c_count = 0, c_next = 0;   // wave-uniform: staged chunk size / next unassigned entry
    bool drained = false;               // wave-uniform: the global queue is empty
    const uint32_t wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    bool first_chunk = true;            // wave-uniform
    bool all_finite = true;             // wave-uniform: every live lane's current slab constants are finite

    uint32_t payload = 0;
    float t_max = 0.0F, stop_t = -kInf;
    RaySlab wray, ray;
    slab_setup(wray, mk3(0.0F), mk3(0.0F, 0.0F, 1.0F));
    ray = wray;
    Hit hit; hit.t = kInf; hit.objid = -1; hit.tri = -1; hit.a = 0.0F; hit.b = 0.0F; hit.meshid = -1;
    int32_t node = kLinkEnd, objid = -1, meshid = -1, top_hit = kLinkEnd, top_miss = kLinkEnd;
    // a lane is idle <=> node == kLinkEnd at the top of an iteration

    for (;;) {
        // ---- refill
        const unsigned long long m_idle = __ballot(node == kLinkEnd);
        const uint32_t n_idle = (uint32_t)__popcll(m_idle);
        if (n_idle >= kRefillLanes) {
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
                if (node == kLinkEnd) {
                    const uint32_t k = (uint32_t)__popcll(m_idle & lt);
                    if (k < avail) {
                        const float4 a = stage[c_next + k][0];
                        const float4 b = stage[c_next + k][1];
                        t_max = a.w;
                        stop_t = stage_stop[c_next + k];
                        payload = __float_as_uint(b.w);
                        hit.t = kInf; hit.objid = -1; hit.tri = -1; hit.a = 0.0F; hit.b = 0.0F; hit.meshid = -1;
                        slab_setup(wray, mk3(a), mk3(b));
                        ray = wray;
                        node = sc.root_link; objid = -1; meshid = -1; top_hit = kLinkEnd; top_miss = kLinkEnd;
                    }
                }
                c_next += n_idle < avail ? n_idle : avail;
                all_finite = __all(node == kLinkEnd || ray.finite) != 0;
            }
            else if (n_idle == 64u) {
                break;          // drained, chunk empty, nothing in flight
            }
        }

        // ---- burst of inner-node steps (dead leaves are typed inner, tag kTagDead).  kLinkEnd has both type bits set,
        // so `(node & 3) == 0` alone selects the live lanes on inner nodes.  An inner node's hit link is the next
        // record and never kLinkEnd (checked at upload): a list that ends here ended on a MISS link.
        const bool live = node != kLinkEnd;
#pragma unroll 1
        for (int k = 0; k < kInnerBurst; k++) {
            if (!(node & kLinkTypeMask)) {
                const uint32_t off = (uint32_t)node;        // type bits are 0: the link is the byte offset
                const float4 q0 = ld16(nb, off);
                const float4 q1 = ld16(nb, off + 16u);
                if (COUNT) cnt->nodes++;
                bool box;
                if (all_finite) box = slab_hit_fast(ray, mk3(q0), mk3(q1), t_min, t_max);
                else box = slab_hit_exact(ray, mk3(q0), mk3(q1), t_min, t_max);
                const int32_t tag = __float_as_int(q0.w);
                node = (box && tag != kTagDead) ? ((int32_t)(off + kNodeBytes) | tag) : __float_as_int(q1.w);  // hit link = next record, typed by the tag
            }
        }
        bool ended = live && node == kLinkEnd;      // this lane's walk left a list in this iteration ...
        bool is_hit = false;                        // ... and this was the result of its last step

        // ---- one step for the lanes on a triangle leaf or a TLAS leaf (both read the record's first two quarters)
        const bool at_tlas = node != kLinkEnd && (node & kLinkTypeMask) == kLinkTlasBit;
        if (node != kLinkEnd && (node & kLinkTypeMask)) {
            const uint32_t off = (uint32_t)node & kLinkOffsetMask;
            const float4 q0 = ld16(nb, off);
            const float4 q1 = ld16(nb, off + 16u);
            if (COUNT) cnt->nodes++;
            if (node & kLinkLeafBit) {
                const float4 q2 = ld16(nb, off + 32u);
                if (COUNT) cnt->tris++;
                bool accept; float t;
                is_hit = leaf_test(ray, q0, q1, q2, t_min, hit, t_max, objid, meshid, accept, t);
                node = __float_as_int(q1.w);        // leaf: hit link == miss link
                if (accept && t <= stop_t) { node = kLinkEnd; top_hit = kLinkEnd; top_miss = kLinkEnd; }    // see Job::fetch
                ended = node == kLinkEnd;
            }
            else {
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
                    const f3 o = m4_apply(m, wray.org);
                    const f3 d = normalize(m4_applyXYZ(m, wray.dir));
                    slab_setup(ray, o, d);
                }
                else {
                    ray = wray;
                }
                is_hit = true;
                node = __float_as_int(q0.z);        // BLAS root link (never kLinkEnd: empty lists are rejected at upload)
                ended = false;
            }
        }

        // ---- a list ended: leave the bottom layer (top_* are kLinkEnd inside the top layer), or finish
        if (ended) {
            node = is_hit ? top_hit : top_miss;
            top_hit = kLinkEnd; top_miss = kLinkEnd;
            ray = wray;
            if (node == kLinkEnd) job.finish(payload, hit, hit.objid >= 0);
        }
        // rays changed in the two blocks above: refresh the wave's slab-form flag (cheap, and only then)
        if (__any(ended || at_tlas) || !all_finite) all_finite = __all(node == kLinkEnd || ray.finite) != 0;
    }
}

} // namespace atn
