// Host-side conversion of the caller's flat scene (include/aten_layout.h) into the device layout
// of device/scene_dev.hpp.  Pure host C++ (no HIP calls) so that it can be unit-tested on CPU.
#pragma once
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../device/scene_dev.hpp"
#include "anyhit_twin.hpp"

namespace atn {

struct HostSceneImage {
    std::vector<float4> nodes;          // the byte image of all records (16-byte units)
    uint64_t n_nodes = 0;
    std::vector<uint32_t> list_root;    // byte offset of each list's first record
    std::vector<int32_t> list_root_link; // typed link of each list's root
    std::vector<uint32_t> list_bytes;   // bytes of each list's contiguous region starting at list_root
    std::vector<int32_t> list_twin_delta; // byte distance from a bottom-level list's root record to the root of its any-hit twin (anyhit_twin.hpp), 0 = none
    struct TlasRef { uint32_t offset, list; };
    std::vector<TlasRef> tlas_refs;     // every TLAS-leaf record (byte offset) and the list it enters: where a twin is switched off later (LBVH rebuild)
    std::vector<uint32_t> list_tri_leaves, list_inner;  // record counts of each list
    std::vector<atn_triangle_param> tris;
    std::vector<float4> vtx_pos, vtx_nml;
    std::vector<atn_object_param> objects;
    std::vector<float4> matrices;
    std::vector<DevMaterial> materials;
    std::vector<float4> carpaint;
    std::vector<atn_toon_param> toon;           // per material + the fallback slot
    std::vector<atn_light_param> npr_lights;
    std::vector<float> screen_shadow;
    std::vector<atn_light_param> lights;
    std::vector<float4> light_plane;    // per light: {unit normal of a planar area light's plane in world space, 1}, or zeros (planar_area_light)
    std::vector<float4> texels;
    std::vector<uint32_t> texels8;
    std::vector<DevTexture> textures;
    DevScene params{};                  // scalar fields filled; pointers left null
    uint64_t n_inner = 0, n_tri_leaf = 0, n_tlas_leaf = 0;
};

inline float i2f(int32_t i) { float f; std::memcpy(&f, &i, 4); return f; }
inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

// Walk order of a threaded list = following hit links from node 0 (every node's hit link is the
// next node in the depth-first order the traverser visits when nothing is culled).
inline bool walk_order(const atn_bvh_node* nodes, uint32_t count, std::vector<int32_t>& new_index, std::vector<uint32_t>& order, std::string& err)
{
    new_index.assign(count, -1);
    order.clear();
    order.reserve(count);
    int32_t id = count ? 0 : -1;
    while (id >= 0) {
        if ((uint32_t)id >= count) { err = "BVH hit link out of range"; return false; }
        if (new_index[id] >= 0) { err = "BVH hit links form a cycle"; return false; }
        new_index[id] = (int32_t)order.size();
        order.push_back((uint32_t)id);
        id = (int32_t)nodes[id].hit;
    }
    return true;
}

enum NodeKind : uint8_t { KIND_INNER = 0, KIND_TRI = 1, KIND_TLAS = 2, KIND_DEAD = 3 };
inline uint32_t record_bytes(uint8_t kind) { return kind == KIND_TRI ? kTriLeafBytes : kInnerBytes; }
inline int32_t kind_type_bits(uint8_t kind) { return kind == KIND_TRI ? kLinkToLeaf : (kind == KIND_TLAS ? kLinkToTlas : 0); }

// One threaded list analysed: walk order, node kinds, depth of every node, and (filled by the layout pass) the byte
// offset of every node's device record.
struct ListLayout {
    std::vector<uint32_t> order;        // walk position -> caller's node index
    std::vector<int32_t> new_index;     // caller's node index -> walk position (-1 = unreachable)
    std::vector<uint8_t> kind;          // by walk position
    std::vector<int32_t> depth;         // by walk position
    std::vector<uint32_t> offset;       // by walk position: byte offset of the record in the device image
};

// Validates a list (reachability, forward links, the structural rules the walk relies on) and fills order/kind/depth.
inline bool analyse_list(ListLayout& L, const atn_bvh_node* src, uint32_t count, bool top, std::string& err)
{
    if (!walk_order(src, count, L.new_index, L.order, err)) return false;
    const uint32_t n = (uint32_t)L.order.size();
    L.kind.assign(n, KIND_INNER); L.depth.assign(n, 0); L.offset.assign(n, 0);
    std::vector<uint32_t> open_end;     // walk positions at which the open subtrees end (a stack)
    for (uint32_t j = 0; j < n; j++) {
        const atn_bvh_node& nd = src[L.order[j]];
        // every link in range, reachable, and pointing FORWARD in walk order: the device walk has no other termination
        // argument (a corrupted or hand-edited .sbvh with a backward miss link would spin a wave forever)
        for (const float link : { nd.hit, nd.miss }) {
            const int32_t l = (int32_t)link;
            if (l < 0) continue;
            if ((uint32_t)l >= count || L.new_index[l] < 0) { err = "BVH link points to an unreachable node"; return false; }
            if (L.new_index[l] <= (int32_t)j) { err = "BVH link points backward in walk order (the walk would not terminate)"; return false; }
        }
        const bool leaf = (nd.f0 >= 0 || nd.f1 >= 0);       // ThreadedBvhNode::isLeaf, threaded_bvh.h:41-44
        uint8_t kind;
        if (!leaf) {
            kind = KIND_INNER;
            // the walk treats "a list ended on an inner node" as "ended on its miss link" (traverse.hpp)
            if ((int32_t)nd.hit < 0) { err = "inner node without a hit link"; return false; }
        }
        else if (nd.f2 >= 0) {
            if (!top) { err = "nested BVH reference inside a bottom-level list"; return false; }
            kind = KIND_TLAS;
        }
        else if (nd.f1 >= 0) {
            if ((int32_t)nd.hit != (int32_t)nd.miss) { err = "triangle leaf with hit != miss link"; return false; }
            kind = KIND_TRI;
        }
        else kind = KIND_DEAD;      // leaf without triangle or nested tree (sphere instance): never tested on this path
        L.kind[j] = kind;
        while (!open_end.empty() && open_end.back() <= j) open_end.pop_back();
        L.depth[j] = (int32_t)open_end.size();
        if (kind == KIND_INNER) {
            const int32_t m = (int32_t)nd.miss;
            open_end.push_back(m < 0 ? 0xffffffffu : (uint32_t)L.new_index[m]);       // the subtree is skipped by the miss link
        }
    }
    return true;
}

// Where a list's records go, from `off` on.  Links are explicit, so the layout is free (the root stays first: a list's typed root
// link and an LBVH rebuild's region start there).  top_levels = 0: walk (pre-)order, what r01-r04 used "for locality".  Default
// (kLayoutTopLevels): the nodes of the first 16 levels level by level -- the records most walks visit, packed into one stretch with
// the two children of a node next to each other (a miss goes to the sibling: same 64-byte chunk for two inner records) -- and everything
// below in walk order.  Measured (r05, profiles/r05_variants_node_layout.txt): sponza_lod 3.25 -> 3.18 ms per frame, the atrium (20 MB
// of records, L2-bound) 5.02 -> 4.85; 8 / 12 levels gain less, 20 / 24 / all levels the same; sibling pairs below the top levels
// instead of walk order: the same on sponza_lod, slightly worse on the atrium; padding so that no record straddles a 64-byte chunk:
// -0.5 % on the atrium, not kept (the twins of a list must all be of the list's size).
constexpr int kLayoutTopLevels = 16;
inline void assign_offsets(ListLayout& L, uint64_t& off, int top_levels)
{
    const uint32_t n = (uint32_t)L.order.size();
    auto place = [&](uint32_t j) { L.offset[j] = (uint32_t)std::min<uint64_t>(off, 0xfffffff0u); off += record_bytes(L.kind[j]); };
    if (top_levels > 0) {
        // (bucket the walk positions by depth: one pass, not one per level)
        std::vector<std::vector<uint32_t>> level((size_t)top_levels);
        for (uint32_t j = 0; j < n; j++) if (L.depth[j] < top_levels) level[(size_t)L.depth[j]].push_back(j);
        for (const auto& lv : level) for (const uint32_t j : lv) place(j);
    }
    for (uint32_t j = 0; j < n; j++) if (top_levels <= 0 || L.depth[j] >= top_levels) place(j);
}

// What a list's records need from the rest of the scene.
struct ListEmitCtx {
    const atn_object_param* objects = nullptr; uint32_t n_objects = 0; uint32_t n_matrices = 0;
    const atn_triangle_param* tris = nullptr; const atn_vec4* vtx_pos = nullptr; uint32_t n_triangles = 0, n_vertices = 0;
    const int32_t* list_root_link = nullptr; uint32_t n_lists = 0;     // typed link of list k's root; kLinkEnd = empty list
    const int32_t* list_twin_delta = nullptr;       // HostSceneImage::list_twin_delta (null: no twins)
    std::vector<HostSceneImage::TlasRef>* tlas_refs = nullptr;     // out (optional)
    const atn_mat4* matrices = nullptr;     // the matrices the TLAS leaves' rows index (null: identity instances are not recognised)
    mutable int32_t ident_row = -1;         // out: w2l_row of an instance whose W2L is bit for bit the identity
};

// mat4 == identity, bit for bit (+0.0 zeros): applying it is the same arithmetic for every instance that has it
inline bool is_exact_identity(const atn_mat4& m)
{
    static const float I[4][4] = { { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 }, { 0, 0, 0, 1 } };
    return std::memcmp(m.m, I, sizeof(I)) == 0;
}

// DevScene::root_*: filled when the top layer's root is a TLAS leaf whose two top links end the walk (a one-node top layer).
// `image` = host copy of the node image from byte `bias` on.
inline void fill_root_direct(DevScene& p, const float4* image, uint32_t bias, const atn_mat4* matrices = nullptr, uint32_t n_matrices = 0)
{
    for (float& v : p.root_m) v = 0.0F;
    p.root_direct = 0; p.root_objid = -1; p.root_meshid = -1; p.root_w2l = -1; p.root_blas = kLinkEnd; p.root_flags = 0; p.root_twin = 0;
    if (p.root_link == kLinkEnd || p.root_link >= 0 || (p.root_link & kLinkTypeMask) != kLinkTlasBit) return;
    const uint32_t off = (uint32_t)p.root_link & kLinkOffsetMask;
    if (off < bias) return;
    const float4* q = image + (off - bias) / 16;
    auto f2i = [](float f) { int32_t i; std::memcpy(&i, &f, 4); return i; };
    if (f2i(q[1].y) != kLinkEnd || f2i(q[1].z) != kLinkEnd) return;
    p.root_direct = 1;
    p.root_objid = f2i(q[0].x); p.root_w2l = f2i(q[0].y); p.root_blas = f2i(q[0].z); p.root_flags = f2i(q[0].w);
    p.root_meshid = f2i(q[1].x);
    p.root_twin = f2i(q[1].w);
    if (p.root_w2l >= 0) {
        const uint32_t mi = (uint32_t)p.root_w2l / 4u;
        if (!matrices || mi >= n_matrices) { p.root_direct = 0; return; }      // (no host copy of the matrices: walks start at root_link)
        for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) p.root_m[4 * r + c] = matrices[mi].m[r][c];
    }
}

// Writes the device records of one analysed list (offsets already assigned) into the byte image `img`.
// (`img` + offset - write_bias is where a record goes: write_bias > 0 when `img` holds only the image's tail.)
inline bool emit_list(char* img, const ListLayout& L, const atn_bvh_node* src, const ListEmitCtx& c,
                      int32_t& root_link, uint64_t counts[3], std::string& err, uint32_t write_bias = 0)
{
    const uint32_t n = (uint32_t)L.order.size();
    auto typed = [&](float link) -> int32_t {
        const int32_t l = (int32_t)link;
        if (l < 0) return kLinkEnd;
        const uint32_t j = (uint32_t)L.new_index[l];
        return (int32_t)L.offset[j] | kind_type_bits(L.kind[j]);
    };
    root_link = n ? ((int32_t)L.offset[0] | kind_type_bits(L.kind[0])) : kLinkEnd;
    for (uint32_t j = 0; j < n; j++) {
        const atn_bvh_node& nd = src[L.order[j]];
        float4* q = reinterpret_cast<float4*>(img + (L.offset[j] - write_bias));
        const int32_t h = typed(nd.hit), m = typed(nd.miss);
        switch (L.kind[j]) {
        case KIND_INNER:
            q[0] = make_float4(nd.boxmin[0], nd.boxmin[1], nd.boxmin[2], i2f(h));
            q[1] = make_float4(nd.boxmax[0], nd.boxmax[1], nd.boxmax[2], i2f(m));
            counts[0]++;
            break;
        case KIND_DEAD:     // an inner record whose both links are the miss link: whatever the slab test says, the walk goes on
            q[0] = make_float4(0, 0, 0, i2f(m));
            q[1] = make_float4(0, 0, 0, i2f(m));
            break;
        case KIND_TLAS: {
            // nested tree (exid bit-field, threaded_bvh.h:29-37)
            const int32_t objid = (int32_t)nd.f0;
            if (objid < 0 || (uint32_t)objid >= c.n_objects) { err = "TLAS leaf object id out of range"; return false; }
            const uint32_t bits = f2u(nd.f2);
            const int32_t exid = ATN_EXID_MAIN(bits);
            if (exid <= 0 || (uint32_t)exid >= c.n_lists || c.list_root_link[exid] == kLinkEnd) { err = "TLAS leaf references a missing BLAS list"; return false; }
            const atn_object_param& obj = c.objects[objid];
            int32_t w2l_row = -1;
            if (obj.mtx_id >= 0) {
                if ((uint32_t)obj.mtx_id + 1 >= c.n_matrices) { err = "object matrix index out of range"; return false; }
                w2l_row = 4 * (obj.mtx_id + 1);              // traverser reads GetMatrix(mtx_id + 1), :153
            }
            int32_t flags = 0;
            if (w2l_row >= 0 && c.matrices && is_exact_identity(c.matrices[obj.mtx_id + 1])) { flags |= kTlasIdentity; if (c.ident_row < 0) c.ident_row = w2l_row; }
            q[0] = make_float4(i2f(objid), i2f(w2l_row), i2f(c.list_root_link[exid]), i2f(flags));
            q[1] = make_float4(i2f((int32_t)nd.f3), i2f(h), i2f(m), i2f(c.list_twin_delta ? c.list_twin_delta[exid] : 0));
            if (c.tlas_refs) c.tlas_refs->push_back({ L.offset[j], (uint32_t)exid });
            counts[2]++;
            break;
        }
        default: {  // KIND_TRI
            if (!c.tris) { err = "triangle leaves in this list need a full scene upload"; return false; }
            const uint32_t tri = (uint32_t)nd.f1;
            if (tri >= c.n_triangles) { err = "leaf triangle id out of range"; return false; }
            const atn_triangle_param& t = c.tris[tri];
            for (int v = 0; v < 3; v++)
                if (t.idx[v] < 0 || (uint32_t)t.idx[v] >= c.n_vertices) { err = "triangle vertex index out of range"; return false; }
            const atn_vec4& a = c.vtx_pos[t.idx[0]];
            const atn_vec4& b = c.vtx_pos[t.idx[1]];
            const atn_vec4& cc = c.vtx_pos[t.idx[2]];
            // e1 = v1 - v0, e2 = v2 - v0: the same fp32 subtractions intersectTriangle performs
            // per test (math/intersect.h:61-62), hoisted to upload time.
            q[0] = make_float4(a.x, a.y, a.z, i2f((int32_t)tri));
            q[1] = make_float4(b.x - a.x, b.y - a.y, b.z - a.z, i2f(h));
            q[2] = make_float4(cc.x - a.x, cc.y - a.y, cc.z - a.z, 0.0F);
            counts[1]++;
            break;
        }
        }
    }
    return true;
}

// Range checks of every id the kernels index with (a bad id is a device out-of-bounds read, not an error code).
// `s` may be null for a top-layer update (objects / matrices only).
inline bool validate_ranges(const atn_object_param* objs, uint32_t n_objs, uint32_t n_mtx, const atn_scene_desc* s, std::string& err)
{
    const uint32_t n_tris = s ? s->n_triangles : 0xffffffffu, n_lights = s ? s->n_lights : 0xffffffffu;
    for (uint32_t i = 0; i < n_objs; i++) {
        const atn_object_param& o = objs[i];
        if (o.type == ATN_OBJ_INSTANCE && (o.object_id < 0 || (uint32_t)o.object_id >= n_objs)) { err = "instance refers to an object id out of range"; return false; }
        if (o.mtx_id >= 0 && (uint32_t)o.mtx_id + 1 >= n_mtx) { err = "object matrix index out of range"; return false; }
        if (s && o.light_id >= 0 && (uint32_t)o.light_id >= n_lights) { err = "object light id out of range"; return false; }
        if (s && o.type == ATN_OBJ_POLYGONS && o.triangle_num > 0
            && (o.triangle_id < 0 || (uint64_t)o.triangle_id + (uint64_t)o.triangle_num > n_tris)) { err = "object triangle range out of range"; return false; }
    }
    if (!s) return true;
    for (uint32_t i = 0; i < s->n_triangles; i++) {
        const atn_triangle_param& t = s->triangles[i];
        if (t.mtrlid >= 0 && (uint32_t)t.mtrlid >= s->n_materials) { err = "triangle material id out of range"; return false; }
        for (int v = 0; v < 3; v++)
            if (t.idx[v] < 0 || (uint32_t)t.idx[v] >= s->n_vertices) { err = "triangle vertex index out of range"; return false; }
    }
    for (uint32_t i = 0; i < s->n_lights; i++) {
        const atn_light_param& l = s->lights[i];
        if (l.arealight_objid >= 0 && (uint32_t)l.arealight_objid >= s->n_objects) { err = "light refers to an object id out of range"; return false; }
    }
    return true;
}

// Node image (bytes): [BLAS list 1 | BLAS list 2 | ... | top layer (list 0)].
//  * every bottom-level list followed by its any-hit twins; inside a list the records lie as assign_offsets puts them (the links are
//    explicit, so correctness does not depend on it)
//  * the top layer comes last so that update_top_layer (≙ Renderer::updateBVH, "only for top layer") can replace it
//    without moving the others; top-layer records are all kInnerBytes long.
// The shadow ray towards an AREA light needs the closest hit's OBJECT (scene::hitLight, scene/scene.h:118-131: visible iff it is the
// light's), so its walk is a closest-hit walk in the list as given.  But when the light's object is PLANAR -- every vertex in the plane
// of its first triangle -- and placed by a rigid matrix (hit distances inside the instance are world distances), the ray, which is aimed
// at a point of that plane at distToLight, meets the light's object there and nowhere else: an accepted hit with t <= 0.999 distToLight
// is on some OTHER object, and since the walk's closest hit can only be nearer still, the closest hit's object is not the light's
// either -- the answer is "blocked" and the walk may stop (ShadowJob::fetch: a finite stop_t, the rule punctual lights already use).
// Same list, same order, same decisions up to that hit: the answer is the reference's.
// "At distToLight" needs care: the ray starts at ray::Offset(p, n) (math/ray.h:26-74: up to 256 ulps or 2^-16 per coordinate away from
// p) but points along pos - p, so it crosses the light's plane at distToLight - ((o' - p) . n_l) / (dir . n_l): for a grazing ray that
// is anywhere.  fetch therefore applies the rule only when |dir . n_l| >= 0.01 and the largest possible |o' - p| is at most
// 5e-4 * distToLight * |dir . n_l| -- then the crossing, and with it every hit on the light's object (Moeller-Trumbore's own rounding
// at such an angle: << 1e-4), lies beyond 0.999 * distToLight.  The plane's world-space normal per light: HostSceneImage::light_plane.  Measured: atrium shadow-ray node visits
// 296.7 M -> 273.8 M per frame, 4.88 -> 4.77 ms; the Cornell box (most of its shadow rays reach the lamp) unchanged.
inline bool planar_area_light(const atn_scene_desc* s, const atn_light_param& l, float world_normal[3])
{
    const atn_mat4* l2w = nullptr;
    if (l.type != ATN_LIGHT_AREA || l.arealight_objid < 0 || (uint32_t)l.arealight_objid >= s->n_objects) return false;
    const atn_object_param* o = &s->objects[l.arealight_objid];
    if (o->type == ATN_OBJ_INSTANCE) {
        if (o->mtx_id >= 0) {
            if ((uint32_t)o->mtx_id + 1 >= s->n_matrices) return false;
            const atn_mat4& m = s->matrices[o->mtx_id + 1];                 // W2L: rows 0..2 orthonormal <=> distances are preserved
            for (int a = 0; a < 3; a++)
                for (int b = a; b < 3; b++) {
                    double d = 0;
                    for (int c = 0; c < 3; c++) d += (double)m.m[a][c] * (double)m.m[b][c];
                    if (std::fabs(d - (a == b ? 1.0 : 0.0)) > 1e-6) return false;
                }
            l2w = &s->matrices[o->mtx_id];                                   // (L2W sits before W2L: instance.h:253-269)
        }
        if (o->object_id < 0 || (uint32_t)o->object_id >= s->n_objects) return false;
        o = &s->objects[o->object_id];
    }
    if (o->type != ATN_OBJ_POLYGONS || o->triangle_num <= 0 || o->triangle_id < 0
        || (uint64_t)o->triangle_id + (uint64_t)o->triangle_num > s->n_triangles) return false;
    double p0[3] = { 0, 0, 0 }, n[3] = { 0, 0, 0 }, lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
    bool have_plane = false;
    auto vtx = [&](int32_t i, double v[3]) { v[0] = s->vtx_pos[i].x; v[1] = s->vtx_pos[i].y; v[2] = s->vtx_pos[i].z; };
    for (int pass = 0; pass < 2; pass++) {
        for (int32_t t = o->triangle_id; t < o->triangle_id + o->triangle_num; t++) {
            const atn_triangle_param& tr = s->triangles[t];
            double v[3][3];
            for (int k = 0; k < 3; k++) {
                if (tr.idx[k] < 0 || (uint32_t)tr.idx[k] >= s->n_vertices) return false;
                vtx(tr.idx[k], v[k]);
            }
            if (pass == 0) {
                for (int k = 0; k < 3; k++) for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], v[k][a]); hi[a] = std::max(hi[a], v[k][a]); }
                if (!have_plane) {
                    const double e1[3] = { v[1][0] - v[0][0], v[1][1] - v[0][1], v[1][2] - v[0][2] }, e2[3] = { v[2][0] - v[0][0], v[2][1] - v[0][1], v[2][2] - v[0][2] };
                    const double c[3] = { e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0] };
                    const double len = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
                    if (len > 0) { for (int a = 0; a < 3; a++) { n[a] = c[a] / len; p0[a] = v[0][a]; } have_plane = true; }
                }
            }
            else {
                const double ext = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
                for (int k = 0; k < 3; k++) {
                    const double d = n[0] * (v[k][0] - p0[0]) + n[1] * (v[k][1] - p0[1]) + n[2] * (v[k][2] - p0[2]);
                    if (std::fabs(d) > 1e-9 * ext) return false;
                }
            }
        }
        if (!have_plane) return false;
    }
    // the plane's unit normal in world space (a rigid matrix: its rotation part applied to the normal)
    double w[3] = { n[0], n[1], n[2] };
    if (l2w) {
        for (int a = 0; a < 3; a++) w[a] = (double)l2w->m[a][0] * n[0] + (double)l2w->m[a][1] * n[1] + (double)l2w->m[a][2] * n[2];
        const double len = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        if (!(std::fabs(len - 1.0) <= 1e-5)) return false;
        for (int a = 0; a < 3; a++) w[a] /= len;
    }
    for (int a = 0; a < 3; a++) world_normal[a] = (float)w[a];
    return true;
}

// anyhit_twins: 0 = none; 1 = for the bottom-level lists whose twin the surface-area model expects to cost an any-hit walk at most
// kTwinPays of what the list as given costs it (the second copy takes cache: measured +3 % per frame where it saves no visits, -5 %
// where it saves a quarter of them, profiles/r05_variants_direction_lists.txt); 2 = for every list that can have one.
// twin_dirs: 8 = one twin per octant of the ray's direction inside the instance (make_anyhit_twin's dir_sign: back to front along the
// ray where the direction decides, the model elsewhere), stored back to back behind the list; 1 = the one direction-free twin.
// Measured on the headline scene: shadow-ray node visits 261.6 M per frame as given, 198.8 M with one twin, 134.6 M with eight;
// 3.555 / 3.42 / 3.27 ms per frame.
constexpr double kTwinPays = 0.95;
// What the twins may cost in node memory: eight twins are nine times a list's records.  A list whose eight twins do not fit what is
// left of the scene's budget gets the one direction-free twin (twice the records), or none; and a scene small enough to be walked
// from an LDS copy (kLdsNodesMaxBytes of records) gets no twin that would push it out of that path.  (The 31-bit offset limit of the
// links is a separate, later check.)
constexpr uint64_t kTwinBudgetBytes = 512ull << 20;
inline bool build_host_image(HostSceneImage& img, const atn_scene_desc* s, std::string& err, int anyhit_twins = 0, int twin_dirs = 8,
                             int node_layout_top_levels = kLayoutTopLevels, bool planar_lights = true, uint64_t twin_budget_bytes = kTwinBudgetBytes)
{
    if (!s || s->n_bvh_lists == 0 || !s->bvh_lists) { err = "scene has no BVH lists"; return false; }
    const uint32_t nl = s->n_bvh_lists;
    std::string range_err;
    if (!validate_ranges(s->objects, s->n_objects, s->n_matrices, s, range_err)) { err = range_err; return false; }

    const uint32_t n_dir_wanted = twin_dirs == 8 ? 8u : 1u;
    uint64_t twin_bytes = 0, plain_bytes = 0;       // (48 bytes per record: an upper bound, inner records take 32)
    for (uint32_t k = 0; k < nl; k++) plain_bytes += (uint64_t)s->bvh_lists[k].count * 48u;
    std::vector<std::vector<ListLayout>> twin_lay(nl);
    std::vector<ListLayout> lay(nl);
    std::vector<std::vector<std::vector<atn_bvh_node>>> twin(nl);    // the any-hit twin(s) of list k as threaded lists of their own (empty: none)
    uint64_t total_nodes = 0;
    for (uint32_t k = 0; k < nl; k++) {
        if (!analyse_list(lay[k], s->bvh_lists[k].nodes, s->bvh_lists[k].count, k == 0, err)) return false;
        total_nodes += lay[k].order.size();
        if (k == 0 || !anyhit_twins || lay[k].order.size() != s->bvh_lists[k].count) continue;
        AnyhitTwin tw;
        if (!make_anyhit_twin(s->bvh_lists[k].nodes, s->bvh_lists[k].count, tw)) continue;
        if (anyhit_twins != 2 && !(tw.cost_twin <= kTwinPays * tw.cost_as_given)) continue;
        const uint64_t one = (uint64_t)s->bvh_lists[k].count * 48u;
        uint32_t n_dir = n_dir_wanted;
        auto fits = [&](uint32_t nd) {
            if (twin_bytes + one * nd > twin_budget_bytes) return false;
            return !(plain_bytes <= kLdsNodesMaxBytes && plain_bytes + twin_bytes + one * nd > kLdsNodesMaxBytes);
        };
        if (!fits(n_dir)) n_dir = 1u;
        if (!fits(n_dir)) continue;
        twin_bytes += one * n_dir;
        twin[k].resize(n_dir); twin_lay[k].resize(n_dir);
        bool ok = true;
        for (uint32_t g = 0; g < n_dir && ok; g++) {
            if (n_dir > 1) {
                const int sg[3] = { (g & 1u) ? 1 : -1, (g & 2u) ? 1 : -1, (g & 4u) ? 1 : -1 };
                ok = make_anyhit_twin(s->bvh_lists[k].nodes, s->bvh_lists[k].count, tw, sg);
            }
            std::string twin_err;
            ok = ok && analyse_list(twin_lay[k][g], tw.nodes.data(), (uint32_t)tw.nodes.size(), false, twin_err);
            twin[k][g].swap(tw.nodes);
        }
        if (!ok) { twin[k].clear(); twin_lay[k].clear(); }
    }
    uint64_t off = 0;
    img.list_root.assign(nl, 0);
    img.list_bytes.assign(nl, 0); img.list_tri_leaves.assign(nl, 0); img.list_inner.assign(nl, 0); img.list_twin_delta.assign(nl, 0);
    for (uint32_t kk = 1; kk <= nl; kk++) {
        const uint32_t k = kk % nl;         // 1, 2, ..., nl-1, 0
        img.list_root[k] = (uint32_t)off;
        for (uint32_t j = 0; j < lay[k].order.size(); j++) {
            if (lay[k].kind[j] == KIND_TRI) img.list_tri_leaves[k]++;
            else if (lay[k].kind[j] == KIND_INNER) img.list_inner[k]++;
        }
        assign_offsets(lay[k], off, k == 0 ? 0 : node_layout_top_levels);
        if (off >= (1ull << 31)) { err = "too many BVH nodes for 31-bit byte-offset links"; return false; }
        img.list_bytes[k] = (uint32_t)(off - img.list_root[k]);
        if (!twin[k].empty()) {
            // the twin's records follow the list's own (not part of list_bytes: the list's region is what an LBVH rebuild rewrites);
            // both roots are the same record kind, so the twin's typed root link is the list's plus the distance -- which is the list's
            // own size, and the size of every further twin: twin g starts at root + (1 + g) * distance
            if (twin_lay[k][0].kind[0] != lay[k].kind[0] || off + (uint64_t)img.list_bytes[k] * twin[k].size() >= (1ull << 31)) { twin[k].clear(); continue; }
            img.list_twin_delta[k] = (int32_t)(off - img.list_root[k]) | (twin[k].size() > 1 ? 1 : 0);
            for (size_t g = 0; g < twin[k].size(); g++)
                assign_offsets(twin_lay[k][g], off, node_layout_top_levels);
        }
    }
    if (off >= (1ull << 31)) { err = "too many BVH nodes for 31-bit byte-offset links"; return false; }
    img.nodes.assign((size_t)(off / 16), make_float4(0, 0, 0, 0));
    img.list_root_link.assign(nl, kLinkEnd);
    img.n_nodes = total_nodes;

    ListEmitCtx c;
    c.objects = s->objects; c.n_objects = s->n_objects; c.n_matrices = s->n_matrices; c.matrices = s->matrices;
    c.tris = s->triangles; c.vtx_pos = s->vtx_pos; c.n_triangles = s->n_triangles; c.n_vertices = s->n_vertices;
    c.n_lists = nl;
    uint64_t counts[3] = { 0, 0, 0 };
    img.tlas_refs.clear();
    for (uint32_t kk = 1; kk <= nl; kk++) {
        const uint32_t k = kk % nl;
        c.list_root_link = img.list_root_link.data();
        c.list_twin_delta = img.list_twin_delta.data();
        c.tlas_refs = &img.tlas_refs;
        int32_t root = kLinkEnd;
        if (!emit_list(reinterpret_cast<char*>(img.nodes.data()), lay[k], s->bvh_lists[k].nodes, c, root, counts, err)) return false;
        img.list_root_link[k] = root;
        for (size_t g = 0; g < twin[k].size(); g++) {
            int32_t twin_root = kLinkEnd;
            uint64_t twin_counts[3] = { 0, 0, 0 };      // (not part of the scene's record statistics)
            if (!emit_list(reinterpret_cast<char*>(img.nodes.data()), twin_lay[k][g], twin[k][g].data(), c, twin_root, twin_counts, err)) return false;
            if (twin_root != root + (int32_t)(1 + g) * (img.list_twin_delta[k] & ~15)) { err = "internal: any-hit twin root"; return false; }
        }
    }
    img.n_inner = counts[0]; img.n_tri_leaf = counts[1]; img.n_tlas_leaf = counts[2];
    img.params.node_bytes = (uint32_t)off;
    img.params.ident_row = c.ident_row;

    // ---- plain copies
    img.tris.assign(s->triangles, s->triangles + s->n_triangles);
    img.vtx_pos.resize(s->n_vertices); img.vtx_nml.resize(s->n_vertices);
    for (uint32_t i = 0; i < s->n_vertices; i++) {
        img.vtx_pos[i] = make_float4(s->vtx_pos[i].x, s->vtx_pos[i].y, s->vtx_pos[i].z, s->vtx_pos[i].w);
        img.vtx_nml[i] = make_float4(s->vtx_nml[i].x, s->vtx_nml[i].y, s->vtx_nml[i].z, s->vtx_nml[i].w);
    }
    img.objects.assign(s->objects, s->objects + s->n_objects);
    img.matrices.resize((size_t)s->n_matrices * 4);
    for (uint32_t i = 0; i < s->n_matrices; i++)
        for (int r = 0; r < 4; r++)
            img.matrices[4 * (size_t)i + r] = make_float4(s->matrices[i].m[r][0], s->matrices[i].m[r][1], s->matrices[i].m[r][2], s->matrices[i].m[r][3]);
    std::vector<uint8_t> tex_has_alpha(s->n_textures, 0);
    for (uint32_t i = 0; i < s->n_textures; i++) {
        const atn_texture_desc& t = s->textures[i];
        const size_t n = (size_t)t.width * t.height;
        for (size_t j = 0; j < n; j++) if (t.texels[j].w < 1.0F) { tex_has_alpha[i] = 1; break; }
    }
    img.materials.resize(s->n_materials);
    img.carpaint.assign((size_t)(s->n_materials + 1) * 4, make_float4(0, 0, 0, 0));
    for (uint32_t i = 0; i < s->n_materials; i++) {
        const float* c = s->materials[i].u.carpaint;
        for (int k = 0; k < 4; k++) img.carpaint[4 * (size_t)i + k] = make_float4(c[4 * k], c[4 * k + 1], c[4 * k + 2], c[4 * k + 3]);
    }
    for (uint32_t i = 0; i < s->n_materials; i++) {
        const atn_material_param& m = s->materials[i];
        DevMaterial& d = img.materials[i];
        d.baseColor = make_float4(m.baseColor.x, m.baseColor.y, m.baseColor.z, m.baseColor.w);
        d.type = m.type; d.attrib = (m.attrib & 0xFu) | (m.isIdealRefraction ? kAttrIdealRefraction : 0u); d.id = m.id;
        if (m.baseColor.w < 1.0F || (m.albedoMap >= 0 && (uint32_t)m.albedoMap < s->n_textures && tex_has_alpha[m.albedoMap])) d.attrib |= kAttrMaybeAlpha;
        if (m.stencil_type == 1) d.attrib |= kAttrStencilAlways;
        if (m.stencil_type == 2) d.attrib |= kAttrStencilStencil;
        d.albedoMap = m.albedoMap; d.normalMap = m.normalMap; d.roughnessMap = m.roughnessMap;
        const atn_standard_mtrl& st = m.u.standard;
        d.ior = st.ior; d.roughness = st.roughness; d.subsurface = st.subsurface; d.metallic = st.metallic;
        d.specular = st.specular; d.specularTint = st.specularTint; d.sheen = st.sheen; d.sheenTint = st.sheenTint;
        d.clearcoat = st.clearcoat; d.clearcoatGloss = st.clearcoatGloss;
    }
    {   // FillMaterial's fallback for mtrl_id < 0 (material_impl.h:253-259), stored at index n_materials
        DevMaterial d{};
        d.baseColor = make_float4(1, 1, 1, 1); d.type = ATN_MTRL_DIFFUSE; d.attrib = 0; d.id = 0;
        d.albedoMap = d.normalMap = d.roughnessMap = -1; d.ior = 1.0F; d.roughness = 0.5F;
        d.subsurface = d.metallic = d.specular = d.specularTint = 0.5F;
        d.sheen = d.sheenTint = d.clearcoat = d.clearcoatGloss = 0.5F;
        img.materials.push_back(d);
    }
    img.lights.assign(s->lights, s->lights + s->n_lights);
    // area lights whose shadow rays may stop early (planar_area_light, below): the flag lives in OUR copy's padding word
    img.light_plane.assign(s->n_lights ? s->n_lights : 1, make_float4(0, 0, 0, 0));       // {world-space unit normal, 1} of a planar area light, else 0
    for (uint32_t i = 0; i < s->n_lights; i++) {
        float wn[3];
        img.lights[i]._pad = 0;
        if (planar_lights && planar_area_light(s, s->lights[i], wn)) { img.light_plane[i] = make_float4(wn[0], wn[1], wn[2], 1.0F); img.lights[i]._pad = 1; }
    }
    img.params.planar_lights = planar_lights ? 1 : 0;
    // textures: RGBA8 where every channel of every texel is exactly k / 255.0f (DevTexture), float4 otherwise
    img.textures.resize(s->n_textures);
    img.texels.clear(); img.texels8.clear();
    for (uint32_t i = 0; i < s->n_textures; i++) {
        const atn_texture_desc& t = s->textures[i];
        const size_t n = (size_t)t.width * t.height;
        // which 8-bit -> float conversion reproduces EVERY channel of EVERY texel?  1: k / 255.0f,  2: k * (1.0f / 255)
        // (aten::Image::Load multiplies by `norm = 1.0F / 255`, image/image.cpp:76-80)
        int32_t fmt = 0;
        const size_t at8 = img.texels8.size();
        std::vector<uint32_t> packed(n);
        for (int32_t cand = 1; cand <= 2 && !fmt && n > 0; cand++) {
            auto decode = [cand](uint32_t k) { return cand == 1 ? (float)k / 255.0F : (float)k * (1.0F / 255); };
            auto code = [&](float v, uint32_t& k) {
                if (!(v >= 0.0F && v <= 1.0F)) return false;
                k = (uint32_t)(v * 255.0F + 0.5F);
                return k <= 255u && decode(k) == v;
            };
            bool ok = true;
            for (size_t j = 0; j < n && ok; j++) {
                uint32_t r = 0, g = 0, b = 0, a = 0;
                ok = code(t.texels[j].x, r) && code(t.texels[j].y, g) && code(t.texels[j].z, b) && code(t.texels[j].w, a);
                packed[j] = r | (g << 8) | (b << 16) | (a << 24);
            }
            if (ok) fmt = cand;
        }
        const bool unorm8 = fmt != 0;
        img.textures[i].width = t.width; img.textures[i].height = t.height;
        if (unorm8) {
            img.textures[i].offset = (uint32_t)at8; img.textures[i].format = fmt;
            img.texels8.insert(img.texels8.end(), packed.begin(), packed.end());
        }
        else {
            const size_t at = img.texels.size();
            img.textures[i].offset = (uint32_t)at; img.textures[i].format = 0;
            img.texels.resize(at + n);
            for (size_t j = 0; j < n; j++) img.texels[at + j] = make_float4(t.texels[j].x, t.texels[j].y, t.texels[j].z, t.texels[j].w);
        }
    }

    DevScene& p = img.params;
    p.root_link = img.list_root_link[0];
    fill_root_direct(p, img.nodes.data(), 0, s->matrices, s->n_matrices);
    p.n_lights = (int32_t)s->n_lights; p.inv_n_lights = s->n_lights ? 1.0f / (float)s->n_lights : 0.0f; p.n_textures = (int32_t)s->n_textures; p.n_materials = (int32_t)s->n_materials;
    p.bvh_hit_min = s->config.bvh_hit_min;
    p.bg_color[0] = s->config.bg.bg_color[0]; p.bg_color[1] = s->config.bg.bg_color[1]; p.bg_color[2] = s->config.bg.bg_color[2];
    p.envmap_tex_idx = s->config.bg.envmap_tex_idx;
    p.avgIllum = s->config.bg.avgIllum;
    p.multiplyer = s->config.bg.multiplyer;
    p.enable_env_map = s->config.bg.enable_env_map;
    p.any_alpha = 0;
    for (const DevMaterial& dm : img.materials) if (dm.attrib & (kAttrMaybeAlpha | kAttrStencilStencil)) p.any_alpha = 1;
    p.enable_alpha_blending = s->config.enable_alpha_blending ? 1 : 0;
    // NPR: ToonParameter per material, the target lights, the screen-space shadow texture (x channel)
    img.toon.assign((size_t)s->n_materials + 1, atn_toon_param{});
    for (uint32_t i = 0; i < s->n_materials; i++) {
        img.toon[i] = s->materials[i].toon;
        const int32_t t = s->materials[i].type;
        if (t == ATN_MTRL_TOON || t == ATN_MTRL_STYLIZED_BRDF) {
            const atn_toon_param& tp = img.toon[i];
            if (tp.target_light_idx >= 0 && (uint32_t)tp.target_light_idx >= s->n_npr_target_lights) { err = "toon material's target light index out of range"; return false; }
            if (tp.target_light_idx >= 0 && !s->npr_target_lights) { err = "null NPR target light array"; return false; }
        }
    }
    if (s->n_npr_target_lights && s->npr_target_lights) {
        img.npr_lights.assign(s->npr_target_lights, s->npr_target_lights + s->n_npr_target_lights);
        for (const atn_light_param& l : img.npr_lights)
            if (l.arealight_objid >= 0 && (uint32_t)l.arealight_objid >= s->n_objects) { err = "NPR target light refers to an object id out of range"; return false; }
    }
    p.n_npr_lights = (int32_t)img.npr_lights.size();
    p.enable_shadowray_base_stylized_shadow = s->enable_shadowray_base_stylized_shadow ? 1 : 0;
    p.ss_w = p.ss_h = 0;
    if (s->screen_space_texture.texels && s->screen_space_texture.width > 0 && s->screen_space_texture.height > 0) {
        p.ss_w = s->screen_space_texture.width; p.ss_h = s->screen_space_texture.height;
        const size_t n = (size_t)p.ss_w * p.ss_h;
        img.screen_shadow.resize(n);
        for (size_t i = 0; i < n; i++) img.screen_shadow[i] = s->screen_space_texture.texels[i].x;
    }
    p.material_set = kMsCore;
    for (const DevMaterial& dm : img.materials) {
        const int32_t t = dm.type;
        const bool core = t == ATN_MTRL_EMISSIVE || t == ATN_MTRL_DIFFUSE || t == ATN_MTRL_SPECULAR || t == ATN_MTRL_GGX;
        const int32_t need = (t == ATN_MTRL_TOON || t == ATN_MTRL_STYLIZED_BRDF) ? kMsToon
                           : t == ATN_MTRL_CARPAINT ? kMsCarPaint : t == ATN_MTRL_DISNEY ? kMsDisney : core ? kMsCore : kMsAnalytic;
        if (need > p.material_set) p.material_set = need;
    }
    // ImageBasedLight::sample's scene_radius (light/ibl.h:106-111; aabb::IsValid / getCenter /
    // ComputeDistanceToCoverBoundingSphere, math/aabb.h:176-180,231-234,346-362), evaluated once on the host.
    {
        const float* mn = s->scene_bbox_min; const float* mx = s->scene_bbox_max;
        float radius = 10000.0F;
        const bool valid = !((mn[0] >= mx[0]) || (mn[1] >= mx[1]) || (mn[2] >= mx[2]));
        if (valid) {
            const float cx = (mn[0] + mx[0]) * 0.5F, cy = (mn[1] + mx[1]) * 0.5F, cz = (mn[2] + mx[2]) * 0.5F;
            const float dx = mx[0] - cx, dy = mx[1] - cy, dz = mx[2] - cz;
            const float r = std::sqrt((dx * dx + dy * dy) + dz * dz);
            const float theta = (3.14159265358979323846F * (30.0F) / 180.0F);
            radius = r / std::tan(theta / 2);
        }
        p.ibl_scene_radius = radius;
    }
    return true;
}

} // namespace atn
