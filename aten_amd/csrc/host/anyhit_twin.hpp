// A second THREADING of a bottom-level list for the rays that only ask "is anything in the way" (shadow rays).
//
// A threaded list is walked in one fixed child order by every ray.  For a closest-hit walk the reference's order is part of the
// answer (exact-t ties, the 1e-6 bias of aabb::hit) and is kept as given.  An ANY-hit walk -- scene::hitLight's question
// "is the closest hit nearer than the light" (scene/scene.h:64-134) as traverse.hpp asks it: stop at the first accepted hit
// with t <= stop_t -- gives the same answer in any child order: until that hit t_max is the constant the ray came with, so
// which boxes pass aabb::hit, and therefore which triangles are reachable, does not depend on the order; only how SOON an
// occluder is met does.  Measured on MI355X (profiles/r05_variants_direction_lists.txt): shadow rays of sponza_lod visit 27 %
// fewer nodes when the child that is cheap and likely to block comes first, 46 % fewer back to front along the ray.
//
// make_anyhit_twin re-threads the SAME tree -- same boxes, same leaves, same parent-child relations; only the order of the two
// children of an inner node may change -- by the expected cost of an any-hit walk under the usual surface-area model:
//   a child c of a node n whose box was hit is reached, costs V(c) and blocks the ray with probability p(c):
//     triangle leaf: V = kLeafCost (it is tested whenever it is reached: threaded_bvh_traverser.h:190-225),  p = h * 1/2
//     inner node:    V = 1 + h * D(c),                                                                         p = h * occ(c)
//     with h = area(c) / area(n), occ(c) = 1 - (1 - p(c1)) (1 - p(c2)), and D(n) = V(first) + (1 - p(first)) V(second);
//   the twin takes, at every inner node, the order with the smaller D.
// cost_as_given / cost_twin are 1 + D(root) for the list's own order and for the twin's: the caller builds the twin only when
// the model says it pays (the extra copy costs cache: the atrium's regular grids gain nothing and lose 3 % to it).
// Pure host C++; used by the upload (scene_upload.hpp) and exported for tools and tests as atns_anyhit_twin.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "../../../include/aten_layout.h"

namespace atn {

struct AnyhitTwin {
    std::vector<atn_bvh_node> nodes;        // the twin: a threaded list in depth-first pre-order (an inner node's hit link = index + 1)
    double cost_as_given = 0, cost_twin = 0;
    uint32_t flipped = 0;                   // inner nodes whose children changed places
};

constexpr float kAnyhitLeafCost = 3.0f;     // a triangle test against a box test, in issue slots (~80 against ~26 VALU, traverse.hpp)

// false: `src` is not a binary tree in pre-order along its hit links (nothing is written); the walk itself accepts more
// general lists (scene_upload.hpp: analyse_list), those simply get no twin.
// dir_sign (optional): the twin for the any-hit rays that travel with these signs (+1 / -1 per axis): where the EXIT faces of the two
// children along that direction lie more than dir_tolerance of the node's extent apart, the child that reaches further comes first
// (back to front: what blocks a ray that leaves a surface lies far along it, profiles/r05_variants_direction_lists.txt); elsewhere
// the model decides as in the direction-free twin.
inline bool make_anyhit_twin(const atn_bvh_node* src, uint32_t count, AnyhitTwin& out, const int* dir_sign = nullptr, float dir_tolerance = 0.05f)
{
    if (!src || count < 3) return false;
    // pre-order along the hit links; a subtree ends where the miss link points
    std::vector<uint32_t> order; order.reserve(count);
    std::vector<int32_t> pos_of(count, -1);
    for (int32_t id = 0; id >= 0; id = (int32_t)src[id].hit) {
        if ((uint32_t)id >= count || pos_of[id] >= 0) return false;
        pos_of[id] = (int32_t)order.size();
        order.push_back((uint32_t)id);
    }
    const uint32_t n = (uint32_t)order.size();
    if (n != count) return false;
    std::vector<uint32_t> end(n);
    std::vector<uint8_t> leaf(n);
    for (uint32_t j = 0; j < n; j++) {
        const atn_bvh_node& nd = src[order[j]];
        leaf[j] = (nd.f0 >= 0 || nd.f1 >= 0) ? 1 : 0;
        if (leaf[j]) {
            if (nd.f2 >= 0) return false;               // a nested tree: not a bottom-level list
            if ((int32_t)nd.hit != (int32_t)nd.miss) return false;
            end[j] = j + 1;
        }
        else {
            const int32_t m = (int32_t)nd.miss;
            if (m >= (int32_t)count || (m >= 0 && pos_of[m] <= (int32_t)j)) return false;
            end[j] = m < 0 ? n : (uint32_t)pos_of[m];
        }
    }
    for (uint32_t j = 0; j < n; j++) {
        if (leaf[j]) continue;
        if (j + 1 >= n || end[j + 1] >= end[j] || end[end[j + 1]] != end[j]) return false;      // exactly two children
    }
    auto half_area = [&](uint32_t j) {
        const atn_bvh_node& nd = src[order[j]];
        const float dx = nd.boxmax[0] - nd.boxmin[0], dy = nd.boxmax[1] - nd.boxmin[1], dz = nd.boxmax[2] - nd.boxmin[2];
        const float a = dx * dy + dy * dz + dz * dx;
        return a > 0.0f ? a : 0.0f;
    };
    // bottom-up (children lie behind their parent in pre-order)
    std::vector<float> occ(n, 0.0f), dg(n, 0.0f), dt(n, 0.0f), area(n);
    std::vector<uint8_t> flip(n, 0);
    for (uint32_t j = 0; j < n; j++) area[j] = half_area(j);
    out.flipped = 0;
    for (uint32_t j = n; j-- > 0;) {
        if (leaf[j]) continue;
        const uint32_t c[2] = { j + 1, end[j + 1] };
        const float an = std::max(area[j], 1e-30f);
        float p[2], vg[2], vt[2];
        for (int k = 0; k < 2; k++) {
            const float h = std::min(area[c[k]] / an, 1.0f);
            if (leaf[c[k]]) {
                const bool tri = src[order[c[k]]].f1 >= 0;
                p[k] = tri ? 0.5f * h : 0.0f;
                vg[k] = vt[k] = tri ? kAnyhitLeafCost : 1.0f;
            }
            else {
                p[k] = h * occ[c[k]];
                vg[k] = 1.0f + h * dg[c[k]];
                vt[k] = 1.0f + h * dt[c[k]];
            }
        }
        occ[j] = 1.0f - (1.0f - p[0]) * (1.0f - p[1]);
        dg[j] = vg[0] + (1.0f - p[0]) * vg[1];
        const float ab = vt[0] + (1.0f - p[0]) * vt[1], ba = vt[1] + (1.0f - p[1]) * vt[0];
        flip[j] = ba < ab ? 1 : 0;
        if (dir_sign) {
            const atn_bvh_node& na = src[order[c[0]]]; const atn_bvh_node& nb = src[order[c[1]]]; const atn_bvh_node& nn = src[order[j]];
            float ea = 0, eb = 0, ext = 0;
            for (int a = 0; a < 3; a++) {
                if (dir_sign[a] > 0) { ea += na.boxmax[a]; eb += nb.boxmax[a]; }
                else { ea -= na.boxmin[a]; eb -= nb.boxmin[a]; }
                ext += nn.boxmax[a] - nn.boxmin[a];
            }
            if (eb - ea > dir_tolerance * ext) flip[j] = 1;
            else if (ea - eb > dir_tolerance * ext) flip[j] = 0;
        }
        dt[j] = flip[j] ? ba : ab;
        out.flipped += flip[j];
    }
    out.cost_as_given = 1.0 + dg[0];
    out.cost_twin = 1.0 + dt[0];
    // the twin in ITS pre-order: subtree sizes do not change, so a node's subtree ends at (its new position) + size
    out.nodes.resize(n);
    std::vector<uint32_t> stack; stack.reserve(64);
    stack.push_back(0);
    uint32_t at = 0;
    while (!stack.empty()) {
        const uint32_t j = stack.back(); stack.pop_back();
        atn_bvh_node o = src[order[j]];
        const uint32_t size = end[j] - j;
        const float next = at + 1 < n ? (float)(at + 1) : -1.0f;
        const float after = at + size < n ? (float)(at + size) : -1.0f;
        o.hit = next;
        o.miss = leaf[j] ? next : after;
        out.nodes[at++] = o;
        if (!leaf[j]) {
            uint32_t a = j + 1, b = end[j + 1];
            if (flip[j]) std::swap(a, b);
            stack.push_back(b); stack.push_back(a);
        }
    }
    return at == n;
}

} // namespace atn
