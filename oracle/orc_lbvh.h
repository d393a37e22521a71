/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h).
 *
 * orc_lbvh.h: sequential CPU restatement of the reference's GPU LBVH builder, idaten::LBVHBuilder::onBuild
 * (src/libidaten/kernel/LBVHBuilder.cu:700-810), the builder the reference runs once per tick for a deformable mesh
 * (src/deformation_renderer/main.cpp:636-710).  Stages, in the reference's order:
 *   genMortonCode        kernel/MortonCode.cuh:147-197 (32-bit codes: AT_ENABLE_64BIT_LBVH_MORTON_CODE is not defined,
 *                        LBVHBuilder.h:10), computeMortonCode :62-80, expandBits :28-47
 *   sort                 kernel/RadixSort.cu:425-437 -> thrust::sort_by_key on uint32 keys, i.e. a radix sort: equal keys
 *                        keep their input order (values ascending).  std::stable_sort here.
 *   buildTree            LBVHBuilder.cu:193-350 (the live #else branch: computeLongestCommonPrefix, findSpan, findSplit)
 *   applyTraverseOrder   LBVHBuilder.cu:353-489; leaves carry object_id = 1 because GPGPU_TRAVERSE_SBVH is defined
 *                        (libaten/accelerator/GpuPayloadDefs.h:9, LBVHBuilder.cu:391-394)
 *   computeBoudingBox    LBVHBuilder.cu:491-680: leaf box = min / max of the three vertices, inner box = union of its
 *                        children's (min / max are exact, so the order in which the device merges them does not matter)
 *
 * PARITY STATUS: the builder is CUDA (.cu, thrust) and cannot be compiled in this image, and the reference holds no
 * expected outputs for it; its only test data are the eight keys of the disabled self-test LBVHBuilder.cu:877-880,
 * {1, 19, 24, 25, 30, 2, 4, 5} -- which, sorted, are the keys of Figure 3 of Karras, "Maximizing Parallelism in the
 * Construction of BVHs, Octrees, and k-d Trees" (HPG 2012), the paper the file cites (:11-12).  tests/ pin the hierarchy
 * stage against that published figure; the other stages are "parity unpinned" restatements.
 */
#pragma once
#include "../include/aten_layout.h"
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace orc { namespace lbvh {

struct Node { int32_t order, left, right, parent; bool isLeaf; };      // LBVHBuilder::LBVHNode, LBVHBuilder.h:42-50

inline uint32_t expandBits(uint32_t value)                             // MortonCode.cuh:28-47
{
    value = (value | value << 16) & 0xFF0000FFu;
    value = (value | value << 8) & 0x0F00F00Fu;
    value = (value | value << 4) & 0xC30C30C3u;
    value = (value | value << 2) & 0x49249249u;
    return value;
}

// CUDA's min / max on floats are fminf / fmaxf: a NaN operand is dropped (0 / 0 on a flat axis gives bit 0)
inline uint32_t computeMortonCode(float x, float y, float z)           // MortonCode.cuh:62-80
{
    uint32_t dx = (uint32_t)std::fmin(std::fmax(x * 1024.0f, 0.0f), 1023.0f);
    uint32_t dy = (uint32_t)std::fmin(std::fmax(y * 1024.0f, 0.0f), 1023.0f);
    uint32_t dz = (uint32_t)std::fmin(std::fmax(z * 1024.0f, 0.0f), 1023.0f);
    dx = expandBits(dx); dy = expandBits(dy); dz = expandBits(dz);
    return dx << 2 | dy << 1 | dz;
}

inline void triangle_box(const atn_triangle_param& t, const atn_vec4* vtx, int32_t vtxOffset, float mn[3], float mx[3])
{
    const atn_vec4& v0 = vtx[t.idx[0] + vtxOffset];
    const atn_vec4& v1 = vtx[t.idx[1] + vtxOffset];
    const atn_vec4& v2 = vtx[t.idx[2] + vtxOffset];
    mn[0] = std::fmin(std::fmin(v0.x, v1.x), v2.x); mn[1] = std::fmin(std::fmin(v0.y, v1.y), v2.y); mn[2] = std::fmin(std::fmin(v0.z, v1.z), v2.z);
    mx[0] = std::fmax(std::fmax(v0.x, v1.x), v2.x); mx[1] = std::fmax(std::fmax(v0.y, v1.y), v2.y); mx[2] = std::fmax(std::fmax(v0.z, v1.z), v2.z);
}

// genMortonCode, MortonCode.cuh:147-197 (onComputeMortonCode :107-123: centre of the triangle's box, normalised by the
// scene box the caller passes; the axis order a0..a2 is only used by the 64-bit variant)
inline void gen_morton(const atn_triangle_param* tris, uint32_t n, const float bmin[3], const float bmax[3],
                       const atn_vec4* vtx, int32_t vtxOffset, std::vector<uint32_t>& codes, std::vector<uint32_t>& indices)
{
    codes.resize(n); indices.resize(n);
    const float size[3] = { bmax[0] - bmin[0], bmax[1] - bmin[1], bmax[2] - bmin[2] };   // aabb::size
    for (uint32_t i = 0; i < n; i++) {
        float mn[3], mx[3];
        triangle_box(tris[i], vtx, vtxOffset, mn, mx);
        float c[3];
        for (int k = 0; k < 3; k++) {
            c[k] = (mn[k] + mx[k]) * 0.5f;
            c[k] = (c[k] - bmin[k]) / size[k];
        }
        codes[i] = computeMortonCode(c[0], c[1], c[2]);
        indices[i] = i;
    }
}

inline void sort_by_key(std::vector<uint32_t>& codes, std::vector<uint32_t>& indices)
{
    const size_t n = codes.size();
    std::vector<uint32_t> perm(n);
    for (size_t i = 0; i < n; i++) perm[i] = (uint32_t)i;
    std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return codes[a] < codes[b]; });
    std::vector<uint32_t> k(n), v(n);
    for (size_t i = 0; i < n; i++) { k[i] = codes[perm[i]]; v[i] = indices[perm[i]]; }
    codes.swap(k); indices.swap(v);
}

inline int32_t clz32(uint32_t x) { return x ? __builtin_clz(x) : 32; }     // __clz(0) == 32

inline int32_t lcp(const uint32_t* keys, uint32_t num, int32_t i1, int32_t i2)      // LBVHBuilder.cu:194-216
{
    const int32_t left = std::min(i1, i2), right = std::max(i1, i2);
    if (left < 0 || right >= (int32_t)num) return -1;
    const uint32_t a = keys[left], b = keys[right];
    return a != b ? clz32(a ^ b) : (32 + clz32((uint32_t)(left ^ right)));
}

inline void findSpan(const uint32_t* keys, uint32_t num, int32_t idx, int32_t span[3])  // LBVHBuilder.cu:218-263
{
    const int32_t lcp1 = lcp(keys, num, idx, idx + 1), lcp2 = lcp(keys, num, idx, idx - 1);
    const int32_t d = (lcp1 - lcp2) < 0 ? -1 : 1;
    const int32_t minLcp = lcp(keys, num, idx, idx - d);
    int32_t lmax = 2;
    while (lcp(keys, num, idx, idx + lmax * d) > minLcp) lmax *= 2;
    int32_t l = 0, t = lmax;
    do {
        t /= 2;
        if (lcp(keys, num, idx, idx + (l + t) * d) > minLcp) l = l + t;
    } while (t > 1);
    span[0] = std::min(idx, idx + l * d); span[1] = std::max(idx, idx + l * d); span[2] = d;
}

inline int32_t findSplit(const uint32_t* keys, uint32_t num, const int32_t span[3])    // LBVHBuilder.cu:266-297
{
    int32_t left = span[0], right = span[1];
    const int32_t numIdentical = lcp(keys, num, left, right);
    do {
        const int32_t newSplit = (right + left) / 2;
        if (lcp(keys, num, left, newSplit) > numIdentical) left = newSplit;
        else right = newSplit;
    } while (right > left + 1);
    return left;
}

inline void build_tree(const uint32_t* keys, uint32_t num, std::vector<Node>& nodes)   // buildTree, LBVHBuilder.cu:299-350
{
    nodes.assign(2 * (size_t)num - 1, Node{ 0, -1, -1, -1, false });
    for (int32_t idx = 0; idx < (int32_t)num - 1; idx++) {
        int32_t range[3];
        findSpan(keys, num, idx, range);
        const int32_t split = findSplit(keys, num, range);
        Node& node = nodes[idx];
        if (idx == 0) node.parent = -1;
        node.order = idx; node.isLeaf = false;
        for (int side = 0; side < 2; side++) {
            const int32_t s = split + side;
            const bool leaf = side == 0 ? (split == range[0]) : (split + 1 == range[1]);
            const int32_t child = leaf ? s + (int32_t)num - 1 : s;
            (side == 0 ? node.left : node.right) = child;
            Node& c = nodes[child];
            c.order = child; c.parent = idx; c.isLeaf = leaf;
            if (leaf) { c.left = -1; c.right = -1; }
        }
    }
}

// onApplyTraverseOrder, LBVHBuilder.cu:353-470
inline void apply_traverse_order(int32_t idx, int32_t numberOfTris, int32_t triIdOffset, const std::vector<Node>& src,
                                 const uint32_t* sortedIndices, atn_bvh_node* dst)
{
    const Node* node = &src[idx];
    const Node* next = node->left >= 0 ? &src[node->left] : nullptr;
    atn_bvh_node* g = &dst[idx];
    g->f0 = -1; g->f2 = -1; g->f3 = -1;                // object_id, exid, meshid
    if (node->isLeaf) {
        const int32_t leafId = node->order - (numberOfTris - 1);
        g->f1 = (float)(triIdOffset + (int32_t)sortedIndices[leafId]);
        g->f0 = 1;                                     // GPGPU_TRAVERSE_SBVH: "isleaf"
    }
    else g->f1 = -1;
    g->hit = -1; g->miss = -1;
    bool isOrdered = false;
    if (node->isLeaf) {
        const Node* parent = &src[node->parent];
        const Node* left = parent->left >= 0 ? &src[parent->left] : nullptr;
        const Node* right = parent->right >= 0 ? &src[parent->right] : nullptr;
        if (left == node) { g->hit = (float)right->order; g->miss = (float)right->order; isOrdered = true; }
    }
    else g->hit = next ? (float)next->order : -1.0f;
    if (isOrdered) return;
    const Node* parent = node->parent >= 0 ? &src[node->parent] : nullptr;
    if (!parent) { g->miss = -1; return; }
    const Node* left = parent->left >= 0 ? &src[parent->left] : nullptr;
    const Node* right = parent->right >= 0 ? &src[parent->right] : nullptr;
    if (left == node && right) { g->miss = (float)right->order; return; }
    const Node* cur = parent;
    for (;;) {
        const Node* grand = cur->parent >= 0 ? &src[cur->parent] : nullptr;
        if (!grand) { g->miss = -1; break; }
        const Node* sibling = grand->right >= 0 ? &src[grand->right] : nullptr;
        if (sibling && sibling != cur) {
            g->miss = (float)sibling->order;
            if (node->isLeaf && g->hit < 0) g->hit = (float)sibling->order;
            break;
        }
        cur = grand;
    }
}

// computeBoudingBox, LBVHBuilder.cu:533-680, as a post-order recursion (explicit stack: LBVH trees can be deep)
inline void compute_boxes(const std::vector<Node>& src, int32_t numberOfTris, const uint32_t* sortedIndices,
                          const atn_triangle_param* tris, const atn_vec4* vtx, int32_t vtxOffset, atn_bvh_node* dst)
{
    const int32_t leafBase = numberOfTris - 1;
    for (int32_t i = 0; i < numberOfTris; i++) {
        const int32_t triId = (int32_t)sortedIndices[src[leafBase + i].order - leafBase];
        triangle_box(tris[triId], vtx, vtxOffset, dst[leafBase + i].boxmin, dst[leafBase + i].boxmax);
    }
    std::vector<int32_t> stack{ 0 };
    std::vector<uint8_t> expanded(src.size(), 0);
    while (!stack.empty()) {
        const int32_t i = stack.back();
        if (src[i].isLeaf) { stack.pop_back(); continue; }
        if (!expanded[i]) { expanded[i] = 1; stack.push_back(src[i].left); stack.push_back(src[i].right); continue; }
        stack.pop_back();
        const atn_bvh_node& a = dst[src[i].left];
        const atn_bvh_node& b = dst[src[i].right];
        for (int k = 0; k < 3; k++) {
            dst[i].boxmin[k] = std::fmin(a.boxmin[k], b.boxmin[k]);
            dst[i].boxmax[k] = std::fmax(a.boxmax[k], b.boxmax[k]);
        }
    }
}

// LBVHBuilder::onBuild, LBVHBuilder.cu:700-810.  `out` holds 2 n - 1 nodes: inner nodes 0 .. n-2, leaves n-1 .. 2n-2.
inline bool build(const atn_triangle_param* tris, uint32_t n, int32_t triIdOffset, const float bmin[3], const float bmax[3],
                  const atn_vec4* vtx, int32_t vtxOffset, atn_bvh_node* out, uint32_t* out_codes, uint32_t* out_indices)
{
    if (n < 2) return false;        // the reference's buildTree does nothing for one triangle and leaves node 0 unset
    std::vector<uint32_t> codes, indices;
    gen_morton(tris, n, bmin, bmax, vtx, vtxOffset, codes, indices);
    sort_by_key(codes, indices);
    std::vector<Node> nodes;
    build_tree(codes.data(), n, nodes);
    for (int32_t i = 0; i < (int32_t)(2 * n - 1); i++) apply_traverse_order(i, (int32_t)n, triIdOffset, nodes, indices.data(), out);
    compute_boxes(nodes, (int32_t)n, indices.data(), tris, vtx, vtxOffset, out);
    if (out_codes) std::copy(codes.begin(), codes.end(), out_codes);
    if (out_indices) std::copy(indices.begin(), indices.end(), out_indices);
    return true;
}

} } // namespace orc::lbvh
