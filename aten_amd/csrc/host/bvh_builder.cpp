// Host-side BVH construction emitting aten's 48-byte threaded node format.
//
// Output contract (what the traversal kernels and the reference's
// ThreadedBvhTraverser::Traverse, src/libaten/accelerator/threaded_bvh_traverser.h:98-304, rely on):
//   * inner node : f0 = f1 = -1, hit = first child, miss = next subtree (or -1)
//   * BLAS leaf  : f0 (isleaf) = 1, f1 (triid) = global triangle id, f2 (voxeldepth) = -1,
//                  f3 (mtrlid) = -1, hit == miss == next node in walk order (or -1)
//                  (src/libaten/accelerator/sbvh.cpp:880-899)
//   * TLAS leaf  : f0 = instance object id, f1 = -1, f2 = exid bit-field punned to float,
//                  f3 = mesh id, hit == miss == next (src/libaten/accelerator/threaded_bvh.cpp:212-246,266-279)
// Layout is depth-first pre-order, so hit of an inner node is always index + 1 and every link
// points forward: a walk is a monotone sweep through memory.
#include "../../../include/aten_amd_scene.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

namespace {

struct Box {
    float mn[3], mx[3];
    void reset()
    {
        for (int k = 0; k < 3; k++) { mn[k] = std::numeric_limits<float>::max(); mx[k] = -std::numeric_limits<float>::max(); }
    }
    void grow(const Box& b)
    {
        for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], b.mn[k]); mx[k] = std::max(mx[k], b.mx[k]); }
    }
    void grow(const float* p)
    {
        for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], p[k]); mx[k] = std::max(mx[k], p[k]); }
    }
    float half_area() const
    {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Prim {
    Box box;
    float c[3];
    int32_t id;     // payload: triangle id (BLAS) or instance slot (TLAS)
};

struct BuildNode {
    Box box;
    int32_t prim;       // >= 0 : leaf payload
    uint32_t end;       // index one past this node's subtree (pre-order)
};

constexpr int kBins = 32;

class Builder {
public:
    std::vector<BuildNode> nodes;

    void run(std::vector<Prim>& prims)
    {
        nodes.clear();
        nodes.reserve(prims.size() * 2);
        if (!prims.empty()) build(prims, 0, (uint32_t)prims.size());
    }

private:
    void build(std::vector<Prim>& p, uint32_t lo, uint32_t hi)
    {
        const uint32_t self = (uint32_t)nodes.size();
        nodes.push_back(BuildNode{});
        Box bb; bb.reset();
        Box cb; cb.reset();
        for (uint32_t i = lo; i < hi; i++) { bb.grow(p[i].box); cb.grow(p[i].c); }
        nodes[self].box = bb;
        nodes[self].prim = -1;

        if (hi - lo == 1) {
            nodes[self].prim = p[lo].id;
            nodes[self].end = self + 1;
            return;
        }

        uint32_t mid = split(p, lo, hi, cb);
        build(p, lo, mid);
        build(p, mid, hi);
        nodes[self].end = (uint32_t)nodes.size();
    }

    // Binned surface-area-heuristic object split; falls back to a median split on the widest
    // centroid axis when binning cannot separate the primitives.
    uint32_t split(std::vector<Prim>& p, uint32_t lo, uint32_t hi, const Box& cb)
    {
        const uint32_t n = hi - lo;
        int best_axis = -1, best_bin = -1;
        float best_cost = std::numeric_limits<float>::max();

        if (n > 2) {
            for (int axis = 0; axis < 3; axis++) {
                const float ext = cb.mx[axis] - cb.mn[axis];
                if (!(ext > 0.f)) continue;
                const float scale = kBins / ext;
                Box bbox[kBins]; uint32_t cnt[kBins];
                for (int b = 0; b < kBins; b++) { bbox[b].reset(); cnt[b] = 0; }
                for (uint32_t i = lo; i < hi; i++) {
                    int b = (int)((p[i].c[axis] - cb.mn[axis]) * scale);
                    b = std::min(std::max(b, 0), kBins - 1);
                    bbox[b].grow(p[i].box); cnt[b]++;
                }
                float right_area[kBins]; uint32_t right_cnt[kBins];
                Box acc; acc.reset(); uint32_t c = 0;
                for (int b = kBins - 1; b > 0; b--) {
                    acc.grow(bbox[b]); c += cnt[b];
                    right_area[b] = acc.half_area(); right_cnt[b] = c;
                }
                acc.reset(); c = 0;
                for (int b = 0; b < kBins - 1; b++) {
                    acc.grow(bbox[b]); c += cnt[b];
                    if (c == 0 || right_cnt[b + 1] == 0) continue;
                    float cost = acc.half_area() * c + right_area[b + 1] * right_cnt[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
                }
            }
        }

        if (best_axis >= 0) {
            const float ext = cb.mx[best_axis] - cb.mn[best_axis];
            const float scale = kBins / ext;
            const float mn = cb.mn[best_axis];
            auto it = std::stable_partition(p.begin() + lo, p.begin() + hi, [&](const Prim& q) {
                int b = (int)((q.c[best_axis] - mn) * scale);
                b = std::min(std::max(b, 0), kBins - 1);
                return b <= best_bin;
            });
            uint32_t mid = (uint32_t)(it - p.begin());
            if (mid > lo && mid < hi) return mid;
        }

        // median split
        int axis = 0;
        float e0 = cb.mx[0] - cb.mn[0], e1 = cb.mx[1] - cb.mn[1], e2 = cb.mx[2] - cb.mn[2];
        if (e1 > e0 && e1 >= e2) axis = 1; else if (e2 > e0 && e2 > e1) axis = 2;
        uint32_t mid = lo + n / 2;
        std::stable_sort(p.begin() + lo, p.begin() + hi, [axis](const Prim& a, const Prim& b) {
            return a.c[axis] < b.c[axis];
        });
        return mid;
    }
};

atn_bvh_node* emit(const std::vector<BuildNode>& bn)
{
    const uint32_t n = (uint32_t)bn.size();
    atn_bvh_node* out = (atn_bvh_node*)std::malloc(sizeof(atn_bvh_node) * std::max<uint32_t>(n, 1));
    if (!out) return nullptr;
    for (uint32_t i = 0; i < n; i++) {
        atn_bvh_node& o = out[i];
        for (int k = 0; k < 3; k++) { o.boxmin[k] = bn[i].box.mn[k]; o.boxmax[k] = bn[i].box.mx[k]; }
        const float next = (i + 1 < n) ? (float)(i + 1) : -1.0f;
        const float after = (bn[i].end < n) ? (float)bn[i].end : -1.0f;
        if (bn[i].prim >= 0) { o.hit = next; o.miss = next; }
        else { o.hit = next; o.miss = after; }
        o.f0 = o.f1 = o.f2 = o.f3 = -1.0f;
    }
    return out;
}

} // namespace

extern "C" {

int atns_build_blas(const atn_vec4* vtx_pos, const atn_triangle_param* tris,
                    const uint32_t* tri_ids, uint32_t n_tris,
                    atn_bvh_node** out_nodes, uint32_t* out_count,
                    float out_bbox_min[3], float out_bbox_max[3])
{
    if (!vtx_pos || !tris || !tri_ids || !out_nodes || !out_count || n_tris == 0) return -1;
    std::vector<Prim> prims(n_tris);
    for (uint32_t i = 0; i < n_tris; i++) {
        if (tri_ids[i] >= (1u << 24)) return -2;   // ids are stored as float: exact below 2^24
        const atn_triangle_param& t = tris[tri_ids[i]];
        Prim& p = prims[i];
        p.box.reset();
        for (int k = 0; k < 3; k++) {
            const atn_vec4& v = vtx_pos[t.idx[k]];
            const float q[3] = { v.x, v.y, v.z };
            p.box.grow(q);
        }
        for (int k = 0; k < 3; k++) p.c[k] = 0.5f * (p.box.mn[k] + p.box.mx[k]);
        p.id = (int32_t)tri_ids[i];
    }
    Builder b;
    b.run(prims);
    atn_bvh_node* nodes = emit(b.nodes);
    if (!nodes) return -3;
    for (size_t i = 0; i < b.nodes.size(); i++) {
        if (b.nodes[i].prim >= 0) {
            nodes[i].f0 = 1.0f;                         // isleaf
            nodes[i].f1 = (float)b.nodes[i].prim;       // triid
            nodes[i].f2 = -1.0f;                        // AT_DISABLE_VOXEL
            nodes[i].f3 = -1.0f;
        }
    }
    *out_nodes = nodes;
    *out_count = (uint32_t)b.nodes.size();
    if (out_bbox_min && out_bbox_max) {
        for (int k = 0; k < 3; k++) { out_bbox_min[k] = b.nodes[0].box.mn[k]; out_bbox_max[k] = b.nodes[0].box.mx[k]; }
    }
    return 0;
}

int atns_build_tlas(const float* boxes, const int32_t* object_ids, const int32_t* blas_list_ids,
                    const int32_t* mesh_ids, uint32_t n,
                    atn_bvh_node** out_nodes, uint32_t* out_count)
{
    if (!boxes || !object_ids || !blas_list_ids || !out_nodes || !out_count || n == 0) return -1;
    std::vector<Prim> prims(n);
    for (uint32_t i = 0; i < n; i++) {
        Prim& p = prims[i];
        for (int k = 0; k < 3; k++) { p.box.mn[k] = boxes[6 * i + k]; p.box.mx[k] = boxes[6 * i + 3 + k]; }
        for (int k = 0; k < 3; k++) p.c[k] = 0.5f * (p.box.mn[k] + p.box.mx[k]);
        p.id = (int32_t)i;
    }
    Builder b;
    b.run(prims);
    atn_bvh_node* nodes = emit(b.nodes);
    if (!nodes) return -3;
    for (size_t i = 0; i < b.nodes.size(); i++) {
        if (b.nodes[i].prim >= 0) {
            const int32_t slot = b.nodes[i].prim;
            nodes[i].f0 = (float)object_ids[slot];
            nodes[i].f1 = -1.0f;
            const int32_t exid = blas_list_ids[slot];
            if (exid >= 0) {
                // ThreadedBvhNode::ConstructExternalBvhIdxFlag(exid, -1), threaded_bvh.h:46-54
                uint32_t bits = (uint32_t)exid & 0x7fffu;   // lodExid = 0, hasLod = 0, noExternal = 0
                float f; std::memcpy(&f, &bits, 4);
                nodes[i].f2 = f;
            }
            else {
                nodes[i].f2 = -1.0f;
            }
            nodes[i].f3 = mesh_ids ? (float)mesh_ids[slot] : -1.0f;
        }
    }
    *out_nodes = nodes;
    *out_count = (uint32_t)b.nodes.size();
    return 0;
}

void atns_free(void* p) { std::free(p); }

int64_t atns_validate_nodes(const atn_bvh_node* nodes, uint32_t count)
{
    if (!nodes) return -1;
    int64_t leaves = 0;
    for (uint32_t i = 0; i < count; i++) {
        const int32_t h = (int32_t)nodes[i].hit, m = (int32_t)nodes[i].miss;
        if (h < -1 || h >= (int32_t)count || m < -1 || m >= (int32_t)count) return -2;
    }
    // hit-only walk must visit every node exactly once
    int32_t id = 0; uint32_t steps = 0;
    while (id >= 0) {
        if (++steps > count) return -3;
        if (nodes[id].f0 >= 0 || nodes[id].f1 >= 0) leaves++;
        id = (int32_t)nodes[id].hit;
    }
    return leaves;
}

} // extern "C"
