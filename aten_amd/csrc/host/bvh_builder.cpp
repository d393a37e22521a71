// Host-side BVH construction emitting aten's 48-byte threaded node format.
//
// What it replaces: aten::sbvh::onBuild + sbvh::convert (src/libaten/accelerator/sbvh.cpp:190-421, 827-950) for the
// bottom level -- a split BVH (Stich et al. 2009): per node the best OBJECT split (surface-area heuristic over the
// references' centroids, findObjectSplit sbvh.cpp:430-547) is compared with the best SPATIAL split (references chopped
// at bin planes, findSpatialSplit sbvh.cpp:548-680; straddling references duplicated or "unsplit" to one side,
// spatialSort sbvh.cpp:682-779), and the spatial one is only looked at when the object split's children overlap by more
// than `spatial_alpha` of the root's area (sbvh.cpp:275-288) -- and aten::ThreadedBVH::build (threaded_bvh.cpp:178-357)
// for the top level (object splits only).
//
// Where it differs from the reference's builder, on purpose (the tree is an INPUT of the walk, on the CPU as on the GPU:
// closest hits do not depend on it, the number of node visits does -- tools/tree_quality.py measures that):
//   * a straddling reference is clipped as a TRIANGLE at the plane (the reference chops the reference's box only,
//     sbvh.cpp:603-618, 752-770): tighter child boxes;
//   * object splits are an exact sweep over the sorted centroids for nodes below `sweep_below` references and binned
//     above (the reference: 16 bins everywhere, sbvh.h:393);
//   * nodes are laid out in depth-first pre-order (the reference: creation order, links through getOrderIndex), so the
//     hit link of an inner node is index + 1 and every link points forward: a walk is a monotone sweep through memory;
//   * the child that the fixed-order walk enters first is CHOSEN (`child_order`), not "the lower side of the plane";
//   * after the build the tree is re-threaded bottom-up with the boxes of what the subtrees really hold, never larger
//     than what the split assigned (the reference keeps the split's boxes, sbvh.cpp:392-393).
//
// Output contract (what the traversal kernels and the reference's
// ThreadedBvhTraverser::Traverse, src/libaten/accelerator/threaded_bvh_traverser.h:98-304, rely on):
//   * inner node : f0 = f1 = -1, hit = first child, miss = next subtree (or -1)
//   * BLAS leaf  : f0 (isleaf) = 1, f1 (triid) = global triangle id, f2 (voxeldepth) = -1,
//                  f3 (mtrlid) = -1, hit == miss == next node in walk order (or -1)
//                  (src/libaten/accelerator/sbvh.cpp:880-899)
//   * TLAS leaf  : f0 = instance object id, f1 = -1, f2 = exid bit-field punned to float,
//                  f3 = mesh id, hit == miss == next (src/libaten/accelerator/threaded_bvh.cpp:212-246,266-279)
#include "../../../include/aten_amd_scene.h"
#include "anyhit_twin.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <vector>

namespace {

constexpr float kInf = std::numeric_limits<float>::max();

struct Box {
    float mn[3], mx[3];
    void reset()
    {
        for (int k = 0; k < 3; k++) { mn[k] = kInf; mx[k] = -kInf; }
    }
    void grow(const Box& b)
    {
        for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], b.mn[k]); mx[k] = std::max(mx[k], b.mx[k]); }
    }
    void grow(const float* p)
    {
        for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], p[k]); mx[k] = std::max(mx[k], p[k]); }
    }
    void clip(const Box& b)
    {
        for (int k = 0; k < 3; k++) { mn[k] = std::max(mn[k], b.mn[k]); mx[k] = std::min(mx[k], b.mx[k]); }
    }
    bool valid() const { return mn[0] <= mx[0] && mn[1] <= mx[1] && mn[2] <= mx[2]; }
    float half_area() const
    {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.f;
        return dx * dy + dy * dz + dz * dx;
    }
    float centre(int k) const { return 0.5f * (mn[k] + mx[k]); }
};

inline Box empty_box() { Box b; b.reset(); return b; }

inline float overlap_half_area(const Box& a, const Box& b)
{
    Box d;
    for (int k = 0; k < 3; k++) { d.mn[k] = std::max(a.mn[k], b.mn[k]); d.mx[k] = std::min(a.mx[k], b.mx[k]); }
    return d.valid() ? d.half_area() : 0.f;
}

// A reference: (part of) one primitive.  `box` is the bound of the part that lies in the node that owns the reference.
struct Ref {
    Box box;
    uint32_t prim;
};

struct Tri { float v[3][3]; };

struct BuildNode {
    Box box;
    int32_t prim;       // >= 0 : leaf payload (index into the caller's primitive table)
    uint32_t end;       // index one past this node's subtree (pre-order)
};

struct Options {
    bool spatial = true;
    float alpha = 1e-5f;            // sbvh.cpp:232 areaAlpha
    int object_bins = 32;
    int spatial_bins = 64;
    uint32_t sweep_below = 4096;    // exact sweep for nodes with fewer references than this
    int child_order = ATNS_ORDER_NEAR_POINT;
    float max_refs_factor = 4.0f;   // duplication budget: references <= factor * primitives
    float order_point[3] = { 0, 0, 0 };
    bool order_point_given = false; // else: the area-weighted centroid of the triangles
    int reinsert_iterations = 100;  // rounds of the insertion-based optimisation (0 = off), each over the worst `reinsert_batch` of the inner nodes
    float reinsert_batch = 0.1f;
};

class Builder {
public:
    std::vector<BuildNode> nodes;
    uint64_t n_spatial = 0, n_refs_out = 0;

    Builder(const Options& o, const Tri* tris) : opt(o), tris_(tris) {}

    void run(std::vector<Ref>& refs)
    {
        nodes.clear();
        nodes.reserve(refs.size() * 2);
        if (refs.empty()) return;
        Box bb = empty_box();
        for (const Ref& r : refs) bb.grow(r.box);
        root_area_ = std::max(bb.half_area(), 1e-30f);
        ref_budget_ = (uint64_t)((double)opt.max_refs_factor * (double)refs.size());
        n_refs_live_ = refs.size();
        build(std::move(refs), bb);
        tighten();
    }

private:
    Options opt;
    const Tri* tris_;
    float root_area_ = 1.f;
    uint64_t ref_budget_ = 0, n_refs_live_ = 0;

    struct ObjectSplit { float cost = kInf; int axis = -1; uint32_t at = 0; int bin = -1; Box lb, rb; };
    struct SpatialSplit { float cost = kInf; int axis = -1; float plane = 0; Box lb, rb; uint32_t ln = 0, rn = 0; };

    // ---- object split ---------------------------------------------------------------------------------------------
    // exact: for every axis, every position in the centroid order
    void object_split_sweep(std::vector<Ref>& refs, ObjectSplit& best, std::vector<float>& right_area)
    {
        const uint32_t n = (uint32_t)refs.size();
        right_area.resize(n);
        std::vector<uint32_t>& ord = order_;
        for (int axis = 0; axis < 3; axis++) {
            ord.resize(n);
            for (uint32_t i = 0; i < n; i++) ord[i] = i;
            std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) {
                const float ca = refs[a].box.centre(axis), cb = refs[b].box.centre(axis);
                return ca < cb || (ca == cb && a < b);
            });
            Box acc = empty_box();
            for (uint32_t i = n - 1; i > 0; i--) { acc.grow(refs[ord[i]].box); right_area[i] = acc.half_area(); }
            acc.reset();
            for (uint32_t i = 1; i < n; i++) {
                acc.grow(refs[ord[i - 1]].box);
                const float cost = acc.half_area() * (float)i + right_area[i] * (float)(n - i);
                if (cost < best.cost) { best.cost = cost; best.axis = axis; best.at = i; best.bin = -1; }
            }
        }
        if (best.axis < 0) return;
        const int axis = best.axis;
        std::stable_sort(refs.begin(), refs.end(), [axis](const Ref& a, const Ref& b) { return a.box.centre(axis) < b.box.centre(axis); });
        best.lb.reset(); best.rb.reset();
        for (uint32_t i = 0; i < n; i++) (i < best.at ? best.lb : best.rb).grow(refs[i].box);
    }

    // binned (findObjectSplit's scheme, sbvh.cpp:430-547, with `object_bins` bins)
    void object_split_binned(std::vector<Ref>& refs, ObjectSplit& best)
    {
        const uint32_t n = (uint32_t)refs.size();
        const int nb = opt.object_bins;
        Box cb = empty_box();
        for (const Ref& r : refs) { const float c[3] = { r.box.centre(0), r.box.centre(1), r.box.centre(2) }; cb.grow(c); }
        std::vector<Box> bbox(nb), racc(nb);
        std::vector<uint32_t> cnt(nb), rcnt(nb);
        for (int axis = 0; axis < 3; axis++) {
            const float ext = cb.mx[axis] - cb.mn[axis];
            if (!(ext > 0.f)) continue;
            const float scale = (float)nb / ext;
            for (int b = 0; b < nb; b++) { bbox[b].reset(); cnt[b] = 0; }
            for (const Ref& r : refs) {
                int b = (int)((r.box.centre(axis) - cb.mn[axis]) * scale);
                b = std::min(std::max(b, 0), nb - 1);
                bbox[b].grow(r.box); cnt[b]++;
            }
            Box acc = empty_box(); uint32_t c = 0;
            for (int b = nb - 1; b > 0; b--) { acc.grow(bbox[b]); c += cnt[b]; racc[b] = acc; rcnt[b] = c; }
            acc.reset(); c = 0;
            for (int b = 0; b < nb - 1; b++) {
                acc.grow(bbox[b]); c += cnt[b];
                if (c == 0 || rcnt[b + 1] == 0) continue;
                const float cost = acc.half_area() * (float)c + racc[b + 1].half_area() * (float)rcnt[b + 1];
                if (cost < best.cost) { best.cost = cost; best.axis = axis; best.bin = b; best.at = c; best.lb = acc; best.rb = racc[b + 1]; }
            }
        }
        if (best.axis < 0) return;
        const int axis = best.axis;
        const float scale = (float)nb / (cb.mx[axis] - cb.mn[axis]), mn = cb.mn[axis];
        const int bin = best.bin;
        std::stable_partition(refs.begin(), refs.end(), [&](const Ref& r) {
            int b = (int)((r.box.centre(axis) - mn) * scale);
            b = std::min(std::max(b, 0), nb - 1);
            return b <= bin;
        });
        (void)n;
    }

    // the fallback of sbvh.cpp:318-355: halves of the centroid order along the widest axis
    void median_split(std::vector<Ref>& refs, const Box& bb, ObjectSplit& best)
    {
        int axis = 0;
        const float e0 = bb.mx[0] - bb.mn[0], e1 = bb.mx[1] - bb.mn[1], e2 = bb.mx[2] - bb.mn[2];
        if (e1 > e0 && e1 >= e2) axis = 1; else if (e2 > e0 && e2 > e1) axis = 2;
        std::stable_sort(refs.begin(), refs.end(), [axis](const Ref& a, const Ref& b) { return a.box.centre(axis) < b.box.centre(axis); });
        best.axis = axis; best.at = (uint32_t)refs.size() / 2; best.bin = -1;
        best.lb.reset(); best.rb.reset();
        for (uint32_t i = 0; i < refs.size(); i++) (i < best.at ? best.lb : best.rb).grow(refs[i].box);
        best.cost = best.lb.half_area() * (float)best.at + best.rb.half_area() * (float)(refs.size() - best.at);
    }

    // ---- spatial split ----------------------------------------------------------------------------------------------
    // The two parts of a reference either side of the plane x[axis] = pos: the triangle's vertices go to their side, the
    // points where its edges cross the plane to both; each part is then cut back to the reference's own box.
    void split_ref(const Ref& r, int axis, float pos, Box& lo, Box& hi) const
    {
        lo.reset(); hi.reset();
        const Tri& t = tris_[r.prim];
        for (int e = 0; e < 3; e++) {
            const float* a = t.v[e];
            const float* b = t.v[(e + 1) % 3];
            const float pa = a[axis], pb = b[axis];
            if (pa <= pos) lo.grow(a);
            if (pa >= pos) hi.grow(a);
            if ((pa < pos && pb > pos) || (pa > pos && pb < pos)) {
                const float s = std::min(std::max((pos - pa) / (pb - pa), 0.f), 1.f);
                // the crossing point, widened by a few ulps off the split axis: the rounded point may lie a hair inside
                // the true edge, and a part's box must never lose a sliver of the triangle
                float p[3], q[3];
                for (int k = 0; k < 3; k++) {
                    const float x = a[k] + s * (b[k] - a[k]);
                    const float pad = (k == axis) ? 0.f : 4.8e-7f * std::max(std::fabs(a[k]), std::fabs(b[k]));
                    p[k] = x - pad; q[k] = x + pad;
                }
                p[axis] = q[axis] = pos;
                lo.grow(p); lo.grow(q); hi.grow(p); hi.grow(q);
            }
        }
        lo.mx[axis] = std::min(lo.mx[axis], pos);
        hi.mn[axis] = std::max(hi.mn[axis], pos);
        lo.clip(r.box); hi.clip(r.box);
    }

    void spatial_split_find(const std::vector<Ref>& refs, const Box& bb, SpatialSplit& best)
    {
        const int nb = opt.spatial_bins;
        std::vector<Box> bbox(nb), racc(nb);
        std::vector<uint32_t> enter(nb), leave(nb);
        for (int axis = 0; axis < 3; axis++) {
            const float ext = bb.mx[axis] - bb.mn[axis];
            if (!(ext > 0.f)) continue;
            const float scale = (float)nb / ext, width = ext / (float)nb, org = bb.mn[axis];
            for (int b = 0; b < nb; b++) { bbox[b].reset(); enter[b] = leave[b] = 0; }
            for (const Ref& r : refs) {
                int b0 = std::min(std::max((int)((r.box.mn[axis] - org) * scale), 0), nb - 1);
                int b1 = std::min(std::max((int)((r.box.mx[axis] - org) * scale), b0), nb - 1);
                enter[b0]++; leave[b1]++;
                if (b0 == b1) { bbox[b0].grow(r.box); continue; }
                Ref cur = r;
                for (int b = b0; b < b1; b++) {
                    Box lo, hi;
                    split_ref(cur, axis, org + width * (float)(b + 1), lo, hi);
                    if (lo.valid()) bbox[b].grow(lo);
                    cur.box = hi;
                    if (!hi.valid()) break;
                }
                if (cur.box.valid()) bbox[b1].grow(cur.box);
            }
            Box acc = empty_box();
            for (int b = nb - 1; b > 0; b--) { acc.grow(bbox[b]); racc[b] = acc; }
            acc.reset();
            uint32_t ln = 0, rn = (uint32_t)refs.size();
            for (int b = 0; b < nb - 1; b++) {
                acc.grow(bbox[b]); ln += enter[b]; rn -= leave[b];
                if (ln == 0 || rn == 0) continue;
                const float cost = acc.half_area() * (float)ln + racc[b + 1].half_area() * (float)rn;
                if (cost < best.cost) {
                    best.cost = cost; best.axis = axis; best.plane = org + width * (float)(b + 1);
                    best.lb = acc; best.rb = racc[b + 1]; best.ln = ln; best.rn = rn;
                }
            }
        }
    }

    // spatialSort, sbvh.cpp:682-779: references on one side go there; a straddling one goes to ONE side if that is
    // cheaper than duplicating it (the paper's "reference unsplitting"), else both sides get their clipped part.
    bool spatial_split_apply(const std::vector<Ref>& refs, SpatialSplit sp, std::vector<Ref>& left, std::vector<Ref>& right, Box& lb, Box& rb)
    {
        const int axis = sp.axis;
        left.clear(); right.clear();
        lb = sp.lb; rb = sp.rb;
        float la = lb.half_area(), ra = rb.half_area();
        float ln = (float)sp.ln, rn = (float)sp.rn;
        for (const Ref& r : refs) {
            if (r.box.mx[axis] <= sp.plane) { left.push_back(r); continue; }
            if (r.box.mn[axis] >= sp.plane) { right.push_back(r); continue; }
            Box lu = lb; lu.grow(r.box);
            Box ru = rb; ru.grow(r.box);
            const float c_split = la * ln + ra * rn;
            const float c_left = lu.half_area() * ln + ra * (rn - 1.f);
            const float c_right = la * (ln - 1.f) + ru.half_area() * rn;
            if (c_left < c_split && c_left <= c_right) {
                left.push_back(r); lb = lu; la = lu.half_area(); rn -= 1.f;
            }
            else if (c_right < c_split) {
                right.push_back(r); rb = ru; ra = ru.half_area(); ln -= 1.f;
            }
            else {
                Box lo, hi;
                split_ref(r, axis, sp.plane, lo, hi);
                if (lo.valid()) left.push_back(Ref{ lo, r.prim });
                if (hi.valid()) right.push_back(Ref{ hi, r.prim });
                if (!lo.valid() && !hi.valid()) left.push_back(r);
            }
        }
        // both sides must make progress, or the recursion need not end
        return !left.empty() && !right.empty() && left.size() < refs.size() && right.size() < refs.size();
    }

    void build(std::vector<Ref> refs, const Box& bb)
    {
        const uint32_t self = (uint32_t)nodes.size();
        nodes.push_back(BuildNode{});
        nodes[self].box = bb;
        nodes[self].prim = -1;
        const uint32_t n = (uint32_t)refs.size();
        if (n == 1) {
            nodes[self].prim = (int32_t)refs[0].prim;
            nodes[self].box = refs[0].box;
            nodes[self].end = self + 1;
            n_refs_out++;
            return;
        }

        ObjectSplit ob;
        if (n > 2) {
            if (n < opt.sweep_below) object_split_sweep(refs, ob, scratch_area_);
            else object_split_binned(refs, ob);
        }
        if (ob.axis < 0) median_split(refs, bb, ob);

        std::vector<Ref> left, right;
        Box lb, rb;
        bool done = false;
        if (opt.spatial && tris_ && n_refs_live_ < ref_budget_ && overlap_half_area(ob.lb, ob.rb) / root_area_ >= opt.alpha) {
            SpatialSplit sp;
            spatial_split_find(refs, bb, sp);
            if (sp.axis >= 0 && sp.cost < ob.cost) {
                if (spatial_split_apply(refs, sp, left, right, lb, rb) && n_refs_live_ + left.size() + right.size() - n <= ref_budget_) {
                    done = true;
                    n_spatial++;
                    n_refs_live_ += left.size() + right.size() - n;
                }
            }
        }
        if (!done) {
            left.assign(refs.begin(), refs.begin() + ob.at);
            right.assign(refs.begin() + ob.at, refs.end());
            lb = ob.lb; rb = ob.rb;
        }
        std::vector<Ref>().swap(refs);

        build(std::move(left), lb);
        build(std::move(right), rb);
        nodes[self].end = (uint32_t)nodes.size();
    }

    // Boxes of inner nodes = union of their children's, bottom-up.  Never larger than what the split assigned (a child
    // box is the union of its references, each inside the split's box), sometimes smaller after unsplitting / clipping.
    void tighten()
    {
        for (uint32_t i = (uint32_t)nodes.size(); i-- > 0;) {
            if (nodes[i].prim >= 0) continue;
            const uint32_t a = i + 1, b = nodes[a].end;
            Box u = nodes[a].box; u.grow(nodes[b].box);
            nodes[i].box = u;
        }
    }

    std::vector<uint32_t> order_;
    std::vector<float> scratch_area_;
};



// ---- post passes on the finished tree -----------------------------------------------------------------------------------
// (1) Insertion-based optimisation (Bittner, Hapala, Havran 2013): the top-down build is greedy -- a split is chosen by the
//     areas of ITS two children and never revisited.  Afterwards, nodes whose box is large for what their children need
//     (inefficiency = area^3 / (mean child area * smallest child area)) are taken out of the tree, two subtrees at a time, and each
//     subtree is re-inserted where it adds the least area: a branch-and-bound search over "become the sibling of X" for every X,
//     cost = area(X u subtree) + the growth of X's ancestors.  The sum of node areas / root area -- the expected number of box
//     tests of a random ray, `sah_cost` -- only goes down.  Works on split-BVH trees as on any other: references are leaves.
// (2) Which child the fixed-order walk enters first (`child_order`), decided from the final boxes, and the nodes laid out in
//     depth-first pre-order of that choice.
class Restructure {
public:
    Restructure(const std::vector<BuildNode>& pre, const Options& o) : opt(o)
    {
        const uint32_t n = (uint32_t)pre.size();
        t.resize(n);
        for (uint32_t i = 0; i < n; i++) { t[i].box = pre[i].box; t[i].prim = pre[i].prim; t[i].parent = -1; t[i].left = t[i].right = -1; t[i].leaves = 1; }
        for (uint32_t i = 0; i < n; i++) {
            if (pre[i].prim >= 0) continue;
            const uint32_t a = i + 1, b = pre[a].end;
            t[i].left = (int32_t)a; t[i].right = (int32_t)b; t[a].parent = (int32_t)i; t[b].parent = (int32_t)i;
        }
        root = 0;
    }

    double area_sum() const
    {
        double s = 0;
        for (const X& x : t) s += (double)x.box.half_area();
        return s;
    }

    // returns the number of re-insertions that were kept
    uint64_t optimise(int iterations, float batch_fraction)
    {
        const uint32_t n = (uint32_t)t.size();
        if (n < 16 || iterations <= 0) return 0;
        uint64_t moved = 0;
        std::vector<std::pair<float, int32_t>> cand;
        uint64_t rng = 0x9e3779b97f4a7c15ull;
        double last = area_sum();
        for (int it = 0; it < iterations; it++) {
            cand.clear();
            const bool random_round = (it % 4) == 3;       // every fourth round: a random sample (the metric alone keeps picking the same nodes)
            for (uint32_t i = 0; i < n; i++) {
                const X& x = t[i];
                if (x.left < 0 || (int32_t)i == root || x.parent == root) continue;
                float key;
                if (random_round) { rng = rng * 6364136223846793005ull + 1442695040888963407ull; key = (float)(rng >> 40); }
                else {
                    const float a = x.box.half_area(), al = t[x.left].box.half_area(), ar = t[x.right].box.half_area();
                    key = a * (a / std::max(0.5f * (al + ar), 1e-30f)) * (a / std::max(std::min(al, ar), 1e-30f));
                }
                cand.emplace_back(key, (int32_t)i);
            }
            size_t k = std::max<size_t>(1, (size_t)((double)cand.size() * batch_fraction));
            if (k > cand.size()) k = cand.size();
            std::partial_sort(cand.begin(), cand.begin() + k, cand.end(), [](const std::pair<float, int32_t>& a, const std::pair<float, int32_t>& b) {
                return a.first > b.first || (a.first == b.first && a.second < b.second);
            });
            for (size_t c = 0; c < k; c++) moved += reinsert_children_of(cand[c].second);
            // stop when a whole cycle (three rounds by the metric, one random) gained less than 0.02 %
            if ((it % 4) == 3) {
                const double now = area_sum();
                if (last - now < 2e-4 * last) break;
                last = now;
            }
        }
        return moved;
    }

    // pre-order emission with the child-order rule
    std::vector<BuildNode> emit()
    {
        count_leaves();
        std::vector<BuildNode> out;
        out.reserve(t.size());
        std::vector<int32_t> stack;
        std::vector<uint32_t> new_of(t.size(), 0), size_of(t.size(), 1);
        // subtree sizes, bottom-up by an explicit post-order
        {
            std::vector<int32_t> order; order.reserve(t.size());
            stack.push_back(root);
            while (!stack.empty()) { const int32_t i = stack.back(); stack.pop_back(); order.push_back(i); if (t[i].left >= 0) { stack.push_back(t[i].left); stack.push_back(t[i].right); } }
            for (size_t k = order.size(); k-- > 0;) { const int32_t i = order[k]; if (t[i].left >= 0) size_of[i] = 1 + size_of[t[i].left] + size_of[t[i].right]; }
        }
        stack.push_back(root);
        while (!stack.empty()) {
            const int32_t i = stack.back(); stack.pop_back();
            BuildNode bn; bn.box = t[i].box; bn.prim = t[i].prim; bn.end = (uint32_t)out.size() + size_of[i];
            out.push_back(bn);
            if (t[i].left >= 0) {
                int32_t a = t[i].left, b = t[i].right;
                if (right_first(t[a].box, t[a].leaves, t[b].box, t[b].leaves)) std::swap(a, b);
                stack.push_back(b); stack.push_back(a);
            }
        }
        return out;
    }

private:
    struct X { Box box; int32_t parent, left, right, prim; uint32_t leaves; };
    struct Item { float induced; int32_t node; };
    mutable std::vector<Item> heap_;
    std::vector<X> t;
    int32_t root = 0;
    Options opt;

    void count_leaves()
    {
        std::vector<int32_t> order, stack;
        stack.push_back(root);
        while (!stack.empty()) { const int32_t i = stack.back(); stack.pop_back(); order.push_back(i); if (t[i].left >= 0) { stack.push_back(t[i].left); stack.push_back(t[i].right); } }
        for (size_t k = order.size(); k-- > 0;) { const int32_t i = order[k]; t[i].leaves = t[i].left >= 0 ? t[t[i].left].leaves + t[t[i].right].leaves : 1u; }
    }

    bool right_first(const Box& lb, uint32_t ln, const Box& rb, uint32_t rn) const
    {
        switch (opt.child_order) {
        case ATNS_ORDER_AREA: return rb.half_area() > lb.half_area();
        case ATNS_ORDER_AREA_SMALL: return rb.half_area() < lb.half_area();
        case ATNS_ORDER_COUNT: return rn > ln;
        case ATNS_ORDER_COUNT_SMALL: return rn < ln;
        case ATNS_ORDER_NEAR_POINT: {
            float dl = 0, dr = 0;
            for (int k = 0; k < 3; k++) {
                const float p = opt.order_point[k];
                const float a = std::max(std::max(lb.mn[k] - p, p - lb.mx[k]), 0.f);
                const float b = std::max(std::max(rb.mn[k] - p, p - rb.mx[k]), 0.f);
                dl += a * a; dr += b * b;
            }
            if (dl != dr) return dr < dl;
            float cl = 0, cr = 0;
            for (int k = 0; k < 3; k++) {
                const float a = lb.centre(k) - opt.order_point[k], b = rb.centre(k) - opt.order_point[k];
                cl += a * a; cr += b * b;
            }
            return cr < cl;
        }
        default: return false;      // ATNS_ORDER_AS_SPLIT: as built
        }
    }

    double delta_ = 0;      // area added to the tree by the operation in progress (sum over the boxes it changed)
    void refit_from(int32_t i)
    {
        while (i >= 0) {
            Box u = t[t[i].left].box; u.grow(t[t[i].right].box);
            const float before = t[i].box.half_area(), after = u.half_area();
            const bool same = std::memcmp(&u, &t[i].box, sizeof(Box)) == 0;
            t[i].box = u;
            if (same) break;        // nothing above changes either
            delta_ += (double)after - (double)before;
            i = t[i].parent;
        }
    }

    // the node X next to which the detached subtree `sub` costs the least: area(X u sub) + growth of X's ancestors
    int32_t best_sibling(const Box& sub) const
    {
        auto worse = [](const Item& a, const Item& b) { return a.induced > b.induced; };
        std::vector<Item>& heap = heap_;
        heap.clear();
        heap.push_back(Item{ 0.f, root });
        const float sub_area = sub.half_area();
        float best = kInf; int32_t best_node = root;
        while (!heap.empty()) {
            std::pop_heap(heap.begin(), heap.end(), worse);
            const Item it = heap.back(); heap.pop_back();
            if (it.induced + sub_area >= best) break;           // every remaining candidate costs at least this much
            const X& x = t[it.node];
            Box u = x.box; u.grow(sub);
            const float direct = u.half_area();
            const float total = it.induced + direct;
            if (total < best) { best = total; best_node = it.node; }
            const float below = total - x.box.half_area();      // what the children inherit: this node grows to u
            if (x.left >= 0 && below + sub_area < best) {
                heap.push_back(Item{ below, x.left }); std::push_heap(heap.begin(), heap.end(), worse);
                heap.push_back(Item{ below, x.right }); std::push_heap(heap.begin(), heap.end(), worse);
            }
        }
        return best_node;
    }

    // `sub` (detached, parent = -1) becomes the sibling of x under the free node `fresh`
    void attach(int32_t sub, int32_t x, int32_t fresh)
    {
        const int32_t p = t[x].parent;
        t[fresh].left = x; t[fresh].right = sub; t[fresh].parent = p; t[fresh].prim = -1;
        if (p >= 0) { if (t[p].left == x) t[p].left = fresh; else t[p].right = fresh; }
        else root = fresh;
        t[x].parent = fresh; t[sub].parent = fresh;
        // the fresh node's box is whatever it held in its previous place: set it here (an "unchanged" box must not stop the
        // refit below it -- its NEW parent has not seen `sub` yet), then refit upwards
        Box u = t[x].box; u.grow(t[sub].box);
        delta_ += (double)u.half_area() - (double)t[fresh].box.half_area();
        t[fresh].box = u;
        refit_from(p);
    }

    // the inverse of attach: `sub`'s sibling takes the place of their parent, which is free again
    void detach(int32_t sub)
    {
        const int32_t f = t[sub].parent;
        const int32_t x = t[f].left == sub ? t[f].right : t[f].left;
        const int32_t gp = t[f].parent;
        if (gp >= 0) { if (t[gp].left == f) t[gp].left = x; else t[gp].right = x; }
        else root = x;
        t[x].parent = gp;
        t[sub].parent = -1;
        refit_from(gp);
    }

    // take node n and its parent out of the tree, re-insert n's two children one by one where each adds the least area;
    // 1 if the tree got cheaper -- else the operation is undone (two greedy insertions need not beat what was there)
    uint64_t reinsert_children_of(int32_t n)
    {
        if (t[n].left < 0 || n == root) return 0;
        const int32_t p = t[n].parent;
        if (p < 0 || p == root) return 0;
        const int32_t g = t[p].parent;
        const int32_t s = t[p].left == n ? t[p].right : t[p].left;
        const int32_t l0 = t[n].left, r0 = t[n].right;
        const bool n_was_left = t[p].left == n, p_was_left = t[g].left == p;
        const Box box_n = t[n].box, box_p = t[p].box;
        delta_ = 0;
        // detach: the sibling takes the parent's place
        if (p_was_left) t[g].left = s; else t[g].right = s;
        t[s].parent = g;
        refit_from(g);
        t[l0].parent = -1; t[r0].parent = -1;
        int32_t l = l0, r = r0;
        if (t[r].box.half_area() > t[l].box.half_area()) std::swap(l, r);      // the larger one first
        attach(l, best_sibling(t[l].box), p);
        attach(r, best_sibling(t[r].box), n);
        if (delta_ < 0) return 1;
        // undo: both out again (the tree is then exactly what it was after the first detach), then the old arrangement
        detach(r); detach(l);
        t[n].left = l0; t[n].right = r0; t[n].parent = p; t[l0].parent = n; t[r0].parent = n; t[n].box = box_n;
        t[p].left = n_was_left ? n : s; t[p].right = n_was_left ? s : n; t[p].parent = g; t[p].box = box_p;
        const int32_t gs = t[s].parent;     // == g
        if (t[gs].left == s) t[gs].left = p; else t[gs].right = p;
        t[s].parent = p;
        refit_from(g);
        return 0;
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

Options options_from(const atns_bvh_options* o)
{
    Options r;
    if (!o) return r;
    r.spatial = o->spatial_splits != 0;
    if (o->spatial_alpha >= 0.f) r.alpha = o->spatial_alpha;
    if (o->object_bins >= 2) r.object_bins = std::min(o->object_bins, 1024);
    if (o->spatial_bins >= 2) r.spatial_bins = std::min(o->spatial_bins, 1024);
    if (o->sweep_below >= 0) r.sweep_below = (uint32_t)o->sweep_below;
    if (o->child_order >= 0 && o->child_order <= ATNS_ORDER_NEAR_POINT) r.child_order = o->child_order;
    r.order_point_given = o->order_point_given != 0;
    if (o->reinsert_iterations >= 0) r.reinsert_iterations = std::min(o->reinsert_iterations, 1000);
    if (o->reinsert_batch > 0.f && o->reinsert_batch <= 1.f) r.reinsert_batch = o->reinsert_batch;
    if (o->max_refs_factor >= 1.f) r.max_refs_factor = o->max_refs_factor;
    for (int k = 0; k < 3; k++) r.order_point[k] = o->order_point[k];
    return r;
}

// A threaded list somebody else built (an imported .sbvh): the same boxes and leaves, re-arranged by the two post passes.
int optimise_nodes(const atn_bvh_node* in, uint32_t count, const atns_bvh_options* user, atn_bvh_node** out_nodes, uint32_t* out_count,
                   atns_bvh_stats* stats)
{
    if (!in || !out_nodes || !out_count || count == 0) return -1;
    // the threaded list as a pre-order array: follow the hit links (every node once), a node's subtree ends where its miss link points
    // -- or, for a leaf, at the next node
    std::vector<uint32_t> order; order.reserve(count);
    std::vector<int32_t> pos_of(count, -1);
    {
        int32_t id = 0;
        while (id >= 0) {
            if ((uint32_t)id >= count || pos_of[id] >= 0 || order.size() >= count) return -4;      // not a threaded tree
            pos_of[id] = (int32_t)order.size();
            order.push_back((uint32_t)id);
            id = (int32_t)in[id].hit;
        }
        if (order.size() != count) return -4;
    }
    std::vector<BuildNode> pre(count);
    for (uint32_t k = 0; k < count; k++) {
        const atn_bvh_node& n = in[order[k]];
        for (int a = 0; a < 3; a++) { pre[k].box.mn[a] = n.boxmin[a]; pre[k].box.mx[a] = n.boxmax[a]; }
        const bool leaf = n.f0 >= 0.f || n.f1 >= 0.f;
        pre[k].prim = leaf ? (int32_t)k : -1;       // payload = position in `order` (the leaf's four payload floats are copied from there)
        const int32_t m = (int32_t)n.miss;
        if (leaf) pre[k].end = k + 1;
        else {
            if (m >= (int32_t)count || (m >= 0 && pos_of[m] <= (int32_t)k)) return -4;
            pre[k].end = m < 0 ? count : (uint32_t)pos_of[m];
        }
    }
    // a binary tree in pre-order: every inner node has exactly two children, the second where the first one's subtree ends
    for (uint32_t k = 0; k < count; k++) {
        if (pre[k].prim >= 0) continue;
        if (k + 1 >= count || pre[k + 1].end >= pre[k].end) return -4;
        const uint32_t b = pre[k + 1].end;
        if (pre[b].end != pre[k].end) return -4;
    }
    Options opt = options_from(user);
    if (opt.child_order == ATNS_ORDER_NEAR_POINT && !opt.order_point_given)
        for (int a = 0; a < 3; a++) opt.order_point[a] = pre[0].box.centre(a);
    Restructure rs(pre, opt);
    const uint64_t moved = rs.optimise(opt.reinsert_iterations, opt.reinsert_batch);
    const std::vector<BuildNode> fin = rs.emit();
    atn_bvh_node* nodes = emit(fin);
    if (!nodes) return -3;
    uint32_t leaves = 0;
    for (size_t i = 0; i < fin.size(); i++) {
        if (fin[i].prim < 0) continue;
        const atn_bvh_node& src = in[order[fin[i].prim]];
        nodes[i].f0 = src.f0; nodes[i].f1 = src.f1; nodes[i].f2 = src.f2; nodes[i].f3 = src.f3;
        leaves++;
    }
    *out_nodes = nodes; *out_count = (uint32_t)fin.size();
    if (stats) {
        stats->n_nodes = (uint32_t)fin.size(); stats->n_leaves = leaves; stats->n_spatial_splits = 0;
        stats->n_reinsertions = (uint32_t)std::min<uint64_t>(moved, 0xffffffffu);
        double sah = 0; const double ra = std::max((double)fin[0].box.half_area(), 1e-30);
        for (const BuildNode& nd : fin) sah += (double)nd.box.half_area() / ra;
        stats->sah_cost = (float)sah;
    }
    return 0;
}

int build_blas(const atn_vec4* vtx_pos, const atn_triangle_param* tris, const uint32_t* tri_ids, uint32_t n_tris,
               const atns_bvh_options* user, atn_bvh_node** out_nodes, uint32_t* out_count, float* out_bbox_min, float* out_bbox_max,
               atns_bvh_stats* stats)
{
    if (!vtx_pos || !tris || !tri_ids || !out_nodes || !out_count || n_tris == 0) return -1;
    std::vector<Ref> refs(n_tris);
    std::vector<Tri> geo(n_tris);
    for (uint32_t i = 0; i < n_tris; i++) {
        if (tri_ids[i] >= (1u << 24)) return -2;   // ids are stored as float: exact below 2^24
        const atn_triangle_param& t = tris[tri_ids[i]];
        Ref& r = refs[i];
        r.box.reset();
        for (int k = 0; k < 3; k++) {
            const atn_vec4& v = vtx_pos[t.idx[k]];
            geo[i].v[k][0] = v.x; geo[i].v[k][1] = v.y; geo[i].v[k][2] = v.z;
            r.box.grow(geo[i].v[k]);
        }
        r.prim = i;
    }
    Options opt = options_from(user);
    if (opt.child_order == ATNS_ORDER_NEAR_POINT && !opt.order_point_given) {
        // where the rays of an interior come from, for want of a better guess: the middle of the surface
        double acc = 0, c[3] = { 0, 0, 0 };
        for (const Tri& t : geo) {
            double e0[3], e1[3];
            for (int k = 0; k < 3; k++) { e0[k] = (double)t.v[1][k] - t.v[0][k]; e1[k] = (double)t.v[2][k] - t.v[0][k]; }
            const double cx = e0[1] * e1[2] - e0[2] * e1[1], cy = e0[2] * e1[0] - e0[0] * e1[2], cz = e0[0] * e1[1] - e0[1] * e1[0];
            const double area = 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz);
            for (int k = 0; k < 3; k++) c[k] += area * ((double)t.v[0][k] + t.v[1][k] + t.v[2][k]) / 3.0;
            acc += area;
        }
        for (int k = 0; k < 3; k++) opt.order_point[k] = acc > 0 ? (float)(c[k] / acc) : 0.5f * (geo[0].v[0][k] + geo[0].v[1][k]);
    }
    Builder b(opt, geo.data());
    b.run(refs);
    uint64_t moved = 0;
    {
        Restructure rs(b.nodes, opt);
        moved = rs.optimise(opt.reinsert_iterations, opt.reinsert_batch);
        b.nodes = rs.emit();
    }
    atn_bvh_node* nodes = emit(b.nodes);
    if (!nodes) return -3;
    for (size_t i = 0; i < b.nodes.size(); i++) {
        if (b.nodes[i].prim >= 0) {
            nodes[i].f0 = 1.0f;                                 // isleaf
            nodes[i].f1 = (float)tri_ids[b.nodes[i].prim];      // triid
            nodes[i].f2 = -1.0f;                                // AT_DISABLE_VOXEL
            nodes[i].f3 = -1.0f;
        }
    }
    *out_nodes = nodes;
    *out_count = (uint32_t)b.nodes.size();
    if (out_bbox_min && out_bbox_max) {
        for (int k = 0; k < 3; k++) { out_bbox_min[k] = b.nodes[0].box.mn[k]; out_bbox_max[k] = b.nodes[0].box.mx[k]; }
    }
    if (stats) {
        stats->n_nodes = (uint32_t)b.nodes.size();
        stats->n_leaves = (uint32_t)b.n_refs_out;
        stats->n_spatial_splits = (uint32_t)b.n_spatial;
        stats->n_reinsertions = (uint32_t)std::min<uint64_t>(moved, 0xffffffffu);
        double sah = 0;
        const double ra = std::max((double)b.nodes[0].box.half_area(), 1e-30);
        for (const BuildNode& nd : b.nodes) sah += (double)nd.box.half_area() / ra;
        stats->sah_cost = (float)sah;
    }
    return 0;
}

} // namespace

extern "C" {

uint32_t atns_abi_version(void) { return ATNS_ABI_VERSION; }

void atns_bvh_default_options(atns_bvh_options* o)
{
    if (!o) return;
    const Options d;
    o->spatial_splits = d.spatial ? 1 : 0;
    o->spatial_alpha = d.alpha;
    o->object_bins = d.object_bins;
    o->spatial_bins = d.spatial_bins;
    o->sweep_below = (int32_t)d.sweep_below;
    o->child_order = d.child_order;
    o->max_refs_factor = d.max_refs_factor;
    for (int k = 0; k < 3; k++) o->order_point[k] = d.order_point[k];
    o->order_point_given = d.order_point_given ? 1 : 0;
    o->reinsert_iterations = d.reinsert_iterations;
    o->reinsert_batch = d.reinsert_batch;
}

int atns_build_blas(const atn_vec4* vtx_pos, const atn_triangle_param* tris,
                    const uint32_t* tri_ids, uint32_t n_tris,
                    atn_bvh_node** out_nodes, uint32_t* out_count,
                    float out_bbox_min[3], float out_bbox_max[3])
{
    // The entry without options builds with OBJECT splits only: exactly one leaf per triangle, n - 1 inner nodes -- the shape
    // atn_lbvh_rebuild_list requires of a list it rebuilds in place (a spatial split duplicates references: more leaves than triangles).
    // Spatial splits are what atns_build_blas_opt's defaults add.
    try {
        atns_bvh_options o;
        atns_bvh_default_options(&o);
        o.spatial_splits = 0;
        return build_blas(vtx_pos, tris, tri_ids, n_tris, &o, out_nodes, out_count, out_bbox_min, out_bbox_max, nullptr);
    }
    catch (const std::bad_alloc&) { return -3; }
    catch (...) { return -5; }
}

int atns_build_blas_opt(const atn_vec4* vtx_pos, const atn_triangle_param* tris,
                        const uint32_t* tri_ids, uint32_t n_tris, const atns_bvh_options* options,
                        atn_bvh_node** out_nodes, uint32_t* out_count,
                        float out_bbox_min[3], float out_bbox_max[3], atns_bvh_stats* out_stats)
{
    try { return build_blas(vtx_pos, tris, tri_ids, n_tris, options, out_nodes, out_count, out_bbox_min, out_bbox_max, out_stats); }
    catch (const std::bad_alloc&) { return -3; }
    catch (...) { return -5; }
}

int atns_optimize_nodes(const atn_bvh_node* nodes, uint32_t count, const atns_bvh_options* options,
                         atn_bvh_node** out_nodes, uint32_t* out_count, atns_bvh_stats* out_stats)
{
    try { return optimise_nodes(nodes, count, options, out_nodes, out_count, out_stats); }
    catch (const std::bad_alloc&) { return -3; }
    catch (...) { return -5; }
}

int atns_build_tlas(const float* boxes, const int32_t* object_ids, const int32_t* blas_list_ids,
                    const int32_t* mesh_ids, uint32_t n,
                    atn_bvh_node** out_nodes, uint32_t* out_count)
{
    if (!boxes || !object_ids || !blas_list_ids || !out_nodes || !out_count || n == 0) return -1;
    try {
        std::vector<Ref> refs(n);
        for (uint32_t i = 0; i < n; i++) {
            Ref& r = refs[i];
            for (int k = 0; k < 3; k++) { r.box.mn[k] = boxes[6 * i + k]; r.box.mx[k] = boxes[6 * i + 3 + k]; }
            r.prim = i;
        }
        Options o;
        o.spatial = false;
        // (no viewer hint at this entry point: children nearer to the middle of the instances are threaded first)
        Box all = empty_box();
        for (const Ref& r : refs) all.grow(r.box);
        for (int k = 0; k < 3; k++) o.order_point[k] = all.centre(k);
        Builder b(o, nullptr);
        b.run(refs);
        {
            Restructure rs(b.nodes, o);
            rs.optimise(o.reinsert_iterations, o.reinsert_batch);
            b.nodes = rs.emit();
        }
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
    catch (const std::bad_alloc&) { return -3; }
    catch (...) { return -5; }
}

int atns_anyhit_twin(const atn_bvh_node* nodes, uint32_t count, atn_bvh_node** out_nodes, double* out_cost_as_given, double* out_cost_twin)
{
    if (!nodes || !out_nodes) return -1;
    try {
        atn::AnyhitTwin tw;
        if (!atn::make_anyhit_twin(nodes, count, tw)) return -4;
        atn_bvh_node* o = (atn_bvh_node*)std::malloc(sizeof(atn_bvh_node) * tw.nodes.size());
        if (!o) return -3;
        std::memcpy(o, tw.nodes.data(), sizeof(atn_bvh_node) * tw.nodes.size());
        *out_nodes = o;
        if (out_cost_as_given) *out_cost_as_given = tw.cost_as_given;
        if (out_cost_twin) *out_cost_twin = tw.cost_twin;
        return 0;
    }
    catch (const std::bad_alloc&) { return -3; }
    catch (...) { return -5; }
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
