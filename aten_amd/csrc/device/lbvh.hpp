// LBVH build on the device: the bottom-level tree of a deforming mesh, rebuilt every tick.
//
// ≙ idaten::LBVHBuilder::onBuild (src/libidaten/kernel/LBVHBuilder.cu:700-810), which the reference runs per tick on its
// deformable (src/deformation_renderer/main.cpp:636-710): Morton codes of the triangle-box centres, a key / value radix
// sort, Karras' hierarchy over the sorted codes, hit / miss threading, bottom-up boxes.  The stages produce the SAME
// tree, node for node and bit for bit, as the reference's (tests/test_gpu_lbvh.py compares with a CPU restatement of it);
// how they run is this file's own:
//
//   k_lbvh_morton      one thread per triangle                                   (MortonCode.cuh:147-197)
//   k_radix_count / k_radix_scan / k_radix_scatter  x 4 digits of 8 bits: a stable LSD radix sort written for 64-wide
//                      waves -- the in-block rank of a key is popcount(lanes below me with my digit), found with 8
//                      ballots, no LDS atomics in the scatter             (the reference calls thrust::sort_by_key)
//   k_lbvh_hierarchy   one thread per inner node: findSpan / findSplit           (LBVHBuilder.cu:193-350)
//   k_lbvh_order       one thread per node: hit / miss links, leaf payload       (LBVHBuilder.cu:353-489)
//   k_lbvh_bounds      one thread per leaf walking up; the second child to arrive at a node merges (an atomic counter
//                      per inner node; min / max are exact, so arrival order cannot change a bit) (LBVHBuilder.cu:533-680)
//   k_lbvh_layout      byte offset of every node's record in walk (pre-)order: a node whose range starts at sorted leaf
//                      `a`, with L left-turns on the way down from the root, is preceded by `a` leaves and a + L inner
//                      nodes -> offset = base + 32 (a + L) + 48 a.  No scan, no sort.
//   k_lbvh_emit        writes the records of device/scene_dev.hpp (32-byte inner, 48-byte triangle leaf with v0 / e1 / e2
//                      hoisted) straight into the scene's node image: the rebuilt tree never visits the host.
//
// Everything here is HBM / latency bound integer and pointer work on a few MB (n triangles -> 2 n - 1 nodes); per
// triangle the pipeline moves ~0.5 KB.
#pragma once
#include "scene_dev.hpp"

namespace atn {

constexpr uint32_t kSortThreads = 256;
constexpr uint32_t kSortRounds = 16;
constexpr uint32_t kSortTile = kSortThreads * kSortRounds;      // keys per block and pass
constexpr uint32_t kLbvhMaxTris = 1u << 23;                     // node indices are stored as floats (ThreadedBvhNode::hit)

struct LbvhTopo {
    int32_t* left;      // [2n-1]  children (-1 on leaves)
    int32_t* right;
    int32_t* parent;    // [2n-1]  -1 on the root
    int32_t* first;     // [n-1]   first sorted leaf of an inner node's range
};

__device__ __forceinline__ uint32_t lbvh_expand_bits(uint32_t v)
{
    v = (v | v << 16) & 0xFF0000FFu;
    v = (v | v << 8) & 0x0F00F00Fu;
    v = (v | v << 4) & 0xC30C30C3u;
    v = (v | v << 2) & 0x49249249u;
    return v;
}

// min(max(x * 1024, 0), 1023) with CUDA's float min / max = fminf / fmaxf (a NaN from a flat axis becomes 0)
__device__ __forceinline__ uint32_t lbvh_quantise(float x) { return (uint32_t)fminf(fmaxf(x * 1024.0F, 0.0F), 1023.0F); }

__device__ __forceinline__ void lbvh_triangle_box(const atn_triangle_param* tris, const float4* vtx, int32_t vtx_offset, uint32_t tri,
                                                  f3& mn, f3& mx)
{
    const int32_t i0 = tris[tri].idx[0] + vtx_offset, i1 = tris[tri].idx[1] + vtx_offset, i2 = tris[tri].idx[2] + vtx_offset;
    const float4 v0 = vtx[i0], v1 = vtx[i1], v2 = vtx[i2];
    mn = mk3(fminf(fminf(v0.x, v1.x), v2.x), fminf(fminf(v0.y, v1.y), v2.y), fminf(fminf(v0.z, v1.z), v2.z));
    mx = mk3(fmaxf(fmaxf(v0.x, v1.x), v2.x), fmaxf(fmaxf(v0.y, v1.y), v2.y), fmaxf(fmaxf(v0.z, v1.z), v2.z));
}

__global__ __launch_bounds__(256) void k_lbvh_morton(const atn_triangle_param* __restrict__ tris, const float4* __restrict__ vtx,
                                                     int32_t vtx_offset, uint32_t n, f3 bmin, f3 bmax,
                                                     uint32_t* __restrict__ codes, uint32_t* __restrict__ indices)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 mn, mx;
    lbvh_triangle_box(tris, vtx, vtx_offset, i, mn, mx);
    const f3 size = bmax - bmin;                    // aabb::size
    f3 c = (mn + mx) * 0.5F;
    c = (c - bmin) / size;
    codes[i] = lbvh_expand_bits(lbvh_quantise(c.x)) << 2 | lbvh_expand_bits(lbvh_quantise(c.y)) << 1 | lbvh_expand_bits(lbvh_quantise(c.z));
    indices[i] = i;
}

// ---------------------------------------------------------------------------------------------------------------------
// stable LSD radix sort, 8 bits per pass.  counts[digit * n_blocks + block].
__global__ __launch_bounds__(kSortThreads) void k_radix_count(const uint32_t* __restrict__ keys, uint32_t n, uint32_t shift,
                                                              uint32_t* __restrict__ counts, uint32_t n_blocks)
{
    __shared__ uint32_t hist[256];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kSortTile;
    for (uint32_t r = 0; r < kSortRounds; r++) {
        const uint32_t g = base + r * kSortThreads + threadIdx.x;
        if (g < n) atomicAdd(&hist[(keys[g] >> shift) & 255u], 1u);
    }
    __syncthreads();
    counts[threadIdx.x * n_blocks + blockIdx.x] = hist[threadIdx.x];
}

// exclusive scan of `total` counters in place, one block of 1024 threads (total = 256 * n_blocks: 64 K entries for a
// million keys)
__global__ __launch_bounds__(1024) void k_radix_scan(uint32_t* __restrict__ counts, uint32_t total)
{
    __shared__ uint32_t part[1024];
    const uint32_t per = (total + 1023u) / 1024u;
    const uint32_t lo = threadIdx.x * per, hi = min(lo + per, total);
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; i++) s += counts[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;
    for (uint32_t i = lo; i < hi; i++) { const uint32_t c = counts[i]; counts[i] = run; run += c; }
}

__global__ __launch_bounds__(kSortThreads) void k_radix_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                                uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                                uint32_t n, uint32_t shift, const uint32_t* __restrict__ offsets, uint32_t n_blocks)
{
    constexpr uint32_t kWaves = kSortThreads / 64;
    __shared__ uint32_t run[256];               // where this block's next key with digit d goes
    __shared__ uint32_t wcount[kWaves][256];    // keys with digit d in wave w of the current round
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    run[tid] = offsets[tid * n_blocks + blockIdx.x];
    for (uint32_t w = 0; w < kWaves; w++) wcount[w][tid] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kSortTile;
    for (uint32_t r = 0; r < kSortRounds; r++) {
        const uint32_t g = base + r * kSortThreads + tid;
        const bool valid = g < n;
        const uint32_t key = valid ? keys_in[g] : 0u, val = valid ? vals_in[g] : 0u;
        const uint32_t d = (key >> shift) & 255u;
        // lanes of my wave holding the same digit
        unsigned long long peers = __ballot(valid);
        for (uint32_t b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t rank = __popcll(peers & ((1ull << lane) - 1ull));
        if (valid && rank == 0) wcount[wave][d] = __popcll(peers);
        __syncthreads();
        if (valid) {
            uint32_t pos = run[d] + rank;
            for (uint32_t w = 0; w < wave; w++) pos += wcount[w][d];
            keys_out[pos] = key; vals_out[pos] = val;
        }
        __syncthreads();
        uint32_t add = 0;
        for (uint32_t w = 0; w < kWaves; w++) { add += wcount[w][tid]; wcount[w][tid] = 0; }
        run[tid] += add;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t lbvh_clz(uint32_t x) { return x ? __builtin_clz(x) : 32; }

// computeLongestCommonPrefix, LBVHBuilder.cu:194-216: equal codes fall back to the positions
__device__ __forceinline__ int32_t lbvh_lcp(const uint32_t* __restrict__ keys, int32_t n, int32_t i1, int32_t i2)
{
    const int32_t l = min(i1, i2), r = max(i1, i2);
    if (l < 0 || r >= n) return -1;
    const uint32_t a = keys[l], b = keys[r];
    return a != b ? lbvh_clz(a ^ b) : 32 + lbvh_clz((uint32_t)(l ^ r));
}

__global__ __launch_bounds__(256) void k_lbvh_hierarchy(const uint32_t* __restrict__ keys, uint32_t n, LbvhTopo t)
{
    const int32_t idx = blockIdx.x * blockDim.x + threadIdx.x, num = (int32_t)n;
    if (idx >= num - 1) return;
    // findSpan, LBVHBuilder.cu:218-263
    const int32_t d = (lbvh_lcp(keys, num, idx, idx + 1) - lbvh_lcp(keys, num, idx, idx - 1)) < 0 ? -1 : 1;
    const int32_t min_lcp = lbvh_lcp(keys, num, idx, idx - d);
    int32_t lmax = 2;
    while (lbvh_lcp(keys, num, idx, idx + lmax * d) > min_lcp) lmax *= 2;
    int32_t l = 0, s = lmax;
    do {
        s /= 2;
        if (lbvh_lcp(keys, num, idx, idx + (l + s) * d) > min_lcp) l += s;
    } while (s > 1);
    const int32_t lo = min(idx, idx + l * d), hi = max(idx, idx + l * d);
    // findSplit, LBVHBuilder.cu:266-297
    int32_t left = lo, right = hi;
    const int32_t identical = lbvh_lcp(keys, num, left, right);
    do {
        const int32_t mid = (right + left) / 2;
        if (lbvh_lcp(keys, num, left, mid) > identical) left = mid;
        else right = mid;
    } while (right > left + 1);
    const int32_t split = left;
    // buildTree, LBVHBuilder.cu:299-350
    const int32_t cl = split == lo ? split + num - 1 : split;
    const int32_t cr = split + 1 == hi ? split + 1 + num - 1 : split + 1;
    if (idx == 0) t.parent[0] = -1;
    t.left[idx] = cl; t.right[idx] = cr; t.first[idx] = lo;
    t.parent[cl] = idx; t.parent[cr] = idx;
    if (cl >= num - 1) { t.left[cl] = -1; t.right[cl] = -1; }
    if (cr >= num - 1) { t.left[cr] = -1; t.right[cr] = -1; }
}

// onApplyTraverseOrder, LBVHBuilder.cu:353-470.  An inner node's hit link is its left child; a node's miss link is the
// right sibling of the nearest ancestor-or-self that is a left child (-1 when there is none); a leaf's hit link equals
// its miss link.  Leaves carry isleaf = 1 (GPGPU_TRAVERSE_SBVH) and the triangle id as a float.
__global__ __launch_bounds__(256) void k_lbvh_order(uint32_t n, int32_t tri_id_offset, LbvhTopo t, const uint32_t* __restrict__ sorted_indices,
                                                    atn_bvh_node* __restrict__ out)
{
    const int32_t idx = blockIdx.x * blockDim.x + threadIdx.x, num = (int32_t)n;
    if (idx >= 2 * num - 1) return;
    const bool leaf = idx >= num - 1;
    int32_t cur = idx, miss = -1;
    for (;;) {
        const int32_t p = t.parent[cur];
        if (p < 0) break;
        if (t.left[p] == cur) { miss = t.right[p]; break; }
        cur = p;
    }
    atn_bvh_node& g = out[idx];
    g.hit = leaf ? (float)miss : (float)t.left[idx];
    g.miss = (float)miss;
    g.f0 = leaf ? 1.0F : -1.0F;
    g.f1 = leaf ? (float)(tri_id_offset + (int32_t)sorted_indices[idx - (num - 1)]) : -1.0F;
    g.f2 = -1.0F; g.f3 = -1.0F;
}

__device__ __forceinline__ float lbvh_ld_coherent(const float* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// computeBoudingBox, LBVHBuilder.cu:533-680.  `arrived` [n-1] is zero on entry.
__global__ __launch_bounds__(256) void k_lbvh_bounds(uint32_t n, LbvhTopo t, const uint32_t* __restrict__ sorted_indices,
                                                     const atn_triangle_param* __restrict__ tris, const float4* __restrict__ vtx, int32_t vtx_offset,
                                                     atn_bvh_node* out, uint32_t* arrived)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x, num = (int32_t)n;
    if (i >= num) return;
    f3 mn, mx;
    lbvh_triangle_box(tris, vtx, vtx_offset, sorted_indices[i], mn, mx);
    int32_t cur = i + num - 1;
    for (;;) {
        atn_bvh_node& g = out[cur];
        g.boxmin[0] = mn.x; g.boxmin[1] = mn.y; g.boxmin[2] = mn.z;
        g.boxmax[0] = mx.x; g.boxmax[1] = mx.y; g.boxmax[2] = mx.z;
        const int32_t p = t.parent[cur];
        if (p < 0) return;
        __threadfence();                                    // my box before my arrival
        if (atomicAdd(&arrived[p], 1u) == 0u) return;       // the sibling subtree is not finished: its thread goes on
        __threadfence();
        const int32_t other = t.left[p] == cur ? t.right[p] : t.left[p];
        const atn_bvh_node& o = out[other];
        mn = mk3(fminf(mn.x, lbvh_ld_coherent(&o.boxmin[0])), fminf(mn.y, lbvh_ld_coherent(&o.boxmin[1])), fminf(mn.z, lbvh_ld_coherent(&o.boxmin[2])));
        mx = mk3(fmaxf(mx.x, lbvh_ld_coherent(&o.boxmax[0])), fmaxf(mx.y, lbvh_ld_coherent(&o.boxmax[1])), fmaxf(mx.z, lbvh_ld_coherent(&o.boxmax[2])));
        cur = p;
    }
}

// Byte offset of every node's device record, walk (pre-)order from `base`.
__global__ __launch_bounds__(256) void k_lbvh_layout(uint32_t n, LbvhTopo t, uint32_t base, uint32_t* __restrict__ offs)
{
    const int32_t idx = blockIdx.x * blockDim.x + threadIdx.x, num = (int32_t)n;
    if (idx >= 2 * num - 1) return;
    const uint32_t a = idx >= num - 1 ? (uint32_t)(idx - (num - 1)) : (uint32_t)t.first[idx];
    uint32_t lefts = 0;
    for (int32_t cur = idx;;) {
        const int32_t p = t.parent[cur];
        if (p < 0) break;
        lefts += t.left[p] == cur ? 1u : 0u;
        cur = p;
    }
    offs[idx] = base + kInnerBytes * (a + lefts) + kTriLeafBytes * a;
}

__global__ __launch_bounds__(256) void k_lbvh_emit(uint32_t n, const atn_bvh_node* __restrict__ nodes, const uint32_t* __restrict__ offs,
                                                   const atn_triangle_param* __restrict__ scene_tris, const float4* __restrict__ scene_vtx,
                                                   float4* __restrict__ image)
{
    const int32_t idx = blockIdx.x * blockDim.x + threadIdx.x, num = (int32_t)n;
    if (idx >= 2 * num - 1) return;
    const atn_bvh_node nd = nodes[idx];
    auto typed = [&](float link) -> int32_t {
        const int32_t l = (int32_t)link;
        if (l < 0) return kLinkEnd;
        return (int32_t)offs[l] | (l >= num - 1 ? kLinkLeafBit : 0);
    };
    float4* q = reinterpret_cast<float4*>(reinterpret_cast<char*>(image) + offs[idx]);
    if (idx < num - 1) {
        q[0] = make_float4(nd.boxmin[0], nd.boxmin[1], nd.boxmin[2], __int_as_float(typed(nd.hit)));
        q[1] = make_float4(nd.boxmax[0], nd.boxmax[1], nd.boxmax[2], __int_as_float(typed(nd.miss)));
    }
    else {
        const int32_t tri = (int32_t)nd.f1;             // an id in the SCENE's triangle array
        const float4 a = scene_vtx[scene_tris[tri].idx[0]], b = scene_vtx[scene_tris[tri].idx[1]], c = scene_vtx[scene_tris[tri].idx[2]];
        q[0] = make_float4(a.x, a.y, a.z, __int_as_float(tri));
        q[1] = make_float4(b.x - a.x, b.y - a.y, b.z - a.z, __int_as_float(typed(nd.hit)));
        q[2] = make_float4(c.x - a.x, c.y - a.y, c.z - a.z, 0.0F);
    }
}

} // namespace atn
