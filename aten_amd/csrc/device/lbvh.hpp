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
//   k_lbvh_links       one thread per node walking to the root: hit / miss links, leaf payload (LBVHBuilder.cu:353-489)
//                      and, from the left turns it counts on the way, the byte offset of the node's record in walk
//                      (pre-)order: a node whose range starts at sorted leaf `a`, L left turns below the root, is
//                      preceded by `a` leaves and a + L inner nodes -> offset = base + 32 (a + L) + 48 a.  No scan.
//   k_lbvh_bounds_*    boxes = unions over ranges of sorted leaves, evaluated block-locally plus a range query for
//                      the nodes that cross block boundaries -- no device-wide hand-over (see there) (LBVHBuilder.cu:533-680)
//   k_lbvh_emit        writes the records of device/scene_dev.hpp (32-byte inner, 48-byte triangle leaf with v0 / e1 / e2
//                      hoisted) straight into the scene's node image: the rebuilt tree never visits the host.
//
// Everything here is HBM / latency bound integer and pointer work on a few MB (n triangles -> 2 n - 1 nodes); per
// triangle the pipeline moves ~0.5 KB.
#pragma once
#include "scene_dev.hpp"

namespace atn {

constexpr uint32_t kSortThreads = 256;
constexpr uint32_t kSortMaxRounds = 16;                         // a block sorts up to 256 x 16 keys per pass ...
// ... and fewer when that would leave the chip idle: small meshes (the usual deformable is 10^4 .. 10^5 triangles) get
// at least ~256 blocks, down to one round of 256 keys per block (measured at 12 852 triangles: 15 -> 3 us per scatter)
inline uint32_t radix_rounds(uint32_t n)
{
    const uint32_t r = n / (256u * kSortThreads);
    return r < 1u ? 1u : (r > kSortMaxRounds ? kSortMaxRounds : r);
}
constexpr uint32_t kLbvhMaxTris = 1u << 23;                     // node indices are stored as floats (ThreadedBvhNode::hit)

struct LbvhTopo {
    int32_t* left;      // [2n-1]  children (-1 on leaves)
    int32_t* right;
    int32_t* parent;    // [2n-1]  -1 on the root
    int32_t* first;     // [n-1]   first sorted leaf of an inner node's range
    int32_t* last;      // [n-1]   last
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
                                                              uint32_t* __restrict__ counts, uint32_t n_blocks, uint32_t rounds)
{
    __shared__ uint32_t hist[256];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kSortThreads * rounds;
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t g = base + r * kSortThreads + threadIdx.x;
        if (g < n) atomicAdd(&hist[(keys[g] >> shift) & 255u], 1u);
    }
    __syncthreads();
    counts[threadIdx.x * n_blocks + blockIdx.x] = hist[threadIdx.x];
}

// Row d of counts[256][n_blocks] -> its exclusive scan over the blocks, and totals[d] = keys with digit d.  One wave per
// digit (grid 256): coalesced 64-entry reads, a shuffle prefix sum.  The scatter adds the digits' own prefix itself.
__global__ __launch_bounds__(64) void k_radix_scan(uint32_t* __restrict__ counts, uint32_t n_blocks, uint32_t* __restrict__ totals)
{
    const uint32_t lane = threadIdx.x, row = blockIdx.x;
    uint32_t run = 0;
    for (uint32_t base = 0; base < n_blocks; base += 64) {
        const uint32_t i = base + lane;
        const uint32_t v = i < n_blocks ? counts[row * n_blocks + i] : 0u;
        uint32_t incl = v;
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (i < n_blocks) counts[row * n_blocks + i] = run + incl - v;
        run += __shfl(incl, 63);
    }
    if (lane == 0) totals[row] = run;
}

__global__ __launch_bounds__(kSortThreads) void k_radix_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                                uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                                uint32_t n, uint32_t shift, const uint32_t* __restrict__ offsets, uint32_t n_blocks,
                                                                const uint32_t* __restrict__ totals, uint32_t rounds)
{
    constexpr uint32_t kWaves = kSortThreads / 64;
    __shared__ uint32_t run[256];               // where this block's next key with digit d goes
    __shared__ uint32_t wcount[kWaves][256];    // keys with digit d in wave w of the current round
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    {   // keys with a smaller digit come first: exclusive scan of the 256 digit totals (4 waves x 64 lanes)
        const uint32_t v = totals[tid];
        uint32_t incl = v;
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (lane == 63) wcount[0][wave] = incl;
        __syncthreads();
        uint32_t base = 0;
        for (uint32_t w = 0; w < wave; w++) base += wcount[0][w];
        run[tid] = base + incl - v + offsets[tid * n_blocks + blockIdx.x];
        __syncthreads();
    }
    for (uint32_t w = 0; w < kWaves; w++) wcount[w][tid] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kSortThreads * rounds;
    for (uint32_t r = 0; r < rounds; r++) {
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

__global__ __launch_bounds__(256) void k_lbvh_hierarchy(const uint32_t* __restrict__ keys, uint32_t n, LbvhTopo t, uint32_t* __restrict__ arrived)
{
    const int32_t idx = blockIdx.x * blockDim.x + threadIdx.x, num = (int32_t)n;
    if (idx >= num - 1) return;
    arrived[idx] = 0;       // k_lbvh_bounds_block's hand-over counters
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
    t.left[idx] = cl; t.right[idx] = cr; t.first[idx] = lo; t.last[idx] = hi;
    t.parent[cl] = idx; t.parent[cr] = idx;
    if (cl >= num - 1) { t.left[cl] = -1; t.right[cl] = -1; }
    if (cr >= num - 1) { t.left[cr] = -1; t.right[cr] = -1; }
}

// onApplyTraverseOrder, LBVHBuilder.cu:353-470.  An inner node's hit link is its left child; a node's miss link is the
// right sibling of the nearest ancestor-or-self that is a left child (-1 when there is none); a leaf's hit link equals
// its miss link.  Leaves carry isleaf = 1 (GPGPU_TRAVERSE_SBVH) and the triangle id as a float.
// The same walk to the root counts the left turns, which gives the byte offset of the node's device record in walk
// (pre-)order from `base`: a node whose range starts at sorted leaf `a` is preceded by `a` leaves and a + lefts inner
// nodes.  No scan, no sort.
__global__ __launch_bounds__(256) void k_lbvh_links(uint32_t n, int32_t tri_id_offset, LbvhTopo t, const uint32_t* __restrict__ sorted_indices,
                                                    atn_bvh_node* __restrict__ out, uint32_t base, uint32_t* __restrict__ offs)
{
    const int32_t idx = blockIdx.x * blockDim.x + threadIdx.x, num = (int32_t)n;
    if (idx >= 2 * num - 1) return;
    const bool leaf = idx >= num - 1;
    int32_t miss = -1;
    uint32_t lefts = 0;
    for (int32_t cur = idx;;) {
        const int32_t p = t.parent[cur];
        if (p < 0) break;
        if (t.left[p] == cur) { if (lefts == 0) miss = t.right[p]; lefts++; }
        cur = p;
    }
    atn_bvh_node& g = out[idx];
    g.hit = leaf ? (float)miss : (float)t.left[idx];
    g.miss = (float)miss;
    g.f0 = leaf ? 1.0F : -1.0F;
    g.f1 = leaf ? (float)(tri_id_offset + (int32_t)sorted_indices[idx - (num - 1)]) : -1.0F;
    g.f2 = -1.0F; g.f3 = -1.0F;
    const uint32_t a = leaf ? (uint32_t)(idx - (num - 1)) : (uint32_t)t.first[idx];
    offs[idx] = base + kInnerBytes * (a + lefts) + kTriLeafBytes * a;
}

// computeBoudingBox, LBVHBuilder.cu:533-680, without a single device-wide hand-over.  The reference lets one thread per
// leaf climb and the second child to arrive at a node merge, synchronised by an atomic per node; on this chip every such
// hand-over between workgroups costs an agent-scope release, i.e. a write-back of an XCD's L2 (measured: 6.7 ms of a
// 7.4 ms rebuild of 10^6 triangles).  A node's box is the union of the leaf boxes of its RANGE of sorted leaves, and
// min / max are exact, so any other evaluation order gives the same bits:
//
//   k_lbvh_bounds_block  a block owns 256 consecutive sorted leaves.  Nodes whose range lies inside them are only ever
//                        touched by this block: the climb of the reference, with workgroup-scope hand-overs (one CU, one
//                        L1).  It also leaves, per leaf i, the union of the block's leaves up to i (`pre`) and from i on
//                        (`suf`) -- two 8-step scans in LDS.
//   k_lbvh_bounds_super  union of every 64 blocks (a wave per super-block).
//   k_lbvh_bounds_cross  a thread per inner node whose range [a, b] crosses block boundaries (a few per cent of them):
//                        suf[a] U whole blocks U whole super-blocks U whole blocks U pre[b].
constexpr uint32_t kBoundsBlock = 256;
constexpr uint32_t kBoundsSuper = 64;       // blocks per super-block

struct LbvhBox { float mn[3], mx[3]; };

__device__ __forceinline__ void lbvh_merge(f3& mn, f3& mx, const f3& omn, const f3& omx)
{
    mn = mk3(fminf(mn.x, omn.x), fminf(mn.y, omn.y), fminf(mn.z, omn.z));
    mx = mk3(fmaxf(mx.x, omx.x), fmaxf(mx.y, omx.y), fmaxf(mx.z, omx.z));
}
__device__ __forceinline__ void lbvh_merge(f3& mn, f3& mx, const LbvhBox& b)
{
    lbvh_merge(mn, mx, mk3(b.mn[0], b.mn[1], b.mn[2]), mk3(b.mx[0], b.mx[1], b.mx[2]));
}
__device__ __forceinline__ void lbvh_store(LbvhBox& b, const f3& mn, const f3& mx)
{
    b.mn[0] = mn.x; b.mn[1] = mn.y; b.mn[2] = mn.z; b.mx[0] = mx.x; b.mx[1] = mx.y; b.mx[2] = mx.z;
}

__global__ __launch_bounds__(kBoundsBlock) void k_lbvh_bounds_block(uint32_t n, LbvhTopo t, const uint32_t* __restrict__ sorted_indices,
                                                                    const atn_triangle_param* __restrict__ tris, const float4* __restrict__ vtx,
                                                                    int32_t vtx_offset, atn_bvh_node* out, uint32_t* arrived,
                                                                    LbvhBox* __restrict__ pre, LbvhBox* __restrict__ suf)
{
    __shared__ float sc[2][6][kBoundsBlock];     // [prefix | suffix][component][leaf]
    const int32_t tid = threadIdx.x, i = blockIdx.x * kBoundsBlock + tid, num = (int32_t)n;
    const int32_t blk_lo = blockIdx.x * kBoundsBlock, blk_hi = blk_lo + (int32_t)kBoundsBlock - 1;
    const bool valid = i < num;
    f3 mn = mk3(INFINITY), mx = mk3(-INFINITY);
    if (valid) lbvh_triangle_box(tris, vtx, vtx_offset, sorted_indices[i], mn, mx);
    for (int s = 0; s < 2; s++) {
        sc[s][0][tid] = mn.x; sc[s][1][tid] = mn.y; sc[s][2][tid] = mn.z;
        sc[s][3][tid] = mx.x; sc[s][4][tid] = mx.y; sc[s][5][tid] = mx.z;
    }
    __syncthreads();
    for (int32_t d = 1; d < (int32_t)kBoundsBlock; d <<= 1) {
        float v[2][6];
        const bool lo_ok = tid >= d, hi_ok = tid + d < (int32_t)kBoundsBlock;
        for (int c = 0; c < 6; c++) {
            v[0][c] = lo_ok ? sc[0][c][tid - d] : sc[0][c][tid];
            v[1][c] = hi_ok ? sc[1][c][tid + d] : sc[1][c][tid];
        }
        __syncthreads();
        for (int c = 0; c < 3; c++) {
            sc[0][c][tid] = fminf(sc[0][c][tid], v[0][c]); sc[0][c + 3][tid] = fmaxf(sc[0][c + 3][tid], v[0][c + 3]);
            sc[1][c][tid] = fminf(sc[1][c][tid], v[1][c]); sc[1][c + 3][tid] = fmaxf(sc[1][c + 3][tid], v[1][c + 3]);
        }
        __syncthreads();
    }
    if (!valid) return;
    lbvh_store(pre[i], mk3(sc[0][0][tid], sc[0][1][tid], sc[0][2][tid]), mk3(sc[0][3][tid], sc[0][4][tid], sc[0][5][tid]));
    lbvh_store(suf[i], mk3(sc[1][0][tid], sc[1][1][tid], sc[1][2][tid]), mk3(sc[1][3][tid], sc[1][4][tid], sc[1][5][tid]));
    // the reference's climb, inside the block
    int32_t cur = i + num - 1;
    for (;;) {
        atn_bvh_node& g = out[cur];
        g.boxmin[0] = mn.x; g.boxmin[1] = mn.y; g.boxmin[2] = mn.z;
        g.boxmax[0] = mx.x; g.boxmax[1] = mx.y; g.boxmax[2] = mx.z;
        const int32_t p = t.parent[cur];
        if (p < 0 || t.first[p] < blk_lo || t.last[p] > blk_hi) return;        // the root, or a node of k_lbvh_bounds_cross
        if (__hip_atomic_fetch_add(&arrived[p], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) return;
        const atn_bvh_node& o = out[t.left[p] == cur ? t.right[p] : t.left[p]];
        lbvh_merge(mn, mx, mk3(o.boxmin[0], o.boxmin[1], o.boxmin[2]), mk3(o.boxmax[0], o.boxmax[1], o.boxmax[2]));
        cur = p;
    }
}

// sup[s] = union of blocks [64 s, 64 s + 63]; a block's union is the `pre` of its last leaf
__global__ __launch_bounds__(64) void k_lbvh_bounds_super(uint32_t n, const LbvhBox* __restrict__ pre, LbvhBox* __restrict__ sup)
{
    const uint32_t n_blocks = (n + kBoundsBlock - 1) / kBoundsBlock;
    const uint32_t blk = blockIdx.x * kBoundsSuper + threadIdx.x;
    f3 mn = mk3(INFINITY), mx = mk3(-INFINITY);
    if (blk < n_blocks) lbvh_merge(mn, mx, pre[min(blk * kBoundsBlock + kBoundsBlock - 1, n - 1)]);
    for (int d = 32; d > 0; d >>= 1)
        lbvh_merge(mn, mx, mk3(__shfl_xor(mn.x, d), __shfl_xor(mn.y, d), __shfl_xor(mn.z, d)), mk3(__shfl_xor(mx.x, d), __shfl_xor(mx.y, d), __shfl_xor(mx.z, d)));
    if (threadIdx.x == 0) lbvh_store(sup[blockIdx.x], mn, mx);
}

__global__ __launch_bounds__(256) void k_lbvh_bounds_cross(uint32_t n, LbvhTopo t, const LbvhBox* __restrict__ pre, const LbvhBox* __restrict__ suf,
                                                           const LbvhBox* __restrict__ sup, atn_bvh_node* __restrict__ out)
{
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx + 1 >= n) return;
    const uint32_t a = (uint32_t)t.first[idx], b = (uint32_t)t.last[idx];
    const uint32_t A = a / kBoundsBlock, B = b / kBoundsBlock;
    if (A == B) return;
    f3 mn = mk3(INFINITY), mx = mk3(-INFINITY);
    lbvh_merge(mn, mx, suf[a]);
    lbvh_merge(mn, mx, pre[b]);
    uint32_t x = A + 1;                                        // whole blocks x .. B - 1
    auto block_box = [&](uint32_t blk) -> const LbvhBox& { return pre[min(blk * kBoundsBlock + kBoundsBlock - 1, n - 1)]; };
    for (; x < B && (x % kBoundsSuper) != 0; x++) lbvh_merge(mn, mx, block_box(x));
    for (; x + kBoundsSuper <= B; x += kBoundsSuper) lbvh_merge(mn, mx, sup[x / kBoundsSuper]);
    for (; x < B; x++) lbvh_merge(mn, mx, block_box(x));
    atn_bvh_node& g = out[idx];
    g.boxmin[0] = mn.x; g.boxmin[1] = mn.y; g.boxmin[2] = mn.z;
    g.boxmax[0] = mx.x; g.boxmax[1] = mx.y; g.boxmax[2] = mx.z;
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
        return (int32_t)offs[l] | (l >= num - 1 ? kLinkToLeaf : 0);
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

// ---- the any-hit twins of a rebuilt list (host: csrc/host/anyhit_twin.hpp, scene_upload.hpp) ---------------------------------------
// A twin is another THREADING of the same tree -- only the order of an inner node's two children may differ -- that the shadow rays of
// infinite lights walk; any order gives the same answer, a good one finds the occluder sooner.  After a rebuild the list is a new
// tree, so its twins are re-threaded here, on the device, from the tree that was just built: twin g (one per octant of the ray's
// direction, bit a = dir[a] > 0) enters FIRST the child whose exit face along that octant's sign vector lies further out (back to
// front: what blocks a ray that leaves a surface lies far along it) when the two exit faces are more than 5 % of the node's extent
// apart -- the host rule's direction part; where they are not, the children stay in the list's own order (the host consults its
// surface-area model there; the film does not depend on the choice, tests/test_gpu_anyhit_twin.py).
// k_lbvh_twin_flips: one byte per inner node, bit g = "twin g swaps the children".
__global__ __launch_bounds__(256) void k_lbvh_twin_flips(uint32_t n, LbvhTopo t, const atn_bvh_node* __restrict__ nodes, uint32_t n_dir, uint8_t* __restrict__ flips)
{
    const int32_t idx = blockIdx.x * blockDim.x + threadIdx.x, num = (int32_t)n;
    if (idx >= num - 1) return;
    uint32_t m = 0;
    if (n_dir == 8u) {
        const atn_bvh_node na = nodes[t.left[idx]], nb = nodes[t.right[idx]], nn = nodes[idx];
        float ext = 0.0F;
        for (int a = 0; a < 3; a++) ext += nn.boxmax[a] - nn.boxmin[a];
        for (uint32_t g = 0; g < 8u; g++) {
            float ea = 0.0F, eb = 0.0F;
            for (int a = 0; a < 3; a++) {
                if (g & (1u << a)) { ea += na.boxmax[a]; eb += nb.boxmax[a]; }
                else { ea -= na.boxmin[a]; eb -= nb.boxmin[a]; }
            }
            if (eb - ea > 0.05F * ext) m |= 1u << g;
        }
    }
    flips[idx] = (uint8_t)m;
}

// k_lbvh_twin_emit: the records of twin g (blockIdx.y) -- the list's records with the links of the other child order -- at the list's
// own relative offsets, (1 + g) * delta bytes further on (the layout the upload gives a list's twins: anyhit_root, traverse.hpp).
__global__ __launch_bounds__(256) void k_lbvh_twin_emit(uint32_t n, LbvhTopo t, const atn_bvh_node* __restrict__ nodes, const uint32_t* __restrict__ offs,
                                                        const uint8_t* __restrict__ flips, uint32_t delta,
                                                        const atn_triangle_param* __restrict__ scene_tris, const float4* __restrict__ scene_vtx,
                                                        float4* __restrict__ image)
{
    const int32_t idx = blockIdx.x * blockDim.x + threadIdx.x, num = (int32_t)n;
    if (idx >= 2 * num - 1) return;
    const uint32_t g = blockIdx.y, shift = (1u + g) * delta;
    auto first_of = [&](int32_t p) { return ((flips[p] >> g) & 1u) ? t.right[p] : t.left[p]; };
    auto second_of = [&](int32_t p) { return ((flips[p] >> g) & 1u) ? t.left[p] : t.right[p]; };
    int32_t next = -1;      // where a walk goes when it is done with this node's subtree
    for (int32_t cur = idx;;) {
        const int32_t p = t.parent[cur];
        if (p < 0) break;
        if (first_of(p) == cur) { next = second_of(p); break; }
        cur = p;
    }
    auto typed = [&](int32_t l) -> int32_t {
        if (l < 0) return kLinkEnd;
        return (int32_t)(offs[l] + shift) | (l >= num - 1 ? kLinkToLeaf : 0);
    };
    const atn_bvh_node nd = nodes[idx];
    float4* q = reinterpret_cast<float4*>(reinterpret_cast<char*>(image) + offs[idx] + shift);
    if (idx < num - 1) {
        q[0] = make_float4(nd.boxmin[0], nd.boxmin[1], nd.boxmin[2], __int_as_float(typed(first_of(idx))));
        q[1] = make_float4(nd.boxmax[0], nd.boxmax[1], nd.boxmax[2], __int_as_float(typed(next)));
    }
    else {
        const int32_t tri = (int32_t)nd.f1;
        const float4 a = scene_vtx[scene_tris[tri].idx[0]], b = scene_vtx[scene_tris[tri].idx[1]], c = scene_vtx[scene_tris[tri].idx[2]];
        q[0] = make_float4(a.x, a.y, a.z, __int_as_float(tri));
        q[1] = make_float4(b.x - a.x, b.y - a.y, b.z - a.z, __int_as_float(typed(next)));
        q[2] = make_float4(c.x - a.x, c.y - a.y, c.z - a.z, 0.0F);
    }
}

} // namespace atn
