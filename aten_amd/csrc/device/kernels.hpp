// Wavefront path-tracing kernels.  One launch sequence per sample:
//
//   gen_path -> [ trace_closest -> shade (+ miss) -> trace_shadow ] x maxDepth -> accumulate_sample
//   ... -> gather (film write)
//
// Path state lives in SoA float4 arrays indexed by the LOCAL path slot (one slot per pixel this
// GPU owns); queues hold slot indices of live paths and are filled with wave64 ballot + popcount
// prefix inside a wave and ONE atomicAdd per 1024-entry block chunk (no scan kernels: the reference's StreamCompaction.cu needs
// ~6 launches per bounce, src/libidaten/kernel/StreamCompaction.cu:175-316).
//
// Semantics are those of the CPU path aten::PathTracing (renderer/pathtracing/pathtracing.cpp:22-236,
// 269-366 and pathtracing_impl.h) -- NOT of the CUDA backend, which re-seeds CMJ per bounce.
#pragma once
#include "shading.hpp"
#include "toon.hpp"

// Three translation units include this file: aten_amd.hip (everything but the regeneration kernels), regen.hip (ATN_REGEN_TU: the
// regeneration kernels and the templates they instantiate) and shade_relaxed.hip (ATN_TEMPLATES_ONLY: k_shade under other compiler
// flags).  Kernels that are not templates belong to exactly one of them.
#if defined(ATN_REGEN_TU) || defined(ATN_TEMPLATES_ONLY)
#define ATN_MAIN_TU 0
#else
#define ATN_MAIN_TU 1
#endif
#ifdef ATN_REGEN_TU
#define ATN_REGEN_KERNELS 1
#else
#define ATN_REGEN_KERNELS 0
#endif

namespace atn {

constexpr uint32_t F_TERMINATED = 1u, F_SINGULAR = 2u, F_HIT = 4u;
constexpr uint32_t kShadowSlotMask = (1u << 26) - 1u;      // shadow-job payload: slot bits (ShadowJob)
constexpr uint32_t kShadowStencilFlag = 0x40000000u;      // in sh_d.w next to the light index: the shaded surface's material is StencilType::ALWAYS
constexpr uint32_t F_LAST_SPECULAR = 8u;     // SVGF shade only: paths.attrib.last_hit_mtrl_idx names a Specular material

// ---- path regeneration (the pool form of the sample loop, PathTracing::run_regen in aten_amd.hip) -------------------------
// The burst's work is a list of ITEMS -- (frame k of the burst, pixel p of this shard), frame-major, pixels in tile order: item =
// k * n_valid + j, p = valid_list[j] -- and a slot of the pool is a CONTAINER: it traces an item's `spp` samples one after the other
// (the accumulation order of the pixel's samples is the serial loop's), and the moment the item's last path ends k_regen_shade hands
// the slot the next item nobody has taken yet (one atomic per 1024-entry chunk).  So every launch of the burst works on a full
// population until the items run out -- not on one that decays bounce by bounce, and not on one in which a pixel whose paths are
// short has run through its frames and idles while the long ones finish (the Cornell box: a third of the pixels see the background).
// Frames of one pixel may be in flight in different slots at once: they meet again in the burst's staging planes (RegenOut::frames),
// which k_regen_end folds into the film in frame order.  What the serial loop keeps in launch arguments becomes per-path state, packed
// into words that travel anyway:
//   ray_d.w : flags (bits 0-4) | bounce (bits 5-12) | sample of the item (bits 13-23) | item, high 8 bits (bits 24-31)
//   thr.w   : CMJ dimension (bits 0-11) | item, low 20 bits (bits 12-31)
// Two things a pool that lives for tens of stages needs, and the serial loop's five bounces do not:
//  * QUEUES THAT STAY IN SLOT ORDER.  The serial loop's append reserves room for a block's entries with one atomic, so the blocks' runs
//    land in the order the blocks finish -- harmless for five bounces after a freshly generated, sorted queue.  In a pool every
//    stage shuffles the runs again and a chunk of the next stage straddles two unrelated runs: after ten stages a wave's 64 entries
//    are 64 unrelated pixels -- state reads uncoalesced, primary rays incoherent: shade +56 %, trace +8 % on sponza_lod after 8 frames,
//    growing with the length of the burst.  So k_regen_shade writes a chunk's entries into the chunk's OWN region (no atomic, a count
//    per chunk) and k_regen_compact squeezes the regions into the next stage's dense queues in order: a stable compaction, and the
//    queue remains the sorted list of live slots for the whole burst.
//  * a frame's finished pixel value goes to a staging plane of the burst (RegenOut::frames), not into the film: the film is a running
//    mean, ordered frame by frame, and a burst that wrote it from its first stage on could not overlap with the burst before it.
//    k_regen_end applies the burst's frames to the film, pixel by pixel in frame order, once the previous burst's k_regen_end is done.
constexpr uint32_t F_PENDING = 16u;         // the pixel's PREVIOUS sample still waits for its last shadow ray: its epilogue runs at the next shade
constexpr uint32_t kRegenFlagMask = 31u;
constexpr uint32_t kRegenBounceShift = 5u, kRegenBounceMask = 255u;
constexpr uint32_t kRegenSampleShift = 13u, kRegenSampleMask = 2047u;
constexpr uint32_t kRegenItemHiShift = 24u;
constexpr uint32_t kRegenDimMask = 4095u, kRegenItemLoShift = 12u, kRegenItemLoMask = (1u << 20) - 1u;
constexpr uint32_t kRegenMaxSpp = 1u << 11, kRegenMaxItems = 1u << 28;
constexpr uint32_t kShadowFinalFlag = 0x20000000u;        // in sh_c.w / sh_d.w: the shadow ray of a path's LAST vertex -- its light goes to `pend`, not `contrib`

struct PathBuffers {
    float4* ray_o;      // org.xyz, pdfb
    float4* ray_d;      // dir.xyz, flags (bit pattern)
    float4* thr;        // throughput.xyz, CMJ dimension counter (bit pattern)
    float4* contrib;    // contrib.xyz, -
    const uint32_t* seeds;  // aten::getRandom(): the CMJ scramble is recomputed from seed + frame + sample instead of being
                            // carried through HBM (32 B of traffic per path and bounce); only the dimension counter is state
    float4* isect;      // instance object id (bit pattern; < 0 = miss), a, b, triangle id (bit pattern): what shade reads of the hit
    float4* sh_o;       // shadow org.xyz, distToLight
    float4* sh_d;       // shadow dir.xyz, target light id (bit pattern)
    float4* sh_c;       // lightcontrib.xyz, -
    float4* accum;      // sum of valid sample contribs.xyz, count
    uint32_t* done;     // pixel stopped sampling (pathtracing.cpp:350-352)
    uint32_t* queue[2]; // live path slots, ping-pong per bounce
    uint32_t* shadow_q; // slots with an active shadow ray this bounce
    uint32_t* q_count;  // [maxDepth + 1] live count entering bounce b
    uint32_t* sh_count; // [maxDepth]
    uint32_t* fetch_closest;    // [maxDepth] dynamic job-fetch cursors of the trace kernels
    uint32_t* fetch_shadow;     // [maxDepth]
    uint32_t* cost;     // [2 * slots] node visits / triangle tests of the pixel's walks this sample (count_stats frames; else null)
    unsigned long long* stats; // [8]: closest rays, shadow rays, hits, closest node visits, closest tri tests, shadow node visits, shadow tri tests
    float4* pend;       // regeneration only: contrib.xyz of the pixel's previous sample while its last shadow ray is in flight (F_PENDING)
    // regeneration only: what k_regen_shade writes for k_regen_compact -- chunk c's surviving / regenerated entries and shadow entries
    // in region c (chunk_size entries wide) of q_regions / sh_regions, their numbers in region_counts[2 * c], [2 * c + 1], and the
    // sums of every kRegenGroup chunks' numbers in group_counts[2 * g], [2 * g + 1] (atomic adds: sums do not depend on their order)
    uint32_t* q_regions;
    uint32_t* sh_regions;
    uint32_t* region_counts;
    uint32_t* group_counts;
    const uint32_t* valid_list;     // regeneration only: [n_valid] the shard's pixel slots that lie inside the frame, ascending (items index it)
    uint32_t* next_item;            // regeneration only: the first item no slot has taken yet
};

struct FrameParams {
    int32_t width, height;
    int32_t n_slots;            // local path slots (n_local_tiles * 64)
    int32_t slot_begin, slot_end;   // the slots this launch works on (one batch of the frame, see PathTracing::render)
    int32_t chunk_items;            // 1..kChunkItems queue entries per thread and chunk in k_shade: 4 keeps the queue atomics
                                    // rare on full frames, fewer give small launches enough blocks to fill the chip
    int32_t tiles_x, tiles_y;
    uint32_t tiles_x_rcp;       // udiv_rcp(tiles_x): slot -> pixel divides by a scalar, without a hoisted vector-register reciprocal
    int32_t rank, world;        // screen-space shard: tile t belongs to rank t % world
    int32_t max_depth, rr_depth;
    int32_t sample;
    uint32_t frame;
    uint32_t n_seeds;
    int32_t break_on_terminate;
    int32_t progressive;
    int32_t burst_frames, spp;  // regeneration only: progressive frames frame .. frame + burst_frames - 1, spp samples each
    uint32_t n_valid, n_valid_rcp, n_items;     // regeneration only: pixels of this shard inside the frame (udiv_rcp of it), items of the burst
};

// slot -> pixel.  Slots are grouped in 8x8 pixel tiles (one tile per wave64): a wave's primary
// rays stay spatially coherent and a tile is the multi-GPU sharding unit.
ATN_DEV bool slot_to_pixel(const FrameParams& fp, uint32_t slot, int32_t& x, int32_t& y)
{
    const uint32_t local_tile = slot >> 6;
    const uint32_t in_tile = slot & 63u;
    const uint32_t tile = local_tile * (uint32_t)fp.world + (uint32_t)fp.rank;
    if (tile >= (uint32_t)(fp.tiles_x * fp.tiles_y)) return false;
    uint32_t tx, ty;
    udivmod(tile, (uint32_t)fp.tiles_x, fp.tiles_x_rcp, ty, tx);
    x = (int32_t)tx * 8 + (int32_t)(in_tile & 7u);
    y = (int32_t)ty * 8 + (int32_t)(in_tile >> 3);
    return x < fp.width && y < fp.height;
}

// Block-aggregated queue append.
//
// A same-address device atomic retires at ~88 per microsecond on MI355X (MI355X_MICROARCH.md,
// row "dequeue"), so one atomicAdd per wave (32 K waves at 1080p) costs ~370 us per queue per
// kernel -- more than the shading itself.  Instead a 256-thread block walks a CHUNK of
// kChunkItems * 256 queue entries, remembers one flag bit per entry, and reserves space for the
// whole chunk with a single atomicAdd per queue: wave ballot + popcount prefix inside the wave,
// LDS for the 4 wave totals.  Output order inside the queue is irrelevant (all per-path state is
// indexed by slot), so nothing is sorted.
constexpr int kChunkItems = 4;
constexpr uint32_t kChunk = 256u * kChunkItems;

struct BlockAppendShared { uint32_t wave_total[2][4]; uint32_t base[2]; };

// flagsA/flagsB: bit k set <=> this thread's k-th entry goes to queue A/B.  entry(k) returns the
// value to append for item k.  Must be called by all 256 threads of the block.
// REGION: the queues are this block's own (a region per chunk, k_regen_shade): entries from index 0, the totals STORED to the counters.
template <class EntryFn, bool REGION = false>
ATN_DEV void block_append2(BlockAppendShared& sh, uint32_t* qA, uint32_t* cntA, uint32_t flagsA,
                           uint32_t* qB, uint32_t* cntB, uint32_t flagsB, EntryFn entry)
{
    const uint32_t tid = here_v(threadIdx.x);
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    uint32_t totA = 0, totB = 0;
#pragma unroll
    for (int k = 0; k < kChunkItems; k++) {
        totA += (uint32_t)__popcll(__ballot((flagsA >> k) & 1u));
        totB += (uint32_t)__popcll(__ballot((flagsB >> k) & 1u));
    }
    if (lane == 0) { sh.wave_total[0][wave] = totA; sh.wave_total[1][wave] = totB; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t a = sh.wave_total[0][0] + sh.wave_total[0][1] + sh.wave_total[0][2] + sh.wave_total[0][3];
        const uint32_t b = sh.wave_total[1][0] + sh.wave_total[1][1] + sh.wave_total[1][2] + sh.wave_total[1][3];
        if constexpr (REGION) { *cntA = a; *cntB = b; sh.base[0] = 0u; sh.base[1] = 0u; }
        else {
        sh.base[0] = a ? atomicAdd(cntA, a) : 0u;
        sh.base[1] = (b && cntB) ? atomicAdd(cntB, b) : 0u;
        }
    }
    __syncthreads();
    uint32_t offA = sh.base[0], offB = sh.base[1];
    for (uint32_t w = 0; w < wave; w++) { offA += sh.wave_total[0][w]; offB += sh.wave_total[1][w]; }
#pragma unroll
    for (int k = 0; k < kChunkItems; k++) {
        const unsigned long long mA = __ballot((flagsA >> k) & 1u);
        const unsigned long long mB = __ballot((flagsB >> k) & 1u);
        if ((flagsA >> k) & 1u) qA[offA + bits_below_lane(mA)] = entry(k);
        if ((flagsB >> k) & 1u) qB[offB + bits_below_lane(mB)] = entry(k);
        offA += (uint32_t)__popcll(mA);
        offB += (uint32_t)__popcll(mB);
    }
    __syncthreads();    // sh is reused by the next chunk
}

ATN_DEV void wave_add_stat(unsigned long long* dst, uint32_t v)
{
    // wave reduction, one atomic per wave
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (__lane_id() == 0 && v) atomicAdd(dst, (unsigned long long)v);
}

// PinholeCamera::sample, camera/pinhole.cpp:97-118
ATN_DEV void pinhole_sample(const atn_camera_param& cam, float s, float t, f3& org, f3& dir)
{
    s = 2.0F * s - 1.0F;
    t = 2.0F * t - 1.0F;
    const f3 cu = mk3(cam.u[0], cam.u[1], cam.u[2]), cv = mk3(cam.v[0], cam.v[1], cam.v[2]);
    const f3 center = mk3(cam.center[0], cam.center[1], cam.center[2]);
    const f3 origin = mk3(cam.origin[0], cam.origin[1], cam.origin[2]);
    f3 pos_on_lens = s * cu + t * cv;
    pos_on_lens = pos_on_lens + center;
    dir = normalize(pos_on_lens - origin);
    org = origin;
}

#if ATN_MAIN_TU
// GeneratePath, renderer/pathtracing/pathtracing_impl.h:65-110
__global__ void __launch_bounds__(256) k_gen_path(PathBuffers pb, FrameParams fp, atn_camera_param cam,
                                                   const uint32_t* __restrict__ seeds)
{
    __shared__ BlockAppendShared sh;
    const uint32_t n = (uint32_t)fp.slot_end;
    for (uint32_t chunk = (uint32_t)fp.slot_begin + blockIdx.x * kChunk; chunk < n; chunk += gridDim.x * kChunk) {
        uint32_t flags = 0;
#pragma unroll 1
        for (int k = 0; k < kChunkItems; k++) {
            const uint32_t slot = chunk + (uint32_t)k * 256u + threadIdx.x;
            int32_t ix = 0, iy = 0;
            bool valid = slot < n && slot_to_pixel(fp, slot, ix, iy);
            if (valid && fp.sample > 0) valid = pb.done[slot] == 0;
            if (!valid) continue;
            const uint32_t idx = (uint32_t)(iy * fp.width + ix);
            const uint32_t rnd = seeds[idx % fp.n_seeds];
            const uint32_t fs = fp.frame + (uint32_t)fp.sample;
            const uint32_t scramble = rnd * 0x1fe3434fu * ((fs + 133u * rnd) / 256u);
            Cmj smp; smp.idx = fs % 256u; smp.dim = 0; smp.scramble = scramble;
            const float r1 = cmj_next(smp);
            const float r2 = cmj_next(smp);
            const float s = ((float)ix + r1) / (float)cam.width;
            const float t = ((float)iy + r2) / (float)cam.height;
            f3 org, dir;
            pinhole_sample(cam, s, t, org, dir);
            pb.ray_o[slot] = make_float4(org.x, org.y, org.z, 1.0F);                    // pdfb = 1
            pb.ray_d[slot] = make_float4(dir.x, dir.y, dir.z, __uint_as_float(0u));     // flags cleared
            pb.thr[slot] = make_float4(1.0F, 1.0F, 1.0F, __uint_as_float(smp.dim));
            pb.contrib[slot] = make_float4(0.0F, 0.0F, 0.0F, 0.0F);
            if (fp.sample == 0) { pb.accum[slot] = make_float4(0, 0, 0, 0); pb.done[slot] = 0; }
            flags |= 1u << k;
        }
        block_append2(sh, pb.queue[0], &pb.q_count[0], flags, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u,
                      [&](int k) { return chunk + (uint32_t)k * 256u + threadIdx.x; });
    }
}

#endif  // ATN_MAIN_TU

// ---- path regeneration: the pieces k_regen_begin / k_regen_shade / k_regen_end share ---------------------------------------
struct RegenState { float4 o, d, t; };     // what a slot's ray_o / ray_d / thr hold
// item -> (frame of the burst, pixel slot of the shard)
ATN_DEV void regen_item(const PathBuffers& pb, const FrameParams& fp, uint32_t item, uint32_t& frame_k, uint32_t& pixel_slot)
{
    uint32_t j;
    udivmod(item, fp.n_valid, fp.n_valid_rcp, frame_k, j);
    pixel_slot = pb.valid_list[j];
}
ATN_DEV uint32_t regen_pack_d(uint32_t flags, uint32_t bounce, uint32_t sample_s, uint32_t item)
{
    return flags | (bounce << kRegenBounceShift) | (sample_s << kRegenSampleShift) | ((item >> 20) << kRegenItemHiShift);
}
ATN_DEV uint32_t regen_pack_t(uint32_t dim, uint32_t item) { return dim | ((item & kRegenItemLoMask) << kRegenItemLoShift); }
// GeneratePath (pathtracing_impl.h:65-110) for sample `sample_s` of item `item`, sampler frame fs = frame + k + sample: the same
// operations on the same operands as k_gen_path, so the same bits.
ATN_DEV RegenState regen_primary_state(const PathBuffers& pb, const FrameParams& fp, const atn_camera_param& cam, uint32_t item, uint32_t sample_s, uint32_t flags)
{
    uint32_t frame_k, p;
    regen_item(pb, fp, item, frame_k, p);
    int32_t ix = 0, iy = 0;
    slot_to_pixel(fp, p, ix, iy);
    const uint32_t idx = (uint32_t)(iy * fp.width + ix);
    const uint32_t rnd = pb.seeds[idx % fp.n_seeds];
    const uint32_t fs = fp.frame + frame_k + sample_s;
    const uint32_t scramble = rnd * 0x1fe3434fu * ((fs + 133u * rnd) / 256u);
    Cmj smp; smp.idx = fs % 256u; smp.dim = 0; smp.scramble = scramble;
    const float r1 = cmj_next(smp);
    const float r2 = cmj_next(smp);
    const float s = ((float)ix + r1) / (float)cam.width;
    const float t = ((float)iy + r2) / (float)cam.height;
    f3 org, dir;
    pinhole_sample(cam, s, t, org, dir);
    RegenState st;
    st.o = make_float4(org.x, org.y, org.z, 1.0F);                    // pdfb = 1
    st.d = make_float4(dir.x, dir.y, dir.z, __uint_as_float(regen_pack_d(flags, 0u, sample_s, item)));
    st.t = make_float4(1.0F, 1.0F, 1.0F, __uint_as_float(regen_pack_t(smp.dim, item)));
    return st;
}

struct RegenOut {
    float4* frames;     // [burst_frames][n_slots]: the pixel value (col / cnt, 1) every frame of the burst hands to Film::put, by PIXEL slot
    float4* film;       // full-frame vec4[w*h] (k_regen_end only)
    float4* tile_out;   // this GPU's pixels in slot order (k_regen_end only; may be null)
};
constexpr uint32_t kRegenGroup = 64u;       // chunks per group of k_regen_compact's two-level prefix sum

// One sample's epilogue -- OnRender's inner loop, pathtracing.cpp:339-352 = k_accumulate_sample -- and, when it was the item's last
// sample, the value k_gather hands to Film::put / FilmProgressive::put (film.cpp:33-45,61-71), stored for k_regen_end.
// `c`: the sample's contribution; `sample_s`: its index in the frame (0 starts the sum: the serial loop clears accum in k_gen_path).
// The samples of an item pass through ONE slot one after the other: the operations and their order are the serial loop's.
ATN_DEV bool regen_invalid_color(const f3& c)     // Renderer::isInvalidColor, renderer.h:58-68
{
    return isnan(c.x) || isinf(c.x) || isnan(c.y) || isinf(c.y) || isnan(c.z) || isinf(c.z) || c.x < 0 || c.y < 0 || c.z < 0;
}
ATN_DEV void regen_epilogue(const PathBuffers& pb, const FrameParams& fp, const RegenOut& ro, uint32_t slot,
                            const f3& c, uint32_t sample_s, uint32_t item, bool item_last)
{
    float4 a = make_float4(0.0F, 0.0F, 0.0F, 0.0F);
    if (sample_s != 0u) a = pb.accum[slot];
    const bool invalid = regen_invalid_color(c);
    if (!invalid) { a.x += c.x; a.y += c.y; a.z += c.z; a.w += 1.0F; }
    if (!item_last) { pb.accum[slot] = a; return; }
    const float cnt = a.w;
    uint32_t frame_k, p;
    regen_item(pb, fp, item, frame_k, p);
    ro.frames[(size_t)frame_k * (uint32_t)fp.n_slots + p] = make_float4(a.x / cnt, a.y / cnt, a.z / cnt, 1.0F);
}

#if ATN_REGEN_KERNELS
// The pool's first population: slot j takes item j (sample 0), for the n_slots_pool = min(n_valid, n_items) slots of the pool; written
// as regions for k_regen_compact(0) like a shade launch's output ("stage -1": q_count[-1] = the slots it worked on).
__global__ void __launch_bounds__(256) k_regen_begin(PathBuffers pb, FrameParams fp, atn_camera_param cam)
{
    __shared__ BlockAppendShared sh;
    const uint32_t n = fp.n_valid < fp.n_items ? fp.n_valid : fp.n_items;
    const int items = fp.chunk_items;
    const uint32_t chunk_size = 256u * (uint32_t)items;
    if (blockIdx.x == 0u && threadIdx.x == 0u) { pb.q_count[-1] = n; *pb.next_item = n; }
    for (uint32_t chunk = blockIdx.x * chunk_size; chunk < n; chunk += gridDim.x * chunk_size) {
        uint32_t flags = 0;
#pragma unroll 1
        for (int k = 0; k < items; k++) {
            const uint32_t slot = chunk + (uint32_t)k * 256u + threadIdx.x;
            if (slot >= n) continue;
            const RegenState st = regen_primary_state(pb, fp, cam, slot, 0u, 0u);
            pb.ray_o[slot] = st.o; pb.ray_d[slot] = st.d; pb.thr[slot] = st.t;
            pb.contrib[slot] = make_float4(0.0F, 0.0F, 0.0F, 0.0F);
            flags |= 1u << k;
        }
        auto entry_of = [&](int k) { return chunk + (uint32_t)k * 256u + threadIdx.x; };
        const uint32_t c = chunk / chunk_size;
        block_append2<decltype(entry_of), true>(sh, pb.q_regions + chunk, &pb.region_counts[2u * c], flags,
                                                pb.sh_regions + chunk, &pb.region_counts[2u * c + 1u], 0u, entry_of);
        if (threadIdx.x == 0u) { const uint32_t v = pb.region_counts[2u * c]; if (v) atomicAdd(&pb.group_counts[2u * (c / kRegenGroup)], v); }
    }
}

// The stable compaction between two stages: region c of q_regions / sh_regions (what chunk c of shade(stage - 1) kept or regenerated,
// and its shadow rays) goes to the dense queues at the sum of the counts before it -- groups of kRegenGroup chunks first, then the
// chunks of its own group -- so the next stage's queues list their slots in the order of this stage's.  Block c = region c; the
// last valid region also writes the totals (q_count[stage], sh_count[stage - 1]); block 0 clears the group sums of the other parity
// for shade(stage).
__global__ void __launch_bounds__(256) k_regen_compact(PathBuffers pb, int32_t stage, uint32_t chunk_size, uint32_t* group_counts_next, uint32_t n_groups)
{
    __shared__ uint32_t red[2][4];
    const uint32_t n_prev = pb.q_count[stage - 1];
    const uint32_t n_chunks = (n_prev + chunk_size - 1u) / chunk_size;      // chunks shade(stage - 1) worked on
    const uint32_t c = blockIdx.x;
    if (c == 0u) { for (uint32_t g = threadIdx.x; g < 2u * n_groups; g += 256u) group_counts_next[g] = 0u; }
    if (c >= n_chunks) return;
    const uint32_t g0 = c / kRegenGroup;
    uint32_t sq = 0, ss = 0;
    for (uint32_t g = threadIdx.x; g < g0; g += 256u) { sq += pb.group_counts[2u * g]; ss += pb.group_counts[2u * g + 1u]; }
    for (uint32_t r = g0 * kRegenGroup + threadIdx.x; r < c; r += 256u) { sq += pb.region_counts[2u * r]; ss += pb.region_counts[2u * r + 1u]; }
    for (int off = 32; off > 0; off >>= 1) { sq += __shfl_down(sq, off); ss += __shfl_down(ss, off); }
    if ((threadIdx.x & 63u) == 0u) { red[0][threadIdx.x >> 6] = sq; red[1][threadIdx.x >> 6] = ss; }
    __syncthreads();
    const uint32_t base_q = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const uint32_t base_s = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const uint32_t nq = pb.region_counts[2u * c], ns = pb.region_counts[2u * c + 1u];
    uint32_t* __restrict__ qd = pb.queue[stage & 1];
    for (uint32_t e = threadIdx.x; e < nq; e += 256u) qd[base_q + e] = pb.q_regions[(size_t)c * chunk_size + e];
    for (uint32_t e = threadIdx.x; e < ns; e += 256u) pb.shadow_q[base_s + e] = pb.sh_regions[(size_t)c * chunk_size + e];
    if (c == n_chunks - 1u && threadIdx.x == 0u) { pb.q_count[stage] = base_q + nq; pb.sh_count[stage - 1] = base_s + ns; }
}

// After the last stage, per SLOT: the epilogue of an item whose last sample ended with a shadow ray in flight and whose slot found no
// further item to take (the F_PENDING marker k_regen_shade leaves in a retired slot).
__global__ void __launch_bounds__(256) k_regen_flush(PathBuffers pb, FrameParams fp, RegenOut ro)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= (fp.n_valid < fp.n_items ? fp.n_valid : fp.n_items)) return;
    const uint32_t w = __float_as_uint(pb.ray_d[slot].w);
    if (w & F_PENDING) {
        const float4 pd = pb.pend[slot];
        regen_epilogue(pb, fp, ro, slot, mk3(pd), (uint32_t)fp.spp - 1u, __float_as_uint(pd.w), true);
    }
}

// Then, per PIXEL slot: the burst's frames into the film -- Film::put / FilmProgressive::put (film.cpp:33-45,61-71) = k_gather, frame
// after frame for every pixel -- and the zero k_gather writes into the tile buffer for slots outside the frame.
__global__ void __launch_bounds__(256) k_regen_end(PathBuffers pb, FrameParams fp, RegenOut ro)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= (uint32_t)fp.n_slots) return;
    int32_t x, y;
    if (!slot_to_pixel(fp, slot, x, y)) {
        if (ro.tile_out) ro.tile_out[slot] = make_float4(0.0F, 0.0F, 0.0F, 0.0F);
        return;
    }
    const uint32_t pixel = (uint32_t)(y * fp.width + x);
    float4 out = make_float4(0.0F, 0.0F, 0.0F, 0.0F);
    if (fp.progressive) out = ro.film[pixel];
    for (int32_t k = 0; k < fp.burst_frames; k++) {
        const float4 v = ro.frames[(size_t)k * (uint32_t)fp.n_slots + slot];
        if (fp.progressive) {
            const float4 cur = out;
            const float n = (float)((int32_t)cur.w);
            const float d = n + 1;
            out = make_float4((n * cur.x + v.x) / d, (n * cur.y + v.y) / d, (n * cur.z + v.z) / d, n + 1);
        }
        else {
            out = v;
        }
    }
    ro.film[pixel] = out;
    if (ro.tile_out) ro.tile_out[slot] = out;
}
#endif  // ATN_REGEN_KERNELS

// REFILL selects the persistent, lane-refilling walk (large trees) or the plain walk (small trees and small launches,
// where the refill bookkeeping costs more than the idle lanes it removes).
#ifdef ATN_TRACE_WPE
#define ATN_TRACE_ATTR __attribute__((amdgpu_waves_per_eu(ATN_TRACE_WPE, ATN_TRACE_WPE)))
#else
#define ATN_TRACE_ATTR
#endif

template <bool COUNT, bool REFILL, class Job, bool LDSN = false>
ATN_DEV void trace_dispatch(const DevScene& sc, uint32_t count, uint32_t* fetch_counter, const Job& job, TravCounters* tc)
{
    if constexpr (REFILL) {
        __shared__ TraceShared sh;
        if (LDSN) lds_scene_copy(sc);       // the whole node image + the matrices (small scenes)
        trace_refill<COUNT, Job, LDSN>(sc, sh, count, fetch_counter, job, tc);
    }
    else {
        trace_simple<COUNT, Job, LDSN>(sc, count, job, tc);
    }
}

struct ClosestJob {
    PathBuffers pb;
    const uint32_t* __restrict__ q;
    float t_min;
    ATN_DEV void fetch(uint32_t j, float4& a, float4& b, float& stop_t) const { fetch_slot(q[j], a, b, stop_t); }
    ATN_DEV void fetch_slot(uint32_t slot, float4& a, float4& b, float& stop_t) const
    {
        const float4 ro = pb.ray_o[slot], rd = pb.ray_d[slot];
        stop_t = -kInf;
        a = make_float4(ro.x, ro.y, ro.z, kInf);
        b = make_float4(rd.x, rd.y, rd.z, __uint_as_float(slot));
    }
    ATN_DEV bool finish(uint32_t slot, const Hit& h, bool, float4&, float4&, float&) const
    {
        pb.isect[slot] = make_float4(__int_as_float(h.objid), h.a, h.b, __int_as_float(h.tri));
        return false;
    }
    ATN_DEV void cost(uint32_t slot, uint32_t nodes, uint32_t tris) const
    {
        if (pb.cost) { atomicAdd(&pb.cost[2u * slot], nodes); atomicAdd(&pb.cost[2u * slot + 1u], tris); }
    }
};

template <bool COUNT, bool REFILL>
__global__ void __launch_bounds__(kTraceBlock > 256 ? kTraceBlock : 256) k_trace_closest(PathBuffers pb, DevScene sc, int32_t bounce)
{
    const uint32_t count = pb.q_count[bounce];
    const ClosestJob job{ pb, pb.queue[bounce & 1], kEps };
    TravCounters tc{};
    trace_dispatch<COUNT, REFILL>(sc, count, &pb.fetch_closest[bounce], job, &tc);
    if (COUNT) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&pb.stats[0], (unsigned long long)count);
        wave_add_stat(&pb.stats[3], tc.nodes);
        wave_add_stat(&pb.stats[4], tc.tris);
    }
}

// AOV outputs of the SVGF flavour of shade (SVGFRenderer::Shade, src/libaten/renderer/svgf/svgf.cpp:89-229):
// full-frame buffers indexed by pixel, and the fourth row of mtx_W2C (clip w = view depth).
struct SvgfShade {
    float4* nd;         // normal.xyz, clip-space w
    float4* am;         // albedo texel.rgb, material id (svgf.cpp:131)
    float4* primary;    // bounce-0 hit position, w = 1 (0 on a miss)
    float w2c3[4];
};

// PathTracing::shade (pathtracing.cpp:91-236) + ShadeMiss (pathtracing_impl.h:112-176) for one path.
// Returns updated flags; fills the next ray and the shadow ray.
// SVGF = true is SVGFRenderer::Shade + ShadeMiss with AOV spans: AOVs at bounce 0 (and at bounce 1 behind a Specular
// hit), albedo read with default (1,1,1,1) and demodulated where the AOV took it.
// MS: the material set of the scene (DevScene::material_set, shading.hpp): BSDFs outside it are compiled out.
#ifdef ATN_SHADE_WPE
#define ATN_SHADE_ATTR __attribute__((amdgpu_waves_per_eu(ATN_SHADE_WPE, ATN_SHADE_WPE)))
#else
#define ATN_SHADE_ATTR
#endif
// Hits and misses of a chunk are shaded by different waves (ATN_SHADE_PARTITION): a block's 256 x items queue entries are
// partitioned -- stable, hits first -- through an LDS permutation before they are shaded.  A wave that mixes the two
// runs the long hit path with the miss lanes idle (lane utilisation of k_shade: 0.87 on sponza_lod, 0.51 on the
// open-sky atrium); partitioned, all but one wave of a chunk are homogeneous.  Per-pixel results do not depend on the
// order in which a block shades its entries.
#ifndef ATN_SHADE_PARTITION
#define ATN_SHADE_PARTITION 1
#endif
struct ShadePartShared { uint32_t perm[kChunk]; uint32_t wcount[kChunkItems][4][2]; };
// REGEN: what brings a chunk's results back into the order of its queue entries (inv: where in the chunk an entry of the permutation
// came from; oflag: per queue entry, bit 0 = goes on to the next stage, bit 1 = has a shadow ray).  The hits-first permutation is for
// the shading only: a queue written in permuted order is scrambled a little more by every stage, and after some tens of stages a
// wave's 64 entries are 64 slots from all over the chunk's neighbourhood -- every state read one cache line per lane.
template <bool ON> struct ShadeOrderShared { uint16_t inv[ON ? kChunk : 1]; uint8_t oflag[ON ? kChunk : 1]; };

// REGEN: the path-regeneration flavour (k_regen_shade).  `bounce_arg` is then the STAGE of the pool -- it selects queues and counters --
// and a path's own bounce, sample and frame come out of its state words; a path that ends runs its sample epilogue here and the
// pixel's next primary ray takes its place in the next stage's queue.
template <bool SVGF, int MS, bool REGEN = false>
ATN_DEV void shade_body(const PathBuffers& pb, const DevScene& sc, const FrameParams& fp, const atn_camera_param& cam, int32_t bounce_arg, const SvgfShade& sv,
                        const RegenOut& ro = RegenOut{})
{
    __shared__ BlockAppendShared sh;
#if ATN_SHADE_PARTITION
    __shared__ ShadePartShared part;
    __shared__ ShadeOrderShared<REGEN> order[1];
#endif
    const uint32_t count = pb.q_count[bounce_arg];
    const uint32_t* __restrict__ q = pb.queue[bounce_arg & 1];
    uint32_t* __restrict__ qn = pb.queue[(bounce_arg + 1) & 1];
#if !ATN_SHADE_PARTITION
    uint32_t nhits = 0;
#endif

    const int items = fp.chunk_items;
    const uint32_t chunk_size = 256u * (uint32_t)items;
    for (uint32_t chunk = blockIdx.x * chunk_size; chunk < count; chunk += gridDim.x * chunk_size) {
      uint32_t push_bits = 0;     // bit k: item k goes on to the next bounce; bit 16 + k: it has a shadow ray (one register, not two)
#if ATN_SHADE_PARTITION
      const uint32_t n_valid = count - chunk < chunk_size ? count - chunk : chunk_size;
      {
          const uint32_t tid = here_v(threadIdx.x);      // (the LDS addresses below are computed per chunk, not carried across it)
          const uint32_t lane = tid & 63u, wave = tid >> 6;
          uint32_t hitmask = 0;
          __syncthreads();            // the previous chunk's append has read `part.perm`
#pragma unroll 1
          for (int k = 0; k < items; k++) {
              const uint32_t j = chunk + (uint32_t)k * 256u + threadIdx.x;
              const bool valid = j < count;
              const bool hit = valid && __float_as_int(pb.isect[q[j]].x) >= 0;
              if (hit) hitmask |= 1u << k;
              const unsigned long long bh = __ballot(hit), bm = __ballot(valid && !hit);
              if (lane == 0) { part.wcount[k][wave][0] = (uint32_t)__popcll(bh); part.wcount[k][wave][1] = (uint32_t)__popcll(bm); }
          }
          if (pb.stats) wave_add_stat(&pb.stats[2], (uint32_t)__popc(hitmask));   // (counted frames only; per chunk, so that no counter lives across the shading)
          __syncthreads();
          uint32_t total_hits = 0;
          for (int k = 0; k < items; k++) for (uint32_t w = 0; w < 4; w++) total_hits += part.wcount[k][w][0];
          uint32_t before_h = 0, before_m = 0;
#pragma unroll 1
          for (int k = 0; k < items; k++) {
              const uint32_t j = chunk + (uint32_t)k * 256u + threadIdx.x;
              const bool valid = j < count;
              const bool hit = (hitmask >> k) & 1u;
              const unsigned long long bh = __ballot(hit), bm = __ballot(valid && !hit);
              uint32_t bh_w = before_h, bm_w = before_m;
              for (uint32_t w = 0; w < 4; w++) {
                  if (w < wave) { bh_w += part.wcount[k][w][0]; bm_w += part.wcount[k][w][1]; }
                  before_h += part.wcount[k][w][0]; before_m += part.wcount[k][w][1];
              }
              if (valid) {
                  const uint32_t pos = hit ? bh_w + bits_below_lane(bh) : total_hits + bm_w + bits_below_lane(bm);
                  part.perm[pos] = q[j];
                  if constexpr (REGEN) order[0].inv[pos] = (uint16_t)(j - chunk);
              }
          }
          __syncthreads();
      }
#endif
#pragma unroll 1
      for (int k = 0; k < items; k++) {
#if ATN_SHADE_PARTITION
        const uint32_t e = (uint32_t)k * 256u + threadIdx.x;
        const bool valid = e < n_valid;
#else
        const uint32_t j = chunk + (uint32_t)k * 256u + threadIdx.x;
        const bool valid = j < count;
#endif
        bool push_next = false, push_shadow = false;
        uint32_t rg_need_bits = 0u;     // REGEN: bit 2 = the slot needs a new item, bit 3 = ... and its last path's epilogue is pending
        uint32_t slot = 0;

        if (valid) {
#if ATN_SHADE_PARTITION
            slot = part.perm[e];
#else
            slot = q[j];
#endif
            const float4 ro4 = pb.ray_o[slot], rd4 = pb.ray_d[slot];
            const f3 ray_org = mk3(ro4), ray_dir = mk3(rd4);
            float pdfb = ro4.w;
            uint32_t flags = __float_as_uint(rd4.w);
            uint32_t rg_sample = 0u, rg_item = 0u;      // REGEN: the path's sample of its item, the item (frame of the burst, pixel)
            int32_t bounce = bounce_arg;
            if constexpr (REGEN) {
                rg_sample = (flags >> kRegenSampleShift) & kRegenSampleMask;
                rg_item = (flags >> kRegenItemHiShift) << 20;
                bounce = (int32_t)((flags >> kRegenBounceShift) & kRegenBounceMask);
                flags &= kRegenFlagMask;
            }
            const float4 is4 = pb.isect[slot];
            const int32_t hit_objid = __float_as_int(is4.x);
            const float4 thr4 = pb.thr[slot];
            f3 throughput = mk3(thr4);
            bool wrote_ray = false, thr_stored = false;
            f3 contrib_add = mk3(0.0F);         // contrib is read-modify-written only by the paths that add to it
            bool contrib_changed = false;
            // sampler state: GeneratePath's scramble (pathtracing_impl.h:75-81) from the pixel's seed
            uint4 s4;
            uint32_t pixel_slot = slot;         // the slot of the serial layout that IS this path's pixel (REGEN: the item's)
            {
                uint32_t fs = fp.frame + (uint32_t)fp.sample;
                s4.y = __float_as_uint(thr4.w);
                if constexpr (REGEN) {
                    rg_item |= s4.y >> kRegenItemLoShift;
                    s4.y &= kRegenDimMask;
                    uint32_t rg_frame;
                    regen_item(pb, fp, rg_item, rg_frame, pixel_slot);
                    fs = fp.frame + rg_frame + rg_sample;
                }
                int32_t px = 0, py = 0;
                slot_to_pixel(fp, pixel_slot, px, py);
                s4.w = (uint32_t)(py * fp.width + px);
                const uint32_t rnd = pb.seeds[s4.w < fp.n_seeds ? s4.w : s4.w % here(fp.n_seeds)];    // (one seed per pixel is the rule)
                s4.x = fs % 256u;
                s4.z = rnd * 0x1fe3434fu * ((fs + 133u * rnd) / 256u);
            }
            Cmj smp; smp.idx = s4.x; smp.dim = s4.y; smp.scramble = s4.z;
            if constexpr (REGEN) {
                // the pixel's previous sample ended with a shadow ray in flight: that ray has been traced since (its light is in
                // `pend`), the sample's epilogue runs now, before anything of this sample can reach accum or the film
                if (flags & F_PENDING) {
                    flags &= ~F_PENDING;
                    const bool prev_last = rg_sample == 0u;     // (such a sample was not terminated: no break, the next index is the next sample)
                    const float4 pd = pb.pend[slot];            // (w: the item it belongs to -- another one when it was that item's last sample)
                    regen_epilogue(pb, fp, ro, slot, mk3(pd), prev_last ? (uint32_t)fp.spp - 1u : rg_sample - 1u, __float_as_uint(pd.w), prev_last);
                }
            }
            // REGEN: where the path ends.  The sample epilogue -- at once, or handed over (`pend`, F_PENDING) when the path ran out of
            // depth without being terminated: the shadow ray of its last vertex, if NEE casts one, is yet to be traced.  Then either the
            // item's next sample (its state is returned: the caller stores it with the continued paths' state, all lanes together), or,
            // when that was the item's last sample, a NEW ITEM for the slot -- taken after the chunk's entries are all shaded, one atomic
            // for the whole chunk (need_item; the slot's state is written there).
            bool rg_stored = false;
            uint32_t rg_end = 0u;       // how the path ended: 0 = it did not; 1 = the item goes on (next sample: state stored); 4 / 12 = the slot needs a new item (12: with a pending epilogue)
            auto regen_end = [&](bool out_of_depth) -> uint32_t {
                f3 ctot = mk3(pb.contrib[slot]);
                if (contrib_changed) ctot = ctot + contrib_add;
                const bool pending = out_of_depth;      // (not terminated)
                // pathtracing.cpp:339-352: an INVALID sample `continue`s past the break -- a terminated path whose colour is NaN / negative does
                // not stop the pixel's sample loop (k_accumulate_sample returns before it sets `done`)
                const bool item_last = rg_sample + 1u >= (uint32_t)fp.spp || (fp.break_on_terminate && !out_of_depth && !regen_invalid_color(ctot));
                if (pending) pb.pend[slot] = make_float4(ctot.x, ctot.y, ctot.z, __uint_as_float(rg_item));
                else regen_epilogue(pb, fp, ro, slot, ctot, rg_sample, rg_item, item_last);
                if (item_last) return pending ? 12u : 4u;
                const RegenState st = regen_primary_state(pb, fp, cam, rg_item, rg_sample + 1u, pending ? F_PENDING : 0u);
                pb.ray_o[slot] = st.o; pb.ray_d[slot] = st.d; pb.thr[slot] = st.t;
                pb.contrib[slot] = make_float4(0.0F, 0.0F, 0.0F, 0.0F);
                return 1u;
            };

            flags &= ~F_HIT;
            const bool is_hit = hit_objid >= 0;

            if (!is_hit) {
                // ---------------- ShadeMiss
                if (!(flags & F_TERMINATED)) {
                    f3 dir = ray_dir;
                    if (bounce == 0) {
                        int32_t ix = 0, iy = 0;
                        slot_to_pixel(fp, pixel_slot, ix, iy);
                        const float s = (float)ix / (float)here(fp.width);
                        const float t = (float)iy / (float)here(fp.height);
                        f3 o;
                        pinhole_sample(cam, s, t, o, dir);
                    }
                    const float4 emit = background_sample(sc, dir);
                    float misW = 1.0f;
                    if (SVGF && (bounce == 0 || (bounce == 1 && (flags & F_SINGULAR)))) {
                        // FillBasicAOVsIfHitMiss (renderer/aov.h:183-198)
                        sv.nd[s4.w] = make_float4(0.0F, 0.0F, 0.0F, -1.0F);
                        sv.am[s4.w] = make_float4(emit.x, emit.y, emit.z, -1.0F);
                        if (bounce == 0) sv.primary[s4.w] = make_float4(0.0F, 0.0F, 0.0F, 0.0F);
                    }
                    if (!(bounce == 0 || (bounce == 1 && (flags & F_SINGULAR)))) {
                        // ImageBasedLight::samplePdf, light/ibl.h:46-58
                        float pdfLight = luminance(emit.x, emit.y, emit.z) / sc.avgIllum;
                        pdfLight /= (2.0f * kPi);
                        if (sc.ibl_importance) pdfLight = ibl_direction_pdf(sc, dir);   // optional table sampler: its own density
                        misW = pdfb / (pdfLight + pdfb);
                    }
                    f3 c = 1.0F * mk3(mul4(misW, emit)) + mk3(0.0F);   // ApplyAlphaBlend (transmission 1, throughput 0)
                    c = c * throughput;
                    contrib_add = c; contrib_changed = true;
                    flags |= F_TERMINATED;
                }
            }
            else {
                flags |= F_HIT;
#if !ATN_SHADE_PARTITION
                nhits++;
#endif
                // ---------------- shade
                const int32_t tri_id = __float_as_int(is4.w);
                HitRec rec;
                evaluate_hit(rec, sc, hit_objid, tri_id, is4.y, is4.z);
                const int32_t mtrlid = triangle_mtrlid(sc, tri_id);

                const bool isBackfacing = dot(rec.normal, -ray_dir) < 0.0F;
                f3 orienting_normal = rec.normal;

                // FillMaterial (material_impl.h:232-262): a negative id selects the white-diffuse fallback, which
                // the upload appends after the last real material
                const DevMaterial* mp = &sc.materials[mtrlid >= 0 ? mtrlid : sc.n_materials];
                DevMaterial toon_base;      // material set 3: a toon surface deeper in the path is its base material
                bool toon_first_hit = false;
                if (MS >= kMsToon && (mp->type == ATN_MTRL_TOON || mp->type == ATN_MTRL_STYLIZED_BRDF)) {
                    // PathTracing::shade, pathtracing.cpp:160-184.  toon_type is Diffuse or Specular, so the reference's
                    // `is_singular = (toon_type == ToonSpecular)` is always false: an ideal mirror that NEE treats as non-singular
                    toon_first_hit = bounce == 0;
                    toon_base = *mp;
                    const int32_t tt = sc.toon[mtrlid >= 0 ? mtrlid : sc.n_materials].toon_type;
                    toon_base.type = tt == ATN_MTRL_DIFFUSE ? ATN_MTRL_DIFFUSE : ATN_MTRL_SPECULAR;
                    toon_base.attrib = (toon_base.attrib & ~(uint32_t)ATN_MTRL_ATTR_SINGULAR) | (tt == ATN_MTRL_TOON_SPECULAR ? ATN_MTRL_ATTR_SINGULAR : 0u);
                    if (!toon_first_hit) mp = &toon_base;
                }
                const DevMaterial& m = *mp;
                float4 albedo4;
                if (SVGF) {
                    int32_t albedo_map = m.albedoMap;
                    if (bounce == 0 || (bounce == 1 && (flags & F_LAST_SPECULAR))) {
                        // FillBasicAOVs (renderer/aov.h:158-181): the normal BEFORE back-face flip and normal map
                        const float4 texcolor = sample_texture(sc, albedo_map, rec.u, rec.v, make_float4(1.0F, 1.0F, 1.0F, 1.0F));
                        const float depth = sv.w2c3[0] * rec.p.x + sv.w2c3[1] * rec.p.y + sv.w2c3[2] * rec.p.z + sv.w2c3[3] * 1.0F;
                        sv.nd[s4.w] = make_float4(rec.normal.x, rec.normal.y, rec.normal.z, depth);
                        sv.am[s4.w] = make_float4(texcolor.x, texcolor.y, texcolor.z, (float)mtrlid);
                        albedo_map = -1;        // "for exporting separated albedo"
                        if (bounce == 0) sv.primary[s4.w] = make_float4(rec.p.x, rec.p.y, rec.p.z, 1.0F);
                    }
                    albedo4 = sample_texture(sc, albedo_map, rec.u, rec.v, make_float4(1.0F, 1.0F, 1.0F, 1.0F));
                }
                else {
                    albedo4 = sample_texture(sc, m.albedoMap, rec.u, rec.v, m.baseColor);
                    albedo4 = add4(mul4(1.0F, albedo4), make_float4(0, 0, 0, 0));
                }
                const f3 albedo = mk3(albedo4);

                bool shaded_out = false;
                if (MS >= kMsToon && toon_first_hit) {
                    // HitTeminatedMaterial, pathtracing_impl.h:482-503: "treat toon as a light" -- the stylised colour is the
                    // path's contribution and the path ends (what PathTracing::shade still computes after it --
                    // pathtracing.cpp:174-184 -- is never observed: HitShadowRay and the next bounce skip terminated paths)
                    int32_t px = 0, py = 0;
                    slot_to_pixel(fp, pixel_slot, px, py);
                    const f3 toon = toon_bsdf(sc, m, mtrlid >= 0 ? mtrlid : sc.n_materials, smp, rec.p, rec.normal, ray_dir, rec.u, rec.v, px, py);
                    contrib_add = (throughput * toon) * albedo; contrib_changed = true;
                    flags |= F_TERMINATED;
                    shaded_out = true;
                }
                // HitTeminatedMaterial -> HitImplicitLight, pathtracing_impl.h:395-509
                if (m.type == ATN_MTRL_EMISSIVE && (m.attrib & ATN_MTRL_ATTR_EMISSIVE) && !isBackfacing) {
                    // an emissive surface that is not registered as a light (light_id < 0) has no LightParameter to
                    // read: the reference indexes lights[-1] there; here it emits nothing
                    const int32_t lid = sc.objects[hit_objid].light_id;
                    const f3 light_color = (lid >= 0 && lid < sc.n_lights) ? area_light_color(sc.lights[lid], rec.area) : mk3(0.0F);
                    float weight = 1.0f;
                    if (bounce > 0) {
                        const float cosLight = dot(rec.normal, -ray_dir);
                        const f3 dv = rec.p - ray_org;
                        const float dist2 = dot(dv, dv);
                        if (cosLight >= 0) {
                            float pdfLight = 1 / rec.area;
                            pdfLight = (pdfLight * dist2) / cosLight;
                            weight = pdfb / (pdfb + pdfLight);
                        }
                    }
                    contrib_add = (throughput * weight) * light_color; contrib_changed = true;
                    flags |= F_TERMINATED;
                    shaded_out = true;
                }

                if (!shaded_out) {
                    if (!(m.attrib & ATN_MTRL_ATTR_TRANSLUCENT) && isBackfacing) orienting_normal = -orienting_normal;
                    // material::applyNormal: the normal map -- or, for CarPaint, the flake normal and the random number it shares
                    const int32_t mtrl_slot = mtrlid >= 0 ? mtrlid : sc.n_materials;
                    const float pre_r = apply_normal<MS>(sc, m, mtrl_slot, orienting_normal, rec.u, rec.v, ray_dir, smp);

                    // ---- FillShadowRay / SampleLight, pathtracing_impl.h:178-264
                    // The reference evaluates NEE HERE, before Russian roulette and the BSDF sample, and HitShadowRay later drops
                    // it when the path was terminated in this bounce.  Only the random DRAWS are taken here, in the reference's
                    // order -- the light pick, then as many dimensions as Light::sample will consume -- and the evaluation runs
                    // at the end of the bounce from the saved sampler position (CMJ draws are pure functions of index,
                    // dimension and scramble): nothing of the shadow ray stays live across the BSDF block (12 registers in a
                    // kernel held to 128), and paths that end in this bounce skip the light sample and the second BSDF
                    // evaluation altogether.  Same operations on the same operands for every surviving path: same bits.
                    const bool invalid_mtrl = (m.attrib & (ATN_MTRL_ATTR_SINGULAR | ATN_MTRL_ATTR_TRANSLUCENT)) != 0;
                    const bool nee = sc.n_lights > 0 && !invalid_mtrl;
                    const uint32_t nee_dim = smp.dim;
                    const f3 thr_in = throughput;
                    // what the BSDF sample, the NEE evaluation and the light sample share at this vertex (shading.hpp, HitPre)
                    HitPre hp;
                    tangent_coordinate(orienting_normal, hp.t, hp.b);
                    hp.rough = ggx_roughness(sc, m, rec.u, rec.v);
                    hp.lambda_v = m.type == ATN_MTRL_GGX ? ggx_lambda(hp.rough, -ray_dir, orienting_normal) : 0.0F;
                    int32_t nee_light = 0;      // the light NEE picked (kept: the draw that picked it is not repeated)
                    if (nee) {
                        // with ONE light the pick is 0 whatever the draw says ((int)(r * 1) with r < 1): the draw is skipped, its
                        // dimension consumed
                        if (here(sc.n_lights) > 1) {
                            nee_light = (int32_t)(cmj_next(smp) * (float)here(sc.n_lights));
                            nee_light = nee_light < sc.n_lights - 1 ? nee_light : sc.n_lights - 1;
                        }
                        else smp.dim++;
                        smp.dim += light_sample_draws(sc.lights[nee_light], sc);
                    }

                    // ---- ComputeRussianProbability, pathtracing_impl.h:680-698
                    float russian_prob = 1.0f;
                    if (bounce > fp.rr_depth) {
                        if (dot(throughput, throughput) > 0) {
                            russian_prob = max3(throughput);
                            const float p = cmj_next(smp);
                            if (p >= russian_prob) flags |= F_TERMINATED; else flags &= ~F_TERMINATED;
                        }
                    }

                    // ---- sampleMaterial + PrepareForNextBounce, pathtracing_impl.h:700-743
                    MtrlSample ms;
                    sample_material<MS>(ms, sc, m, orienting_normal, ray_dir, smp, rec.u, rec.v, mtrl_slot, pre_r, &hp);
                    const f3 next_dir = normalize(ms.dir);
                    const f3 ray_along_normal = dot(orienting_normal, next_dir) >= 0.0f ? orienting_normal : -orienting_normal;
                    const float c = dot(ray_along_normal, next_dir);
                    if (ms.pdf > 0 && c > 0) {
                        throughput = throughput * ((((albedo * ms.bsdf) * c) / ms.pdf));
                        if (russian_prob != 1.0F) throughput = throughput / russian_prob;      // (x / 1 is x, bit for bit: three IEEE divisions on every vertex below the roulette depth)
                    }
                    else {
                        flags |= F_TERMINATED;
                    }
                    // (the path's throughput and sampler position are final here: stored now, not carried across the NEE block)
                    bool nee_final = false;     // REGEN: the vertex is the path's last and NEE's shadow ray adds to `pend`
                    if constexpr (REGEN) {
                        bool cont = false;
                        if (!(flags & F_TERMINATED)) {
                            pdfb = ms.pdf;
                            flags = (m.attrib & ATN_MTRL_ATTR_SINGULAR) ? (flags | F_SINGULAR) : (flags & ~F_SINGULAR);
                            cont = bounce + 1 < fp.max_depth;
                            nee_final = !cont;
                        }
                        if (cont) {
                            const f3 no = ray_offset(rec.p, ray_along_normal);
                            const f3 nd = normalize(next_dir);     // ray(o, d, n) constructor re-normalises (ray.h:17-24)
                            pb.ray_o[slot] = make_float4(no.x, no.y, no.z, pdfb);
                            pb.ray_d[slot] = make_float4(nd.x, nd.y, nd.z, __uint_as_float(regen_pack_d(flags, (uint32_t)(bounce + 1), rg_sample, rg_item)));
                            pb.thr[slot] = make_float4(throughput.x, throughput.y, throughput.z, __uint_as_float(regen_pack_t(smp.dim, rg_item)));
                            push_next = true;
                        }
                        else {
                            rg_end = regen_end(nee_final);
                            contrib_changed = false;
                            push_next = rg_end == 1u;
                        }
                        rg_stored = true; thr_stored = true; wrote_ray = true;
                    }
                    else {
                    pb.thr[slot] = make_float4(throughput.x, throughput.y, throughput.z, __uint_as_float(smp.dim));
                    thr_stored = true;
                    if (!(flags & F_TERMINATED)) {
                        pdfb = ms.pdf;
                        flags = (m.attrib & ATN_MTRL_ATTR_SINGULAR) ? (flags | F_SINGULAR) : (flags & ~F_SINGULAR);
                        if (SVGF) {     // last_hit_mtrl_idx = mtrl.id (pathtracing_impl.h:739), read back at svgf.cpp:142-144
                            const bool last_spec = m.id >= 0 && m.id < sc.n_materials && sc.materials[m.id].type == ATN_MTRL_SPECULAR;
                            flags = last_spec ? (flags | F_LAST_SPECULAR) : (flags & ~F_LAST_SPECULAR);
                        }
                        const f3 no = ray_offset(rec.p, ray_along_normal);
                        const f3 nd = normalize(next_dir);     // ray(o, d, n) constructor re-normalises (ray.h:17-24)
                        pb.ray_o[slot] = make_float4(no.x, no.y, no.z, pdfb);
                        pb.ray_d[slot] = make_float4(nd.x, nd.y, nd.z, __uint_as_float(flags));
                        wrote_ray = true;
                        push_next = (bounce + 1 < fp.max_depth);
                    }
                    }
                    // ---- the NEE evaluation (see above); HitShadowRay runs only for non-terminated paths (pathtracing_impl.h:362-368)
                    if (nee && !(flags & F_TERMINATED)) {
                        Cmj sl; sl.idx = smp.idx; sl.dim = nee_dim + 1u; sl.scramble = smp.scramble;     // (behind the light pick's dimension)
                        const int32_t li = nee_light;
                        const float lightSelectPdf = sc.inv_n_lights;       // 1.0f / (float)n_lights, divided once at upload
                        LightSample ls;
                        sample_light(ls, sc.lights[li], sc, rec.p, orienting_normal, sl, &hp);
                        push_shadow = radiance_nee_then<MS>(sc, ray_dir, orienting_normal, m, rec.u, rec.v, lightSelectPdf, ls, mtrl_slot, pre_r, nullptr,
                                                            [&](const f3& radiance) {
                            // (next to the light index: HitShadowRay's surface_mtrl.stencil_type == ALWAYS, pathtracing.cpp:59-66)
                            const float lbits = __uint_as_float((uint32_t)li | ((m.attrib & kAttrStencilAlways) ? kShadowStencilFlag : 0u)
                                                                | (nee_final ? kShadowFinalFlag : 0u));
                            // the contribution first: `radiance` is dead before the shadow ray's geometry is worked out
                            const f3 lightcontrib = (thr_in * radiance) * albedo;
                            pb.sh_c[slot] = make_float4(lightcontrib.x, lightcontrib.y, lightcontrib.z, lbits);     // (the light bits again: all finish() needs)
                            const f3 dirToLight = normalize(ls.dir);
                            const float distToLight = length(ls.pos - rec.p);
                            const f3 so = ray_offset(rec.p, orienting_normal);
                            pb.sh_o[slot] = make_float4(so.x, so.y, so.z, distToLight);
                            pb.sh_d[slot] = make_float4(dirToLight.x, dirToLight.y, dirToLight.z, lbits);
                        }, &hp);
                    }
                }
            }
            if constexpr (REGEN) {
                if (!rg_stored) {       // a miss, an emissive or first-hit toon surface: the path ended without reaching the block above
                    rg_end = regen_end(false);
                    contrib_changed = false;
                    push_next = rg_end == 1u;
                    thr_stored = true;
                }
            }
            else if (!push_next && !wrote_ray) {
                // path ends here (terminated; a path that merely ran out of depth stored its flags with its last ray):
                // keep the flags for the sample epilogue
                pb.ray_d[slot] = make_float4(rd4.x, rd4.y, rd4.z, __uint_as_float(flags));
            }
            if (!thr_stored) pb.thr[slot] = make_float4(throughput.x, throughput.y, throughput.z, __uint_as_float(smp.dim));
            if (contrib_changed) {
                const f3 contrib = mk3(pb.contrib[slot]) + contrib_add;
                pb.contrib[slot] = make_float4(contrib.x, contrib.y, contrib.z, 0.0F);
            }
            if constexpr (REGEN) rg_need_bits = rg_end & 12u;
        }
        if constexpr (REGEN) {
            // (declared inside `if (valid)`: carried out through the flag byte)
            if (valid) order[0].oflag[order[0].inv[e]] = (uint8_t)((push_next ? 1u : 0u) | (push_shadow ? 2u : 0u) | rg_need_bits);
        }
        else push_bits |= (push_next ? 1u << k : 0u) | (push_shadow ? 0x10000u << k : 0u);
      }
#if ATN_SHADE_PARTITION
      auto entry_of = [&](int k) { return part.perm[(uint32_t)k * 256u + threadIdx.x]; };
#else
      auto entry_of = [&](int k) { return q[chunk + (uint32_t)k * 256u + threadIdx.x]; };
#endif
      if constexpr (REGEN) {
          // into this chunk's own regions, in the order of the chunk's queue entries, no atomic on a queue cursor (k_regen_compact makes
          // the dense queues of them)
          __syncthreads();
          // ---- new items for the slots whose item is finished: one atomic on the burst's item cursor for the whole chunk, consecutive
          // items (= neighbouring pixels of one frame) to the entries in queue order, GeneratePath for each into its slot
          {
              const uint32_t tid2 = here_v(threadIdx.x);
              const uint32_t lane2 = tid2 & 63u, wave2 = tid2 >> 6;
              uint32_t tot = 0;
#pragma unroll 1
              for (int k = 0; k < items; k++) {
                  const uint32_t jl = (uint32_t)k * 256u + threadIdx.x;
                  const bool need = jl < n_valid && (order[0].oflag[jl] & 4u) != 0u;
                  tot += (uint32_t)__popcll(__ballot(need));
              }
              if (lane2 == 0u) sh.wave_total[0][wave2] = tot;
              __syncthreads();
              if (threadIdx.x == 0u) {
                  const uint32_t a = sh.wave_total[0][0] + sh.wave_total[0][1] + sh.wave_total[0][2] + sh.wave_total[0][3];
                  sh.base[0] = a ? atomicAdd(pb.next_item, a) : 0u;
              }
              __syncthreads();
              uint32_t off = sh.base[0];
              for (uint32_t w = 0; w < wave2; w++) off += sh.wave_total[0][w];
#pragma unroll 1
              for (int k = 0; k < items; k++) {
                  const uint32_t jl = (uint32_t)k * 256u + threadIdx.x;
                  uint32_t f = jl < n_valid ? (uint32_t)order[0].oflag[jl] : 0u;
                  const bool need = (f & 4u) != 0u;
                  const unsigned long long m = __ballot(need);
                  if (need) {
                      const uint32_t item = off + bits_below_lane(m);
                      const uint32_t slot2 = q[chunk + jl];
                      const uint32_t pend_flag = (f & 8u) ? F_PENDING : 0u;
                      if (item < fp.n_items) {
                          const RegenState st = regen_primary_state(pb, fp, cam, item, 0u, pend_flag);
                          pb.ray_o[slot2] = st.o; pb.ray_d[slot2] = st.d; pb.thr[slot2] = st.t;
                          pb.contrib[slot2] = make_float4(0.0F, 0.0F, 0.0F, 0.0F);
                          f |= 1u;
                      }
                      else {
                          pb.ray_d[slot2] = make_float4(0.0F, 0.0F, 0.0F, __uint_as_float(pend_flag));       // retired (k_regen_flush looks for the flag)
                      }
                  }
                  off += (uint32_t)__popcll(m);
                  push_bits |= ((f & 1u) << k) | (((f >> 1) & 1u) << (16 + k));
              }
              __syncthreads();        // (sh is block_append2's next)
          }
          auto entry_in_order = [&](int k) { return q[chunk + (uint32_t)k * 256u + threadIdx.x]; };
          const uint32_t c = chunk / chunk_size;
          block_append2<decltype(entry_in_order), true>(sh, pb.q_regions + chunk, &pb.region_counts[2u * c], push_bits & 0xffffu,
                                                        pb.sh_regions + chunk, &pb.region_counts[2u * c + 1u], push_bits >> 16, entry_in_order);
          if (threadIdx.x < 2u) { const uint32_t v = pb.region_counts[2u * c + threadIdx.x]; if (v) atomicAdd(&pb.group_counts[2u * (c / kRegenGroup) + threadIdx.x], v); }
      }
      else
      block_append2(sh, qn, &pb.q_count[bounce_arg + 1], push_bits & 0xffffu, pb.shadow_q, &pb.sh_count[bounce_arg], push_bits >> 16, entry_of);
    }
#if !ATN_SHADE_PARTITION
    if (pb.stats) wave_add_stat(&pb.stats[2], nhits);
#endif
}

template <bool SVGF, int MS>
__global__ void ATN_SHADE_ATTR __launch_bounds__(256) k_shade(PathBuffers pb, DevScene sc, FrameParams fp, atn_camera_param cam, int32_t bounce, SvgfShade sv)
{
    shade_body<SVGF, MS>(pb, sc, fp, cam, bounce, sv);
}
// The three smaller material sets need 124 / 125 / 131 VGPRs (built without the SLP vectoriser, build.py): held to 128 they run
// 4 waves per SIMD without a spill.  WAVES = 5 holds them to 96 registers with 16-18 spilled ones: alone that launch is 3-6 %
// SLOWER, but with frames in flight the shade waves of one frame share the SIMDs with the persistent trace waves of another
// (76 VGPRs each), and two 96-register waves fit where one 124-register wave did: frame THROUGHPUT -2.9 % on the atrium, -1.7 % on
// the Cornell box, -0.2 % on sponza_lod; latency +1-3 % (profiles/r04_variants_shade_waves.txt).  So the host launches WAVES = 5
// when frames overlap and 4 when a caller waits for every frame.  6 waves (80 registers, 36-47 spilled) lose everywhere.
// The larger sets (139 .. 205 VGPRs) would spill too much: they keep the compiler's own allocation (k_shade).
template <bool SVGF, int MS, int WAVES>
__global__ void __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) __launch_bounds__(256) k_shade_wn(PathBuffers pb, DevScene sc, FrameParams fp, atn_camera_param cam, int32_t bounce, SvgfShade sv)
{
    shade_body<SVGF, MS>(pb, sc, fp, cam, bounce, sv);
}

// HitShadowRay -> HitTestToTargetLight -> scene::hitLight
// (pathtracing_impl.h:266-393, scene/scene.h:64-134): closest hit toward the light, visible iff
// the hit object IS the light object (or nothing is hit / infinite / singular rules).
// ALPHA = false compiles the "ignored hit" rules (alpha translucency, stencil) out: chosen by the host when no uploaded
// material can be ignored (DevScene::any_alpha == 0), which takes evaluate_hit + a texture fetch -- and their registers
// -- out of the walk.
//
// Lookups (HitTestToTargetLight's loop, pathtracing_impl.h:295-336): a hit on an alpha-translucent surface -- or, when
// the SHADED surface's material has StencilType::ALWAYS, on a STENCIL surface -- is ignored and the ray restarts behind
// it, up to 10 times when scene_rendering_config.enable_alpha_blending is set or the stencil check applies, ONCE
// otherwise (the ray then simply counts as blocked).  A restart re-uses the lane: finish() returns true with the new
// ray.  Payload bits: 0-25 slot, 26 "an ignored hit was the light object", 27-30 lookups done, 31 (FusedJob) shadow.

// REGEN (path regeneration): the shadow ray of a path's last vertex (kShadowFinalFlag) adds its light to `pend` -- by then the slot's
// `contrib` belongs to the pixel's next sample.
template <bool ALPHA, bool REGEN = false>
struct ShadowJob {
    PathBuffers pb;
    DevScene sc;
    float t_min;
    ATN_DEV void fetch(uint32_t j, float4& a, float4& b, float& stop_t) const { fetch_slot(pb.shadow_q[j], a, b, stop_t); }
    ATN_DEV void fetch_slot(uint32_t slot, float4& a, float4& b, float& stop_t) const
    {
        const float4 so = pb.sh_o[slot], sd = pb.sh_d[slot];
        const f3 dir = normalize(mk3(sd));      // aten::ray(org, dir) constructor re-normalises (pathtracing_impl.h:380)
        // scene::hitLight (scene/scene.h:118-131) needs the closest hit's OBJECT only for area lights.  For an
        // infinite light without object its answer is exactly `!isHit`: any accepted hit settles it.  For a point /
        // spot light it is `hit.t > distToLight`: the walk's t_max caps only box tests, so a triangle BEHIND the light
        // can be accepted first (triangle hits are accepted against isect.t = inf, threaded_bvh_traverser.h:236-262)
        // while the closest hit is a nearer blocker -- only an accepted hit with t <= distToLight settles it early.
        const uint32_t lbits = __float_as_uint(sd.w);
        const atn_light_param* lp = &sc.lights[lbits & 0xffffffu];
        const bool has_obj = lp->type == ATN_LIGHT_AREA && lp->arealight_objid >= 0;
        // (a light with neither object nor attribute is visible iff nothing is hit, like an infinite one)
        const bool near_only = (lp->attrib & ATN_LIGHT_ATTR_SINGULAR) && !(lp->attrib & ATN_LIGHT_ATTR_INFINITE);
        // An area light whose object is planar and rigidly placed is met where the ray crosses its plane and nowhere else; when that
        // crossing is certain to lie beyond 0.999 distToLight, a hit nearer than that is on another object and settles "blocked"
        // (scene_upload.hpp, planar_area_light, has the argument and the two conditions below).
        bool planar = false;
        if (has_obj && sc.planar_lights != 0) {
            const float4 pl = sc.light_plane[lbits & 0xffffffu];
            const float cos_l = fabsf(dot(mk3(pl), dir));
            // the origin is ray::Offset(p, n): at most 256 ulps (or 2^-16) per coordinate from p
            const float big = fmaxf(fmaxf(fabsf(so.x), fabsf(so.y)), fabsf(so.z));
            const float off = 1.7320508F * fmaxf(big * (300.0F * 1.1920929e-7F), 1.0F / 65536.0F);
            planar = pl.w != 0.0F && cos_l >= 0.01F && off <= 5e-4F * so.w * cos_l;
        }
        stop_t = has_obj ? (planar ? so.w * 0.999F : -kInf) : (near_only ? so.w : kInf);
        // with more than one lookup an ignored hit restarts the ray BEHIND it, so it has to be the closest one
        if (ALPHA && sc.any_alpha && (sc.enable_alpha_blending || (lbits & kShadowStencilFlag))) stop_t = -kInf;
        a = make_float4(so.x, so.y, so.z, so.w - kEps);         // t_max = distToLight - AT_MATH_EPSILON (:304)
        b = make_float4(dir.x, dir.y, dir.z, __uint_as_float(slot));
    }
    ATN_DEV void cost(uint32_t payload, uint32_t nodes, uint32_t tris) const
    {
        const uint32_t slot = payload & kShadowSlotMask;
        if (pb.cost) { atomicAdd(&pb.cost[2u * slot], nodes); atomicAdd(&pb.cost[2u * slot + 1u], tris); }
    }
    ATN_DEV bool finish(uint32_t payload, const Hit& h, bool isHit, float4& ra, float4& rb, float& rstop) const
    {
        const uint32_t slot = payload & kShadowSlotMask;
        const uint32_t lookups = (payload >> 27) & 15u;
        // one 16-byte read settles the common case: the light contribution carries the light bits; the distance and
        // the ray itself are read only by the branches that need them
        const float4 lc = pb.sh_c[slot];
        const uint32_t lbits = __float_as_uint(lc.w);
        const atn_light_param* lp = &sc.lights[lbits & 0xffffffu];
        const int32_t ltype = lp->type, lobj = lp->arealight_objid;
        const uint32_t lattr = lp->attrib;
        const int32_t lightobj = (ltype == ATN_LIGHT_AREA && lobj >= 0) ? lobj : -1;
        // `hitobj` survives ignored hits (the reference keeps the pointer across lookups, :291,309): after a miss it is
        // the last ignored object, or the light object if nothing was ever hit
        const bool same_obj = isHit ? h.objid == lightobj : (lookups == 0u || (payload & (1u << 26)) != 0u);
        bool visible;
        if (same_obj) visible = true;
        else if (lattr & ATN_LIGHT_ATTR_INFINITE) visible = !isHit;
        else if (lattr & ATN_LIGHT_ATTR_SINGULAR) visible = h.t > pb.sh_o[slot].w;     // distToLight
        else visible = false;
        if (ALPHA && sc.any_alpha && isHit) {
            const bool need_stencil = (lbits & kShadowStencilFlag) != 0u;
            const uint32_t max_lookups = (sc.enable_alpha_blending || need_stencil) ? 10u : 1u;
            // With a budget of one an ignored hit just means "blocked", which every hit that is not `visible` means
            // anyway: only then is the material worth a look.  (A walk that stopped early -- fetch -- is never
            // `visible` and has a budget of one, so `h` is the exact closest hit whenever it matters.)
            if (visible || max_lookups > 1u) {
                const int32_t mid = triangle_mtrlid(sc, h.tri);
                const uint32_t mattr = mid >= 0 ? sc.materials[mid].attrib : 0u;
                bool ignore = need_stencil && (mattr & kAttrStencilStencil);
                f3 hit_p = mk3(0.0F), hit_n = mk3(0.0F, 1.0F, 0.0F);
                if (ignore || (mattr & kAttrMaybeAlpha)) {
                    HitRec rec;
                    evaluate_hit(rec, sc, h.objid, h.tri, h.a, h.b);
                    hit_p = rec.p; hit_n = rec.normal;
                    if (mattr & kAttrMaybeAlpha) {
                        // material::isTranslucentByAlpha (material.cpp:193-210); only flagged materials can have alpha < 1
                        const DevMaterial& hm = sc.materials[mid];
                        const float4 albedo = sample_texture(sc, hm.albedoMap, rec.u, rec.v, make_float4(1.0F, 1.0F, 1.0F, 1.0F));
                        if (albedo.w * hm.baseColor.w < 1.0F) ignore = true;
                    }
                }
                if (ignore) {
                    if (lookups + 1u >= max_lookups) return false;      // budget spent: is_hit_to_light stays false
                    // r = ray(rec.p, original_ray.dir, normal facing along the ray), :319-330
                    const float distToLight = pb.sh_o[slot].w;
                    const f3 odir = normalize(mk3(pb.sh_d[slot]));
                    const bool is_same_facing = dot(hit_n, odir) > 0.0F;
                    const f3 on = is_same_facing ? hit_n : -hit_n;
                    const f3 o = ray_offset(hit_p, on);
                    const f3 d = normalize(odir);
                    ra = make_float4(o.x, o.y, o.z, distToLight - kEps);
                    rb = make_float4(d.x, d.y, d.z, __uint_as_float(slot | ((lookups + 1u) << 27) | (h.objid == lightobj ? (1u << 26) : 0u)));
                    rstop = -kInf;
                    return true;
                }
            }
        }
        if (visible) {
            float4* dst = pb.contrib;
            if constexpr (REGEN) { if (lbits & kShadowFinalFlag) dst = pb.pend; }
            const float4 c = dst[slot];
            dst[slot] = make_float4(c.x + lc.x, c.y + lc.y, c.z + lc.z, REGEN ? c.w : 0.0F);       // (pend.w: the item the sample belongs to)
        }
        return false;
    }
};

template <bool COUNT, bool REFILL>
__global__ void __launch_bounds__(kTraceBlock > 256 ? kTraceBlock : 256) k_trace_shadow(PathBuffers pb, DevScene sc, int32_t bounce)
{
    const uint32_t count = pb.sh_count[bounce];
    const ShadowJob<true> job{ pb, sc, kEps };
    TravCounters tc{};
    trace_dispatch<COUNT, REFILL>(sc, count, &pb.fetch_shadow[bounce], job, &tc);
    if (COUNT) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&pb.stats[1], (unsigned long long)count);
        wave_add_stat(&pb.stats[5], tc.nodes);
        wave_add_stat(&pb.stats[6], tc.tris);
    }
}

// One launch for the shadow rays of bounce `bs` AND the closest-hit rays of bounce `bc` (both are products of
// shade(bs); bc = bs + 1): every trace launch carries a size-independent ~0.13 ms (the dependent-load chain of its
// longest ray), so a frame pays it depth + 1 times instead of 2 * depth times, and each launch has twice the rays to
// fill the machine with.  The two job kinds touch disjoint state (shadow: contrib; closest: isect).  bs < 0 or
// bc < 0 = that half is absent (first / last launch of a sample).
template <bool ALPHA, bool REGEN = false>
struct FusedJob {
    ShadowJob<ALPHA, REGEN> s;
    ClosestJob c;
    uint32_t n_shadow;
    float t_min;
    ATN_DEV void fetch(uint32_t j, float4& a, float4& b, float& stop_t) const
    {
        if (j < n_shadow) {
            s.fetch(j, a, b, stop_t);
            b.w = __uint_as_float(__float_as_uint(b.w) | 0x80000000u);
        }
        else {
            c.fetch(j - n_shadow, a, b, stop_t);
        }
    }
    ATN_DEV bool finish(uint32_t payload, const Hit& h, bool is_hit, float4& ra, float4& rb, float& rstop) const
    {
        if (payload & 0x80000000u) {
            const bool again = s.finish(payload & 0x7fffffffu, h, is_hit, ra, rb, rstop);
            if (again) rb.w = __uint_as_float(__float_as_uint(rb.w) | 0x80000000u);
            return again;
        }
        return c.finish(payload, h, is_hit, ra, rb, rstop);
    }
    ATN_DEV void cost(uint32_t payload, uint32_t nodes, uint32_t tris) const
    {
        if (payload & 0x80000000u) s.cost(payload & 0x7fffffffu, nodes, tris); else c.cost(payload, nodes, tris);
    }
};

// LDSN: the walk over an LDS copy of the whole node image (small scenes)
template <bool REFILL, bool ALPHA, bool LDSN = false, bool REGEN = false>
__global__ void ATN_TRACE_ATTR __launch_bounds__(kTraceBlock > 256 ? kTraceBlock : 256) k_trace_fused(PathBuffers pb, DevScene sc, int32_t bs, int32_t bc, int32_t launch)
{
    const uint32_t n_shadow = bs >= 0 ? pb.sh_count[bs] : 0u;
    const uint32_t n_closest = bc >= 0 ? pb.q_count[bc] : 0u;
    const FusedJob<ALPHA, REGEN> job{ ShadowJob<ALPHA, REGEN>{ pb, sc, kEps }, ClosestJob{ pb, pb.queue[(bc >= 0 ? bc : 0) & 1], kEps }, n_shadow, kEps };
    TravCounters tc{};
    trace_dispatch<false, REFILL, FusedJob<ALPHA, REGEN>, LDSN>(sc, n_shadow + n_closest, &pb.fetch_closest[launch], job, &tc);
}

// k_shade of the regenerated pool (PathTracing::run_regen): stage `stage` of the burst
template <int MS>
__global__ void __launch_bounds__(256) k_regen_shade(PathBuffers pb, DevScene sc, FrameParams fp, atn_camera_param cam, int32_t stage, RegenOut ro)
{
    shade_body<false, MS, true>(pb, sc, fp, cam, stage, SvgfShade{}, ro);
}
template <int MS, int WAVES>
__global__ void __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) __launch_bounds__(256) k_regen_shade_wn(PathBuffers pb, DevScene sc, FrameParams fp, atn_camera_param cam, int32_t stage, RegenOut ro)
{
    shade_body<false, MS, true>(pb, sc, fp, cam, stage, SvgfShade{}, ro);
}

#if ATN_MAIN_TU
// Per-sample epilogue of OnRender's inner loop (pathtracing.cpp:339-352): skip invalid colours,
// accumulate, stop sampling this pixel once its path terminated.
__global__ void __launch_bounds__(256) k_accumulate_sample(PathBuffers pb, FrameParams fp)
{
    const uint32_t slot = (uint32_t)fp.slot_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= (uint32_t)fp.slot_end) return;
    int32_t x, y;
    if (!slot_to_pixel(fp, slot, x, y)) return;
    if (fp.sample > 0 && pb.done[slot]) return;
    const float4 c = pb.contrib[slot];
    const bool invalid = isnan(c.x) || isinf(c.x) || isnan(c.y) || isinf(c.y) || isnan(c.z) || isinf(c.z)
        || c.x < 0 || c.y < 0 || c.z < 0;                     // Renderer::isInvalidColor, renderer.h:58-68
    if (invalid) return;
    float4 a = pb.accum[slot];
    a.x += c.x; a.y += c.y; a.z += c.z; a.w += 1.0F;
    pb.accum[slot] = a;
    const uint32_t flags = __float_as_uint(pb.ray_d[slot].w);
    if (fp.break_on_terminate && (flags & F_TERMINATED)) pb.done[slot] = 1;
}
#endif  // ATN_MAIN_TU

// col / cnt -> Film::put / FilmProgressive::put (renderer/film.cpp:33-45,61-71).
// film: full-frame vec4[w*h] (row 0 = bottom); tile_out: this GPU's pixels in slot order
// (the buffer that is all-gathered over RCCL when the screen is sharded).
// ONE_SAMPLE: the frame has a single sample per pixel, so the per-sample epilogue (k_accumulate_sample) is folded in:
// col = valid ? contrib : 0, cnt = valid ? 1 : 0 -- the same values the two kernels produce, without the accum round trip.
template <bool ONE_SAMPLE>
__global__ void __launch_bounds__(256) k_gather(PathBuffers pb, FrameParams fp, float4* film, float4* tile_out)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= (uint32_t)fp.n_slots) return;
    int32_t x, y;
    float4 out = make_float4(0, 0, 0, 0);
    if (slot_to_pixel(fp, slot, x, y)) {
        float4 a;
        if (ONE_SAMPLE) {
            const float4 c = pb.contrib[slot];
            const bool invalid = isnan(c.x) || isinf(c.x) || isnan(c.y) || isinf(c.y) || isnan(c.z) || isinf(c.z)
                || c.x < 0 || c.y < 0 || c.z < 0;                     // Renderer::isInvalidColor, renderer.h:58-68
            a = invalid ? make_float4(0.0F, 0.0F, 0.0F, 0.0F) : make_float4(0.0F + c.x, 0.0F + c.y, 0.0F + c.z, 1.0F);
        }
        else {
            a = pb.accum[slot];
        }
        const float cnt = a.w;      // (float)cnt of an integer counter
        const float4 v = make_float4(a.x / cnt, a.y / cnt, a.z / cnt, 1.0F);
        const uint32_t idx = (uint32_t)(y * fp.width + x);
        if (fp.progressive) {
            const float4 cur = film[idx];
            const float n = (float)((int32_t)cur.w);
            const float d = n + 1;
            out = make_float4((n * cur.x + v.x) / d, (n * cur.y + v.y) / d, (n * cur.z + v.z) / d, n + 1);
        }
        else {
            out = v;
        }
        film[idx] = out;
    }
    if (tile_out) tile_out[slot] = out;
}

#if ATN_MAIN_TU       // (to the end of the file: stage kernels and helpers of aten_amd.hip)
// Scatter all-gathered tile buffers (rank-major, each n_slots_per_rank float4) into a full frame.
__global__ void __launch_bounds__(256) k_assemble_tiles(const float4* __restrict__ gathered, float4* film,
                                                        int32_t width, int32_t height, int32_t tiles_x, int32_t tiles_y,
                                                        int32_t world, int32_t slots_per_rank)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (uint32_t)(world * slots_per_rank)) return;
    FrameParams fp{};
    fp.width = width; fp.height = height; fp.tiles_x = tiles_x; fp.tiles_y = tiles_y; fp.tiles_x_rcp = udiv_rcp((uint32_t)tiles_x);
    fp.world = world; fp.rank = (int32_t)(g / (uint32_t)slots_per_rank);
    const uint32_t slot = g % (uint32_t)slots_per_rank;
    int32_t x, y;
    if (slot_to_pixel(fp, slot, x, y)) film[y * width + x] = gathered[g];
}

// ---- stage kernels used by the parity tests through the C-ABI --------------------------------
struct BatchJob {
    DevScene sc;
    const atn_ray* __restrict__ rays;
    atn_intersection* out;
    float t_min, t_max;
    ATN_DEV void fetch(uint32_t j, float4& a, float4& b, float& stop_t) const
    {
        const atn_ray r = rays[j];
        stop_t = -kInf;
        a = make_float4(r.org[0], r.org[1], r.org[2], t_max);
        b = make_float4(r.dir[0], r.dir[1], r.dir[2], __uint_as_float(j));
    }
    ATN_DEV bool finish(uint32_t j, const Hit& h, bool, float4&, float4&, float&) const
    {
        atn_intersection o;
        o.t = h.t; o.objid = h.objid; o.tri_id = h.tri; o.a = h.a; o.b = h.b; o.isVoxel = 0;
        o.mtrlid = -1; o.meshid = -1;
        if (h.objid >= 0) {
            const atn_triangle_param tp = sc.tris[h.tri];
            o.mtrlid = tp.mtrlid;
            o.meshid = tp.mesh_id < 0 ? h.meshid : tp.mesh_id;     // threaded_bvh_traverser.h:206-209
        }
        out[j] = o;
        return false;
    }
    ATN_DEV void cost(uint32_t, uint32_t, uint32_t) const {}
};

// The renderer's traversal core over caller-provided rays (parity probe, atn_trace_closest).
template <bool COUNT, bool REFILL>
__global__ void __launch_bounds__(kTraceBlock > 256 ? kTraceBlock : 256) k_trace_batch(DevScene sc, const atn_ray* __restrict__ rays, uint32_t n,
                                                     float t_min, float t_max, atn_intersection* out,
                                                     unsigned long long* stats)
{
    const BatchJob job{ sc, rays, out, t_min, t_max };
    TravCounters tc{};
    trace_dispatch<COUNT, REFILL>(sc, n, reinterpret_cast<uint32_t*>(&stats[7]), job, &tc);     // stats[7]: zeroed fetch cursor
    if (COUNT) { wave_add_stat(&stats[3], tc.nodes); wave_add_stat(&stats[4], tc.tris); }
}

__global__ void __launch_bounds__(256) k_export_rays(PathBuffers pb, FrameParams fp, atn_ray* out)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= (uint32_t)fp.n_slots) return;
    int32_t x, y;
    if (!slot_to_pixel(fp, slot, x, y)) return;
    const float4 o = pb.ray_o[slot], d = pb.ray_d[slot];
    atn_ray r; r.org[0] = o.x; r.org[1] = o.y; r.org[2] = o.z; r.dir[0] = d.x; r.dir[1] = d.y; r.dir[2] = d.z;
    out[y * fp.width + x] = r;
}

__global__ void __launch_bounds__(64) k_cmj_samples(uint32_t index, uint32_t dim, uint32_t scramble, int n, float* out)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        Cmj s; s.idx = index; s.dim = dim; s.scramble = scramble;
        for (int i = 0; i < n; i++) out[i] = cmj_next(s);
    }
}

// the math-library functions the float path calls, as this build answers them (atn_libm_probe; kinds: include/aten_amd.h)
__global__ void __launch_bounds__(256) k_libm_probe(int32_t kind, uint32_t n, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = a[i], y = b[i];
    float r = 0.0F;
    switch (kind) {
    case 0: r = sinf(x); break;
    case 1: r = cosf(x); break;
    case 2: r = atanf(x); break;
    case 3: r = acosf(x); break;
    case 4: r = atan2f(x, y); break;
    case 5: r = logf(x); break;
    case 6: r = expf(x); break;
    case 7: r = powf(x, y); break;
    case 8: r = sqrtf(x); break;
    case 9: r = x / y; break;
    case 10: r = 1.0F / sqrtf(x); break;
    default: break;
    }
    out[i] = r;
}

__global__ void __launch_bounds__(256) k_ray_offset(uint32_t n, const float* __restrict__ o, const float* __restrict__ nrm, float* __restrict__ out)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const f3 r = ray_offset(mk3(o[3 * k], o[3 * k + 1], o[3 * k + 2]), mk3(nrm[3 * k], nrm[3 * k + 1], nrm[3 * k + 2]));
    out[3 * k] = r.x; out[3 * k + 1] = r.y; out[3 * k + 2] = r.z;
}

// one thread per (index, dimension, scramble) triple: `draws` successive nextSample()
__global__ void __launch_bounds__(256) k_cmj_batch(uint32_t n, const uint32_t* __restrict__ index, const uint32_t* __restrict__ dim,
                                                   const uint32_t* __restrict__ scramble, int draws, float* __restrict__ out)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    Cmj s; s.idx = index[k]; s.dim = dim[k]; s.scramble = scramble[k];
    for (int d = 0; d < draws; d++) out[(size_t)k * draws + d] = cmj_next(s);
}

__global__ void __launch_bounds__(256) k_sample_texture(DevScene sc, int32_t texid, uint32_t n, const float* __restrict__ uv, float* out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 c = sample_texture(sc, texid, uv[2 * i], uv[2 * i + 1], make_float4(0, 0, 0, 0));
    out[4 * i] = c.x; out[4 * i + 1] = c.y; out[4 * i + 2] = c.z; out[4 * i + 3] = c.w;
}

// BSDF table: for case i sample at (n, wi) with sampler (index, dim 0, scramble), then re-evaluate pdf/bsdf at the sampled dir
__global__ void __launch_bounds__(64) k_material_table(DevScene sc, int32_t mtrl_id, uint32_t n, const float* nrm, const float* wi,
                                                       const uint32_t* index, const uint32_t* dimension, const uint32_t* scramble, const float* uv,
                                                       float* out_sample, float* out_eval)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DevMaterial m = sc.materials[mtrl_id];
    Cmj s; s.idx = index[i]; s.dim = dimension ? dimension[i] : 0u; s.scramble = scramble[i];
    f3 N = mk3(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]);
    const f3 WI = mk3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]);
    // CarPaint: material::applyNormal runs first, as in shade (it draws the shared random number and may swap the normal)
    const float pre_r = m.type == ATN_MTRL_CARPAINT ? apply_normal<kMsCarPaint>(sc, m, mtrl_id, N, uv[2 * i], uv[2 * i + 1], WI, s) : 0.0F;
    MtrlSample ms;
    sample_material(ms, sc, m, N, WI, s, uv[2 * i], uv[2 * i + 1], mtrl_id, pre_r);
    float* o = out_sample + 7 * i;
    o[0] = ms.dir.x; o[1] = ms.dir.y; o[2] = ms.dir.z; o[3] = ms.bsdf.x; o[4] = ms.bsdf.y; o[5] = ms.bsdf.z; o[6] = ms.pdf;
    const float p = material_pdf(sc, m, N, WI, ms.dir, uv[2 * i], uv[2 * i + 1], mtrl_id);
    const MtrlSample ev = material_bsdf(sc, m, N, WI, ms.dir, uv[2 * i], uv[2 * i + 1], mtrl_id, pre_r);
    float* e = out_eval + 5 * i;
    e[0] = p; e[1] = ev.bsdf.x; e[2] = ev.bsdf.y; e[3] = ev.bsdf.z; e[4] = ev.pdf;
}

// material::samplePDF / sampleBSDF (material_impl.h:90-206) at CALLER-GIVEN outgoing directions: what the closed-form
// invariants integrate (the pdf over the sphere, bsdf * cos over the hemisphere; tests/test_gpu_invariants.py)
__global__ void __launch_bounds__(256) k_material_eval(DevScene sc, int32_t mtrl_id, uint32_t n, const float* nrm, const float* wi, const float* wo,
                                                       const float* uv, float* out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DevMaterial m = sc.materials[mtrl_id];
    const f3 N = mk3(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]);
    const f3 WI = mk3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]);
    const f3 WO = mk3(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]);
    const float p = material_pdf(sc, m, N, WI, WO, uv[2 * i], uv[2 * i + 1], mtrl_id);
    const MtrlSample ev = material_bsdf(sc, m, N, WI, WO, uv[2 * i], uv[2 * i + 1], mtrl_id, 0.0F);
    float* e = out + 5 * i;
    e[0] = p; e[1] = ev.bsdf.x; e[2] = ev.bsdf.y; e[3] = ev.bsdf.z; e[4] = ev.pdf;
}

// The renderer's queue append (block_append2) over caller-provided flags: entry i goes to queue A when
// flags_a[i] > 0 and to queue B when flags_b[i] > 0 (flags_b may be null) -- exactly the call k_shade makes for
// its next-bounce and shadow queues, same chunking (kChunkItems x 256 entries per block and atomic), grid-stride.
// The queues come out UNORDERED (slot order is irrelevant to the renderer); atn_compact sorts them on the host
// to present the stable contract of idaten::StreamCompaction::compact (StreamCompaction.cu:175-316).
__global__ void __launch_bounds__(256) k_compact_append(const int32_t* __restrict__ flags_a, const int32_t* __restrict__ flags_b, uint32_t n,
                                                        uint32_t* out_a, uint32_t* cnt_a, uint32_t* out_b, uint32_t* cnt_b)
{
    __shared__ BlockAppendShared sh;
    for (uint32_t chunk = blockIdx.x * kChunk; chunk < n; chunk += gridDim.x * kChunk) {
        uint32_t fa = 0, fb = 0;
#pragma unroll
        for (int k = 0; k < kChunkItems; k++) {
            const uint32_t i = chunk + (uint32_t)k * 256u + threadIdx.x;
            if (i < n) {
                if (flags_a[i] > 0) fa |= 1u << k;
                if (flags_b && flags_b[i] > 0) fb |= 1u << k;
            }
        }
        block_append2(sh, out_a, cnt_a, fa, out_b, flags_b ? cnt_b : (uint32_t*)nullptr, fb,
                          [&](int k) { return chunk + (uint32_t)k * 256u + threadIdx.x; });
    }
}

} // namespace atn

namespace atn {
// scene_dev.hpp's packed per-triangle shading record, (re)built on the device from the scene arrays
__global__ __launch_bounds__(256) void k_pack_shade_tris(const atn_triangle_param* __restrict__ tris, const float4* __restrict__ vtx_pos,
                                                         const float4* __restrict__ vtx_nml, uint32_t first, uint32_t count, float4* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t t = first + i;
    const float4* tp = reinterpret_cast<const float4*>(&tris[t]);
    const float4 h0 = tp[0], h1 = tp[1];
    const int32_t i0 = __float_as_int(h0.x), i1 = __float_as_int(h0.y), i2 = __float_as_int(h0.z);
    float4* q = out + (size_t)kShadeTriQuads * t;
    q[0] = vtx_pos[i0]; q[1] = vtx_pos[i1]; q[2] = vtx_pos[i2];
    q[3] = vtx_nml[i0]; q[4] = vtx_nml[i1]; q[5] = vtx_nml[i2];
    q[6] = h1;
    q[7] = make_float4(h0.x, h0.y, h0.z, 0.0F);
}
} // namespace atn

namespace atn {
// The per-pixel cost map of a count_stats frame (≙ the heat map the reference builds from PathTimeProfiler's per-path GPU
// timer, renderer/pathtracing/path_time_profiler.h:15-60 -- here the deterministic quantity behind the time: BVH node
// visits and triangle tests of all the pixel's walks, closest and shadow, all samples of the frame).
__global__ __launch_bounds__(256) void k_cost_to_pixels(FrameParams fp, const uint32_t* __restrict__ cost, uint32_t* __restrict__ out)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= (uint32_t)fp.n_slots) return;
    int32_t x, y;
    if (!slot_to_pixel(fp, slot, x, y)) return;
    const uint32_t p = (uint32_t)(y * fp.width + x);
    out[2u * p] = cost[2u * slot];
    out[2u * p + 1u] = cost[2u * slot + 1u];
}
} // namespace atn
#else
} // namespace atn
#endif  // ATN_MAIN_TU
