// libaten_amd.so: host-side renderer object + C-ABI (include/aten_amd.h) over the HIP kernels.
//
// atn::PathTracing mirrors the public surface of the reference's GPU seam --
// idaten::Renderer / idaten::PathTracing (src/libidaten/kernel/renderer.h:17-179,
// src/libidaten/kernel/pathtracing.cpp:23-153): UpdateSceneData, updateCamera, render, reset --
// with the CPU renderer's semantics (frame passed in, CMJ seeded once per sample).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/aten_amd.h"
#include "device/kernels.hpp"
#include "device/regen_launch.hpp"
#include "device/relaxed_launch.hpp"
#include "device/svgf.hpp"
#include "device/lbvh.hpp"
#include "host/scene_upload.hpp"
#include "host/ibl_precompute.hpp"

namespace atn {

#define ATN_HIP(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) { return fail(ATN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } \
    } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    ~DevBuf() { release(); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
    hipError_t resize(size_t count)
    {
        if (count <= n && p) return hipSuccess;
        release();
        if (count == 0) return hipSuccess;
        hipError_t e = hipMalloc((void**)&p, count * sizeof(T) + 256);     // slack: the walk's pair fetch reads 32 B past a record
        if (e == hipSuccess) n = count;
        return e;
    }
    void swap(DevBuf& o) { std::swap(p, o.p); std::swap(n, o.n); }
    hipError_t upload(const std::vector<T>& v, hipStream_t s)
    {
        hipError_t e = resize(v.size() ? v.size() : 1);
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
    }
};

// ---- which streams really run side by side ---------------------------------------------------------------------------
// HIP streams are multiplexed onto a few hardware queues (4 by default; under RCCL that number holds whatever
// GPU_MAX_HW_QUEUES says), chosen by the runtime when a stream is created, and two streams that share a queue execute
// strictly one after the other.  Frames in flight live on one stream per bank: measured with rocprofv3 (tools/trace_queues.py)
// under torch.distributed two of the three bank streams shared a queue -- two frames in flight instead of three, 4.20 ->
// 4.46 ms per frame at full size, and on an 8-way shard the 2-in-flight figure (0.83 instead of 0.67 ms).
// The API does not tell which queue a stream got, so it is MEASURED: two spin kernels of kSpinTicks, one per stream; side by
// side they take one spin, on one queue two (a false "clash" under host jitter only costs a needless replacement).
__global__ void k_spin(unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();      // constant 100 MHz counter
    while (wall_clock64() - t0 < ticks) { }
}
constexpr unsigned long long kSpinTicks = 30000;        // 300 us: long against launch + synchronise overhead and host jitter
inline bool streams_run_side_by_side(hipStream_t a, hipStream_t b, bool& ok)
{
    const double spin_us = (double)kSpinTicks / 100.0;
    for (int rep = 0; rep < 3; rep++) {                 // the first round also loads the kernel; one round under the limit settles it
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) { ok = false; return false; }
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, a, kSpinTicks);
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, b, kSpinTicks);
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) { ok = false; return false; }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        ok = true;
        if (us < 1.5 * spin_us) return true;            // two spins on one queue cannot take less than 2 x spin_us
    }
    ok = true;
    return false;
}

class PathTracing {
public:
    std::string last_error;
    int device = 0;
    hipStream_t stream = nullptr;
    // A frame's paths are cut into n_batches independent batches (contiguous slot ranges = groups of screen tiles),
    // each with its own queues and counters, and every batch runs its gen -> [trace, shade, shadow] x depth sequence on
    // its own stream.  A trace launch has a size-independent part of ~0.2 ms (ramp-up, and a tail in which a few rays
    // that visit 10x the mean number of nodes keep a handful of waves alive); with several batches in flight the other
    // batches' kernels fill the machine meanwhile.  Measured (sponza_lod 1080p): 6.83 -> 6.58 ms
    // on the whole frame, 4.35 -> 3.71 ms on half of it (the 2-GPU shard), no gain below ~200 K paths per batch.
    static constexpr int kMaxBatches = 8;
    static constexpr int kMaxShards = 64;       // screen shards of one node (atn_mgpu_*)
    hipStream_t bstream[kMaxBatches] = {};
    hipEvent_t ev_fork = nullptr, ev_join[kMaxBatches] = {};
    int n_batches = 3;
    bool batches_forced = false;    // ATEN_AMD_BATCHES / atn_set_path_batches given: no size policy on top
    bool fuse_traces = true;    // shadow(b) + closest(b+1) in one launch (k_trace_fused); ATEN_AMD_FUSE=0 disables (experiments)
    bool env_probe_streams = true;  // ATEN_AMD_PROBE_STREAMS=0: take the bank streams as the runtime hands them out
    bool env_lds_nodes = true;  // small node images are walked from an LDS copy (ATEN_AMD_LDS_NODES=0: from global memory)
    int env_anyhit_twin = 1;    // ATEN_AMD_ANYHIT_TWIN: 0 = no any-hit twins, 1 = where the model says they pay (scene_upload.hpp, kTwinPays), 2 = wherever possible
    int env_anyhit_twin_dirs = 8;   // ATEN_AMD_ANYHIT_TWIN_DIRS: 8 = one twin per direction octant (default), 1 = the one direction-free twin
    bool opt_node_layout = true, opt_planar_lights = true;      // ATEN_AMD_NODE_LAYOUT / ATEN_AMD_PLANAR_LIGHTS at creation; atn_set_upload_options
    bool env_atrous4 = true;    // SVGF a-trous levels with four pixels per thread (k_svgf_atrous4); ATEN_AMD_SVGF_ATROUS4=0: one pixel per thread
    uint32_t env_trace_blocks = 0;
    int env_shade_waves = 0;    // ATEN_AMD_SHADE_WAVES=4|5 forces the k_shade_wn flavour (default: 5 when frames are in flight, else 4)
    int env_shade_items = 0, env_flavour = -1;

    // scene (HBM-resident after UpdateSceneData)
    DevBuf<float4> nodes, vtx_pos, vtx_nml, matrices, texels, carpaint, shade_tris;
    DevBuf<atn_triangle_param> tris;
    DevBuf<atn_object_param> objects;
    DevBuf<DevMaterial> materials;
    DevBuf<atn_light_param> lights;
    DevBuf<float4> light_plane;
    DevBuf<DevTexture> textures;
    DevBuf<atn_toon_param> toon;
    DevBuf<atn_light_param> npr_lights;
    DevBuf<float> screen_shadow;
    DevBuf<uint32_t> texels8;
    DevScene scene{};
    bool has_scene = false, has_camera = false;
    std::vector<int32_t> list_root_link;    // typed root link of every BVH list (top layer = list 0, stored last)
    uint32_t n_planar_lights = 0;           // area lights flagged planar + rigid at upload (scene.planar_lights says whether the flags still hold)
    // What a planar light's flag was derived from (scene_upload.hpp: planar_area_light): the instance's and the mesh's object records,
    // the instance's two matrices, the mesh's triangles and the vertices they name.  An update that leaves ALL of it byte for byte
    // as it was leaves the flag true; any other update drops every flag until the next full upload (never re-derived: the host
    // keeps no copy of the vertices).  r06: a deformation tick (new blob vertices, rebuilt list, same lamp) used to drop them --
    // deformable room 1.97 -> 2.04 ms per frame after the first tick (profiles/r06_rebuilt_layout_bound.jsonl).
    struct PlanarCert {
        int32_t inst_obj = -1, mesh_obj = -1, mtx_id = -1;
        atn_object_param inst{}, mesh{};
        atn_mat4 l2w{}, w2l{};
        uint32_t tri_first = 0, tri_count = 0, vtx_lo = 0, vtx_hi = 0;     // [tri_first, tri_first + tri_count), [vtx_lo, vtx_hi]
    };
    std::vector<PlanarCert> planar_certs;
    void collect_planar_certs(const atn_scene_desc* s, const std::vector<atn_light_param>& lights)
    {
        planar_certs.clear();
        for (size_t i = 0; i < lights.size() && i < s->n_lights; i++) {
            if (!lights[i]._pad) continue;
            PlanarCert c;                                          // (planar_area_light accepted this light: every index below is in range)
            c.inst_obj = s->lights[i].arealight_objid;
            c.inst = s->objects[c.inst_obj];
            const atn_object_param* o = &s->objects[c.inst_obj];
            if (o->type == ATN_OBJ_INSTANCE) {
                c.mtx_id = o->mtx_id;
                if (c.mtx_id >= 0) { c.l2w = s->matrices[c.mtx_id]; c.w2l = s->matrices[c.mtx_id + 1]; }
                c.mesh_obj = o->object_id;
                c.mesh = s->objects[c.mesh_obj];
                o = &s->objects[c.mesh_obj];
            }
            c.tri_first = (uint32_t)o->triangle_id; c.tri_count = (uint32_t)o->triangle_num;
            c.vtx_lo = 0xFFFFFFFFu; c.vtx_hi = 0;
            for (uint32_t t = c.tri_first; t < c.tri_first + c.tri_count; t++)
                for (int k = 0; k < 3; k++) {
                    const uint32_t v = (uint32_t)s->triangles[t].idx[k];
                    c.vtx_lo = std::min(c.vtx_lo, v); c.vtx_hi = std::max(c.vtx_hi, v);
                }
            planar_certs.push_back(c);
        }
    }
    // the objects (and, when given, the matrices) of an atn_update_tlas leave every planar light's instance as it was uploaded
    bool planar_certs_survive_objects(const atn_object_param* objs, uint32_t n_objs, const atn_mat4* mtxs, uint32_t n_mtxs) const
    {
        for (const PlanarCert& c : planar_certs) {
            if ((uint32_t)c.inst_obj >= n_objs || std::memcmp(&objs[c.inst_obj], &c.inst, sizeof(atn_object_param)) != 0) return false;
            if (c.mesh_obj >= 0 && ((uint32_t)c.mesh_obj >= n_objs || std::memcmp(&objs[c.mesh_obj], &c.mesh, sizeof(atn_object_param)) != 0)) return false;
            if (c.mtx_id >= 0 && n_mtxs
                && ((uint32_t)c.mtx_id + 1 >= n_mtxs || std::memcmp(&mtxs[c.mtx_id], &c.l2w, sizeof(atn_mat4)) != 0
                    || std::memcmp(&mtxs[c.mtx_id + 1], &c.w2l, sizeof(atn_mat4)) != 0)) return false;
        }
        return true;
    }
    // an atn_update_geometry over these vertex / triangle ranges touches none of a planar light's triangles or vertices
    bool planar_certs_survive_geometry(uint32_t vtx_offset, uint32_t n_vtx, uint32_t tri_offset, uint32_t n_tr) const
    {
        for (const PlanarCert& c : planar_certs) {
            if (n_vtx && vtx_offset <= c.vtx_hi && (uint64_t)vtx_offset + n_vtx > c.vtx_lo) return false;
            if (n_tr && tri_offset < c.tri_first + c.tri_count && (uint64_t)tri_offset + n_tr > c.tri_first) return false;
        }
        return true;
    }
    std::vector<int32_t> list_twin_delta;   // HostSceneImage::list_twin_delta: where each list's any-hit twin starts (0 = none)
    std::vector<HostSceneImage::TlasRef> tlas_refs;     // the top layer's TLAS-leaf records and the lists they enter
    uint32_t top_base = 0, n_host_matrices = 0;
    std::vector<atn_mat4> host_matrices;    // the caller's matrices as last uploaded
    std::vector<uint32_t> list_base, list_bytes, list_tri_leaves, list_inner;   // region of every list in the node image
    uint32_t n_scene_tris = 0, n_scene_vtx = 0, n_scene_mtrls = 0;

    // LBVH rebuild scratch (atn_lbvh_*), grown on demand
    struct LbvhScratch {
        DevBuf<uint32_t> codes[2], indices[2], counts, totals, arrived, offs;
        DevBuf<uint8_t> flips;          // per inner node, bit g: twin g swaps the children (k_lbvh_twin_flips)
        DevBuf<int32_t> left, right, parent, first, last;
        DevBuf<atn_bvh_node> ref_nodes;
        DevBuf<LbvhBox> pre, suf, sup;
        DevBuf<atn_triangle_param> tris;     // atn_lbvh_build's own inputs
        DevBuf<float4> vtx;
    } lb;
    uint64_t n_bottom_nodes = 0;
    atn_camera_param camera{};

    // optional samplers (atn_set_sampling_options)
    std::vector<atn_vec4> env_host;         // host copy of the environment map (the tables are built on demand)
    int32_t env_w = 0, env_h = 0;
    float env_multiplyer = 1.0F;
    DevBuf<float> ibl_cdf_v, ibl_cdf_u;
    bool ibl_tables_ready = false;
    int32_t opt_ibl_importance = 0, opt_tex_bilinear = 0;

    // sampler
    DevBuf<uint32_t> seeds;
    uint32_t n_seeds = 0;

    // path state
    DevBuf<float4> ray_o, ray_d, thr, contrib, isect, sh_o, sh_d, sh_c, accum, film, tile_out;
    DevBuf<float4> pend;                    // path regeneration: a pixel's previous sample while its last shadow ray is in flight
    DevBuf<float4> rg_frames;               // path regeneration: [frames of the burst][slots] pixel values on their way to the film (k_regen_end)
    DevBuf<uint32_t> done, queue0, queue1, shadow_q, counters;
    DevBuf<uint32_t> rg_counters;           // path regeneration: [3][rg_stages + 3] live paths, shadow rays, fetch cursor of every stage (from stage -1)
    DevBuf<uint32_t> rg_regions;            // path regeneration: what a shade launch hands to the compaction: [2][slots + chunk] entries, counts per chunk and group
    int32_t rg_stages = 0;                  // stages the counters hold (of the bank's last regenerated burst: atn_regen_stage_counts)
    DevBuf<unsigned long long> stats;
    DevBuf<uint32_t> cost, cost_film;       // per-slot / per-pixel {node visits, triangle tests} of the last count_stats frame
    int32_t cost_w = 0, cost_h = 0;
    int32_t film_w = 0, film_h = 0;
    int32_t rank = 0, world = 1;
    uint32_t n_slots = 0;
    int32_t counters_depth = 0;

    // Frames in flight (atn_set_frames_in_flight): everything a frame's kernels write except the film lives in a BANK --
    // path state, queues, counters, tile buffer, and the streams / events the frame runs on.  The members above ARE the
    // current bank; render() rotates them with the spare banks, so frame f + 1 is enqueued on another stream while frame
    // f's launch tails (a few long rays keep a handful of waves alive at the end of every trace launch) still run.
    // Only the film orders consecutive frames: a frame's k_gather waits for the previous frame's.
    static constexpr int kMaxInFlight = 4;      // (5 / 6 / 8 banks with 8 hardware queues: no gain, profiles/r04_variants_shade_waves.txt)
    struct Bank {
        DevBuf<float4> ray_o, ray_d, thr, contrib, isect, sh_o, sh_d, sh_c, accum, tile_out, pend, rg_frames;
        DevBuf<uint32_t> done, queue0, queue1, shadow_q, counters, rg_counters, rg_regions;
        int32_t rg_stages = 0;
        uint32_t n_slots = 0;
        int32_t counters_depth = 0;
        uint64_t bank_epoch = 0;
        int scene_set = 0;
        hipEvent_t ev_read[3] = {};     // [scene set id]: the last frame of this bank that read that set has finished with the scene
        hipStream_t stream = nullptr, bstream[kMaxBatches] = {};
        hipEvent_t ev_fork = nullptr, ev_join[kMaxBatches] = {}, ev_gather = nullptr;
        int batch_streams_checked = 0;
    };
    Bank spare[kMaxInFlight - 1];
    int frames_in_flight = 1, n_spare_ready = 0;
    uint64_t frame_seq = 0;
    hipEvent_t ev_gather = nullptr;         // the current bank's "film updated" event
    // the film is one running mean shared by every bank: its last writer (a render()'s k_gather, or reset()'s clear).  One
    // event owned by the context, NOT by a bank -- svgf_render rotates banks without writing the film, so "the previous
    // bank's ev_gather" is not the film's last writer.  (A wait refers to the record that precedes it; re-recording is fine.)
    hipEvent_t ev_film = nullptr;
    bool film_pending = false;

    void swap_bank(Bank& b)
    {
        ray_o.swap(b.ray_o); ray_d.swap(b.ray_d); thr.swap(b.thr); contrib.swap(b.contrib); isect.swap(b.isect);
        sh_o.swap(b.sh_o); sh_d.swap(b.sh_d); sh_c.swap(b.sh_c); accum.swap(b.accum); tile_out.swap(b.tile_out);
        done.swap(b.done); queue0.swap(b.queue0); queue1.swap(b.queue1); shadow_q.swap(b.shadow_q);
        counters.swap(b.counters); pend.swap(b.pend); rg_frames.swap(b.rg_frames); rg_counters.swap(b.rg_counters); rg_regions.swap(b.rg_regions); std::swap(rg_stages, b.rg_stages);
        std::swap(n_slots, b.n_slots); std::swap(counters_depth, b.counters_depth); std::swap(bank_epoch, b.bank_epoch); std::swap(bank_scene_set, b.scene_set);
        for (int k = 0; k < 3; k++) std::swap(ev_read[k], b.ev_read[k]);
        std::swap(stream, b.stream); std::swap(ev_fork, b.ev_fork); std::swap(ev_gather, b.ev_gather);
        for (int k = 0; k < kMaxBatches; k++) { std::swap(bstream[k], b.bstream[k]); std::swap(ev_join[k], b.ev_join[k]); }
        std::swap(batch_streams_checked, b.batch_streams_checked);
    }

    int set_frames_in_flight(int n)
    {
        if (n < 1 || n > kMaxInFlight) return fail(ATN_ERR_INVALID_ARG, "frames in flight out of range");
        ATN_HIP(hipSetDevice(device));
        int rc = quiesce();
        if (rc) return rc;
        for (; n_spare_ready < n - 1; n_spare_ready++) {
            Bank& b = spare[n_spare_ready];
            ATN_HIP(hipStreamCreateWithFlags(&b.stream, hipStreamNonBlocking));
            ATN_HIP(hipEventCreateWithFlags(&b.ev_fork, hipEventDisableTiming));
            ATN_HIP(hipEventCreateWithFlags(&b.ev_gather, hipEventDisableTiming));
            for (int k = 0; k < kMaxBatches; k++) {
                ATN_HIP(hipStreamCreateWithFlags(&b.bstream[k], hipStreamNonBlocking));
                ATN_HIP(hipEventCreateWithFlags(&b.ev_join[k], hipEventDisableTiming));
            }
        }
        // the spare scene sets are only kept current while ticks flip through them: an in-place update (one frame in
        // flight) is not logged, so sets kept across N -> 1 -> N would come back stale.  Re-clone at the next flip.
        if (n != frames_in_flight) drop_alt_set();
        frames_in_flight = n;
        if (n > 1 && env_probe_streams) { int rc2 = separate_bank_streams(n); if (rc2) return rc2; }
        return ATN_OK;
    }

    // Every bank's main stream on a hardware queue of its own (see streams_run_side_by_side): a bank stream that shares its
    // queue with an earlier bank's is replaced by a newly created one, up to kStreamTries times (the rejected ones stay
    // alive until the end, so that the runtime's next choice differs).  Best effort: with fewer queues than banks the last
    // candidates are kept as they are.
    static constexpr int kStreamTries = 12;
    int bank_streams_checked = 0;       // banks 0 .. n-1 are known to be pairwise concurrent
    int n_stream_swaps = 0;             // (diagnostics: atn_get_stream_swaps)
    // st[first .. n-1] each made concurrent with every stream before it in the list (replaced in place when they clash)
    int separate_streams(hipStream_t** st, int n, int first)
    {
        std::vector<hipStream_t> rejected;
        for (int i = (first > 1 ? first : 1); i < n; i++) {
            for (int t = 0; t < kStreamTries; t++) {
                bool clash = false;
                for (int j = 0; j < i && !clash; j++) {
                    bool ok = true;
                    const bool par = streams_run_side_by_side(*st[j], *st[i], ok);
                    if (!ok) { for (auto r : rejected) (void)hipStreamDestroy(r); return fail(ATN_ERR_HIP, "stream probe failed"); }
                    clash = !par;
                }
                if (!clash) break;
                if (t == kStreamTries - 1) break;        // out of tries: keep it
                hipStream_t fresh = nullptr;
                if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) != hipSuccess) break;
                rejected.push_back(*st[i]);
                *st[i] = fresh;
                n_stream_swaps++;
            }
        }
        for (auto r : rejected) (void)hipStreamDestroy(r);
        return ATN_OK;
    }
    // The list the new banks are probed against: the main stream, the caller's side stream if it was handed out already (both
    // FIXED: the caller holds their handles), then the spare banks' streams.
    int separate_bank_streams(int n)
    {
        if (n <= bank_streams_checked) return ATN_OK;
        hipStream_t* st[kMaxInFlight + 1];
        int m = 0;
        st[m++] = &stream;
        if (side_stream) st[m++] = &side_stream;
        const int fixed = m;
        for (int i = 1; i < n; i++) st[m++] = &spare[i - 1].stream;
        const int first = fixed + (bank_streams_checked > 1 ? bank_streams_checked - 1 : 0);
        const int rc = separate_streams(st, m, first);
        if (rc) return rc;
        bank_streams_checked = n;
        return ATN_OK;
    }
    // The batch streams a frame forks into when it is rendered in several batches (run_paths).  LAZY: probed the first time
    // a frame of this bank really forks into that many batches -- most contexts (one batch: the refill walk, frames in flight)
    // never fork.  The probe times spin kernels against each other, so it runs on an IDLE device (quiesce: the other banks'
    // frames would read as a clash and cost stream re-creations in the middle of a frame); the count of probed streams is
    // recorded only when the probe succeeded, and a later, larger batch count probes the additional streams.
    int batch_streams_checked = 0;      // bstream[0 .. n-1] of this bank are known to be pairwise concurrent
    int separate_batch_streams(int nb)
    {
        const int n = nb < 3 ? nb : 3;        // the size policy never uses more than two; three when forced
        if (n <= batch_streams_checked || n < 2) return ATN_OK;
        int rc = quiesce();
        if (rc) return rc;
        hipStream_t* st[kMaxBatches];
        for (int i = 0; i < n; i++) st[i] = &bstream[i];
        rc = separate_streams(st, n, batch_streams_checked > 1 ? batch_streams_checked : 1);
        if (rc) return rc;
        batch_streams_checked = n;
        return ATN_OK;
    }

    // the caller's side stream (atn_side_stream): concurrent with every bank stream if a queue is left
    hipStream_t side_stream = nullptr;
    hipStream_t get_side_stream()
    {
        // handed out once: the caller holds the handle (a torch ExternalStream, a C++ app's own work); banks created later
        // are probed against it (separate_bank_streams), it is never replaced
        if (side_stream) return side_stream;
        if (hipSetDevice(device) != hipSuccess) return nullptr;
        hipStream_t banks[kMaxInFlight];
        banks[0] = stream;
        for (int i = 1; i < frames_in_flight; i++) banks[i] = spare[i - 1].stream;
        auto clashes = [&](hipStream_t c, bool& ok) {
            for (int j = 0; j < frames_in_flight; j++) {
                if (!streams_run_side_by_side(banks[j], c, ok) || !ok) return true;
            }
            return false;
        };
        std::vector<hipStream_t> rejected;
        for (int t = 0; t < kStreamTries; t++) {
            if (!side_stream && hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking) != hipSuccess) { side_stream = nullptr; break; }
            bool ok = true;
            const bool clash = env_probe_streams ? clashes(side_stream, ok) : false;
            if (!ok) break;
            if (!clash || t == kStreamTries - 1) break;
            rejected.push_back(side_stream);
            side_stream = nullptr;
        }
        // out of tries (fewer queues than banks + 1): any stream will do
        if (!side_stream && !rejected.empty()) { side_stream = rejected.back(); rejected.pop_back(); }
        for (auto r : rejected) (void)hipStreamDestroy(r);
        return side_stream;
    }

    // every frame in flight finished (all banks' streams idle): required before anything but the next render()
    int quiesce()
    {
        ATN_HIP(hipStreamSynchronize(stream));
        for (int i = 0; i < n_spare_ready; i++) ATN_HIP(hipStreamSynchronize(spare[i].stream));
        if (sv_stream) ATN_HIP(hipStreamSynchronize(sv_stream));
        if (scene_stream) ATN_HIP(hipStreamSynchronize(scene_stream));
        film_pending = false;
        sv_prepare_recorded[0] = sv_prepare_recorded[1] = false;
        return ATN_OK;
    }

    // Scene updates between frames (atn_update_geometry / atn_lbvh_rebuild_list / atn_update_tlas) do not stop the host:
    // they are enqueued on the current bank's stream behind the frames in flight (every bank's "film updated" event is
    // its last read of the scene), their host inputs travel through a pinned staging arena, and the next render on every
    // bank waits for `ev_scene`.  The host only ever waits for the staging arena of the PREVIOUS update to drain.
    // With frames in flight the MUTABLE part of the scene (node image, vertices, triangles, shading records, objects,
    // matrices) exists twice: an update is written into the set no recent frame reads, after copying over what the
    // previous update changed in the other one (its dirty ranges, device to device), and only frames enqueued afterwards
    // read it -- so the frames in flight keep running while the next tick's geometry, LBVH and top layer are built.
    // The members below ARE the set being written / read by new frames; `alt` is the other one (allocated at the first
    // update of a scene).  In place, behind all frames, when there is one frame in flight or the caller writes the arrays
    // itself (atn_scene_device_arrays).
    enum SceneBufId { SB_NODES, SB_VTX_POS, SB_VTX_NML, SB_SHADE, SB_MATRICES, SB_TRIS, SB_OBJECTS, SB_COUNT };
    struct SceneRange { int buf; size_t off, bytes; };
    struct SceneSet {
        DevBuf<float4> nodes, vtx_pos, vtx_nml, shade_tris, matrices;
        DevBuf<atn_triangle_param> tris;
        DevBuf<atn_object_param> objects;
        int id = 0;                 // which set this is (banks remember the id their last frame read)
        uint64_t tick = 0;          // the update after which this set was complete
        void release() { nodes.release(); vtx_pos.release(); vtx_nml.release(); shade_tris.release(); matrices.release(); tris.release(); objects.release(); }
    };
    // One set per frame in flight, up to three: the set written by a tick is the one that was current longest ago, so
    // its last readers are (frames in flight) ticks old and have normally finished -- the tick does not wait.
    static constexpr int kMaxSceneSets = 3;
    SceneSet alt[kMaxSceneSets - 1];
    int n_alt = 0;                                  // allocated sets besides the current one
    bool frame_since_update = true, scene_in_place = false;
    int cur_set = 0, bank_scene_set = 0;            // ids; bank_scene_set: the set this bank's last frame read (travels with the bank)
    hipEvent_t ev_read[3] = {};                     // the current bank's (see Bank::ev_read)
    uint64_t cur_tick = 0, tick_counter = 0;
    std::vector<SceneRange> log_now;                // what the updates since the last flip wrote into the current set
    struct TickLog { uint64_t tick; std::vector<SceneRange> ranges; };
    std::vector<TickLog> log_hist;                  // the last kMaxSceneSets - 1 finished ticks

    char* set_ptr(int b)
    {
        switch (b) {
        case SB_NODES: return (char*)nodes.p; case SB_VTX_POS: return (char*)vtx_pos.p; case SB_VTX_NML: return (char*)vtx_nml.p;
        case SB_SHADE: return (char*)shade_tris.p; case SB_MATRICES: return (char*)matrices.p; case SB_TRIS: return (char*)tris.p;
        default: return (char*)objects.p;
        }
    }
    static char* alt_ptr(SceneSet& a, int b)
    {
        switch (b) {
        case SB_NODES: return (char*)a.nodes.p; case SB_VTX_POS: return (char*)a.vtx_pos.p; case SB_VTX_NML: return (char*)a.vtx_nml.p;
        case SB_SHADE: return (char*)a.shade_tris.p; case SB_MATRICES: return (char*)a.matrices.p; case SB_TRIS: return (char*)a.tris.p;
        default: return (char*)a.objects.p;
        }
    }
    size_t set_bytes(int b)
    {
        switch (b) {
        case SB_NODES: return nodes.n * sizeof(float4); case SB_VTX_POS: return vtx_pos.n * sizeof(float4); case SB_VTX_NML: return vtx_nml.n * sizeof(float4);
        case SB_SHADE: return shade_tris.n * sizeof(float4); case SB_MATRICES: return matrices.n * sizeof(float4);
        case SB_TRIS: return tris.n * sizeof(atn_triangle_param); default: return objects.n * sizeof(atn_object_param);
        }
    }
    void log_range(int b, size_t off, size_t bytes) { if (bytes) log_now.push_back(SceneRange{ b, off, bytes }); }
    void drop_alt_set()
    {
        for (int k = 0; k < n_alt; k++) alt[k].release();
        n_alt = 0; log_now.clear(); log_hist.clear();
    }
    void point_scene_at_current_set()
    {
        scene.nodes = nodes.p; scene.tris = tris.p; scene.shade_tris = shade_tris.p; scene.vtx_pos = vtx_pos.p; scene.vtx_nml = vtx_nml.p;
        scene.objects = objects.p; scene.matrices = matrices.p;
    }
    // switch to the set that was current longest ago (see above); everything is enqueued on `upd`
    int flip_scene_set()
    {
        // the tick that just ended (the updates since the last flip) becomes history
        if (!log_now.empty()) {
            log_hist.push_back(TickLog{ cur_tick, std::move(log_now) });
            log_now.clear();
            if (log_hist.size() > (size_t)(kMaxSceneSets - 1)) log_hist.erase(log_hist.begin());
        }
        const int want = (frames_in_flight < kMaxSceneSets ? frames_in_flight : kMaxSceneSets) - 1;     // sets besides the current one
        SceneSet* t = nullptr;
        if (n_alt < want) {
            // a new set: a clone of the current one
            t = &alt[n_alt];
            ATN_HIP(t->nodes.resize(nodes.n)); ATN_HIP(t->vtx_pos.resize(vtx_pos.n)); ATN_HIP(t->vtx_nml.resize(vtx_nml.n));
            ATN_HIP(t->shade_tris.resize(shade_tris.n)); ATN_HIP(t->matrices.resize(matrices.n));
            ATN_HIP(t->tris.resize(tris.n)); ATN_HIP(t->objects.resize(objects.n));
            for (int b = 0; b < SB_COUNT; b++)
                if (set_bytes(b)) ATN_HIP(hipMemcpyAsync(alt_ptr(*t, b), set_ptr(b), set_bytes(b), hipMemcpyDeviceToDevice, upd));
            // ids 0 .. kMaxSceneSets-1, none in use twice
            bool used[kMaxSceneSets] = {};
            used[cur_set] = true;
            for (int k = 0; k < n_alt; k++) used[alt[k].id] = true;
            for (int id = 0; id < kMaxSceneSets; id++) if (!used[id]) { t->id = id; break; }
            t->tick = cur_tick;
            n_alt++;
        }
        else {
            if (n_alt == 0) return ATN_OK;      // (one frame in flight never gets here)
            t = &alt[0];
            for (int k = 1; k < n_alt; k++) if (alt[k].tick < t->tick) t = &alt[k];
            // the frames that still read it: its writers go behind them
            // (per bank and set, not "the bank's last frame": an older frame of a bank may still be running behind a newer
            // one that reads another set)
            for (int i = 0; i < n_spare_ready; i++)
                if (spare[i].ev_read[t->id]) ATN_HIP(hipStreamWaitEvent(upd, spare[i].ev_read[t->id], 0));
            if (ev_read[t->id]) ATN_HIP(hipStreamWaitEvent(upd, ev_read[t->id], 0));
            // what the ticks since its own wrote into the other sets is missing there: the current set has all of it
#ifndef ATN_DEBUG_NO_REPLAY      /* (the tests' negative control: without the replay test_ticks_touching_different_ranges_are_replayed fails) */
            for (const TickLog& L : log_hist)
                if (L.tick > t->tick)
                    for (const SceneRange& r : L.ranges)
                        ATN_HIP(hipMemcpyAsync(alt_ptr(*t, r.buf) + r.off, set_ptr(r.buf) + r.off, r.bytes, hipMemcpyDeviceToDevice, upd));
#endif
        }
        // swap: the target becomes the members, the old current takes its slot
        nodes.swap(t->nodes); vtx_pos.swap(t->vtx_pos); vtx_nml.swap(t->vtx_nml); shade_tris.swap(t->shade_tris);
        matrices.swap(t->matrices); tris.swap(t->tris); objects.swap(t->objects);
        std::swap(cur_set, t->id);
        t->tick = cur_tick;                 // the old current set is complete up to the tick that just ended
        cur_tick = ++tick_counter;          // the tick being written now
        point_scene_at_current_set();
        return ATN_OK;
    }

    hipEvent_t ev_scene = nullptr, ev_stage = nullptr;
    hipStream_t scene_stream = nullptr;             // updates run here when frames are in flight: not behind the newest frame
    hipStream_t upd = nullptr;                      // the stream of the update in progress (scene_stream, or `stream` with one frame in flight)
    uint64_t scene_epoch = 0, bank_epoch = 0;       // bank_epoch travels with the bank (swap_bank)
    bool stage_busy = false;
    char* stage_p = nullptr;
    size_t stage_cap = 0, stage_used = 0;

    int begin_scene_update(size_t stage_bytes)
    {
        if (!ev_scene) { ATN_HIP(hipEventCreateWithFlags(&ev_scene, hipEventDisableTiming)); ATN_HIP(hipEventCreateWithFlags(&ev_stage, hipEventDisableTiming)); }
        if (stage_busy) { ATN_HIP(hipEventSynchronize(ev_stage)); stage_busy = false; }       // the previous update's copies
        stage_used = 0;
        if (stage_bytes > stage_cap) {
            if (stage_p) ATN_HIP(hipHostFree(stage_p));
            stage_p = nullptr; stage_cap = 0;
            const size_t cap = stage_bytes + stage_bytes / 2 + 4096;
            ATN_HIP(hipHostMalloc((void**)&stage_p, cap, hipHostMallocDefault));
            stage_cap = cap;
        }
        if (frames_in_flight > 1 && !scene_stream) ATN_HIP(hipStreamCreateWithFlags(&scene_stream, hipStreamNonBlocking));
        upd = frames_in_flight > 1 ? scene_stream : stream;
        if (frames_in_flight > 1 && n_spare_ready > 0 && !scene_in_place) {
            // a frame may be reading the current set: write the other one.  (Several updates in a row -- geometry, LBVH,
            // top layer -- flip once: no frame has been enqueued in between.)
            if (frame_since_update) { int r = flip_scene_set(); if (r) return r; frame_since_update = false; }
            return ATN_OK;
        }
        if (n_alt > 0 && frames_in_flight <= 1) drop_alt_set();   // written in place and unlogged: spare sets would miss this update (set_frames_in_flight quiesced)
        log_now.clear();        // (no other set to replay into)
        // in place, behind every frame in flight: a bank's ev_gather is recorded after its last kernel that reads the scene
        // (the filter stream of pipelined SVGF frames never reads the scene)
        if (upd != stream) {
            for (int i = 0; i < n_spare_ready; i++) if (spare[i].ev_gather) ATN_HIP(hipStreamWaitEvent(upd, spare[i].ev_gather, 0));
            if (ev_gather) ATN_HIP(hipStreamWaitEvent(upd, ev_gather, 0));
        }
        return ATN_OK;
    }
    // host bytes -> device, through the arena (the caller's memory is free again when this returns)
    int stage_copy(void* dst_dev, const void* src_host, size_t bytes)
    {
        if (!bytes) return ATN_OK;
        const size_t off = (stage_used + 63) & ~(size_t)63;
        if (off + bytes > stage_cap) return fail(ATN_ERR_INVALID_ARG, "staging arena too small (internal)");
        std::memcpy(stage_p + off, src_host, bytes);
        stage_used = off + bytes;
        ATN_HIP(hipMemcpyAsync(dst_dev, stage_p + off, bytes, hipMemcpyHostToDevice, upd));
        return ATN_OK;
    }
    int end_scene_update()
    {
        if (stage_used) { ATN_HIP(hipEventRecord(ev_stage, upd)); stage_busy = true; }
        ATN_HIP(hipEventRecord(ev_scene, upd));
        scene_epoch++;
        if (upd == stream) bank_epoch = scene_epoch;       // this bank's stream is already behind the update
        return ATN_OK;
    }
    // a frame's last read of the scene is behind it on its stream
    int record_scene_read()
    {
        if (frames_in_flight <= 1) return ATN_OK;
        hipEvent_t& e = ev_read[bank_scene_set];
        if (!e) ATN_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ATN_HIP(hipEventRecord(e, stream));
        return ATN_OK;
    }
    // first thing a frame does on its bank's stream
    int wait_scene_epoch()
    {
        if (bank_epoch != scene_epoch && ev_scene) ATN_HIP(hipStreamWaitEvent(stream, ev_scene, 0));
        bank_epoch = scene_epoch;
        bank_scene_set = cur_set;
        frame_since_update = true;
        return ATN_OK;
    }

    // profiling
    float k_ms[ATN_K_COUNT] = {};
    uint32_t k_launches[ATN_K_COUNT] = {};
    std::vector<hipEvent_t> ev_pool;
    struct Span { int kind; size_t e0, e1; };
    static constexpr size_t kMaxSpans = 8192;
    std::vector<Span> spans;
    size_t ev_used = 0;
    hipStream_t prof_stream = nullptr;
    uint64_t host_stats[8] = {};

    int fail(int code, const std::string& msg) { last_error = msg; return code; }

    int init(int device_ordinal)
    {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0) return fail(ATN_ERR_NO_DEVICE, "no HIP device available (libaten_amd has no CPU fallback)");
        if (device_ordinal < 0 || device_ordinal >= n) return fail(ATN_ERR_INVALID_ARG, "device ordinal out of range");
        device = device_ordinal;
        ATN_HIP(hipSetDevice(device));
        ATN_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        for (int k = 0; k < kMaxBatches; k++) {
            ATN_HIP(hipStreamCreateWithFlags(&bstream[k], hipStreamNonBlocking));
            ATN_HIP(hipEventCreateWithFlags(&ev_join[k], hipEventDisableTiming));
        }
        ATN_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        ATN_HIP(hipEventCreateWithFlags(&ev_gather, hipEventDisableTiming));
        // environment switches (README.md "Environment variables": which are supported controls, which are experiment hooks);
        // read once here, never inside a frame
        if (const char* e = std::getenv("ATEN_AMD_FUSE")) fuse_traces = e[0] != '0';
        if (const char* e = std::getenv("ATEN_AMD_SVGF_ATROUS4")) env_atrous4 = e[0] != '0';     // 0: the one-pixel-per-thread a-trous kernel
        if (const char* e = std::getenv("ATEN_AMD_TRACE_BLOCKS")) env_trace_blocks = (uint32_t)std::atoi(e);
        if (const char* e = std::getenv("ATEN_AMD_SHADE_WAVES")) { const int v = std::atoi(e); if (v == 4 || v == 5) env_shade_waves = v; }   // else: by frames in flight
        if (const char* e = std::getenv("ATEN_AMD_SHADE_ITEMS")) { const int v = std::atoi(e); if (v >= 1 && v <= kChunkItems) env_shade_items = v; }
        if (const char* e = std::getenv("ATEN_AMD_LDS_NODES")) env_lds_nodes = std::atoi(e) != 0;
        if (const char* e = std::getenv("ATEN_AMD_ANYHIT_TWIN")) env_anyhit_twin = std::max(0, std::min(2, std::atoi(e)));
        if (const char* e = std::getenv("ATEN_AMD_ANYHIT_TWIN_DIRS")) env_anyhit_twin_dirs = std::atoi(e) == 1 ? 1 : 8;
        if (const char* e = std::getenv("ATEN_AMD_NODE_LAYOUT")) opt_node_layout = std::atoi(e) != 0;     // 0: bottom-level records in walk order
        if (const char* e = std::getenv("ATEN_AMD_PLANAR_LIGHTS")) opt_planar_lights = std::atoi(e) != 0; // 0: shadow rays towards area lights always walk to their closest hit
        if (const char* e = std::getenv("ATEN_AMD_PROBE_STREAMS")) env_probe_streams = std::atoi(e) != 0;
        if (const char* e = std::getenv("ATEN_AMD_TRACE")) { env_flavour = e[0] == 'r' ? 1 : 0; }   // 'r'efill / 's'imple
        if (const char* e = std::getenv("ATEN_AMD_SIMPLE_BLOCK")) { const int v = std::atoi(e); if (v == 64 || v == 128 || v == 256) simple_block = (uint32_t)v; }
        if (const char* e = std::getenv("ATEN_AMD_BATCHES")) {
            n_batches = std::atoi(e);
            batches_forced = true;
            if (n_batches < 1) n_batches = 1;
            if (n_batches > kMaxBatches) n_batches = kMaxBatches;
        }
        // (the streams a frame's batches run on are probed lazily: separate_batch_streams, run_paths)
        return ATN_OK;
    }

    ~PathTracing()
    {
        for (auto& e : ev_pool) (void)hipEventDestroy(e);
        for (int k = 0; k < kMaxBatches; k++) {
            if (ev_join[k]) (void)hipEventDestroy(ev_join[k]);
            if (bstream[k]) (void)hipStreamDestroy(bstream[k]);
        }
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_gather) (void)hipEventDestroy(ev_gather);
        for (auto& e : sv_ev_prepare) if (e) (void)hipEventDestroy(e);
        if (ev_film) (void)hipEventDestroy(ev_film);
        if (sv_stream) (void)hipStreamDestroy(sv_stream);
        for (int i = 0; i < n_spare_ready; i++) {
            Bank& b = spare[i];
            for (int k = 0; k < kMaxBatches; k++) {
                if (b.ev_join[k]) (void)hipEventDestroy(b.ev_join[k]);
                if (b.bstream[k]) (void)hipStreamDestroy(b.bstream[k]);
            }
            if (b.ev_fork) (void)hipEventDestroy(b.ev_fork);
            if (b.ev_gather) (void)hipEventDestroy(b.ev_gather);
            for (auto& e : b.ev_read) if (e) (void)hipEventDestroy(e);
            if (b.stream) (void)hipStreamDestroy(b.stream);
        }
        if (stream) (void)hipStreamDestroy(stream);
        if (side_stream) (void)hipStreamDestroy(side_stream);
        if (scene_stream) (void)hipStreamDestroy(scene_stream);
        for (auto& e : ev_read) if (e) (void)hipEventDestroy(e);
        if (ev_scene) (void)hipEventDestroy(ev_scene);
        if (ev_stage) (void)hipEventDestroy(ev_stage);
        if (stage_p) (void)hipHostFree(stage_p);
    }

    // ≙ idaten::Renderer::UpdateSceneData, src/libidaten/kernel/renderer.cpp:12-131
    int UpdateSceneData(const atn_scene_desc* s)
    {
        ATN_HIP(hipSetDevice(device));
        { int q = quiesce(); if (q) return q; }
        drop_alt_set(); cur_set = 0; scene_in_place = false; frame_since_update = true;
        HostSceneImage img;
        std::string err;
        // (the upload options are context state: the environment was read ONCE, when the context was created; atn_set_upload_options
        // changes them -- every shard of a node, every re-upload gets the same layout whatever happened to the process environment since)
        const int layout_top = opt_node_layout ? kLayoutTopLevels : 0;
        const bool planar_lights = opt_planar_lights;
        if (!build_host_image(img, s, err, env_anyhit_twin, env_anyhit_twin_dirs, layout_top, planar_lights)) return fail(ATN_ERR_UNSUPPORTED, err);
        n_planar_lights = 0;
        for (const atn_light_param& l : img.lights) n_planar_lights += l._pad != 0 ? 1u : 0u;
        collect_planar_certs(s, img.lights);
        ATN_HIP(nodes.upload(img.nodes, stream));
        ATN_HIP(tris.upload(img.tris, stream));
        ATN_HIP(vtx_pos.upload(img.vtx_pos, stream));
        ATN_HIP(vtx_nml.upload(img.vtx_nml, stream));
        ATN_HIP(objects.upload(img.objects, stream));
        ATN_HIP(matrices.upload(img.matrices, stream));
        ATN_HIP(materials.upload(img.materials, stream));
        ATN_HIP(carpaint.upload(img.carpaint, stream));
        ATN_HIP(toon.upload(img.toon, stream));
        ATN_HIP(npr_lights.upload(img.npr_lights, stream));
        ATN_HIP(screen_shadow.upload(img.screen_shadow, stream));
        ATN_HIP(lights.upload(img.lights, stream));
        ATN_HIP(light_plane.upload(img.light_plane, stream));
        ATN_HIP(texels.upload(img.texels, stream));
        ATN_HIP(texels8.upload(img.texels8, stream));
        ATN_HIP(textures.upload(img.textures, stream));
        ATN_HIP(shade_tris.resize((size_t)kShadeTriQuads * (img.tris.size() ? img.tris.size() : 1)));
        if (!img.tris.empty())
            hipLaunchKernelGGL(k_pack_shade_tris, dim3(((uint32_t)img.tris.size() + 255) / 256), dim3(256), 0, stream, (const atn_triangle_param*)tris.p,
                               (const float4*)vtx_pos.p, (const float4*)vtx_nml.p, 0u, (uint32_t)img.tris.size(), shade_tris.p);
        ATN_HIP(hipGetLastError());
        ATN_HIP(hipStreamSynchronize(stream));      // `img` is pageable host memory
        scene = img.params;
        scene.nodes = nodes.p; scene.tris = tris.p; scene.shade_tris = shade_tris.p; scene.vtx_pos = vtx_pos.p; scene.vtx_nml = vtx_nml.p;
        scene.objects = objects.p; scene.matrices = matrices.p; scene.materials = materials.p; scene.carpaint = carpaint.p;
        scene.toon = toon.p; scene.npr_lights = npr_lights.p; scene.screen_shadow = img.screen_shadow.empty() ? nullptr : screen_shadow.p;
        scene.lights = lights.p; scene.light_plane = light_plane.p; scene.texels = texels.p; scene.texels8 = texels8.p; scene.textures = textures.p;
        scene.mtx_quads = (uint32_t)img.matrices.size();
        has_scene = true;
        env_host.clear(); env_w = env_h = 0; ibl_tables_ready = false;
        {
            const int32_t ei = s->config.bg.envmap_tex_idx;
            if (ei >= 0 && (uint32_t)ei < s->n_textures && s->config.bg.enable_env_map) {
                const atn_texture_desc& t = s->textures[ei];
                env_host.assign(t.texels, t.texels + (size_t)t.width * t.height);
                env_w = t.width; env_h = t.height; env_multiplyer = s->config.bg.multiplyer;
            }
        }
        {
            int orc = apply_sampling_options();
            if (orc) return orc;
        }
        list_root_link = img.list_root_link; list_twin_delta = img.list_twin_delta; tlas_refs = img.tlas_refs;
        list_base = img.list_root; list_bytes = img.list_bytes; list_tri_leaves = img.list_tri_leaves; list_inner = img.list_inner;
        n_scene_tris = s->n_triangles; n_scene_vtx = s->n_vertices; n_scene_mtrls = s->n_materials;
        top_base = img.list_root[0];        // byte offset of the top layer's first record (the image's tail)
        n_bottom_nodes = img.n_nodes - s->bvh_lists[0].count;
        n_host_matrices = s->n_matrices;
        host_matrices.assign(s->matrices, s->matrices + s->n_matrices);     // (update_tlas without matrices still recognises identity instances)
        tree_is_deep = img.n_nodes >= kRefillMinNodes;
        use_refill = tree_is_deep;
        flavour_forced = false;
        if (env_flavour >= 0) { use_refill = env_flavour == 1; flavour_forced = true; }
        return ATN_OK;
    }

    // Optional samplers, both OFF by default because they leave the sample stream of aten::PathTracing (the parity
    // path): the IBL light sampled from ImageBasedLight::preCompute's tables (light/ibl.cpp:10-118,180-230) and
    // texture::AtWithBilinear (image/texture.cpp:77-125) instead of texture::at for every texture lookup.
    int apply_sampling_options()
    {
        scene.tex_bilinear = opt_tex_bilinear;
        scene.ibl_importance = 0;
        scene.ibl_cdf_v = nullptr; scene.ibl_cdf_u = nullptr; scene.ibl_w = 0; scene.ibl_h = 0;
        if (!opt_ibl_importance || env_host.empty()) return ATN_OK;
        if (!ibl_tables_ready) {
            IblTables t;
            const int32_t w = env_w, h = env_h;
            const atn_vec4* tex = env_host.data();
            const float mul = env_multiplyer;
            ibl_precompute(t, w, h, [&](int32_t x, int32_t y, float& r, float& g, float& b) {
                // SampleFromUVWithTexture(u, v) = texture::at(u, v) * multiplyer at the texel centre (ibl.cpp:57-60)
                const float u = (float)(x + 0.5) / w, v = (float)(y + 0.5) / h;
                int32_t iu = (int32_t)(u * (float)(w - 1)), iv = (int32_t)(v * (float)(h - 1));
                iu = iu < 0 ? 0 : (iu > w - 1 ? w - 1 : iu); iv = iv < 0 ? 0 : (iv > h - 1 ? h - 1 : iv);
                const atn_vec4& c = tex[(size_t)iv * w + iu];
                r = c.x * mul; g = c.y * mul; b = c.z * mul;
            });
            ATN_HIP(ibl_cdf_v.upload(t.cdf_v, stream));
            ATN_HIP(ibl_cdf_u.upload(t.cdf_u, stream));
            ATN_HIP(hipStreamSynchronize(stream));
            ibl_tables_ready = true;
        }
        scene.ibl_cdf_v = ibl_cdf_v.p; scene.ibl_cdf_u = ibl_cdf_u.p; scene.ibl_w = env_w; scene.ibl_h = env_h;
        scene.ibl_importance = 1;
        return ATN_OK;
    }
    int set_sampling_options(int32_t ibl_importance, int32_t tex_bilinear)
    {
        ATN_HIP(hipSetDevice(device));
        { int q = quiesce(); if (q) return q; }
        opt_ibl_importance = ibl_importance ? 1 : 0;
        opt_tex_bilinear = tex_bilinear ? 1 : 0;
        return has_scene ? apply_sampling_options() : ATN_OK;
    }

    // ≙ idaten::Renderer::updateBVH, src/libidaten/kernel/renderer.cpp:133-153: new object parameters and matrices
    // plus a rebuilt top layer; the bottom-level lists stay where they are in HBM.
    int updateBVH(const atn_object_param* objs, uint32_t n_objs, const atn_mat4* mtxs, uint32_t n_mtxs,
                  const atn_bvh_node* top, uint32_t n_top)
    {
        if (!has_scene) return fail(ATN_ERR_NO_SCENE, "atn_upload_scene has not been called");
        if (!objs || n_objs == 0 || !top || n_top == 0) return fail(ATN_ERR_INVALID_ARG, "empty object or top-layer array");
        ATN_HIP(hipSetDevice(device));
        if ((uint64_t)top_base + (uint64_t)n_top * kInnerBytes >= (1ull << 31)) return fail(ATN_ERR_UNSUPPORTED, "too many BVH nodes for 31-bit byte-offset links");
        if (n_mtxs && !mtxs) return fail(ATN_ERR_INVALID_ARG, "null matrix array");
        {
            std::string rerr;
            if (!validate_ranges(objs, n_objs, n_mtxs ? n_mtxs : n_host_matrices, nullptr, rerr)) return fail(ATN_ERR_UNSUPPORTED, rerr);
            for (uint32_t i = 0; i < n_objs; i++)
                if (objs[i].light_id >= scene.n_lights) return fail(ATN_ERR_UNSUPPORTED, "object light id out of range");
        }
        // the new top layer: every top-layer record is kInnerBytes long, laid out in walk order at the image's tail
        ListLayout lay;
        std::string err;
        if (!analyse_list(lay, top, n_top, true, err)) return fail(ATN_ERR_UNSUPPORTED, err);
        for (uint32_t j = 0; j < lay.order.size(); j++) {
            if (lay.kind[j] == KIND_TRI) return fail(ATN_ERR_UNSUPPORTED, "triangle leaves in this list need a full scene upload");
            lay.offset[j] = top_base + j * kInnerBytes;
        }
        ListEmitCtx c;
        c.objects = objs; c.n_objects = n_objs; c.n_matrices = n_mtxs ? n_mtxs : n_host_matrices;
        c.matrices = n_mtxs ? mtxs : (host_matrices.size() == n_host_matrices ? host_matrices.data() : nullptr);
        c.list_root_link = list_root_link.data(); c.n_lists = (uint32_t)list_root_link.size();
        c.list_twin_delta = list_twin_delta.size() == list_root_link.size() ? list_twin_delta.data() : nullptr;
        std::vector<HostSceneImage::TlasRef> new_refs;
        c.tlas_refs = &new_refs;
        const size_t top_bytes = lay.order.size() * (size_t)kInnerBytes;
        std::vector<float4> rec(top_bytes / 16 + 1, make_float4(0, 0, 0, 0));
        int32_t root = kLinkEnd;
        uint64_t counts[3] = { 0, 0, 0 };
        if (!emit_list(reinterpret_cast<char*>(rec.data()), lay, top, c, root, counts, err, top_base)) return fail(ATN_ERR_UNSUPPORTED, err);
        std::vector<float4> mv;
        if (n_mtxs) {
            mv.resize((size_t)n_mtxs * 4);
            for (uint32_t i = 0; i < n_mtxs; i++)
                for (int r = 0; r < 4; r++)
                    mv[4 * (size_t)i + r] = make_float4(mtxs[i].m[r][0], mtxs[i].m[r][1], mtxs[i].m[r][2], mtxs[i].m[r][3]);
        }
        const size_t need = ((size_t)top_base + top_bytes + 15) / 16;
        const size_t obj_bytes = (size_t)n_objs * sizeof(atn_object_param), mtx_bytes = mv.size() * sizeof(float4);
        // Buffers that must grow are replaced behind a full stop (rare: the usual tick keeps every count); otherwise the
        // update is enqueued behind the frames in flight and the host goes on.
        const bool grow = need > nodes.n || n_objs > objects.n || mv.size() > matrices.n;
        if (grow) {
            { int q = quiesce(); if (q) return q; }
            drop_alt_set();         // re-cloned from the grown buffers at the next flip
            if (need > nodes.n) {
                // keep the bottom-level lists (device-to-device), drop the old top layer
                float4* bigger = nullptr;
                ATN_HIP(hipMalloc((void**)&bigger, need * sizeof(float4)));
                hipError_t e = hipMemcpyAsync(bigger, nodes.p, (size_t)top_base, hipMemcpyDeviceToDevice, stream);
                if (e == hipSuccess) e = hipStreamSynchronize(stream);
                if (e != hipSuccess) { (void)hipFree(bigger); return fail(ATN_ERR_HIP, hipGetErrorString(e)); }
                nodes.release();
                nodes.p = bigger; nodes.n = need;
            }
            if (n_objs > objects.n) ATN_HIP(objects.resize(n_objs));
            if (mv.size() > matrices.n) ATN_HIP(matrices.resize(mv.size()));
        }
        { int r = begin_scene_update(top_bytes + obj_bytes + mtx_bytes + 512); if (r) return r; }
        { int r = stage_copy(reinterpret_cast<char*>(nodes.p) + top_base, rec.data(), top_bytes); if (r) return r; log_range(SB_NODES, top_base, top_bytes); }
        { int r = stage_copy(objects.p, objs, obj_bytes); if (r) return r; log_range(SB_OBJECTS, 0, obj_bytes); }
        if (n_mtxs) { int r = stage_copy(matrices.p, mv.data(), mtx_bytes); if (r) return r; log_range(SB_MATRICES, 0, mtx_bytes); n_host_matrices = n_mtxs; host_matrices.assign(mtxs, mtxs + n_mtxs); }
        { int r = end_scene_update(); if (r) return r; }
        list_root_link[0] = root;
        // a light's instance may have a new matrix, or point at another object / matrix (an objs-only update too): the flags hold only
        // if every planar light's records came back byte for byte
        if (scene.planar_lights && !planar_certs_survive_objects(objs, n_objs, mtxs, n_mtxs)) scene.planar_lights = 0;
        tlas_refs.swap(new_refs);
        scene.root_link = root;
        fill_root_direct(scene, rec.data(), top_base, n_mtxs ? mtxs : (host_matrices.size() == n_host_matrices ? host_matrices.data() : nullptr), n_mtxs ? n_mtxs : n_host_matrices);
        scene.node_bytes = (uint32_t)(top_base + top_bytes);
        scene.ident_row = c.ident_row;
        if (n_mtxs) scene.mtx_quads = (uint32_t)mv.size();
        point_scene_at_current_set();
        tree_is_deep = n_bottom_nodes + n_top >= kRefillMinNodes;
        if (!flavour_forced) use_refill = tree_is_deep;
        return ATN_OK;
    }

    // ---------------------------------------------------------------------------------------------------------------
    // Dynamic geometry: ≙ the per-tick sequence of src/deformation_renderer/main.cpp:636-710
    //   skinning -> LBVHBuilder::build into the renderer's node list -> Renderer::updateGeometry -> Renderer::updateBVH.

    // ≙ idaten::Renderer::updateGeometry, src/libidaten/kernel/renderer.cpp:155-215: overwrite a range of the scene's
    // vertices and triangles.  (The BVH records of lists over these triangles hold hoisted vertex data: rebuild them.)
    int updateGeometry(const atn_vec4* pos, const atn_vec4* nml, uint32_t n_vtx, uint32_t vtx_offset,
                       const atn_triangle_param* tr, uint32_t n_tr, uint32_t tri_offset)
    {
        if (!has_scene) return fail(ATN_ERR_NO_SCENE, "atn_upload_scene has not been called");
        if ((n_vtx && !pos && !nml) || (n_tr && !tr)) return fail(ATN_ERR_INVALID_ARG, "null geometry array");
        if ((uint64_t)vtx_offset + n_vtx > n_scene_vtx) return fail(ATN_ERR_INVALID_ARG, "vertex range outside the uploaded scene");
        if ((uint64_t)tri_offset + n_tr > n_scene_tris) return fail(ATN_ERR_INVALID_ARG, "triangle range outside the uploaded scene");
        for (uint32_t i = 0; i < n_tr; i++) {
            if (tr[i].mtrlid >= 0 && (uint32_t)tr[i].mtrlid >= n_scene_mtrls) return fail(ATN_ERR_UNSUPPORTED, "triangle material id out of range");
            for (int v = 0; v < 3; v++)
                if (tr[i].idx[v] < 0 || (uint32_t)tr[i].idx[v] >= n_scene_vtx) return fail(ATN_ERR_UNSUPPORTED, "triangle vertex index out of range");
        }
        ATN_HIP(hipSetDevice(device));
        // a light's vertices or triangles may be among these: then its shadow rays walk to their closest hit from here on
        if (scene.planar_lights && !planar_certs_survive_geometry(vtx_offset, n_vtx, tri_offset, n_tr)) scene.planar_lights = 0;
        const size_t vb = (size_t)n_vtx * sizeof(float4), tb = (size_t)n_tr * sizeof(atn_triangle_param);
        { int r = begin_scene_update((pos ? vb : 0) + (nml ? vb : 0) + tb + 256); if (r) return r; }
        if (n_vtx && pos) { int r = stage_copy(vtx_pos.p + vtx_offset, pos, vb); if (r) return r; log_range(SB_VTX_POS, (size_t)vtx_offset * sizeof(float4), vb); }
        if (n_vtx && nml) { int r = stage_copy(vtx_nml.p + vtx_offset, nml, vb); if (r) return r; log_range(SB_VTX_NML, (size_t)vtx_offset * sizeof(float4), vb); }
        if (n_tr) { int r = stage_copy(tris.p + tri_offset, tr, tb); if (r) return r; log_range(SB_TRIS, (size_t)tri_offset * sizeof(atn_triangle_param), tb); }
        // which triangles use the new vertices is not known here: repack every shading record (a copy of the scene arrays)
        { int r = repack_shade_tris(0, n_scene_tris); if (r) return r; }
        return end_scene_update();
    }

    int repack_shade_tris(uint32_t first, uint32_t count)
    {
        if (!count) return ATN_OK;
        hipLaunchKernelGGL(k_pack_shade_tris, dim3((count + 255) / 256), dim3(256), 0, upd, (const atn_triangle_param*)tris.p,
                           (const float4*)vtx_pos.p, (const float4*)vtx_nml.p, first, count, shade_tris.p);
        ATN_HIP(hipGetLastError());
        log_range(SB_SHADE, (size_t)first * kShadeTriQuads * sizeof(float4), (size_t)count * kShadeTriQuads * sizeof(float4));
        return ATN_OK;
    }

    int lbvh_reserve(uint32_t n)
    {
        const uint32_t rounds = radix_rounds(n), tile = kSortThreads * rounds, nb = (n + tile - 1) / tile, nn = 2 * n - 1;
        for (int k = 0; k < 2; k++) {
            if (lb.codes[k].n < n) ATN_HIP(lb.codes[k].resize(n));
            if (lb.indices[k].n < n) ATN_HIP(lb.indices[k].resize(n));
        }
        if (lb.counts.n < 256u * nb) ATN_HIP(lb.counts.resize(256u * nb));
        if (lb.totals.n < 256u) ATN_HIP(lb.totals.resize(256u));
        if (lb.arrived.n < n) ATN_HIP(lb.arrived.resize(n));
        if (lb.offs.n < nn) ATN_HIP(lb.offs.resize(nn));
        if (lb.left.n < nn) ATN_HIP(lb.left.resize(nn));
        if (lb.right.n < nn) ATN_HIP(lb.right.resize(nn));
        if (lb.parent.n < nn) ATN_HIP(lb.parent.resize(nn));
        if (lb.first.n < n) ATN_HIP(lb.first.resize(n));
        if (lb.last.n < n) ATN_HIP(lb.last.resize(n));
        if (lb.ref_nodes.n < nn) ATN_HIP(lb.ref_nodes.resize(nn));
        if (lb.pre.n < n) ATN_HIP(lb.pre.resize(n));
        if (lb.suf.n < n) ATN_HIP(lb.suf.resize(n));
        { const uint32_t ns = (n + kBoundsBlock * kBoundsSuper - 1) / (kBoundsBlock * kBoundsSuper); if (lb.sup.n < ns) ATN_HIP(lb.sup.resize(ns)); }
        return ATN_OK;
    }

    // Morton codes -> sort -> hierarchy -> links -> boxes, all enqueued on `stream`; the tree is left in lb.ref_nodes in
    // the reference's node order.  `tr` points at the first of the n triangles, `vtx` at the vertex array.
    int lbvh_enqueue(const atn_triangle_param* tr, uint32_t n, int32_t tri_id_offset, const float* bmin, const float* bmax,
                     const float4* vtx, int32_t vtx_offset, uint32_t image_base, hipStream_t st)
    {
        { int r = lbvh_reserve(n); if (r) return r; }
        const uint32_t rounds = radix_rounds(n), tile = kSortThreads * rounds, nb = (n + tile - 1) / tile, nn = 2 * n - 1;
        const dim3 b256(256);
        f3 mn, mx;
        mn.x = bmin[0]; mn.y = bmin[1]; mn.z = bmin[2]; mx.x = bmax[0]; mx.y = bmax[1]; mx.z = bmax[2];
        hipLaunchKernelGGL(k_lbvh_morton, dim3((n + 255) / 256), b256, 0, st, tr, vtx, vtx_offset, n, mn, mx, lb.codes[0].p, lb.indices[0].p);
        for (uint32_t pass = 0; pass < 4; pass++) {
            const int in = pass & 1, out = in ^ 1;
            hipLaunchKernelGGL(k_radix_count, dim3(nb), dim3(kSortThreads), 0, st, (const uint32_t*)lb.codes[in].p, n, pass * 8u, lb.counts.p, nb, rounds);
            hipLaunchKernelGGL(k_radix_scan, dim3(256), dim3(64), 0, st, lb.counts.p, nb, lb.totals.p);
            hipLaunchKernelGGL(k_radix_scatter, dim3(nb), dim3(kSortThreads), 0, st, (const uint32_t*)lb.codes[in].p, (const uint32_t*)lb.indices[in].p,
                               lb.codes[out].p, lb.indices[out].p, n, pass * 8u, (const uint32_t*)lb.counts.p, nb, (const uint32_t*)lb.totals.p, rounds);
        }
        LbvhTopo t{ lb.left.p, lb.right.p, lb.parent.p, lb.first.p, lb.last.p };
        hipLaunchKernelGGL(k_lbvh_hierarchy, dim3((n + 255) / 256), b256, 0, st, (const uint32_t*)lb.codes[0].p, n, t, lb.arrived.p);
        hipLaunchKernelGGL(k_lbvh_links, dim3((nn + 255) / 256), b256, 0, st, n, tri_id_offset, t, (const uint32_t*)lb.indices[0].p, lb.ref_nodes.p,
                           image_base, lb.offs.p);
        {
            const uint32_t nblk = (n + kBoundsBlock - 1) / kBoundsBlock, nsup = (nblk + kBoundsSuper - 1) / kBoundsSuper;
            hipLaunchKernelGGL(k_lbvh_bounds_block, dim3(nblk), dim3(kBoundsBlock), 0, st, n, t, (const uint32_t*)lb.indices[0].p, tr, vtx, vtx_offset,
                               lb.ref_nodes.p, lb.arrived.p, lb.pre.p, lb.suf.p);
            hipLaunchKernelGGL(k_lbvh_bounds_super, dim3(nsup), dim3(64), 0, st, n, (const LbvhBox*)lb.pre.p, lb.sup.p);
            hipLaunchKernelGGL(k_lbvh_bounds_cross, dim3((n + 255) / 256), b256, 0, st, n, t, (const LbvhBox*)lb.pre.p, (const LbvhBox*)lb.suf.p,
                               (const LbvhBox*)lb.sup.p, lb.ref_nodes.p);
        }
        ATN_HIP(hipGetLastError());
        return ATN_OK;
    }

    // ≙ idaten::LBVHBuilder::build(dst, std::vector<TriangleParameter>&, ..., threadedBvhNodes), LBVHBuilder.cu:812-833:
    // host arrays in, the ThreadedBvhNode array out, in the reference's node order.
    int lbvh_build(const atn_triangle_param* tr, uint32_t n, int32_t tri_id_offset, const float* bmin, const float* bmax,
                   const atn_vec4* vtx, uint32_t n_vtx, int32_t vtx_offset, atn_bvh_node* out, uint32_t* out_codes, uint32_t* out_indices)
    {
        if (!tr || !vtx || !out || !bmin || !bmax) return fail(ATN_ERR_INVALID_ARG, "null argument");
        if (n < 2) return fail(ATN_ERR_INVALID_ARG, "an LBVH needs at least two triangles");
        if (n > kLbvhMaxTris) return fail(ATN_ERR_UNSUPPORTED, "too many triangles: node indices are stored as floats");
        for (uint32_t i = 0; i < n; i++)
            for (int v = 0; v < 3; v++) {
                const int64_t j = (int64_t)tr[i].idx[v] + vtx_offset;
                if (j < 0 || j >= (int64_t)n_vtx) return fail(ATN_ERR_INVALID_ARG, "triangle vertex index out of range");
            }
        ATN_HIP(hipSetDevice(device));
        ATN_HIP(hipStreamSynchronize(stream));
        if (lb.tris.n < n) ATN_HIP(lb.tris.resize(n));
        if (lb.vtx.n < n_vtx) ATN_HIP(lb.vtx.resize(n_vtx));
        ATN_HIP(hipMemcpyAsync(lb.tris.p, tr, (size_t)n * sizeof(atn_triangle_param), hipMemcpyHostToDevice, stream));
        ATN_HIP(hipMemcpyAsync(lb.vtx.p, vtx, (size_t)n_vtx * sizeof(float4), hipMemcpyHostToDevice, stream));
        { int r = lbvh_enqueue(lb.tris.p, n, tri_id_offset, bmin, bmax, lb.vtx.p, vtx_offset, 0, stream); if (r) return r; }
        ATN_HIP(hipMemcpyAsync(out, lb.ref_nodes.p, (size_t)(2 * n - 1) * sizeof(atn_bvh_node), hipMemcpyDeviceToHost, stream));
        if (out_codes) ATN_HIP(hipMemcpyAsync(out_codes, lb.codes[0].p, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
        if (out_indices) ATN_HIP(hipMemcpyAsync(out_indices, lb.indices[0].p, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
        ATN_HIP(hipStreamSynchronize(stream));
        return ATN_OK;
    }

    // ≙ lbvh_.build(nodes[deformPos], tris, tri_offset_, sceneBbox, vtxPos, ...) of deformation_renderer/main.cpp:686-693:
    // rebuild bottom-level list `list` as an LBVH over the scene's triangles [tri_offset, tri_offset + n) as they are on
    // the device NOW, and write its records over the list's region of the node image.  The region keeps its size: an
    // LBVH over n triangles is n - 1 inner records and n triangle leaves, so the list must have been uploaded with one
    // leaf per triangle (any full binary tree over the n triangles, e.g. atns_build_blas's).
    int lbvh_rebuild_list(uint32_t list, uint32_t tri_offset, uint32_t n, const float* bmin, const float* bmax, bool sync)
    {
        if (!has_scene) return fail(ATN_ERR_NO_SCENE, "atn_upload_scene has not been called");
        if (!bmin || !bmax) return fail(ATN_ERR_INVALID_ARG, "null bounding box");
        if (list == 0 || list >= list_base.size()) return fail(ATN_ERR_INVALID_ARG, "not a bottom-level BVH list");
        if (n < 2) return fail(ATN_ERR_INVALID_ARG, "an LBVH needs at least two triangles");
        if (n > kLbvhMaxTris) return fail(ATN_ERR_UNSUPPORTED, "too many triangles: node indices are stored as floats");
        if ((uint64_t)tri_offset + n > n_scene_tris) return fail(ATN_ERR_INVALID_ARG, "triangle range outside the uploaded scene");
        if (list_tri_leaves[list] != n || list_inner[list] != n - 1 || list_bytes[list] != (n - 1) * kInnerBytes + n * kTriLeafBytes)
            return fail(ATN_ERR_UNSUPPORTED, "the list was not uploaded as a binary tree with one leaf per triangle of this range");
        ATN_HIP(hipSetDevice(device));
        // The list's any-hit twins (scene_upload.hpp) are threadings of the tree that is about to be replaced: they are re-threaded on
        // the device from the new tree (lbvh.hpp, k_lbvh_twin_*), into the region the upload gave them; the TLAS leaves' twin words stay.
        const int32_t twin_word = list < list_twin_delta.size() ? list_twin_delta[list] : 0;
        const uint32_t twin_delta = (uint32_t)(twin_word & ~15), twin_dirs = twin_word == 0 ? 0u : ((twin_word & 1) ? 8u : 1u);
        if (twin_word != 0 && twin_delta != list_bytes[list]) return fail(ATN_ERR_UNSUPPORTED, "the list's twins do not lie at the list's own size apart");
        { int r = begin_scene_update(64); if (r) return r; }
        // a caller may have written the scene arrays in place (atn_scene_device_arrays): refresh this mesh's shading records
        { int r = repack_shade_tris(tri_offset, n); if (r) return r; }
        { int r = lbvh_enqueue(tris.p + tri_offset, n, (int32_t)tri_offset, bmin, bmax, vtx_pos.p, 0, list_base[list], upd); if (r) return r; }
        const uint32_t nn = 2 * n - 1;
        hipLaunchKernelGGL(k_lbvh_emit, dim3((nn + 255) / 256), dim3(256), 0, upd, n, (const atn_bvh_node*)lb.ref_nodes.p, (const uint32_t*)lb.offs.p,
                           (const atn_triangle_param*)tris.p, (const float4*)vtx_pos.p, nodes.p);
        ATN_HIP(hipGetLastError());
        log_range(SB_NODES, list_base[list], list_bytes[list]);
        if (twin_dirs) {
            if (lb.flips.n < n) ATN_HIP(lb.flips.resize(n));
            LbvhTopo t{ lb.left.p, lb.right.p, lb.parent.p, lb.first.p, lb.last.p };
            hipLaunchKernelGGL(k_lbvh_twin_flips, dim3((n + 255) / 256), dim3(256), 0, upd, n, t, (const atn_bvh_node*)lb.ref_nodes.p, twin_dirs, lb.flips.p);
            hipLaunchKernelGGL(k_lbvh_twin_emit, dim3((nn + 255) / 256, twin_dirs), dim3(256), 0, upd, n, t, (const atn_bvh_node*)lb.ref_nodes.p,
                               (const uint32_t*)lb.offs.p, (const uint8_t*)lb.flips.p, twin_delta, (const atn_triangle_param*)tris.p, (const float4*)vtx_pos.p, nodes.p);
            ATN_HIP(hipGetLastError());
            log_range(SB_NODES, list_base[list] + twin_delta, (size_t)twin_dirs * twin_delta);
        }
        // the root is node 0 = an inner record at the start of the region: the TLAS leaves' root link (and twin word) stay valid
        { int r = end_scene_update(); if (r) return r; }
        if (sync) ATN_HIP(hipStreamSynchronize(upd));
        return ATN_OK;
    }

    // ≙ idaten::Renderer::updateCamera, renderer.cpp:202-205
    int updateCamera(const atn_camera_param* c)
    {
        if (!c) return fail(ATN_ERR_INVALID_ARG, "null camera");
        camera = *c;
        has_camera = true;
        return ATN_OK;
    }

    int setRandom(const uint32_t* v, uint32_t n)
    {
        if (!v || n == 0) return fail(ATN_ERR_INVALID_ARG, "empty seed array");
        ATN_HIP(hipSetDevice(device));
        { int q = quiesce(); if (q) return q; }
        ATN_HIP(seeds.resize(n));
        ATN_HIP(hipMemcpyAsync(seeds.p, v, (size_t)n * 4, hipMemcpyHostToDevice, stream));
        ATN_HIP(hipStreamSynchronize(stream));
        n_seeds = n;
        return ATN_OK;
    }

    // ≙ aten::getRandom(), src/libaten/sampler/sampler.cpp:20-23: the table as the kernels read it
    int getRandom(uint32_t* out, uint32_t n)
    {
        if (!out || n == 0 || n > n_seeds) return fail(ATN_ERR_INVALID_ARG, "atn_get_random: null output or more entries than the sampler holds");
        ATN_HIP(hipSetDevice(device));
        { int q = quiesce(); if (q) return q; }
        ATN_HIP(hipMemcpyAsync(out, seeds.p, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
        ATN_HIP(hipStreamSynchronize(stream));
        return ATN_OK;
    }

    // ≙ aten::initSampler, src/libaten/sampler/sampler.cpp:8-18
    int initSampler(int32_t w, int32_t h, int32_t seed)
    {
        if (w <= 0 || h <= 0) return fail(ATN_ERR_INVALID_ARG, "bad sampler size");
        std::vector<uint32_t> v((size_t)w * h);
        std::mt19937 src(seed);
        for (auto& x : v) x = (uint32_t)src();
        return setRandom(v.data(), (uint32_t)v.size());
    }

    int ensure_frame(int32_t w, int32_t h, int32_t max_depth)
    {
        const int32_t tx = (w + 7) / 8, ty = (h + 7) / 8;
        const uint32_t n_tiles = (uint32_t)tx * ty;
        const uint32_t tiles_per_rank = (n_tiles + world - 1) / world;
        const uint32_t slots = tiles_per_rank * 64;
        if (slots > kShadowSlotMask) return fail(ATN_ERR_UNSUPPORTED, "more than 2^26 path slots per GPU (shard the screen)");
        if (slots != n_slots) {
            ATN_HIP(ray_o.resize(slots)); ATN_HIP(ray_d.resize(slots)); ATN_HIP(thr.resize(slots));
            ATN_HIP(contrib.resize(slots)); ATN_HIP(isect.resize(slots));
            ATN_HIP(sh_o.resize(slots)); ATN_HIP(sh_d.resize(slots)); ATN_HIP(sh_c.resize(slots));
            ATN_HIP(accum.resize(slots)); ATN_HIP(done.resize(slots));
            ATN_HIP(queue0.resize(slots)); ATN_HIP(queue1.resize(slots)); ATN_HIP(shadow_q.resize(slots));
            ATN_HIP(tile_out.resize(slots));
            n_slots = slots;
        }
        if (w != film_w || h != film_h) {
            if (frames_in_flight > 1) { int qrc = quiesce(); if (qrc) return qrc; }
            ATN_HIP(film.resize((size_t)w * h));
            ATN_HIP(hipMemsetAsync(film.p, 0, (size_t)w * h * sizeof(float4), stream));
            film_w = w; film_h = h;
        }
        if (max_depth + 2 > counters_depth) {
            ATN_HIP(counters.resize((size_t)kMaxBatches * 4 * (max_depth + 2)));
            counters_depth = max_depth + 2;
        }
        if (!stats.p) {
            ATN_HIP(stats.resize(8));
            ATN_HIP(hipMemsetAsync(stats.p, 0, 64, stream));
        }
        return ATN_OK;
    }

    // batch k's queues live in [slot_begin, slot_end) of the queue arrays (a batch never has more live paths than
    // slots) and its counters in the k-th counter block
    PathBuffers buffers(bool count, int batch = 0, uint32_t slot_begin = 0)
    {
        PathBuffers pb{};
        pb.ray_o = ray_o.p; pb.ray_d = ray_d.p; pb.thr = thr.p; pb.contrib = contrib.p; pb.seeds = seeds.p;
        pb.isect = isect.p; pb.sh_o = sh_o.p; pb.sh_d = sh_d.p; pb.sh_c = sh_c.p;
        pb.accum = accum.p; pb.done = done.p; pb.queue[0] = queue0.p + slot_begin; pb.queue[1] = queue1.p + slot_begin;
        pb.shadow_q = shadow_q.p + slot_begin;
        uint32_t* cb = counters.p + (size_t)batch * 4 * counters_depth;
        pb.q_count = cb; pb.sh_count = cb + counters_depth;
        pb.fetch_closest = cb + 2 * counters_depth; pb.fetch_shadow = cb + 3 * counters_depth;
        pb.stats = count ? stats.p : nullptr;
        pb.cost = count ? cost.p : nullptr;
        return pb;
    }

    FrameParams frame_params(const atn_destination& d)
    {
        FrameParams fp{};
        fp.width = d.width; fp.height = d.height; fp.n_slots = (int32_t)n_slots;
        fp.tiles_x = (d.width + 7) / 8; fp.tiles_y = (d.height + 7) / 8; fp.tiles_x_rcp = udiv_rcp((uint32_t)fp.tiles_x);
        fp.rank = rank; fp.world = world;
        fp.max_depth = d.maxDepth;
        fp.rr_depth = d.russianRouletteDepth;
        if (fp.rr_depth > fp.max_depth) fp.rr_depth = fp.max_depth - 1;    // pathtracing.cpp:282-284
        fp.slot_begin = 0; fp.slot_end = (int32_t)n_slots;
        fp.chunk_items = kChunkItems;
        fp.sample = 0; fp.frame = d.frame; fp.n_seeds = n_seeds;
        fp.break_on_terminate = d.break_on_terminate; fp.progressive = d.progressive;
        return fp;
    }

    // launch geometry: enough waves to fill 256 CUs several times over; grid-stride loops do the rest
    static uint32_t grid_for(uint32_t n, uint32_t cap_blocks = 256u * 16u)
    {
        uint32_t b = (n + 255u) / 256u;
        if (b == 0) b = 1;
        return b < cap_blocks ? b : cap_blocks;
    }

    // Traversal flavour.  Measured on MI355X: the persistent lane-refilling walk wins
    // on sponza_lod (38 K nodes, ~56 node visits per ray: trace 3.25 -> 2.90 ms, shadow 4.40 -> 3.60 ms per
    // 1080p frame) and loses on the Cornell box (71 nodes, ~20 visits per ray: 0.83 -> 1.18 ms).
    // Since the trace launches are fused (r01-g) the refill walk only pays when a launch carries >= ~1.5 M paths as
    // well: on the 2-, 4- and 8-way shards of the 1080p frame the plain walk is 12-16 % faster (3.28 / 2.10 / 1.48 ms
    // against 3.71 / 2.39 / 1.77), so the flavour is picked per frame from tree size AND frame size.
    bool use_refill = false, tree_is_deep = false, flavour_forced = false;
    uint32_t simple_block = 64;     // threads per block of the plain walk's fused launches: one wave per block retires on its own (1-2 % over 256)
    static constexpr size_t kRefillMinNodes = 2048;
    // Re-measured after the burst walk (r02_e, sponza_lod, 3 frames in flight, ms per frame of an N-way shard: refill 4.29 /
    // 2.31 / 1.29 / 0.71 vs plain 5.51 / 2.93 / 1.39 / 0.75 at 2.07 M / 1.04 M / 0.52 M / 0.26 M paths): the r01 crossover of
    // 1.9 M paths is gone, deep trees take the refill walk at every size that fills the machine at all
    // r04, after the plain walk dropped from 84 to 63 VGPRs (8 waves per SIMD; no SLP pairing): it wins again on the 4- and 8-way shards of a
    // 1080p frame (sponza_lod 1.157 -> 1.097 / 0.640 -> 0.608 ms, atrium 1.753 -> 1.700 / 1.112 -> 1.032 with 3 frames in flight) and loses on
    // the 2-way shard of sponza_lod (2.13 vs 2.21) and on full frames (4.04 vs 4.47): profiles/r04_shard_matrix.txt
    // Re-measured after the refill walk's r04 changes (direct start, rays finished with the refill, TLAS step every third iteration): the two
    // walks tie on the 4-way shard (sponza_lod 1.075 / 1.076 ms, atrium 1.667 refill / 1.704 plain), the plain walk keeps the 8-way shard
    // (0.606 vs 0.627, atrium 1.036 vs 1.075): 800 K -> 400 K paths
    static constexpr uint32_t kRefillMinPaths = 400u * 1000u;

    // bytes of the LDS copy a small scene is walked from (node image + matrix rows), 0 = the scene is walked from global memory
    uint32_t lds_scene_bytes() const
    {
        const uint64_t b = (uint64_t)scene.node_bytes + (uint64_t)scene.mtx_quads * 16u;
        return (env_lds_nodes && b <= kLdsNodesMaxBytes) ? (uint32_t)b : 0u;
    }

    uint32_t trace_grid(uint32_t n_jobs) const
    {
        if (use_refill) {
            // persistent waves pulling kFetchChunk-job chunks: about as many waves as fit on the chip
            const uint32_t waves_per_block = (uint32_t)kTraceBlock / 64u;
            uint32_t blocks = ((n_jobs + atn::kFetchChunk - 1u) / atn::kFetchChunk + waves_per_block - 1u) / waves_per_block;
            if (blocks < 1u) blocks = 1u;
            // One frame at a time: 8 waves per SIMD's worth (6 are resident at 78 VGPRs; the rest start as those retire: best
            // isolated launch, 3.28 ms of trace per sponza_lod frame).  With frames in flight the other frames' kernels want
            // room beside this launch: 4.5 waves per SIMD's worth is slower alone (3.35 ms) and faster in the pipeline (r04 sweep,
            // profiles/r04_sweep_trace_grid.txt: sponza_lod 4.13 -> 4.04 ms per frame, atrium 5.84 -> 5.71 with 4 frames in flight;
            // 768 .. 1280 blocks within 1 %).
            const uint32_t cap = env_trace_blocks ? env_trace_blocks
                               : ((frames_in_flight > 1 ? 256u * 18u : 256u * 32u) / waves_per_block);
            return blocks < cap ? blocks : cap;
        }
        return grid_for(n_jobs);
    }

    // the persistent kernels run kTraceBlock threads per block
    template <bool SHADOW>
    void launch_trace(const PathBuffers& pb, uint32_t grid, bool count, int32_t b, hipStream_t stream)
    {
        const dim3 g(grid), t(use_refill ? (uint32_t)kTraceBlock : 256u);
        const uint32_t lds = 0u;
        if (SHADOW) {
            if (count) { if (use_refill) hipLaunchKernelGGL((k_trace_shadow<true, true>), g, t, lds, stream, pb, scene, b); else hipLaunchKernelGGL((k_trace_shadow<true, false>), g, t, lds, stream, pb, scene, b); }
            else { if (use_refill) hipLaunchKernelGGL((k_trace_shadow<false, true>), g, t, lds, stream, pb, scene, b); else hipLaunchKernelGGL((k_trace_shadow<false, false>), g, t, lds, stream, pb, scene, b); }
        }
        else {
            if (count) { if (use_refill) hipLaunchKernelGGL((k_trace_closest<true, true>), g, t, lds, stream, pb, scene, b); else hipLaunchKernelGGL((k_trace_closest<true, false>), g, t, lds, stream, pb, scene, b); }
            else { if (use_refill) hipLaunchKernelGGL((k_trace_closest<false, true>), g, t, lds, stream, pb, scene, b); else hipLaunchKernelGGL((k_trace_closest<false, false>), g, t, lds, stream, pb, scene, b); }
        }
    }

    template <bool SVGF>
    void launch_shade(uint32_t g_shade, hipStream_t st, const PathBuffers& pb, const FrameParams& fp, int32_t b, const SvgfShade& sv)
    {
        const dim3 g(g_shade), t(256);
        // 5 waves per SIMD (96 registers, a few spilled) pays where shade shares the SIMDs with another frame's trace waves AND the launch is
        // large enough not to be latency-bound itself: the Disney / analytic sets on every shard size measured, the core set on full frames
        // only (profiles/r04_variants_shade_waves.txt, r04_shard_matrix.txt)
        const bool big = (uint32_t)(fp.slot_end - fp.slot_begin) >= 1500u * 1000u;
        // (the SVGF flavour -- AOV writes, 25 registers spilled at 96 -- stays at 4: C5 4.66 vs 4.68 ms per frame)
        const int shade_waves = env_shade_waves ? env_shade_waves
                              : (!SVGF && frames_in_flight > 1 && (scene.material_set != kMsCore || big) ? 5 : 4);
        if (!SVGF && shade_math_relaxed) {      // atn_set_shade_math(1): the same kernel under --use_fast_math rules (shade_relaxed.hip); not the parity path
            relaxed_launch_shade(scene.material_set, shade_waves, g_shade, st, pb, scene, fp, camera, b);
            return;
        }
        switch (scene.material_set) {       // BSDFs no uploaded material uses are compiled out of the instantiation launched
        // (small sets: 5 waves per SIMD when frames overlap -- they share the SIMDs with another frame's trace waves --, 4 otherwise:
        // kernels.hpp, k_shade_wn)
        case kMsCore:
            if (shade_waves == 5) hipLaunchKernelGGL((k_shade_wn<SVGF, kMsCore, 5>), g, t, 0, st, pb, scene, fp, camera, b, sv);
            else hipLaunchKernelGGL((k_shade_wn<SVGF, kMsCore, 4>), g, t, 0, st, pb, scene, fp, camera, b, sv);
            break;
        case kMsDisney:
            if (shade_waves == 5) hipLaunchKernelGGL((k_shade_wn<SVGF, kMsDisney, 5>), g, t, 0, st, pb, scene, fp, camera, b, sv);
            else hipLaunchKernelGGL((k_shade_wn<SVGF, kMsDisney, 4>), g, t, 0, st, pb, scene, fp, camera, b, sv);
            break;
        case kMsAnalytic:
            if (shade_waves == 5) hipLaunchKernelGGL((k_shade_wn<SVGF, kMsAnalytic, 5>), g, t, 0, st, pb, scene, fp, camera, b, sv);
            else hipLaunchKernelGGL((k_shade_wn<SVGF, kMsAnalytic, 4>), g, t, 0, st, pb, scene, fp, camera, b, sv);
            break;
        case kMsCarPaint: hipLaunchKernelGGL((k_shade<SVGF, kMsCarPaint>), g, t, 0, st, pb, scene, fp, camera, b, sv); break;
        default: hipLaunchKernelGGL((k_shade<SVGF, kMsToon>), g, t, 0, st, pb, scene, fp, camera, b, sv); break;
        }
    }

    void prof_begin(bool on, int kind, hipStream_t st = nullptr)
    {
        if (!st) st = stream;
        prof_stream = st;
        if (!on) return;
        if (spans.size() >= kMaxSpans) {    // nobody collected for thousands of launches: resolve them now (bounded pool)
            (void)hipStreamSynchronize(stream);
            for (int k = 0; k < kMaxBatches; k++) (void)hipStreamSynchronize(bstream[k]);
            prof_collect();
        }
        if (ev_used + 2 > ev_pool.size()) {
            for (int i = 0; i < 64; i++) { hipEvent_t e; (void)hipEventCreate(&e); ev_pool.push_back(e); }
        }
        spans.push_back(Span{ kind, ev_used, ev_used + 1 });
        (void)hipEventRecord(ev_pool[ev_used], st);
    }
    void prof_end(bool on)
    {
        if (!on) return;
        (void)hipEventRecord(ev_pool[ev_used + 1], prof_stream);
        ev_used += 2;
    }
    void prof_collect()
    {
        for (const Span& s : spans) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, ev_pool[s.e0], ev_pool[s.e1]) == hipSuccess) { k_ms[s.kind] += ms; k_launches[s.kind]++; }
        }
        spans.clear();
        ev_used = 0;
    }

    // The sample loop of OnRender for every batch of the frame, each on its own stream, forked from and joined
    // back into `stream`.  SVGF = the SVGFRenderer flavour (AOV-writing shade, its own sample epilogue).
    template <bool SVGF>
    int run_paths(const atn_destination* d, FrameParams fp, bool count, bool prof, const SvgfShade& sv, const SvgfFrame& sf)
    {
        // Measured policy (sponza_lod / Cornell 1080p and its 2-, 4-, 8-way shards): overlap pays
        // while launches are latency-bound (<= ~1 M paths in flight); on a full 1080p frame the concurrent shade kernel
        // streams path state through the L2 that the walk wants for its nodes and the deep-tree walk loses more than
        // the overlap wins (6.13 vs 6.59 ms), the shallow-tree walk still gains with two batches.
        if (!flavour_forced) use_refill = tree_is_deep && n_slots >= kRefillMinPaths;
        int nb;
        if (batches_forced) {
            nb = n_batches;
            const uint32_t min_batch = 200u * 1000u;
            while (nb > 1 && n_slots / (uint32_t)nb < min_batch) nb--;
        }
        else {
            // the refill walk wants the machine to itself; the plain walk gains from a second batch down to ~0.8 M paths
            // (with frames in flight the next frame fills the gaps a second batch was for: measured r02_e on Cornell, 3 in
            // flight, 1.04 M paths: 1 batch 0.82 ms, 2 batches 0.96; 2.07 M paths: 1.66 vs 1.63)
            // (r03, node image in LDS, 3 in flight, Cornell 2.07 M paths: 1 batch 1.52 ms, 2 batches 1.58 -- one batch whenever frames overlap)
            const bool lds_nodes = lds_scene_bytes() != 0u;
            nb = use_refill ? 1 : (lds_nodes && frames_in_flight > 1) ? 1 : (n_slots >= (frames_in_flight > 1 ? 1500u : 800u) * 1000u ? 2 : 1);
            if (nb > n_batches) nb = n_batches;
        }
        // the streams a frame's batches run on must not share a hardware queue (see streams_run_side_by_side)
        if (nb > batch_streams_checked && nb > 1 && env_probe_streams) { int rc = separate_batch_streams(nb); if (rc) return rc; }
        uint32_t per = (n_slots + (uint32_t)nb - 1u) / (uint32_t)nb;
        per = (per + kChunk - 1u) / kChunk * kChunk;        // whole 1024-slot chunks (16 screen tiles)
        ATN_HIP(hipMemsetAsync(counters.p, 0, (size_t)nb * 4 * counters_depth * 4, stream));
        ATN_HIP(hipEventRecord(ev_fork, stream));
        for (int k = 0; k < nb; k++) {
            const uint32_t begin = (uint32_t)k * per;
            const uint32_t end = begin + per < n_slots ? begin + per : n_slots;
            if (begin >= end) continue;
            hipStream_t st = nb > 1 ? bstream[k] : stream;
            if (nb > 1) ATN_HIP(hipStreamWaitEvent(st, ev_fork, 0));
            PathBuffers pb = buffers(count, k, begin);
            fp.slot_begin = (int32_t)begin; fp.slot_end = (int32_t)end;
            const uint32_t n = end - begin;
            // k_shade works on chunks of `items` x 256 queue entries per block (one queue atomic per chunk): 4 on full
            // frames; below ~0.4 M paths a launch has too few such blocks to fill 256 CUs (measured on the 8-way
            // shard: 2 -> 1.38 ms, 4 -> 1.46 ms, 1 -> 1.41 ms per frame)
            int items = n >= 400u * 1000u ? kChunkItems : 2;
            if (env_shade_items) items = env_shade_items;
            fp.chunk_items = items;
            const uint32_t g_shade = grid_for((n + (uint32_t)items - 1u) / (uint32_t)items);
            const uint32_t g_slots = grid_for(n), g_trace = trace_grid(n), g_all = (n + 255u) / 256u;
            for (int32_t s = 0; s < d->sample; s++) {
                fp.sample = s;
                if (s > 0) ATN_HIP(hipMemsetAsync(pb.q_count, 0, (size_t)4 * counters_depth * 4, st));
                prof_begin(prof, ATN_K_GEN, st);
                hipLaunchKernelGGL(k_gen_path, dim3(g_slots), dim3(256), 0, st, pb, fp, camera, (const uint32_t*)seeds.p);
                prof_end(prof);
                if (count || !fuse_traces) {
                    for (int32_t b = 0; b < d->maxDepth; b++) {
                        prof_begin(prof, ATN_K_TRACE_CLOSEST, st);
                        launch_trace<false>(pb, g_trace, count, b, st);
                        prof_end(prof);
                        prof_begin(prof, ATN_K_SHADE, st);
                        launch_shade<SVGF>(g_shade, st, pb, fp, b, sv);
                        prof_end(prof);
                        prof_begin(prof, ATN_K_TRACE_SHADOW, st);
                        launch_trace<true>(pb, g_trace, count, b, st);
                        prof_end(prof);
                    }
                }
                else {
                    // depth + 1 trace launches: [closest 0], [shadow b + closest b+1] ..., [shadow depth-1]
                    const uint32_t g_fused = trace_grid(2u * n);
                    for (int32_t b = 0; b <= d->maxDepth; b++) {
                        const int32_t bs = b - 1, bc = b < d->maxDepth ? b : -1;
                        // the first launch holds only primary rays: coherent, they finish together, and the refill bookkeeping buys
                        // nothing (sponza_lod 4.33 -> 4.30 ms, atrium 4K 221 -> 219 ms)
                        const bool refill_now = use_refill && b != 0;     // (primary rays, coherent, take the plain walk: DESIGN.md section 7)
                        // (timed under "trace_closest" when it is a different kernel from the other launches: the roofline of
                        // k_trace_fused<true, .> is about those)
                        prof_begin(prof, (use_refill && !refill_now && b == 0) ? ATN_K_TRACE_CLOSEST : ATN_K_TRACE_FUSED, st);
                        // a node image of a few KB is walked from an LDS copy (trace_simple<., ., true>); above 8 KB per copy the blocks get
                        // four waves to share it
                        const bool lds_nodes = lds_scene_bytes() != 0u;
                        const uint32_t sb = (lds_nodes && lds_scene_bytes() > 8192u) ? 256u : simple_block;
                        const dim3 gr(refill_now ? g_fused : g_fused * (256u / sb)), tb(refill_now ? (uint32_t)kTraceBlock : sb);
                        const uint32_t lds = lds_nodes ? lds_scene_bytes() : 0u;
                        if (lds_nodes && refill_now) {
                            if (scene.any_alpha) hipLaunchKernelGGL((k_trace_fused<true, true, true>), gr, tb, lds, st, pb, scene, bs, bc, b);
                            else hipLaunchKernelGGL((k_trace_fused<true, false, true>), gr, tb, lds, st, pb, scene, bs, bc, b);
                        }
                        else if (lds_nodes) {
                            if (scene.any_alpha) hipLaunchKernelGGL((k_trace_fused<false, true, true>), gr, tb, lds, st, pb, scene, bs, bc, b);
                            else hipLaunchKernelGGL((k_trace_fused<false, false, true>), gr, tb, lds, st, pb, scene, bs, bc, b);
                        }
                        else if (refill_now) {
                            if (scene.any_alpha) hipLaunchKernelGGL((k_trace_fused<true, true>), gr, tb, lds, st, pb, scene, bs, bc, b);
                            else hipLaunchKernelGGL((k_trace_fused<true, false>), gr, tb, lds, st, pb, scene, bs, bc, b);
                        }
                        else if (b == 0) {
                            // only closest-hit rays (there is no bounce -1 to cast shadows): the closest-hit kernel is the same walk
                            // without the shadow job's code in it (every ray's stop_t is a constant there; the fused kernel's plain
                            // flavour grew by the any-hit twins' root selection: primary rays 0.168 -> 0.203 ms per frame, back at
                            // 0.168 through this launch)
                            hipLaunchKernelGGL((k_trace_closest<false, false>), gr, tb, lds, st, pb, scene, 0);
                        }
                        else {
                            if (scene.any_alpha) hipLaunchKernelGGL((k_trace_fused<false, true>), gr, tb, lds, st, pb, scene, bs, bc, b);
                            else hipLaunchKernelGGL((k_trace_fused<false, false>), gr, tb, lds, st, pb, scene, bs, bc, b);
                        }
                        prof_end(prof);
                        if (b < d->maxDepth) {
                            prof_begin(prof, ATN_K_SHADE, st);
                            launch_shade<SVGF>(g_shade, st, pb, fp, b, sv);
                            prof_end(prof);
                        }
                    }
                }
                if (SVGF || d->sample > 1) {     // with one sample per pixel k_gather<true> does the epilogue itself
                    prof_begin(prof, ATN_K_ACCUM, st);
                    if (SVGF) hipLaunchKernelGGL(k_svgf_sample_end, dim3(g_all), dim3(256), 0, st, pb, fp, sf);
                    else hipLaunchKernelGGL(k_accumulate_sample, dim3(g_all), dim3(256), 0, st, pb, fp);
                    prof_end(prof);
                }
            }
            if (nb > 1) {
                ATN_HIP(hipEventRecord(ev_join[k], st));
                ATN_HIP(hipStreamWaitEvent(stream, ev_join[k], 0));
            }
        }
        return ATN_OK;
    }

    // ≙ idaten::PathTracing::render + OnRender (src/libidaten/kernel/pathtracing.cpp:49-153), loop
    // structure of aten::PathTracing::OnRender/radiance (src/libaten/renderer/pathtracing/pathtracing.cpp:22-89,269-366)
    int render(const atn_destination* d, atn_vec4* out_host)
    {
        if (!d) return fail(ATN_ERR_INVALID_ARG, "null destination");
        if (!has_scene) return fail(ATN_ERR_NO_SCENE, "atn_upload_scene has not been called");
        if (!has_camera) return fail(ATN_ERR_INVALID_ARG, "atn_update_camera has not been called");
        if (n_seeds == 0) return fail(ATN_ERR_INVALID_ARG, "atn_init_sampler / atn_set_random has not been called");
        if (d->width <= 0 || d->height <= 0 || d->maxDepth <= 0 || d->sample <= 0) return fail(ATN_ERR_INVALID_ARG, "bad destination");
        ATN_HIP(hipSetDevice(device));
        if (d->sample > 1 && regen_applies(*d, 1)) return render_regen(d, 1, out_host);
        const bool count = d->count_stats != 0, prof = d->profile != 0;
        int rc;
        if (frames_in_flight > 1) {
            if (count) { rc = quiesce(); if (rc) return rc; }       // the counters are one set, read back synchronously
            else {
                // rotate: the bank that has been idle longest becomes the current one
                swap_bank(spare[frame_seq % (uint64_t)(frames_in_flight - 1)]);
            }
        }
        frame_seq++;
        rc = wait_scene_epoch();
        if (rc) return rc;
        rc = ensure_frame(d->width, d->height, d->maxDepth);
        if (rc) return rc;
        FrameParams fp = frame_params(*d);
        if (count) {
            ATN_HIP(hipMemsetAsync(stats.p, 0, 64, stream));
            ATN_HIP(cost.resize((size_t)2 * n_slots));
            ATN_HIP(cost_film.resize((size_t)2 * d->width * d->height));
            ATN_HIP(hipMemsetAsync(cost.p, 0, (size_t)2 * n_slots * sizeof(uint32_t), stream));
            ATN_HIP(hipMemsetAsync(cost_film.p, 0, (size_t)2 * d->width * d->height * sizeof(uint32_t), stream));
            cost_w = d->width; cost_h = d->height;
        }
        PathBuffers pb = buffers(count);

        const uint32_t g_all = (n_slots + 255u) / 256u;
        rc = run_paths<false>(d, fp, count, prof, SvgfShade{}, SvgfFrame{});
        if (rc) return rc;
        fp.slot_begin = 0; fp.slot_end = (int32_t)n_slots;
        if (film_pending) ATN_HIP(hipStreamWaitEvent(stream, ev_film, 0));     // the film is a running mean: frame order
        prof_begin(prof, ATN_K_GATHER);
        if (d->sample == 1) hipLaunchKernelGGL((k_gather<true>), dim3(g_all), dim3(256), 0, stream, pb, fp, film.p, tile_out.p);
        else hipLaunchKernelGGL((k_gather<false>), dim3(g_all), dim3(256), 0, stream, pb, fp, film.p, tile_out.p);
        prof_end(prof);
        ATN_HIP(hipGetLastError());
        if (frames_in_flight > 1) {
            ATN_HIP(hipEventRecord(ev_gather, stream)); rc = record_scene_read(); if (rc) return rc;
            if (!ev_film) ATN_HIP(hipEventCreateWithFlags(&ev_film, hipEventDisableTiming));
            ATN_HIP(hipEventRecord(ev_film, stream)); film_pending = true;
        }

        if (count) {
            hipLaunchKernelGGL(k_cost_to_pixels, dim3(g_all), dim3(256), 0, stream, fp, (const uint32_t*)cost.p, cost_film.p);
            ATN_HIP(hipMemcpyAsync(host_stats, stats.p, 64, hipMemcpyDeviceToHost, stream));
        }
        if (out_host) {
            ATN_HIP(hipMemcpyAsync(out_host, film.p, (size_t)d->width * d->height * sizeof(float4), hipMemcpyDeviceToHost, stream));
        }
        if (out_host || count) ATN_HIP(hipStreamSynchronize(stream));
        return ATN_OK;      // profiling spans are resolved lazily in kernel_times() (no sync in the frame loop)
    }


    // ------------------------------------------------------------------------------------------------
    // Path regeneration (BASELINE.json north_star "path compaction/regeneration"; atn_set_regeneration, atn_render_burst).
    //
    // The serial loop above -- the reference's, src/libidaten/kernel/pathtracing.cpp:105-138 -- is `for sample { generate; for bounce
    // { trace; shade } ; accumulate } gather`: every launch of a sample works on what is left of the population after the bounces
    // before it, a pixel whose path ended early idles until the longest path of the sample is over, and the next sample (or the next
    // progressive frame) starts from an empty machine.  Here one POOL of slots -- a slot is a pixel -- runs a whole burst of
    // `n_frames` progressive frames x `spp` samples: `begin; for stage { trace; shade } ; end`.  A path that ends in shade(stage)
    // has its sample epilogue run there (accumulate; after the frame's last sample Film::put), and the pixel's next primary ray is
    // written into the same slot and queued for trace(stage + 1); bounce, sample and frame are per-path state (kernels.hpp).  Per
    // pixel the samples and frames follow each other in the serial order with the serial operations, so films are BYTE-equal to
    // the serial loop's (tests/test_gpu_regen.py); what changes is only which launch a piece of work rides in.
    //
    // A path that runs out of depth with a shadow ray to trace hands its contribution over (`pend`) and the next sample starts at
    // once; the shadow ray adds its light to `pend` in the next stage's trace launch and the epilogue runs at the start of that
    // stage's shade (F_PENDING) -- before anything of the new sample can reach accum or the film.
    //
    // The film itself is written once per burst, by k_regen_end, from the burst's staging planes (one pixel value per frame): only that
    // launch is ordered behind the previous burst's (ev_film), so bursts on different banks (atn_set_frames_in_flight) overlap like
    // serial frames in flight do.  A burst is cut into pieces of at most kRegenStagingBytes of staging planes.
    //
    // Stages are launched up to the bound n_frames * spp * maxDepth (a pixel's worst case); the kernels read their counts from
    // device memory, and a stage whose queue is empty costs a kernel start.
    // ------------------------------------------------------------------------------------------------
    bool shade_math_relaxed = false;    // atn_set_shade_math
    int regen_mode = 0;         // atn_set_regeneration: 0 = serial sample loop, 1 = regenerated pool wherever it applies
    uint64_t rg_host_totals[4] = {};
    static constexpr int32_t kRegenMaxStages = 1 << 16;
    static constexpr int32_t kRegenMaxDepth = 128;      // 4096 CMJ dimensions / 32 per bounce (kRegenDimMask)
    static constexpr size_t kRegenStagingBytes = (size_t)1 << 30;

    bool regen_applies(const atn_destination& d, int32_t n_frames) const
    {
        if (regen_mode == 0 || d.count_stats) return false;        // (counted frames use the serial loop's counting kernels)
        if (d.maxDepth > kRegenMaxDepth || (uint32_t)d.sample >= kRegenMaxSpp) return false;
        if ((uint64_t)n_frames * (uint64_t)d.width * (uint64_t)d.height >= kRegenMaxItems) return false;     // (items are 28-bit)
        if (n_frames > 1 && !d.progressive) return false;           // (a burst of overwriting frames is its last frame)
        return ((int64_t)n_frames + 1) * d.sample * d.maxDepth <= kRegenMaxStages;
    }

    // ≙ n_frames x (idaten::PathTracing::render, pathtracing.cpp:49-153) with frame = d->frame, d->frame + 1, ...
    int render_burst(const atn_destination* d, int32_t n_frames, atn_vec4* out_host)
    {
        if (!d) return fail(ATN_ERR_INVALID_ARG, "null destination");
        if (n_frames <= 0) return fail(ATN_ERR_INVALID_ARG, "bad burst length");
        if (regen_applies(*d, n_frames)) {
            if (!has_scene) return fail(ATN_ERR_NO_SCENE, "atn_upload_scene has not been called");
            if (!has_camera) return fail(ATN_ERR_INVALID_ARG, "atn_update_camera has not been called");
            if (n_seeds == 0) return fail(ATN_ERR_INVALID_ARG, "atn_init_sampler / atn_set_random has not been called");
            if (d->width <= 0 || d->height <= 0 || d->maxDepth <= 0 || d->sample <= 0) return fail(ATN_ERR_INVALID_ARG, "bad destination");
            ATN_HIP(hipSetDevice(device));
            // pieces of at most kRegenStagingBytes of staging planes (1080p: 32 frames; a 4K frame is 133 MB)
            const uint64_t tiles = (uint64_t)((d->width + 7) / 8) * ((d->height + 7) / 8);
            const uint64_t plane = ((tiles + world - 1) / world) * 64u * sizeof(float4);
            int32_t piece = (int32_t)(kRegenStagingBytes / (plane ? plane : 1u));
            if (piece < 1) piece = 1;
            atn_destination part = *d;
            for (int32_t k = 0; k < n_frames; k += piece) {
                const int32_t n = n_frames - k < piece ? n_frames - k : piece;
                part.frame = d->frame + (uint32_t)k;
                const int rc = render_regen(&part, n, k + n == n_frames ? out_host : nullptr);
                if (rc) return rc;
            }
            return ATN_OK;
        }
        atn_destination one = *d;
        for (int32_t k = 0; k < n_frames; k++) {
            one.frame = d->frame + (uint32_t)k;
            const int rc = render(&one, k + 1 == n_frames ? out_host : nullptr);
            if (rc) return rc;
        }
        return ATN_OK;
    }

    int render_regen(const atn_destination* d, int32_t n_frames, atn_vec4* out_host)
    {
        const bool prof = d->profile != 0;
        int rc;
        if (frames_in_flight > 1) swap_bank(spare[frame_seq % (uint64_t)(frames_in_flight - 1)]);
        frame_seq++;
        rc = wait_scene_epoch();
        if (rc) return rc;
        rc = ensure_frame(d->width, d->height, d->maxDepth);
        if (rc) return rc;
        // Stages: while items are left every slot is busy, so the cursor passes the last item after at most (path-stages of all items) /
        // (slots) <= n_frames * spp * maxDepth stages; the items then in flight need at most spp * maxDepth more.
        const int32_t stages = (n_frames + 1) * d->sample * d->maxDepth;
        rc = regen_valid_list(d->width, d->height);
        if (rc) return rc;
        if (pend.n < n_slots) ATN_HIP(pend.resize(n_slots));
        const size_t cstride = (size_t)stages + 3;        // counters of stages -1 .. stages + 1
        if (rg_counters.n < 3 * cstride + 1) ATN_HIP(rg_counters.resize(3 * cstride + 1));
        if (rg_frames.n < (size_t)n_frames * n_slots) ATN_HIP(rg_frames.resize((size_t)n_frames * n_slots));
        rg_stages = stages;
        FrameParams fp = frame_params(*d);
        fp.burst_frames = n_frames; fp.spp = d->sample;
        fp.n_valid = rg_n_valid; fp.n_valid_rcp = udiv_rcp(rg_n_valid); fp.n_items = rg_n_valid * (uint32_t)n_frames;
        PathBuffers pb = buffers(false);
        pb.pend = pend.p;
        pb.valid_list = rg_valid.p; pb.next_item = rg_counters.p + 3 * cstride;
        pb.q_count = rg_counters.p + 1; pb.sh_count = rg_counters.p + cstride + 1; pb.fetch_closest = rg_counters.p + 2 * cstride + 1;
        pb.fetch_shadow = nullptr;
        const RegenOut ro{ rg_frames.p, film.p, tile_out.p };

        if (!flavour_forced) use_refill = tree_is_deep && n_slots >= kRefillMinPaths;
        const uint32_t n = rg_n_valid;      // the pool: one slot per pixel of the shard (n_frames >= 1: never more slots than items)
        int items = n >= 400u * 1000u ? kChunkItems : 2;
        if (env_shade_items) items = env_shade_items;
        fp.chunk_items = items;
        const uint32_t g_shade = grid_for((n + (uint32_t)items - 1u) / (uint32_t)items);
        const uint32_t g_all = (n_slots + 255u) / 256u, g_fused = trace_grid(2u * n);
        // the regions a shade launch writes for the compaction in front of the next stage (kernels.hpp, k_regen_compact)
        const uint32_t chunk_size = 256u * (uint32_t)items, n_chunks = (n + chunk_size - 1u) / chunk_size, n_groups = (n_chunks + kRegenGroup - 1u) / kRegenGroup;
        const size_t region_words = (size_t)n_chunks * chunk_size;
        const size_t rg_words = 2 * region_words + 2 * (size_t)n_chunks + 4 * (size_t)n_groups;
        if (rg_regions.n < rg_words) ATN_HIP(rg_regions.resize(rg_words));
        pb.q_regions = rg_regions.p; pb.sh_regions = rg_regions.p + region_words; pb.region_counts = rg_regions.p + 2 * region_words;
        uint32_t* const group_counts[2] = { pb.region_counts + 2 * (size_t)n_chunks, pb.region_counts + 2 * (size_t)n_chunks + 2 * (size_t)n_groups };
        const bool big = n >= 1500u * 1000u;
        const bool small_set = scene.material_set == kMsCore || scene.material_set == kMsDisney || scene.material_set == kMsAnalytic;
        // (the regenerated shade kernel needs all 128 registers of the 4-wave budget: held to 96 for 5 waves it spills 21-44 of them)
        const int shade_waves = !small_set ? 0 : env_shade_waves ? env_shade_waves : 4;
        (void)big;
        const bool lds_nodes = lds_scene_bytes() != 0u;
        const uint32_t sb = (lds_nodes && lds_scene_bytes() > 8192u) ? 256u : simple_block;

        ATN_HIP(hipMemsetAsync(rg_counters.p, 0, (3 * cstride + 1) * sizeof(uint32_t), stream));
        ATN_HIP(hipMemsetAsync(group_counts[0], 0, 4 * (size_t)n_groups * sizeof(uint32_t), stream));
        prof_begin(prof, ATN_K_GEN);
        pb.group_counts = group_counts[0];
        regen_launch_begin(g_shade, stream, pb, fp, camera);
        regen_launch_compact(n_chunks, stream, pb, 0, chunk_size, group_counts[1], n_groups);
        prof_end(prof);
        for (int32_t i = 0; i <= stages; i++) {
            // trace(i): the shadow rays shade(i - 1) cast + the closest-hit rays of the paths (continued and regenerated) it queued
            const bool refill_now = use_refill && i != 0;       // (stage 0 holds primary rays only: coherent, the plain walk)
            RegenTraceLaunch tl{};
            tl.refill = refill_now; tl.alpha = scene.any_alpha != 0; tl.lds_nodes = lds_nodes;
            tl.grid = refill_now ? g_fused : g_fused * (256u / sb); tl.block = refill_now ? (uint32_t)kTraceBlock : sb;
            tl.lds_bytes = lds_nodes ? lds_scene_bytes() : 0u;
            prof_begin(prof, (use_refill && i == 0) ? ATN_K_TRACE_CLOSEST : ATN_K_TRACE_FUSED);
            regen_launch_trace(tl, stream, pb, scene, i - 1, i < stages ? i : -1, i);
            prof_end(prof);
            if (i < stages) {
                prof_begin(prof, ATN_K_SHADE);
                pb.group_counts = group_counts[(i + 1) & 1];
                regen_launch_shade(scene.material_set, shade_waves, g_shade, stream, pb, scene, fp, camera, i, ro);
                prof_end(prof);
                prof_begin(prof, ATN_K_ACCUM);      // (timed under the serial loop's per-sample epilogue kind)
                regen_launch_compact(n_chunks, stream, pb, i + 1, chunk_size, group_counts[i & 1], n_groups);
                prof_end(prof);
            }
        }
        if (film_pending) ATN_HIP(hipStreamWaitEvent(stream, ev_film, 0));     // the film is a running mean: burst order
        prof_begin(prof, ATN_K_GATHER);
        regen_launch_flush((n + 255u) / 256u, stream, pb, fp, ro);
        regen_launch_end(g_all, stream, pb, fp, ro);
        prof_end(prof);
        ATN_HIP(hipGetLastError());
        if (frames_in_flight > 1) {
            ATN_HIP(hipEventRecord(ev_gather, stream)); rc = record_scene_read(); if (rc) return rc;
            if (!ev_film) ATN_HIP(hipEventCreateWithFlags(&ev_film, hipEventDisableTiming));
            ATN_HIP(hipEventRecord(ev_film, stream)); film_pending = true;
        }
        if (out_host) {
            ATN_HIP(hipMemcpyAsync(out_host, film.p, (size_t)d->width * d->height * sizeof(float4), hipMemcpyDeviceToHost, stream));
            ATN_HIP(hipStreamSynchronize(stream));
        }
        return ATN_OK;
    }

    // The shard's pixel slots that lie inside the frame, ascending (kernels.hpp: the items of a burst index it).  Built on the host from
    // the tile geometry once per (size, shard) and shared by every bank.
    DevBuf<uint32_t> rg_valid;
    uint32_t rg_n_valid = 0;
    int32_t rg_valid_key[4] = { -1, -1, -1, -1 };
    int regen_valid_list(int32_t w, int32_t h)
    {
        if (rg_valid_key[0] == w && rg_valid_key[1] == h && rg_valid_key[2] == rank && rg_valid_key[3] == world && rg_valid.p) return ATN_OK;
        const int32_t tx = (w + 7) / 8, ty = (h + 7) / 8;
        std::vector<uint32_t> v;
        v.reserve(n_slots);
        for (uint32_t slot = 0; slot < n_slots; slot++) {
            const uint32_t tile = (slot >> 6) * (uint32_t)world + (uint32_t)rank;       // slot_to_pixel, kernels.hpp
            if (tile >= (uint32_t)(tx * ty)) continue;
            const int32_t x = (int32_t)(tile % (uint32_t)tx) * 8 + (int32_t)(slot & 7u), y = (int32_t)(tile / (uint32_t)tx) * 8 + (int32_t)((slot & 63u) >> 3);
            if (x < w && y < h) v.push_back(slot);
        }
        { int q = quiesce(); if (q) return q; }        // (other banks' bursts may be reading the old list)
        ATN_HIP(rg_valid.resize(v.size() ? v.size() : 1));
        ATN_HIP(hipMemcpy(rg_valid.p, v.data(), v.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        rg_n_valid = (uint32_t)v.size();
        rg_valid_key[0] = w; rg_valid_key[1] = h; rg_valid_key[2] = rank; rg_valid_key[3] = world;
        return ATN_OK;
    }

    // per-stage populations of the current bank's last regenerated burst: closest-hit rays and shadow rays of every stage
    int regen_stage_counts(uint32_t* closest, uint32_t* shadow, uint32_t capacity, uint32_t* n_stages)
    {
        ATN_HIP(hipSetDevice(device));
        ATN_HIP(hipStreamSynchronize(stream));
        const uint32_t ns = (uint32_t)rg_stages + 1u;
        if (n_stages) *n_stages = rg_stages > 0 ? ns : 0u;
        if (rg_stages <= 0 || !rg_counters.p) return ATN_OK;
        const size_t cstride = (size_t)rg_stages + 3;
        std::vector<uint32_t> h(2 * cstride);
        ATN_HIP(hipMemcpy(h.data(), rg_counters.p, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < ns && i < capacity; i++) {
            if (closest) closest[i] = h[1 + i];
            if (shadow) shadow[i] = h[cstride + 1 + i];
        }
        return ATN_OK;
    }

    // ------------------------------------------------------------------------------------------------
    // SVGF (aten::SVGFRenderer, src/libaten/renderer/svgf/svgf.cpp): frame-persistent state = SVGFParams
    // (svgf_types.h:54-166) + MatricesForRendering (pt_params.h:150-185)
    // ------------------------------------------------------------------------------------------------
    DevBuf<float4> sv_aov[2][4], sv_scratch, sv_atrous[2], sv_tmp, sv_motion, sv_out, sv_stages;
    // what the path pass hands to the filters, two slots so that frame f + 1's path pass can run while frame f is filtered:
    // G-buffer staging (normal+depth, albedo+id), primary hit positions, contributions
    DevBuf<float4> sv_gnd[2], sv_gam[2], sv_primary[2], sv_contribs[2];
    int32_t sv_slot = 0, sv_last_slot = 0;      // slot of the next frame / of the last frame, upload or denoise
    hipEvent_t sv_ev_prepare[2] = { nullptr, nullptr };     // the prepare pass that consumed slot k has finished
    bool sv_prepare_recorded[2] = { false, false };
    float4* sv_cv[2] = { nullptr, nullptr };    // colour+variance of the two AOV sets (the variance pass swaps with sv_spare)
    float4* sv_spare = nullptr;
    int32_t sv_w = 0, sv_h = 0, sv_curr = 0, sv_atrous_iters = 5;
    bool sv_motion_set = false;
    hipStream_t sv_stream = nullptr;        // filter stream of pipelined SVGF frames (frames in flight > 1)
    bool w_or_h_changed(int32_t w, int32_t h) const { return w != sv_w || h != sv_h || w != film_w || h != film_h; }
    DevBuf<float> sv_weight;        // scalar plane of the optional temporal-weight dilation
    int32_t sv_dilate_weight = 0;
    size_t sv_motion_count = 0;     // elements of the last atn_svgf_set_motion_depth / upload (sv_motion.n is the capacity)
    float sv_W2V[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
    float sv_V2C[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
    float sv_prevW2V[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };

    static void mat_mul(const float* a, const float* b, float* out)        // mat4::operator*=, mat4.h:140-156
    {
        float tmp[16];
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                float acc = 0.0f;
                for (int k = 0; k < 4; k++) acc += a[4 * i + k] * b[4 * k + j];
                tmp[4 * i + j] = acc;
            }
        std::memcpy(out, tmp, sizeof(tmp));
    }

    // MatricesForRendering::Reset = Camera::ComputeCameraMatrices: mat4::lookat(origin, center, up) and
    // mat4::perspective(znear, zfar, vfov, aspect) written into the existing matrices (mat4.h:457-513)
    void svgf_reset_matrices()
    {
        std::memcpy(sv_prevW2V, sv_W2V, sizeof(sv_W2V));
        const float* e = camera.origin; const float* at = camera.center; const float* up = camera.up;
        float z[3] = { e[0] - at[0], e[1] - at[1], e[2] - at[2] };
        float il = 1.0f / std::sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);        // glm::normalize = v * inversesqrt(dot(v, v))
        z[0] *= il; z[1] *= il; z[2] *= il;
        float x[3] = { up[1] * z[2] - z[1] * up[2], up[2] * z[0] - z[2] * up[0], up[0] * z[1] - z[0] * up[1] };
        il = 1.0f / std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
        x[0] *= il; x[1] *= il; x[2] *= il;
        const float y[3] = { z[1] * x[2] - x[1] * z[2], z[2] * x[0] - x[2] * z[0], z[0] * x[1] - x[0] * z[1] };
        float* m = sv_W2V;
        m[0] = x[0]; m[4] = y[0]; m[8] = z[0];
        m[1] = x[1]; m[5] = y[1]; m[9] = z[1];
        m[2] = x[2]; m[6] = y[2]; m[10] = z[2];
        m[3] = -(x[0] * e[0] + x[1] * e[1] + x[2] * e[2]);
        m[7] = -(y[0] * e[0] + y[1] * e[1] + y[2] * e[2]);
        m[11] = -(z[0] * e[0] + z[1] * e[1] + z[2] * e[2]);
        m[15] = 1;
        const float fH = 1 / std::tan((3.14159265358979323846F * (camera.vfov) / 180.0F) * 0.5f);
        const float fW = fH / camera.aspect;
        float* p = sv_V2C;
        p[0] = fW; p[5] = fH;
        p[10] = camera.zfar / (camera.znear - camera.zfar);
        p[11] = camera.znear * camera.zfar / (camera.znear - camera.zfar);
        p[14] = -1.0f; p[15] = 0.0f;
    }

    int svgf_fill(float4* p, size_t n, float4 v)
    {
        hipLaunchKernelGGL(k_svgf_fill, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, p, (uint32_t)n, v);
        return ATN_OK;
    }

    // SVGFParams::InitBuffers: vec4() = (0, 0, 0, 1) for every AOV texel
    int svgf_ensure(int32_t w, int32_t h, bool stages)
    {
        const size_t n = (size_t)w * h;
        if (w != sv_w || h != sv_h) {
            const float4 init = make_float4(0.0F, 0.0F, 0.0F, 1.0F);
            for (auto& set : sv_aov) for (auto& b : set) { ATN_HIP(b.resize(n)); svgf_fill(b.p, n, init); }
            ATN_HIP(sv_scratch.resize(n)); svgf_fill(sv_scratch.p, n, init);
            for (auto& b : sv_atrous) { ATN_HIP(b.resize(n)); svgf_fill(b.p, n, init); }
            ATN_HIP(sv_tmp.resize(n)); svgf_fill(sv_tmp.p, n, init);
            for (int k = 0; k < 2; k++) {
                ATN_HIP(sv_primary[k].resize(n)); svgf_fill(sv_primary[k].p, n, make_float4(0, 0, 0, 0));
                ATN_HIP(sv_contribs[k].resize(n)); ATN_HIP(sv_gnd[k].resize(n)); ATN_HIP(sv_gam[k].resize(n));
            }
            ATN_HIP(sv_out.resize(n));
            if (!sv_motion_set || sv_motion_count < n) { ATN_HIP(sv_motion.resize(n)); sv_motion_set = false; sv_motion_count = 0; }
            sv_cv[0] = sv_aov[0][2].p; sv_cv[1] = sv_aov[1][2].p; sv_spare = sv_scratch.p;
            sv_w = w; sv_h = h; sv_curr = 0; sv_slot = 0; sv_last_slot = 0;
        }
        if (stages) ATN_HIP(sv_stages.resize(3 * n));
        return ATN_OK;
    }

    // ≙ SVGFRenderer::SetMotionDepthBuffer (svgf.cpp:441-450)
    int svgf_set_motion_depth(const atn_vec4* md, uint32_t n)
    {
        if (!md || n == 0) return fail(ATN_ERR_INVALID_ARG, "empty motion/depth buffer");
        ATN_HIP(hipSetDevice(device));
        ATN_HIP(sv_motion.resize(n));
        ATN_HIP(hipMemcpyAsync(sv_motion.p, md, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, stream));
        ATN_HIP(hipStreamSynchronize(stream));
        sv_motion_set = true;
        sv_motion_count = n;
        return ATN_OK;
    }

    int svgf_reset()
    {
        sv_w = 0; sv_h = 0; sv_curr = 0;        // buffers are re-initialised by the next svgf_render
        const float id[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
        std::memcpy(sv_W2V, id, sizeof(id)); std::memcpy(sv_V2C, id, sizeof(id)); std::memcpy(sv_prevW2V, id, sizeof(id));
        return ATN_OK;
    }

    // uploads address the slot the NEXT frame / denoise call reads, downloads the slot the last one used
    float4* svgf_buffer(int32_t which, bool for_upload = false)
    {
        const int32_t c = sv_curr, p = 1 - sv_curr;
        const int32_t k = for_upload ? sv_slot : sv_last_slot;
        if (for_upload && (which == 10 || which == 14)) sv_last_slot = sv_slot;
        if (which >= 0 && which < 4) return which == 2 ? sv_cv[c] : sv_aov[c][which].p;
        if (which < 8) return which == 6 ? sv_cv[p] : sv_aov[p][which - 4].p;
        switch (which) {
        case 8: return sv_tmp.p; case 9: return sv_motion.p; case 10: return sv_primary[k].p;
        case 11: return sv_atrous[0].p; case 12: return sv_atrous[1].p; case 13: return sv_out.p; case 14: return sv_contribs[k].p;
        }
        return nullptr;
    }

    // ≙ aten::SVGFRenderer::OnRender (svgf.cpp:452-637)
    int svgf_render(const atn_destination* d, int32_t compute_motion, atn_vec4* out_host, atn_vec4* stages_host, bool path_pass = true)
    {
        if (!d) return fail(ATN_ERR_INVALID_ARG, "null destination");
        if (!has_scene) return fail(ATN_ERR_NO_SCENE, "atn_upload_scene has not been called");
        if (!has_camera) return fail(ATN_ERR_INVALID_ARG, "atn_update_camera has not been called");
        if (n_seeds == 0) return fail(ATN_ERR_INVALID_ARG, "atn_init_sampler / atn_set_random has not been called");
        if (d->width <= 0 || d->height <= 0 || d->maxDepth <= 0 || d->sample <= 0) return fail(ATN_ERR_INVALID_ARG, "bad destination");
        if (world != 1) return fail(ATN_ERR_UNSUPPORTED, "SVGF needs the whole frame on one GPU (filter footprints cross tiles)");
        ATN_HIP(hipSetDevice(device));
        // Frames in flight (atn_set_frames_in_flight > 1): the path pass of frame f + 1 runs on the next bank's stream while
        // frame f is still being traced and filtered.  The path pass writes only its bank and slot (f + 1) % 2 of the
        // hand-over planes (G-buffer staging, primary positions, contributions), so all it waits for is the prepare pass
        // that consumed that slot two frames ago; the filter passes run on one filter stream, frame after frame, each
        // waiting for its own path pass.
        const bool pipelined = frames_in_flight > 1 && path_pass;
        int rc;
        if (!pipelined && frames_in_flight > 1) { rc = quiesce(); if (rc) return rc; }
        else if (w_or_h_changed(d->width, d->height)) { rc = quiesce(); if (rc) return rc; }
        if (pipelined) {
            if (!sv_stream) {
                ATN_HIP(hipStreamCreateWithFlags(&sv_stream, hipStreamNonBlocking));
                for (auto& e : sv_ev_prepare) ATN_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            }
            swap_bank(spare[frame_seq % (uint64_t)(frames_in_flight - 1)]);
        }
        frame_seq++;
        rc = wait_scene_epoch();
        if (rc) return rc;
        hipStream_t fs = pipelined ? sv_stream : stream;       // the stream the filter passes run on
        rc = ensure_frame(d->width, d->height, d->maxDepth);
        if (rc) return rc;
        rc = svgf_ensure(d->width, d->height, stages_host != nullptr);
        if (rc) return rc;
        if (!compute_motion && (!sv_motion_set || sv_motion_count < (size_t)d->width * d->height))
            return fail(ATN_ERR_INVALID_ARG, "no motion/depth buffer: call atn_svgf_set_motion_depth or pass compute_motion = 1");
        const bool prof = d->profile != 0;
        PathBuffers pb = buffers(false);
        FrameParams fp = frame_params(*d);
        fp.break_on_terminate = 1;

        svgf_reset_matrices();
        const int32_t cur = sv_curr, prv = 1 - sv_curr;
        SvgfFrame sf{};
        sf.nd = sv_aov[cur][0].p; sf.am = sv_aov[cur][1].p; sf.cv = sv_cv[cur]; sf.mt = sv_aov[cur][3].p;
        sf.pnd = sv_aov[prv][0].p; sf.pam = sv_aov[prv][1].p; sf.pcv = sv_cv[prv]; sf.pmt = sv_aov[prv][3].p;
        sf.cv_out = sv_spare;
        sf.atrous[0] = sv_atrous[0].p; sf.atrous[1] = sv_atrous[1].p;
        const int32_t slot = sv_slot;
        sv_last_slot = slot;
        sf.tmp = sv_tmp.p; sf.motion = sv_motion.p; sf.primary = sv_primary[slot].p; sf.contribs = sv_contribs[slot].p;
        sf.g_nd = path_pass ? sv_gnd[slot].p : nullptr; sf.g_am = path_pass ? sv_gam[slot].p : nullptr;
        sf.out = sv_out.p; sf.stages = stages_host ? sv_stages.p : nullptr;
        mat_mul(sv_V2C, sv_W2V, sf.w2c);
        mat_mul(sv_V2C, sv_prevW2V, sf.prev_w2c);
        sf.width = d->width; sf.height = d->height; sf.frame = d->frame; sf.atrous_iter_cnt = sv_atrous_iters;
        // Camera::ComputeScreenDistance (camera.h:216-221): tan of half the fov IN DEGREES, as the reference writes it
        sf.camera_distance = (float)d->height / (2.0f * std::tan(0.5f * camera.vfov));
        sf.compute_motion = compute_motion;
        SvgfShade sv{};
        sv.nd = sv_gnd[slot].p; sv.am = sv_gam[slot].p; sv.primary = sf.primary;
        sv.w2c3[0] = sf.w2c[12]; sv.w2c3[1] = sf.w2c[13]; sv.w2c3[2] = sf.w2c[14]; sv.w2c3[3] = sf.w2c[15];

        if (path_pass) {
            if (pipelined && sv_prepare_recorded[slot]) ATN_HIP(hipStreamWaitEvent(stream, sv_ev_prepare[slot], 0));
            rc = run_paths<true>(d, fp, false, prof, sv, sf);
            if (rc) return rc;
            if (pipelined) {
                ATN_HIP(hipEventRecord(ev_gather, stream));         // this bank's "path pass done"
                rc = record_scene_read();
                if (rc) return rc;
                ATN_HIP(hipStreamWaitEvent(fs, ev_gather, 0));
            }
        }
        const dim3 gp((((d->width + 7) / 8) + 7) / 8 * 8, (d->height + 31) / 32), tp(256);     // x: multiple of 8 (XCD strips)
        prof_begin(prof, ATN_K_SVGF_PREPARE, fs);
        hipLaunchKernelGGL(k_svgf_prepare, gp, tp, 0, fs, sf);
        prof_end(prof);
        if (path_pass) sv_slot = 1 - sv_slot;
        if (d->frame > 0) {
            prof_begin(prof, ATN_K_SVGF_TEMPORAL, fs);
            hipLaunchKernelGGL(k_svgf_temporal, gp, tp, 0, fs, sf, 0.98f, 0.05f);
            if (sv_dilate_weight) {
                ATN_HIP(sv_weight.resize((size_t)d->width * d->height));
                hipLaunchKernelGGL(k_svgf_dilate_weight, gp, tp, 0, fs, sf, sv_weight.p);
                hipLaunchKernelGGL(k_svgf_store_weight, gp, tp, 0, fs, sf, (const float*)sv_weight.p);
            }
            prof_end(prof);
        }
        else if (sf.stages) {
            // frame 0: the temporal pass only re-puts the raw contribution (svgf.cpp:549-551)
            ATN_HIP(hipMemcpyAsync(sf.stages + (size_t)d->width * d->height, sf.stages, (size_t)d->width * d->height * sizeof(float4), hipMemcpyDeviceToDevice, fs));
        }
        // the slot is free again only now: k_svgf_temporal is the last reader of the hand-over planes (it re-reads
        // sf.contribs[slot]); recording this right behind k_svgf_prepare let frame f + 2's k_svgf_sample_end overwrite them
        if (pipelined) { ATN_HIP(hipEventRecord(sv_ev_prepare[slot], fs)); sv_prepare_recorded[slot] = true; }
        prof_begin(prof, ATN_K_SVGF_VARIANCE, fs);
        hipLaunchKernelGGL(k_svgf_variance, gp, tp, 0, fs, sf);
        prof_end(prof);
        std::swap(sv_cv[cur], sv_spare);        // the pass wrote the new colour+variance into the spare buffer
        sf.cv = sv_cv[cur]; sf.cv_out = sv_spare;
        for (int32_t i = 0; i < sv_atrous_iters; i++) {
            prof_begin(prof, ATN_K_SVGF_ATROUS, fs);
            if (env_atrous4) {
                // four pixels per thread (svgf.hpp, k_svgf_atrous4): the thread grid covers the 2 x 2 pixel GROUPS of pitch 2^i
                const int32_t s = 1 << i;
                const int32_t nx = ((d->width + 2 * s - 1) / (2 * s)) * s, ny = ((d->height + 2 * s - 1) / (2 * s)) * s;
                const dim3 g4((((nx + 7) / 8) + 7) / 8 * 8, (ny + 31) / 32);
                hipLaunchKernelGGL(k_svgf_atrous4, g4, tp, 0, fs, sf, i);
            }
            else hipLaunchKernelGGL(k_svgf_atrous, gp, tp, 0, fs, sf, i);
            prof_end(prof);
        }
        prof_begin(prof, ATN_K_SVGF_PREPARE, fs);
        hipLaunchKernelGGL(k_svgf_copy, gp, tp, 0, fs, sf);
        prof_end(prof);
        ATN_HIP(hipGetLastError());
        sv_curr = 1 - sv_curr;

        const size_t n = (size_t)d->width * d->height;
        if (out_host) ATN_HIP(hipMemcpyAsync(out_host, sv_out.p, n * sizeof(float4), hipMemcpyDeviceToHost, fs));
        if (stages_host) ATN_HIP(hipMemcpyAsync(stages_host, sv_stages.p, 3 * n * sizeof(float4), hipMemcpyDeviceToHost, fs));
        if (out_host || stages_host) ATN_HIP(hipStreamSynchronize(fs));
        return ATN_OK;
    }

    // ≙ idaten::Renderer::reset, renderer.h:40-43
    int reset()
    {
        if (film.p) {
            ATN_HIP(hipSetDevice(device));
            { int q = quiesce(); if (q) return q; }
            ATN_HIP(hipMemsetAsync(film.p, 0, film.n * sizeof(float4), stream));
            if (frames_in_flight > 1) {     // the next frame runs on another bank's stream: its k_gather goes behind the clear
                if (!ev_film) ATN_HIP(hipEventCreateWithFlags(&ev_film, hipEventDisableTiming));
                ATN_HIP(hipEventRecord(ev_film, stream)); film_pending = true;
            }
        }
        return ATN_OK;
    }
};

} // namespace atn

#include "host/mgpu.hpp"

struct atn_ctx { atn::PathTracing r; };
struct atn_mgpu { atn::MultiGpu m; };

using atn::PathTracing;

#define CTX_OR_FAIL(ctx) do { if (!(ctx)) return ATN_ERR_INVALID_ARG; } while (0)
// every entry point except atn_render first lets the frames in flight finish (atn_set_frames_in_flight)
#define CTX_QUIET_OR_FAIL(ctx) do { if (!(ctx)) return ATN_ERR_INVALID_ARG; \
    if ((ctx)->r.frames_in_flight > 1) { int q_ = (ctx)->r.quiesce(); if (q_) return q_; } } while (0)

// No exception may cross the C boundary (std::vector growth in the upload paths can throw std::bad_alloc).
template <class F>
static int guarded(atn_ctx* ctx, F&& f) noexcept
{
    try { return f(); }
    catch (const std::bad_alloc&) {
        try { return ctx->r.fail(ATN_ERR_OUT_OF_MEMORY, "out of host memory"); } catch (...) { return ATN_ERR_OUT_OF_MEMORY; }
    }
    catch (const std::exception& e) {
        try { return ctx->r.fail(ATN_ERR_INVALID_ARG, std::string("unexpected exception: ") + e.what()); } catch (...) { return ATN_ERR_INVALID_ARG; }
    }
    catch (...) { return ATN_ERR_INVALID_ARG; }
}
#define C_HIP(r, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (r).fail(ATN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

#define MG_OR_FAIL(mg) do { if (!(mg)) return ATN_ERR_INVALID_ARG; } while (0)
template <class F>
static int mg_guarded(atn_mgpu* mg, F&& f) noexcept
{
    try { return f(); }
    catch (const std::bad_alloc&) { try { return mg->m.fail(ATN_ERR_OUT_OF_MEMORY, "out of host memory"); } catch (...) { return ATN_ERR_OUT_OF_MEMORY; } }
    catch (...) { return ATN_ERR_INVALID_ARG; }
}

extern "C" {

int atn_create(atn_ctx** out, int device_ordinal)
{
    if (!out) return ATN_ERR_INVALID_ARG;
    *out = nullptr;
    atn_ctx* c = new (std::nothrow) atn_ctx();
    if (!c) return ATN_ERR_OUT_OF_MEMORY;
    int rc = c->r.init(device_ordinal);
    if (rc != ATN_OK) {
        std::fprintf(stderr, "atn_create: %s\n", c->r.last_error.c_str());
        delete c;
        return rc;
    }
    *out = c;
    return ATN_OK;
}

void atn_destroy(atn_ctx* ctx) { delete ctx; }

const char* atn_last_error(atn_ctx* ctx) { return ctx ? ctx->r.last_error.c_str() : "null context"; }

int atn_upload_scene(atn_ctx* ctx, const atn_scene_desc* scene)
{
    CTX_QUIET_OR_FAIL(ctx);
    if (!scene) return ctx->r.fail(ATN_ERR_INVALID_ARG, "null scene");
    return guarded(ctx, [&] { return ctx->r.UpdateSceneData(scene); });
}

int atn_set_upload_options(atn_ctx* ctx, int32_t anyhit_twin, int32_t anyhit_twin_dirs, int32_t node_layout, int32_t planar_lights)
{
    CTX_QUIET_OR_FAIL(ctx);
    if (anyhit_twin > 2 || (anyhit_twin_dirs >= 0 && anyhit_twin_dirs != 1 && anyhit_twin_dirs != 8)) return ctx->r.fail(ATN_ERR_INVALID_ARG, "upload option out of range");
    if (anyhit_twin >= 0) ctx->r.env_anyhit_twin = anyhit_twin;
    if (anyhit_twin_dirs >= 0) ctx->r.env_anyhit_twin_dirs = anyhit_twin_dirs;
    if (node_layout >= 0) ctx->r.opt_node_layout = node_layout != 0;
    if (planar_lights >= 0) ctx->r.opt_planar_lights = planar_lights != 0;
    return ATN_OK;
}
int atn_update_camera(atn_ctx* ctx, const atn_camera_param* camera) { CTX_OR_FAIL(ctx); return guarded(ctx, [&] { return ctx->r.updateCamera(camera); }); }

int atn_update_tlas(atn_ctx* ctx, const atn_object_param* objects, uint32_t n_objects, const atn_mat4* matrices, uint32_t n_matrices,
                    const atn_bvh_node* top_nodes, uint32_t n_top_nodes)
{
    CTX_OR_FAIL(ctx);      // enqueued behind the frames in flight (begin_scene_update), no host stop
    return guarded(ctx, [&] { return ctx->r.updateBVH(objects, n_objects, matrices, n_matrices, top_nodes, n_top_nodes); });
}
int atn_update_geometry(atn_ctx* ctx, const atn_vec4* vtx_pos, const atn_vec4* vtx_nml, uint32_t n_vertices, uint32_t vtx_offset,
                        const atn_triangle_param* triangles, uint32_t n_triangles, uint32_t tri_offset)
{
    CTX_OR_FAIL(ctx);
    return guarded(ctx, [&] { return ctx->r.updateGeometry(vtx_pos, vtx_nml, n_vertices, vtx_offset, triangles, n_triangles, tri_offset); });
}
int atn_lbvh_rebuild_list(atn_ctx* ctx, uint32_t list_index, uint32_t tri_offset, uint32_t n_triangles, const float* bbox_min, const float* bbox_max)
{
    CTX_OR_FAIL(ctx);
    return guarded(ctx, [&] { return ctx->r.lbvh_rebuild_list(list_index, tri_offset, n_triangles, bbox_min, bbox_max, false); });
}
int atn_lbvh_build(atn_ctx* ctx, const atn_triangle_param* triangles, uint32_t n_triangles, int32_t tri_id_offset,
                   const float* bbox_min, const float* bbox_max, const atn_vec4* vtx_pos, uint32_t n_vertices, int32_t vtx_offset,
                   atn_bvh_node* out_nodes, uint32_t* out_sorted_codes, uint32_t* out_sorted_indices)
{
    CTX_QUIET_OR_FAIL(ctx);
    return guarded(ctx, [&] { return ctx->r.lbvh_build(triangles, n_triangles, tri_id_offset, bbox_min, bbox_max, vtx_pos, n_vertices, vtx_offset,
                                                       out_nodes, out_sorted_codes, out_sorted_indices); });
}
int atn_scene_device_arrays(atn_ctx* ctx, void** vtx_pos, void** vtx_nml, void** triangles)
{
    CTX_QUIET_OR_FAIL(ctx);
    if (!ctx->r.has_scene) return ctx->r.fail(ATN_ERR_NO_SCENE, "atn_upload_scene has not been called");
    // the caller writes these arrays itself from now on: one copy of the scene, updates in place behind the frames in flight
    { int q = ctx->r.quiesce(); if (q) return q; }
    ctx->r.drop_alt_set(); ctx->r.scene_in_place = true;
    ctx->r.scene.planar_lights = 0;     // (the caller writes vertices from now on)
    if (vtx_pos) *vtx_pos = ctx->r.vtx_pos.p;
    if (vtx_nml) *vtx_nml = ctx->r.vtx_nml.p;
    if (triangles) *triangles = ctx->r.tris.p;
    return ATN_OK;
}
int atn_init_sampler(atn_ctx* ctx, int32_t w, int32_t h, int32_t seed) { CTX_QUIET_OR_FAIL(ctx); return guarded(ctx, [&] { return ctx->r.initSampler(w, h, seed); }); }
int atn_get_random(atn_ctx* ctx, uint32_t* out_host, uint32_t n) { CTX_QUIET_OR_FAIL(ctx); return guarded(ctx, [&] { return ctx->r.getRandom(out_host, n); }); }
uint32_t atn_random_count(atn_ctx* ctx) { return ctx ? ctx->r.n_seeds : 0; }
int atn_set_random(atn_ctx* ctx, const uint32_t* seeds, uint32_t n) { CTX_QUIET_OR_FAIL(ctx); return guarded(ctx, [&] { return ctx->r.setRandom(seeds, n); }); }

int atn_set_screen_shard(atn_ctx* ctx, int32_t rank, int32_t world)
{
    CTX_OR_FAIL(ctx);
    if (world <= 0 || rank < 0 || rank >= world) return ctx->r.fail(ATN_ERR_INVALID_ARG, "bad screen shard");
    ctx->r.rank = rank; ctx->r.world = world;
    return ATN_OK;
}

int atn_render(atn_ctx* ctx, const atn_destination* dst, atn_vec4* out_host) { CTX_OR_FAIL(ctx); return guarded(ctx, [&] { return ctx->r.render(dst, out_host); }); }
int atn_render_burst(atn_ctx* ctx, const atn_destination* dst, int32_t n_frames, atn_vec4* out_host) { CTX_OR_FAIL(ctx); return guarded(ctx, [&] { return ctx->r.render_burst(dst, n_frames, out_host); }); }
int atn_set_regeneration(atn_ctx* ctx, int32_t mode)
{
    CTX_QUIET_OR_FAIL(ctx);
    if (mode < 0 || mode > 1) return ctx->r.fail(ATN_ERR_INVALID_ARG, "regeneration mode out of range");
    ctx->r.regen_mode = mode;
    return ATN_OK;
}
int atn_set_shade_math(atn_ctx* ctx, int32_t mode)
{
    CTX_QUIET_OR_FAIL(ctx);
    if (mode < 0 || mode > 1) return ctx->r.fail(ATN_ERR_INVALID_ARG, "shade math mode out of range");
    ctx->r.shade_math_relaxed = mode == 1;
    return ATN_OK;
}
int32_t atn_get_regeneration(atn_ctx* ctx) { return ctx ? ctx->r.regen_mode : 0; }
int atn_regen_stage_counts(atn_ctx* ctx, uint32_t* closest, uint32_t* shadow, uint32_t capacity, uint32_t* n_stages)
{
    CTX_QUIET_OR_FAIL(ctx);
    return guarded(ctx, [&] { return ctx->r.regen_stage_counts(closest, shadow, capacity, n_stages); });
}
int atn_reset(atn_ctx* ctx) { CTX_QUIET_OR_FAIL(ctx); return ctx->r.reset(); }
int atn_set_path_batches(atn_ctx* ctx, int32_t n)
{
    CTX_OR_FAIL(ctx);
    if (n < 1 || n > PathTracing::kMaxBatches) return ctx->r.fail(ATN_ERR_INVALID_ARG, "batch count out of range");
    ctx->r.n_batches = n;
    ctx->r.batches_forced = n == 1;     // 1 = strictly serial; larger values stay subject to the size policy
    return ATN_OK;
}

int atn_set_sampling_options(atn_ctx* ctx, int32_t ibl_importance, int32_t tex_bilinear)
{
    CTX_OR_FAIL(ctx);
    return guarded(ctx, [&] { return ctx->r.set_sampling_options(ibl_importance, tex_bilinear); });
}

// texture::at / texture::AtWithBilinear probe (parity tests): n lookups of texture `texid` at uv[2n] -> out[4n]
int atn_sample_texture(atn_ctx* ctx, int32_t texid, uint32_t n, const float* uv_host, float* out_host)
{
    CTX_QUIET_OR_FAIL(ctx);
    PathTracing& r = ctx->r;
    if (!r.has_scene) return r.fail(ATN_ERR_NO_SCENE, "atn_upload_scene has not been called");
    if (texid < 0 || texid >= r.scene.n_textures || n == 0 || !uv_host || !out_host) return r.fail(ATN_ERR_INVALID_ARG, "bad texture id / count");
    return guarded(ctx, [&]() -> int {
        C_HIP(r, hipSetDevice(r.device));
        atn::DevBuf<float> duv, dout;
        C_HIP(r, duv.resize(2 * (size_t)n)); C_HIP(r, dout.resize(4 * (size_t)n));
        C_HIP(r, hipMemcpyAsync(duv.p, uv_host, 8 * (size_t)n, hipMemcpyHostToDevice, r.stream));
        hipLaunchKernelGGL(atn::k_sample_texture, dim3((n + 255) / 256), dim3(256), 0, r.stream, r.scene, texid, n, (const float*)duv.p, dout.p);
        C_HIP(r, hipGetLastError());
        C_HIP(r, hipMemcpyAsync(out_host, dout.p, 16 * (size_t)n, hipMemcpyDeviceToHost, r.stream));
        C_HIP(r, hipStreamSynchronize(r.stream));
        return ATN_OK;
    });
}

int atn_set_frames_in_flight(atn_ctx* ctx, int32_t n)
{
    CTX_OR_FAIL(ctx);
    return guarded(ctx, [&] { return ctx->r.set_frames_in_flight(n); });
}

int atn_bank_streams(atn_ctx* ctx, int32_t* swaps, int32_t* concurrent)
{
    CTX_OR_FAIL(ctx);
    return guarded(ctx, [&] {
        atn::PathTracing& r = ctx->r;
        C_HIP(r, hipSetDevice(r.device));
        { int q = r.quiesce(); if (q) return q; }
        if (swaps) *swaps = r.n_stream_swaps;
        if (concurrent) {
            *concurrent = 1;
            hipStream_t st[atn::PathTracing::kMaxInFlight];
            st[0] = r.stream;
            for (int i = 1; i < r.frames_in_flight; i++) st[i] = r.spare[i - 1].stream;
            for (int i = 0; i < r.frames_in_flight; i++)
                for (int j = 0; j < i; j++) {
                    bool ok = true;
                    if (!atn::streams_run_side_by_side(st[j], st[i], ok)) *concurrent &= ~1;
                    if (!ok) return r.fail(ATN_ERR_HIP, "stream probe failed");
                }
            // bit 1: the side stream, if it was handed out, runs beside every bank stream
            if (r.side_stream) {
                *concurrent |= 2;
                for (int i = 0; i < r.frames_in_flight; i++) {
                    bool ok = true;
                    if (!atn::streams_run_side_by_side(st[i], r.side_stream, ok)) *concurrent &= ~2;
                    if (!ok) return r.fail(ATN_ERR_HIP, "stream probe failed");
                }
            }
        }
        return (int)ATN_OK;
    });
}

int atn_svgf_render(atn_ctx* ctx, const atn_destination* dst, int32_t compute_motion, atn_vec4* out_host, atn_vec4* stages_host)
{
    CTX_OR_FAIL(ctx);
    return guarded(ctx, [&] { return ctx->r.svgf_render(dst, compute_motion, out_host, stages_host); });
}
int atn_svgf_set_motion_depth(atn_ctx* ctx, const atn_vec4* motion_depth, uint32_t n) { CTX_QUIET_OR_FAIL(ctx); return guarded(ctx, [&] { return ctx->r.svgf_set_motion_depth(motion_depth, n); }); }
int atn_svgf_reset(atn_ctx* ctx) { CTX_QUIET_OR_FAIL(ctx); return ctx->r.svgf_reset(); }
int atn_svgf_set_atrous_iterations(atn_ctx* ctx, int32_t n)
{
    CTX_OR_FAIL(ctx);
    if (n < 1 || n > 8) return ctx->r.fail(ATN_ERR_INVALID_ARG, "a-trous iteration count out of range");
    ctx->r.sv_atrous_iters = n;
    return ATN_OK;
}
int atn_svgf_set_dilate_temporal_weight(atn_ctx* ctx, int32_t on)
{
    CTX_OR_FAIL(ctx);
    ctx->r.sv_dilate_weight = on ? 1 : 0;
    return ATN_OK;
}
int atn_svgf_download(atn_ctx* ctx, int32_t which, atn_vec4* out_host)
{
    CTX_QUIET_OR_FAIL(ctx);
    PathTracing& r = ctx->r;
    float4* p = r.sv_w > 0 ? r.svgf_buffer(which) : nullptr;
    if (!p || !out_host) return r.fail(ATN_ERR_INVALID_ARG, "no such SVGF buffer (or atn_svgf_render has not run)");
    C_HIP(r, hipSetDevice(r.device));
    C_HIP(r, hipMemcpyAsync(out_host, p, (size_t)r.sv_w * r.sv_h * sizeof(float4), hipMemcpyDeviceToHost, r.stream));
    C_HIP(r, hipStreamSynchronize(r.stream));
    return ATN_OK;
}
int atn_svgf_denoise(atn_ctx* ctx, const atn_destination* dst, int32_t compute_motion, atn_vec4* out_host, atn_vec4* stages_host)
{
    CTX_QUIET_OR_FAIL(ctx);
    return guarded(ctx, [&] { return ctx->r.svgf_render(dst, compute_motion, out_host, stages_host, false); });
}
int atn_svgf_upload(atn_ctx* ctx, int32_t which, int32_t width, int32_t height, const atn_vec4* host)
{
    CTX_QUIET_OR_FAIL(ctx);
    PathTracing& r = ctx->r;
    if (!host || width <= 0 || height <= 0) return r.fail(ATN_ERR_INVALID_ARG, "bad SVGF upload");
    C_HIP(r, hipSetDevice(r.device));
    int rc = r.svgf_ensure(width, height, false);
    if (rc) return rc;
    float4* p = r.svgf_buffer(which, true);
    if (!p) return r.fail(ATN_ERR_INVALID_ARG, "no such SVGF buffer");
    C_HIP(r, hipMemcpyAsync(p, host, (size_t)width * height * sizeof(float4), hipMemcpyHostToDevice, r.stream));
    C_HIP(r, hipStreamSynchronize(r.stream));
    if (which == 9) { r.sv_motion_set = true; r.sv_motion_count = (size_t)width * height; }
    return ATN_OK;
}
void* atn_svgf_output_device(atn_ctx* ctx) { return ctx ? (void*)ctx->r.sv_out.p : nullptr; }

void* atn_film_device(atn_ctx* ctx) { return ctx ? (void*)ctx->r.film.p : nullptr; }
void* atn_tile_device(atn_ctx* ctx) { return ctx ? (void*)ctx->r.tile_out.p : nullptr; }
uint32_t atn_tile_slots(atn_ctx* ctx) { return ctx ? ctx->r.n_slots : 0; }
uint32_t atn_planar_area_lights(atn_ctx* ctx) { return (ctx && ctx->r.scene.planar_lights) ? ctx->r.n_planar_lights : 0u; }
uint32_t atn_anyhit_twins(atn_ctx* ctx)
{
    uint32_t n = 0;
    if (ctx) for (const int32_t d : ctx->r.list_twin_delta) n += d != 0 ? 1u : 0u;
    return n;
}
void* atn_stream(atn_ctx* ctx) { return ctx ? (void*)ctx->r.stream : nullptr; }
void* atn_side_stream(atn_ctx* ctx)
{
    if (!ctx) return nullptr;
    try { if (ctx->r.quiesce() != ATN_OK) return nullptr; return (void*)ctx->r.get_side_stream(); } catch (...) { return nullptr; }
}

int atn_synchronize(atn_ctx* ctx)
{
    CTX_QUIET_OR_FAIL(ctx);
    C_HIP(ctx->r, hipStreamSynchronize(ctx->r.stream));
    return ATN_OK;
}

int atn_assemble_tiles(atn_ctx* ctx, const void* gathered_dev, int32_t world, void* film_dev_out)
{
    return atn_assemble_tiles_on(ctx, gathered_dev, world, film_dev_out, nullptr);
}

int atn_assemble_tiles_on(atn_ctx* ctx, const void* gathered_dev, int32_t world, void* film_dev_out, void* hip_stream)
{
    CTX_OR_FAIL(ctx);
    PathTracing& r = ctx->r;
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : r.stream;
    if (!gathered_dev || world <= 0 || r.film_w <= 0) return r.fail(ATN_ERR_INVALID_ARG, "atn_assemble_tiles: nothing rendered yet");
    float4* dst = film_dev_out ? (float4*)film_dev_out : r.film.p;
    const uint32_t total = (uint32_t)world * r.n_slots;
    hipLaunchKernelGGL(atn::k_assemble_tiles, dim3((total + 255) / 256), dim3(256), 0, st,
                       (const float4*)gathered_dev, dst, r.film_w, r.film_h, (r.film_w + 7) / 8, (r.film_h + 7) / 8,
                       world, (int32_t)r.n_slots);
    C_HIP(r, hipGetLastError());
    return ATN_OK;
}

int atn_download_film(atn_ctx* ctx, atn_vec4* out_host)
{
    CTX_QUIET_OR_FAIL(ctx);
    PathTracing& r = ctx->r;
    if (!out_host || !r.film.p) return r.fail(ATN_ERR_INVALID_ARG, "atn_download_film: nothing rendered yet");
    C_HIP(r, hipMemcpyAsync(out_host, r.film.p, (size_t)r.film_w * r.film_h * sizeof(float4), hipMemcpyDeviceToHost, r.stream));
    C_HIP(r, hipStreamSynchronize(r.stream));
    return ATN_OK;
}

// Resume: the film of an earlier run (atn_download_film) becomes the state the next progressive frame continues from --
// FilmProgressive keeps {running mean, sample count in .w} per pixel (src/libaten/renderer/film.cpp:61-71).
int atn_upload_film(atn_ctx* ctx, int32_t width, int32_t height, const atn_vec4* film_host)
{
    CTX_QUIET_OR_FAIL(ctx);
    PathTracing& r = ctx->r;
    if (!film_host || width <= 0 || height <= 0) return r.fail(ATN_ERR_INVALID_ARG, "atn_upload_film: bad size / null film");
    return guarded(ctx, [&]() -> int {
        C_HIP(r, hipSetDevice(r.device));
        C_HIP(r, r.film.resize((size_t)width * height));
        C_HIP(r, hipMemcpyAsync(r.film.p, film_host, (size_t)width * height * sizeof(float4), hipMemcpyHostToDevice, r.stream));
        C_HIP(r, hipStreamSynchronize(r.stream));
        r.film_w = width; r.film_h = height;        // the next render of this size keeps the film instead of clearing it
        return ATN_OK;
    });
}

int atn_download_path_cost(atn_ctx* ctx, uint32_t* out_host)
{
    CTX_QUIET_OR_FAIL(ctx);
    PathTracing& r = ctx->r;
    if (!out_host) return r.fail(ATN_ERR_INVALID_ARG, "null output");
    if (!r.cost_film.p || r.cost_w <= 0) return r.fail(ATN_ERR_INVALID_ARG, "no frame has been rendered with count_stats = 1");
    return guarded(ctx, [&]() -> int {
        if (hipSetDevice(r.device) != hipSuccess) return r.fail(ATN_ERR_HIP, "hipSetDevice");
        if (hipMemcpyAsync(out_host, r.cost_film.p, (size_t)2 * r.cost_w * r.cost_h * sizeof(uint32_t), hipMemcpyDeviceToHost, r.stream) != hipSuccess
            || hipStreamSynchronize(r.stream) != hipSuccess) return r.fail(ATN_ERR_HIP, "cost map download");
        return ATN_OK;
    });
}
int atn_get_stats(atn_ctx* ctx, uint64_t out[8])
{
    CTX_QUIET_OR_FAIL(ctx);
    for (int i = 0; i < 8; i++) out[i] = ctx->r.host_stats[i];
    return ATN_OK;
}

int atn_get_kernel_times(atn_ctx* ctx, float ms[ATN_K_COUNT], uint32_t launches[ATN_K_COUNT])
{
    CTX_QUIET_OR_FAIL(ctx);
    C_HIP(ctx->r, hipStreamSynchronize(ctx->r.stream));
    ctx->r.prof_collect();
    for (int i = 0; i < ATN_K_COUNT; i++) { ms[i] = ctx->r.k_ms[i]; launches[i] = ctx->r.k_launches[i]; }
    return ATN_OK;
}

int atn_reset_kernel_times(atn_ctx* ctx)
{
    CTX_QUIET_OR_FAIL(ctx);
    C_HIP(ctx->r, hipStreamSynchronize(ctx->r.stream));
    ctx->r.prof_collect();
    for (int i = 0; i < ATN_K_COUNT; i++) { ctx->r.k_ms[i] = 0; ctx->r.k_launches[i] = 0; }
    return ATN_OK;
}

// ---------------------------------------------------------------- one node, every GPU (host/mgpu.hpp)
int atn_mgpu_create(atn_mgpu** out, const int32_t* devices, int32_t n_devices)
{
    if (!out) return ATN_ERR_INVALID_ARG;
    *out = nullptr;
    atn_mgpu* mg = new (std::nothrow) atn_mgpu();
    if (!mg) return ATN_ERR_OUT_OF_MEMORY;
    int rc = mg_guarded(mg, [&] { return mg->m.init(devices, n_devices); });
    if (rc != ATN_OK) {
        std::fprintf(stderr, "atn_mgpu_create: %s\n", mg->m.last_error.c_str());
        delete mg;
        return rc;
    }
    *out = mg;
    return ATN_OK;
}
void atn_mgpu_destroy(atn_mgpu* mg) { delete mg; }
const char* atn_mgpu_last_error(atn_mgpu* mg) { return mg ? mg->m.last_error.c_str() : "null context"; }
int32_t atn_mgpu_shard_count(atn_mgpu* mg) { return mg ? mg->m.n : 0; }
int32_t atn_mgpu_shard_device(atn_mgpu* mg, int32_t i) { return (mg && i >= 0 && i < mg->m.n) ? mg->m.shard[i]->device : -1; }

int atn_mgpu_upload_scene(atn_mgpu* mg, const atn_scene_desc* scene)
{
    MG_OR_FAIL(mg);
    if (!scene) return mg->m.fail(ATN_ERR_INVALID_ARG, "null scene");
    return mg_guarded(mg, [&] { return mg->m.on_all([&](int i) { return mg->m.shard[i]->UpdateSceneData(scene); }); });
}
int atn_mgpu_update_tlas(atn_mgpu* mg, const atn_object_param* objects, uint32_t n_objects, const atn_mat4* matrices, uint32_t n_matrices,
                         const atn_bvh_node* top_nodes, uint32_t n_top_nodes)
{
    MG_OR_FAIL(mg);
    return mg_guarded(mg, [&] { return mg->m.on_all([&](int i) { return mg->m.shard[i]->updateBVH(objects, n_objects, matrices, n_matrices, top_nodes, n_top_nodes); }); });
}
int atn_mgpu_update_geometry(atn_mgpu* mg, const atn_vec4* vtx_pos, const atn_vec4* vtx_nml, uint32_t n_vertices, uint32_t vtx_offset,
                             const atn_triangle_param* triangles, uint32_t n_triangles, uint32_t tri_offset)
{
    MG_OR_FAIL(mg);
    return mg_guarded(mg, [&] { return mg->m.on_all([&](int i) {
        PathTracing& r = *mg->m.shard[i];
        if (hipSetDevice(r.device) != hipSuccess) return r.fail(ATN_ERR_HIP, "hipSetDevice");
        return r.updateGeometry(vtx_pos, vtx_nml, n_vertices, vtx_offset, triangles, n_triangles, tri_offset); }); });
}
int atn_mgpu_lbvh_rebuild_list(atn_mgpu* mg, uint32_t list_index, uint32_t tri_offset, uint32_t n_triangles, const float* bbox_min, const float* bbox_max)
{
    MG_OR_FAIL(mg);
    return mg_guarded(mg, [&] { return mg->m.on_all([&](int i) {
        PathTracing& r = *mg->m.shard[i];
        if (hipSetDevice(r.device) != hipSuccess) return r.fail(ATN_ERR_HIP, "hipSetDevice");
        return r.lbvh_rebuild_list(list_index, tri_offset, n_triangles, bbox_min, bbox_max, false); }); });
}
int atn_mgpu_update_camera(atn_mgpu* mg, const atn_camera_param* camera)
{
    MG_OR_FAIL(mg);
    for (auto& s : mg->m.shard) { int rc = s->updateCamera(camera); if (rc) return mg->m.fail(rc, s->last_error); }
    return ATN_OK;
}
int atn_mgpu_init_sampler(atn_mgpu* mg, int32_t w, int32_t h, int32_t seed)
{
    MG_OR_FAIL(mg);
    return mg_guarded(mg, [&]() -> int {
        if (w <= 0 || h <= 0) return mg->m.fail(ATN_ERR_INVALID_ARG, "bad sampler size");
        std::vector<uint32_t> v((size_t)w * h);         // one mt19937 pass, shared by every shard (seeds are global)
        std::mt19937 src(seed);
        for (auto& x : v) x = (uint32_t)src();
        return mg->m.on_all([&](int i) { return mg->m.shard[i]->setRandom(v.data(), (uint32_t)v.size()); });
    });
}
int atn_mgpu_set_random(atn_mgpu* mg, const uint32_t* seeds, uint32_t n)
{
    MG_OR_FAIL(mg);
    return mg_guarded(mg, [&] { return mg->m.on_all([&](int i) { return mg->m.shard[i]->setRandom(seeds, n); }); });
}
int atn_mgpu_render(atn_mgpu* mg, const atn_destination* dst, atn_vec4* out_host)
{
    MG_OR_FAIL(mg);
    return mg_guarded(mg, [&] { return mg->m.render(dst, out_host); });
}
int atn_mgpu_render_burst(atn_mgpu* mg, const atn_destination* dst, int32_t n_frames, atn_vec4* out_host)
{
    MG_OR_FAIL(mg);
    return mg_guarded(mg, [&] { return mg->m.render_burst(dst, n_frames, out_host); });
}
int atn_mgpu_set_regeneration(atn_mgpu* mg, int32_t mode)
{
    MG_OR_FAIL(mg);
    if (mode < 0 || mode > 1) return mg->m.fail(ATN_ERR_INVALID_ARG, "regeneration mode out of range");
    return mg_guarded(mg, [&] {
        int rc = mg->m.synchronize();
        if (rc) return rc;
        for (int i = 0; i < mg->m.n; i++) mg->m.shard[i]->regen_mode = mode;
        return (int)ATN_OK;
    });
}
int atn_mgpu_reset(atn_mgpu* mg)
{
    MG_OR_FAIL(mg);
    return mg_guarded(mg, [&] { return mg->m.on_all([&](int i) { return mg->m.shard[i]->reset(); }); });
}
int atn_mgpu_synchronize(atn_mgpu* mg) { MG_OR_FAIL(mg); return mg_guarded(mg, [&] { return mg->m.synchronize(); }); }
int atn_mgpu_set_frames_in_flight(atn_mgpu* mg, int32_t n)
{
    MG_OR_FAIL(mg);
    return mg_guarded(mg, [&] { return mg->m.on_all([&](int i) { return mg->m.shard[i]->set_frames_in_flight(n); }); });
}
void* atn_mgpu_film_device(atn_mgpu* mg) { return mg ? (void*)mg->m.full.p : nullptr; }
int atn_mgpu_download_film(atn_mgpu* mg, atn_vec4* out_host)
{
    MG_OR_FAIL(mg);
    atn::MultiGpu& m = mg->m;
    if (!out_host || !m.full.p || m.width <= 0) return m.fail(ATN_ERR_INVALID_ARG, "atn_mgpu_download_film: nothing rendered yet");
    if (hipSetDevice(m.shard[0]->device) != hipSuccess
        || hipMemcpyAsync(out_host, m.full.p, (size_t)m.width * m.height * sizeof(float4), hipMemcpyDeviceToHost, m.comm) != hipSuccess
        || hipStreamSynchronize(m.comm) != hipSuccess) return m.fail(ATN_ERR_HIP, "film download failed");
    return ATN_OK;
}

// ---------------------------------------------------------------- stage entry points
int atn_generate_paths(atn_ctx* ctx, int32_t width, int32_t height, int32_t sample, uint32_t frame, atn_ray* out_host)
{
    CTX_QUIET_OR_FAIL(ctx);
    PathTracing& r = ctx->r;
    if (!r.has_camera || r.n_seeds == 0 || !out_host) return r.fail(ATN_ERR_INVALID_ARG, "atn_generate_paths: camera / sampler / output missing");
    C_HIP(r, hipSetDevice(r.device));
    const int32_t sr = r.rank, sw = r.world;
    r.rank = 0; r.world = 1;
    int rc = r.ensure_frame(width, height, 1);
    if (rc) { r.rank = sr; r.world = sw; return rc; }
    atn_destination d{}; d.width = width; d.height = height; d.maxDepth = 1; d.russianRouletteDepth = 1; d.sample = 1; d.frame = frame;
    atn::FrameParams fp = r.frame_params(d);
    fp.sample = sample;
    atn::PathBuffers pb = r.buffers(false);
    atn::DevBuf<atn_ray> out;
    C_HIP(r, out.resize((size_t)width * height));
    C_HIP(r, hipMemsetAsync(r.counters.p, 0, (size_t)4 * r.counters_depth * 4, r.stream));
    if (sample > 0) C_HIP(r, hipMemsetAsync(r.done.p, 0, (size_t)r.n_slots * 4, r.stream));
    hipLaunchKernelGGL(atn::k_gen_path, dim3(PathTracing::grid_for(r.n_slots)), dim3(256), 0, r.stream, pb, fp, r.camera, (const uint32_t*)r.seeds.p);
    hipLaunchKernelGGL(atn::k_export_rays, dim3((r.n_slots + 255) / 256), dim3(256), 0, r.stream, pb, fp, out.p);
    C_HIP(r, hipMemcpyAsync(out_host, out.p, (size_t)width * height * sizeof(atn_ray), hipMemcpyDeviceToHost, r.stream));
    C_HIP(r, hipStreamSynchronize(r.stream));
    r.rank = sr; r.world = sw;      // (the next render re-validates its path buffers against its own shard's slot count)
    return ATN_OK;
}

int atn_trace_closest(atn_ctx* ctx, const atn_ray* rays_host, uint32_t n, float t_min, float t_max,
                      atn_intersection* out_host, uint64_t* stats_out)
{
    CTX_QUIET_OR_FAIL(ctx);
    PathTracing& r = ctx->r;
    if (!r.has_scene) return r.fail(ATN_ERR_NO_SCENE, "atn_upload_scene has not been called");
    if (!rays_host || !out_host) return r.fail(ATN_ERR_INVALID_ARG, "null rays / output");
    if (n == 0) { if (stats_out) { stats_out[0] = stats_out[1] = 0; } return ATN_OK; }
    C_HIP(r, hipSetDevice(r.device));
    atn::DevBuf<atn_ray> rays; atn::DevBuf<atn_intersection> out; atn::DevBuf<unsigned long long> st;
    C_HIP(r, rays.resize(n)); C_HIP(r, out.resize(n)); C_HIP(r, st.resize(8));
    C_HIP(r, hipMemcpyAsync(rays.p, rays_host, (size_t)n * sizeof(atn_ray), hipMemcpyHostToDevice, r.stream));
    C_HIP(r, hipMemsetAsync(st.p, 0, 64, r.stream));
    {
        // the probe exercises the walk the scene's tree calls for (or the forced one), whatever n is
        const bool probe_refill = r.flavour_forced ? r.use_refill : r.tree_is_deep;
        r.use_refill = probe_refill;        // trace_grid sizes the launch for it
        const dim3 g(r.trace_grid(n)), t(probe_refill ? (uint32_t)atn::kTraceBlock : 256u);
        const uint32_t lds = 0u;
        const atn_ray* rp = rays.p;
        if (stats_out) {
            if (probe_refill) hipLaunchKernelGGL((atn::k_trace_batch<true, true>), g, t, lds, r.stream, r.scene, rp, n, t_min, t_max, out.p, st.p);
            else hipLaunchKernelGGL((atn::k_trace_batch<true, false>), g, t, lds, r.stream, r.scene, rp, n, t_min, t_max, out.p, st.p);
        }
        else {
            if (probe_refill) hipLaunchKernelGGL((atn::k_trace_batch<false, true>), g, t, lds, r.stream, r.scene, rp, n, t_min, t_max, out.p, st.p);
            else hipLaunchKernelGGL((atn::k_trace_batch<false, false>), g, t, lds, r.stream, r.scene, rp, n, t_min, t_max, out.p, st.p);
        }
    }
    C_HIP(r, hipGetLastError());
    C_HIP(r, hipMemcpyAsync(out_host, out.p, (size_t)n * sizeof(atn_intersection), hipMemcpyDeviceToHost, r.stream));
    unsigned long long hs[8] = {};
    C_HIP(r, hipMemcpyAsync(hs, st.p, 64, hipMemcpyDeviceToHost, r.stream));
    C_HIP(r, hipStreamSynchronize(r.stream));
    if (stats_out) { stats_out[0] = hs[3]; stats_out[1] = hs[4]; }
    return ATN_OK;
}

int atn_cmj_samples(atn_ctx* ctx, uint32_t index, uint32_t dimension, uint32_t scramble, int32_t n, float* out_host)
{
    CTX_QUIET_OR_FAIL(ctx);
    PathTracing& r = ctx->r;
    if (n <= 0 || !out_host) return r.fail(ATN_ERR_INVALID_ARG, "bad sample count");
    C_HIP(r, hipSetDevice(r.device));
    atn::DevBuf<float> out;
    C_HIP(r, out.resize(n));
    hipLaunchKernelGGL(atn::k_cmj_samples, dim3(1), dim3(64), 0, r.stream, index, dimension, scramble, n, out.p);
    C_HIP(r, hipMemcpyAsync(out_host, out.p, (size_t)n * 4, hipMemcpyDeviceToHost, r.stream));
    C_HIP(r, hipStreamSynchronize(r.stream));
    return ATN_OK;
}

int atn_cmj_batch(atn_ctx* ctx, uint32_t n, const uint32_t* index, const uint32_t* dimension, const uint32_t* scramble,
                  int32_t draws, float* out_host)
{
    CTX_QUIET_OR_FAIL(ctx);
    PathTracing& r = ctx->r;
    if (n == 0 || draws <= 0 || !index || !dimension || !scramble || !out_host) return r.fail(ATN_ERR_INVALID_ARG, "bad cmj batch");
    C_HIP(r, hipSetDevice(r.device));
    atn::DevBuf<uint32_t> di, dd, ds; atn::DevBuf<float> out;
    C_HIP(r, di.resize(n)); C_HIP(r, dd.resize(n)); C_HIP(r, ds.resize(n)); C_HIP(r, out.resize((size_t)n * draws));
    C_HIP(r, hipMemcpyAsync(di.p, index, 4 * (size_t)n, hipMemcpyHostToDevice, r.stream));
    C_HIP(r, hipMemcpyAsync(dd.p, dimension, 4 * (size_t)n, hipMemcpyHostToDevice, r.stream));
    C_HIP(r, hipMemcpyAsync(ds.p, scramble, 4 * (size_t)n, hipMemcpyHostToDevice, r.stream));
    hipLaunchKernelGGL(atn::k_cmj_batch, dim3((n + 255) / 256), dim3(256), 0, r.stream, n, (const uint32_t*)di.p,
                       (const uint32_t*)dd.p, (const uint32_t*)ds.p, (int)draws, out.p);
    C_HIP(r, hipGetLastError());
    C_HIP(r, hipMemcpyAsync(out_host, out.p, 4 * (size_t)n * draws, hipMemcpyDeviceToHost, r.stream));
    C_HIP(r, hipStreamSynchronize(r.stream));
    return ATN_OK;
}

int atn_ray_offset(atn_ctx* ctx, uint32_t n, const float* origins, const float* normals, float* out_host)
{
    CTX_QUIET_OR_FAIL(ctx);
    PathTracing& r = ctx->r;
    if (n == 0 || !origins || !normals || !out_host) return r.fail(ATN_ERR_INVALID_ARG, "bad ray offset batch");
    C_HIP(r, hipSetDevice(r.device));
    atn::DevBuf<float> o, nm, out;
    C_HIP(r, o.resize(3 * (size_t)n)); C_HIP(r, nm.resize(3 * (size_t)n)); C_HIP(r, out.resize(3 * (size_t)n));
    C_HIP(r, hipMemcpyAsync(o.p, origins, 12 * (size_t)n, hipMemcpyHostToDevice, r.stream));
    C_HIP(r, hipMemcpyAsync(nm.p, normals, 12 * (size_t)n, hipMemcpyHostToDevice, r.stream));
    hipLaunchKernelGGL(atn::k_ray_offset, dim3((n + 255) / 256), dim3(256), 0, r.stream, n, (const float*)o.p, (const float*)nm.p, out.p);
    C_HIP(r, hipGetLastError());
    C_HIP(r, hipMemcpyAsync(out_host, out.p, 12 * (size_t)n, hipMemcpyDeviceToHost, r.stream));
    C_HIP(r, hipStreamSynchronize(r.stream));
    return ATN_OK;
}

int atn_libm_probe(atn_ctx* ctx, int32_t kind, uint32_t n, const float* a, const float* b, float* out_host)
{
    CTX_QUIET_OR_FAIL(ctx);
    PathTracing& r = ctx->r;
    if (n == 0 || !a || !b || !out_host || kind < 0 || kind > 10) return r.fail(ATN_ERR_INVALID_ARG, "bad libm probe");
    C_HIP(r, hipSetDevice(r.device));
    atn::DevBuf<float> da, db, out;
    C_HIP(r, da.resize(n)); C_HIP(r, db.resize(n)); C_HIP(r, out.resize(n));
    C_HIP(r, hipMemcpyAsync(da.p, a, 4 * (size_t)n, hipMemcpyHostToDevice, r.stream));
    C_HIP(r, hipMemcpyAsync(db.p, b, 4 * (size_t)n, hipMemcpyHostToDevice, r.stream));
    hipLaunchKernelGGL(atn::k_libm_probe, dim3((n + 255) / 256), dim3(256), 0, r.stream, kind, n, (const float*)da.p, (const float*)db.p, out.p);
    C_HIP(r, hipGetLastError());
    C_HIP(r, hipMemcpyAsync(out_host, out.p, 4 * (size_t)n, hipMemcpyDeviceToHost, r.stream));
    C_HIP(r, hipStreamSynchronize(r.stream));
    return ATN_OK;
}

int atn_material_table(atn_ctx* ctx, int32_t mtrl_id, uint32_t n, const float* nrm, const float* wi,
                       const uint32_t* index, const uint32_t* dimension, const uint32_t* scramble, const float* uv,
                       float* out_sample, float* out_eval)
{
    CTX_QUIET_OR_FAIL(ctx);
    PathTracing& r = ctx->r;
    if (!r.has_scene) return r.fail(ATN_ERR_NO_SCENE, "atn_upload_scene has not been called");
    if (mtrl_id < 0 || mtrl_id >= r.scene.n_materials || n == 0) return r.fail(ATN_ERR_INVALID_ARG, "bad material id / count");
    C_HIP(r, hipSetDevice(r.device));
    atn::DevBuf<float> dn, dw, duv, ds, de; atn::DevBuf<uint32_t> di, dsc, ddim;
    C_HIP(r, dn.resize(3 * (size_t)n)); C_HIP(r, dw.resize(3 * (size_t)n)); C_HIP(r, duv.resize(2 * (size_t)n));
    C_HIP(r, ds.resize(7 * (size_t)n)); C_HIP(r, de.resize(5 * (size_t)n)); C_HIP(r, di.resize(n)); C_HIP(r, dsc.resize(n));
    if (dimension) { C_HIP(r, ddim.resize(n)); C_HIP(r, hipMemcpyAsync(ddim.p, dimension, 4 * (size_t)n, hipMemcpyHostToDevice, r.stream)); }
    C_HIP(r, hipMemcpyAsync(dn.p, nrm, 12 * (size_t)n, hipMemcpyHostToDevice, r.stream));
    C_HIP(r, hipMemcpyAsync(dw.p, wi, 12 * (size_t)n, hipMemcpyHostToDevice, r.stream));
    C_HIP(r, hipMemcpyAsync(duv.p, uv, 8 * (size_t)n, hipMemcpyHostToDevice, r.stream));
    C_HIP(r, hipMemcpyAsync(di.p, index, 4 * (size_t)n, hipMemcpyHostToDevice, r.stream));
    C_HIP(r, hipMemcpyAsync(dsc.p, scramble, 4 * (size_t)n, hipMemcpyHostToDevice, r.stream));
    hipLaunchKernelGGL(atn::k_material_table, dim3((n + 63) / 64), dim3(64), 0, r.stream, r.scene, mtrl_id, n,
                       (const float*)dn.p, (const float*)dw.p, (const uint32_t*)di.p, (const uint32_t*)(dimension ? ddim.p : nullptr), (const uint32_t*)dsc.p, (const float*)duv.p, ds.p, de.p);
    C_HIP(r, hipGetLastError());
    C_HIP(r, hipMemcpyAsync(out_sample, ds.p, 28 * (size_t)n, hipMemcpyDeviceToHost, r.stream));
    C_HIP(r, hipMemcpyAsync(out_eval, de.p, 20 * (size_t)n, hipMemcpyDeviceToHost, r.stream));
    C_HIP(r, hipStreamSynchronize(r.stream));
    return ATN_OK;
}

int atn_material_eval(atn_ctx* ctx, int32_t mtrl_id, uint32_t n, const float* nrm, const float* wi, const float* wo, const float* uv, float* out_eval)
{
    CTX_QUIET_OR_FAIL(ctx);
    PathTracing& r = ctx->r;
    if (!r.has_scene) return r.fail(ATN_ERR_NO_SCENE, "atn_upload_scene has not been called");
    if (mtrl_id < 0 || mtrl_id >= r.scene.n_materials || n == 0) return r.fail(ATN_ERR_INVALID_ARG, "bad material id / count");
    if (!nrm || !wi || !wo || !uv || !out_eval) return r.fail(ATN_ERR_INVALID_ARG, "null argument");
    return guarded(ctx, [&]() -> int {
        C_HIP(r, hipSetDevice(r.device));
        atn::DevBuf<float> dn, dw, dwo, duv, de;
        C_HIP(r, dn.resize(3 * (size_t)n)); C_HIP(r, dw.resize(3 * (size_t)n)); C_HIP(r, dwo.resize(3 * (size_t)n));
        C_HIP(r, duv.resize(2 * (size_t)n)); C_HIP(r, de.resize(5 * (size_t)n));
        C_HIP(r, hipMemcpyAsync(dn.p, nrm, 12 * (size_t)n, hipMemcpyHostToDevice, r.stream));
        C_HIP(r, hipMemcpyAsync(dw.p, wi, 12 * (size_t)n, hipMemcpyHostToDevice, r.stream));
        C_HIP(r, hipMemcpyAsync(dwo.p, wo, 12 * (size_t)n, hipMemcpyHostToDevice, r.stream));
        C_HIP(r, hipMemcpyAsync(duv.p, uv, 8 * (size_t)n, hipMemcpyHostToDevice, r.stream));
        hipLaunchKernelGGL(atn::k_material_eval, dim3((n + 255) / 256), dim3(256), 0, r.stream, r.scene, mtrl_id, n,
                           (const float*)dn.p, (const float*)dw.p, (const float*)dwo.p, (const float*)duv.p, de.p);
        C_HIP(r, hipGetLastError());
        C_HIP(r, hipMemcpyAsync(out_eval, de.p, 20 * (size_t)n, hipMemcpyDeviceToHost, r.stream));
        C_HIP(r, hipStreamSynchronize(r.stream));
        return ATN_OK;
    });
}

int atn_compact2(atn_ctx* ctx, const int32_t* flags_a_host, const int32_t* flags_b_host, uint32_t n, uint32_t grid_blocks,
                 int32_t* out_a_host, uint32_t* out_count_a, int32_t* out_b_host, uint32_t* out_count_b)
{
    CTX_QUIET_OR_FAIL(ctx);
    PathTracing& r = ctx->r;
    if (!flags_a_host || !out_a_host || !out_count_a) return r.fail(ATN_ERR_INVALID_ARG, "null argument");
    if (flags_b_host && (!out_b_host || !out_count_b)) return r.fail(ATN_ERR_INVALID_ARG, "null output for the second queue");
    *out_count_a = 0;
    if (out_count_b) *out_count_b = 0;
    if (n == 0) return ATN_OK;
    return guarded(ctx, [&]() -> int {
        C_HIP(r, hipSetDevice(r.device));
        atn::DevBuf<int32_t> fa, fb; atn::DevBuf<uint32_t> oa, ob, c;
        C_HIP(r, fa.resize(n)); C_HIP(r, oa.resize(n)); C_HIP(r, c.resize(2));
        C_HIP(r, hipMemcpyAsync(fa.p, flags_a_host, 4 * (size_t)n, hipMemcpyHostToDevice, r.stream));
        if (flags_b_host) {
            C_HIP(r, fb.resize(n)); C_HIP(r, ob.resize(n));
            C_HIP(r, hipMemcpyAsync(fb.p, flags_b_host, 4 * (size_t)n, hipMemcpyHostToDevice, r.stream));
        }
        C_HIP(r, hipMemsetAsync(c.p, 0, 8, r.stream));
        uint32_t grid = grid_blocks ? grid_blocks : PathTracing::grid_for((n + (uint32_t)atn::kChunkItems - 1u) / (uint32_t)atn::kChunkItems);
        hipLaunchKernelGGL(atn::k_compact_append, dim3(grid), dim3(256), 0, r.stream, (const int32_t*)fa.p, (const int32_t*)fb.p, n,
                           oa.p, c.p, ob.p, c.p + 1);
        C_HIP(r, hipGetLastError());
        uint32_t hc[2] = { 0, 0 };
        C_HIP(r, hipMemcpyAsync(hc, c.p, 8, hipMemcpyDeviceToHost, r.stream));
        C_HIP(r, hipStreamSynchronize(r.stream));
        if (hc[0] > n || hc[1] > n) return r.fail(ATN_ERR_HIP, "queue append reserved more entries than exist");
        *out_count_a = hc[0];
        if (hc[0]) C_HIP(r, hipMemcpyAsync(out_a_host, oa.p, 4 * (size_t)hc[0], hipMemcpyDeviceToHost, r.stream));
        if (flags_b_host) {
            *out_count_b = hc[1];
            if (hc[1]) C_HIP(r, hipMemcpyAsync(out_b_host, ob.p, 4 * (size_t)hc[1], hipMemcpyDeviceToHost, r.stream));
        }
        C_HIP(r, hipStreamSynchronize(r.stream));
        return ATN_OK;
    });
}

int atn_compact(atn_ctx* ctx, const int32_t* flags_host, uint32_t n, int32_t* out_idx_host, uint32_t* out_count)
{
    // the product's unordered block append, put into index order on the host: the entries ARE the indices, so the
    // sorted queue is the stable compaction
    int rc = atn_compact2(ctx, flags_host, nullptr, n, 0, out_idx_host, out_count, nullptr, nullptr);
    if (rc == ATN_OK && *out_count) std::sort(out_idx_host, out_idx_host + *out_count);
    return rc;
}

uint32_t atn_sizeof_scene_desc(void) { return (uint32_t)sizeof(atn_scene_desc); }
uint32_t atn_sizeof_destination(void) { return (uint32_t)sizeof(atn_destination); }
// what this binary was built from: aten_amd.build.kernel_sources_sha16() of the sources + the extra compile flags, passed in by
// the build recipe (aten_amd/build.py, tools/build_variants.sh).  bench.py uses PMC records only for the binary they were taken on.
#ifndef ATN_BUILD_ID
#define ATN_BUILD_ID "unknown"
#endif
const char* atn_build_id(void) { return ATN_BUILD_ID; }
uint32_t atn_abi_version(void) { return 4; }     // 4: atn_bank_streams' *concurrent is a bit mask (bit 0 = banks overlap, bit 1 = the side stream does) since r05; r06 adds atn_set_regeneration / atn_render_burst / atn_regen_stage_counts / atn_set_upload_options / atn_mgpu_render_burst
//     // 3: atn_material_table takes the starting dimension, atn_compact3 dropped (r04); 2: atn_scene_desc grew the NPR fields, atn_toon_param spelled out (r02)

} // extern "C"
