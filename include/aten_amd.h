/*
 * aten_amd.h -- C-ABI of the MI355X path-tracing integrator (libaten_amd.so).
 *
 * Drop-in boundary: these entry points take exactly the flat arrays that aten's existing GPU
 * seam, idaten::Renderer (src/libidaten/kernel/renderer.h:17-179 of the reference), extracts
 * from aten::context, and perform what idaten::PathTracing does with them -- but with the
 * sample stream and semantics of the CPU renderer aten::PathTracing (the parity target).
 * INTEGRATION.md shows the C++ adapter an aten application adds.
 *
 * All functions return 0 on success or a negative atn_status; they never throw.  A context is
 * not thread-safe (the reference's renderer objects are single-threaded, one per window thread);
 * distinct contexts are independent.  The library fails loudly (ATN_ERR_NO_DEVICE) when no HIP
 * device is present: there is no CPU fallback.
 */
#ifndef ATEN_AMD_H_
#define ATEN_AMD_H_

#include "aten_layout.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct atn_ctx atn_ctx;

typedef enum atn_status {
    ATN_OK = 0,
    ATN_ERR_INVALID_ARG = -1,
    ATN_ERR_NO_DEVICE = -2,
    ATN_ERR_HIP = -3,
    ATN_ERR_NO_SCENE = -4,
    ATN_ERR_UNSUPPORTED = -5,
    ATN_ERR_OUT_OF_MEMORY = -6
} atn_status;

/* aten::Destination (src/libaten/renderer/renderer.h:15-23) plus what the reference keeps as
 * renderer state: the frame counter (renderer.h:31-39,71; CPU starts at 0, CUDA at 1) and the
 * film flavour (Film vs FilmProgressive, src/libaten/renderer/film.cpp:33-71). */
typedef struct atn_destination {
    int32_t width;
    int32_t height;
    int32_t maxDepth;
    int32_t russianRouletteDepth;
    int32_t sample;                 /* samples per pixel this frame */
    uint32_t frame;                 /* aten::Renderer::GetFrameCount() */
    int32_t progressive;            /* 1 = FilmProgressive::put running mean, 0 = Film::put overwrite */
    int32_t break_on_terminate;     /* 1 = reproduce pathtracing.cpp:350-352 (stop sampling a pixel after a
                                       terminated path; the CPU renderer's behaviour), 0 = take all samples */
    int32_t count_stats;            /* 1 = count rays / node visits / triangle tests (slower; atn_get_stats) */
    int32_t profile;                /* 1 = bracket every launch with HIP events (atn_get_kernel_times) */
} atn_destination;

/* ≙ idaten::Renderer construction + cudaSetDevice.  device_ordinal: HIP device index. */
int atn_create(atn_ctx** out, int device_ordinal);
void atn_destroy(atn_ctx* ctx);
const char* atn_last_error(atn_ctx* ctx);

/* ≙ idaten::Renderer::UpdateSceneData (src/libidaten/kernel/renderer.cpp:12-131): copies the
 * scene to HBM (and re-lays-out the BVH).  The caller keeps ownership of `scene`. */
int atn_upload_scene(atn_ctx* ctx, const atn_scene_desc* scene);

/* ≙ idaten::Renderer::updateBVH (src/libidaten/kernel/renderer.cpp:133-153): new ObjectParameters and
 * matrices (aten::context::GetObjectParametersAndMatrices) and a rebuilt TOP layer (`nodes[0]`); the
 * bottom-level lists uploaded by atn_upload_scene stay in place.  n_matrices == 0 keeps the old
 * matrices, like the reference.  The list indices in the leaves' exid refer to the uploaded lists. */
int atn_update_tlas(atn_ctx* ctx, const atn_object_param* objects, uint32_t n_objects,
                    const atn_mat4* matrices, uint32_t n_matrices,
                    const atn_bvh_node* top_nodes, uint32_t n_top_nodes);

/* How atn_upload_scene lays the scene out.  Context state, initialised ONCE at atn_create from the environment variables of the same
 * names (README.md) and changed only here -- not re-read from the process environment at every upload.  -1 leaves an option as it is.
 *   anyhit_twin       0 / 1 / 2: no any-hit twins / where the surface-area model expects them to pay (default) / wherever possible
 *   anyhit_twin_dirs  8 / 1: one twin per octant of the ray direction (default) / the one direction-free twin
 *   node_layout       1 / 0: first levels of a list level by level (default) / walk order
 *   planar_lights     1 / 0: shadow rays towards planar, rigidly placed area lights stop at the first nearer hit (default) / never
 * Films do not depend on any of them (tests/test_gpu_anyhit_twin.py). */
int atn_set_upload_options(atn_ctx* ctx, int32_t anyhit_twin, int32_t anyhit_twin_dirs, int32_t node_layout, int32_t planar_lights);

/* ≙ idaten::Renderer::updateCamera (renderer.cpp:202-205). */
int atn_update_camera(atn_ctx* ctx, const atn_camera_param* camera);

/* Scene updates between frames -- atn_update_tlas, atn_update_geometry, atn_lbvh_rebuild_list -- return when they are
 * ENQUEUED: the caller's arrays have been copied (they may be reused at once), the frames in flight keep running, and
 * every frame rendered afterwards sees the update.  With atn_set_frames_in_flight > 1 the mutable part of the scene is
 * kept once per frame in flight (up to three copies) on the device for this. */

/* ---- dynamic geometry: the per-tick sequence of the reference's deformation renderer
 * (src/deformation_renderer/main.cpp:636-710): skinned vertices -> LBVHBuilder::build into the renderer's node list ->
 * Renderer::updateGeometry -> Renderer::updateBVH (= atn_update_tlas). */

/* ≙ idaten::Renderer::updateGeometry (src/libidaten/kernel/renderer.cpp:155-215): overwrite vertices
 * [vtx_offset, vtx_offset + n_vertices) and triangles [tri_offset, tri_offset + n_triangles) of the uploaded scene.  Host
 * pointers; vtx_pos or vtx_nml may be NULL (left as they are).  Bottom-level lists over these triangles must be rebuilt
 * (atn_lbvh_rebuild_list) before the next render: their leaf records hold vertex data. */
int atn_update_geometry(atn_ctx* ctx, const atn_vec4* vtx_pos, const atn_vec4* vtx_nml, uint32_t n_vertices, uint32_t vtx_offset,
                        const atn_triangle_param* triangles, uint32_t n_triangles, uint32_t tri_offset);
/* The scene arrays in device memory (float4[n_vertices] x 2, atn_triangle_param[n_triangles]) for a caller whose own HIP
 * skinning kernel writes them in place (≙ the interop VBO the reference's skinning writes, main.cpp:673-676).  Waits for
 * the frames in flight and switches the context to ONE copy of the scene, updated in place behind the frames in flight
 * (the caller's kernel must itself run after them, e.g. after atn_synchronize); the pointers stay valid until the next
 * atn_upload_scene. */
int atn_scene_device_arrays(atn_ctx* ctx, void** vtx_pos, void** vtx_nml, void** triangles);
/* ≙ lbvh_.build(nodes[deformPos], tris, tri_offset_, sceneBbox, vtxPos, ...) (main.cpp:686-693; idaten::LBVHBuilder::onBuild,
 * src/libidaten/kernel/LBVHBuilder.cu:700-810): rebuild bottom-level list `list_index` ON THE DEVICE as an LBVH over the
 * scene's triangles [tri_offset, tri_offset + n_triangles) as they are in device memory now; bbox_* is the box the Morton
 * codes are normalised with (the reference passes the skinned mesh's box).  The tree is the reference's, node for node;
 * its records replace the list's region of the node image, which must hold exactly n - 1 inner and n leaf records (a list
 * uploaded as a binary tree with one leaf per triangle).  ATN_ERR_UNSUPPORTED otherwise. */
int atn_lbvh_rebuild_list(atn_ctx* ctx, uint32_t list_index, uint32_t tri_offset, uint32_t n_triangles,
                          const float bbox_min[3], const float bbox_max[3]);
/* ≙ idaten::LBVHBuilder::build(dst, std::vector<TriangleParameter>&, triIdOffset, sceneBbox, vtxPos, vtxOffset,
 * threadedBvhNodes) (LBVHBuilder.cu:812-833): the same builder as a function of host arrays, returning
 * ThreadedBvhNode[2 n - 1] in the reference's node order (inner nodes 0 .. n-2, leaves n-1 .. 2n-2; leaf f0 = 1,
 * f1 = tri_id_offset + triangle index).  out_sorted_codes / out_sorted_indices (n each) may be NULL.  Needs no scene. */
int atn_lbvh_build(atn_ctx* ctx, const atn_triangle_param* triangles, uint32_t n_triangles, int32_t tri_id_offset,
                   const float bbox_min[3], const float bbox_max[3], const atn_vec4* vtx_pos, uint32_t n_vertices, int32_t vtx_offset,
                   atn_bvh_node* out_nodes, uint32_t* out_sorted_codes, uint32_t* out_sorted_indices);

/* ≙ aten::initSampler(width, height, seed) (src/libaten/sampler/sampler.cpp:8-18) followed by
 * idaten::PathTracing::initSamplerParameter's upload of aten::getRandom()
 * (src/libidaten/kernel/renderer.h:113-124): one std::mt19937(seed) draw per pixel. */
int atn_init_sampler(atn_ctx* ctx, int32_t width, int32_t height, int32_t seed);
/* Same, with caller-provided seeds (aten::getRandom()). */
int atn_set_random(atn_ctx* ctx, const uint32_t* seeds, uint32_t n);
/* ≙ aten::getRandom() (src/libaten/sampler/sampler.cpp:20-23): the first n entries of the table the kernels
 * read, copied back from HBM; atn_random_count = its size (0 before atn_init_sampler / atn_set_random). */
int atn_get_random(atn_ctx* ctx, uint32_t* out_host, uint32_t n);
uint32_t atn_random_count(atn_ctx* ctx);

/* Screen-space sharding for multi-GPU: the image is cut into 8x8-pixel tiles, tile t (row-major)
 * is rendered by rank t % world.  Default (0, 1) = whole image.  No reference analogue (the
 * reference is single-GPU). */
int atn_set_screen_shard(atn_ctx* ctx, int32_t rank, int32_t world);

/* ≙ idaten::PathTracing::render(width, height, maxSamples, maxBounce)
 * (src/libidaten/kernel/pathtracing.cpp:49-153) with aten::PathTracing::OnRender's semantics
 * (src/libaten/renderer/pathtracing/pathtracing.cpp:269-366).
 * out_host: optional vec4[width*height], row 0 = bottom (like aten::Film); NULL = stay on device. */
int atn_render(atn_ctx* ctx, const atn_destination* dst, atn_vec4* out_host);

/* ---- path regeneration (BASELINE.json north_star: "path compaction/regeneration") ---------------------------------------------
 * The loop behind atn_render is the reference's (src/libidaten/kernel/pathtracing.cpp:105-138): sample by sample, bounce by bounce, one
 * compaction per bounce, on a ray population that decays until the sample's longest path is over.  With regeneration on, a frame's
 * samples -- and, through atn_render_burst, a run of consecutive progressive frames -- share one POOL of path slots, one per pixel: the
 * moment a pixel's path ends, the shade kernel runs its sample epilogue (pathtracing.cpp:339-352; after the frame's last sample
 * Film::put / FilmProgressive::put, film.cpp:33-71) and writes the pixel's next primary ray (GeneratePath, pathtracing_impl.h:65-110)
 * into the same slot, so every launch works on a full population.  Per pixel the samples and frames keep the serial order and
 * arithmetic: films are BYTE-equal to the serial loop's, with both values of break_on_terminate, on any screen shard.
 *
 * atn_set_regeneration: 0 = the serial loop (default), 1 = the pool wherever it applies: atn_render with sample > 1 and atn_render_burst
 * (not with count_stats -- counted frames use the serial loop's counting kernels -- and not atn_svgf_render, whose AOVs are per frame).
 * atn_render_burst: n_frames x atn_render with frame = dst->frame, dst->frame + 1, ... (the serial loop does exactly that when
 * regeneration is off or does not apply; n_frames > 1 regenerates only with dst->progressive); out_host receives the film after the last.
 * atn_regen_stage_counts: closest-hit rays and shadow rays of every launch of the last regenerated burst (the pool's occupancy),
 * n_stages entries (at most `capacity` written); either array may be NULL. */
int atn_set_regeneration(atn_ctx* ctx, int32_t mode);
int32_t atn_get_regeneration(atn_ctx* ctx);
int atn_render_burst(atn_ctx* ctx, const atn_destination* dst, int32_t n_frames, atn_vec4* out_host);
int atn_regen_stage_counts(atn_ctx* ctx, uint32_t* closest, uint32_t* shadow, uint32_t capacity, uint32_t* n_stages);

/* Floating-point rules of the shade kernel.  0 (default): every operation rounds as the CPU renderer's SSE2 build does -- no fused
 * multiply-add, correctly rounded division and square root, ocml's sinf / cosf / ...: the parity path, what every test and the
 * headline number use.  1: the rules of the reference's own GPU build (src/libidaten/CMakeLists.txt:188, nvcc --use_fast_math): fused
 * multiply-adds, approximate division / square root, the hardware's transcendentals, denormals flushed.  Frames then differ from the
 * CPU renderer's beyond the parity tolerance (measured: DESIGN.md section 7f); serial loop only (not the regenerated pool, not SVGF). */
int atn_set_shade_math(atn_ctx* ctx, int32_t mode);

/* ≙ idaten::Renderer::reset (renderer.h:40-43): clears the progressive film. */
int atn_reset(atn_ctx* ctx);

/* Device-side results of the last atn_render (valid until the next call on ctx). */
void* atn_film_device(atn_ctx* ctx);            /* float4[width*height]; only this rank's pixels are written */
void* atn_tile_device(atn_ctx* ctx);            /* float4[atn_tile_slots]: this rank's pixels in slot order */
/* How many bottom-level lists of the uploaded scene have an any-hit twin right now (csrc/host/anyhit_twin.hpp: a second threading
 * of the same tree that the shadow rays of infinite lights walk; results do not depend on it).  ATEN_AMD_ANYHIT_TWIN = 0 / 1 / 2 at
 * upload: none / where the surface-area model expects it to pay (default) / wherever a list is a binary tree.  An LBVH rebuild of a
 * list re-threads its twins on the device from the new tree. */
uint32_t atn_anyhit_twins(atn_ctx* ctx);
/* How many area lights' shadow rays may stop at the first hit nearer than the light right now: lights whose object is planar and placed
 * by a rigid matrix, found at upload (csrc/host/scene_upload.hpp, planar_area_light; ATEN_AMD_PLANAR_LIGHTS=0 switches the rule off).
 * Such a ray meets the light's object at distToLight and nowhere else, so a nearer hit is on another object and scene::hitLight's answer
 * is "blocked" whatever else the walk would find.  An update keeps the flags when it hands back every such light's object records
 * (and, if it carries matrices, the instance's two matrices) byte for byte and writes none of the light's vertices or triangles -- a
 * deformation tick of another mesh, other instances moving; 0 after any other update, until the next atn_upload_scene. */
uint32_t atn_planar_area_lights(atn_ctx* ctx);
uint32_t atn_tile_slots(atn_ctx* ctx);          /* identical on every rank: ceil(n_tiles / world) * 64 */
void* atn_stream(atn_ctx* ctx);                 /* hipStream_t all work is enqueued on */
int atn_synchronize(atn_ctx* ctx);

/* Scatter an all-gathered tile buffer (rank-major, world * atn_tile_slots float4, device memory)
 * into film_dev_out (float4[width*height], device memory; NULL = the context's own film). */
int atn_assemble_tiles(atn_ctx* ctx, const void* gathered_dev, int32_t world, void* film_dev_out);
/* Same, on a caller's HIP stream (hipStream_t; NULL = the context's): lets the exchange + assembly of frame f run
 * on a communication stream while the context's stream already renders frame f + 1. */
int atn_assemble_tiles_on(atn_ctx* ctx, const void* gathered_dev, int32_t world, void* film_dev_out, void* hip_stream);
int atn_download_film(atn_ctx* ctx, atn_vec4* out_host);
/* Checkpoint / resume of the progressive film (FilmProgressive's running mean + sample count in .w,
 * src/libaten/renderer/film.cpp:61-71): what atn_download_film returned is put back, and the next progressive
 * atn_render of that size continues from it exactly as if the earlier frames had run on this context. */
int atn_upload_film(atn_ctx* ctx, int32_t width, int32_t height, const atn_vec4* film_host);

/* Counters of the last atn_render with count_stats = 1:
 * {closest rays, shadow rays, shaded hits, closest node visits, closest triangle tests,
 *  shadow node visits, shadow triangle tests, 0}. */
int atn_get_stats(atn_ctx* ctx, uint64_t out[8]);
/* The per-pixel cost map of the last frame rendered with count_stats = 1: uint32 {BVH node visits, triangle tests}[h][w] of all
 * the pixel's walks (closest and shadow, every sample and bounce).  ≙ the heat map the reference builds from its per-path GPU
 * timer (PathTimeProfiler, src/libaten/renderer/pathtracing/path_time_profiler.h:15-60; ComputeTemperature maps it to colours) --
 * the deterministic quantity behind that time.  Pixels of other ranks' tiles are 0. */
int atn_download_path_cost(atn_ctx* ctx, uint32_t* out_host);

/* Kernel classes for atn_get_kernel_times. */
enum { ATN_K_GEN = 0, ATN_K_TRACE_CLOSEST = 1, ATN_K_SHADE = 2, ATN_K_TRACE_SHADOW = 3,
       ATN_K_ACCUM = 4, ATN_K_GATHER = 5,
       ATN_K_SVGF_PREPARE = 6, ATN_K_SVGF_TEMPORAL = 7, ATN_K_SVGF_VARIANCE = 8, ATN_K_SVGF_ATROUS = 9,
       ATN_K_TRACE_FUSED = 10,      /* shadow rays of bounce b + closest-hit rays of bounce b+1 in one launch */
       ATN_K_COUNT = 11 };
/* HIP-event time (ms) and launch count per kernel class, accumulated over every atn_render with
 * profile = 1 since the last atn_reset_kernel_times. */
int atn_get_kernel_times(atn_ctx* ctx, float ms[ATN_K_COUNT], uint32_t launches[ATN_K_COUNT]);
int atn_reset_kernel_times(atn_ctx* ctx);

/* Execution knob, no effect on results: the frame's paths are processed as up to `n` independent batches on
 * `n` HIP streams so that one batch's kernels fill the GPU during another's launch tails (default 3, automatically
 * fewer for small frames / shards; 1 = strictly one kernel at a time, which is what per-kernel timings of an
 * isolated kernel need). */
int atn_set_path_batches(atn_ctx* ctx, int32_t n);

/* Execution knob, no effect on results: how many consecutive atn_render frames may be in flight on the GPU at once
 * (1 .. 4, default 1).  With n > 1 the context keeps n banks of path state and streams and atn_render(f + 1) is enqueued
 * on another stream than frame f: the launch tails of one frame overlap the bulk of the next (what limits a small frame
 * -- one GPU's share of a sharded 1080p image -- is the ~0.13 ms tail of each of its depth + 1 trace launches, DESIGN.md
 * section 8).  Only the film orders the frames.  atn_stream / atn_tile_device then refer to the LAST rendered frame;
 * every other entry point first waits for all frames in flight. */
int atn_set_frames_in_flight(atn_ctx* ctx, int32_t n);
/* Diagnostics of the above: HIP streams share a few hardware queues and two streams on one queue run one after the other, so
 * atn_set_frames_in_flight MEASURES which of its bank streams run side by side and replaces the ones that do not (DESIGN.md
 * section 8).  *swaps = streams replaced so far; *concurrent: bit 0 = every pair of the current bank streams was measured
 * to overlap (re-measured by this call; 0 = fewer hardware queues than frames in flight), bit 1 = the side stream
 * (atn_side_stream), if it was handed out, overlaps with every bank stream.  Either may be NULL. */
int atn_bank_streams(atn_ctx* ctx, int32_t* swaps, int32_t* concurrent);
/* A stream for the CALLER's own work beside the frames in flight (a tile exchange, a display copy): created on first use and
 * chosen -- by the same measurement -- to run side by side with every bank stream when a hardware queue is left for it
 * (a stream the caller creates itself lands on SOME queue, possibly a bank's, and then waits behind that bank's kernels).
 * Owned by the context; NULL on failure.  Call after atn_set_frames_in_flight. */
void* atn_side_stream(atn_ctx* ctx);

/* Optional samplers, both off by default: they change the sample stream, i.e. they leave the parity path of
 * aten::PathTracing (every default-mode result still matches the CPU renderer).
 *   ibl_importance != 0: the IBL light is sampled from the luminance tables of the environment map --
 *       ImageBasedLight::preCompute (src/libaten/light/ibl.cpp:10-118) and the table sampler ImageBasedLight::sample
 *       (ibl.cpp:180-230) -- instead of light/ibl.h:98-100's cosine-hemisphere sampling; BSDF-sampled misses are weighted
 *       with the same density.
 *   tex_bilinear != 0: every texture lookup is aten::texture::AtWithBilinear (src/libaten/image/texture.cpp:77-125)
 *       instead of texture::at (the CUDA backend filters with tex2DLod, material/sample_texture.h:18-40). */
int atn_set_sampling_options(atn_ctx* ctx, int32_t ibl_importance, int32_t tex_bilinear);
/* Stage probe: n lookups of texture `texid` at uv_host[2n] -> out_host[4n], with the current texture mode. */
int atn_sample_texture(atn_ctx* ctx, int32_t texid, uint32_t n, const float* uv_host, float* out_host);

/* ---- one node, every GPU ------------------------------------------------------------------------
 * The reference renderer is single-device (idaten::Renderer, src/libidaten/kernel/renderer.h:17-179; its caller
 * src/device_renderer/main.cpp:133-149,196-204 makes one UpdateSceneData and one render(dst) per frame).  An
 * atn_mgpu keeps exactly that call shape and spreads the frame over the node: one context + one host worker thread
 * per shard, the scene replicated, the screen cut into 8x8-pixel tiles (tile t -> shard t % N, seeds and pixel indices
 * global, so the image does not depend on N), and per frame ONE exchange: every shard pushes its tile buffer to
 * shard 0's device with a peer copy over xGMI and shard 0 scatters them into the full frame.  The film of frame f is
 * assembled while frame f + 1 renders (two gather buffers, events both ways).
 *
 * devices: HIP ordinals of the shards, n_devices entries; NULL = ordinals 0 .. n_devices-1, and NULL with
 * n_devices <= 0 = every visible device.  An ordinal may repeat (shards sharing a GPU): with that the whole N > 1 path
 * runs on a one-GPU machine.  All calls return 0 or a negative atn_status; none throws. */
typedef struct atn_mgpu atn_mgpu;
int atn_mgpu_create(atn_mgpu** out, const int32_t* devices, int32_t n_devices);
void atn_mgpu_destroy(atn_mgpu* mg);
const char* atn_mgpu_last_error(atn_mgpu* mg);
int32_t atn_mgpu_shard_count(atn_mgpu* mg);
int32_t atn_mgpu_shard_device(atn_mgpu* mg, int32_t shard);
/* ≙ UpdateSceneData / updateBVH / updateCamera / initSampler on every shard (uploads run in parallel). */
int atn_mgpu_upload_scene(atn_mgpu* mg, const atn_scene_desc* scene);
int atn_mgpu_update_tlas(atn_mgpu* mg, const atn_object_param* objects, uint32_t n_objects,
                         const atn_mat4* matrices, uint32_t n_matrices,
                         const atn_bvh_node* top_nodes, uint32_t n_top_nodes);
int atn_mgpu_update_geometry(atn_mgpu* mg, const atn_vec4* vtx_pos, const atn_vec4* vtx_nml, uint32_t n_vertices, uint32_t vtx_offset,
                             const atn_triangle_param* triangles, uint32_t n_triangles, uint32_t tri_offset);
int atn_mgpu_lbvh_rebuild_list(atn_mgpu* mg, uint32_t list_index, uint32_t tri_offset, uint32_t n_triangles,
                               const float bbox_min[3], const float bbox_max[3]);
int atn_mgpu_update_camera(atn_mgpu* mg, const atn_camera_param* camera);
int atn_mgpu_init_sampler(atn_mgpu* mg, int32_t width, int32_t height, int32_t seed);
int atn_mgpu_set_random(atn_mgpu* mg, const uint32_t* seeds, uint32_t n);
/* ≙ idaten::PathTracing::render for the whole node.  out_host NULL: returns when the frame is enqueued; the
 * assembled film (float4[w*h], row 0 = bottom) is atn_mgpu_film_device on shard 0's device, complete after
 * atn_mgpu_synchronize.  out_host != NULL: returns with the frame in host memory. */
int atn_mgpu_render(atn_mgpu* mg, const atn_destination* dst, atn_vec4* out_host);
int atn_mgpu_reset(atn_mgpu* mg);
int atn_mgpu_synchronize(atn_mgpu* mg);
void* atn_mgpu_film_device(atn_mgpu* mg);
int atn_mgpu_download_film(atn_mgpu* mg, atn_vec4* out_host);
/* atn_set_frames_in_flight on every shard. */
int atn_mgpu_set_frames_in_flight(atn_mgpu* mg, int32_t n);
/* atn_set_regeneration on every shard; atn_render_burst on every shard followed by ONE exchange of the tiles (a progressive film is
 * looked at after the burst: its tiles travel once per burst, not once per frame). */
int atn_mgpu_set_regeneration(atn_mgpu* mg, int32_t mode);
int atn_mgpu_render_burst(atn_mgpu* mg, const atn_destination* dst, int32_t n_frames, atn_vec4* out_host);

/* ---- SVGF (next tier, BASELINE config 5) -------------------------------------------------------
 * ≙ aten::SVGFRenderer (src/libaten/renderer/svgf/svgf.{h,cpp}): the path pass with AOV outputs
 * (SVGFRenderer::Shade / ShadeMiss with AOV spans) followed by the svgf_impl.h passes as HIP kernels --
 * PrepareForDenoise, TemporalReprojection + AccumulateMoments (frame > 0), EstimateVariance,
 * atrous_iter_cnt x (3x3 Gauss of the variance + ExecAtrousWaveletFilter + PostProcessForAtrousFilter),
 * CopyFromTeporaryColorBufferToAov -- with the AOV / moment buffers of two frames resident in HBM.
 * `dst->frame == 0` takes the first-frame path (svgf.cpp:519-525,543-551).  One GPU only.
 *
 * out_host (may be NULL): what dst.buffer holds when OnRender returns (the last a-trous iteration's
 * albedo-multiplied colour), vec4[w*h], row 0 = bottom.  stages_host (may be NULL): 3 x vec4[w*h], the
 * values OnRender puts after the path pass, the temporal pass and the variance pass.
 * compute_motion != 0: the motion/depth buffer is computed from the primary hits and the current/previous
 * camera matrices (MatricesForRendering, pt_params.h:150-185) -- the compute pass that stands in for the
 * reference's GL raster pass (src/shader/ssrt_fs.glsl:31-47); 0: use atn_svgf_set_motion_depth's buffer. */
int atn_svgf_render(atn_ctx* ctx, const atn_destination* dst, int32_t compute_motion,
                    atn_vec4* out_host, atn_vec4* stages_host);
/* ≙ SVGFRenderer::SetMotionDepthBuffer (svgf.cpp:441-450): {motion.xy in screen fractions, depth, 1}. */
int atn_svgf_set_motion_depth(atn_ctx* ctx, const atn_vec4* motion_depth, uint32_t n);
/* Forget the frame history (AOV sets, moments, previous camera matrices). */
int atn_svgf_reset(atn_ctx* ctx);
/* SVGFParams::atrous_iter_cnt (svgf_types.h:72), default 5. */
int atn_svgf_set_atrous_iterations(atn_ctx* ctx, int32_t n);
/* Optional pass, off by default (the CPU SVGFRenderer, the parity target, never runs it; the CUDA twin does, right after
 * temporal reprojection: src/libidaten/svgf/svgf_tp.cu:150-216): RecomputeTemporalWeightFromSurroundingPixels
 * (src/libaten/renderer/svgf/svgf_impl.h:386-423) -- a non-background pixel's temporal weight becomes the minimum over
 * its 3x3 neighbourhood, every tap reading the weights the pass started with. */
int atn_svgf_set_dilate_temporal_weight(atn_ctx* ctx, int32_t on);
/* Parity probe.  which: 0-3 SVGFParams::GetCurrAovBuffer() as the state stands (OnRender toggles at its end, so
 * this is the set the NEXT frame writes: normal+depth, albedo+meshid, colour+variance, moments+temporal weight),
 * 4-7 GetPrevAovBuffer() (the set the last frame wrote), 8 temporary colour, 9 motion/depth, 10 primary hit
 * position, 11/12 a-trous ping-pong buffers, 13 output, 14 contributions. */
int atn_svgf_download(atn_ctx* ctx, int32_t which, atn_vec4* out_host);
/* The filter passes alone (everything of OnRender after the sample loop, svgf.cpp:515-637) on whatever the
 * path pass -- or atn_svgf_upload -- left in the buffers: contributions (which = 14: contrib.xyz, sample count),
 * the current AOVs (0, 1), primary hit positions (10), motion/depth (9).  This is how a caller that already has a
 * noisy frame and a G-buffer uses the denoiser, and how the parity tests feed both sides identical inputs. */
int atn_svgf_denoise(atn_ctx* ctx, const atn_destination* dst, int32_t compute_motion,
                     atn_vec4* out_host, atn_vec4* stages_host);
int atn_svgf_upload(atn_ctx* ctx, int32_t which, int32_t width, int32_t height, const atn_vec4* host);
void* atn_svgf_output_device(atn_ctx* ctx);

/* ---- stage entry points (parity tests; each mirrors one reference function) ---------------- */
/* GeneratePath for every pixel (src/libaten/renderer/pathtracing/pathtracing_impl.h:65-110). */
int atn_generate_paths(atn_ctx* ctx, int32_t width, int32_t height, int32_t sample, uint32_t frame, atn_ray* out_host);
/* ThreadedBvhTraverser::Traverse<Closest> (src/libaten/accelerator/threaded_bvh_traverser.h:98-304).
 * stats_out (optional): {node visits, triangle tests}. */
int atn_trace_closest(atn_ctx* ctx, const atn_ray* rays_host, uint32_t n, float t_min, float t_max,
                      atn_intersection* out_host, uint64_t* stats_out);
/* n successive CMJ::nextSample() (src/libaten/sampler/cmj.h:32-37). */
int atn_cmj_samples(atn_ctx* ctx, uint32_t index, uint32_t dimension, uint32_t scramble, int32_t n, float* out_host);
/* For each of n (index, dimension, scramble) triples: CMJ::init then `draws` x nextSample()
 * (src/libaten/sampler/cmj.h:21-37); out_host[k * draws + d]. */
int atn_cmj_batch(atn_ctx* ctx, uint32_t n, const uint32_t* index, const uint32_t* dimension, const uint32_t* scramble,
                  int32_t draws, float* out_host);
/* ray::Offset (src/libaten/math/ray.h:26-74: "A Fast and Robust Method for Avoiding Self-Intersection", Ray Tracing Gems ch. 6)
 * as the kernels compute it for every next ray and shadow ray: n origins (xyz) and normals (xyz) -> n offset origins. */
/* The math-library functions the float path calls, evaluated by THIS build on the device (ocml sinf / cosf / atanf / acosf / atan2f /
 * logf / expf / powf, and the correctly rounded sqrtf, a / b, 1 / sqrtf): kind 0..10 in that order, b is the second argument where
 * there is one.  What the float tolerance of the parity contract is made of: tools/ulp_study.py holds these against the CPU build's
 * libm. */
int atn_libm_probe(atn_ctx* ctx, int32_t kind, uint32_t n, const float* a, const float* b, float* out_host);
int atn_ray_offset(atn_ctx* ctx, uint32_t n, const float* origins, const float* normals, float* out_host);
/* material::sampleMaterial / samplePDF / sampleBSDF tables (src/libaten/material/material_impl.h:24-206).
 * Case i samples with CMJ::init(index[i], dimension[i], scramble[i]) (dimension == NULL: 0 -- note that CMJ's pattern seed is
 * dimension * scramble, cmj.h:118-123, so the draw of dimension 0 ignores the scramble).
 * out_sample: n*7 {dir, bsdf, pdf}; out_eval: n*5 {samplePDF, sampleBSDF.bsdf, sampleBSDF.pdf} at wo = dir. */
int atn_material_table(atn_ctx* ctx, int32_t mtrl_id, uint32_t n, const float* nrm, const float* wi,
                       const uint32_t* index, const uint32_t* dimension, const uint32_t* scramble, const float* uv,
                       float* out_sample, float* out_eval);
/* material::samplePDF / sampleBSDF (material_impl.h:90-206) at caller-given outgoing directions wo (n * 3):
 * out_eval n*5 {samplePDF, sampleBSDF.bsdf.xyz, sampleBSDF.pdf}.  For invariants that need no oracle (the pdf integrates to
 * one, sampled directions follow it, bsdf * cos stays below one). */
int atn_material_eval(atn_ctx* ctx, int32_t mtrl_id, uint32_t n, const float* nrm, const float* wi, const float* wo,
                      const float* uv, float* out_eval);
/* Stable compaction of indices with flag > 0 (contract of idaten::StreamCompaction::compact,
 * src/libidaten/kernel/StreamCompaction.cu:175-316): the renderer's own queue append (one ballot + popcount
 * prefix per wave, one atomic per 1024-entry block chunk -- the call k_shade makes) followed by a host sort,
 * because the renderer's queues are unordered. */
int atn_compact(atn_ctx* ctx, const int32_t* flags_host, uint32_t n, int32_t* out_idx_host, uint32_t* out_count);
/* The queue append as the renderer uses it: two queues filled in one pass (entry i -> A when flags_a[i] > 0, -> B
 * when flags_b[i] > 0; flags_b may be NULL), outputs in the (racy) order the device produced.  grid_blocks = 0
 * picks the renderer's launch geometry; any other value forces that many 256-thread blocks (grid-stride path). */
int atn_compact2(atn_ctx* ctx, const int32_t* flags_a_host, const int32_t* flags_b_host, uint32_t n, uint32_t grid_blocks,
                 int32_t* out_a_host, uint32_t* out_count_a, int32_t* out_b_host, uint32_t* out_count_b);

/* ABI self-description for binding checks. */
uint32_t atn_sizeof_scene_desc(void);
uint32_t atn_sizeof_destination(void);
uint32_t atn_abi_version(void);         /* 3 (r04: atn_material_table's dimension argument, atn_compact3 dropped); 2 since atn_scene_desc carries the NPR fields */
/* "<sha16 of the kernel sources>|<extra compile flags>" of the loaded binary (the build recipe passes it in; "unknown" for a
 * hand-rolled build): profiling records name the build they were taken on. */
const char* atn_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* ATEN_AMD_H_ */
