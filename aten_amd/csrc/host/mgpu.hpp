// One-node multi-GPU renderer behind the C-ABI (atn_mgpu_*, include/aten_amd.h).
//
// The reference's GPU renderer is single-device (idaten::Renderer, src/libidaten/kernel/renderer.h:17-179, driven from
// one window thread: src/device_renderer/main.cpp:133-149,196-204).  A C++ aten application that wants every GPU of the
// node keeps that call shape -- UpdateSceneData once, render(dst) per frame -- and this object does the rest:
//
//   * one atn::PathTracing per shard, each with its own host WORKER THREAD (kernel launches of 8 devices are not
//     serialised through one thread; a frame is ~13 launches per device);
//   * the scene is replicated (it is << 288 GB), the screen is cut into 8x8 tiles, tile t -> shard t % N
//     (atn_set_screen_shard), seeds and pixel indices stay global, so the image does not depend on N;
//   * the only exchange of a frame: every shard PUSHES its tile buffer into the gather buffer on device 0 with a peer
//     copy over xGMI on its own stream (33 MB per 1080p frame in total, 7/8 of it crossing links, each over a different
//     link into device 0), then device 0 scatters the gathered tiles into the full frame (k_assemble_tiles).  Two
//     gather buffers + events both ways: frame f + 1 renders while frame f is being assembled, no host
//     synchronisation inside a frame unless the caller asks for the film in host memory.
//
// No collective library is needed for a gather-to-one of this size; bench.py's one-process-per-GPU mode uses RCCL's
// all_gather for the same exchange (every rank gets the frame).
//
// A device ordinal may appear several times in the shard list: the shards then share that GPU.  That is how the whole
// N > 1 path (threads, shards, peer copies, events, assembly) is exercised on a one-GPU box.
#pragma once
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

namespace atn {

class Worker {
public:
    Worker() : th_([this] { loop(); }) {}
    ~Worker()
    {
        { std::lock_guard<std::mutex> g(m_); quit_ = true; }
        cv_.notify_all();
        th_.join();
    }
    void post(std::function<int()> f)
    {
        { std::lock_guard<std::mutex> g(m_); job_ = std::move(f); has_job_ = true; done_ = false; }
        cv_.notify_all();
    }
    int wait()
    {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [this] { return done_; });
        return rc_;
    }

private:
    void loop()
    {
        for (;;) {
            std::function<int()> f;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [this] { return has_job_ || quit_; });
                if (quit_ && !has_job_) return;
                f = std::move(job_);
                has_job_ = false;
            }
            int rc;
            try { rc = f(); }
            catch (const std::bad_alloc&) { rc = ATN_ERR_OUT_OF_MEMORY; }
            catch (...) { rc = ATN_ERR_INVALID_ARG; }
            { std::lock_guard<std::mutex> g(m_); rc_ = rc; done_ = true; }
            cv_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::function<int()> job_;
    bool has_job_ = false, done_ = true, quit_ = false;
    int rc_ = 0;
    std::thread th_;        // last member: the thread starts after everything it touches exists
};

class MultiGpu {
public:
    std::string last_error;
    std::vector<std::unique_ptr<PathTracing>> shard;
    std::vector<std::unique_ptr<Worker>> worker;
    int n = 0;

    // exchange state, all on shard 0's device
    hipStream_t comm = nullptr;
    DevBuf<float4> gathered[2], full;
    hipEvent_t ev_pushed[2][PathTracing::kMaxShards] = {};  // shard i's tiles of the frame using buffer k have arrived
    hipEvent_t ev_assembled[2] = {};                         // buffer k has been scattered into `full`
    bool assembled_recorded[2] = { false, false };
    uint64_t frames = 0;
    int32_t width = 0, height = 0;

    int fail(int code, const std::string& msg) { last_error = msg; return code; }

    int init(const int32_t* devices, int32_t count)
    {
        int visible = 0;
        if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0)
            return fail(ATN_ERR_NO_DEVICE, "no HIP device available (libaten_amd has no CPU fallback)");
        std::vector<int32_t> list;
        if (devices && count > 0) list.assign(devices, devices + count);
        else {
            const int32_t want = count > 0 ? count : visible;
            if (want > visible) return fail(ATN_ERR_INVALID_ARG, "more devices requested than visible");
            for (int32_t i = 0; i < want; i++) list.push_back(i);
        }
        if (list.empty() || list.size() > (size_t)PathTracing::kMaxShards) return fail(ATN_ERR_INVALID_ARG, "shard count out of range");
        n = (int)list.size();
        for (int i = 0; i < n; i++) {
            shard.emplace_back(new PathTracing());
            int rc = shard[i]->init(list[i]);
            if (rc != ATN_OK) return fail(rc, shard[i]->last_error);
            shard[i]->rank = i; shard[i]->world = n;
            worker.emplace_back(new Worker());
        }
        const int d0 = shard[0]->device;
        for (int i = 1; i < n; i++) {
            const int di = shard[i]->device;
            if (di == d0) continue;
            int can = 0;
            (void)hipDeviceCanAccessPeer(&can, di, d0);
            if (can) {
                (void)hipSetDevice(di);
                hipError_t e = hipDeviceEnablePeerAccess(d0, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(ATN_ERR_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
                (void)hipGetLastError();
            }       // otherwise hipMemcpyPeerAsync stages through host memory: slower, still correct
        }
        ATN_HIP(hipSetDevice(d0));
        ATN_HIP(hipStreamCreateWithFlags(&comm, hipStreamNonBlocking));
        for (int k = 0; k < 2; k++) {
            ATN_HIP(hipEventCreateWithFlags(&ev_assembled[k], hipEventDisableTiming));
            for (int i = 0; i < n; i++) {
                ATN_HIP(hipSetDevice(shard[i]->device));
                ATN_HIP(hipEventCreateWithFlags(&ev_pushed[k][i], hipEventDisableTiming));
            }
        }
        return ATN_OK;
    }

    ~MultiGpu()
    {
        worker.clear();     // joins the threads
        for (int k = 0; k < 2; k++) {
            if (ev_assembled[k]) (void)hipEventDestroy(ev_assembled[k]);
            for (int i = 0; i < n; i++) if (ev_pushed[k][i]) (void)hipEventDestroy(ev_pushed[k][i]);
        }
        if (comm) (void)hipStreamDestroy(comm);
        if (!shard.empty()) (void)hipSetDevice(shard[0]->device);
        gathered[0].release(); gathered[1].release(); full.release();
    }

    // run f(i) on every shard's worker thread, wait for all; first failure wins
    int on_all(const std::function<int(int)>& f)
    {
        for (int i = 0; i < n; i++) worker[i]->post([&f, i] { return f(i); });
        int rc = ATN_OK, who = -1;
        for (int i = 0; i < n; i++) {
            const int r = worker[i]->wait();
            if (r != ATN_OK && rc == ATN_OK) { rc = r; who = i; }
        }
        if (rc != ATN_OK) last_error = "shard " + std::to_string(who) + ": " + shard[who]->last_error;
        return rc;
    }

    // ≙ idaten::PathTracing::render for the whole node.  Returns once everything is ENQUEUED (or, with out_host,
    // once the frame is in host memory).
    int render(const atn_destination* d, atn_vec4* out_host) { return render_burst(d, 1, out_host); }

    // n_frames consecutive frames on every shard (atn_render_burst: one regenerated pool per shard when regeneration is on), then ONE
    // exchange: the progressive film is only looked at after the burst, so its tiles travel once per burst, not once per frame.
    int render_burst(const atn_destination* d, int32_t n_frames, atn_vec4* out_host)
    {
        if (!d || d->width <= 0 || d->height <= 0) return fail(ATN_ERR_INVALID_ARG, "bad destination");
        if (n_frames <= 0) return fail(ATN_ERR_INVALID_ARG, "bad burst length");
        const int k = (int)(frames & 1u);
        const int d0 = shard[0]->device;
        const size_t px = (size_t)d->width * d->height;
        ATN_HIP(hipSetDevice(d0));
        if (d->width != width || d->height != height) {
            ATN_HIP(hipStreamSynchronize(comm));
            ATN_HIP(full.resize(px));
            ATN_HIP(hipMemsetAsync(full.p, 0, px * sizeof(float4), comm));
            width = d->width; height = d->height;
        }
        float4* gbuf[2] = { nullptr, nullptr };
        {
            const uint32_t tiles = (uint32_t)((d->width + 7) / 8) * (uint32_t)((d->height + 7) / 8);
            const size_t slots = (size_t)((tiles + n - 1) / n) * 64;
            for (int b = 0; b < 2; b++) { ATN_HIP(gathered[b].resize(slots * n)); gbuf[b] = gathered[b].p; }
        }
        const bool wait_assembled = assembled_recorded[k];
        int rc = on_all([&](int i) -> int {
            PathTracing& r = *shard[i];
            int rr = n_frames == 1 ? r.render(d, nullptr) : r.render_burst(d, n_frames, nullptr);
            if (rr != ATN_OK) return rr;
            // push this shard's tiles into slot i of the gather buffer on device 0 (peer copy on the shard's stream)
            if (wait_assembled) { if (hipStreamWaitEvent(r.stream, ev_assembled[k], 0) != hipSuccess) return r.fail(ATN_ERR_HIP, "hipStreamWaitEvent"); }
            const size_t bytes = (size_t)r.n_slots * sizeof(float4);
            hipError_t e = (r.device == d0)
                ? hipMemcpyAsync(gbuf[k] + (size_t)i * r.n_slots, r.tile_out.p, bytes, hipMemcpyDeviceToDevice, r.stream)
                : hipMemcpyPeerAsync(gbuf[k] + (size_t)i * r.n_slots, d0, r.tile_out.p, r.device, bytes, r.stream);
            if (e == hipSuccess) e = hipEventRecord(ev_pushed[k][i], r.stream);
            if (e != hipSuccess) return r.fail(ATN_ERR_HIP, std::string("tile push: ") + hipGetErrorString(e));
            return ATN_OK;
        });
        if (rc != ATN_OK) return rc;
        ATN_HIP(hipSetDevice(d0));
        for (int i = 0; i < n; i++) ATN_HIP(hipStreamWaitEvent(comm, ev_pushed[k][i], 0));
        const uint32_t total = (uint32_t)n * shard[0]->n_slots;
        hipLaunchKernelGGL(k_assemble_tiles, dim3((total + 255) / 256), dim3(256), 0, comm, (const float4*)gbuf[k], full.p,
                           d->width, d->height, (d->width + 7) / 8, (d->height + 7) / 8, n, (int32_t)shard[0]->n_slots);
        ATN_HIP(hipGetLastError());
        ATN_HIP(hipEventRecord(ev_assembled[k], comm));
        assembled_recorded[k] = true;
        frames++;
        if (out_host) {
            ATN_HIP(hipMemcpyAsync(out_host, full.p, px * sizeof(float4), hipMemcpyDeviceToHost, comm));
            ATN_HIP(hipStreamSynchronize(comm));
        }
        return ATN_OK;
    }

    int synchronize()
    {
        int rc = on_all([&](int i) -> int {
            PathTracing& r = *shard[i];
            if (hipSetDevice(r.device) != hipSuccess) return r.fail(ATN_ERR_HIP, "hipSetDevice");
            return r.quiesce();
        });
        if (rc != ATN_OK) return rc;
        ATN_HIP(hipSetDevice(shard[0]->device));
        ATN_HIP(hipStreamSynchronize(comm));
        return ATN_OK;
    }
};

} // namespace atn
