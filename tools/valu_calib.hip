// Calibration micro-benchmarks for the counters bench.py's roofline object is built from.
//
// What does `SQ_ACTIVE_INST_VALU / (256 CUs x cycles)` read when a kernel SATURATES VALU issue on gfx950, and what does
// `TCP_TOTAL_ACCESSES / cycle / CU` read when the per-CU vector L1 (TCP) is saturated by the walk's access shape
// (16 B per lane, gathered)?  rocprofiler's derived metrics fall back to gfx94x formulas on this chip
// (MI355X_MICROARCH.md, "rocprofv3 PMC slots"), so the ceilings are MEASURED here rather than assumed:
//
//   k_cal_valu_fma      16 independent v_fma_f32 chains per lane, 8 waves per SIMD            -> VALU issue ceiling (plain)
//   k_cal_valu_walkmix  the instruction mix of the walk's inner step (v_pk_mul / v_pk_add / v_min3 / v_max3 / v_cmp /
//                       v_cndmask), independent chains, 8 waves per SIMD                      -> VALU issue ceiling (this mix)
//   k_cal_l1_gather     independent 16-B gathers at random records of a 16 KB table (L1-resident) -> TCP access-rate ceiling
//   k_cal_l1_rows       the same with 8 lanes per 128-B line (what a coherent wave does)
//   k_cal_l1_chase<W>   DEPENDENT chain of 2 x 16-B gathers per hop (a walk step: both halves of a 32-B record, the next
//                       address comes out of the data), table 16 KB (L1), 1 MB (L2) -- at W waves per SIMD
//                                                                                              -> latency-bound step rate
//
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_calib tools/valu_calib.hip && /tmp/valu_calib
// The program prints one JSON line per kernel with its wall time (HIP events) and the work it did; tools/calib_collect.sh
// runs it under `rocprofv3 --pmc` (one pass per counter set) and tools/calib_to_json.py joins both into
// profiles/r03_calibration.json.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static constexpr int kCUs = 256;

// ---- VALU: plain FMA ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_cal_valu_fma(float* out, int iters, float b, float c)
{
    float a[16];
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = (float)(threadIdx.x + k);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) s += a[k];
    if (s == 12345.678f) out[0] = s;        // never true; keeps the chains alive
}

// the same with the multiplier / addend in an SGPR (one VGPR source: no register-bank pressure)
__global__ void __launch_bounds__(256) k_cal_valu_fma_sgpr(float* out, int iters, float b)
{
    float a[16];
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = (float)(threadIdx.x + k);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[k]) : "s"(b));
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) s += a[k];
    if (s == 12345.678f) out[0] = s;
}

// ---- VALU: the walk's inner-step mix (traverse.hpp, burst loop: 6 slab products as 3 v_pk_mul + 3 v_pk_add, min3/max3
// x 4, 2 compares-and-selects) ---------------------------------------------------------------------------------------
typedef float float2v __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k_cal_valu_walkmix(float* out, int iters, float b, float c)
{
    float2v p[4], q[4];
    float t[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { p[k] = float2v{ (float)threadIdx.x + k, 1.0f + k }; q[k] = float2v{ 0.5f + k, 2.0f }; t[k] = (float)k; }
    const float2v bb = { b, b }, cc = { c, c };
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(bb));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(q[k]) : "v"(cc));
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(q[k]) : "v"(bb));
            asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(t[k]) : "v"(p[k].x), "v"(q[k].x));
            asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(t[k]) : "v"(p[k].y), "v"(q[k].y));
            asm volatile("v_cmp_le_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(t[k]) : "v"(p[k].x), "v"(q[k].y) : "vcc");
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) s += t[k] + p[k].x + q[k].y;
    if (s == 12345.678f) out[0] = s;
}

// ---- L1: independent 16-B gathers -----------------------------------------------------------------------------------
// `rows`: 0 = every lane its own random record; 1 = groups of 8 lanes read the 8 records of one 128-B line;
// 2 = PAIRS of neighbouring lanes read the two 16-B halves of one random 32-B record (a 32-B node record fetched by two lanes)
// `active`: 0 = all 64 lanes load; 1 = one lane per quad (lane % 4 == 0); 2 = the first 16 lanes of the wave; 3 = 27 lanes
// picked pseudo-randomly (the trace kernel's mean).  The other lanes are exec-masked for the loads: does the TCP spend
// time on them?
__global__ void __launch_bounds__(256) k_cal_l1_gather(const float4* __restrict__ table, uint32_t mask, float* out, int iters, int rows, int active)
{
    const uint32_t ln = threadIdx.x & 63u;
    const bool on = active == 0 || (active == 1 && (ln & 3u) == 0u) || (active == 2 && ln < 16u) || (active == 3 && ((ln * 2654435761u) >> 16) % 64u < 27u);
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    if (rows == 1) x = ((blockIdx.x * 256u + threadIdx.x) >> 3) * 2654435761u + 12345u;
    if (rows == 2) x = ((blockIdx.x * 256u + threadIdx.x) >> 1) * 2654435761u + 12345u;
    const uint32_t lane8 = threadIdx.x & 7u;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int i = 0; i < iters; i++) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            x = x * 1664525u + 1013904223u;
            uint32_t idx = (x >> 8) & mask;
            if (rows == 1) idx = (idx & ~7u) | lane8;
            if (rows == 2) idx = (idx & ~1u) | (lane8 & 1u);
            v[k] = on ? table[idx] : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

// ---- HBM side: independent 16-byte point lookups in a table far larger than the caches (the shape of a texel / environment
// map lookup of an incoherent path): what do FETCH_SIZE and TCC_EA0_RDREQ read PER LOOKUP?  (the guide's "x2" correction of
// FETCH_SIZE is calibrated on streaming reads only)
__global__ void __launch_bounds__(256) k_cal_hbm_gather(const float4* __restrict__ table, uint32_t mask, float* out, int iters)
{
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 99991u;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int i = 0; i < iters; i++) {
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            x = x * 1664525u + 1013904223u;
            v[k] = table[(x >> 4) & mask];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

// streaming twin: every lane reads consecutive 16-byte records (1 KiB per wave load, whole 128-byte lines)
__global__ void __launch_bounds__(256) k_cal_hbm_stream(const float4* __restrict__ table, uint32_t n_rec, float* out)
{
    float4 acc = make_float4(0, 0, 0, 0);
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_rec; i += gridDim.x * 256u) {
        const float4 v = table[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

// ---- L1 / L2: dependent chain, one hop = the two 16-B halves of a 32-B record, next index from the data --------------
template <int WAVES_PER_SIMD>
__global__ void __launch_bounds__(256) k_cal_l1_chase(const float4* __restrict__ table, uint32_t mask, float* out, int hops)
{
    uint32_t idx = ((blockIdx.x * 256u + threadIdx.x) * 2654435761u >> 7) & mask;
    float acc = 0;
    for (int i = 0; i < hops; i++) {
        const float4 a = table[2 * idx];
        const float4 b = table[2 * idx + 1];
        acc += (a.x + a.y) + (a.z + b.x) + (b.y + b.z);      // every component used: the loads stay global_load_dwordx4
        idx = (__float_as_uint(a.w) + __float_as_uint(b.w)) & mask;      // b.w holds 0: the hop waits for BOTH halves
    }
    if (acc == 12345.678f) out[0] = acc;
}

struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    void start() { CK(hipEventRecord(a, 0)); }
    float stop() { CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }
};

template <class F>
static float best_of(F f, int reps = 5)
{
    Timer t; float best = 1e30f;
    f();            // warm-up (also brings the table into the caches)
    CK(hipDeviceSynchronize());
    for (int r = 0; r < reps; r++) { t.start(); f(); float ms = t.stop(); if (ms < best) best = ms; }
    return best;
}

int main()
{
    float* out; CK(hipMalloc(&out, 64));
    // tables: records hold the NEXT index in .w (a random permutation cycle), so the chase is a real dependent chain
    auto make_table = [](uint32_t n_rec, bool pairs) {
        std::vector<float4> h((size_t)n_rec * (pairs ? 2 : 1));
        std::vector<uint32_t> perm(n_rec);
        for (uint32_t i = 0; i < n_rec; i++) perm[i] = i;
        uint64_t s = 88172645463325252ull;
        for (uint32_t i = n_rec - 1; i > 0; i--) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; uint32_t j = (uint32_t)(s % (i + 1)); std::swap(perm[i], perm[j]); }
        for (uint32_t i = 0; i < n_rec; i++) {
            uint32_t nxt = perm[(i + 1) % n_rec];      // one random cycle through all records
            float w; memcpy(&w, &nxt, 4);
            if (pairs) { h[2 * (size_t)perm[i]] = make_float4(1, 2, 3, w); h[2 * (size_t)perm[i] + 1] = make_float4(4, 5, 6, 0); }
            else h[perm[i]] = make_float4(1, 2, 3, w);
        }
        float4* d; CK(hipMalloc(&d, h.size() * sizeof(float4)));
        CK(hipMemcpy(d, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice));
        return d;
    };
    const int full = kCUs * 8;       // 256-thread blocks for 8 waves per SIMD on every CU
    {
        const int iters = 20000;
        float ms = best_of([&] { hipLaunchKernelGGL(k_cal_valu_fma, dim3(full), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); });
        double winst = (double)full * 4 * iters * 16;
        printf("{\"kernel\": \"k_cal_valu_fma\", \"ms\": %.4f, \"wave_valu_insts\": %.0f, \"waves_per_simd\": 8, \"valu_per_simd_per_us\": %.2f}\n",
               ms, winst, winst / 1024.0 / (ms * 1e3));
    }
    {
        const int iters = 20000;
        float ms = best_of([&] { hipLaunchKernelGGL(k_cal_valu_fma_sgpr, dim3(full), dim3(256), 0, 0, out, iters, 1.0001f); });
        double winst = (double)full * 4 * iters * 16;
        printf("{\"kernel\": \"k_cal_valu_fma_sgpr\", \"ms\": %.4f, \"wave_valu_insts\": %.0f, \"waves_per_simd\": 8, \"valu_per_simd_per_us\": %.2f}\n",
               ms, winst, winst / 1024.0 / (ms * 1e3));
    }
    {
        const int iters = 12000;
        float ms = best_of([&] { hipLaunchKernelGGL(k_cal_valu_walkmix, dim3(full), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); });
        double winst = (double)full * 4 * iters * 4 * 7;
        printf("{\"kernel\": \"k_cal_valu_walkmix\", \"ms\": %.4f, \"wave_valu_insts\": %.0f, \"waves_per_simd\": 8, \"valu_per_simd_per_us\": %.2f}\n",
               ms, winst, winst / 1024.0 / (ms * 1e3));
    }
    {
        float4* t16k = make_table(1024, false);
        for (int v = 0; v < 6; v++) {
            const int rows = v < 3 ? v : 0, active = v < 3 ? 0 : v - 2;
            const int iters = 2000;
            float ms = best_of([&] { hipLaunchKernelGGL(k_cal_l1_gather, dim3(full), dim3(256), 0, 0, (const float4*)t16k, 1023u, out, iters, rows, active); });
            int n_on = 0;
            for (uint32_t ln = 0; ln < 64; ln++) n_on += active == 0 || (active == 1 && (ln & 3u) == 0u) || (active == 2 && ln < 16u) || (active == 3 && ((ln * 2654435761u) >> 16) % 64u < 27u);
            double lane_loads = (double)full * 4 * n_on * iters * 8;
            printf("{\"kernel\": \"k_cal_l1_gather\", \"active_lanes\": %d, \"cycles_per_wave_load_at_2p4GHz\": %.2f, \"variant\": \"%s\", \"ms\": %.4f, \"lane_loads_16B\": %.0f, \"wave_loads\": %.0f, \"GBps\": %.1f, \"lane_loads_per_cu_per_us\": %.1f}\n",
                   n_on, ms * 1e-3 * 2.4e9 / ((double)full * 4 * iters * 8 / 256.0),
                   active == 1 ? "random, one lane per quad active" : active == 2 ? "random, lanes 0-15 active" : active == 3 ? "random, 27 scattered lanes active" :
                   rows == 1 ? "rows (8 lanes per 128-B line)" : rows == 2 ? "pairs (2 lanes per 32-B record)" : "random (one record per lane)", ms, lane_loads, (double)full * 4 * iters * 8, lane_loads * 16 / (ms * 1e6), lane_loads / kCUs / (ms * 1e3));
        }
        CK(hipFree(t16k));
    }
    {
        struct { const char* name; uint32_t n_rec; } tabs[] = { { "16KB (L1)", 512 }, { "1MB (L2)", 32768 }, { "64MB (MALL/HBM)", 2097152 } };
        for (auto& tb : tabs) {
            float4* t = make_table(tb.n_rec, true);
            const int hops = tb.n_rec > 100000 ? 400 : 4000;
            auto run = [&](int w) {
                const int blocks = kCUs * w;
                float ms;
                switch (w) {
                case 1: ms = best_of([&] { hipLaunchKernelGGL(k_cal_l1_chase<1>, dim3(blocks), dim3(256), 0, 0, (const float4*)t, tb.n_rec - 1, out, hops); }); break;
                case 5: ms = best_of([&] { hipLaunchKernelGGL(k_cal_l1_chase<5>, dim3(blocks), dim3(256), 0, 0, (const float4*)t, tb.n_rec - 1, out, hops); }); break;
                default: ms = best_of([&] { hipLaunchKernelGGL(k_cal_l1_chase<8>, dim3(blocks), dim3(256), 0, 0, (const float4*)t, tb.n_rec - 1, out, hops); }); break;
                }
                double lane_hops = (double)blocks * 256 * hops;
                printf("{\"kernel\": \"k_cal_l1_chase<%d>\", \"table\": \"%s\", \"ms\": %.4f, \"hops\": %d, \"waves_per_simd\": %d, \"ns_per_hop\": %.2f, \"lane_hops_per_us\": %.1f, \"lane_hops_per_cu_per_us\": %.2f}\n",
                       w, tb.name, ms, hops, w, ms * 1e6 / hops, lane_hops / (ms * 1e3), lane_hops / kCUs / (ms * 1e3));
            };
            run(1); run(5); run(8);
            CK(hipFree(t));
        }
    }
    {
        // 1 GiB of float4 (2^26 records): beyond L2 (32 MiB) and the Infinity Cache (256 MiB)
        const uint32_t n_rec = 1u << 26;
        float4* big; CK(hipMalloc(&big, (size_t)n_rec * sizeof(float4)));
        CK(hipMemset(big, 0, (size_t)n_rec * sizeof(float4)));
        const int iters = 64;
        float ms = best_of([&] { hipLaunchKernelGGL(k_cal_hbm_gather, dim3(full), dim3(256), 0, 0, (const float4*)big, n_rec - 1, out, iters); });
        double lookups = (double)full * 256 * iters * 4;
        printf("{\"kernel\": \"k_cal_hbm_gather\", \"table\": \"1 GiB\", \"ms\": %.4f, \"lookups_16B\": %.0f, \"lookups_per_us\": %.1f, \"useful_GBps\": %.1f}\n",
               ms, lookups, lookups / (ms * 1e3), lookups * 16 / (ms * 1e6));
        ms = best_of([&] { hipLaunchKernelGGL(k_cal_hbm_stream, dim3(full), dim3(256), 0, 0, (const float4*)big, n_rec, out); });
        printf("{\"kernel\": \"k_cal_hbm_stream\", \"table\": \"1 GiB\", \"ms\": %.4f, \"lookups_16B\": %.0f, \"bytes\": %.0f, \"useful_GBps\": %.1f}\n",
               ms, (double)n_rec, (double)n_rec * 16, (double)n_rec * 16 / (ms * 1e6));
        CK(hipFree(big));
    }
    CK(hipFree(out));
    return 0;
}
