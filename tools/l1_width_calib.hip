// What does the per-CU vector L1 (TCP) charge for a wave load as a function of its WIDTH (1 / 2 / 3 / 4 dwords per lane), its
// instruction kind (global_load vs raw buffer_load) and the lanes that take part?  tools/valu_calib.hip calibrated the 16-byte
// gather only ("~16 clocks per wave load + ~0.3 per distinct 64-byte chunk, masked lanes are not free"); the
// walk's inner record is 24 B of box + links, so whether a 12-byte or 8-byte second load is cheaper than a 16-byte one decides
// whether a narrower record could pay.
//
// Every variant: independent gathers (8 in flight per lane) at random records of a 16 KB table (L1-resident), 8 waves per SIMD on
// every CU; reported: clocks of one CU's TCP per wave load at 2.4 GHz.
//   two-load variants ("pair"): the two halves of ONE record per lane, as the walk's inner step issues them (same 64-byte chunk):
//   16+16 (today's 32-byte record), 16+12 (28 B), 16+8 (24 B).
//
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -o /tmp/l1w tools/l1_width_calib.hip && /tmp/l1w
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int kCUs = 256;

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v3i __attribute__((ext_vector_type(3)));
typedef int v2i __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bool lane_on(int active)
{
    const uint32_t ln = threadIdx.x & 63u;
    if (active >= 10) {
        // lane-packing study (r06): the same NUMBER of active lanes packed into the low lanes (whole quads where the count allows)
        // against scattered over the wave.  10 + count: packed (lanes 0 .. count-1); 100 + count: scattered (a fixed pseudo-random subset)
        const bool scattered = active >= 100;
        const uint32_t count = (uint32_t)(scattered ? active - 100 : active - 10);
        if (!scattered) return ln < count;
        // rank of this lane in a fixed permutation of 0..63 (odd multiplier mod 64 is a bijection)
        return ((ln * 37u + 11u) & 63u) < count;
    }
    return active == 0 || (active == 1 && (ln & 3u) == 0u) || (active == 2 && ln < 16u) || (active == 3 && ((ln * 2654435761u) >> 16) % 64u < 27u);
}

// independent gathers: 8 loads in flight per lane (plain C++ loads of the right width: the compiler keeps them apart and waits once)
template <int W, int W2, bool BUF>
__global__ void __launch_bounds__(256) k_width(const char* __restrict__ table, uint32_t table_bytes, uint32_t mask, int* out, int iters, int active)
{
    const bool on = lane_on(active);
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, (int)table_bytes, 0x00020000);
    int acc = 0;
    for (int i = 0; i < iters; i++) {
        int v[8], u[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            x = x * 1664525u + 1013904223u;
            const uint32_t off = ((x >> 8) & mask) * 32u;
            v[k] = 0; u[k] = 0;
            if (on) {
                if constexpr (BUF) {
                    if constexpr (W == 4) { v4i t = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0); v[k] = t.x ^ t.y ^ t.z ^ t.w; }
                    else if constexpr (W == 3) { v3i t = __builtin_amdgcn_raw_buffer_load_b96(rs, (int)off, 0, 0); v[k] = t.x ^ t.y ^ t.z; }
                    else if constexpr (W == 2) { v2i t = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0); v[k] = t.x ^ t.y; }
                    else v[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0);
                    if constexpr (W2 == 4) { v4i t = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off + 16, 0, 0); u[k] = t.x ^ t.y ^ t.z ^ t.w; }
                    else if constexpr (W2 == 3) { v3i t = __builtin_amdgcn_raw_buffer_load_b96(rs, (int)off + 16, 0, 0); u[k] = t.x ^ t.y ^ t.z; }
                    else if constexpr (W2 == 2) { v2i t = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off + 16, 0, 0); u[k] = t.x ^ t.y; }
                    else if constexpr (W2 == 1) u[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off + 16, 0, 0);
                }
                else {
                    const char* p = table + off;
                    if constexpr (W == 4) { v4i t = *reinterpret_cast<const v4i*>(p); v[k] = t.x ^ t.y ^ t.z ^ t.w; }
                    else if constexpr (W == 3) { v3i t = *reinterpret_cast<const v3i*>(p); v[k] = t.x ^ t.y ^ t.z; }
                    else if constexpr (W == 2) { v2i t = *reinterpret_cast<const v2i*>(p); v[k] = t.x ^ t.y; }
                    else v[k] = *reinterpret_cast<const int*>(p);
                    if constexpr (W2 == 4) { v4i t = *reinterpret_cast<const v4i*>(p + 16); u[k] = t.x ^ t.y ^ t.z ^ t.w; }
                    else if constexpr (W2 == 3) { v3i t = *reinterpret_cast<const v3i*>(p + 16); u[k] = t.x ^ t.y ^ t.z; }
                    else if constexpr (W2 == 2) { v2i t = *reinterpret_cast<const v2i*>(p + 16); u[k] = t.x ^ t.y; }
                    else if constexpr (W2 == 1) u[k] = *reinterpret_cast<const int*>(p + 16);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) acc += v[k] + u[k];
    }
    if (acc == 0x12345678) out[0] = acc;
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
    f();
    CK(hipDeviceSynchronize());
    for (int r = 0; r < reps; r++) { t.start(); f(); float ms = t.stop(); if (ms < best) best = ms; }
    return best;
}

template <int W, int W2, bool BUF>
static void run(const char* table, uint32_t bytes, int* out)
{
    const int full = kCUs * 8, iters = 2000;
    static const char* act_name[] = { "64 lanes", "16 lanes, one per quad", "lanes 0-15", "27 scattered lanes" };
    for (int active : { 0, 3, 1 }) {
        float ms = best_of([&] { hipLaunchKernelGGL((k_width<W, W2, BUF>), dim3(full), dim3(256), 0, 0, table, bytes, bytes / 32 - 1, out, iters, active); });
        const double wave_insts_per_cu = (double)full * 4 * iters * 8 * (W2 ? 2 : 1) / kCUs;
        printf("{\"kind\": \"%s\", \"dwords\": \"%d%s\", \"active\": \"%s\", \"ms\": %.4f, \"clocks_per_wave_load\": %.2f, \"clocks_per_record\": %.2f}\n",
               BUF ? "buffer" : "global", W, W2 == 0 ? "" : W2 == 4 ? "+4" : W2 == 3 ? "+3" : W2 == 2 ? "+2" : "+1", act_name[active], ms,
               ms * 1e-3 * 2.4e9 / wave_insts_per_cu, ms * 1e-3 * 2.4e9 / (wave_insts_per_cu / (W2 ? 2 : 1)));
    }
}

// r06: does it matter WHERE in the wave the live lanes sit?  (if packed lanes cost the TCP >= 30 % less, keeping live
// rays in the low lanes at refill time could pay; if not, the refill walk's lane occupancy is closed.)
template <int W, int W2, bool BUF>
static void run_packing(const char* table, uint32_t bytes, int* out)
{
    const int full = kCUs * 8, iters = 2000;
    for (int count : { 64, 48, 32, 28, 27, 16, 8 }) {
        for (int scattered = 0; scattered < 2; scattered++) {
            if (count == 64 && scattered) continue;
            const int active = (scattered ? 100 : 10) + count;
            float ms = best_of([&] { hipLaunchKernelGGL((k_width<W, W2, BUF>), dim3(full), dim3(256), 0, 0, table, bytes, bytes / 32 - 1, out, iters, active); });
            const double wave_insts_per_cu = (double)full * 4 * iters * 8 * (W2 ? 2 : 1) / kCUs;
            printf("{\"study\": \"lane packing\", \"kind\": \"%s\", \"dwords\": \"%d%s\", \"active_lanes\": %d, \"placement\": \"%s\", \"ms\": %.4f, \"clocks_per_wave_load\": %.2f, \"clocks_per_record\": %.2f}\n",
                   BUF ? "buffer" : "global", W, W2 == 0 ? "" : "+4", count, scattered ? "scattered over the wave" : "packed into the low lanes", ms,
                   ms * 1e-3 * 2.4e9 / wave_insts_per_cu, ms * 1e-3 * 2.4e9 / (wave_insts_per_cu / (W2 ? 2 : 1)));
        }
    }
}

int main(int argc, char** argv)
{
    int* out; CK(hipMalloc(&out, 64));
    const uint32_t bytes = 16384;
    std::vector<int> h(bytes / 4);
    for (size_t i = 0; i < h.size(); i++) h[i] = (int)(i * 2654435761u);
    char* table; CK(hipMalloc(&table, bytes));
    CK(hipMemcpy(table, h.data(), bytes, hipMemcpyHostToDevice));
    if (argc > 1 && !strcmp(argv[1], "--packing")) {
        run_packing<4, 0, true>(table, bytes, out);     // one 16-byte buffer load per lane
        run_packing<4, 4, true>(table, bytes, out);     // the walk's inner step: both halves of one 32-byte record
        CK(hipFree(table)); CK(hipFree(out));
        return 0;
    }
    run<1, 0, false>(table, bytes, out); run<2, 0, false>(table, bytes, out); run<3, 0, false>(table, bytes, out); run<4, 0, false>(table, bytes, out);
    run<1, 0, true>(table, bytes, out); run<2, 0, true>(table, bytes, out); run<3, 0, true>(table, bytes, out); run<4, 0, true>(table, bytes, out);
    run<4, 4, false>(table, bytes, out); run<4, 3, false>(table, bytes, out); run<4, 2, false>(table, bytes, out); run<4, 1, false>(table, bytes, out);
    run<4, 4, true>(table, bytes, out); run<4, 3, true>(table, bytes, out); run<4, 2, true>(table, bytes, out);
    CK(hipFree(table)); CK(hipFree(out));
    return 0;
}
