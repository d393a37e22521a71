// Device fp32 vector math with a fixed evaluation order.
//
// Parity contract (DESIGN.md "float parity"): every expression below is evaluated as the listed
// sequence of individually rounded IEEE fp32 operations -- the TU is compiled with
// -ffp-contract=off and correctly rounded divide/sqrt -- so that results are bit-identical to an
// x86-64 SSE2 build of the same formulas (the reference's CPU build has no FMA).
// The formulas are those aten evaluates through glm (src/libaten/math/vec3.h:213):
//   dot = (x*x' + y*y') + z*z' ; cross ; normalize = v * (1 / sqrt(dot(v,v))).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ATN_DEV __device__ __forceinline__

namespace atn {

constexpr float kPi = 3.14159265358979323846F;
constexpr float kPi2 = kPi * 2;
constexpr float kInf = 3.402823466e+38F;    // AT_MATH_INF = numeric_limits<float>::max()
constexpr float kEps = 1e-9F;               // AT_MATH_EPSILON

struct f3 { float x, y, z; };

// A uniform value (a kernel argument) that the optimiser must treat as produced HERE.  What is derived from it -- (float)n,
// the reciprocal a division by n needs -- is then computed where it is used instead of being hoisted out of the kernel's
// main loop into a vector register that lives (at 5 waves per SIMD: spills) through everything else.  No instruction.
ATN_DEV int32_t here(int32_t x) { asm volatile("" : "+s"(x)); return x; }
ATN_DEV uint32_t here(uint32_t x) { asm volatile("" : "+s"(x)); return x; }
ATN_DEV uint32_t here_v(uint32_t x) { asm volatile("" : "+v"(x)); return x; }     // the same for a per-lane value (threadIdx.x)
// x / d and x % d for a uniform divisor with rcp = floor(2^32 / d) computed on the host (capped at 2^32 - 1 for d = 1): the
// estimate mulhi(x, rcp) is the quotient or one less for every 32-bit x, so one correction step makes it exact.
__host__ __device__ inline uint32_t udiv_rcp(uint32_t d) { return d <= 1u ? 0xffffffffu : (uint32_t)(0x100000000ull / d); }
ATN_DEV void udivmod(uint32_t x, uint32_t d, uint32_t rcp, uint32_t& q, uint32_t& r)
{
    q = __umulhi(x, rcp);
    r = x - q * d;
    if (r >= d) { q += 1u; r -= d; }
}
// Set bits of a wave mask below this lane (the rank of the lane among the flagged ones): v_mbcnt_lo / _hi on the ballot's two
// halves -- no 64-bit lane mask held in vector registers.
ATN_DEV uint32_t bits_below_lane(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
ATN_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
ATN_DEV f3 mk3(float s) { return mk3(s, s, s); }
ATN_DEV f3 mk3(const float4& v) { return mk3(v.x, v.y, v.z); }
ATN_DEV f3 operator+(const f3& a, const f3& b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
ATN_DEV f3 operator-(const f3& a, const f3& b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
ATN_DEV f3 operator*(const f3& a, const f3& b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
ATN_DEV f3 operator/(const f3& a, const f3& b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
ATN_DEV f3 operator*(const f3& a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
ATN_DEV f3 operator*(float s, const f3& a) { return mk3(s * a.x, s * a.y, s * a.z); }
ATN_DEV f3 operator/(const f3& a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
ATN_DEV f3 operator/(float s, const f3& a) { return mk3(s / a.x, s / a.y, s / a.z); }
ATN_DEV f3 operator+(const f3& a, float s) { return mk3(a.x + s, a.y + s, a.z + s); }
ATN_DEV f3 operator-(const f3& a) { return mk3(-a.x, -a.y, -a.z); }

// std::max / std::min of the host build (NaN-sensitive): max(a,b) = a<b ? b : a; min(a,b) = b<a ? b : a
ATN_DEV float smax(float a, float b) { return (a < b) ? b : a; }
ATN_DEV float smin(float a, float b) { return (b < a) ? b : a; }
ATN_DEV float sclamp(float v, float lo, float hi) { return (v < lo) ? lo : (hi < v) ? hi : v; }

ATN_DEV float dot(const f3& a, const f3& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
ATN_DEV f3 cross(const f3& x, const f3& y)
{
    return mk3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
ATN_DEV f3 normalize(const f3& v) { return v * (1.0F / sqrtf(dot(v, v))); }
ATN_DEV float length(const f3& v) { return sqrtf(dot(v, v)); }
ATN_DEV float max3(const f3& v) { return smax(smax(v.x, v.y), v.z); }
ATN_DEV float min3(const f3& v) { return smin(smin(v.x, v.y), v.z); }
ATN_DEV f3 mix3(const f3& a, const f3& b, float t) { return a * (1.0F - t) + b * t; }
ATN_DEV float mixf(float a, float b, float t) { return a * (1 - t) + b * t; }
ATN_DEV float sqr(float f) { return f * f; }
ATN_DEV float luminance(float r, float g, float b) { return (0.212639F * r + 0.71517F * g) + 0.0721926F * b; }

// aten::vec4 arithmetic used by hit evaluation (src/libaten/math/vec4.h)
ATN_DEV float4 mul4(float s, const float4& v) { return make_float4(s * v.x, s * v.y, s * v.z, s * v.w); }
ATN_DEV float4 add4(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
ATN_DEV float4 sub4(const float4& a, const float4& b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
ATN_DEV float4 cross4(const float4& a, const float4& b)
{
    return make_float4(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x, 0.0F);
}
ATN_DEV float4 normalize4(const float4& v)
{
    float d = ((v.x * v.x + v.y * v.y) + v.z * v.z) + v.w * v.w;
    float inv = 1.0F / sqrtf(d);
    return make_float4(inv * v.x, inv * v.y, inv * v.z, inv * v.w);
}

// aten::mat4 rows as float4 (row-major, M*p): src/libaten/math/mat4.h:171-221
struct m4 { float4 r0, r1, r2, r3; };
ATN_DEV m4 m4_identity()
{
    m4 m;
    m.r0 = make_float4(1, 0, 0, 0); m.r1 = make_float4(0, 1, 0, 0);
    m.r2 = make_float4(0, 0, 1, 0); m.r3 = make_float4(0, 0, 0, 1);
    return m;
}
ATN_DEV f3 m4_apply(const m4& m, const f3& p)
{
    return mk3(((m.r0.x * p.x + m.r0.y * p.y) + m.r0.z * p.z) + m.r0.w,
               ((m.r1.x * p.x + m.r1.y * p.y) + m.r1.z * p.z) + m.r1.w,
               ((m.r2.x * p.x + m.r2.y * p.y) + m.r2.z * p.z) + m.r2.w);
}
ATN_DEV f3 m4_apply_w1(const m4& m, const f3& p)   // vec4 (p, 1) apply, xyz of the result
{
    return mk3(((m.r0.x * p.x + m.r0.y * p.y) + m.r0.z * p.z) + m.r0.w * 1.0F,
               ((m.r1.x * p.x + m.r1.y * p.y) + m.r1.z * p.z) + m.r1.w * 1.0F,
               ((m.r2.x * p.x + m.r2.y * p.y) + m.r2.z * p.z) + m.r2.w * 1.0F);
}
ATN_DEV f3 m4_applyXYZ(const m4& m, const f3& p)
{
    return mk3((m.r0.x * p.x + m.r0.y * p.y) + m.r0.z * p.z,
               (m.r1.x * p.x + m.r1.y * p.y) + m.r1.z * p.z,
               (m.r2.x * p.x + m.r2.y * p.y) + m.r2.z * p.z);
}

// aten::ray::Offset, src/libaten/math/ray.h:26-74
ATN_DEV f3 ray_offset(const f3& o, const f3& n)
{
    constexpr float origin = 1.0F / 32.0F;
    constexpr float float_scale = 1.0F / 65536.0F;
    constexpr float int_scale = 256.0F;
    int32_t ix = (int32_t)(int_scale * n.x);
    int32_t iy = (int32_t)(int_scale * n.y);
    int32_t iz = (int32_t)(int_scale * n.z);
    float px = __int_as_float(__float_as_int(o.x) + (o.x < 0.0F ? -ix : ix));
    float py = __int_as_float(__float_as_int(o.y) + (o.y < 0.0F ? -iy : iy));
    float pz = __int_as_float(__float_as_int(o.z) + (o.z < 0.0F ? -iz : iz));
    return mk3(fabsf(o.x) < origin ? o.x + float_scale * n.x : px,
               fabsf(o.y) < origin ? o.y + float_scale * n.y : py,
               fabsf(o.z) < origin ? o.z + float_scale * n.z : pz);
}

// GetOrthoVector / GetTangentCoordinate, src/libaten/math/vec3.h:290-337
ATN_DEV f3 ortho_vector(const f3& n)
{
    f3 p;
    if (fabsf(n.z) > 0.0F) {
        float k = sqrtf(n.y * n.y + n.z * n.z);
        p = mk3(0.0F, -n.z / k, n.y / k);
    }
    else {
        float k = sqrtf(n.x * n.x + n.y * n.y);
        p = mk3(n.y / k, -n.x / k, 0.0F);
    }
    return normalize(p);
}
ATN_DEV void tangent_coordinate(const f3& n, f3& t, f3& b)
{
    t = ortho_vector(n);
    b = cross(n, t);
    t = cross(b, n);
}

} // namespace atn
