/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * orc_math.h: scalar fp32 vector math restating, operation for operation, what aten's CPU
 * build evaluates for vec3/vec4/mat4.
 *
 * aten::vec3 IS glm::highp_vec3 (src/libaten/math/vec3.h:213) and unqualified dot/cross/
 * normalize resolve into glm by ADL.  glm (g-truc/glm, submodule 3rdparty/glm, pinned commit
 * unknown: the snapshot holds an empty directory) is ABSENT from /root/reference, so the
 * reference cannot be compiled here and this file restates glm's published scalar
 * definitions (glm/detail/func_geometric.inl, func_exponential.inl):
 *     dot(a,b)       = { tmp = a*b; tmp.x + tmp.y + tmp.z }
 *     cross(x,y)     = ( x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x, x.x*y.y - y.x*x.y )
 *     inversesqrt(x) = 1 / sqrt(x)
 *     normalize(v)   = v * inversesqrt(dot(v,v))
 *     length(v)      = sqrt(dot(v,v))
 *     v / s, s / v, v * s : component-wise, no reciprocal trick
 * No reference test pins results at this boundary -> PARITY UNPINNED for it.
 *
 * Build rule: -ffp-contract=off, no -ffast-math (the reference is built -O3 for x86-64 SSE2
 * without FMA, src/CMakeLists.txt:18-19).
 */
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <limits>

namespace orc {

constexpr float PI = 3.14159265358979323846F;      // src/libaten/math/math.h:10
constexpr float PI_2 = PI * 2;                      // math.h:11
constexpr float INF = std::numeric_limits<float>::max();   // AT_MATH_INF, math.h:14
constexpr float EPS = 1e-9F;                        // AT_MATH_EPSILON, math.h:15

// std::max / std::min exactly as the host build uses them (math.h:148-152,176-180):
// max(a,b) = (a < b) ? b : a ; min(a,b) = (b < a) ? b : a.  NaN behaviour matters in aabb::hit.
inline float fmax_(float a, float b) { return (a < b) ? b : a; }
inline float fmin_(float a, float b) { return (b < a) ? b : a; }
inline float clamp_(float v, float lo, float hi) { return (v < lo) ? lo : (hi < v) ? hi : v; } // std::clamp
inline float saturate_(float v) { return clamp_(v, 0.0F, 1.0F); }
inline float sqr(float f) { return f * f; }
inline int32_t float_as_int(float f) { int32_t i; std::memcpy(&i, &f, 4); return i; }
inline float int_as_float(int32_t i) { float f; std::memcpy(&f, &i, 4); return f; }

struct v3 {
    float x, y, z;
    v3() : x(0), y(0), z(0) {}
    explicit v3(float f) : x(f), y(f), z(f) {}
    v3(float a, float b, float c) : x(a), y(b), z(c) {}
};
inline v3 operator+(const v3& a, const v3& b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline v3 operator-(const v3& a, const v3& b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline v3 operator*(const v3& a, const v3& b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline v3 operator/(const v3& a, const v3& b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline v3 operator*(const v3& a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline v3 operator*(float s, const v3& a) { return v3(s * a.x, s * a.y, s * a.z); }
inline v3 operator/(const v3& a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
inline v3 operator/(float s, const v3& a) { return v3(s / a.x, s / a.y, s / a.z); }
inline v3 operator+(const v3& a, float s) { return v3(a.x + s, a.y + s, a.z + s); }
inline v3 operator-(const v3& a) { return v3(-a.x, -a.y, -a.z); }
inline v3& operator+=(v3& a, const v3& b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
inline v3& operator*=(v3& a, const v3& b) { a.x *= b.x; a.y *= b.y; a.z *= b.z; return a; }
inline v3& operator/=(v3& a, float s) { a.x /= s; a.y /= s; a.z /= s; return a; }

inline float dot(const v3& a, const v3& b) { v3 t = a * b; return t.x + t.y + t.z; }
inline v3 cross(const v3& x, const v3& y)
{
    return v3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
inline float inversesqrt(float x) { return 1.0F / std::sqrt(x); }
inline v3 normalize(const v3& v) { return v * inversesqrt(dot(v, v)); }
inline float length(const v3& v) { return std::sqrt(dot(v, v)); }
inline float squared_length(const v3& v) { return dot(v, v); }
inline float max_from_vec3(const v3& v) { return fmax_(fmax_(v.x, v.y), v.z); }   // vec3.h:345-348
inline float min_from_vec3(const v3& v) { return fmin_(fmin_(v.x, v.y), v.z); }   // vec3.h:359-362
inline v3 vmin(const v3& a, const v3& b) { return v3(fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z)); }
inline v3 vmax(const v3& a, const v3& b) { return v3(fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z)); }
inline v3 mix(const v3& a, const v3& b, float t) { return a * (1.0F - t) + b * t; }     // vec3.h:260-264
inline float mix(float a, float b, float t) { return a * (1 - t) + b * t; }             // math.h:319-325

// aten::vec4 (src/libaten/math/vec4.h).  Default w = 1 (vec4.h:22-26).
struct v4 {
    float x, y, z, w;
    v4() : x(0), y(0), z(0), w(1) {}
    v4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    explicit v4(float f) : x(f), y(f), z(f), w(f) {}
    v4(const v3& v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
    explicit v4(const v3& v) : x(v.x), y(v.y), z(v.z), w(0) {}       // vec4.h:75-79
    v3 xyz() const { return v3(x, y, z); }
};
inline v4 operator+(const v4& a, const v4& b) { return v4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline v4 operator-(const v4& a, const v4& b) { return v4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
inline v4 operator*(const v4& a, const v4& b) { return v4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
inline v4 operator*(float t, const v4& v) { return v4(t * v.x, t * v.y, t * v.z, t * v.w); }
inline v4 operator*(const v4& v, float t) { return v4(t * v.x, t * v.y, t * v.z, t * v.w); }
inline float dot(const v4& a, const v4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
inline v4 cross(const v4& a, const v4& b)    // vec4.h:282-291
{
    return v4(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x, 0);
}
inline v4 normalize(const v4& v) { float inv = 1.0F / std::sqrt(dot(v, v)); return v * inv; }  // vec4.h:293-298
inline float length3(const v4& v) { return std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z); }      // vec4::length

// aten::mat4 row-major, M*p (src/libaten/math/mat4.h:171-235)
struct m4 {
    float m[4][4];
    static m4 identity()
    {
        m4 r; std::memset(&r, 0, sizeof(r));
        r.m[0][0] = r.m[1][1] = r.m[2][2] = r.m[3][3] = 1; return r;
    }
    v3 apply(const v3& p) const
    {
        v3 r;
        r.x = m[0][0] * p.x + m[0][1] * p.y + m[0][2] * p.z + m[0][3];
        r.y = m[1][0] * p.x + m[1][1] * p.y + m[1][2] * p.z + m[1][3];
        r.z = m[2][0] * p.x + m[2][1] * p.y + m[2][2] * p.z + m[2][3];
        return r;
    }
    v4 apply(const v4& p) const
    {
        v4 r;
        r.x = m[0][0] * p.x + m[0][1] * p.y + m[0][2] * p.z + m[0][3] * p.w;
        r.y = m[1][0] * p.x + m[1][1] * p.y + m[1][2] * p.z + m[1][3] * p.w;
        r.z = m[2][0] * p.x + m[2][1] * p.y + m[2][2] * p.z + m[2][3] * p.w;
        r.w = m[3][0] * p.x + m[3][1] * p.y + m[3][2] * p.z + m[3][3] * p.w;
        return r;
    }
    v3 applyXYZ(const v3& p) const
    {
        v3 r;
        r.x = m[0][0] * p.x + m[0][1] * p.y + m[0][2] * p.z;
        r.y = m[1][0] * p.x + m[1][1] * p.y + m[1][2] * p.z;
        r.z = m[2][0] * p.x + m[2][1] * p.y + m[2][2] * p.z;
        return r;
    }
};

// aten::ray (src/libaten/math/ray.h).  Both constructors re-normalise dir.
struct Ray {
    v3 org, dir;
    Ray() {}
    Ray(const v3& o, const v3& d) { dir = normalize(d); org = o; }
    Ray(const v3& o, const v3& d, const v3& n) { dir = normalize(d); org = Offset(o, n); }

    // ray::Offset, ray.h:26-74 ("A Fast and Robust Method for Avoiding Self-Intersection").
    static v3 Offset(const v3& o, const v3& n)
    {
        constexpr float origin = 1.0F / 32.0F;
        constexpr float float_scale = 1.0F / 65536.0F;
        constexpr float int_scale = 256.0F;
        int32_t of_ix = static_cast<int32_t>(int_scale * n.x);
        int32_t of_iy = static_cast<int32_t>(int_scale * n.y);
        int32_t of_iz = static_cast<int32_t>(int_scale * n.z);
        v3 p_i(
            int_as_float(float_as_int(o.x) + (o.x < 0.0F ? -of_ix : of_ix)),
            int_as_float(float_as_int(o.y) + (o.y < 0.0F ? -of_iy : of_iy)),
            int_as_float(float_as_int(o.z) + (o.z < 0.0F ? -of_iz : of_iz)));
        return v3(
            std::fabs(o.x) < origin ? o.x + float_scale * n.x : p_i.x,
            std::fabs(o.y) < origin ? o.y + float_scale * n.y : p_i.y,
            std::fabs(o.z) < origin ? o.z + float_scale * n.z : p_i.z);
    }
};

// GetOrthoVector / GetTangentCoordinate, src/libaten/math/vec3.h:290-337
inline v3 GetOrthoVector(const v3& n)
{
    v3 p;
    if (std::fabs(n.z) > 0.0F) {
        float k = std::sqrt(n.y * n.y + n.z * n.z);
        p.x = 0; p.y = -n.z / k; p.z = n.y / k;
    }
    else {
        float k = std::sqrt(n.x * n.x + n.y * n.y);
        p.x = n.y / k; p.y = -n.x / k; p.z = 0;
    }
    return normalize(p);
}
inline void GetTangentCoordinate(const v3& n, v3& t, v3& b)
{
    t = GetOrthoVector(n);
    b = cross(n, t);
    t = cross(b, n);
}

// color::luminance, src/libaten/misc/color.h:61-73
inline float luminance(float r, float g, float b) { return 0.212639F * r + 0.71517F * g + 0.0721926F * b; }
inline float luminance(const v3& c) { return luminance(c.x, c.y, c.z); }

} // namespace orc
