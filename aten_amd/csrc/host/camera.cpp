// Host-side camera block for callers that do not link libaten: what aten::PinholeCamera hands to the
// renderer through Camera::param() (src/libaten/camera/pinhole.cpp:34-75, camera/camera.h:15-36).
// An aten application passes its own CameraParameter to atn_update_camera; this entry exists so that the
// scene stand-in (aten_amd/scene), bench.py and smoke() do not need anything outside the product tree.
//
// Built with -ffp-contract=off: the block must carry the same fp32 values the reference's x86-64 SSE2
// build computes, every operation rounded on its own.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "../../../include/aten_amd_scene.h"

namespace {

struct V { float x, y, z; };

inline V sub(const V& a, const V& b) { return V{ a.x - b.x, a.y - b.y, a.z - b.z }; }
inline V add(const V& a, const V& b) { return V{ a.x + b.x, a.y + b.y, a.z + b.z }; }
inline V scale(float s, const V& a) { return V{ s * a.x, s * a.y, s * a.z }; }
inline float dot3(const V& a, const V& b) { const float px = a.x * b.x, py = a.y * b.y, pz = a.z * b.z; return px + py + pz; }
inline V cross3(const V& a, const V& b) { return V{ a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y }; }
// glm::normalize: v * inversesqrt(dot(v, v)), inversesqrt(x) = 1 / sqrt(x)
inline V unit(const V& a) { const float inv = 1.0F / std::sqrt(dot3(a, a)); return V{ a.x * inv, a.y * inv, a.z * inv }; }
inline void put(float* d, const V& s) { d[0] = s.x; d[1] = s.y; d[2] = s.z; }

} // namespace

extern "C" int atns_create_camera(atn_camera_param* out, const float origin[3], const float lookat[3], const float up[3],
                                  float vfov, float z_near, float z_far, int32_t width, int32_t height)
{
    if (!out || !origin || !lookat || !up || width <= 0 || height <= 0) return -1;
    std::memset(out, 0, sizeof(*out));
    const V eye{ origin[0], origin[1], origin[2] }, at{ lookat[0], lookat[1], lookat[2] }, world_up{ up[0], up[1], up[2] };

    const float theta = 3.14159265358979323846F * vfov / 180.0F;      // aten::Deg2Rad, math/math.h:18-21
    out->aspect = (float)width / (float)height;
    const float half_h = std::tan(theta / 2);
    const float half_w = out->aspect * half_h;

    const V dir = unit(sub(at, eye));
    const V right = unit(cross3(dir, world_up));
    const V cam_up = cross3(right, dir);

    put(out->origin, eye);
    put(out->lookat, at);
    put(out->dir, dir);
    put(out->right, right);
    put(out->up, cam_up);
    put(out->center, add(eye, dir));
    put(out->u, scale(half_w, right));          // screen half-extent vectors
    put(out->v, scale(half_h, cam_up));
    out->dist = (float)height / (2.0F * std::tan(theta / 2));
    out->vfov = vfov;
    out->width = width;
    out->height = height;
    out->znear = std::min(z_near, z_far);
    out->zfar = std::max(z_near, z_far);
    return 0;
}
