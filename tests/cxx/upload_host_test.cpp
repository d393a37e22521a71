// Host-side unit test of the upload (aten_amd/csrc/host/scene_upload.hpp: pure host C++, compiled here with hipcc for its
// headers only -- no HIP call is made, no GPU is needed): the node image of a small scene with any-hit twins and the
// top-levels-first layout is a set of lists that can be walked along their typed links; planar_area_light accepts a flat lamp
// under a rigid matrix and refuses a bent one and a scaled one.  Prints "ok" or the first thing that is wrong (exit code 1).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../aten_amd/csrc/host/scene_upload.hpp"
#include "../../include/aten_amd_scene.h"

using namespace atn;

static int fail(const std::string& what) { std::printf("FAILED: %s\n", what.c_str()); return 1; }
static int32_t f2i_(float f) { int32_t i; std::memcpy(&i, &f, 4); return i; }

struct Scene {
    std::vector<atn_vec4> pos, nml;
    std::vector<atn_triangle_param> tris;
    std::vector<atn_object_param> objs;
    std::vector<atn_mat4> mtx;
    std::vector<atn_material_param> mats;
    std::vector<atn_light_param> lights;
    std::vector<atn_bvh_node> blas, tlas;
    std::vector<atn_bvh_list> lists;
    atn_scene_desc d{};
};

// a bumpy (or flat) n x n grid of quads over [-1, 1]^2, one polygon object + one instance with matrix pair (I or `scale`)
static bool make_scene(Scene& s, int n, float bump, float scale, float bend)
{
    for (int i = 0; i <= n; i++)
        for (int j = 0; j <= n; j++) {
            const float x = -1.f + 2.f * i / n, z = -1.f + 2.f * j / n;
            float y = bump * std::sin(3.f * x) * std::cos(2.f * z);
            if (i == n && j == n) y += bend;
            s.pos.push_back({ x, y, z, 0.f }); s.nml.push_back({ 0.f, 1.f, 0.f, 0.f });
        }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            const int a = i * (n + 1) + j, b = a + 1, c = a + n + 1, e = c + 1;
            for (int k = 0; k < 2; k++) {
                atn_triangle_param t{};
                if (k == 0) { t.idx[0] = a; t.idx[1] = b; t.idx[2] = e; } else { t.idx[0] = a; t.idx[1] = e; t.idx[2] = c; }
                t.area = 1.f; t.mtrlid = 0; t.mesh_id = 0;
                s.tris.push_back(t);
            }
        }
    std::vector<uint32_t> ids(s.tris.size());
    for (size_t i = 0; i < ids.size(); i++) ids[i] = (uint32_t)i;
    atn_bvh_node* out = nullptr; uint32_t cnt = 0; float bmin[3], bmax[3];
    if (atns_build_blas(s.pos.data(), s.tris.data(), ids.data(), (uint32_t)ids.size(), &out, &cnt, bmin, bmax) != 0) return false;
    s.blas.assign(out, out + cnt); atns_free(out);
    atn_object_param poly{}; poly.type = ATN_OBJ_POLYGONS; poly.object_id = -1; poly.mtx_id = -1; poly.triangle_id = 0; poly.triangle_num = (int32_t)s.tris.size(); poly.light_id = -1;
    atn_object_param inst{}; inst.type = ATN_OBJ_INSTANCE; inst.object_id = 0; inst.mtx_id = 0; inst.triangle_id = -1; inst.light_id = 0;
    s.objs = { poly, inst };
    atn_mat4 l2w{}, w2l{};
    for (int a = 0; a < 4; a++) { l2w.m[a][a] = a < 3 ? scale : 1.f; w2l.m[a][a] = a < 3 ? 1.f / scale : 1.f; }
    s.mtx = { l2w, w2l };
    s.mats.resize(1); std::memset(&s.mats[0], 0, sizeof(s.mats[0])); s.mats[0].type = ATN_MTRL_DIFFUSE; s.mats[0].albedoMap = s.mats[0].normalMap = s.mats[0].roughnessMap = -1;
    atn_light_param l{}; l.type = ATN_LIGHT_AREA; l.arealight_objid = 1; l.envmapidx = -1;
    s.lights = { l };
    const float box[6] = { bmin[0] * scale, bmin[1] * scale, bmin[2] * scale, bmax[0] * scale, bmax[1] * scale, bmax[2] * scale };
    const int32_t oid = 1, lid = 1, mid = 0;
    if (atns_build_tlas(box, &oid, &lid, &mid, 1, &out, &cnt) != 0) return false;
    s.tlas.assign(out, out + cnt); atns_free(out);
    s.lists = { { s.tlas.data(), (uint32_t)s.tlas.size(), 0 }, { s.blas.data(), (uint32_t)s.blas.size(), 0 } };
    atn_scene_desc& d = s.d;
    d.objects = s.objs.data(); d.n_objects = 2; d.matrices = s.mtx.data(); d.n_matrices = 2; d.materials = s.mats.data(); d.n_materials = 1;
    d.lights = s.lights.data(); d.n_lights = 1; d.triangles = s.tris.data(); d.n_triangles = (uint32_t)s.tris.size();
    d.vtx_pos = s.pos.data(); d.vtx_nml = s.nml.data(); d.n_vertices = (uint32_t)s.pos.size();
    d.bvh_lists = s.lists.data(); d.n_bvh_lists = 2; d.textures = nullptr; d.n_textures = 0;
    d.config.bvh_hit_min = -1.f; d.config.bg.envmap_tex_idx = -1;
    for (int a = 0; a < 3; a++) { d.scene_bbox_min[a] = box[a]; d.scene_bbox_max[a] = box[3 + a]; }
    return true;
}

// follow the hit links of the records from `root` (a typed link): every record once, triangle ids collected; false on a bad link
static bool walk(const HostSceneImage& img, int32_t root, uint32_t lo, uint32_t hi, size_t& n_records, std::multiset<int32_t>& tri_ids, bool& offsets_ascend)
{
    const char* base = reinterpret_cast<const char*>(img.nodes.data());
    std::set<uint32_t> seen;
    int32_t link = root;
    uint32_t prev = 0;
    offsets_ascend = true;
    while (link != kLinkEnd) {
        const uint32_t off = (uint32_t)link & kLinkOffsetMask;
        if (off < lo || off >= hi || !seen.insert(off).second) return false;
        if (!seen.empty() && off < prev) offsets_ascend = false;
        prev = off;
        const float4* q = reinterpret_cast<const float4*>(base + off);
        if (link >= 0) link = f2i_(q[0].w);                                  // inner record: hit link
        else if (link & kLinkLeafBit) { tri_ids.insert(f2i_(q[0].w)); link = f2i_(q[1].w); }
        else return false;                                                      // a TLAS leaf inside a bottom-level list
        if (seen.size() > (hi - lo) / 16u) return false;
    }
    n_records = seen.size();
    return true;
}

int main()
{
    {   // ---- twins + layout
        Scene s;
        if (!make_scene(s, 24, 0.15f, 1.0f, 0.0f)) return fail("scene");
        for (int layout : { 0, kLayoutTopLevels }) {
            for (int dirs : { 1, 8 }) {
                HostSceneImage img; std::string err;
                if (!build_host_image(img, &s.d, err, 2, dirs, layout, true)) return fail("build_host_image: " + err);
                const uint32_t lb = img.list_bytes[1], root = img.list_root[1];
                if ((img.list_twin_delta[1] & ~15) != (int32_t)lb || (img.list_twin_delta[1] & 1) != (dirs == 8 ? 1 : 0)) return fail("twin word");
                if (img.list_root[0] != root + (1u + (uint32_t)dirs) * lb) return fail("the top layer does not follow the list and its twins");
                if (((uint32_t)img.list_root_link[1] & kLinkOffsetMask) != root) return fail("the list's root record is not its first");
                std::multiset<int32_t> want;
                for (int g = 0; g <= dirs; g++) {
                    size_t n = 0; std::multiset<int32_t> got; bool asc = false;
                    const uint32_t lo = root + (uint32_t)g * lb;
                    if (!walk(img, img.list_root_link[1] + (int32_t)((uint32_t)g * lb), lo, lo + lb, n, got, asc)) return fail("a list / twin cannot be walked inside its own region");
                    if (n != s.blas.size()) return fail("a walk does not visit every record once");
                    if (g == 0) want = got; else if (got != want) return fail("a twin holds other triangles than its list");
                    if (layout == 0 && !asc) return fail("walk order layout: offsets do not ascend along the hit links");
                    if (layout != 0 && asc && g == 0) return fail("top-levels-first layout equals walk order");
                }
                if (want.size() < s.tris.size()) return fail("triangles missing from the leaves");
                // the TLAS leaf carries the twin word, and the direct-start copy of it
                if (img.tlas_refs.size() != 1 || img.tlas_refs[0].list != 1) return fail("tlas_refs");
                const float4* q = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(img.nodes.data()) + img.tlas_refs[0].offset);
                if (f2i_(q[1].w) != img.list_twin_delta[1] || img.params.root_twin != img.list_twin_delta[1] || !img.params.root_direct) return fail("TLAS leaf twin word");
            }
        }
        HostSceneImage none; std::string err;
        if (!build_host_image(none, &s.d, err, 0, 8, kLayoutTopLevels, true) || none.list_twin_delta[1] != 0 || none.params.root_twin != 0) return fail("twins off");
        // the twins' memory budget (kTwinBudgetBytes): eight twins of this list are 8 x its records -- a budget of 3 x leaves room for the
        // one direction-free twin only, a budget below 1 x for none
        const uint64_t one = (uint64_t)s.blas.size() * 48u;
        HostSceneImage b1, b0;
        if (!build_host_image(b1, &s.d, err, 2, 8, kLayoutTopLevels, true, 3 * one) || b1.list_twin_delta[1] == 0 || (b1.list_twin_delta[1] & 1) != 0) return fail("budget: one twin");
        if (b1.list_root[0] != b1.list_root[1] + 2u * b1.list_bytes[1]) return fail("budget: one twin's records");
        if (!build_host_image(b0, &s.d, err, 2, 8, kLayoutTopLevels, true, one / 2) || b0.list_twin_delta[1] != 0) return fail("budget: no twin");
        // the model gives this bumpy sheet no twin by itself (nothing occludes anything)
        HostSceneImage adaptive;
        if (!build_host_image(adaptive, &s.d, err, 1, 8, kLayoutTopLevels, true)) return fail("adaptive: " + err);
        std::printf("adaptive twin on the sheet: %s\n", adaptive.list_twin_delta[1] ? "yes" : "no");
    }
    {   // ---- planar area lights
        float n[3];
        Scene flat; if (!make_scene(flat, 4, 0.0f, 1.0f, 0.0f)) return fail("scene");
        if (!planar_area_light(&flat.d, flat.lights[0], n) || std::fabs(std::fabs(n[1]) - 1.f) > 1e-6f || std::fabs(n[0]) > 1e-6f) return fail("a flat lamp under the identity is planar, normal +-y");
        HostSceneImage img; std::string err;
        if (!build_host_image(img, &flat.d, err, 0, 8, kLayoutTopLevels, true) || img.light_plane[0].w != 1.0f || !img.params.planar_lights) return fail("light_plane table");
        if (!build_host_image(img, &flat.d, err, 0, 8, kLayoutTopLevels, false) || img.light_plane[0].w != 0.0f || img.params.planar_lights) return fail("planar lights off");
        Scene bent; if (!make_scene(bent, 4, 0.0f, 1.0f, 1e-4f)) return fail("scene");
        if (planar_area_light(&bent.d, bent.lights[0], n)) return fail("a lamp with one vertex 1e-4 out of its plane is not planar");
        Scene scaled; if (!make_scene(scaled, 4, 0.0f, 2.0f, 0.0f)) return fail("scene");
        if (planar_area_light(&scaled.d, scaled.lights[0], n)) return fail("a lamp under a scaling matrix: hit distances are not world distances");
        Scene rot; if (!make_scene(rot, 4, 0.0f, 1.0f, 0.0f)) return fail("scene");
        const float c = std::cos(0.5f), sn = std::sin(0.5f);                  // rotation about z by 0.5 rad: L2W, and W2L = its transpose
        atn_mat4 r{}, rt{};
        r.m[0][0] = c; r.m[0][1] = -sn; r.m[1][0] = sn; r.m[1][1] = c; r.m[2][2] = 1; r.m[3][3] = 1;
        for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) rt.m[a][b] = r.m[b][a];
        rot.mtx = { r, rt }; rot.d.matrices = rot.mtx.data();
        if (!planar_area_light(&rot.d, rot.lights[0], n) || std::fabs(n[0] * n[0] + n[1] * n[1] + n[2] * n[2] - 1.f) > 1e-5f
            || std::fabs(std::fabs(n[0]) - sn) > 1e-5f || std::fabs(std::fabs(n[1]) - c) > 1e-5f) return fail("a rotated lamp: planar, normal rotated with it");
        atn_light_param point = flat.lights[0]; point.type = ATN_LIGHT_POINT;
        if (planar_area_light(&flat.d, point, n)) return fail("only area lights");
    }
    std::printf("ok\n");
    return 0;
}
