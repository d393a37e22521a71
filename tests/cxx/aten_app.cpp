// A minimal "aten application" in C++ that drives the renderer ONLY through the C-ABI of include/*.h, the way
// INTEGRATION.md's adapter does (≙ src/device_renderer/main.cpp:133-149,196-204: UpdateSceneData once, render per frame).
// It builds a small scene with the host library (atns_build_blas / atns_build_tlas / atns_create_camera), uploads it,
// renders progressive frames, and dumps the scene arrays, the camera and the film so that the Python test can hand
// exactly the same arrays to the CPU oracle.  No HIP, no torch, no Python on this side of the boundary.
//
//   g++ -std=c++17 -I include tests/cxx/aten_app.cpp -L aten_amd -laten_amd -laten_amd_scene -Wl,-rpath,$PWD/aten_amd -o aten_app
//   ./aten_app <out_dir> <width> <height> <frames>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "aten_amd.h"
#include "aten_amd_scene.h"

namespace {

struct App {
    std::vector<atn_vec4> pos, nml;
    std::vector<atn_triangle_param> tris;
    std::vector<atn_material_param> mtrls;
    std::vector<atn_object_param> objs;
    std::vector<atn_mat4> mtxs;
    std::vector<atn_light_param> lights;
    std::vector<std::vector<atn_bvh_node>> lists{ 1 };     // [0] = top layer
    struct Inst { int32_t obj, list; float mn[3], mx[3]; };
    std::vector<Inst> insts;

    int add_material(int32_t type, uint32_t attrib, float r, float g, float b, float roughness = 0.5F, float ior = 1.0F)
    {
        atn_material_param m;
        std::memset(&m, 0, sizeof(m));
        m.baseColor = atn_vec4{ r, g, b, 1.0F };
        m.type = type; m.attrib = attrib; m.id = (uint16_t)mtrls.size();
        m.albedoMap = m.normalMap = m.roughnessMap = -1;
        m.u.standard = atn_standard_mtrl{ ior, roughness, 1.0F, 0.5F, 0.5F, 0.5F, 0.5F, 0.5F, 0.5F, 0.5F, 0.5F, 0.5F };
        mtrls.push_back(m);
        return (int)mtrls.size() - 1;
    }

    // one polygon object made of quads (two triangles each, unshared vertices), instanced with the identity
    int add_quads(const std::vector<float>& q /* 12 floats per quad */, int mtrl, int mesh_id)
    {
        const uint32_t first = (uint32_t)tris.size();
        for (size_t k = 0; k + 12 <= q.size(); k += 12) {
            const float* v = &q[k];
            const int order[6] = { 0, 1, 2, 0, 2, 3 };
            for (int t = 0; t < 2; t++) {
                atn_triangle_param tp;
                std::memset(&tp, 0, sizeof(tp));
                float p[3][3];
                for (int c = 0; c < 3; c++) {
                    const int vi = order[3 * t + c];
                    tp.idx[c] = (int32_t)pos.size();
                    for (int d = 0; d < 3; d++) p[c][d] = v[3 * vi + d];
                    pos.push_back(atn_vec4{ p[c][0], p[c][1], p[c][2], 0.0F });
                    nml.push_back(atn_vec4{ 0.0F, 1.0F, 0.0F, 0.0F });
                }
                // triangle::BuildTriangle: area = 0.5 * |cross(e0, e1)| (triangle.cpp:125-130)
                const float e0[3] = { p[1][0] - p[0][0], p[1][1] - p[0][1], p[1][2] - p[0][2] };
                const float e1[3] = { p[2][0] - p[0][0], p[2][1] - p[0][1], p[2][2] - p[0][2] };
                const float cx = e0[1] * e1[2] - e0[2] * e1[1], cy = e0[2] * e1[0] - e0[0] * e1[2], cz = e0[0] * e1[1] - e0[1] * e1[0];
                tp.area = 0.5F * std::sqrt((cx * cx + cy * cy) + cz * cz);
                tp.needNormal = 1; tp.mtrlid = mtrl; tp.mesh_id = mesh_id;
                tris.push_back(tp);
            }
        }
        const uint32_t num = (uint32_t)tris.size() - first;
        atn_object_param o;
        std::memset(&o, 0, sizeof(o));
        o.type = ATN_OBJ_POLYGONS; o.object_id = -1; o.mtx_id = -1; o.light_id = -1;
        o.triangle_id = (int32_t)first; o.triangle_num = (int32_t)num;
        float area = 0.0F;
        for (uint32_t t = first; t < first + num; t++) area += tris[t].area;
        o.area = area;
        const int poly = (int)objs.size();
        objs.push_back(o);
        // the bottom-level tree
        std::vector<uint32_t> ids(num);
        for (uint32_t t = 0; t < num; t++) ids[t] = first + t;
        atn_bvh_node* nodes = nullptr; uint32_t cnt = 0; float mn[3], mx[3];
        if (atns_build_blas(pos.data(), tris.data(), ids.data(), num, &nodes, &cnt, mn, mx) != 0) { std::fprintf(stderr, "atns_build_blas failed\n"); std::exit(2); }
        lists.emplace_back(nodes, nodes + cnt);
        atns_free(nodes);
        // TransformableFactory::createInstance: an (L2W, W2L) matrix pair and an Instance entry
        atn_mat4 id;
        std::memset(&id, 0, sizeof(id));
        for (int i = 0; i < 4; i++) id.m[i][i] = 1.0F;
        atn_object_param in;
        std::memset(&in, 0, sizeof(in));
        in.type = ATN_OBJ_INSTANCE; in.object_id = poly; in.mtx_id = (int32_t)mtxs.size(); in.light_id = -1; in.triangle_id = -1;
        mtxs.push_back(id); mtxs.push_back(id);
        objs.push_back(in);
        Inst it; it.obj = (int)objs.size() - 1; it.list = (int)lists.size() - 1;
        std::memcpy(it.mn, mn, sizeof(mn)); std::memcpy(it.mx, mx, sizeof(mx));
        insts.push_back(it);
        return it.obj;
    }
};

void dump(const std::string& dir, const char* name, const void* p, size_t bytes)
{
    const std::string path = dir + "/" + name;
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f || (bytes && std::fwrite(p, 1, bytes, f) != bytes)) { std::fprintf(stderr, "cannot write %s\n", path.c_str()); std::exit(2); }
    std::fclose(f);
}

#define CHECK(call) do { int rc_ = (call); if (rc_ != ATN_OK) { std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ctx ? atn_last_error(ctx) : "(no context)"); return 1; } } while (0)

} // namespace

int main(int argc, char** argv)
{
    if (argc < 5) { std::fprintf(stderr, "usage: aten_app <out_dir> <width> <height> <frames>\n"); return 2; }
    const std::string out = argv[1];
    const int32_t W = std::atoi(argv[2]), H = std::atoi(argv[3]), frames = std::atoi(argv[4]);

    if (atns_abi_version() != ATNS_ABI_VERSION) { std::fprintf(stderr, "libaten_amd_scene.so speaks ABI %u, this program %u\n", atns_abi_version(), ATNS_ABI_VERSION); return 2; }
    App a;
    const int white = a.add_material(ATN_MTRL_DIFFUSE, 0, 0.75F, 0.75F, 0.75F);
    const int red = a.add_material(ATN_MTRL_DIFFUSE, 0, 0.75F, 0.2F, 0.2F);
    const int glossy = a.add_material(ATN_MTRL_GGX, ATN_MTRL_ATTR_GLOSSY, 0.7F, 0.6F, 0.5F, 0.2F, 1.5F);
    const int emit = a.add_material(ATN_MTRL_EMISSIVE, ATN_MTRL_ATTR_EMISSIVE, 1.0F, 1.0F, 1.0F);
    // a room of five quads, a tilted glossy panel, a ceiling light
    a.add_quads({ -1, 0, 1,  1, 0, 1,  1, 0, -1,  -1, 0, -1,        // floor
                  -1, 2, -1,  1, 2, -1,  1, 2, 1,  -1, 2, 1,        // ceiling
                  -1, 0, -1,  1, 0, -1,  1, 2, -1,  -1, 2, -1 },    // back wall
                white, 0);
    a.add_quads({ -1, 0, 1,  -1, 0, -1,  -1, 2, -1,  -1, 2, 1,      // left wall
                  1, 0, -1,  1, 0, 1,  1, 2, 1,  1, 2, -1 },        // right wall
                red, 1);
    a.add_quads({ -0.5F, 0.3F, 0.3F,  0.5F, 0.3F, 0.3F,  0.5F, 1.0F, -0.4F,  -0.5F, 1.0F, -0.4F }, glossy, 2);
    const int light_inst = a.add_quads({ -0.3F, 1.98F, 0.3F,  0.3F, 1.98F, 0.3F,  0.3F, 1.98F, -0.3F,  -0.3F, 1.98F, -0.3F }, emit, 3);
    {   // AreaLight over the light instance (and its polygon object, like the scene builder of the Python harness)
        atn_light_param l;
        std::memset(&l, 0, sizeof(l));
        l.type = ATN_LIGHT_AREA; l.attrib = 0;
        l.light_color[0] = l.light_color[1] = l.light_color[2] = 1.0F;
        l.innerAngle = l.outerAngle = 3.14159265358979323846F;
        l.scale = 1.0F; l.intensity = 60.0F;
        l.arealight_objid = light_inst; l.envmapidx = -1;
        a.lights.push_back(l);
        a.objs[light_inst].light_id = 0;
        a.objs[a.objs[light_inst].object_id].light_id = 0;
    }
    // the top layer
    {
        std::vector<float> boxes; std::vector<int32_t> oids, exids, mesh;
        for (const auto& it : a.insts) {
            boxes.insert(boxes.end(), it.mn, it.mn + 3); boxes.insert(boxes.end(), it.mx, it.mx + 3);
            oids.push_back(it.obj); exids.push_back(it.list); mesh.push_back(-1);
        }
        atn_bvh_node* nodes = nullptr; uint32_t cnt = 0;
        if (atns_build_tlas(boxes.data(), oids.data(), exids.data(), mesh.data(), (uint32_t)a.insts.size(), &nodes, &cnt) != 0) { std::fprintf(stderr, "atns_build_tlas failed\n"); return 2; }
        a.lists[0].assign(nodes, nodes + cnt);
        atns_free(nodes);
    }

    std::vector<atn_bvh_list> lists(a.lists.size());
    for (size_t k = 0; k < a.lists.size(); k++) { lists[k].nodes = a.lists[k].data(); lists[k].count = (uint32_t)a.lists[k].size(); lists[k]._pad = 0; }
    atn_scene_desc d;
    std::memset(&d, 0, sizeof(d));
    d.objects = a.objs.data(); d.n_objects = (uint32_t)a.objs.size();
    d.matrices = a.mtxs.data(); d.n_matrices = (uint32_t)a.mtxs.size();
    d.materials = a.mtrls.data(); d.n_materials = (uint32_t)a.mtrls.size();
    d.lights = a.lights.data(); d.n_lights = (uint32_t)a.lights.size();
    d.triangles = a.tris.data(); d.n_triangles = (uint32_t)a.tris.size();
    d.vtx_pos = a.pos.data(); d.vtx_nml = a.nml.data(); d.n_vertices = (uint32_t)a.pos.size();
    d.bvh_lists = lists.data(); d.n_bvh_lists = (uint32_t)lists.size();
    d.config.bvh_hit_min = -1.0F;
    d.config.epsilon_bias_for_traversing_shadow_ray_in_medium = 1e-3F;
    d.config.bg.envmap_tex_idx = -1; d.config.bg.avgIllum = 1.0F; d.config.bg.multiplyer = 1.0F; d.config.bg.enable_env_map = 1;
    d.scene_bbox_min[0] = -1; d.scene_bbox_min[1] = 0; d.scene_bbox_min[2] = -1;
    d.scene_bbox_max[0] = 1; d.scene_bbox_max[1] = 2; d.scene_bbox_max[2] = 1;
    d.enable_shadowray_base_stylized_shadow = 1;

    atn_camera_param cam;
    const float org[3] = { 0.0F, 1.0F, 3.2F }, at[3] = { 0.0F, 1.0F, 0.0F }, up[3] = { 0.0F, 1.0F, 0.0F };
    if (atns_create_camera(&cam, org, at, up, 40.0F, 0.1F, 10000.0F, W, H) != 0) return 2;

    if (atn_sizeof_scene_desc() != sizeof(atn_scene_desc) || atn_sizeof_destination() != sizeof(atn_destination)) {
        std::fprintf(stderr, "header / library mismatch\n"); return 2;
    }
    atn_ctx* ctx = nullptr;
    CHECK(atn_create(&ctx, 0));
    CHECK(atn_upload_scene(ctx, &d));
    CHECK(atn_update_camera(ctx, &cam));
    CHECK(atn_init_sampler(ctx, W, H, 0));
    std::vector<atn_vec4> film((size_t)W * H);
    for (int32_t f = 0; f < frames; f++) {
        atn_destination dst;
        std::memset(&dst, 0, sizeof(dst));
        dst.width = W; dst.height = H; dst.maxDepth = 5; dst.russianRouletteDepth = 3; dst.sample = 1; dst.frame = (uint32_t)f;
        dst.progressive = 1; dst.break_on_terminate = 1;
        CHECK(atn_render(ctx, &dst, f + 1 == frames ? film.data() : nullptr));
    }
    CHECK(atn_synchronize(ctx));
    // an error must come back as a code and a message, not as a crash
    if (atn_update_camera(ctx, nullptr) == ATN_OK || std::strlen(atn_last_error(ctx)) == 0) { std::fprintf(stderr, "null camera accepted\n"); return 1; }

    // ---- the whole-node renderer (atn_mgpu_*): three shards sharing device 0 must give the same film, byte for byte
    {
        atn_mgpu* mg = nullptr;
        const int32_t devs[3] = { 0, 0, 0 };
        int rc = atn_mgpu_create(&mg, devs, 3);
        if (rc != ATN_OK) { std::fprintf(stderr, "atn_mgpu_create -> %d\n", rc); return 1; }
#define MCHECK(call) do { int rc_ = (call); if (rc_ != ATN_OK) { std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, atn_mgpu_last_error(mg)); return 1; } } while (0)
        MCHECK(atn_mgpu_upload_scene(mg, &d));
        MCHECK(atn_mgpu_update_camera(mg, &cam));
        MCHECK(atn_mgpu_init_sampler(mg, W, H, 0));
        MCHECK(atn_mgpu_set_frames_in_flight(mg, 2));
        std::vector<atn_vec4> mfilm((size_t)W * H);
        for (int32_t f = 0; f < frames; f++) {
            atn_destination dst;
            std::memset(&dst, 0, sizeof(dst));
            dst.width = W; dst.height = H; dst.maxDepth = 5; dst.russianRouletteDepth = 3; dst.sample = 1; dst.frame = (uint32_t)f;
            dst.progressive = 1; dst.break_on_terminate = 1;
            MCHECK(atn_mgpu_render(mg, &dst, f + 1 == frames ? mfilm.data() : nullptr));
        }
        MCHECK(atn_mgpu_synchronize(mg));
        if (atn_mgpu_shard_count(mg) != 3 || std::memcmp(mfilm.data(), film.data(), film.size() * sizeof(atn_vec4)) != 0) {
            std::fprintf(stderr, "3-shard film differs from the single-context film\n"); return 1;
        }
        atn_mgpu_destroy(mg);
    }
    // ---- SVGF on the same context: two frames, finite output
    {
        std::vector<atn_vec4> den((size_t)W * H);
        CHECK(atn_reset(ctx));
        for (int32_t f = 0; f < 2; f++) {
            atn_destination dst;
            std::memset(&dst, 0, sizeof(dst));
            dst.width = W; dst.height = H; dst.maxDepth = 5; dst.russianRouletteDepth = 3; dst.sample = 1; dst.frame = (uint32_t)f;
            dst.break_on_terminate = 1;
            CHECK(atn_svgf_render(ctx, &dst, /*compute_motion*/ 1, den.data(), nullptr));
        }
        double s = 0;
        for (const auto& p : den) { if (!(p.x == p.x) || !(p.y == p.y) || !(p.z == p.z)) { std::fprintf(stderr, "SVGF output has NaN\n"); return 1; } s += p.x + p.y + p.z; }
        if (!(s > 0)) { std::fprintf(stderr, "SVGF output is black\n"); return 1; }
    }
    // ---- a deformation tick: the glossy panel (list 3: one quad = two triangles, three nodes) is moved, its list is
    // rebuilt on the device (atn_lbvh_rebuild_list), the top layer re-sent; the hit records must see the new geometry
    {
        const uint32_t t0 = 10, n = 2, v0 = 30;          // the panel: triangles 10..11, vertices 30..35
        for (uint32_t v = v0; v < v0 + 6; v++) { a.pos[v].x += 0.25F; a.pos[v].y += 0.1F; }
        float mn[3] = { 1e30F, 1e30F, 1e30F }, mx[3] = { -1e30F, -1e30F, -1e30F };
        for (uint32_t v = v0; v < v0 + 6; v++) {
            const float q[3] = { a.pos[v].x, a.pos[v].y, a.pos[v].z };
            for (int c = 0; c < 3; c++) { mn[c] = q[c] < mn[c] ? q[c] : mn[c]; mx[c] = q[c] > mx[c] ? q[c] : mx[c]; }
        }
        CHECK(atn_update_geometry(ctx, &a.pos[v0], &a.nml[v0], 6, v0, &a.tris[t0], n, t0));
        CHECK(atn_lbvh_rebuild_list(ctx, 3, t0, n, mn, mx));
        // the same builder as a function: 2 n - 1 nodes, leaves carry the scene's triangle ids
        atn_bvh_node ln[3];
        CHECK(atn_lbvh_build(ctx, &a.tris[t0], n, (int32_t)t0, mn, mx, a.pos.data(), (uint32_t)a.pos.size(), 0, ln, nullptr, nullptr));
        if (!(ln[0].f0 < 0) || ln[1].f1 + ln[2].f1 != (float)(2 * t0 + 1)) { std::fprintf(stderr, "atn_lbvh_build: unexpected nodes\n"); return 1; }
        // new top layer (the panel's box moved)
        std::vector<float> boxes; std::vector<int32_t> oids, exids, mesh;
        for (auto& it : a.insts) {
            if (it.list == 3) { std::memcpy(it.mn, mn, sizeof(mn)); std::memcpy(it.mx, mx, sizeof(mx)); }
            boxes.insert(boxes.end(), it.mn, it.mn + 3); boxes.insert(boxes.end(), it.mx, it.mx + 3);
            oids.push_back(it.obj); exids.push_back(it.list); mesh.push_back(-1);
        }
        atn_bvh_node* nodes = nullptr; uint32_t cnt = 0;
        if (atns_build_tlas(boxes.data(), oids.data(), exids.data(), mesh.data(), (uint32_t)a.insts.size(), &nodes, &cnt) != 0) return 2;
        CHECK(atn_update_tlas(ctx, a.objs.data(), (uint32_t)a.objs.size(), a.mtxs.data(), (uint32_t)a.mtxs.size(), nodes, cnt));
        atns_free(nodes);
        // a ray straight at the panel's new centre
        atn_ray r;
        const float c3[3] = { 0.5F * (mn[0] + mx[0]), 0.5F * (mn[1] + mx[1]), 0.5F * (mn[2] + mx[2]) };
        r.org[0] = c3[0]; r.org[1] = c3[1]; r.org[2] = 3.0F; r.dir[0] = 0; r.dir[1] = 0; r.dir[2] = -1;
        atn_intersection is;
        CHECK(atn_trace_closest(ctx, &r, 1, 1e-9F, 3.402823466e+38F, &is, nullptr));
        if (is.objid != a.insts[2].obj || (is.tri_id != (int32_t)t0 && is.tri_id != (int32_t)t0 + 1)) {
            std::fprintf(stderr, "after the tick the ray hits object %d triangle %d\n", is.objid, is.tri_id); return 1;
        }
        for (uint32_t v = v0; v < v0 + 6; v++) { a.pos[v].x -= 0.25F; a.pos[v].y -= 0.1F; }       // the dump below is the scene as rendered
    }
    atn_destroy(ctx);

    dump(out, "objects.bin", a.objs.data(), a.objs.size() * sizeof(atn_object_param));
    dump(out, "matrices.bin", a.mtxs.data(), a.mtxs.size() * sizeof(atn_mat4));
    dump(out, "materials.bin", a.mtrls.data(), a.mtrls.size() * sizeof(atn_material_param));
    dump(out, "lights.bin", a.lights.data(), a.lights.size() * sizeof(atn_light_param));
    dump(out, "triangles.bin", a.tris.data(), a.tris.size() * sizeof(atn_triangle_param));
    dump(out, "vtx_pos.bin", a.pos.data(), a.pos.size() * sizeof(atn_vec4));
    dump(out, "vtx_nml.bin", a.nml.data(), a.nml.size() * sizeof(atn_vec4));
    for (size_t k = 0; k < a.lists.size(); k++)
        dump(out, ("bvh_" + std::to_string(k) + ".bin").c_str(), a.lists[k].data(), a.lists[k].size() * sizeof(atn_bvh_node));
    dump(out, "config.bin", &d.config, sizeof(d.config));
    dump(out, "camera.bin", &cam, sizeof(cam));
    dump(out, "film.bin", film.data(), film.size() * sizeof(atn_vec4));
    double sum = 0;
    for (const auto& p : film) sum += p.x + p.y + p.z;
    std::printf("aten_app: %d lists, %zu triangles, %d frames, mean %.6f\n", (int)a.lists.size(), a.tris.size(), frames, sum / (3.0 * film.size()));
    return 0;
}
