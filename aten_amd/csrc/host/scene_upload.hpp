// Host-side conversion of the caller's flat scene (include/aten_layout.h) into the device layout
// of device/scene_dev.hpp.  Pure host C++ (no HIP calls) so that it can be unit-tested on CPU.
#pragma once
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../device/scene_dev.hpp"

namespace atn {

struct HostSceneImage {
    std::vector<float4> nodes;          // 3 per node
    std::vector<uint32_t> list_root;    // absolute index of each list's first node
    std::vector<int32_t> list_root_link; // typed link of each list's root
    std::vector<atn_triangle_param> tris;
    std::vector<float4> vtx_pos, vtx_nml;
    std::vector<atn_object_param> objects;
    std::vector<float4> matrices;
    std::vector<DevMaterial> materials;
    std::vector<atn_light_param> lights;
    std::vector<float4> texels;
    std::vector<DevTexture> textures;
    DevScene params{};                  // scalar fields filled; pointers left null
    uint64_t n_inner = 0, n_tri_leaf = 0, n_tlas_leaf = 0;
};

inline float i2f(int32_t i) { float f; std::memcpy(&f, &i, 4); return f; }
inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

// Walk order of a threaded list = following hit links from node 0 (every node's hit link is the
// next node in the depth-first order the traverser visits when nothing is culled).
inline bool walk_order(const atn_bvh_node* nodes, uint32_t count, std::vector<int32_t>& new_index, std::vector<uint32_t>& order, std::string& err)
{
    new_index.assign(count, -1);
    order.clear();
    order.reserve(count);
    int32_t id = count ? 0 : -1;
    while (id >= 0) {
        if ((uint32_t)id >= count) { err = "BVH hit link out of range"; return false; }
        if (new_index[id] >= 0) { err = "BVH hit links form a cycle"; return false; }
        new_index[id] = (int32_t)order.size();
        order.push_back((uint32_t)id);
        id = (int32_t)nodes[id].hit;
    }
    return true;
}

// What a list's records need from the rest of the scene.
struct ListEmitCtx {
    const atn_object_param* objects = nullptr; uint32_t n_objects = 0; uint32_t n_matrices = 0;
    const atn_triangle_param* tris = nullptr; const atn_vec4* vtx_pos = nullptr; uint32_t n_triangles = 0, n_vertices = 0;
    const int32_t* list_root_link = nullptr; uint32_t n_lists = 0;     // typed link of list k's root; kLinkEnd = empty list
    bool top = false;
};

// Emits the device records of one threaded list in walk order at absolute node index `base`.
// Returns the typed link of the list's root through `root_link`.
inline bool emit_list(float4* out, const atn_bvh_node* src, uint32_t count, uint32_t base, const ListEmitCtx& c,
                      int32_t& root_link, uint64_t counts[3], std::string& err)
{
    std::vector<int32_t> new_index;
    std::vector<uint32_t> order;
    if (!walk_order(src, count, new_index, order, err)) return false;
    auto type_bits = [&](uint32_t old_idx) -> int32_t {
        const atn_bvh_node& n = src[old_idx];
        if (!(n.f0 >= 0 || n.f1 >= 0)) return 0;        // inner
        if (n.f2 >= 0) return kLinkTlasBit;             // nested tree
        if (n.f1 >= 0) return kLinkLeafBit;             // triangle
        return 0;                                       // dead leaf: handled on the inner path by its tag
    };
    // typed link of a float link: kLinkEnd, or absolute byte offset | type bits; -2 = invalid
    auto remap = [&](float link) -> int32_t {
        const int32_t l = (int32_t)link;
        if (l < 0) return kLinkEnd;
        if ((uint32_t)l >= count || new_index[l] < 0) return -2;
        const int32_t abs = (int32_t)base + new_index[l];
        return (int32_t)((uint32_t)abs * kNodeBytes) | type_bits((uint32_t)l);
    };
    root_link = count ? remap(0.0F) : kLinkEnd;
    for (uint32_t j = 0; j < order.size(); j++) {
        const atn_bvh_node& n = src[order[j]];
        const uint32_t abs = base + j;
        float4& q0 = out[3 * (size_t)j + 0];
        float4& q1 = out[3 * (size_t)j + 1];
        float4& q2 = out[3 * (size_t)j + 2];
        q0 = q1 = q2 = make_float4(0, 0, 0, 0);
        const int32_t h = remap(n.hit), m = remap(n.miss);
        if (h == -2 || m == -2) { err = "BVH link points to an unreachable node"; return false; }
        // Every link must point FORWARD in walk order: the device walk has no other termination argument (a
        // corrupted or hand-edited .sbvh with a backward miss link would spin a wave forever).
        auto forward = [&](float link) { const int32_t l = (int32_t)link; return l < 0 || new_index[l] > (int32_t)j; };
        if (!forward(n.hit) || !forward(n.miss)) { err = "BVH link points backward in walk order (the walk would not terminate)"; return false; }
        const bool leaf = (n.f0 >= 0 || n.f1 >= 0);         // ThreadedBvhNode::isLeaf, threaded_bvh.h:41-44
        if (!leaf) {
            if (h == kLinkEnd || ((uint32_t)h & kLinkOffsetMask) != (abs + 1) * kNodeBytes) { err = "inner node whose hit link is not the next node in walk order"; return false; }
            q0 = make_float4(n.boxmin[0], n.boxmin[1], n.boxmin[2], i2f(h & kLinkTypeMask));
            q1 = make_float4(n.boxmax[0], n.boxmax[1], n.boxmax[2], i2f(m));
            counts[0]++;
        }
        else if (n.f2 >= 0) {
            // nested tree (exid bit-field, threaded_bvh.h:29-37)
            if (!c.top) { err = "nested BVH reference inside a bottom-level list"; return false; }
            const int32_t objid = (int32_t)n.f0;
            if (objid < 0 || (uint32_t)objid >= c.n_objects) { err = "TLAS leaf object id out of range"; return false; }
            const uint32_t bits = f2u(n.f2);
            const int32_t exid = ATN_EXID_MAIN(bits);
            if (exid <= 0 || (uint32_t)exid >= c.n_lists || c.list_root_link[exid] == kLinkEnd) { err = "TLAS leaf references a missing BLAS list"; return false; }
            const atn_object_param& obj = c.objects[objid];
            int32_t w2l_row = -1;
            if (obj.mtx_id >= 0) {
                if ((uint32_t)obj.mtx_id + 1 >= c.n_matrices) { err = "object matrix index out of range"; return false; }
                w2l_row = 4 * (obj.mtx_id + 1);              // traverser reads GetMatrix(mtx_id + 1), :153
            }
            q0 = make_float4(i2f(objid), i2f(w2l_row), i2f(c.list_root_link[exid]), 0.0F);
            q1 = make_float4(i2f((int32_t)n.f3), i2f(h), i2f(m), 0.0F);
            counts[2]++;
        }
        else if (n.f1 >= 0) {
            if (!c.tris) { err = "triangle leaves in this list need a full scene upload"; return false; }
            const uint32_t tri = (uint32_t)n.f1;
            if (tri >= c.n_triangles) { err = "leaf triangle id out of range"; return false; }
            if (h != m) { err = "triangle leaf with hit != miss link"; return false; }
            const atn_triangle_param& t = c.tris[tri];
            for (int v = 0; v < 3; v++)
                if (t.idx[v] < 0 || (uint32_t)t.idx[v] >= c.n_vertices) { err = "triangle vertex index out of range"; return false; }
            const atn_vec4& a = c.vtx_pos[t.idx[0]];
            const atn_vec4& b = c.vtx_pos[t.idx[1]];
            const atn_vec4& cc = c.vtx_pos[t.idx[2]];
            // e1 = v1 - v0, e2 = v2 - v0: the same fp32 subtractions intersectTriangle performs
            // per test (math/intersect.h:61-62), hoisted to upload time.
            q0 = make_float4(a.x, a.y, a.z, i2f((int32_t)tri));
            q1 = make_float4(b.x - a.x, b.y - a.y, b.z - a.z, i2f(h));
            q2 = make_float4(cc.x - a.x, cc.y - a.y, cc.z - a.z, 0.0F);
            counts[1]++;
        }
        else {
            // leaf without triangle or nested tree (sphere instance): never tested on this path
            q0 = make_float4(0, 0, 0, i2f(kTagDead));
            q1 = make_float4(0, 0, 0, i2f(m));
        }
    }
    return true;
}

// Range checks of every id the kernels index with (a bad id is a device out-of-bounds read, not an error code).
// `s` may be null for a top-layer update (objects / matrices only).
inline bool validate_ranges(const atn_object_param* objs, uint32_t n_objs, uint32_t n_mtx, const atn_scene_desc* s, std::string& err)
{
    const uint32_t n_tris = s ? s->n_triangles : 0xffffffffu, n_lights = s ? s->n_lights : 0xffffffffu;
    for (uint32_t i = 0; i < n_objs; i++) {
        const atn_object_param& o = objs[i];
        if (o.type == ATN_OBJ_INSTANCE && (o.object_id < 0 || (uint32_t)o.object_id >= n_objs)) { err = "instance refers to an object id out of range"; return false; }
        if (o.mtx_id >= 0 && (uint32_t)o.mtx_id + 1 >= n_mtx) { err = "object matrix index out of range"; return false; }
        if (s && o.light_id >= 0 && (uint32_t)o.light_id >= n_lights) { err = "object light id out of range"; return false; }
        if (s && o.type == ATN_OBJ_POLYGONS && o.triangle_num > 0
            && (o.triangle_id < 0 || (uint64_t)o.triangle_id + (uint64_t)o.triangle_num > n_tris)) { err = "object triangle range out of range"; return false; }
    }
    if (!s) return true;
    for (uint32_t i = 0; i < s->n_triangles; i++) {
        const atn_triangle_param& t = s->triangles[i];
        if (t.mtrlid >= 0 && (uint32_t)t.mtrlid >= s->n_materials) { err = "triangle material id out of range"; return false; }
        for (int v = 0; v < 3; v++)
            if (t.idx[v] < 0 || (uint32_t)t.idx[v] >= s->n_vertices) { err = "triangle vertex index out of range"; return false; }
    }
    for (uint32_t i = 0; i < s->n_lights; i++) {
        const atn_light_param& l = s->lights[i];
        if (l.arealight_objid >= 0 && (uint32_t)l.arealight_objid >= s->n_objects) { err = "light refers to an object id out of range"; return false; }
    }
    return true;
}

// Node image = [BLAS list 1][BLAS list 2]...[top layer (list 0)]: the top layer comes last so that
// update_top_layer (≙ Renderer::updateBVH, "only for top layer") can replace it without moving the others.
inline bool build_host_image(HostSceneImage& img, const atn_scene_desc* s, std::string& err)
{
    if (!s || s->n_bvh_lists == 0 || !s->bvh_lists) { err = "scene has no BVH lists"; return false; }
    const uint32_t nl = s->n_bvh_lists;
    uint64_t total = 0;
    for (uint32_t k = 0; k < nl; k++) total += s->bvh_lists[k].count;
    if (total * kNodeBytes >= (1ull << 31)) { err = "too many BVH nodes for 31-bit byte-offset links"; return false; }
    img.nodes.assign((size_t)total * 3, make_float4(0, 0, 0, 0));
    img.list_root.assign(nl, 0);
    img.list_root_link.assign(nl, kLinkEnd);

    ListEmitCtx c;
    c.objects = s->objects; c.n_objects = s->n_objects; c.n_matrices = s->n_matrices;
    c.tris = s->triangles; c.vtx_pos = s->vtx_pos; c.n_triangles = s->n_triangles; c.n_vertices = s->n_vertices;
    c.n_lists = nl;
    std::string range_err;
    if (!validate_ranges(s->objects, s->n_objects, s->n_matrices, s, range_err)) { err = range_err; return false; }
    uint64_t counts[3] = { 0, 0, 0 };
    uint32_t base = 0;
    for (uint32_t k = 1; k <= nl; k++) {
        const uint32_t list = k % nl;       // 1, 2, ..., nl-1, 0
        c.top = (list == 0);
        c.list_root_link = img.list_root_link.data();
        img.list_root[list] = base;
        int32_t root = kLinkEnd;
        if (!emit_list(img.nodes.data() + 3 * (size_t)base, s->bvh_lists[list].nodes, s->bvh_lists[list].count, base, c, root, counts, err)) return false;
        img.list_root_link[list] = root;
        base += s->bvh_lists[list].count;
    }
    img.n_inner = counts[0]; img.n_tri_leaf = counts[1]; img.n_tlas_leaf = counts[2];

    // ---- plain copies
    img.tris.assign(s->triangles, s->triangles + s->n_triangles);
    img.vtx_pos.resize(s->n_vertices); img.vtx_nml.resize(s->n_vertices);
    for (uint32_t i = 0; i < s->n_vertices; i++) {
        img.vtx_pos[i] = make_float4(s->vtx_pos[i].x, s->vtx_pos[i].y, s->vtx_pos[i].z, s->vtx_pos[i].w);
        img.vtx_nml[i] = make_float4(s->vtx_nml[i].x, s->vtx_nml[i].y, s->vtx_nml[i].z, s->vtx_nml[i].w);
    }
    img.objects.assign(s->objects, s->objects + s->n_objects);
    img.matrices.resize((size_t)s->n_matrices * 4);
    for (uint32_t i = 0; i < s->n_matrices; i++)
        for (int r = 0; r < 4; r++)
            img.matrices[4 * (size_t)i + r] = make_float4(s->matrices[i].m[r][0], s->matrices[i].m[r][1], s->matrices[i].m[r][2], s->matrices[i].m[r][3]);
    std::vector<uint8_t> tex_has_alpha(s->n_textures, 0);
    for (uint32_t i = 0; i < s->n_textures; i++) {
        const atn_texture_desc& t = s->textures[i];
        const size_t n = (size_t)t.width * t.height;
        for (size_t j = 0; j < n; j++) if (t.texels[j].w < 1.0F) { tex_has_alpha[i] = 1; break; }
    }
    img.materials.resize(s->n_materials);
    for (uint32_t i = 0; i < s->n_materials; i++) {
        const atn_material_param& m = s->materials[i];
        DevMaterial& d = img.materials[i];
        d.baseColor = make_float4(m.baseColor.x, m.baseColor.y, m.baseColor.z, m.baseColor.w);
        d.type = m.type; d.attrib = (m.attrib & 0xFu) | (m.isIdealRefraction ? kAttrIdealRefraction : 0u); d.id = m.id;
        if (m.baseColor.w < 1.0F || (m.albedoMap >= 0 && (uint32_t)m.albedoMap < s->n_textures && tex_has_alpha[m.albedoMap])) d.attrib |= kAttrMaybeAlpha;
        d.albedoMap = m.albedoMap; d.normalMap = m.normalMap; d.roughnessMap = m.roughnessMap;
        const atn_standard_mtrl& st = m.u.standard;
        d.ior = st.ior; d.roughness = st.roughness; d.subsurface = st.subsurface; d.metallic = st.metallic;
        d.specular = st.specular; d.specularTint = st.specularTint; d.sheen = st.sheen; d.sheenTint = st.sheenTint;
        d.clearcoat = st.clearcoat; d.clearcoatGloss = st.clearcoatGloss;
    }
    {   // FillMaterial's fallback for mtrl_id < 0 (material_impl.h:253-259), stored at index n_materials
        DevMaterial d{};
        d.baseColor = make_float4(1, 1, 1, 1); d.type = ATN_MTRL_DIFFUSE; d.attrib = 0; d.id = 0;
        d.albedoMap = d.normalMap = d.roughnessMap = -1; d.ior = 1.0F; d.roughness = 0.5F;
        d.subsurface = d.metallic = d.specular = d.specularTint = 0.5F;
        d.sheen = d.sheenTint = d.clearcoat = d.clearcoatGloss = 0.5F;
        img.materials.push_back(d);
    }
    img.lights.assign(s->lights, s->lights + s->n_lights);
    img.textures.resize(s->n_textures);
    size_t ntex = 0;
    for (uint32_t i = 0; i < s->n_textures; i++) ntex += (size_t)s->textures[i].width * s->textures[i].height;
    img.texels.resize(ntex);
    size_t off = 0;
    for (uint32_t i = 0; i < s->n_textures; i++) {
        const atn_texture_desc& t = s->textures[i];
        img.textures[i].offset = (uint32_t)off; img.textures[i].width = t.width; img.textures[i].height = t.height; img.textures[i]._pad = 0;
        const size_t n = (size_t)t.width * t.height;
        for (size_t j = 0; j < n; j++) img.texels[off + j] = make_float4(t.texels[j].x, t.texels[j].y, t.texels[j].z, t.texels[j].w);
        off += n;
    }

    DevScene& p = img.params;
    p.root_link = img.list_root_link[0];
    p.n_lights = (int32_t)s->n_lights; p.n_textures = (int32_t)s->n_textures; p.n_materials = (int32_t)s->n_materials;
    p.bvh_hit_min = s->config.bvh_hit_min;
    p.bg_color[0] = s->config.bg.bg_color[0]; p.bg_color[1] = s->config.bg.bg_color[1]; p.bg_color[2] = s->config.bg.bg_color[2];
    p.envmap_tex_idx = s->config.bg.envmap_tex_idx;
    p.avgIllum = s->config.bg.avgIllum;
    p.multiplyer = s->config.bg.multiplyer;
    p.enable_env_map = s->config.bg.enable_env_map;
    p.any_alpha = 0;
    for (const DevMaterial& dm : img.materials) if (dm.attrib & kAttrMaybeAlpha) p.any_alpha = 1;
    // ImageBasedLight::sample's scene_radius (light/ibl.h:106-111; aabb::IsValid / getCenter /
    // ComputeDistanceToCoverBoundingSphere, math/aabb.h:176-180,231-234,346-362), evaluated once on the host.
    {
        const float* mn = s->scene_bbox_min; const float* mx = s->scene_bbox_max;
        float radius = 10000.0F;
        const bool valid = !((mn[0] >= mx[0]) || (mn[1] >= mx[1]) || (mn[2] >= mx[2]));
        if (valid) {
            const float cx = (mn[0] + mx[0]) * 0.5F, cy = (mn[1] + mx[1]) * 0.5F, cz = (mn[2] + mx[2]) * 0.5F;
            const float dx = mx[0] - cx, dy = mx[1] - cy, dz = mx[2] - cz;
            const float r = std::sqrt((dx * dx + dy * dy) + dz * dz);
            const float theta = (3.14159265358979323846F * (30.0F) / 180.0F);
            radius = r / std::tan(theta / 2);
        }
        p.ibl_scene_radius = radius;
    }
    return true;
}

} // namespace atn
