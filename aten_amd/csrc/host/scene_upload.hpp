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
    std::vector<uint32_t> list_root;    // absolute index of each list's root
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

inline bool build_host_image(HostSceneImage& img, const atn_scene_desc* s, std::string& err)
{
    if (!s || s->n_bvh_lists == 0 || !s->bvh_lists) { err = "scene has no BVH lists"; return false; }
    const uint32_t nl = s->n_bvh_lists;

    // ---- pass 1: walk orders and absolute offsets
    std::vector<std::vector<int32_t>> new_index(nl);
    std::vector<std::vector<uint32_t>> order(nl);
    img.list_root.assign(nl, 0);
    uint32_t total = 0;
    for (uint32_t k = 0; k < nl; k++) {
        if (!walk_order(s->bvh_lists[k].nodes, s->bvh_lists[k].count, new_index[k], order[k], err)) return false;
        img.list_root[k] = total;
        total += (uint32_t)order[k].size();
    }
    if (total >= (1u << 24)) { err = "more than 2^24 BVH nodes: float-encoded links would lose precision"; return false; }
    img.nodes.assign((size_t)total * 3, make_float4(0, 0, 0, 0));

    auto type_bits = [&](uint32_t k, uint32_t old_idx) -> int32_t {
        const atn_bvh_node& n = s->bvh_lists[k].nodes[old_idx];
        if (!(n.f0 >= 0 || n.f1 >= 0)) return 0;        // inner
        if (n.f2 >= 0) return kLinkTlasBit;             // nested tree
        if (n.f1 >= 0) return kLinkLeafBit;             // triangle
        return 0;                                       // dead leaf: handled on the inner path by its tag
    };
    // typed link of a (list, float link): kLinkEnd, or absolute index | leaf bit; -2 = invalid
    auto remap = [&](uint32_t k, float link) -> int32_t {
        const int32_t l = (int32_t)link;
        if (l < 0) return kLinkEnd;
        if ((uint32_t)l >= s->bvh_lists[k].count || new_index[k][l] < 0) return -2;
        const int32_t abs = (int32_t)img.list_root[k] + new_index[k][l];
        return (int32_t)((uint32_t)abs * kNodeBytes) | type_bits(k, (uint32_t)l);
    };
    if ((uint64_t)total * kNodeBytes >= (1ull << 31)) { err = "too many BVH nodes for 31-bit byte-offset links"; return false; }

    // ---- pass 2: emit device records
    for (uint32_t k = 0; k < nl; k++) {
        const atn_bvh_node* src = s->bvh_lists[k].nodes;
        for (uint32_t j = 0; j < order[k].size(); j++) {
            const atn_bvh_node& n = src[order[k][j]];
            const uint32_t abs = img.list_root[k] + j;
            float4& q0 = img.nodes[3 * (size_t)abs + 0];
            float4& q1 = img.nodes[3 * (size_t)abs + 1];
            float4& q2 = img.nodes[3 * (size_t)abs + 2];
            const int32_t h = remap(k, n.hit), m = remap(k, n.miss);
            if (h == -2 || m == -2) { err = "BVH link points to an unreachable node"; return false; }
            const bool leaf = (n.f0 >= 0 || n.f1 >= 0);         // ThreadedBvhNode::isLeaf, threaded_bvh.h:41-44
            if (!leaf) {
                if (h == kLinkEnd || ((uint32_t)h & kLinkOffsetMask) != (abs + 1) * kNodeBytes) { err = "inner node whose hit link is not the next node in walk order"; return false; }
                q0 = make_float4(n.boxmin[0], n.boxmin[1], n.boxmin[2], i2f(h & kLinkTypeMask));
                q1 = make_float4(n.boxmax[0], n.boxmax[1], n.boxmax[2], i2f(m));
                img.n_inner++;
            }
            else if (n.f2 >= 0) {
                // nested tree (exid bit-field, threaded_bvh.h:29-37)
                if (k != 0) { err = "nested BVH reference inside a bottom-level list"; return false; }
                const int32_t objid = (int32_t)n.f0;
                if (objid < 0 || (uint32_t)objid >= s->n_objects) { err = "TLAS leaf object id out of range"; return false; }
                const uint32_t bits = f2u(n.f2);
                const int32_t exid = ATN_EXID_MAIN(bits);
                if (exid <= 0 || (uint32_t)exid >= nl || order[exid].empty()) { err = "TLAS leaf references a missing BLAS list"; return false; }
                const atn_object_param& obj = s->objects[objid];
                int32_t w2l_row = -1;
                if (obj.mtx_id >= 0) {
                    if ((uint32_t)obj.mtx_id + 1 >= s->n_matrices) { err = "object matrix index out of range"; return false; }
                    w2l_row = 4 * (obj.mtx_id + 1);              // traverser reads GetMatrix(mtx_id + 1), :153
                }
                const int32_t root = remap((uint32_t)exid, 0.0F);
                q0 = make_float4(i2f(objid), i2f(w2l_row), i2f(root), 0.0F);
                q1 = make_float4(i2f((int32_t)n.f3), i2f(h), i2f(m), 0.0F);
                img.n_tlas_leaf++;
            }
            else if (n.f1 >= 0) {
                const uint32_t tri = (uint32_t)n.f1;
                if (tri >= s->n_triangles) { err = "leaf triangle id out of range"; return false; }
                if (h != m) { err = "triangle leaf with hit != miss link"; return false; }
                const atn_triangle_param& t = s->triangles[tri];
                const atn_vec4& a = s->vtx_pos[t.idx[0]];
                const atn_vec4& b = s->vtx_pos[t.idx[1]];
                const atn_vec4& c = s->vtx_pos[t.idx[2]];
                // e1 = v1 - v0, e2 = v2 - v0: the same fp32 subtractions intersectTriangle performs
                // per test (math/intersect.h:61-62), hoisted to upload time.
                q0 = make_float4(a.x, a.y, a.z, i2f((int32_t)tri));
                q1 = make_float4(b.x - a.x, b.y - a.y, b.z - a.z, i2f(h));
                q2 = make_float4(c.x - a.x, c.y - a.y, c.z - a.z, 0.0F);
                img.n_tri_leaf++;
            }
            else {
                // leaf without triangle or nested tree (sphere instance): never tested on this path
                q0 = make_float4(0, 0, 0, i2f(kTagDead));
                q1 = make_float4(0, 0, 0, i2f(m));
            }
        }
    }

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
    img.materials.resize(s->n_materials);
    for (uint32_t i = 0; i < s->n_materials; i++) {
        const atn_material_param& m = s->materials[i];
        DevMaterial& d = img.materials[i];
        d.baseColor = make_float4(m.baseColor.x, m.baseColor.y, m.baseColor.z, m.baseColor.w);
        d.type = m.type; d.attrib = m.attrib; d.id = m.id;
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
    p.root_link = remap(0, 0.0F);
    p.n_lights = (int32_t)s->n_lights; p.n_textures = (int32_t)s->n_textures; p.n_materials = (int32_t)s->n_materials;
    p.bvh_hit_min = s->config.bvh_hit_min;
    p.bg_color[0] = s->config.bg.bg_color[0]; p.bg_color[1] = s->config.bg.bg_color[1]; p.bg_color[2] = s->config.bg.bg_color[2];
    p.envmap_tex_idx = s->config.bg.envmap_tex_idx;
    p.avgIllum = s->config.bg.avgIllum;
    p.multiplyer = s->config.bg.multiplyer;
    p.enable_env_map = s->config.bg.enable_env_map;
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
