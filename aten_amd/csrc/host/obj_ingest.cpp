// Host-only scene ingestion: what libatenscene does between a file and aten::context,
// for callers that do not link libaten(scene).  Native twin of aten_amd/scene/obj_loader.py + SceneBuilder.load_obj,
// which stay as the cross-check (tests/test_scene_ingest_cpu.py compares both byte for byte).
//
//   atns_obj_*           Wavefront OBJ / MTL -> one vertex per face corner, triangles, material groups, objects, following
//                        the REGISTRATION rules of aten::ObjLoader::Load (src/libatenscene/ObjLoader.cpp:95-461):
//                          * shapes in file order, a shape per `o` / `g` statement, shapes without faces dropped (tinyobj);
//                          * one ctxt.AddVertex per face corner, never de-duplicated (:140-163);
//                          * uv.z flags: no normal -> 1; normal present -> need_compute_normal_on_the_fly ? 1 : 0;
//                            NaN normal -> (0, 1, 0); no texcoord -> uv = 0, uv.z = -1 (:176-215);
//                          * a new TriangleGroupMesh whenever the material id changes inside a shape (:333-372);
//                          * needNormal of a triangle = any corner flag == 1 or on-the-fly (:387-393);
//                          * will_register_shape_as_separate_obj: one PolygonObject per shape; else one object for the
//                            whole file, except that a mesh with an EMISSIVE material becomes its own object (:408-451).
//                        tinyobjloader (3rdparty/tinyobjloader, pinned commit unknown) is absent from the reference
//                        snapshot: polygons are triangulated as a plain fan (0,1,2), (0,2,3), ... -- stated with every fixture.
//   atns_mtrlxml_*       the XML the reference's MaterialLoader reads (src/libatenscene/MaterialLoader.cpp:82-218):
//                        <root><material><name/><type/><baseColor>r g b</baseColor><ior>..</ior><albedoMap>file</albedoMap>..
//                        Parameter names and kinds are MaterialLoader's table (:82-99); unknown elements are skipped like there;
//                        a second material with a name already seen is dropped (:176-183); a missing type is "Diffuse" (:206-209).
//                        (tinyxml2 is absent too: a reader for exactly this element-with-text subset, comments and
//                        the five predefined entities.)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../../include/aten_amd_scene.h"

namespace {

struct Mtl { std::string name, map_kd, map_bump; float kd[3] = { 1, 1, 1 }, ke[3] = { 0, 0, 0 }; };
struct Corner { int32_t v, vt, vn; };
struct Shape { std::string name; std::vector<Corner> corners; std::vector<int32_t> mtl; };

struct ObjFile {
    std::vector<float> pos, tex, nml;      // 3 / 2 / 3 floats per entry
    std::vector<Shape> shapes;
    std::vector<Mtl> mtls;
    // after atns_obj_register
    std::vector<atn_vec4> vtx_pos, vtx_nml;
    std::vector<atns_obj_triangle> tris;
    std::vector<atns_obj_mesh> meshes;
    std::vector<atns_obj_object> objects;
    std::string error;
};

static std::vector<std::string> split_ws(const std::string& line)
{
    std::vector<std::string> t;
    size_t i = 0;
    while (i < line.size()) {
        while (i < line.size() && isspace((unsigned char)line[i])) i++;
        size_t j = i;
        while (j < line.size() && !isspace((unsigned char)line[j])) j++;
        if (j > i) t.push_back(line.substr(i, j - i));
        i = j;
    }
    return t;
}
static std::string join_from(const std::vector<std::string>& t, size_t k)
{
    std::string s;
    for (size_t i = k; i < t.size(); i++) { if (i > k) s += ' '; s += t[i]; }
    return s;
}
static float to_f(const std::string& s) { return (float)std::strtod(s.c_str(), nullptr); }     // Python float(): double, then float32
static int32_t fix_index(int32_t i, size_t n) { return i > 0 ? i - 1 : (int32_t)n + i; }       // 1-based, negative = relative

static void load_mtl(const std::string& path, std::vector<Mtl>& out)
{
    std::ifstream f(path);
    if (!f) return;
    std::string line;
    Mtl* cur = nullptr;
    while (std::getline(f, line)) {
        const auto t = split_ws(line);
        if (t.empty() || t[0][0] == '#') continue;
        const std::string& k = t[0];
        if (k == "newmtl") { out.emplace_back(); cur = &out.back(); cur->name = join_from(t, 1); }
        else if (!cur) continue;
        else if (k == "Kd" && t.size() >= 4) { for (int c = 0; c < 3; c++) cur->kd[c] = to_f(t[1 + c]); }
        else if (k == "Ke" && t.size() >= 4) { for (int c = 0; c < 3; c++) cur->ke[c] = to_f(t[1 + c]); }
        else if (k == "map_Kd" && t.size() >= 2) cur->map_kd = t.back();
        else if ((k == "map_bump" || k == "map_Bump" || k == "bump") && t.size() >= 2) cur->map_bump = t.back();
    }
}

static bool parse_obj(const std::string& path, ObjFile& o)
{
    std::ifstream f(path);
    if (!f) { o.error = "cannot open " + path; return false; }
    std::string base;
    { const size_t s = path.find_last_of("/\\"); base = s == std::string::npos ? std::string() : path.substr(0, s + 1); }
    std::string line;
    Shape* cur = nullptr;
    int32_t cur_mtl = -1;
    while (std::getline(f, line)) {
        const auto t = split_ws(line);
        if (t.empty()) continue;
        const std::string& k = t[0];
        if (k == "v" && t.size() >= 4) { for (int c = 0; c < 3; c++) o.pos.push_back(to_f(t[1 + c])); }
        else if (k == "vt" && t.size() >= 2) { o.tex.push_back(to_f(t[1])); o.tex.push_back(t.size() > 2 ? to_f(t[2]) : 0.0F); }
        else if (k == "vn" && t.size() >= 4) { for (int c = 0; c < 3; c++) o.nml.push_back(to_f(t[1 + c])); }
        else if (k == "o" || k == "g") { o.shapes.emplace_back(); cur = &o.shapes.back(); cur->name = join_from(t, 1); }
        else if (k == "mtllib" && t.size() >= 2) load_mtl(base + t[1], o.mtls);     // a second library APPENDS (tinyobj's MaterialFileReader): ids already handed out stay valid
        else if (k == "usemtl") {
            const std::string name = join_from(t, 1);
            cur_mtl = -1;
            for (size_t i = 0; i < o.mtls.size(); i++) if (o.mtls[i].name == name) cur_mtl = (int32_t)i;      // last of equal names, like a dict
        }
        else if (k == "f") {
            if (!cur) { o.shapes.emplace_back(); cur = &o.shapes.back(); }
            std::vector<Corner> cs;
            for (size_t w = 1; w < t.size(); w++) {
                Corner c{ 0, -1, -1 };
                const std::string& s = t[w];
                const size_t a = s.find('/');
                const size_t b = a == std::string::npos ? std::string::npos : s.find('/', a + 1);
                c.v = fix_index(std::atoi(s.substr(0, a).c_str()), o.pos.size() / 3);
                if (a != std::string::npos) {
                    const std::string m = s.substr(a + 1, b == std::string::npos ? std::string::npos : b - a - 1);
                    if (!m.empty()) c.vt = fix_index(std::atoi(m.c_str()), o.tex.size() / 2);
                    if (b != std::string::npos && b + 1 < s.size()) c.vn = fix_index(std::atoi(s.substr(b + 1).c_str()), o.nml.size() / 3);
                }
                cs.push_back(c);
            }
            for (size_t j = 1; j + 1 < cs.size(); j++) {       // plain fan
                cur->corners.push_back(cs[0]); cur->corners.push_back(cs[j]); cur->corners.push_back(cs[j + 1]);
                cur->mtl.push_back(cur_mtl);
            }
        }
    }
    std::vector<Shape> kept;
    for (auto& s : o.shapes) if (!s.mtl.empty()) kept.push_back(std::move(s));      // tinyobj drops shapes without faces
    o.shapes.swap(kept);
    for (const auto& s : o.shapes)
        for (const auto& c : s.corners) {
            if (c.v < 0 || (size_t)c.v >= o.pos.size() / 3 || (c.vt >= 0 && (size_t)c.vt >= o.tex.size() / 2) || c.vt < -1
                || (c.vn >= 0 && (size_t)c.vn >= o.nml.size() / 3) || c.vn < -1) { o.error = "face index out of range in " + path; return false; }
        }
    for (const auto& s : o.shapes)
        for (const int32_t m : s.mtl)
            if (m < -1 || (m >= 0 && (size_t)m >= o.mtls.size())) { o.error = "face material id out of range in " + path; return false; }
    return true;
}

// ---- MaterialLoader's XML subset --------------------------------------------------------------------------------------
struct XmlMaterial { std::string name, type; std::vector<std::pair<std::string, std::string>> params; };
struct XmlFile { std::vector<XmlMaterial> mats; std::string error; };

static std::string xml_unescape(const std::string& s)
{
    std::string o;
    for (size_t i = 0; i < s.size(); i++) {
        if (s[i] == '&') {
            const char* ent[5] = { "&amp;", "&lt;", "&gt;", "&quot;", "&apos;" };
            const char rep[5] = { '&', '<', '>', '"', '\'' };
            bool done = false;
            for (int e = 0; e < 5 && !done; e++) {
                const size_t n = std::strlen(ent[e]);
                if (s.compare(i, n, ent[e]) == 0) { o += rep[e]; i += n - 1; done = true; }
            }
            if (!done) o += s[i];
        }
        else o += s[i];
    }
    return o;
}
static std::string trim(const std::string& s)
{
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) a++;
    while (b > a && isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}
// pull parser over "<tag ...>" / "</tag>" / text; comments, <?...?> and <!...> skipped
struct XmlTok { int kind; std::string text; };       // 0 open, 1 close, 2 text, 3 self-closing
static bool xml_tokens(const std::string& s, std::vector<XmlTok>& out, std::string& err)
{
    size_t i = 0;
    while (i < s.size()) {
        if (s[i] == '<') {
            if (s.compare(i, 4, "<!--") == 0) { const size_t e = s.find("-->", i + 4); if (e == std::string::npos) { err = "unterminated comment"; return false; } i = e + 3; continue; }
            if (s.compare(i, 2, "<?") == 0) { const size_t e = s.find("?>", i + 2); if (e == std::string::npos) { err = "unterminated declaration"; return false; } i = e + 2; continue; }
            if (s.compare(i, 2, "<!") == 0) { const size_t e = s.find('>', i + 2); if (e == std::string::npos) { err = "unterminated <!"; return false; } i = e + 1; continue; }
            const size_t e = s.find('>', i + 1);
            if (e == std::string::npos) { err = "unterminated tag"; return false; }
            std::string body = s.substr(i + 1, e - i - 1);
            if (!body.empty() && body[0] == '/') out.push_back({ 1, trim(body.substr(1)) });
            else {
                const bool self = !body.empty() && body.back() == '/';
                if (self) body.pop_back();
                size_t k = 0;
                while (k < body.size() && !isspace((unsigned char)body[k])) k++;
                out.push_back({ self ? 3 : 0, body.substr(0, k) });      // attributes are not part of MaterialLoader's format
            }
            i = e + 1;
        }
        else {
            const size_t e = s.find('<', i);
            const std::string txt = s.substr(i, e == std::string::npos ? std::string::npos : e - i);
            if (!trim(txt).empty()) out.push_back({ 2, xml_unescape(trim(txt)) });
            i = e == std::string::npos ? s.size() : e;
        }
    }
    return true;
}
static bool parse_mtrl_xml(const std::string& text, XmlFile& x)
{
    std::vector<XmlTok> tk;
    if (!xml_tokens(text, tk, x.error)) return false;
    size_t i = 0;
    while (i < tk.size() && !(tk[i].kind == 0 && tk[i].text == "root")) i++;
    if (i == tk.size()) { x.error = "no <root> element"; return false; }       // MaterialLoader::load returns false (:153-160)
    int depth = 0;      // inside root
    XmlMaterial cur; bool in_mat = false, dup = false;
    for (i = i + 1; i < tk.size(); i++) {
        const XmlTok& t = tk[i];
        if (!in_mat) {
            if (t.kind == 0 && t.text == "material" && depth == 0) { in_mat = true; dup = false; cur = XmlMaterial(); }
            else if (t.kind == 0) depth++;
            else if (t.kind == 1) { if (depth == 0) break; depth--; }      // </root>
            continue;
        }
        if (t.kind == 1 && t.text == "material") {
            if (!dup) { if (cur.type.empty()) cur.type = "Diffuse"; x.mats.push_back(cur); }
            in_mat = false;
            continue;
        }
        if (t.kind == 3) continue;      // <x/>: an element without text; tinyxml2's GetText() would be null -- nothing to read
        if (t.kind == 0) {
            // <name>text</name>
            std::string val;
            size_t j = i + 1;
            if (j < tk.size() && tk[j].kind == 2) { val = tk[j].text; j++; }
            if (j >= tk.size() || tk[j].kind != 1 || tk[j].text != t.text) { x.error = "element <" + t.text + "> is not plain text"; return false; }
            i = j;
            if (dup) continue;          // (:176-183: the loop over the duplicate's children is broken off)
            if (t.text == "name") {
                cur.name = val;
                for (const auto& m : x.mats) if (m.name == val) dup = true;
            }
            else if (t.text == "type") cur.type = val;
            else cur.params.emplace_back(t.text, val);
        }
    }
    return true;
}

} // namespace

struct atns_obj { ObjFile f; };
struct atns_mtrlxml { XmlFile x; std::vector<std::string> keep; };

extern "C" {

// Nothing may throw across the C boundary (a std::bad_alloc on a huge or hostile file would end the caller's process):
// every entry point that runs container code returns -2 instead.
int atns_obj_open(const char* path, atns_obj** out)
{
    if (!path || !out) return -1;
    atns_obj* h = new (std::nothrow) atns_obj();
    if (!h) return -2;
    try {
        if (!parse_obj(path, h->f)) { std::fprintf(stderr, "atns_obj_open: %s\n", h->f.error.c_str()); delete h; return -3; }
    }
    catch (...) { delete h; return -2; }
    *out = h;
    return 0;
}
void atns_obj_close(atns_obj* h) { delete h; }
uint32_t atns_obj_material_count(const atns_obj* h) { return h ? (uint32_t)h->f.mtls.size() : 0; }
int atns_obj_material(const atns_obj* h, uint32_t i, atns_obj_material_info* out)
{
    if (!h || !out || i >= h->f.mtls.size()) return -1;
    const Mtl& m = h->f.mtls[i];
    out->name = m.name.c_str(); out->diffuse_texname = m.map_kd.c_str(); out->bump_texname = m.map_bump.c_str();
    for (int c = 0; c < 3; c++) { out->diffuse[c] = m.kd[c]; out->emission[c] = m.ke[c]; }
    return 0;
}

static int obj_register(atns_obj* h, uint32_t first_vertex, uint32_t first_mesh_id, int32_t separate_objs, int32_t normal_on_the_fly,
                        const uint8_t* mtl_is_emissive, uint8_t default_is_emissive);

int atns_obj_register(atns_obj* h, uint32_t first_vertex, uint32_t first_mesh_id, int32_t separate_objs, int32_t normal_on_the_fly,
                      const uint8_t* mtl_is_emissive, uint32_t n_mtl_is_emissive, uint8_t default_is_emissive)
{
    if (!h) return -1;
    // mtl_is_emissive is indexed by OBJ material id: it must cover every material of the file (parse_obj checked the ids)
    if (mtl_is_emissive && n_mtl_is_emissive < h->f.mtls.size()) return -4;
    try { return obj_register(h, first_vertex, first_mesh_id, separate_objs, normal_on_the_fly, mtl_is_emissive, default_is_emissive); }
    catch (...) { return -2; }
}

static int obj_register(atns_obj* h, uint32_t first_vertex, uint32_t first_mesh_id, int32_t separate_objs, int32_t normal_on_the_fly,
                        const uint8_t* mtl_is_emissive, uint8_t default_is_emissive)
{
    ObjFile& o = h->f;
    o.vtx_pos.clear(); o.vtx_nml.clear(); o.tris.clear(); o.meshes.clear(); o.objects.clear();
    auto emissive = [&](int32_t mtl) { return mtl < 0 ? default_is_emissive != 0 : (mtl_is_emissive && mtl_is_emissive[mtl] != 0); };
    uint32_t mesh_id = first_mesh_id;
    int32_t cur_obj = -1;
    int32_t n_returned = 0;
    auto new_obj = [&](int32_t shape) {
        atns_obj_object ob{}; ob.first_mesh = 0; ob.n_meshes = 0; ob.shape = shape; ob.is_emissive_split = 0; ob.return_order = -1;
        o.objects.push_back(ob);
        return (int32_t)o.objects.size() - 1;
    };
    // ObjLoader.cpp:262-283 (a material change inside a shape) and :425-441 (end of a shape, one object per file): a group whose
    // material is Emissive becomes its own object, anything else joins the current object (created on demand, named by the shape)
    auto register_mesh = [&](uint32_t mesh_index, int32_t shape) {
        atns_obj_mesh& m = o.meshes[mesh_index];
        if (emissive(m.mtl)) {
            const int32_t e = new_obj(shape);
            o.objects[e].is_emissive_split = 1;
            o.objects[e].return_order = n_returned++;
            m.object = e;
            return;
        }
        if (cur_obj < 0) cur_obj = new_obj(shape);
        if (o.objects[cur_obj].shape < 0) o.objects[cur_obj].shape = shape;         // create_obj_functor: name it if it has none
        m.object = cur_obj;
    };
    for (size_t si = 0; si < o.shapes.size(); si++) {
        const Shape& sh = o.shapes[si];
        const uint32_t base_v = first_vertex + (uint32_t)o.vtx_pos.size();
        std::vector<float> flags(sh.corners.size());
        for (size_t c = 0; c < sh.corners.size(); c++) {
            const Corner& k = sh.corners[c];
            atn_vec4 p{ o.pos[3 * k.v], o.pos[3 * k.v + 1], o.pos[3 * k.v + 2], 0.0F }, n{ 0.0F, 1.0F, 0.0F, 0.0F };
            float uvz;
            if (k.vn < 0) uvz = 1.0F;                                   // the reference leaves nml unset; needNormal is set
            else { n.x = o.nml[3 * k.vn]; n.y = o.nml[3 * k.vn + 1]; n.z = o.nml[3 * k.vn + 2]; uvz = normal_on_the_fly ? 1.0F : 0.0F; }
            if (std::isnan(n.x) || std::isnan(n.y) || std::isnan(n.z)) { n.x = 0.0F; n.y = 1.0F; n.z = 0.0F; }
            if (k.vt >= 0) { p.w = o.tex[2 * k.vt]; n.w = o.tex[2 * k.vt + 1]; }
            else { p.w = 0.0F; n.w = 0.0F; uvz = -1.0F; }
            o.vtx_pos.push_back(p); o.vtx_nml.push_back(n);
            flags[c] = uvz;
        }
        int32_t mesh = -1, prev = 0;
        for (size_t i = 0; i < sh.mtl.size(); i++) {
            const int32_t mid = sh.mtl[i];
            if (mesh < 0 || prev != mid) {
                if (mesh >= 0) register_mesh((uint32_t)mesh, (int32_t)si);          // in BOTH modes (:262-283)
                atns_obj_mesh m{}; m.mtl = mid; m.mesh_id = mesh_id++; m.first_triangle = (uint32_t)o.tris.size(); m.n_triangles = 0; m.object = -1; m.shape = (int32_t)si;
                o.meshes.push_back(m);
                mesh = (int32_t)o.meshes.size() - 1;
                prev = mid;
            }
            atns_obj_triangle t{};
            t.idx[0] = base_v + 3 * (uint32_t)i; t.idx[1] = t.idx[0] + 1; t.idx[2] = t.idx[0] + 2;
            t.need_normal = (flags[3 * i] == 1.0F || flags[3 * i + 1] == 1.0F || flags[3 * i + 2] == 1.0F || normal_on_the_fly) ? 1 : 0;
            t.mesh = mesh;
            o.tris.push_back(t);
            o.meshes[mesh].n_triangles++;
        }
        if (separate_objs) {
            // the shape's LAST group goes into the shape's object whatever its material; the object is returned, and the next
            // shape's object is created right away (ObjLoader.cpp:408-424) -- before any emissive split of the next shape
            if (cur_obj < 0) cur_obj = new_obj((int32_t)si);
            o.objects[cur_obj].shape = (int32_t)si;
            o.meshes[mesh].object = cur_obj;
            o.objects[cur_obj].return_order = n_returned++;
            cur_obj = si + 1 < o.shapes.size() ? new_obj(-1) : -1;
        }
        else register_mesh((uint32_t)mesh, (int32_t)si);
    }
    if (!separate_objs && cur_obj >= 0) o.objects[cur_obj].return_order = n_returned++;       // the file's object is returned last (:444-451)
    // object order = creation order; within an object, meshes in creation order
    for (auto& ob : o.objects) { ob.n_meshes = 0; }
    for (const auto& m : o.meshes) if (m.object >= 0) o.objects[m.object].n_meshes++;
    return 0;
}

uint32_t atns_obj_vertex_count(const atns_obj* h) { return h ? (uint32_t)h->f.vtx_pos.size() : 0; }
uint32_t atns_obj_triangle_count(const atns_obj* h) { return h ? (uint32_t)h->f.tris.size() : 0; }
uint32_t atns_obj_mesh_count(const atns_obj* h) { return h ? (uint32_t)h->f.meshes.size() : 0; }
uint32_t atns_obj_object_count(const atns_obj* h) { return h ? (uint32_t)h->f.objects.size() : 0; }
uint32_t atns_obj_shape_count(const atns_obj* h) { return h ? (uint32_t)h->f.shapes.size() : 0; }
const char* atns_obj_shape_name(const atns_obj* h, uint32_t i) { return (h && i < h->f.shapes.size()) ? h->f.shapes[i].name.c_str() : ""; }
int atns_obj_copy(const atns_obj* h, atn_vec4* vtx_pos, atn_vec4* vtx_nml, uint32_t cap_vertices, atns_obj_triangle* tris, uint32_t cap_triangles,
                  atns_obj_mesh* meshes, uint32_t cap_meshes, atns_obj_object* objects, uint32_t cap_objects)
{
    if (!h) return -1;
    const ObjFile& o = h->f;
    // a destination smaller than what atns_obj_register produced is an error, not an overrun
    if (((vtx_pos || vtx_nml) && cap_vertices < o.vtx_pos.size()) || (tris && cap_triangles < o.tris.size())
        || (meshes && cap_meshes < o.meshes.size()) || (objects && cap_objects < o.objects.size())) return -4;
    if (vtx_pos) std::memcpy(vtx_pos, o.vtx_pos.data(), o.vtx_pos.size() * sizeof(atn_vec4));
    if (vtx_nml) std::memcpy(vtx_nml, o.vtx_nml.data(), o.vtx_nml.size() * sizeof(atn_vec4));
    if (tris) std::memcpy(tris, o.tris.data(), o.tris.size() * sizeof(atns_obj_triangle));
    if (meshes) std::memcpy(meshes, o.meshes.data(), o.meshes.size() * sizeof(atns_obj_mesh));
    if (objects) std::memcpy(objects, o.objects.data(), o.objects.size() * sizeof(atns_obj_object));
    return 0;
}

int atns_mtrlxml_open(const char* path, atns_mtrlxml** out)
{
    if (!path || !out) return -1;
    atns_mtrlxml* h = nullptr;
    try {
        std::ifstream f(path, std::ios::binary);
        if (!f) return -3;                                   // MaterialLoader::load: "failed to load" (:145-151)
        std::stringstream ss; ss << f.rdbuf();
        h = new (std::nothrow) atns_mtrlxml();
        if (!h) return -2;
        if (!parse_mtrl_xml(ss.str(), h->x)) { std::fprintf(stderr, "atns_mtrlxml_open: %s\n", h->x.error.c_str()); delete h; return -4; }
    }
    catch (...) { delete h; return -2; }
    *out = h;
    return 0;
}
void atns_mtrlxml_close(atns_mtrlxml* h) { delete h; }
uint32_t atns_mtrlxml_count(const atns_mtrlxml* h) { return h ? (uint32_t)h->x.mats.size() : 0; }
const char* atns_mtrlxml_name(const atns_mtrlxml* h, uint32_t i) { return (h && i < h->x.mats.size()) ? h->x.mats[i].name.c_str() : ""; }
const char* atns_mtrlxml_type(const atns_mtrlxml* h, uint32_t i) { return (h && i < h->x.mats.size()) ? h->x.mats[i].type.c_str() : ""; }
uint32_t atns_mtrlxml_param_count(const atns_mtrlxml* h, uint32_t i) { return (h && i < h->x.mats.size()) ? (uint32_t)h->x.mats[i].params.size() : 0; }

// MaterialLoader's parameter table (MaterialLoader.cpp:82-99): 0 = vec3, 1 = texture file name, 2 = float, -1 = not a parameter
static int param_kind(const std::string& n)
{
    if (n == "baseColor") return 0;
    if (n == "albedoMap" || n == "normalMap" || n == "roughnessMap") return 1;
    const char* fl[] = { "ior", "roughness", "shininess", "subsurface", "metallic", "specular", "specularTint", "anisotropic", "sheen",
                         "sheenTint", "clearcoat", "clearcoatGloss" };
    for (const char* f : fl) if (n == f) return 2;
    return -1;
}
int atns_mtrlxml_param(const atns_mtrlxml* h, uint32_t i, uint32_t k, atns_mtrlxml_param_info* out)
{
    if (!h || !out || i >= h->x.mats.size() || k >= h->x.mats[i].params.size()) return -1;
    const auto& p = h->x.mats[i].params[k];
    out->name = p.first.c_str(); out->text = p.second.c_str();
    out->kind = param_kind(p.first);
    out->value[0] = out->value[1] = out->value[2] = 0.0F;
    if (out->kind == 0) {
        // getValue<vec3> (:23-43): split on ' ', atof each, at most three
        try {
            const auto t = split_ws(p.second);
            for (size_t c = 0; c < t.size() && c < 3; c++) out->value[c] = (float)std::atof(t[c].c_str());
        }
        catch (...) { return -2; }
    }
    else if (out->kind == 2) out->value[0] = (float)std::strtod(p.second.c_str(), nullptr);      // XMLElement::DoubleText -> float (:45-51)
    return 0;
}

} // extern "C"
