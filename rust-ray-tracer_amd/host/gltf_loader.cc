// gltf_loader.cc -- glTF -> rrt_scene_desc, the host-side front end in front of the kernels.
//
// COMPAT mode reproduces the output contract of the reference's `Scene::from_gltf_file`
// (src/gltf.rs:314-351) including its quirks; EXTENDED mode is a documented superset (external .bin
// buffers, accessor byteOffset / byteStride, u8/u32 indices, every primitive of a mesh, node transforms
// baked into the vertices, uniform scale + offset).
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <new>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/rrt_ffi.h"
#include "../csrc/rrt_internal.h"
#include "json_min.h"

using rrtjson::Value;

namespace {

struct OwnedScene {
    rrt_scene_desc desc;  // must stay the first member: rrt_scene_free casts back
    std::vector<rrt_material> materials;
    std::vector<rrt_triangle> triangles;
    std::vector<rrt_sphere> spheres;
    std::vector<rrt_object> world;
    std::vector<rrt_light> lights;
};

struct Fail {
    rrt_status st;
    std::string msg;
};

[[noreturn]] void fail(rrt_status st, const std::string& m) { throw Fail{st, m}; }

std::string read_file(const std::string& path, bool binary) {
    std::ifstream f(path, binary ? std::ios::binary : std::ios::in);
    if (!f) fail(RRT_ERR_IO, "cannot open " + path);
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

std::shared_ptr<Value> parse_json_file(const std::string& path) {
    std::string text = read_file(path, false);
    rrtjson::Parser p(text);
    auto v = p.parse();
    if (!v) fail(RRT_ERR_PARSE, path + ": " + p.error());
    if (!v->is_object()) fail(RRT_ERR_PARSE, path + ": top-level JSON value is not an object");
    return v;
}

bool base64_decode(const std::string& in, std::vector<uint8_t>& out) {
    static int8_t table[256];
    static bool init = false;
    if (!init) {
        std::memset(table, -1, sizeof table);
        const char* abc = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
        for (int i = 0; i < 64; ++i) table[(unsigned char)abc[i]] = (int8_t)i;
        init = true;
    }
    out.clear();
    out.reserve(in.size() * 3 / 4);
    uint32_t acc = 0;
    int bits = 0;
    size_t pad = 0;
    for (char ch : in) {
        if (ch == '=') { ++pad; continue; }
        if (pad) return false;
        int8_t v = table[(unsigned char)ch];
        if (v < 0) return false;
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            out.push_back((uint8_t)((acc >> bits) & 0xFF));
        }
    }
    return true;
}

// required-field access with serde-like error messages
const Value& need(const Value& o, const char* key, const std::string& where) {
    const Value* v = o.get(key);
    if (!v) fail(RRT_ERR_PARSE, "missing field `" + std::string(key) + "` in " + where);
    return *v;
}
double need_num(const Value& o, const char* key, const std::string& where) {
    const Value& v = need(o, key, where);
    if (!v.is_number()) fail(RRT_ERR_PARSE, "field `" + std::string(key) + "` in " + where + " is not a number");
    return v.num;
}
// Indices, counts and byte offsets a file supplies: non-negative integers below 2^40 (far beyond any real file, and small
// enough that sums and products of a few of them cannot wrap a size_t).
constexpr double INDEX_LIMIT = 1099511627776.0;
constexpr size_t NO_INDEX = (size_t)-1;
// The value as an index, or NO_INDEX (which no container size exceeds) if it is not a number of that kind.
size_t as_index(const Value& v) {
    if (!v.is_number() || !(v.num >= 0.0) || !(v.num < INDEX_LIMIT) || v.num != std::floor(v.num)) return NO_INDEX;
    return (size_t)v.num;
}
size_t need_index(const Value& o, const char* key, const std::string& where) {
    const size_t i = as_index(need(o, key, where));
    if (i == NO_INDEX) fail(RRT_ERR_PARSE, "field `" + std::string(key) + "` in " + where + " is not a non-negative integer below 2^40");
    return i;
}
// An optional index-valued field: `absent` if the field is missing, an error if it is present and malformed.
size_t opt_index(const Value& o, const char* key, const std::string& where, size_t absent) {
    return o.get(key) ? need_index(o, key, where) : absent;
}
const std::string& need_str(const Value& o, const char* key, const std::string& where) {
    const Value& v = need(o, key, where);
    if (!v.is_string()) fail(RRT_ERR_PARSE, "field `" + std::string(key) + "` in " + where + " is not a string");
    return v.str;
}
const Value& need_arr(const Value& o, const char* key, const std::string& where) {
    const Value& v = need(o, key, where);
    if (!v.is_array()) fail(RRT_ERR_PARSE, "field `" + std::string(key) + "` in " + where + " is not an array");
    return v;
}

// ------------------------------------------------------------------------------------------------
// 4x4 column-major matrices for node transforms (the `gltf` crate's Transform::matrix()):
// m[c][r]
// ------------------------------------------------------------------------------------------------
struct Mat4 {
    float m[4][4];
};
Mat4 mat_identity() {
    Mat4 r;
    std::memset(&r, 0, sizeof r);
    for (int i = 0; i < 4; ++i) r.m[i][i] = 1.f;
    return r;
}
Mat4 mat_mul(const Mat4& a, const Mat4& b) {
    Mat4 r;
    for (int c = 0; c < 4; ++c)
        for (int row = 0; row < 4; ++row) {
            float s = 0.f;
            for (int k = 0; k < 4; ++k) s += a.m[k][row] * b.m[c][k];
            r.m[c][row] = s;
        }
    return r;
}
Mat4 node_matrix(const Value& node) {
    if (const Value* mv = node.get("matrix")) {
        if (mv->is_array() && mv->size() == 16) {
            Mat4 r;
            for (int i = 0; i < 16; ++i) r.m[i / 4][i % 4] = (float)mv->at(i).num;
            return r;
        }
    }
    float t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
    if (const Value* v = node.get("translation")) if (v->is_array() && v->size() == 3) for (int i = 0; i < 3; ++i) t[i] = (float)v->at(i).num;
    if (const Value* v = node.get("rotation")) if (v->is_array() && v->size() == 4) for (int i = 0; i < 4; ++i) q[i] = (float)v->at(i).num;
    if (const Value* v = node.get("scale")) if (v->is_array() && v->size() == 3) for (int i = 0; i < 3; ++i) s[i] = (float)v->at(i).num;
    // rotation matrix from the unit quaternion (x, y, z, w)
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float x2 = x + x, y2 = y + y, z2 = z + z;
    const float xx2 = x2 * x, xy2 = x2 * y, xz2 = x2 * z, yy2 = y2 * y, yz2 = y2 * z, zz2 = z2 * z;
    const float sx2 = x2 * w, sy2 = y2 * w, sz2 = z2 * w;
    Mat4 T = mat_identity(), R = mat_identity(), S = mat_identity();
    T.m[3][0] = t[0]; T.m[3][1] = t[1]; T.m[3][2] = t[2];
    R.m[0][0] = 1.f - yy2 - zz2; R.m[0][1] = xy2 + sz2;       R.m[0][2] = xz2 - sy2;
    R.m[1][0] = xy2 - sz2;       R.m[1][1] = 1.f - xx2 - zz2; R.m[1][2] = yz2 + sx2;
    R.m[2][0] = xz2 + sy2;       R.m[2][1] = yz2 - sx2;       R.m[2][2] = 1.f - xx2 - yy2;
    S.m[0][0] = s[0]; S.m[1][1] = s[1]; S.m[2][2] = s[2];
    return mat_mul(mat_mul(T, R), S);
}

// glam Affine3A: 3x3 part + translation
struct Affine {
    float c0[3], c1[3], c2[3], t[3];
};
Affine affine_from(const Mat4& m) {
    Affine a;
    for (int r = 0; r < 3; ++r) { a.c0[r] = m.m[0][r]; a.c1[r] = m.m[1][r]; a.c2[r] = m.m[2][r]; a.t[r] = m.m[3][r]; }
    return a;
}
Affine affine_identity() { return affine_from(mat_identity()); }
void mat3_mul_vec(const Affine& a, const float* v, float* out) {  // ((c0*x + c1*y) + c2*z)
    for (int r = 0; r < 3; ++r) out[r] = (a.c0[r] * v[0] + a.c1[r] * v[1]) + a.c2[r] * v[2];
}
void transform_point(const Affine& a, const float* p, float* out) {  // Affine3A::transform_point3a
    mat3_mul_vec(a, p, out);
    for (int r = 0; r < 3; ++r) out[r] += a.t[r];
}
Affine affine_mul(const Affine& a, const Affine& b) {  // a * b
    Affine r;
    mat3_mul_vec(a, b.c0, r.c0);
    mat3_mul_vec(a, b.c1, r.c1);
    mat3_mul_vec(a, b.c2, r.c2);
    mat3_mul_vec(a, b.t, r.t);
    for (int i = 0; i < 3; ++i) r.t[i] += a.t[i];
    return r;
}

// ------------------------------------------------------------------------------------------------
// camera + lights from the scene graph -- reference src/gltf.rs:262-312,328-341
// ------------------------------------------------------------------------------------------------
struct CamLight {
    bool has_camera = false;
    rrt_camera camera;
    std::vector<rrt_sphere> light_spheres;  // material index filled in by the caller
    std::vector<std::array<float, 3>> light_emission;
};

rrt_camera make_camera(float aspect, float vfov_deg, const Affine& c2w, float time0, float time1) {  // camera.rs:19-39
    rrt_camera c;
    std::memset(&c, 0, sizeof c);
    const float pi = 3.14159274101257324f;
    const float radians = vfov_deg * (pi / 180.0f);  // f32::to_radians
    c.scale = std::tan(radians * 0.5f);
    c.aspect = aspect;
    c.time0 = time0;
    c.time1 = time1;
    for (int r = 0; r < 3; ++r) { c.c2w[r] = c2w.c0[r]; c.c2w[3 + r] = c2w.c1[r]; c.c2w[6 + r] = c2w.c2[r]; c.c2w[9 + r] = c2w.t[r]; }
    const float zero[3] = {0.f, 0.f, 0.f};
    transform_point(c2w, zero, c.origin);
    return c;
}
rrt_camera default_camera() {  // camera.rs:41-54
    return make_camera(1.0f, 30.0f, affine_identity(), 0.0f, 1.0f);
}
void camera_apply(rrt_camera& c, const Affine& other) {  // camera.rs:72-77
    Affine cur;
    for (int r = 0; r < 3; ++r) { cur.c0[r] = c.c2w[r]; cur.c1[r] = c.c2w[3 + r]; cur.c2[r] = c.c2w[6 + r]; cur.t[r] = c.c2w[9 + r]; }
    Affine n = affine_mul(other, cur);
    for (int r = 0; r < 3; ++r) { c.c2w[r] = n.c0[r]; c.c2w[3 + r] = n.c1[r]; c.c2w[6 + r] = n.c2[r]; c.c2w[9 + r] = n.t[r]; }
    float o[3];
    transform_point(other, c.origin, o);
    for (int r = 0; r < 3; ++r) c.origin[r] = o[r];
}

struct NodeResult {
    int kind = 0;  // 0 none, 1 camera, 2 light
    rrt_camera cam;
    float center[3];
    float emission[3];
};

NodeResult handle_node(const Value& root, const Value& nodes, size_t index, int depth) {
    NodeResult res;
    if (depth > 64 || index >= nodes.size()) return res;
    const Value& node = nodes.at(index);
    if (const Value* cam_ref = node.get("camera")) {
        const Value* cams = root.get("cameras");
        if (cams && cams->is_array() && cam_ref->is_number() && as_index(*cam_ref) < cams->size()) {
            const Value& cam = cams->at(as_index(*cam_ref));
            const Value* persp = cam.get("perspective");
            const Value* type = cam.get("type");
            if (persp && type && type->is_string() && type->str == "perspective") {
                const Value* ar = persp->get("aspectRatio");
                const Value* yfov = persp->get("yfov");
                const float aspect = (ar && ar->is_number()) ? (float)ar->num : 1.0f;
                const float yf = (yfov && yfov->is_number()) ? (float)yfov->num : 0.0f;
                const float deg = yf * 57.2957795130823208767981548141051703f;  // f32::to_degrees
                res.kind = 1;
                res.cam = make_camera(aspect, deg, affine_from(node_matrix(node)), 0.0f, 1.0f);
                return res;
            }
        }
    }
    if (const Value* ext = node.get("extensions")) {
        const Value* kl = ext->get("KHR_lights_punctual");
        const Value* li = kl ? kl->get("light") : nullptr;
        const Value* rext = root.get("extensions");
        const Value* rkl = rext ? rext->get("KHR_lights_punctual") : nullptr;
        const Value* lights = rkl ? rkl->get("lights") : nullptr;
        if (li && li->is_number() && lights && lights->is_array() && as_index(*li) < lights->size()) {
            const Value& light = lights->at(as_index(*li));
            float color[3] = {1.f, 1.f, 1.f};
            float intensity = 1.f;
            if (const Value* c = light.get("color")) if (c->is_array() && c->size() == 3) for (int i = 0; i < 3; ++i) color[i] = (float)c->at(i).num;
            if (const Value* iv = light.get("intensity")) if (iv->is_number()) intensity = (float)iv->num;
            res.kind = 2;
            const Affine a = affine_from(node_matrix(node));
            for (int i = 0; i < 3; ++i) {
                res.emission[i] = color[i] * intensity;
                res.center[i] = 0.0f + a.t[i];  // Sphere::apply_transform: centre += translation only (sphere.rs:43-49)
            }
            return res;
        }
    }
    const Value* children = node.get("children");
    if (children && children->is_array() && children->size() == 1 && children->at(0).is_number()) {
        NodeResult child = handle_node(root, nodes, as_index(children->at(0)), depth + 1);
        if (child.kind != 0) {
            const Affine a = affine_from(node_matrix(node));
            if (child.kind == 1) camera_apply(child.cam, a);
            else for (int i = 0; i < 3; ++i) child.center[i] += a.t[i];
            return child;
        }
    }
    return res;
}

void collect_camera_lights(const Value& root, rrt_camera& camera, std::vector<NodeResult>& lights) {
    const Value* scenes = root.get("scenes");
    const Value* nodes = root.get("nodes");
    if (!scenes || !scenes->is_array() || !nodes || !nodes->is_array()) return;
    for (size_t si = 0; si < scenes->size(); ++si) {
        const Value* roots = scenes->at(si).get("nodes");
        if (!roots || !roots->is_array()) continue;
        for (size_t ni = 0; ni < roots->size(); ++ni) {
            if (!roots->at(ni).is_number()) continue;
            NodeResult r = handle_node(root, *nodes, as_index(roots->at(ni)), 0);
            if (r.kind == 1) camera = r.cam;
            else if (r.kind == 2) lights.push_back(r);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// materials -- reference src/gltf.rs:147-168
// ------------------------------------------------------------------------------------------------
rrt_material map_material(const float* base, float metallic, float roughness) {
    rrt_material m;
    std::memset(&m, 0, sizeof m);
    m.rgb[0] = base[0]; m.rgb[1] = base[1]; m.rgb[2] = base[2];
    if (std::fabs(metallic) < 1e-5f) {
        m.kind = RRT_MAT_LAMBERTIAN;
    } else {
        m.kind = RRT_MAT_METAL;
        m.param = roughness;
    }
    return m;
}

// ------------------------------------------------------------------------------------------------
// COMPAT geometry -- reference src/gltf.rs:108-244
// ------------------------------------------------------------------------------------------------
void load_compat(const Value& root, OwnedScene& sc) {
    // serde: every listed field is required
    (void)need_index(root, "scene", "glTF");
    const Value& scenes = need_arr(root, "scenes", "glTF");
    for (size_t i = 0; i < scenes.size(); ++i) {
        (void)need_str(scenes.at(i), "name", "scenes[]");
        (void)need_arr(scenes.at(i), "nodes", "scenes[]");
    }
    const Value& nodes = need_arr(root, "nodes", "glTF");
    for (size_t i = 0; i < nodes.size(); ++i) (void)need_str(nodes.at(i), "name", "nodes[]");
    const Value& materials = need_arr(root, "materials", "glTF");
    const Value& meshes = need_arr(root, "meshes", "glTF");
    const Value& accessors = need_arr(root, "accessors", "glTF");
    const Value& views = need_arr(root, "bufferViews", "glTF");
    const Value& buffers = need_arr(root, "buffers", "glTF");

    // buffers: base64 data URIs only; a buffer whose payload does not decode is dropped (gltf.rs:118-130)
    std::vector<std::vector<uint8_t>> buf_bytes;
    for (size_t i = 0; i < buffers.size(); ++i) {
        (void)need_index(buffers.at(i), "byteLength", "buffers[]");
        const std::string& uri = need_str(buffers.at(i), "uri", "buffers[]");
        size_t comma = uri.find(',');
        if (comma == std::string::npos)
            fail(RRT_ERR_PARSE, "buffers[" + std::to_string(i) + "].uri is not a data URI (external buffers need RRT_GLTF_EXTENDED)");
        size_t end = uri.find(',', comma + 1);
        std::string payload = uri.substr(comma + 1, end == std::string::npos ? std::string::npos : end - comma - 1);
        std::vector<uint8_t> bytes;
        if (base64_decode(payload, bytes)) buf_bytes.push_back(std::move(bytes));
    }
    std::vector<std::vector<uint8_t>> view_bytes;
    for (size_t i = 0; i < views.size(); ++i) {
        const std::string where = "bufferViews[" + std::to_string(i) + "]";
        size_t b = need_index(views.at(i), "buffer", where);
        size_t len = need_index(views.at(i), "byteLength", where);
        size_t off = need_index(views.at(i), "byteOffset", where);
        if (b >= buf_bytes.size() || off > buf_bytes[b].size() || len > buf_bytes[b].size() - off) fail(RRT_ERR_PARSE, where + " exceeds its buffer");
        view_bytes.emplace_back(buf_bytes[b].begin() + (long)off, buf_bytes[b].begin() + (long)(off + len));
    }
    for (size_t i = 0; i < materials.size(); ++i) {
        const std::string where = "materials[" + std::to_string(i) + "]";
        const Value& m = materials.at(i);
        if (!need(m, "doubleSided", where).is_bool()) fail(RRT_ERR_PARSE, "field `doubleSided` in " + where + " is not a bool");
        (void)need_str(m, "name", where);
        const Value& pbr = need(m, "pbrMetallicRoughness", where);
        const Value& bc = need_arr(pbr, "baseColorFactor", where);
        if (bc.size() < 3) fail(RRT_ERR_PARSE, where + ".baseColorFactor has fewer than 3 components");
        float base[3] = {(float)bc.at(0).num, (float)bc.at(1).num, (float)bc.at(2).num};
        sc.materials.push_back(map_material(base, (float)need_num(pbr, "metallicFactor", where), (float)need_num(pbr, "roughnessFactor", where)));
    }
    // accessors: decoded from the START of the buffer view, tightly packed (byteOffset / byteStride ignored)
    struct Acc {
        std::string type;
        size_t view, count;
    };
    std::vector<Acc> accs;
    for (size_t i = 0; i < accessors.size(); ++i) {
        const std::string where = "accessors[" + std::to_string(i) + "]";
        Acc a;
        a.view = need_index(accessors.at(i), "bufferView", where);
        a.count = need_index(accessors.at(i), "count", where);
        a.type = need_str(accessors.at(i), "type", where);
        if (a.view >= view_bytes.size()) fail(RRT_ERR_PARSE, where + ".bufferView out of range");
        size_t elem = a.type == "VEC3" ? 12 : a.type == "VEC2" ? 8 : a.type == "SCALAR" ? 2 : 0;
        if (elem != 0 && a.count > view_bytes[a.view].size() / elem) fail(RRT_ERR_PARSE, where + " reads past its buffer view");
        accs.push_back(a);
    }
    for (size_t mi = 0; mi < meshes.size(); ++mi) {
        const std::string where = "meshes[" + std::to_string(mi) + "]";
        (void)need_str(meshes.at(mi), "name", where);
        const Value& prims = need_arr(meshes.at(mi), "primitives", where);
        for (size_t pi = 0; pi < prims.size(); ++pi) {
            (void)need_index(need(prims.at(pi), "attributes", where), "POSITION", where);
            (void)need_index(prims.at(pi), "indices", where);
            (void)need_index(prims.at(pi), "material", where);
        }
        if (prims.size() == 0) fail(RRT_ERR_PARSE, where + " has no primitives");
        const Value& p0 = prims.at(0);  // only primitives[0] is read (gltf.rs:216-223)
        size_t ia = need_index(p0, "indices", where), pa = need_index(*p0.get("attributes"), "POSITION", where);
        size_t mat = need_index(p0, "material", where);
        if (ia >= accs.size() || pa >= accs.size()) fail(RRT_ERR_PARSE, where + " accessor index out of range");
        if (mat >= sc.materials.size()) fail(RRT_ERR_PARSE, where + " material index out of range");
        const Acc& iacc = accs[ia];
        const Acc& pacc = accs[pa];
        // indices: SCALAR accessors decode as u16, every other kind maps to 0; unknown kinds decode to nothing
        std::vector<uint32_t> idx;
        if (iacc.type == "SCALAR") {
            const uint8_t* b = view_bytes[iacc.view].data();
            for (size_t i = 0; i < iacc.count; ++i) idx.push_back((uint32_t)b[2 * i] | ((uint32_t)b[2 * i + 1] << 8));
        } else if (iacc.type == "VEC3" || iacc.type == "VEC2") {
            idx.assign(iacc.count, 0u);
        }
        std::vector<float> pos;  // xyz
        size_t npos = 0;
        if (pacc.type == "VEC3") {
            npos = pacc.count;
            pos.resize(3 * npos);
            std::memcpy(pos.data(), view_bytes[pacc.view].data(), 12 * npos);
        } else if (pacc.type == "VEC2" || pacc.type == "SCALAR") {
            npos = pacc.count;
            pos.assign(3 * npos, 0.f);
        }
        for (size_t t = 0; t + 2 < idx.size(); t += 3) {
            rrt_triangle tri;
            const uint32_t i0 = idx[t], i1 = idx[t + 1], i2 = idx[t + 2];
            if (i0 >= npos || i1 >= npos || i2 >= npos) fail(RRT_ERR_PARSE, where + " vertex index out of range");
            for (int a = 0; a < 3; ++a) { tri.v0[a] = pos[3 * i0 + a]; tri.v1[a] = pos[3 * i1 + a]; tri.v2[a] = pos[3 * i2 + a]; }
            tri.material = (uint32_t)mat;
            sc.triangles.push_back(tri);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// EXTENDED geometry
// ------------------------------------------------------------------------------------------------
std::string dir_of(const std::string& path) {
    size_t s = path.find_last_of('/');
    return s == std::string::npos ? std::string(".") : path.substr(0, s);
}

void load_extended(const Value& root, const std::string& path, const rrt_gltf_options& opt, OwnedScene& sc) {
    const float scale = opt.extended_scale == 0.0f ? 1.0f : opt.extended_scale;
    const Value* buffers = root.get("buffers");
    const Value* views = root.get("bufferViews");
    const Value* accessors = root.get("accessors");
    const Value* meshes = root.get("meshes");
    const Value* nodes = root.get("nodes");
    if (!buffers || !views || !accessors || !meshes || !nodes) fail(RRT_ERR_PARSE, "glTF lacks buffers / bufferViews / accessors / meshes / nodes");
    std::vector<std::vector<uint8_t>> buf_bytes(buffers->size());
    for (size_t i = 0; i < buffers->size(); ++i) {
        const Value* uri = buffers->at(i).get("uri");
        if (!uri || !uri->is_string()) fail(RRT_ERR_PARSE, "buffers[] without uri (GLB is not supported)");
        if (uri->str.compare(0, 5, "data:") == 0) {
            size_t comma = uri->str.find(',');
            if (comma == std::string::npos || !base64_decode(uri->str.substr(comma + 1), buf_bytes[i])) fail(RRT_ERR_PARSE, "buffers[] data URI does not decode");
        } else {
            std::string bin = read_file(dir_of(path) + "/" + uri->str, true);
            buf_bytes[i].assign(bin.begin(), bin.end());
        }
    }
    if (const Value* materials = root.get("materials")) {
        for (size_t i = 0; i < materials->size(); ++i) {
            float base[3] = {1.f, 1.f, 1.f};
            float metallic = 1.f, roughness = 1.f;
            if (const Value* pbr = materials->at(i).get("pbrMetallicRoughness")) {
                if (const Value* bc = pbr->get("baseColorFactor")) if (bc->is_array() && bc->size() >= 3) for (int k = 0; k < 3; ++k) base[k] = (float)bc->at(k).num;
                if (const Value* v = pbr->get("metallicFactor")) if (v->is_number()) metallic = (float)v->num;
                if (const Value* v = pbr->get("roughnessFactor")) if (v->is_number()) roughness = (float)v->num;
            }
            sc.materials.push_back(map_material(base, metallic, roughness));
        }
    }
    if (sc.materials.empty()) {
        float base[3] = {1.f, 1.f, 1.f};
        sc.materials.push_back(map_material(base, 0.f, 1.f));
    }

    auto accessor_span = [&](size_t ai, size_t elem_bytes, const uint8_t*& base, size_t& stride, size_t& count) {
        if (ai >= accessors->size()) fail(RRT_ERR_PARSE, "accessor index out of range");
        const Value& a = accessors->at(ai);
        const Value* bv = a.get("bufferView");
        if (!bv || !bv->is_number() || as_index(*bv) >= views->size()) fail(RRT_ERR_PARSE, "accessor without a valid bufferView");
        const Value& view = views->at(as_index(*bv));
        size_t b = need_index(view, "buffer", "bufferViews[]");
        size_t voff = opt_index(view, "byteOffset", "bufferViews[]", 0);
        size_t vlen = need_index(view, "byteLength", "bufferViews[]");
        size_t aoff = opt_index(a, "byteOffset", "accessors[]", 0);
        stride = opt_index(view, "byteStride", "bufferViews[]", 0);
        if (stride == 0) stride = elem_bytes;
        count = need_index(a, "count", "accessors[]");
        if (b >= buf_bytes.size()) fail(RRT_ERR_PARSE, "bufferView buffer index out of range");
        // by subtraction and division, so that no sum or product of file-supplied values can wrap
        if (voff > buf_bytes[b].size() || vlen > buf_bytes[b].size() - voff) fail(RRT_ERR_PARSE, "bufferView exceeds its buffer");
        if (count && (aoff > vlen || elem_bytes > vlen - aoff || (count - 1) > (vlen - aoff - elem_bytes) / stride)) fail(RRT_ERR_PARSE, "accessor reads past its buffer view");
        base = buf_bytes[b].data() + voff + aoff;
    };

    std::function<void(size_t, const Mat4&, int)> visit = [&](size_t ni, const Mat4& parent, int depth) {
        if (depth > 64 || ni >= nodes->size()) return;
        const Value& node = nodes->at(ni);
        const Mat4 world = mat_mul(parent, node_matrix(node));
        if (const Value* mref = node.get("mesh")) {
            if (mref->is_number() && as_index(*mref) < meshes->size()) {
                const Value* prims = meshes->at(as_index(*mref)).get("primitives");
                for (size_t pi = 0; prims && pi < prims->size(); ++pi) {
                    const Value& prim = prims->at(pi);
                    if (const Value* mode = prim.get("mode")) if (as_index(*mode) != 4) continue;
                    const Value* attrs = prim.get("attributes");
                    const Value* posa = attrs ? attrs->get("POSITION") : nullptr;
                    if (!posa || !posa->is_number()) continue;
                    if (as_index(*posa) >= accessors->size()) fail(RRT_ERR_PARSE, "POSITION accessor index out of range");
                    if (const Value* ty = accessors->at(as_index(*posa)).get("type")) if (!ty->is_string() || ty->str != "VEC3") continue;
                    const uint8_t* pbase; size_t pstride, pcount;
                    accessor_span(as_index(*posa), 12, pbase, pstride, pcount);
                    std::vector<float> pos(3 * pcount);
                    for (size_t i = 0; i < pcount; ++i) {
                        float p[3];
                        std::memcpy(p, pbase + i * pstride, 12);
                        for (int r = 0; r < 3; ++r) {
                            float w = ((world.m[0][r] * p[0] + world.m[1][r] * p[1]) + world.m[2][r] * p[2]) + world.m[3][r];
                            pos[3 * i + r] = w * scale + opt.extended_offset[r];
                        }
                    }
                    std::vector<uint32_t> idx;
                    if (const Value* ia = prim.get("indices")) {
                        if (as_index(*ia) >= accessors->size()) fail(RRT_ERR_PARSE, "indices accessor index out of range");
                        const Value& acc = accessors->at(as_index(*ia));
                        const size_t ct = opt_index(acc, "componentType", "accessors[]", 5123);
                        size_t eb = ct == 5121 ? 1 : ct == 5125 ? 4 : 2;
                        const uint8_t* ibase; size_t istride, icount;
                        accessor_span(as_index(*ia), eb, ibase, istride, icount);
                        idx.resize(icount);
                        for (size_t i = 0; i < icount; ++i) {
                            const uint8_t* q = ibase + i * istride;
                            idx[i] = eb == 1 ? q[0] : eb == 2 ? ((uint32_t)q[0] | ((uint32_t)q[1] << 8)) : ((uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24));
                        }
                    } else {
                        idx.resize(pcount);
                        for (size_t i = 0; i < pcount; ++i) idx[i] = (uint32_t)i;
                    }
                    uint32_t mat = 0;
                    if (const Value* mv = prim.get("material")) if (mv->is_number() && as_index(*mv) < sc.materials.size()) mat = (uint32_t)mv->num;
                    for (size_t t = 0; t + 2 < idx.size(); t += 3) {
                        if (idx[t] >= pcount || idx[t + 1] >= pcount || idx[t + 2] >= pcount) fail(RRT_ERR_PARSE, "vertex index out of range");
                        rrt_triangle tri;
                        for (int a = 0; a < 3; ++a) { tri.v0[a] = pos[3 * idx[t] + a]; tri.v1[a] = pos[3 * idx[t + 1] + a]; tri.v2[a] = pos[3 * idx[t + 2] + a]; }
                        tri.material = mat;
                        sc.triangles.push_back(tri);
                    }
                }
            }
        }
        if (const Value* ch = node.get("children"))
            for (size_t i = 0; ch->is_array() && i < ch->size(); ++i)
                if (ch->at(i).is_number()) visit(as_index(ch->at(i)), world, depth + 1);
    };
    const Value* scenes = root.get("scenes");
    size_t scene_index = root.get("scene") && root.get("scene")->is_number() ? as_index(*root.get("scene")) : 0;
    if (scenes && scenes->is_array() && scene_index < scenes->size()) {
        const Value* roots = scenes->at(scene_index).get("nodes");
        for (size_t i = 0; roots && roots->is_array() && i < roots->size(); ++i)
            if (roots->at(i).is_number()) visit(as_index(roots->at(i)), mat_identity(), 0);
    }
}

}  // namespace

extern "C" {

rrt_status rrt_gltf_load(const char* path, const char* camera_light_path, const rrt_gltf_options* opt, rrt_scene_desc** out) {
    if (!path || !out) { rrt::set_thread_error("rrt_gltf_load: NULL argument"); return RRT_ERR_INVALID; }
    *out = nullptr;
    rrt_gltf_options o;
    std::memset(&o, 0, sizeof o);
    if (opt) o = *opt;
    OwnedScene* sc = new (std::nothrow) OwnedScene();
    if (!sc) { rrt::set_thread_error("rrt_gltf_load: out of memory"); return RRT_ERR_OOM; }
    try {
        auto root = parse_json_file(path);
        if (o.flags & RRT_GLTF_EXTENDED) load_extended(*root, path, o, *sc);
        else load_compat(*root, *sc);

        // camera + lights (gltf.rs:317-341): from the companion file, or from `path` itself
        std::shared_ptr<Value> cl_root = root;
        if (camera_light_path && std::string(camera_light_path) != path) cl_root = parse_json_file(camera_light_path);
        rrt_camera camera = default_camera();
        std::vector<NodeResult> lights;
        collect_camera_lights(*cl_root, camera, lights);

        if (!sc->triangles.empty()) sc->world.push_back(rrt_object{RRT_OBJ_TRIANGLES, 0u, (uint32_t)sc->triangles.size()});
        for (const NodeResult& l : lights) {
            rrt_material m;
            std::memset(&m, 0, sizeof m);
            m.kind = RRT_MAT_DIFFUSE_LIGHT;
            for (int i = 0; i < 3; ++i) m.rgb[i] = l.emission[i];
            sc->materials.push_back(m);
            rrt_sphere s;
            for (int i = 0; i < 3; ++i) s.center[i] = l.center[i];
            s.radius = 0.2f;
            s.material = (uint32_t)sc->materials.size() - 1u;
            sc->world.push_back(rrt_object{RRT_OBJ_SPHERES, (uint32_t)sc->spheres.size(), 1u});
            sc->spheres.push_back(s);
            rrt_light lr;
            std::memset(&lr, 0, sizeof lr);
            lr.type = RRT_LIGHT_SPHERE;
            lr.p[0] = s.center[0]; lr.p[1] = s.center[1]; lr.p[2] = s.center[2]; lr.p[3] = s.radius;
            sc->lights.push_back(lr);
        }
        rrt_scene_desc& d = sc->desc;
        std::memset(&d, 0, sizeof d);
        d.materials = sc->materials.data(); d.material_count = (uint32_t)sc->materials.size();
        d.triangles = sc->triangles.data(); d.triangle_count = (uint32_t)sc->triangles.size();
        d.spheres = sc->spheres.data();     d.sphere_count = (uint32_t)sc->spheres.size();
        d.world = sc->world.data();         d.world_count = (uint32_t)sc->world.size();
        d.lights = sc->lights.data();       d.light_count = (uint32_t)sc->lights.size();
        d.camera = camera;
        d.background[0] = d.background[1] = d.background[2] = 0.051f;  // gltf.rs:348
        if (d.material_count == 0) fail(RRT_ERR_PARSE, "scene has no materials and no lights");
    } catch (const Fail& f) {
        delete sc;
        rrt::set_thread_error(std::string("rrt_gltf_load: ") + f.msg);
        return f.st;
    } catch (const std::bad_alloc&) {
        delete sc;
        rrt::set_thread_error("rrt_gltf_load: out of memory");
        return RRT_ERR_OOM;
    } catch (const std::exception& e) {
        delete sc;
        rrt::set_thread_error(std::string("rrt_gltf_load: ") + e.what());
        return RRT_ERR_PARSE;
    }
    *out = &sc->desc;
    return RRT_OK;
}

void rrt_scene_free(rrt_scene_desc* scene) {
    if (scene) delete reinterpret_cast<OwnedScene*>(scene);
}

}  // extern "C"
