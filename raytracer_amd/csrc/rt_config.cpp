// rt_config.cpp — scene.json reader (Config / SetupCamera of src/config.rs:7-28 with serde's default
// representation of Shape, Material, Color and Vec3) and the two file-writing render entry points
// of src/lib.rs:137-184.
#include <unistd.h>

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "rt_internal.h"

namespace rt {
namespace {

// ---------------------------------------------------------------- JSON (RFC 8259) into a small DOM
struct Json {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    bool boolean = false;
    double number = 0.0;
    bool integral = false;  // written without fraction or exponent (what serde accepts for u32)
    std::string string;
    std::vector<Json> array;
    std::vector<std::pair<std::string, Json>> object;

    const Json* find(const char* key) const {
        for (const auto& kv : object)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

struct JsonParser {
    const char* p;
    const char* end;
    std::string error;
    int depth = 0;

    bool fail(const std::string& msg, const char* at) {
        if (error.empty()) {
            // line/column like serde_json's messages
            int line = 1, col = 1;
            for (const char* q = begin; q < at && q < end; ++q) {
                if (*q == '\n') { ++line; col = 1; } else { ++col; }
            }
            error = msg + " at line " + std::to_string(line) + " column " + std::to_string(col);
        }
        return false;
    }
    const char* begin;

    void ws() {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
    }
    bool literal(const char* word) {
        size_t n = std::strlen(word);
        if (size_t(end - p) >= n && std::memcmp(p, word, n) == 0) { p += n; return true; }
        return fail("expected value", p);
    }
    static void utf8(std::string& s, uint32_t cp) {
        if (cp < 0x80) s += char(cp);
        else if (cp < 0x800) { s += char(0xc0 | (cp >> 6)); s += char(0x80 | (cp & 0x3f)); }
        else if (cp < 0x10000) { s += char(0xe0 | (cp >> 12)); s += char(0x80 | ((cp >> 6) & 0x3f)); s += char(0x80 | (cp & 0x3f)); }
        else { s += char(0xf0 | (cp >> 18)); s += char(0x80 | ((cp >> 12) & 0x3f)); s += char(0x80 | ((cp >> 6) & 0x3f)); s += char(0x80 | (cp & 0x3f)); }
    }
    bool hex4(uint32_t* out) {
        if (end - p < 4) return fail("EOF while parsing a string", p);
        uint32_t v = 0;
        for (int i = 0; i < 4; ++i) {
            char c = p[i];
            v <<= 4;
            if (c >= '0' && c <= '9') v |= uint32_t(c - '0');
            else if (c >= 'a' && c <= 'f') v |= uint32_t(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= uint32_t(c - 'A' + 10);
            else return fail("invalid escape", p);
        }
        p += 4;
        *out = v;
        return true;
    }
    bool string(std::string* out) {
        ++p;  // opening quote
        for (;;) {
            if (p >= end) return fail("EOF while parsing a string", p);
            unsigned char c = static_cast<unsigned char>(*p++);
            if (c == '"') return true;
            if (c < 0x20) return fail("control character found while parsing a string", p - 1);
            if (c != '\\') { *out += char(c); continue; }
            if (p >= end) return fail("EOF while parsing a string", p);
            char e = *p++;
            switch (e) {
                case '"': *out += '"'; break;
                case '\\': *out += '\\'; break;
                case '/': *out += '/'; break;
                case 'b': *out += '\b'; break;
                case 'f': *out += '\f'; break;
                case 'n': *out += '\n'; break;
                case 'r': *out += '\r'; break;
                case 't': *out += '\t'; break;
                case 'u': {
                    uint32_t cp;
                    if (!hex4(&cp)) return false;
                    if (cp >= 0xd800 && cp < 0xdc00) {  // surrogate pair
                        uint32_t lo;
                        if (end - p < 2 || p[0] != '\\' || p[1] != 'u') return fail("unexpected end of hex escape", p);
                        p += 2;
                        if (!hex4(&lo)) return false;
                        if (lo < 0xdc00 || lo > 0xdfff) return fail("lone leading surrogate in hex escape", p);
                        cp = 0x10000 + ((cp - 0xd800) << 10) + (lo - 0xdc00);
                    } else if (cp >= 0xdc00 && cp < 0xe000) {
                        return fail("lone trailing surrogate in hex escape", p);
                    }
                    utf8(*out, cp);
                    break;
                }
                default: return fail("invalid escape", p - 1);
            }
        }
    }
    bool number(Json* out) {
        const char* s = p;
        bool integral = true;
        if (p < end && *p == '-') ++p;
        if (p >= end) return fail("EOF while parsing a value", p);
        if (*p == '0') {
            ++p;
        } else if (*p >= '1' && *p <= '9') {
            while (p < end && *p >= '0' && *p <= '9') ++p;
        } else {
            return fail("invalid number", p);
        }
        if (p < end && *p == '.') {
            integral = false;
            ++p;
            if (p >= end || *p < '0' || *p > '9') return fail("invalid number", p);
            while (p < end && *p >= '0' && *p <= '9') ++p;
        }
        if (p < end && (*p == 'e' || *p == 'E')) {
            integral = false;
            ++p;
            if (p < end && (*p == '+' || *p == '-')) ++p;
            if (p >= end || *p < '0' || *p > '9') return fail("invalid number", p);
            while (p < end && *p >= '0' && *p <= '9') ++p;
        }
        std::string text(s, p);
        out->type = Json::Number;
        out->number = std::strtod(text.c_str(), nullptr);  // correctly rounded, like serde_json's float parsing
        out->integral = integral;
        return true;
    }
    bool value(Json* out) {
        ws();
        if (p >= end) return fail("EOF while parsing a value", p);
        if (++depth > 128) return fail("recursion limit exceeded", p);
        bool ok = true;
        switch (*p) {
            case '{': {
                out->type = Json::Object;
                ++p;
                ws();
                if (p < end && *p == '}') { ++p; break; }
                for (;;) {
                    ws();
                    if (p >= end) { ok = fail("EOF while parsing an object", p); break; }
                    if (*p != '"') { ok = fail("key must be a string", p); break; }
                    std::string key;
                    if (!string(&key)) { ok = false; break; }
                    ws();
                    if (p >= end || *p != ':') { ok = fail("expected `:`", p); break; }
                    ++p;
                    Json v;
                    if (!value(&v)) { ok = false; break; }
                    out->object.emplace_back(std::move(key), std::move(v));
                    ws();
                    if (p < end && *p == ',') { ++p; continue; }
                    if (p < end && *p == '}') { ++p; break; }
                    ok = fail(p >= end ? "EOF while parsing an object" : "expected `,` or `}`", p);
                    break;
                }
                break;
            }
            case '[': {
                out->type = Json::Array;
                ++p;
                ws();
                if (p < end && *p == ']') { ++p; break; }
                for (;;) {
                    Json v;
                    if (!value(&v)) { ok = false; break; }
                    out->array.push_back(std::move(v));
                    ws();
                    if (p < end && *p == ',') { ++p; continue; }
                    if (p < end && *p == ']') { ++p; break; }
                    ok = fail(p >= end ? "EOF while parsing a list" : "expected `,` or `]`", p);
                    break;
                }
                break;
            }
            case '"': out->type = Json::String; ok = string(&out->string); break;
            case 't': ok = literal("true"); out->type = Json::Bool; out->boolean = true; break;
            case 'f': ok = literal("false"); out->type = Json::Bool; out->boolean = false; break;
            case 'n': ok = literal("null"); out->type = Json::Null; break;
            default:
                if (*p == '-' || (*p >= '0' && *p <= '9')) ok = number(out);
                else ok = fail("expected value", p);
        }
        --depth;
        return ok;
    }
    bool document(Json* out) {
        begin = p;
        if (!value(out)) return false;
        ws();
        if (p != end) return fail("trailing characters", p);
        return true;
    }
};

// ---------------------------------------------------------------- serde-shaped readers
struct Reader {
    std::string error;
    bool fail(const std::string& m) {
        if (error.empty()) error = m;
        return false;
    }
    const Json* field(const Json& obj, const char* what, const char* key) {
        if (obj.type != Json::Object) { fail(std::string("invalid type: expected struct ") + what); return nullptr; }
        const Json* v = obj.find(key);
        if (!v) fail(std::string("missing field `") + key + "`");
        return v;
    }
    bool f64(const Json* v, double* out) {
        if (!v) return false;
        if (v->type != Json::Number) return fail("invalid type: expected f64");
        *out = v->number;
        return true;
    }
    bool u32(const Json* v, uint32_t* out) {
        if (!v) return false;
        if (v->type != Json::Number || !v->integral) return fail("invalid type: expected u32");
        if (v->number < 0.0 || v->number > 4294967295.0) return fail("invalid value: expected u32");
        *out = uint32_t(v->number);
        return true;
    }
    bool vec3(const Json* v, double out[3]) {  // src/vector.rs:7-8: newtype over [f64; 3]
        if (!v) return false;
        if (v->type != Json::Array) return fail("invalid type: expected an array of length 3");
        if (v->array.size() != 3) return fail("invalid length " + std::to_string(v->array.size()) + ", expected an array of length 3");
        for (int i = 0; i < 3; ++i)
            if (!f64(&v->array[size_t(i)], out + i)) return false;
        return true;
    }
    bool color(const Json* v, double out[3]) {  // src/color.rs:4-9: struct {r, g, b}
        if (!v) return false;
        return f64(field(*v, "Color", "r"), out) && f64(field(*v, "Color", "g"), out + 1) && f64(field(*v, "Color", "b"), out + 2);
    }
    // externally tagged enum: {"Variant": {...}}
    bool variant(const Json* v, const char* what, std::string* name, const Json** body) {
        if (!v) return false;
        if (v->type != Json::Object || v->object.size() != 1)
            return fail(std::string("invalid type: expected a map with a single key naming the variant of ") + what);
        *name = v->object[0].first;
        *body = &v->object[0].second;
        return true;
    }
};

}  // namespace
}  // namespace rt

struct rt_config {
    rt_camera camera;
    std::vector<rt_hittable> hittables;
    std::vector<rt_texture> textures;
    std::vector<uint8_t*> owned;  // decoded images (malloc), one per texture
    ~rt_config() {
        for (uint8_t* p : owned) std::free(p);
    }
};

namespace rt {
namespace {

int parse_config(const char* text, size_t len, const char* texture_dir, rt_config** out) {
    Json doc;
    JsonParser jp{text, text + len, {}, 0, text};
    if (!jp.document(&doc)) return fail(RT_ERR_IO, "unable to deserialize scene information: %s", jp.error.c_str());
    Reader rd;
    std::unique_ptr<rt_config> cfg(new rt_config());
    std::map<std::string, int32_t> texture_index;
    auto bad = [&](const std::string& where) {
        return fail(RT_ERR_IO, "unable to deserialize scene information: %s%s%s", rd.error.c_str(), where.empty() ? "" : " in ",
                    where.c_str());
    };
    // camera: SetupCamera (src/config.rs:13-21) -> Camera::build (src/config.rs:24-27)
    const Json* cam = rd.field(doc, "Config", "camera");
    double lookat[3], lookfrom[3], inv_focal_length = 0, aperture = 0;
    uint32_t horiz_res = 0, vert_res = 0;
    if (!cam || !rd.vec3(rd.field(*cam, "SetupCamera", "lookat"), lookat) ||
        !rd.vec3(rd.field(*cam, "SetupCamera", "lookfrom"), lookfrom) ||
        !rd.f64(rd.field(*cam, "SetupCamera", "inv_focal_length"), &inv_focal_length) ||
        !rd.f64(rd.field(*cam, "SetupCamera", "aperture"), &aperture) ||
        !rd.u32(rd.field(*cam, "SetupCamera", "horiz_res"), &horiz_res) ||
        !rd.u32(rd.field(*cam, "SetupCamera", "vert_res"), &vert_res))
        return bad("camera");
    if (int rc = rt_camera_build(lookat, lookfrom, inv_focal_length, aperture, horiz_res, vert_res, &cfg->camera)) return rc;

    const Json* hs = rd.field(doc, "Config", "hittables");
    if (!hs) return bad("");
    if (hs->type != Json::Array) { rd.fail("invalid type: expected a sequence"); return bad("hittables"); }
    for (size_t i = 0; i < hs->array.size(); ++i) {
        const Json& e = hs->array[i];
        const std::string where = "hittables[" + std::to_string(i) + "]";
        rt_hittable h;
        std::memset(&h, 0, sizeof h);
        h.texture = -1;
        std::string kind;
        const Json* body = nullptr;
        // shape (src/geometry.rs:9-47)
        if (!rd.variant(rd.field(e, "Hittable", "shape"), "Shape", &kind, &body)) return bad(where);
        if (kind == "Sphere") {
            h.shape = RT_SHAPE_SPHERE;
            if (!rd.vec3(rd.field(*body, "Sphere", "centre"), h.centre) || !rd.f64(rd.field(*body, "Sphere", "radius"), &h.radius))
                return bad(where);
        } else if (kind == "Disc") {
            h.shape = RT_SHAPE_DISC;
            if (!rd.vec3(rd.field(*body, "Disc", "centre"), h.centre) || !rd.vec3(rd.field(*body, "Disc", "normal"), h.normal) ||
                !rd.f64(rd.field(*body, "Disc", "radius"), &h.radius))
                return bad(where);
        } else {
            rd.fail("unknown variant `" + kind + "`, expected `Sphere` or `Disc`");
            return bad(where);
        }
        // material (src/materials.rs:15-35)
        if (!rd.variant(rd.field(e, "Hittable", "material"), "Material", &kind, &body)) return bad(where);
        if (kind == "Diffuse") {
            h.material = RT_MAT_DIFFUSE;
            if (!rd.color(rd.field(*body, "Diffuse", "albedo"), h.albedo)) return bad(where);
        } else if (kind == "Metal") {
            h.material = RT_MAT_METAL;
            if (!rd.color(rd.field(*body, "Metal", "albedo"), h.albedo) || !rd.f64(rd.field(*body, "Metal", "fuzz"), &h.fuzz))
                return bad(where);
        } else if (kind == "Dielectric") {
            h.material = RT_MAT_DIELECTRIC;
            if (!rd.f64(rd.field(*body, "Dielectric", "refractive_index"), &h.refractive_index)) return bad(where);
        } else if (kind == "Emitter") {
            h.material = RT_MAT_EMITTER;
            if (!rd.color(rd.field(*body, "Emitter", "albedo"), h.albedo)) return bad(where);
        } else if (kind == "TextureMap") {
            h.material = RT_MAT_TEXTUREMAP;
            const Json* map = rd.field(*body, "TextureMap", "map");
            if (!map) return bad(where);
            if (map->type != Json::String) { rd.fail("invalid type: expected a string (path of the image)"); return bad(where); }
            if (!rd.vec3(rd.field(*body, "TextureMap", "orient_up"), h.orient_up) ||
                !rd.vec3(rd.field(*body, "TextureMap", "orient_around"), h.orient_around))
                return bad(where);
            // "map" is a path, opened while the scene is read (src/materials.rs:37-46)
            std::string path = map->string;
            if (!path.empty() && path[0] != '/' && texture_dir && texture_dir[0]) path = std::string(texture_dir) + "/" + path;
            auto it = texture_index.find(path);
            if (it == texture_index.end()) {
                uint8_t* rgb = nullptr;
                uint32_t w = 0, hh = 0;
                if (int rc = rt_load_jpeg(path.c_str(), &rgb, &w, &hh)) return rc;
                cfg->owned.push_back(rgb);
                cfg->textures.push_back(rt_texture{rgb, w, hh});
                it = texture_index.emplace(path, int32_t(cfg->textures.size() - 1)).first;
            }
            h.texture = it->second;
        } else {
            rd.fail("unknown variant `" + kind + "`, expected one of `Diffuse`, `Metal`, `Dielectric`, `TextureMap`, `Emitter`");
            return bad(where);
        }
        cfg->hittables.push_back(h);
    }
    *out = cfg.release();
    return RT_OK;
}

// "r g b\n" per pixel (src/lib.rs:151) appended to fd
int write_lines(int fd, const uint8_t* rgb8, size_t n_pixels) {
    std::string buf;
    buf.reserve(1 << 20);
    char line[16];
    auto flush = [&]() -> bool {
        const char* p = buf.data();
        size_t left = buf.size();
        while (left) {
            ssize_t w = ::write(fd, p, left);
            if (w < 0) {
                if (errno == EINTR) continue;
                return false;
            }
            p += w;
            left -= size_t(w);
        }
        buf.clear();
        return true;
    };
    for (size_t i = 0; i < n_pixels; ++i) {
        int n = std::snprintf(line, sizeof line, "%u %u %u\n", unsigned(rgb8[3 * i]), unsigned(rgb8[3 * i + 1]), unsigned(rgb8[3 * i + 2]));
        buf.append(line, size_t(n));
        if (buf.size() > (1 << 20) - 16 && !flush()) return fail(RT_ERR_IO, "Unable to write colors.: %s", std::strerror(errno));
    }
    if (!flush()) return fail(RT_ERR_IO, "Unable to write colors.: %s", std::strerror(errno));
    return RT_OK;
}

// the pixels as three raw bytes each (the body of a binary "P6" file) appended to fd
int write_raw(int fd, const uint8_t* rgb8, size_t n_pixels) {
    const uint8_t* p = rgb8;
    size_t left = n_pixels * 3;
    while (left) {
        ssize_t w = ::write(fd, p, left < (size_t(1) << 24) ? left : (size_t(1) << 24));
        if (w < 0) {
            if (errno == EINTR) continue;
            return fail(RT_ERR_IO, "Unable to write colors.: %s", std::strerror(errno));
        }
        p += w;
        left -= size_t(w);
    }
    return RT_OK;
}

int render_into_fd(int fd, const rt_camera* cam, const rt_hittable* hittables, size_t n, const rt_texture* textures,
                   size_t n_textures, uint32_t spp, uint32_t depth, uint64_t seed, int device, bool covering_tree,
                   bool binary = false) {
    if (!cam || (!hittables && n)) return fail(RT_ERR_INVALID_ARGUMENT, "render_into_file: null argument");
    if (spp == 0) return fail(RT_ERR_INVALID_ARGUMENT, "render_into_file: zero samples per pixel");
    if (covering_tree)
        for (size_t i = 0; i < n; ++i)  // make_cover_of has no arm for a Disc (src/boundingvolume.rs:242-259)
            if (hittables[i].shape != RT_SHAPE_SPHERE)
                return fail(RT_ERR_UNSUPPORTED, "accel_render_into_file: hittable %zu is not a sphere; only spheres can be covered", i);
    rt_scene* scene = nullptr;
    if (int rc = rt_scene_create(hittables, n, textures, n_textures, device, &scene)) return rc;
    const size_t n_pixels = size_t(cam->horiz_res) * cam->vert_res;
    std::vector<uint8_t> rgb8(n_pixels * 3);
    rt_render_opts opts;
    std::memset(&opts, 0, sizeof opts);
    opts.sample_begin = 0;
    opts.sample_end = spp;
    opts.depth = depth;
    opts.seed = seed;
    int rc = rt_render(scene, cam, &opts, nullptr, rgb8.data());
    rt_scene_destroy(scene);  // the reference drops the tree at return (src/lib.rs:162)
    if (rc) return rc;
    return binary ? write_raw(fd, rgb8.data(), n_pixels) : write_lines(fd, rgb8.data(), n_pixels);
}

}  // namespace
}  // namespace rt

extern "C" {

int rt_config_parse(const char* json, size_t len, const char* texture_dir, rt_config** out) {
    if (!json || !out) return rt::fail(RT_ERR_INVALID_ARGUMENT, "rt_config_parse: null argument");
    *out = nullptr;
    try {
        return rt::parse_config(json, len, texture_dir, out);
    } catch (const std::bad_alloc&) {  // nothing may unwind across the C boundary
        return rt::fail(RT_ERR_LIMIT, "rt_config_parse: out of memory");
    }
}

int rt_config_load(const char* path, rt_config** out) {
    if (!path || !out) return rt::fail(RT_ERR_INVALID_ARGUMENT, "rt_config_load: null argument");
    *out = nullptr;
    FILE* f = std::fopen(path, "rb");
    if (!f) return rt::fail(RT_ERR_IO, "unable to read scene file %s: %s", path, std::strerror(errno));  // src/main.rs:26
    std::string text;
    char chunk[65536];
    size_t n;
    while ((n = std::fread(chunk, 1, sizeof chunk, f)) > 0) text.append(chunk, n);
    std::fclose(f);
    std::string dir(path);
    const size_t slash = dir.find_last_of('/');
    dir = slash == std::string::npos ? std::string() : dir.substr(0, slash);
    return rt_config_parse(text.data(), text.size(), dir.c_str(), out);
}

void rt_config_destroy(rt_config* cfg) { delete cfg; }

int rt_config_view(const rt_config* cfg, rt_camera* camera, const rt_hittable** hittables, size_t* n_hittables,
                   const rt_texture** textures, size_t* n_textures) {
    if (!cfg) return rt::fail(RT_ERR_INVALID_ARGUMENT, "rt_config_view: null config");
    if (camera) *camera = cfg->camera;
    if (hittables) *hittables = cfg->hittables.data();
    if (n_hittables) *n_hittables = cfg->hittables.size();
    if (textures) *textures = cfg->textures.data();
    if (n_textures) *n_textures = cfg->textures.size();
    return RT_OK;
}

int rt_render_into_file(int fd, const rt_camera* cam, const rt_hittable* hittables, size_t n_hittables,
                        const rt_texture* textures, size_t n_textures, uint32_t spp, uint32_t depth, uint64_t seed,
                        int device) {
    return rt::render_into_fd(fd, cam, hittables, n_hittables, textures, n_textures, spp, depth, seed, device, false);
}

int rt_accel_render_into_file(int fd, const rt_camera* cam, const rt_hittable* hittables, size_t n_hittables,
                              const rt_texture* textures, size_t n_textures, uint32_t spp, uint32_t depth, uint64_t seed,
                              int device) {
    return rt::render_into_fd(fd, cam, hittables, n_hittables, textures, n_textures, spp, depth, seed, device, true);
}

int rt_render_into_file_p6(int fd, const rt_camera* cam, const rt_hittable* hittables, size_t n_hittables,
                           const rt_texture* textures, size_t n_textures, uint32_t spp, uint32_t depth, uint64_t seed,
                           int device) {
    return rt::render_into_fd(fd, cam, hittables, n_hittables, textures, n_textures, spp, depth, seed, device, false, true);
}

int rt_accel_render_into_file_p6(int fd, const rt_camera* cam, const rt_hittable* hittables, size_t n_hittables,
                                 const rt_texture* textures, size_t n_textures, uint32_t spp, uint32_t depth, uint64_t seed,
                                 int device) {
    return rt::render_into_fd(fd, cam, hittables, n_hittables, textures, n_textures, spp, depth, seed, device, true, true);
}

}  // extern "C"
