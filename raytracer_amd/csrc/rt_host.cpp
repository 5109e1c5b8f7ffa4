// rt_host.cpp — host half of the C ABI that needs no device: error channel,
// Camera::build, the seeded scene generators and the P3 writer.
#include <cmath>
#include <cstdio>
#include <cstring>

#include "rt_internal.h"

namespace rt {

static thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

static void sub3(const double a[3], const double b[3], double o[3]) { for (int i = 0; i < 3; ++i) o[i] = a[i] - b[i]; }
static void cross3(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static double norm3(const double a[3]) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
static void unit3(double a[3]) { double n = norm3(a); for (int i = 0; i < 3; ++i) a[i] = a[i] / n; }

static rt_hittable sphere(double x, double y, double z, double r) {
    rt_hittable h;
    std::memset(&h, 0, sizeof h);
    h.shape = RT_SHAPE_SPHERE;
    h.centre[0] = x; h.centre[1] = y; h.centre[2] = z;
    h.radius = r;
    h.texture = -1;
    return h;
}
static void set_albedo(rt_hittable& h, double r, double g, double b) { h.albedo[0] = r; h.albedo[1] = g; h.albedo[2] = b; }

// src/scenegen.rs:81-109 gen_hittable: 80 % diffuse, 15 % metal, 5 % glass.
static void random_material(SplitMix64& rng, rt_hittable& h) {
    double p = rng.uniform();
    if (p < 0.8) {
        h.material = RT_MAT_DIFFUSE;
        double r = rng.uniform(), g = rng.uniform(), b = rng.uniform();
        set_albedo(h, r, g, b);
    } else if (p < 0.95) {
        h.material = RT_MAT_METAL;
        double r = rng.range(0.5, 1.0), g = rng.range(0.5, 1.0), b = rng.range(0.5, 1.0);
        set_albedo(h, r, g, b);
        h.fuzz = rng.range(0.0, 0.5);
    } else {
        h.material = RT_MAT_DIELECTRIC;
        h.refractive_index = rng.range(1.0, 2.0);
    }
}

}  // namespace rt

using namespace rt;

extern "C" {

const char* rt_last_error(void) { return g_last_error.c_str(); }

uint32_t rt_abi_version(void) { return (1u << 16) | 1u; }

int rt_camera_build(const double lookat[3], const double lookfrom[3], double inv_focal_length, double aperture,
                    uint32_t horiz_res, uint32_t vert_res, rt_camera* out) {
    if (!lookat || !lookfrom || !out) return fail(RT_ERR_INVALID_ARGUMENT, "rt_camera_build: null argument");
    if (horiz_res == 0 || vert_res == 0) return fail(RT_ERR_INVALID_ARGUMENT, "rt_camera_build: zero resolution");
    rt_camera c;
    std::memset(&c, 0, sizeof c);
    const double up[3] = {0.0, 1.0, 0.0};  // src/camera.rs:27
    double pointing[3], h[3], v[3];
    sub3(lookat, lookfrom, pointing);
    c.focal_distance = norm3(pointing);
    cross3(up, pointing, h);
    unit3(h);  // src/camera.rs:31
    cross3(pointing, h, v);
    unit3(v);  // src/camera.rs:32
    for (int i = 0; i < 3; ++i) {
        c.lookat[i] = lookat[i]; c.lookfrom[i] = lookfrom[i]; c.lookup[i] = up[i];
        c.horiz_arm[i] = h[i]; c.vert_arm[i] = v[i];
    }
    c.inv_focal_length = inv_focal_length;
    c.aperture = aperture;
    c.horiz_res = horiz_res;
    c.vert_res = vert_res;
    c.aspect_ratio = double(horiz_res) / double(vert_res);
    *out = c;
    return RT_OK;
}

int rt_default_camera(uint32_t horiz_res, uint32_t vert_res, rt_camera* out) {
    const double lookat[3] = {0.0, 0.0, 0.0}, lookfrom[3] = {13.0, 1.5, 3.0};
    return rt_camera_build(lookat, lookfrom, 1.0, 0.1, horiz_res, vert_res, out);
}

int rt_gen_scene(uint64_t seed, rt_hittable* out, size_t cap, size_t* n) {
    if (!n) return fail(RT_ERR_INVALID_ARGUMENT, "rt_gen_scene: null count pointer");
    std::vector<rt_hittable> sc;
    sc.reserve(488);
    SplitMix64 rng{seed};
    rt_hittable ground = sphere(0.0, -1000.0, 0.0, 1000.0);
    ground.material = RT_MAT_DIFFUSE; set_albedo(ground, 0.5, 0.5, 0.5);
    sc.push_back(ground);
    rt_hittable glass = sphere(0.0, 1.0, 0.0, 1.0);
    glass.material = RT_MAT_DIELECTRIC; glass.refractive_index = 1.5;
    sc.push_back(glass);
    rt_hittable matte = sphere(-4.0, 1.0, 0.0, 1.0);
    matte.material = RT_MAT_DIFFUSE; set_albedo(matte, 0.4, 0.2, 0.1);
    sc.push_back(matte);
    rt_hittable metal = sphere(4.0, 1.0, 0.0, 1.0);
    metal.material = RT_MAT_METAL; set_albedo(metal, 0.7, 0.6, 0.5); metal.fuzz = 0.0;
    sc.push_back(metal);
    for (int x = -11; x < 11; ++x)
        for (int z = -11; z < 11; ++z) {
            double cx = double(x) + 0.9 * rng.uniform();  // src/scenegen.rs:53-57: x offset drawn before z offset
            double cz = double(z) + 0.9 * rng.uniform();
            rt_hittable h = sphere(cx, 0.2, cz, 0.2);
            random_material(rng, h);
            sc.push_back(h);
        }
    *n = sc.size();
    if (out) std::memcpy(out, sc.data(), sizeof(rt_hittable) * (cap < sc.size() ? cap : sc.size()));
    return RT_OK;
}

int rt_gen_scene_grid(uint64_t seed, uint32_t side, rt_hittable* out, size_t cap, size_t* n) {
    if (!n) return fail(RT_ERR_INVALID_ARGUMENT, "rt_gen_scene_grid: null count pointer");
    // pitch 0.5: beyond side 2800 the corners of the grid would leave the ground sphere (radius 1000)
    if (side == 0 || side > 2800) return fail(RT_ERR_LIMIT, "rt_gen_scene_grid: side %u outside 1..2800", side);
    const size_t total = size_t(side) * side + 1;
    *n = total;
    if (!out) return RT_OK;
    SplitMix64 rng{seed};
    size_t k = 0;
    rt_hittable ground = sphere(0.0, -1000.0, 0.0, 1000.0);
    ground.material = RT_MAT_DIFFUSE; set_albedo(ground, 0.5, 0.5, 0.5);
    if (k < cap) out[k] = ground;
    ++k;
    const double pitch = 0.5, half = 0.5 * pitch * double(side);
    for (uint32_t ix = 0; ix < side; ++ix)
        for (uint32_t iz = 0; iz < side; ++iz) {
            // jitter < 0.09 keeps neighbouring centres more than 2r = 0.4 apart: no overlaps, no ties
            double cx = double(ix) * pitch - half + 0.09 * rng.uniform();
            double cz = double(iz) * pitch - half + 0.09 * rng.uniform();
            double cy = std::sqrt(1000.2 * 1000.2 - cx * cx - cz * cz) - 1000.0;  // resting on the ground sphere
            rt_hittable h = sphere(cx, cy, cz, 0.2);
            random_material(rng, h);
            if (k < cap) out[k] = h;
            ++k;
        }
    return RT_OK;
}

int rt_gen_scene_shells(uint32_t shells, rt_hittable* out, size_t cap, size_t* n) {
    if (!n) return fail(RT_ERR_INVALID_ARGUMENT, "rt_gen_scene_shells: null count pointer");
    if (shells == 0 || shells > 64) return fail(RT_ERR_LIMIT, "rt_gen_scene_shells: shells %u outside 1..64", shells);
    std::vector<rt_hittable> sc;
    rt_hittable ground = sphere(0.0, -100.5, -1.0, 100.0);  // scene.json:9 geometry, matte instead of metal
    ground.material = RT_MAT_DIFFUSE; set_albedo(ground, 0.5, 0.5, 0.5);
    sc.push_back(ground);
    rt_hittable earth = sphere(0.0, 0.2, 3.0, 0.75);  // scene.json:18-21
    earth.material = RT_MAT_TEXTUREMAP; earth.texture = 0;
    earth.orient_up[0] = 0.0; earth.orient_up[1] = 1.0; earth.orient_up[2] = -0.6;
    earth.orient_around[0] = 1.0; earth.orient_around[1] = 0.0; earth.orient_around[2] = 0.5;
    sc.push_back(earth);
    double radius = 1.0;
    for (uint32_t k = 0; k < shells; ++k) {
        // src/materials.rs:102-106 only knows the ratio across one surface: a glass shell is an
        // outer sphere of index 1.5 around an inner sphere of index 1/1.5 (air inside glass).
        rt_hittable s = sphere(-0.4, 0.5, 0.6, radius);
        s.material = RT_MAT_DIELECTRIC;
        s.refractive_index = (k % 2 == 0) ? 1.5 : 1.0 / 1.5;
        sc.push_back(s);
        radius *= 0.85;
    }
    *n = sc.size();
    if (out) std::memcpy(out, sc.data(), sizeof(rt_hittable) * (cap < sc.size() ? cap : sc.size()));
    return RT_OK;
}

int rt_write_ppm(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height) {
    if (!path || !rgb8) return fail(RT_ERR_INVALID_ARGUMENT, "rt_write_ppm: null argument");
    FILE* f = std::fopen(path, "wb");  // truncates; the reference opens without truncate (src/main.rs:35-39)
    if (!f) return fail(RT_ERR_IO, "rt_write_ppm: cannot open %s", path);
    std::fprintf(f, "P3\n%u %u\n255\n", width, height);  // src/main.rs:41
    const size_t n_pixels = size_t(width) * height;
    std::vector<char> buf;
    buf.reserve(1 << 20);
    auto put = [&buf](unsigned v) {
        if (v >= 100) buf.push_back(char('0' + v / 100));
        if (v >= 10) buf.push_back(char('0' + (v / 10) % 10));
        buf.push_back(char('0' + v % 10));
    };
    bool ok = true;
    for (size_t p = 0; p < n_pixels && ok; ++p) {  // src/lib.rs:151: "{} {} {}\n"
        put(rgb8[3 * p]); buf.push_back(' ');
        put(rgb8[3 * p + 1]); buf.push_back(' ');
        put(rgb8[3 * p + 2]); buf.push_back('\n');
        if (buf.size() > (1 << 20) - 16) { ok = std::fwrite(buf.data(), 1, buf.size(), f) == buf.size(); buf.clear(); }
    }
    if (ok && !buf.empty()) ok = std::fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    if (std::fclose(f) != 0) ok = false;
    return ok ? RT_OK : fail(RT_ERR_IO, "rt_write_ppm: write to %s failed", path);
}

int rt_write_ppm_p6(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height) {
    if (!path || !rgb8) return fail(RT_ERR_INVALID_ARGUMENT, "rt_write_ppm_p6: null argument");
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(RT_ERR_IO, "rt_write_ppm_p6: cannot open %s", path);
    const size_t n = size_t(width) * height * 3;
    bool ok = std::fprintf(f, "P6\n%u %u\n255\n", width, height) > 0;  // the header of src/main.rs:41 with the binary magic number
    ok = ok && (n == 0 || std::fwrite(rgb8, 1, n, f) == n);
    if (std::fclose(f) != 0) ok = false;
    return ok ? RT_OK : fail(RT_ERR_IO, "rt_write_ppm_p6: write to %s failed", path);
}

int rt_finalize(const float* sum_rgb, size_t n_pixels, uint32_t spp_total, uint8_t* rgb8) {
    if (!sum_rgb || !rgb8) return fail(RT_ERR_INVALID_ARGUMENT, "rt_finalize: null argument");
    if (spp_total == 0) return fail(RT_ERR_INVALID_ARGUMENT, "rt_finalize: zero samples");
    const double inv_spp = 1.0 / double(spp_total);
    for (size_t g = 0; g < n_pixels * 3; ++g) {  // the arithmetic of the device's finalize kernel, operation for operation
        const double x = 255.0 * std::sqrt(inv_spp * double(sum_rgb[g]));
        uint8_t q = 0;
        if (x > 0.0) q = (x >= 255.0) ? uint8_t(255) : uint8_t(int(x));
        rgb8[g] = q;
    }
    return RT_OK;
}

// Checkpoint file: 8-byte magic, rt_checkpoint_info, then width*height*3 floats, all little-endian as in memory.
static const char kCheckpointMagic[8] = {'R', 'T', 'A', 'C', 'C', 'U', 'M', '1'};

int rt_checkpoint_save(const char* path, const rt_checkpoint_info* info, const float* sum_rgb) {
    if (!path || !info || !sum_rgb) return fail(RT_ERR_INVALID_ARGUMENT, "rt_checkpoint_save: null argument");
    if (info->width == 0 || info->height == 0) return fail(RT_ERR_INVALID_ARGUMENT, "rt_checkpoint_save: empty image");
    // write beside the target and rename, so that an interrupted save leaves the previous checkpoint intact
    const std::string tmp = std::string(path) + ".part";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return fail(RT_ERR_IO, "rt_checkpoint_save: cannot open %s", tmp.c_str());
    const size_t n = size_t(info->width) * info->height * 3;
    bool ok = std::fwrite(kCheckpointMagic, 1, 8, f) == 8 && std::fwrite(info, sizeof *info, 1, f) == 1 &&
              std::fwrite(sum_rgb, sizeof(float), n, f) == n;
    ok = (std::fclose(f) == 0) && ok;
    if (!ok || std::rename(tmp.c_str(), path) != 0) {
        std::remove(tmp.c_str());
        return fail(RT_ERR_IO, "rt_checkpoint_save: cannot write %s", path);
    }
    return RT_OK;
}

int rt_checkpoint_load(const char* path, rt_checkpoint_info* info, float* sum_rgb, size_t capacity) {
    if (!path || !info) return fail(RT_ERR_INVALID_ARGUMENT, "rt_checkpoint_load: null argument");
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(RT_ERR_IO, "rt_checkpoint_load: cannot open %s", path);
    char magic[8];
    int rc = RT_OK;
    if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, kCheckpointMagic, 8) != 0 || std::fread(info, sizeof *info, 1, f) != 1) {
        rc = fail(RT_ERR_IO, "rt_checkpoint_load: %s is not a checkpoint file", path);
    } else if (info->width == 0 || info->height == 0 || uint64_t(info->width) * info->height > (uint64_t(1) << 32)) {
        rc = fail(RT_ERR_IO, "rt_checkpoint_load: %s has an implausible header", path);
    } else if (sum_rgb) {
        const size_t n = size_t(info->width) * info->height * 3;
        if (capacity < n) rc = fail(RT_ERR_INVALID_ARGUMENT, "rt_checkpoint_load: buffer of %zu floats, %zu needed", capacity, n);
        else if (std::fread(sum_rgb, sizeof(float), n, f) != n) rc = fail(RT_ERR_IO, "rt_checkpoint_load: %s is truncated", path);
    }
    std::fclose(f);
    return rc;
}

uint64_t rt_hash_bytes(const void* bytes, size_t len, uint64_t basis) {
    uint64_t h = basis ? basis : 0xcbf29ce484222325ull;
    const unsigned char* p = static_cast<const unsigned char*>(bytes);
    for (size_t i = 0; i < len; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}

}  // extern "C"
