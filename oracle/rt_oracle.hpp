// rt_oracle.hpp — CPU restatement (f64) of the nabilbrice/raytracer hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library may include, link
// or call this file; only tests/, __graft_entry__.smoke() and the cpu_baseline
// leg of bench.py use it, and only as the checker / the timed CPU baseline.
//
// Parity status: the reference is Rust and cannot be compiled in this
// environment, so this restatement is pinned by the reference's own inline
// known-answer tests only (vector.rs:129-147, ray.rs:21-25, geometry.rs:150-211,
// intervals.rs:77-120, boundingvolume.rs:272-513, camera.rs:95-101; restated in
// selftest.cpp and tests/golden/reference_kats.json).  The integrator
// (lib.rs:40-135), Material::scatter/albedo (materials.rs:48-171), the camera
// sample geometry (camera.rs:49-78) and color_to_ppm (lib.rs:186-192) are
// covered by NO reference test: for those this file follows the source text and
// parity is UNPINNED.  The reference's RNG (rand 0.8.5 ThreadRng, OS-seeded) is
// not reproducible by design; it is replaced with Philox4x32-10.
//
// Every function cites the reference file:line it follows.  Arithmetic keeps
// the reference's operation order; compile with -ffp-contract=off.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <vector>

namespace orc {

// ---------------------------------------------------------------- vector.rs
struct Vec3 {
    double v[3];
    double& operator[](int i) { return v[i]; }
    const double& operator[](int i) const { return v[i]; }
};
inline Vec3 V(double x, double y, double z) { return Vec3{{x, y, z}}; }
// vector.rs:36-45
inline Vec3 operator+(Vec3 a, const Vec3& b) { for (int i = 0; i < 3; ++i) a[i] += b[i]; return a; }
// vector.rs:55-64
inline Vec3 operator-(Vec3 a, const Vec3& b) { for (int i = 0; i < 3; ++i) a[i] -= b[i]; return a; }
// vector.rs:47-53
inline Vec3 operator-(const Vec3& a) { return V(-a[0], -a[1], -a[2]); }
// vector.rs:66-80
inline Vec3 operator*(const Vec3& a, double s) { return V(a[0] * s, a[1] * s, a[2] * s); }
inline Vec3 operator*(double s, const Vec3& a) { return a * s; }
// vector.rs:82-88 (true division per component, not a reciprocal multiply)
inline Vec3 operator/(const Vec3& a, double s) { return V(a[0] / s, a[1] / s, a[2] / s); }
// vector.rs:103-105
inline double dot(const Vec3& a, const Vec3& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
// vector.rs:99-101
inline double norm(const Vec3& a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
// vector.rs:107-113
inline Vec3 cross(const Vec3& a, const Vec3& b) {
    return V(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
// vector.rs:116-118
inline Vec3 normalize(const Vec3& a) { return a / norm(a); }

// ----------------------------------------------------------------- color.rs
struct Color { double r, g, b; };
inline Color operator+(const Color& a, const Color& b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }  // color.rs:18-28
inline Color operator*(const Color& a, const Color& b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }  // color.rs:30-40
inline Color operator*(const Color& a, double s) { return {a.r * s, a.g * s, a.b * s}; }              // color.rs:42-52
inline Color operator*(double s, const Color& a) { return a * s; }                                    // color.rs:54-60

// ------------------------------------------------------------------- ray.rs
struct Ray {
    Vec3 orig, dir;
    // ray.rs:9-11: the constructor normalises; a struct literal does not.
    static Ray make(const Vec3& position, const Vec3& point_to) { return Ray{position, normalize(point_to)}; }
    // ray.rs:12-14
    Vec3 position_at(double t) const { return orig + t * dir; }
};

// ------------------------------------------------------------- intervals.rs
struct Interval {
    double start, end;
    double size() const { return end - start; }                  // intervals.rs:14-16
    double midpoint() const { return 0.5 * (start + end); }      // intervals.rs:18-20
};
// intervals.rs:34-47: open rule, start >= end means no intersection.
inline bool intersection(const Interval& a, const Interval& b, Interval* out) {
    double start = (a.start < b.start) ? b.start : a.start;
    double end = (a.end < b.end) ? a.end : b.end;
    if (start >= end) return false;
    *out = Interval{start, end};
    return true;
}
// intervals.rs:49-58
inline Interval cover(const Interval& a, const Interval& b) {
    double start = (a.start < b.start) ? a.start : b.start;
    double end = (a.end < b.end) ? b.end : a.end;
    return Interval{start, end};
}
// intervals.rs:60-65
inline const Interval& get_larger(const Interval& a, const Interval& b) { return (a.size() < b.size()) ? b : a; }

// ------------------------------------------------------- scene description
// Same memory layout as rt_hittable / rt_texture / rt_camera of the shipped
// C ABI, declared independently here so that the oracle needs no product header.
enum { SHAPE_SPHERE = 0, SHAPE_DISC = 1 };
enum { MAT_DIFFUSE = 0, MAT_METAL = 1, MAT_DIELECTRIC = 2, MAT_TEXTUREMAP = 3, MAT_EMITTER = 4 };

struct Hittable {          // lib.rs:25-29 {shape, material} flattened to POD
    int32_t shape;         // geometry.rs:9-15
    int32_t material;      // materials.rs:15-35
    double centre[3];      // geometry.rs:38 / :44
    double normal[3];      // geometry.rs:45 (disc only; used as given, serde bypasses Disc::new)
    double radius;         // geometry.rs:39 / :46
    double albedo[3];      // Diffuse / Metal / Emitter
    double fuzz;           // Metal
    double refractive_index;  // Dielectric
    double orient_up[3];      // TextureMap
    double orient_around[3];  // TextureMap
    int32_t texture;          // TextureMap: index into the texture table
    int32_t pad_;
};

struct Texture { const uint8_t* rgb8; uint32_t width, height; };

struct Camera {  // camera.rs:4-16, field for field
    double lookat[3], lookfrom[3], lookup[3];
    double focal_distance, inv_focal_length, aperture;
    double horiz_arm[3], vert_arm[3];
    uint32_t horiz_res, vert_res;
    double aspect_ratio;
};

inline Vec3 V(const double* p) { return V(p[0], p[1], p[2]); }

// camera.rs:19-47
inline Camera camera_build(const Vec3& lookat, const Vec3& lookfrom, double inv_focal_length, double aperture,
                           uint32_t horiz_res, uint32_t vert_res) {
    Vec3 lookup = V(0.0, 1.0, 0.0);
    Vec3 pointing = lookat - lookfrom;
    double focal_distance = norm(pointing);
    Vec3 horiz_arm = normalize(cross(lookup, pointing));
    Vec3 vert_arm = normalize(cross(pointing, horiz_arm));
    Camera c;
    for (int i = 0; i < 3; ++i) {
        c.lookat[i] = lookat[i]; c.lookfrom[i] = lookfrom[i]; c.lookup[i] = lookup[i];
        c.horiz_arm[i] = horiz_arm[i]; c.vert_arm[i] = vert_arm[i];
    }
    c.focal_distance = focal_distance;
    c.inv_focal_length = inv_focal_length;
    c.aperture = aperture;
    c.horiz_res = horiz_res; c.vert_res = vert_res;
    c.aspect_ratio = double(horiz_res) / double(vert_res);  // computed, never used (camera.rs:33)
    return c;
}

// -------------------------------------------------------------- geometry.rs
// geometry.rs:104-121.  Returns false for None.
inline bool sphere_intersect(const Vec3& centre, double radius, const Ray& ray, double* t) {
    Vec3 rc = ray.orig - centre;
    double b = 2.0 * dot(rc, ray.dir);
    double c = dot(rc, rc) - radius * radius;
    double discrm = b * b - 4.0 * c;
    if (discrm < 0.0) return false;
    double sq = std::sqrt(discrm);
    double t_smaller = -0.5 * (b + sq);
    if (t_smaller > 0.0) { *t = t_smaller; return true; }
    double t_larger = t_smaller + sq;
    if (t_larger > 1.0e-6) { *t = t_larger; return true; }
    return false;
}
// geometry.rs:123-125
inline Vec3 sphere_normal_at(const Vec3& centre, double radius, const Vec3& p) { return (p - centre) / radius; }
// geometry.rs:133-139 (no h > 0 test)
inline bool disc_intersect(const Vec3& centre, const Vec3& normal, double radius, const Ray& ray, double* t) {
    if (dot(normal, ray.dir) == 0.0) return false;
    double h = dot(centre - ray.orig, normal) / dot(normal, ray.dir);
    Vec3 p = ray.position_at(h) - centre;
    if (dot(p, p) > radius * radius) return false;
    *t = h;
    return true;
}
// geometry.rs:17-25
inline bool shape_intersect(const Hittable& h, const Ray& ray, double* t) {
    if (h.shape == SHAPE_SPHERE) return sphere_intersect(V(h.centre), h.radius, ray, t);
    return disc_intersect(V(h.centre), V(h.normal), h.radius, ray, t);
}
// geometry.rs:27-33, :141-143
inline Vec3 shape_normal_at(const Hittable& h, const Vec3& p) {
    if (h.shape == SHAPE_SPHERE) return sphere_normal_at(V(h.centre), h.radius, p);
    return V(h.normal);
}

// -------------------------------------------------------- boundingvolume.rs
struct BoundingBox {
    Interval dims[3];
    int boxed;  // index of the hittable moved into this box, or -1 (boundingvolume.rs:17-20)
};
// boundingvolume.rs:53-58
inline BoundingBox bbox_empty() { return BoundingBox{{{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}}, -1}; }
// boundingvolume.rs:70-95: slab test of the infinite LINE; a zero direction
// component is replaced by the divisor 1e-4.
inline bool bbox_check_intersection(const BoundingBox& bb, const Ray& ray) {
    Interval times[3];
    for (int i = 0; i < 3; ++i) {
        double divisor = (ray.dir[i] == 0.0) ? 1.0e-4 : ray.dir[i];
        double start = (bb.dims[i].start - ray.orig[i]) / divisor;
        double end = (bb.dims[i].end - ray.orig[i]) / divisor;
        times[i] = Interval{start, end};
        if (times[i].size() < 0.0) times[i] = Interval{end, start};
    }
    Interval xy, xyz;
    if (!intersection(times[0], times[1], &xy)) return false;
    return intersection(xy, times[2], &xyz);
}
// boundingvolume.rs:98-108
inline int bbox_longest_axis(const BoundingBox& bb) {
    double s0 = bb.dims[0].size(), s1 = bb.dims[1].size(), s2 = bb.dims[2].size();
    if (s0 > s1 && s0 > s2) return 0;
    if (s1 > s2) return 1;
    return 2;
}
// boundingvolume.rs:120-129
inline BoundingBox make_cover_of(const BoundingBox& a, const BoundingBox& b) {
    return BoundingBox{{cover(a.dims[0], b.dims[0]), cover(a.dims[1], b.dims[1]), cover(a.dims[2], b.dims[2])}, -1};
}
// boundingvolume.rs:146-149: fold starts from [0,0]^3, so every cover holds the origin.
inline BoundingBox make_all_covering(const BoundingBox* boxes, size_t n) {
    BoundingBox acc = bbox_empty();
    for (size_t i = 0; i < n; ++i) acc = make_cover_of(acc, boxes[i]);
    return acc;
}
// boundingvolume.rs:142-144: the sort key is the EXTENT of each box on the axis.
// The reference uses sort_unstable_by (order of equal keys unspecified); a
// stable sort is used here so the oracle itself is deterministic.
inline void sort_on_index(BoundingBox* boxes, size_t n, int idx) {
    std::stable_sort(boxes, boxes + n, [idx](const BoundingBox& a, const BoundingBox& b) {
        return a.dims[idx].size() < b.dims[idx].size();
    });
}
// boundingvolume.rs:242-253
inline BoundingBox make_covering(const Hittable& h, int index) {
    BoundingBox bb;
    for (int i = 0; i < 3; ++i) bb.dims[i] = Interval{h.centre[i] - h.radius, h.centre[i] + h.radius};
    bb.boxed = index;
    return bb;
}

struct CoveringTree {  // boundingvolume.rs:174-178
    BoundingBox cover;
    std::unique_ptr<CoveringTree> left, right;
};
// boundingvolume.rs:192-208 with split_on_covering (:163-170) inlined.
inline std::unique_ptr<CoveringTree> make_coveringtree(BoundingBox* boxes, size_t n) {
    auto node = std::make_unique<CoveringTree>();
    if (n > 1) {
        node->cover = make_all_covering(boxes, n);
        size_t halfway = n / 2;
        sort_on_index(boxes, n, bbox_longest_axis(node->cover));
        node->left = make_coveringtree(boxes, halfway);
        node->right = make_coveringtree(boxes + halfway, n - halfway);
    } else {
        node->cover = boxes[0];
    }
    return node;
}

struct Candidate { int index; bool some; double t; };

struct TraceCounters {
    uint64_t box_tests = 0, shape_tests = 0, segments = 0, paths = 0;
};

// boundingvolume.rs:214-231: pre-order, no culling, pushes even a None.
inline void tree_filter(const CoveringTree* root, const Hittable* scene, std::vector<Candidate>& out, const Ray& ray,
                        TraceCounters* ctr) {
    if (ctr) ctr->box_tests++;
    if (!bbox_check_intersection(root->cover, ray)) return;
    if (root->cover.boxed >= 0) {
        double t = 0.0;
        bool some = shape_intersect(scene[root->cover.boxed], ray, &t);
        if (ctr) ctr->shape_tests++;
        out.push_back(Candidate{root->cover.boxed, some, t});
    }
    if (root->left) tree_filter(root->left.get(), scene, out, ray, ctr);
    if (root->right) tree_filter(root->right.get(), scene, out, ray, ctr);
}

// lib.rs:31-38 + Iterator::min_by: None sorts last, the FIRST minimum wins.
// With tie_lowest_index the winner among equal parameters is the lowest scene
// index instead (the tree's iteration order is implementation-defined in the
// reference because of sort_unstable_by).
inline bool nearest(const std::vector<Candidate>& c, bool tie_lowest_index, int* index, double* t) {
    bool have = false;
    int bi = -1;
    double bt = 0.0;
    for (const Candidate& k : c) {
        if (!k.some) continue;
        if (!have || k.t < bt || (tie_lowest_index && k.t == bt && k.index < bi)) {
            have = true; bi = k.index; bt = k.t;
        }
    }
    if (have) { *index = bi; *t = bt; }
    return have;
}

// --------------------------------------------------------------------- RNG
// Philox4x32-10 (Salmon et al., SC'11).  Replaces rand::thread_rng.
inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = uint64_t(0xD2511F53u) * c0;
        uint64_t p1 = uint64_t(0xCD9E8D57u) * c2;
        uint32_t n0 = uint32_t(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = uint32_t(p1);
        uint32_t n2 = uint32_t(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = uint32_t(p0);
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

enum { SAMPLER_REFERENCE = 0, SAMPLER_DIRECT = 1 };
// Counter layout shared with the device: (pixel index, sample index, bounce,
// attempt), key = 64-bit seed.  bounce 0 is the camera sample.
// SAMPLER_REFERENCE draws a sequential stream of 53-bit doubles (two words per
// draw) and runs the reference's own rejection loops on it; its counter is
// (pixel, sample, block, 0x80000000) so the two modes never share a block.
struct Sampler {
    int mode;
    uint32_t key[2];
    uint32_t pixel, sample;
    uint32_t block = 0, buf[4] = {0, 0, 0, 0};
    int have = 0;
    Sampler(int m, uint64_t seed, uint32_t px, uint32_t s) : mode(m), pixel(px), sample(s) {
        key[0] = uint32_t(seed); key[1] = uint32_t(seed >> 32);
    }
    double next_f64() {  // rand's Standard f64: 53 random bits scaled into [0,1)
        if (have < 2) {
            uint32_t ctr[4] = {pixel, sample, block++, 0x80000000u};
            philox4x32_10(ctr, key, buf);
            have = 4;
        }
        uint64_t lo = buf[4 - have], hi = buf[5 - have];
        have -= 2;
        uint64_t bits = (hi << 32) | lo;
        return double(bits >> 11) * (1.0 / 9007199254740992.0);
    }
    // four 24-bit uniforms in [0,1), exactly representable in f32
    void direct4(uint32_t bounce, uint32_t attempt, double u[4]) const {
        uint32_t ctr[4] = {pixel, sample, bounce, attempt}, o[4];
        philox4x32_10(ctr, key, o);
        for (int i = 0; i < 4; ++i) u[i] = double(o[i] >> 8) * (1.0 / 16777216.0);
    }
};

static const double PI = 3.14159265358979323846;  // std::f64::consts::PI
static const int FUZZ_MAX_ATTEMPTS = 16;

// materials.rs:150-157 (rejection from the cube, then normalise)
inline Vec3 random_vec3_reference(Sampler& s) {
    for (;;) {
        double a = s.next_f64(), b = s.next_f64(), c = s.next_f64();
        Vec3 w = 2.0 * V(a - 0.5, b - 0.5, c - 0.5);
        if (norm(w) > 1.0) continue;
        return normalize(w);
    }
}
// Same distribution (uniform on the unit sphere) from two uniforms.
inline Vec3 unit_sphere_direct(double u0, double u1) {
    double z = 1.0 - 2.0 * u0;
    double r = std::sqrt(std::max(0.0, 1.0 - z * z));
    double phi = 2.0 * PI * u1;
    return V(r * std::cos(phi), r * std::sin(phi), z);
}

// camera.rs:81-89: uniform on the POSITIVE quarter of the unit disc.
inline void random_in_disc(Sampler& s, const double* direct_u, double out[2]) {
    if (s.mode == SAMPLER_REFERENCE) {
        for (;;) {
            double a = s.next_f64(), b = s.next_f64();
            if (a * a + b * b > 1.0) continue;
            out[0] = a; out[1] = b;
            return;
        }
    }
    double r = std::sqrt(direct_u[0]);
    double th = 0.5 * PI * direct_u[1];
    out[0] = r * std::cos(th); out[1] = r * std::sin(th);
}

// camera.rs:49-57
inline Vec3 get_focus_loc(const Camera& c, const double disc[2]) {
    return V(c.lookfrom) + c.aperture * disc[0] * normalize(V(c.horiz_arm)) +
           c.aperture * disc[1] * normalize(V(c.vert_arm));
}
// camera.rs:59-78
inline Vec3 get_sample_loc(const Camera& c, uint32_t i, uint32_t j, const double u[2]) {
    double horiz_increm = 1.0 / double(c.horiz_res);
    double vert_increm = 1.0 / double(c.vert_res);
    Vec3 horiz_nudge = (u[0] * horiz_increm) * V(c.horiz_arm);
    Vec3 vert_nudge = (u[1] * vert_increm) * V(c.vert_arm);
    Vec3 horiz_span = c.inv_focal_length * c.focal_distance * V(c.horiz_arm);
    Vec3 vert_span = c.inv_focal_length * c.focal_distance * V(c.vert_arm);
    double grid_h = -0.5 + double(i) * horiz_increm;
    double grid_v = 0.5 - double(j) * vert_increm;
    return V(c.lookat) + (grid_h * horiz_span) + (grid_v * vert_span) + horiz_nudge + vert_nudge;
}
// lib.rs:168-170: lens sample first, then the pixel jitter, then Ray::new.
inline Ray camera_ray(const Camera& c, uint32_t i, uint32_t j, Sampler& s) {
    double u4[4] = {0, 0, 0, 0};
    if (s.mode == SAMPLER_DIRECT) s.direct4(0, 0, u4);
    double disc[2];
    random_in_disc(s, u4, disc);
    Vec3 focus = get_focus_loc(c, disc);
    double jit[2];
    if (s.mode == SAMPLER_REFERENCE) { jit[0] = s.next_f64(); jit[1] = s.next_f64(); }
    else { jit[0] = u4[2]; jit[1] = u4[3]; }
    return Ray::make(focus, get_sample_loc(c, i, j, jit) - focus);
}

// ------------------------------------------------------------- materials.rs
// materials.rs:133-139
inline bool schlick(double cosine, double r_idx, double u) {
    double r0 = (1.0 - r_idx) / (1.0 + r_idx);
    r0 = r0 * r0;
    double x = 1.0 - std::fabs(cosine);
    double reflectance = r0 + (1.0 - r0) * (x * x * x * x * x);  // powi(5)
    return u < reflectance;
}

// materials.rs:80-130.  ok=false only when the bounded fuzz loop of the direct
// sampler gives up (the reference would recurse without bound there).
inline Ray scatter(const Hittable& h, const Ray& inc, const Vec3& loc, Sampler& s, uint32_t bounce, bool* ok) {
    *ok = true;
    double u4[4] = {0, 0, 0, 0};
    const bool direct = (s.mode == SAMPLER_DIRECT);
    switch (h.material) {
    case MAT_DIFFUSE:
    case MAT_TEXTUREMAP: {  // materials.rs:82-85, :122-125
        Vec3 rv;
        if (direct) { s.direct4(bounce, 0, u4); rv = unit_sphere_direct(u4[0], u4[1]); }
        else rv = random_vec3_reference(s);
        return Ray::make(loc, shape_normal_at(h, loc) + rv);
    }
    case MAT_METAL: {  // materials.rs:86-95 + fuzzify :141-148
        Vec3 n = shape_normal_at(h, loc);
        Vec3 refl = inc.dir - 2.0 * dot(n, inc.dir) * n;
        for (uint32_t attempt = 0;; ++attempt) {
            Vec3 rv;
            if (direct) {
                if (attempt >= uint32_t(FUZZ_MAX_ATTEMPTS)) { *ok = false; return Ray{loc, refl}; }
                s.direct4(bounce, attempt, u4);
                rv = unit_sphere_direct(u4[0], u4[1]);
            } else rv = random_vec3_reference(s);
            Vec3 fuzzy = refl + (h.fuzz * rv);
            if (dot(fuzzy, n) > 0.0) return Ray::make(loc, fuzzy);
        }
    }
    case MAT_DIELECTRIC: {  // materials.rs:96-121
        Vec3 n = shape_normal_at(h, loc);
        double inc_cos = dot(n, inc.dir);
        Vec3 inc_perp = inc.dir - inc_cos * n;
        double ratio = h.refractive_index;
        // f64::signum: +1 for +0.0 and positives, -1 for -0.0 and negatives, NaN for NaN
        double sign_inc = std::isnan(inc_cos) ? inc_cos : (std::signbit(inc_cos) ? -1.0 : 1.0);
        if (sign_inc < 0.0) ratio = 1.0 / h.refractive_index;
        Vec3 sc_perp = ratio * inc_perp;
        double sin2 = dot(sc_perp, sc_perp);
        bool reflect = sin2 > 1.0;
        if (!reflect) {  // `||` short-circuits: the draw happens only when sin2 <= 1
            double u;
            if (direct) { s.direct4(bounce, 0, u4); u = u4[2]; }
            else u = s.next_f64();
            reflect = schlick(inc_cos, ratio, u);
        }
        if (reflect) return Ray::make(loc, inc_perp - inc_cos * n);
        double sc_cos = sign_inc * std::sqrt(1.0 - sin2);
        return Ray::make(loc, sc_perp + sc_cos * n);
    }
    default:
        *ok = false;  // materials.rs:126-128 panics; unreachable from the integrator
        return inc;
    }
}

// materials.rs:159-171 + lib.rs:194-200
inline Color texture_lookup(const Texture& tex, double longitude, double latitude) {
    double col = 0.5 * longitude / PI * double(tex.width);
    double row = latitude / PI * double(tex.height);
    // Rust `as u32` saturates and maps NaN to 0
    auto to_u32 = [](double x) -> uint32_t {
        if (!(x > 0.0)) return 0u;
        if (x >= 4294967295.0) return 4294967295u;
        return uint32_t(x);
    };
    uint32_t c = to_u32(col) % tex.width, r = to_u32(row) % tex.height;
    const uint8_t* p = tex.rgb8 + 3 * (size_t(r) * tex.width + c);
    return Color{double(p[0]) / 255.0, double(p[1]) / 255.0, double(p[2]) / 255.0};
}
// materials.rs:49-79
inline Color albedo(const Hittable& h, const Texture* textures, const Vec3& loc) {
    switch (h.material) {
    case MAT_DIFFUSE: case MAT_METAL: case MAT_EMITTER:
        return Color{h.albedo[0], h.albedo[1], h.albedo[2]};
    case MAT_DIELECTRIC:
        return Color{1.0, 1.0, 1.0};
    default: {
        Vec3 up = V(h.orient_up), around = V(h.orient_around);
        double latitude = std::acos(dot(normalize(up), loc));
        Vec3 a0 = normalize(around);
        Vec3 a1 = cross(normalize(up), normalize(around));
        double longitude = std::atan2(dot(a0, loc), dot(a1, loc)) + PI;
        return texture_lookup(textures[h.texture], longitude, latitude);
    }
    }
}

// ------------------------------------------------------------------ lib.rs
enum { TRAVERSE_LINEAR = 0, TRAVERSE_TREE = 1, TRAVERSE_GATED_BRUTE = 2 };

struct Scene {
    std::vector<Hittable> hittables;
    std::vector<Texture> textures;
    std::vector<std::vector<uint8_t>> texture_data;
    std::unique_ptr<CoveringTree> tree;  // built on demand, spheres only (boundingvolume.rs:254-256)
    size_t tree_nodes = 0;
    bool tie_lowest_index = true;
    // Test aid, 0 = off: every ray direction (camera ray and scattered rays) is moved by up to this much per
    // component before it is traced.  Running the same paths with and without it shows how far an error of that
    // size in a direction — what an f32 evaluation of the same formulas commits — drifts along a path.
    double dir_noise = 0.0;
};

inline size_t count_nodes(const CoveringTree* t) { return t ? 1 + count_nodes(t->left.get()) + count_nodes(t->right.get()) : 0; }
inline int tree_depth(const CoveringTree* t) { return t ? 1 + std::max(tree_depth(t->left.get()), tree_depth(t->right.get())) : 0; }

inline bool build_tree(Scene& sc) {
    std::vector<BoundingBox> boxes;
    for (size_t i = 0; i < sc.hittables.size(); ++i) {
        if (sc.hittables[i].shape != SHAPE_SPHERE) return false;  // unimplemented!() in the reference
        boxes.push_back(make_covering(sc.hittables[i], int(i)));
    }
    if (boxes.empty()) return false;  // unwrap on an empty slice panics (boundingvolume.rs:203)
    sc.tree = make_coveringtree(boxes.data(), boxes.size());
    sc.tree_nodes = count_nodes(sc.tree.get());
    return true;
}

// Candidate list for one segment.
//  LINEAR       lib.rs:46-49   every hittable, scene order
//  TREE         lib.rs:94-95   tree_filter over the pointer tree
//  GATED_BRUTE  the topology-free statement of TREE: a leaf is reached iff its
//               own box passes the line test, because an ancestor's per-axis
//               interval always contains its descendants'.
inline void candidates(const Scene& sc, int traverse, const Ray& ray, std::vector<Candidate>& out, TraceCounters* ctr) {
    out.clear();
    if (traverse == TRAVERSE_TREE) {
        tree_filter(sc.tree.get(), sc.hittables.data(), out, ray, ctr);
        return;
    }
    for (size_t i = 0; i < sc.hittables.size(); ++i) {
        const Hittable& h = sc.hittables[i];
        if (traverse == TRAVERSE_GATED_BRUTE) {
            if (ctr) ctr->box_tests++;
            if (!bbox_check_intersection(make_covering(h, int(i)), ray)) continue;
        }
        double t = 0.0;
        bool some = shape_intersect(h, ray, &t);
        if (ctr) ctr->shape_tests++;
        out.push_back(Candidate{int(i), some, t});
    }
}

// lib.rs:40-86 and :88-135 (identical bodies apart from the candidate source).
struct SegmentLog { int32_t* id; double* t; uint32_t cap; uint32_t n; };

inline Color trace(const Scene& sc, int traverse, Ray ray, uint32_t depth, Sampler& s, std::vector<Candidate>& scratch,
                   TraceCounters* ctr, SegmentLog* log = nullptr) {
    Color color{1.0, 1.0, 1.0};
    for (uint32_t bounce = 1; bounce <= depth; ++bounce) {
        if (sc.dir_noise > 0.0) {  // not part of the reference: see Scene::dir_noise
            uint32_t c4[4] = {s.pixel, s.sample, bounce, 0x4e4f4953u}, o4[4];
            philox4x32_10(c4, s.key, o4);
            Vec3 d = ray.dir;
            for (int i = 0; i < 3; ++i) d[i] += sc.dir_noise * (2.0 * (double(o4[i] >> 8) * (1.0 / 16777216.0)) - 1.0);
            ray = Ray::make(ray.orig, d);
        }
        candidates(sc, traverse, ray, scratch, ctr);
        if (ctr) ctr->segments++;
        int idx = -1; double t = 0.0;
        const bool found = nearest(scratch, sc.tie_lowest_index || traverse != TRAVERSE_TREE, &idx, &t);
        if (log) { if (log->n < log->cap) { log->id[log->n] = found ? idx : -1; log->t[log->n] = found ? t : 0.0; } log->n++; }
        if (found) {
            const Hittable& h = sc.hittables[idx];
            Vec3 loc = ray.position_at(t);
            if (h.material == MAT_EMITTER) {  // lib.rs:101-104: drops the throughput
                double cosine = dot(ray.dir, shape_normal_at(h, loc));
                return Color{h.albedo[0], h.albedo[1], h.albedo[2]} * std::fabs(cosine);
            }
            bool ok;
            Ray next = scatter(h, ray, loc, s, bounce, &ok);
            if (!ok) return Color{0.0, 0.0, 0.0};
            Vec3 rel = (h.shape == SHAPE_SPHERE) ? normalize(loc - V(h.centre)) : (loc - V(h.centre));
            color = color * albedo(h, sc.textures.data(), rel);
            ray = next;
        } else {  // lib.rs:117-130
            double tt = 0.5 * (ray.dir[1] + 1.0);
            Color sky = (1.0 - tt) * Color{1.0, 1.0, 1.0} + tt * Color{0.5, 0.7, 1.0};
            return color * sky;
        }
    }
    return color;  // lib.rs:134: depth exhausted returns the throughput
}

// lib.rs:186-192: (255*sqrt(c)) as u8 — truncating, saturating, NaN -> 0.
inline uint8_t channel_to_ppm(double c) {
    double x = 255.0 * std::sqrt(c);
    if (!(x > 0.0)) return 0;
    if (x >= 255.0) return 255;
    return uint8_t(x);
}

// SplitMix64 host generator used for the seeded scene (the reference draws the
// scene from an unseeded ThreadRng, scenegen.rs:13).
struct SplitMix64 {
    uint64_t state;
    uint64_t next() {
        uint64_t z = (state += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uniform() { return double(next() >> 11) * (1.0 / 9007199254740992.0); }
    double range(double a, double b) { return a + (b - a) * uniform(); }
};

inline Hittable make_sphere(const Vec3& c, double r) {
    Hittable h;
    std::memset(&h, 0, sizeof h);
    h.shape = SHAPE_SPHERE;
    for (int i = 0; i < 3; ++i) h.centre[i] = c[i];
    h.radius = r;
    h.texture = -1;
    return h;
}

// scenegen.rs:12-109: ground + three unit spheres + 22x22 small spheres.
inline std::vector<Hittable> gen_scene(uint64_t seed) {
    SplitMix64 rng{seed};
    std::vector<Hittable> scene;
    Hittable g = make_sphere(V(0.0, -1000.0, 0.0), 1000.0);
    g.material = MAT_DIFFUSE; g.albedo[0] = g.albedo[1] = g.albedo[2] = 0.5;
    scene.push_back(g);
    Hittable glass = make_sphere(V(0.0, 1.0, 0.0), 1.0);
    glass.material = MAT_DIELECTRIC; glass.refractive_index = 1.5;
    scene.push_back(glass);
    Hittable matte = make_sphere(V(-4.0, 1.0, 0.0), 1.0);
    matte.material = MAT_DIFFUSE; matte.albedo[0] = 0.4; matte.albedo[1] = 0.2; matte.albedo[2] = 0.1;
    scene.push_back(matte);
    Hittable metal = make_sphere(V(4.0, 1.0, 0.0), 1.0);
    metal.material = MAT_METAL; metal.albedo[0] = 0.7; metal.albedo[1] = 0.6; metal.albedo[2] = 0.5; metal.fuzz = 0.0;
    scene.push_back(metal);
    for (int x = -11; x < 11; ++x) {
        for (int z = -11; z < 11; ++z) {
            double cx = double(x) + 0.9 * rng.uniform();
            double cz = double(z) + 0.9 * rng.uniform();
            Hittable h = make_sphere(V(cx, 0.2, cz), 0.2);
            double p = rng.uniform();
            if (p < 0.8) {
                h.material = MAT_DIFFUSE;
                for (int k = 0; k < 3; ++k) h.albedo[k] = rng.uniform();
            } else if (p < 0.95) {
                h.material = MAT_METAL;
                for (int k = 0; k < 3; ++k) h.albedo[k] = rng.range(0.5, 1.0);
                h.fuzz = rng.range(0.0, 0.5);
            } else {
                h.material = MAT_DIELECTRIC;
                h.refractive_index = rng.range(1.0, 2.0);
            }
            scene.push_back(h);
        }
    }
    return scene;
}

}  // namespace orc
