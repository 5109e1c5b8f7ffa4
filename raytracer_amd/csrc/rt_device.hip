// rt_device.hip — gfx950 kernels and the device half of the C ABI.
//
// Precision policy
//   * BVH descent runs in f32 on boxes rounded outward, against a ray whose
//     origin is widened by `eps` on every axis; eps covers every rounding of the
//     f32 slab arithmetic, so the descent never skips a sphere the f64 test hits.
//   * The sphere / disc tests and the hit point use the reference's own f64
//     formulas in the reference's operation order (no FMA contraction), so the
//     nearest-hit primitive and its parameter equal the f64 restatement's.
//   * Shading (scatter directions, throughput, sky) runs in f32 with FMA
//     contraction off, so every kernel that shades a path (render, per-path
//     dumps) produces the same bits whatever code surrounds it.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "rt_internal.h"

namespace rt {

#define RT_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (call);                                                                       \
        if (e__ != hipSuccess) return fail(RT_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(e__)); \
    } while (0)

constexpr int kBlock = 256;             // threads per workgroup of every kernel here
constexpr int kLeafSentinel = INT32_MIN;  // "stack empty" marker; leaves are ~k in [-n, -1]
constexpr int kFuzzMaxAttempts = 16;

struct DiscDev {
    double cx, cy, cz;
    double nx, ny, nz;
    double r;
    double pad;
};
struct TexMatDev {  // orientation of one TextureMap material (src/materials.rs:26-31)
    float up[3];
    float a0[3];
    float a1[3];
    int32_t tex;
};
struct TexImgDev {
    const uchar4* px;
    uint32_t w, h;
};

// Node of the HBM-resident tree as the render kernel reads it: the boxes of both children on a 16-bit grid over
// the scene box (coordinate = qbase + q * qstep per axis), each as CENTRE and HALF EXTENT per axis (the centre is a
// grid point, the half extent is rounded up so that the box only grows) + 2 child links: 32 bytes = two 16-byte
// loads per lane instead of the four of the f32 Node.  Centre / half extent instead of low / high bound: the slab
// test then needs no min / max per axis (near and far parameter are centre -+ half extent times |1/d|), and the two
// children's additions pair up in packed instructions.
struct Node32 {
    uint16_t q[12];  // per axis x, y, z: centre of child 0, centre of child 1, half extent of child 0, of child 1
    int32_t child0, child1;
};
static_assert(sizeof(Node32) == 32, "Node32 must be 32 bytes");

struct SceneDev {
    const Node* nodes;
    const Node32* nodes32;  // the same tree, quantised (render_kernel_ws without the scene in LDS)
    float qbase[3], qstep[3];  // grid of nodes32
    float hbase[3], hstep[3];  // the f16 boxes of the LDS-resident tree are (plane - hbase) / hstep
    float eps_rel;             // origin widening of the render kernels' descent, which works relative to qbase / hbase
    float t_scale;             // the render kernels' descent measures ray parameters in units of 1 / t_scale, a length no segment of this scene reaches
    const double4* sph;     // the n_tree spheres of the BVH in leaf order, then the n_wide spheres kept out of it
    const DiscDev* discs;   // scene order
    const float4* mat;      // spheres in leaf order, then discs: albedo rgb + (fuzz | refractive index)
    const int32_t* kind;    // same indexing: material kind | texmat index << 8
    const int32_t* prim_id; // same indexing: hittable index
    const TexMatDev* texmat;
    const TexImgDev* teximg;
    int32_t n_nodes, n_spheres, n_discs;
    int32_t shell_first, shell_count;  // wide spheres [shell_first, shell_first + shell_count): concentric, radii ascending and
                                       // well separated (nested shells); 0 = no such group
    double shell_far2;          // squared distance from their centre beyond which the shortcut below is not taken
    const double* shell_r2;     // radius * radius of those spheres (ascending), rounded as the sphere test rounds it
    uint32_t wide_same_centre;  // bit i: wide sphere i has the centre of wide sphere i - 1 (they follow one another in the tables)
    int32_t n_tree, n_wide; // n_spheres = n_tree + n_wide; a wide sphere's box covers most of the others, so every
                            // ray would reach its leaf anyway: it is tested once per segment before the descent
    float eps;              // origin widening for the f32 descent
};

// The per-primitive tables a kernel reads while shading: the HBM copies, or their LDS copies.
struct Tables {
    const double4* sph;
    const float4* mat;
    const int32_t* kind;
    const int32_t* prim_id;
    const double* shell_r2;
};
__device__ __forceinline__ Tables global_tables(const SceneDev& S) { return Tables{S.sph, S.mat, S.kind, S.prim_id, S.shell_r2}; }

struct CamDev {
    float lookfrom[3], lookat[3], h[3], v[3];
    float aperture, span, inv_w, inv_h;
    uint32_t W, H;
};

// The render kernel keeps the camera in LDS: twenty scalar registers held across the whole loop for a block of code
// that runs once per path cost more in register spills than five broadcast reads cost there.
typedef __attribute__((address_space(3))) const CamDev lds_cam;
constexpr uint32_t kCamLdsBytes = 96;
static_assert(sizeof(CamDev) <= kCamLdsBytes, "camera copy in LDS");

struct RenderParams {
    uint32_t sample_begin, sample_end, depth, flags;
    uint32_t key0, key1;
    uint32_t n_pixels, tiles_x, tiles_y;
    uint32_t trav_threshold, inner_threshold, leave_threshold;  // schedule of render_kernel_ws
    uint32_t n_chunks, chunk_spp;              // a work item = one 8x8 tile x one chunk of chunk_spp samples
    uint32_t band_items;                       // items per work counter when each counter owns a contiguous band (0: interleaved)
};

// ---------------------------------------------------------------- Philox
__device__ __forceinline__ uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = uint64_t(0xD2511F53u) * c0;
        uint64_t p1 = uint64_t(0xCD9E8D57u) * c2;
        uint32_t n0 = uint32_t(p1 >> 32) ^ c1 ^ k0;
        uint32_t n2 = uint32_t(p0 >> 32) ^ c3 ^ k1;
        c1 = uint32_t(p1);
        c3 = uint32_t(p0);
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}
__device__ __forceinline__ float u01(uint32_t x) { return float(x >> 8) * (1.0f / 16777216.0f); }
// v_rcp_f32 / v_sqrt_f32: one instruction, about 1 ulp (the IEEE-exact forms cost ~10 instructions each)
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// sin/cos of 2*pi*rev (v_sin_f32 / v_cos_f32 take revolutions)
__device__ __forceinline__ void sincos_rev(float rev, float* s, float* c) {
    *s = __builtin_amdgcn_sinf(rev);
    *c = __builtin_amdgcn_cosf(rev);
}
// uniform on the unit sphere: same distribution as random_vec3 (src/materials.rs:150-157)
__device__ __forceinline__ void unit_sphere(float u0, float u1, float* x, float* y, float* z) {
#pragma clang fp contract(off)
    float zz = fmaf(-2.0f, u0, 1.0f);
    float r = fast_sqrt(fmaxf(0.0f, fmaf(-zz, zz, 1.0f)));
    float s, c;
    sincos_rev(u1, &s, &c);
    *x = r * c;
    *y = r * s;
    *z = zz;
}

// ------------------------------------------------------ f64 shape tests
// src/geometry.rs:104-121 in the reference's operation order.
// Correctly rounded f64 square root of a discriminant: the compiler's own sequence (v_rsq_f64, one joint refinement of
// root and half reciprocal root, two corrections of the root) without the rescaling it wraps around it for arguments
// below 2^-767, which a discriminant of coordinates of ordinary size never is unless it is zero.  Same bits as sqrt()
// from 2^-767 up; zero and +infinity are returned as they are; a negative argument gives NaN.
__device__ __forceinline__ double sqrt_discriminant(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    return __builtin_amdgcn_class(x, 0x260) ? x : g;  // +0, -0, +infinity
}

__device__ __forceinline__ bool sphere_intersect_f64(double cx, double cy, double cz, double r, double ox, double oy,
                                                     double oz, double dx, double dy, double dz, double* t) {
#pragma clang fp contract(off)
    double rx = ox - cx, ry = oy - cy, rz = oz - cz;
    double b = 2.0 * (rx * dx + ry * dy + rz * dz);
    double c = (rx * rx + ry * ry + rz * rz) - r * r;
    double discrm = b * b - 4.0 * c;
    if (discrm < 0.0) return false;
    double sq = sqrt_discriminant(discrm);
    double t_smaller = -0.5 * (b + sq);
    if (t_smaller > 0.0) { *t = t_smaller; return true; }
    double t_larger = t_smaller + sq;
    if (t_larger > 1.0e-6) { *t = t_larger; return true; }
    return false;
}
// src/geometry.rs:133-139: no h > 0 test.
__device__ __forceinline__ bool disc_intersect_f64(const DiscDev& d, double ox, double oy, double oz, double dx,
                                                   double dy, double dz, double* t) {
#pragma clang fp contract(off)
    double nd = d.nx * dx + d.ny * dy + d.nz * dz;
    if (nd == 0.0) return false;
    double h = ((d.cx - ox) * d.nx + (d.cy - oy) * d.ny + (d.cz - oz) * d.nz) / nd;
    double px = (ox + h * dx) - d.cx, py = (oy + h * dy) - d.cy, pz = (oz + h * dz) - d.cz;
    if (px * px + py * py + pz * pz > d.r * d.r) return false;
    *t = h;
    return true;
}
// src/boundingvolume.rs:70-95 applied to the sphere's own box [c-r, c+r]
// (src/boundingvolume.rs:246-248): the LINE test with the 1e-4 divisor.
__device__ __forceinline__ bool own_box_line_test_f64(double cx, double cy, double cz, double r, double ox, double oy,
                                                      double oz, double dx, double dy, double dz) {
#pragma clang fp contract(off)
    double c[3] = {cx, cy, cz}, o[3] = {ox, oy, oz}, d[3] = {dx, dy, dz};
    double s[3], e[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double divisor = (d[i] == 0.0) ? 1.0e-4 : d[i];
        double start = ((c[i] - r) - o[i]) / divisor;
        double end = ((c[i] + r) - o[i]) / divisor;
        if (end - start < 0.0) { double tmp = start; start = end; end = tmp; }
        s[i] = start;
        e[i] = end;
    }
    double st = (s[0] < s[1]) ? s[1] : s[0];
    double en = (e[0] < e[1]) ? e[0] : e[1];
    if (st >= en) return false;
    double st2 = (st < s[2]) ? s[2] : st;
    double en2 = (en < e[2]) ? en : e[2];
    return !(st2 >= en2);
}

// ------------------------------------------------------------ traversal
struct Best {
    double t;
    int32_t k;   // index into mat/kind/prim_id, -1 = none
    int32_t id;  // hittable index of k (tie rule: lowest wins)
};

__device__ __forceinline__ void consider(Best& b, double t, int32_t k, int32_t id) {
    if (b.k < 0 || t < b.t || (t == b.t && id < b.id)) { b.t = t; b.k = k; b.id = id; }
}

// Nearest hit of one ray.  `stk` points at this lane's column of the LDS stack
// (entry `level` lives at stk[level * kBlock]).  GATE adds the reference's
// own-box line test in front of every sphere test (bit-exact candidate set).
template <bool GATE>
__device__ __forceinline__ Best closest_hit(const SceneDev& S, double ox, double oy, double oz, double dx, double dy,
                                            double dz, int32_t* stk) {
    Best best;
    best.t = 0.0;
    best.k = -1;
    best.id = 0x7fffffff;
    const float dxf = float(dx), dyf = float(dy), dzf = float(dz);

    // discs first, one by one, with LINE semantics (negative parameters count)
    for (int32_t i = 0; i < S.n_discs; ++i) {
        double t;
        if (disc_intersect_f64(S.discs[i], ox, oy, oz, dx, dy, dz, &t)) consider(best, t, S.n_spheres + i, S.prim_id[S.n_spheres + i]);
    }
    // a sphere parameter is always positive: nothing can beat a non-positive disc hit
    if (best.k >= 0 && best.t <= 0.0) return best;
    for (int32_t i = 0; i < S.n_wide; ++i) {
        const int32_t k = S.n_tree + i;
        const double4 s = S.sph[k];
        double t;
        bool ok = true;
        if (GATE) ok = own_box_line_test_f64(s.x, s.y, s.z, s.w, ox, oy, oz, dx, dy, dz);
        if (ok && sphere_intersect_f64(s.x, s.y, s.z, s.w, ox, oy, oz, dx, dy, dz, &t)) consider(best, t, k, S.prim_id[k]);
    }
    if (S.n_nodes == 0) return best;

    const float of[3] = {float(ox), float(oy), float(oz)};
    const float df[3] = {dxf, dyf, dzf};
    float inv[3], A[3], B[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float d = df[a];
        float i = (fabsf(d) >= 1.0e-18f) ? fast_rcp(d) : copysignf(1.0e18f, d);
        inv[a] = i;
        A[a] = -(of[a] + S.eps) * i;
        B[a] = -(of[a] - S.eps) * i;
    }
    // Culling to the parameter range [0, nearest hit so far] rests on a hit's parameter lying inside the interval in
    // which the ray crosses the sphere's box.  That holds for unit directions (every path of a render: unit_f64).
    // The parity entry point takes directions as given, and Sphere::intersect (src/geometry.rs:104-121) drops the
    // quadratic's leading coefficient: with |d|^2 = 1 + e it answers for a sphere of radius^2 r^2 - e * |o - c|^2,
    // whose hits may lie outside the box's interval (e = 1e-7 at 200 units is 4e-3 against r^2 = 9e-4).  So with
    // GATE the descent follows the reference's LINE test (src/boundingvolume.rs:70-95): no clip at 0, no clip at the
    // nearest hit; the gate and the quadratic then decide exactly as the reference does.
    const float tmin = GATE ? -3.0e38f : 0.0f;
    float tmax = 3.0e38f;
    if (!GATE && best.k >= 0) tmax = float(best.t) * (1.0f + 4.0e-7f);

    int sp = 0;
    int32_t node = 0;
    while (node != kLeafSentinel) {
        while (node >= 0) {
            const float4* q = reinterpret_cast<const float4*>(S.nodes + node);
            const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            // q0 = lo0.xyz hi0.x | q1 = hi0.yz lo1.xy | q2 = lo1.z hi1.xyz | q3 = child0 child1 pad pad
            float t0x = fmaf(q0.x, inv[0], A[0]), t1x = fmaf(q0.w, inv[0], B[0]);
            float t0y = fmaf(q0.y, inv[1], A[1]), t1y = fmaf(q1.x, inv[1], B[1]);
            float t0z = fmaf(q0.z, inv[2], A[2]), t1z = fmaf(q1.y, inv[2], B[2]);
            float en0 = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), tmin));
            float ex0 = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
            float u0x = fmaf(q1.z, inv[0], A[0]), u1x = fmaf(q2.y, inv[0], B[0]);
            float u0y = fmaf(q1.w, inv[1], A[1]), u1y = fmaf(q2.z, inv[1], B[1]);
            float u0z = fmaf(q2.x, inv[2], A[2]), u1z = fmaf(q2.w, inv[2], B[2]);
            float en1 = fmaxf(fmaxf(fminf(u0x, u1x), fminf(u0y, u1y)), fmaxf(fminf(u0z, u1z), tmin));
            float ex1 = fminf(fminf(fmaxf(u0x, u1x), fmaxf(u0y, u1y)), fminf(fmaxf(u0z, u1z), tmax));
            const bool h0 = en0 <= ex0, h1 = en1 <= ex1;
            const int32_t c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y);
            if (h0 && h1) {
                const bool swap = en1 < en0;
                const int32_t near_c = swap ? c1 : c0, far_c = swap ? c0 : c1;
                stk[sp * kBlock] = far_c;
                ++sp;
                node = near_c;
            } else if (h0) {
                node = c0;
            } else if (h1) {
                node = c1;
            } else if (sp > 0) {
                --sp;
                node = stk[sp * kBlock];
            } else {
                node = kLeafSentinel;
            }
        }
        if (node != kLeafSentinel) {
            const int32_t k = ~node;
            const double4 s = S.sph[k];
            double t;
            bool ok = true;
            if (GATE) ok = own_box_line_test_f64(s.x, s.y, s.z, s.w, ox, oy, oz, dx, dy, dz);
            if (ok && sphere_intersect_f64(s.x, s.y, s.z, s.w, ox, oy, oz, dx, dy, dz, &t)) {
                consider(best, t, k, S.prim_id[k]);
                if (!GATE) tmax = float(best.t) * (1.0f + 4.0e-7f);
            }
            if (sp > 0) {
                --sp;
                node = stk[sp * kBlock];
            } else {
                node = kLeafSentinel;
            }
        }
    }
    return best;
}

// ---------------------------------------------------------------- shading
struct Path {
    double ox, oy, oz;
    double dx, dy, dz;  // unit to f64 rounding, see unit_f64
    float tr, tg, tb;   // throughput ("color" in src/lib.rs:89)
    uint32_t pixel, sample, bounce;
};

__device__ __forceinline__ void normalize3(float& x, float& y, float& z) {
#pragma clang fp contract(off)
    float inv = __builtin_amdgcn_rsqf(x * x + y * y + z * z);  // one instruction: the squared length of a direction is no denormal
    x *= inv; y *= inv; z *= inv;
}

// Sphere::intersect (src/geometry.rs:104-121) drops the quadratic's leading
// coefficient, so the hit point lies on the sphere only as well as |dir| = 1
// holds; its 1e-6 self-intersection guard assumes f64 accuracy there.  An
// f32-normalised direction (|d|^2 = 1 + e, |e| ~ 1e-7) is rescaled in f64 by the
// series of 1/sqrt(1 + e), which leaves |d| = 1 to f64 rounding.
__device__ __forceinline__ void unit_f64(float x, float y, float z, double* X, double* Y, double* Z) {
#pragma clang fp contract(off)
    const double a = double(x), b = double(y), c = double(z);
    const double e = (a * a + b * b + c * c) - 1.0;
    const double k = 1.0 - 0.5 * e + 0.375 * (e * e);
    *X = a * k; *Y = b * k; *Z = c * k;
}

// src/camera.rs:49-78 + src/ray.rs:9-11 with the two rejection loops replaced
// by direct draws of the same distributions (quarter disc, unit square).
// r = Philox block (pixel, sample, 0, 0).
// CamT: CamDev (kernel argument, lives in scalar registers) or its copy in LDS (lds_cam below).
// camera_ray_raw leaves the direction as pixel - origin; Ray::new's normalisation is the caller's (the render kernel
// shares it with the scatter step).
template <class CamT>
__device__ __forceinline__ void camera_ray_raw(const CamT& C, uint32_t i, uint32_t j, const uint4 r, float o[3], float d[3]) {
#pragma clang fp contract(off)
    const float rad = fast_sqrt(u01(r.x));
    float s, c;
    sincos_rev(0.25f * u01(r.y), &s, &c);
    const float la = C.aperture * (rad * c), lb = C.aperture * (rad * s);
    const float gh = fmaf(fmaf(float(i), C.inv_w, -0.5f), C.span, u01(r.z) * C.inv_w);
    const float gv = fmaf(fmaf(-float(j), C.inv_h, 0.5f), C.span, u01(r.w) * C.inv_h);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        o[a] = fmaf(lb, C.v[a], fmaf(la, C.h[a], C.lookfrom[a]));
        d[a] = fmaf(gv, C.v[a], fmaf(gh, C.h[a], C.lookat[a])) - o[a];
    }
}

template <class CamT>
__device__ __forceinline__ void camera_ray_from(const CamT& C, uint32_t i, uint32_t j, const uint4 r, float o[3], float d[3]) {
    camera_ray_raw(C, i, j, r, o, d);
    normalize3(d[0], d[1], d[2]);
}

__device__ __forceinline__ void camera_ray(const CamDev& C, const RenderParams& P, uint32_t pixel, uint32_t sample,
                                           float o[3], float d[3]) {
    camera_ray_from(C, pixel % C.W, pixel / C.W, philox4x32_10(pixel, sample, 0u, 0u, P.key0, P.key1), o, d);
}

template <class CamT>
__device__ __forceinline__ void start_path_from(const CamT& C, uint32_t pixel, uint32_t px, uint32_t py, uint32_t sample,
                                                const uint4 r, Path& p) {
    float o[3], d[3];
    camera_ray_from(C, px, py, r, o, d);
    p.ox = double(o[0]); p.oy = double(o[1]); p.oz = double(o[2]);
    unit_f64(d[0], d[1], d[2], &p.dx, &p.dy, &p.dz);
    p.tr = p.tg = p.tb = 1.0f;
    p.pixel = pixel; p.sample = sample; p.bounce = 1;
}

__device__ __forceinline__ void start_path(const CamDev& C, const RenderParams& P, uint32_t pixel, uint32_t sample,
                                           Path& p) {
    start_path_from(C, pixel, pixel % C.W, pixel / C.W, sample, philox4x32_10(pixel, sample, 0u, 0u, P.key0, P.key1), p);
}

// src/materials.rs:59-76 + :159-171 + src/lib.rs:194-200
__device__ __forceinline__ void texture_albedo(const SceneDev& S, int32_t texmat, float lx, float ly, float lz, float* r,
                                               float* g, float* b) {
#pragma clang fp contract(off)
    const TexMatDev& m = S.texmat[texmat];
    const TexImgDev img = S.teximg[m.tex];
    const float kPi = 3.14159265358979323846f;
    float lat = acosf(m.up[0] * lx + m.up[1] * ly + m.up[2] * lz);
    float lon = atan2f(m.a0[0] * lx + m.a0[1] * ly + m.a0[2] * lz, m.a1[0] * lx + m.a1[1] * ly + m.a1[2] * lz) + kPi;
    // (times 1 / pi instead of the two divisions by pi: about ten instructions each; a texel index moves only when the
    // product lies within an ulp of a whole number)
    const float kInvPi = 0.318309886183790671538f;
    float colf = (0.5f * lon) * kInvPi * float(img.w);
    float rowf = lat * kInvPi * float(img.h);
    // Rust `as u32`: saturating, NaN -> 0
    uint32_t col = (colf > 0.0f) ? (colf >= 4294967040.0f ? 4294967295u : uint32_t(colf)) : 0u;
    uint32_t row = (rowf > 0.0f) ? (rowf >= 4294967040.0f ? 4294967295u : uint32_t(rowf)) : 0u;
    // row <= h and col <= w for any finite direction (lat in [0, pi], lon in [0, 2 pi]); the remainder is only
    // computed where it can change something, an integer division is ~25 instructions
    if (row >= img.h) row %= img.h;
    if (col >= img.w) col %= img.w;
    const uchar4 px = img.px[size_t(row) * img.w + col];
    *r = float(px.x) * (1.0f / 255.0f);
    *g = float(px.y) * (1.0f / 255.0f);
    *b = float(px.z) * (1.0f / 255.0f);
}

// What the scatter step needs to know about the surface that was hit.
struct Surf {
    double hx, hy, hz;  // scatter_loc
    float nx, ny, nz;   // normal_at
    float ddn;          // dir . normal
    float w;            // fuzz (metal) or refractive index (dielectric)
    int32_t kind;
};

// First half of the loop body of src/lib.rs:93-131: everything that needs no random number.  Returns
// true when the path scatters on (sf filled, throughput already multiplied by the albedo); false when
// it ends here with the colour (Lr,Lg,Lb): a miss (sky), an emitter, or the depth limit.  At the depth
// limit the reference still calls scatter and then drops the ray (src/lib.rs:105, :134); with a
// counter-based RNG that call has no effect on anything, so it is not made.
// LEAN: what the scene is known not to hold, so that the branches for it are not compiled in.  kLeanNoDiscs: no
// disc; kLeanPlain: spheres of the three common materials only (no disc, texture map or emitter; in trav_begin also no
// nested shells).
constexpr int kLeanNone = 0, kLeanNoDiscs = 1, kLeanPlain = 2;
template <int LEAN = kLeanNone>
__device__ __forceinline__ bool hit_part(const SceneDev& S, const Tables& Tb, const RenderParams& P, Path& p, const Best& hit,
                                         Surf& sf, float& Lr, float& Lg, float& Lb) {
#pragma clang fp contract(off)
    const float dxf = float(p.dx), dyf = float(p.dy), dzf = float(p.dz);
    if (hit.k < 0) {  // src/lib.rs:117-130
        const float t = fmaf(0.5f, dyf, 0.5f);
        Lr = p.tr * fmaf(t, -0.5f, 1.0f);  // (1-t)*1 + t*0.5
        Lg = p.tg * fmaf(t, -0.3f, 1.0f);  // (1-t)*1 + t*0.7
        Lb = p.tb;                         // (1-t)*1 + t*1
        return false;
    }
    const int32_t k = hit.k;
    // the three table reads go out together, ahead of the arithmetic that does not need them
    const int32_t kind_word = Tb.kind[k];
    const float4 m = Tb.mat[k];
    double4 s = make_double4(0.0, 0.0, 0.0, 1.0);
    constexpr bool PLAIN = LEAN == kLeanPlain, NODISC = LEAN != kLeanNone;
    if (NODISC || k < S.n_spheres) s = Tb.sph[k];  // (a disc's index lies beyond the sphere table, which may be empty)
    sf.hx = p.ox + hit.t * p.dx;  // scatter_loc = orig + t*dir (src/ray.rs:12-14)
    sf.hy = p.oy + hit.t * p.dy;
    sf.hz = p.oz + hit.t * p.dz;
    float lx, ly, lz;  // obj_relative_loc (src/lib.rs:106-113)
    if (NODISC || k < S.n_spheres) {
        const float rx = float(sf.hx - s.x), ry = float(sf.hy - s.y), rz = float(sf.hz - s.z);
        const float inv_r = fast_rcp(float(s.w));
        sf.nx = rx * inv_r; sf.ny = ry * inv_r; sf.nz = rz * inv_r;  // src/geometry.rs:123-125
        lx = rx; ly = ry; lz = rz;
    } else {
        const DiscDev& dd = S.discs[k - S.n_spheres];
        sf.nx = float(dd.nx); sf.ny = float(dd.ny); sf.nz = float(dd.nz);  // src/geometry.rs:141-143
        lx = float(sf.hx - dd.cx); ly = float(sf.hy - dd.cy); lz = float(sf.hz - dd.cz);
    }
    sf.kind = kind_word & 0xff;
    sf.w = m.w;
    sf.ddn = fmaf(dzf, sf.nz, fmaf(dyf, sf.ny, dxf * sf.nx));
    const bool glass = sf.kind == RT_MAT_DIELECTRIC;  // src/materials.rs:56-58
    float ar = glass ? 1.0f : m.x, ag = glass ? 1.0f : m.y, ab = glass ? 1.0f : m.z;
    if (!PLAIN && sf.kind > RT_MAT_DIELECTRIC) {  // the two rare kinds share one branch
        if (sf.kind == RT_MAT_EMITTER) {  // src/lib.rs:101-104: the throughput is dropped
            const float c = fabsf(sf.ddn);
            Lr = m.x * c; Lg = m.y * c; Lb = m.z * c;
            return false;
        }
        if (NODISC || k < S.n_spheres) normalize3(lx, ly, lz);  // spheres: normalised, discs: as is (src/lib.rs:108-111)
        texture_albedo(S, kind_word >> 8, lx, ly, lz, &ar, &ag, &ab);
    }
    p.tr *= ar; p.tg *= ag; p.tb *= ab;
    Lr = p.tr; Lg = p.tg; Lb = p.tb;  // src/lib.rs:134: depth exhausted returns the throughput
    return p.bounce < P.depth;
}

// Second half: Material::scatter (src/materials.rs:80-130) + Ray::new.  r0 = Philox block
// (pixel, sample, bounce, 0).  Returns false only when the bounded fuzz loop gives up.
// scatter_raw: the scattered direction before Ray::new normalises it.
__device__ __forceinline__ bool scatter_raw(const RenderParams& P, const Path& p, const Surf& sf, const uint4 r0, float& sx, float& sy,
                                            float& sz) {
#pragma clang fp contract(off)
    const float dxf = float(p.dx), dyf = float(p.dy), dzf = float(p.dz);
    const float nx = sf.nx, ny = sf.ny, nz = sf.nz, ddn = sf.ddn;
    if (sf.kind == RT_MAT_DIELECTRIC) {  // src/materials.rs:96-121
        const float inc_cos = ddn;
        const float px = fmaf(-inc_cos, nx, dxf), py = fmaf(-inc_cos, ny, dyf), pz = fmaf(-inc_cos, nz, dzf);
        const bool entering = __float_as_uint(inc_cos) >> 31;  // signum < 0 (also for -0.0)
        const float ratio = entering ? fast_rcp(sf.w) : sf.w;
        const float qx = ratio * px, qy = ratio * py, qz = ratio * pz;
        const float sin2 = fmaf(qz, qz, fmaf(qy, qy, qx * qx));
        bool reflect = sin2 > 1.0f;
        if (!reflect) {  // the Schlick draw counts only when sin2 <= 1 (`||` short-circuits)
            float r0f = (1.0f - ratio) * fast_rcp(1.0f + ratio);
            r0f = r0f * r0f;
            const float x = 1.0f - fabsf(inc_cos);
            const float x2 = x * x;
            const float reflectance = fmaf(1.0f - r0f, x2 * x2 * x, r0f);
            reflect = u01(r0.z) < reflectance;
        }
        if (reflect) {
            sx = fmaf(-inc_cos, nx, px); sy = fmaf(-inc_cos, ny, py); sz = fmaf(-inc_cos, nz, pz);
        } else {
            const float sc_cos = (entering ? -1.0f : 1.0f) * fast_sqrt(1.0f - sin2);
            sx = fmaf(sc_cos, nx, qx); sy = fmaf(sc_cos, ny, qy); sz = fmaf(sc_cos, nz, qz);
        }
    } else {
        float vx, vy, vz;
        unit_sphere(u01(r0.x), u01(r0.y), &vx, &vy, &vz);
        if (sf.kind == RT_MAT_METAL) {  // src/materials.rs:86-95, fuzzify :141-148
            const float m2 = -2.0f * ddn;
            const float rx = fmaf(m2, nx, dxf), ry = fmaf(m2, ny, dyf), rz = fmaf(m2, nz, dzf);
            sx = fmaf(sf.w, vx, rx); sy = fmaf(sf.w, vy, ry); sz = fmaf(sf.w, vz, rz);
            uint32_t attempt = 1;
            while (!(fmaf(sz, nz, fmaf(sy, ny, sx * nx)) > 0.0f)) {
                if (attempt >= uint32_t(kFuzzMaxAttempts)) return false;  // the reference would recurse without bound
                const uint4 r = philox4x32_10(p.pixel, p.sample, p.bounce, attempt, P.key0, P.key1);
                unit_sphere(u01(r.x), u01(r.y), &vx, &vy, &vz);
                sx = fmaf(sf.w, vx, rx); sy = fmaf(sf.w, vy, ry); sz = fmaf(sf.w, vz, rz);
                ++attempt;
            }
        } else {  // Diffuse and TextureMap, src/materials.rs:82-85, :122-125
            sx = nx + vx; sy = ny + vy; sz = nz + vz;
        }
    }
    return true;
}

__device__ __forceinline__ bool scatter_part(const RenderParams& P, Path& p, const Surf& sf, const uint4 r0) {
    float sx, sy, sz;
    if (!scatter_raw(P, p, sf, r0, sx, sy, sz)) return false;
    normalize3(sx, sy, sz);  // Ray::new
    p.ox = sf.hx; p.oy = sf.hy; p.oz = sf.hz;
    unit_f64(sx, sy, sz, &p.dx, &p.dy, &p.dz);
    ++p.bounce;
    return true;
}

// One segment of accel_raytrace / raytrace (src/lib.rs:93-131).  Returns true
// when the path goes on; otherwise (Lr,Lg,Lb) is the value the reference returns.
__device__ __forceinline__ bool advance(const SceneDev& S, const RenderParams& P, Path& p, int32_t* stk, float& Lr,
                                        float& Lg, float& Lb, int32_t* hit_id = nullptr, double* hit_t = nullptr) {
    const Best hit = closest_hit<false>(S, p.ox, p.oy, p.oz, p.dx, p.dy, p.dz, stk);
    if (hit_id) { *hit_id = hit.k < 0 ? -1 : hit.id; *hit_t = hit.t; }
    Surf sf;
    const Tables Tb = global_tables(S);
    if (!hit_part(S, Tb, P, p, hit, sf, Lr, Lg, Lb)) return false;
    if (!scatter_part(P, p, sf, philox4x32_10(p.pixel, p.sample, p.bounce, 0u, P.key0, P.key1))) {
        Lr = Lg = Lb = 0.0f;
        return false;
    }
    return true;
}

// Sum of one finished item: straight into the image when a pixel is a single item, else into the
// item's own slot of the partial-sum buffer (combine_kernel adds the slots in chunk order).
__device__ __forceinline__ void flush_item(const RenderParams& P, float* sum_rgb, float* partials, uint32_t pixel,
                                           uint32_t chunk, float ar, float ag, float ab) {
    const bool single = P.n_chunks == 1;  // then chunk is 0 and the slot formula below is the pixel's place in the image
    float* out = (single ? sum_rgb : partials) + 3 * (size_t(chunk) * P.n_pixels + pixel);
    if (single && (P.flags & RT_RENDER_ACCUMULATE)) { ar += out[0]; ag += out[1]; ab += out[2]; }
    out[0] = ar; out[1] = ag; out[2] = ab;
}

// Work hand-out.  A work item is one 8x8 pixel tile x one chunk of samples.  A wave draws an item from
// the global counter with ONE atomic and deals its 64 pixels to its lanes as they ask, so the counter
// sees one request per item instead of one per pixel (a single word serves only ~90 requests per
// microsecond).  ws = this wave's four LDS words {item, next undealt pixel of its tile, tile origin x | y << 16, chunk}; called by the
// lanes that need work.  Chunks keep the items small enough to balance the end of the launch: with one
// item per pixel and 100 samples each, a third of the launch was spent waiting for the last items.
// The four words are wave-uniform: they are read once into scalar registers, updated there and written back once.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
// The items are dealt from kCounters counters, each on a cache line of its own (word kCounterWord0 + 32 * c of the
// counter block; the block's first 256 bytes hold the diagnostic words): counter c hands out the items c, c + 8,
// c + 16 ...  A wave starts with the counter of its workgroup's number and moves on to the next when one runs dry.
// One word served all requests before; at ~90 per microsecond it was a fifth busy with 200 k items per frame and
// the waves queued for it.
constexpr uint32_t kCounters = 8, kCounterWord0 = 64, kCounterBlockBytes = 256 + kCounters * 128;
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return uint32_t(__builtin_amdgcn_readfirstlane(int32_t(v))); }

__device__ __forceinline__ void claim_pixels(bool& need, bool& done, uint32_t& pixel, uint32_t& xy, uint32_t& chunk, lds_u32* ws,
                                             uint32_t* work_counter, const RenderParams& P, const CamDev& C) {
    unsigned long long m = __ballot(need);
    if (m == 0) return;
    const uint32_t lane = __lane_id();
    uint32_t next = uniform(ws[1]), origin = uniform(ws[2]), item_chunk = uniform(ws[3]);
    // The usual case by far: the wave's current tile still holds a pixel for every lane that asks.  One pass, no
    // loop, only the tile's fill mark goes back to LDS.  (An exhausted wave keeps next = 64, so it never comes here.)
    const uint32_t asked = uint32_t(__popcll(m));
    if (next + asked <= 64u) {
        if (need) {
            const uint32_t in_tile = next + __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
            const uint32_t x = (origin & 0xffffu) + (in_tile & 7u);  // row-major inside the tile
            const uint32_t y = (origin >> 16) + (in_tile >> 3);
            if (x < C.W && y < C.H) { pixel = y * C.W + x; xy = x | (y << 16); chunk = item_chunk; need = false; }  // outside the frame: ask again
        }
        next += asked;
        ws[1] = next;
        m = __ballot(need);
        if (m == 0) return;
    }
    const uint32_t n_items = P.tiles_x * P.tiles_y * P.n_chunks;
    uint32_t source = uniform(ws[0]);  // counter in use | counters found dry in a row << 8; 0xffffffff = nothing left anywhere
    while (m) {
        if (next >= 64u) {  // the tile is dealt out: draw the next item and decode it once for its 64 pixels
            uint32_t item = 0xffffffffu;
            while (source != 0xffffffffu) {  // an exhausted wave skips this: later calls cost no atomic
                const uint32_t c = source & 0xffu;
                const int first = __ffsll((long long)m) - 1;
                uint32_t w = 0;
                if (int(lane) == first) w = atomicAdd(work_counter + kCounterWord0 + 32u * c, 1u);
                const uint32_t j = uint32_t(__builtin_amdgcn_readlane(int32_t(w), first));
                item = P.band_items ? c * P.band_items + j : j * kCounters + c;
                if (item < n_items && (!P.band_items || j < P.band_items)) { source = c; break; }
                const uint32_t dry = (source >> 8) + 1u;
                source = dry >= kCounters ? 0xffffffffu : (((c + 1u) % kCounters) | (dry << 8));
                item = 0xffffffffu;
            }
            if (item == 0xffffffffu) {  // nothing left
                if (need) { done = true; need = false; }
                break;
            }
            next = 0;
            const uint32_t tile = item / P.n_chunks;
            const uint32_t ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
            origin = (tx * 8u) | ((ty * 8u) << 16);
            item_chunk = item - tile * P.n_chunks;
        }
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
        const uint32_t avail = 64u - next;
        if (need && rank < avail) {
            const uint32_t in_tile = next + rank;
            const uint32_t x = (origin & 0xffffu) + (in_tile & 7u);
            const uint32_t y = (origin >> 16) + (in_tile >> 3);
            if (x < C.W && y < C.H) { pixel = y * C.W + x; xy = x | (y << 16); chunk = item_chunk; need = false; }
        }
        const uint32_t asked_now = uint32_t(__popcll(m));
        next += asked_now < avail ? asked_now : avail;
        m = __ballot(need);
    }
    ws[0] = source; ws[1] = next; ws[2] = origin; ws[3] = item_chunk;
}

// ---------------------------------------------------------------- kernels
// Persistent render kernel.  A lane owns one pixel at a time and walks its
// samples in order, so a pixel's sum is formed in sample order by one lane: no
// atomics on the image and bit-reproducible output.  A lane whose path ends
// starts the pixel's next sample at once; a lane whose pixel is finished draws
// the next pixel index from a global counter.
__global__ __launch_bounds__(kBlock) void render_kernel(SceneDev S, CamDev C, RenderParams P, uint32_t stack_depth,
                                                        float* __restrict__ sum_rgb, float* __restrict__ partials,
                                                        uint32_t* __restrict__ work_counter) {
    extern __shared__ int32_t stack_lds[];
    int32_t* stk = stack_lds + threadIdx.x;
    lds_u32* wave_state = (lds_u32*)(reinterpret_cast<uint32_t*>(stack_lds + stack_depth * kBlock) + 4 * (threadIdx.x >> 6));
    wave_state[0] = blockIdx.x % kCounters; wave_state[1] = 64; wave_state[2] = 0; wave_state[3] = 0;

    Path p;
    p.ox = p.oy = p.oz = 0.0;
    p.dx = p.dy = p.dz = 0.0;
    p.tr = p.tg = p.tb = 0.0f;
    p.pixel = 0; p.sample = 0; p.bounce = 0;
    bool have_pixel = false, alive = false, done = false;
    uint32_t next_sample = 0, item_end = 0, chunk = 0, pix_xy = 0;
    float ar = 0.0f, ag = 0.0f, ab = 0.0f;

    for (;;) {
        if (!alive && !done) {
            if (have_pixel && next_sample >= item_end) {  // item finished: write its sum
                flush_item(P, sum_rgb, partials, p.pixel, chunk, ar, ag, ab);
                have_pixel = false;
            }
            if (!have_pixel) {
                bool need = true;
                claim_pixels(need, done, p.pixel, pix_xy, chunk, wave_state, work_counter, P, C);
                if (!done) {
                    have_pixel = true;
                    next_sample = P.sample_begin + chunk * P.chunk_spp;
                    item_end = min(next_sample + P.chunk_spp, P.sample_end);
                    ar = ag = ab = 0.0f;
                }
            }
            if (have_pixel && next_sample < item_end) {
                if (P.depth == 0) {  // `for _ in 1..=0` never runs: the colour is the initial throughput
                    ar += 1.0f; ag += 1.0f; ab += 1.0f;
                } else {
                    start_path(C, P, p.pixel, next_sample, p);
                    alive = true;
                }
                ++next_sample;
            }
        }
        if (__all(done)) break;
        if (alive) {
            float Lr, Lg, Lb;
            if (!advance(S, P, p, stk, Lr, Lg, Lb)) {
                ar += Lr; ag += Lg; ab += Lb;
                alive = false;
            }
        }
    }
}

// ------------------------------------------------ wave-scheduled render kernel
// Same work assignment and arithmetic as render_kernel, different schedule.  A
// lane no longer walks "traverse, then shade" in lock step with its wave:
//   * a wave keeps stepping its BVH descents while at least kTravThreshold lanes
//     are still descending; lanes that finished wait for the shade phase;
//   * inside that, a run of inner-node steps (f32) lasts until fewer than
//     kInnerThreshold lanes sit on an inner node or kLeaveThreshold lanes have
//     left the inner nodes since the run began, then every lane parked on a leaf
//     runs the f64 sphere test together;
//   * the shade phase scatters / terminates / regenerates all waiting lanes at
//     once and sends them back into the descent.
// With LDS16 the whole scene is staged in LDS: boxes as f16 (rounded outward, so
// the descent stays conservative), child links as int32, spheres as f64.
constexpr int kBlockLds = 512;      // workgroup of the LDS-resident kernel: 2 per CU = 16 waves = 4 per SIMD (128 VGPRs)
constexpr int kTravThreshold = 12;
constexpr int kMaxWide = 4;         // spheres kept out of the tree at most (tested linearly per segment)
constexpr int kShellTable = 16;     // entries of the table of squared shell radii (a power of two above kSmallScene)
constexpr int kSmallScene = 12;     // a scene of up to this many spheres that holds a group of nested shells gets no tree
constexpr int kTinyScene = 5;       // up to this many spheres no scene gets a tree: every sphere is tested per segment
static_assert(kSmallScene < kShellTable && (kShellTable & (kShellTable - 1)) == 0, "the binary search over the shell table");
constexpr int kInnerThreshold = 8;
constexpr int kLeaveThreshold = 48;  // lanes that may leave the inner nodes before the wave turns to the leaves (measured flat from 40 up: with the short inner step a run is best ended by kInnerThreshold alone)

constexpr uint32_t kQ1Offset = 12288;  // LDS bytes from a node's first half to its second (inner_loop_lds16): room for 768 nodes
constexpr uint32_t kQ1OffsetFive = 8192;  // the same in the five-wave build of the kernel, which keeps less in LDS: 512 nodes
struct Node16 {       // 32 bytes: two 16-byte LDS reads
    uint16_t h[12];   // IEEE half bits; per axis x, y, z: centre of child 0, of child 1, half extent of child 0, of child 1
    int32_t child0, child1;
};
static_assert(sizeof(Node16) == 32, "Node16 must be 32 bytes");

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

typedef float float2v __attribute__((ext_vector_type(2)));
struct Trav {
    float inv[3], A[3], E[3];  // per axis: step / d, (base - origin) / d, eps / |d|  (see trav_begin)
    float2v inv_abs[3], A_E[3];  // the same as pairs {inv, |inv|} and {A, E}: operands of the packed FMAs of inner_loop_hbm
    float tmax;
    int32_t node;
    uint32_t sp;  // LDS byte address of the next free slot of this lane's stack column (slots are BLOCK * 4 bytes apart)
    int32_t top;  // top of the stack lives in a register: a pop does not wait for LDS
    Best best;
};

// sphere_intersect_f64 followed by consider() in one piece: the same operations in the same order (a negative
// discriminant gives a NaN root, which fails both comparisons); the arithmetic runs for every lane, only the rare
// comparisons are skipped.  Returns whether `best` changed.
// The test comes in two parts.  What depends on the sphere's CENTRE only (src/geometry.rs:105-107: oc, b, and the
// oc . oc of c) is computed by sphere_centre_terms; spheres that share a centre (the shells of a hollow glass ball,
// src/materials.rs:102-106 models glass by nesting them) share those values bit for bit, so the second and later
// ones skip that part (15 of the test's 50 instructions).
struct CentreTerms {
    double b, bb, oc2;  // 2 oc . d, its square, oc . oc
};
__device__ __forceinline__ CentreTerms sphere_centre_terms(const double4 s, const Path& p) {
#pragma clang fp contract(off)
    const double rx = p.ox - s.x, ry = p.oy - s.y, rz = p.oz - s.z;
    CentreTerms c;
    c.b = 2.0 * (rx * p.dx + ry * p.dy + rz * p.dz);
    c.oc2 = rx * rx + ry * ry + rz * rz;
    c.bb = c.b * c.b;
    return c;
}
// The comparisons short-circuit, so a wave in which no lane needs the second root's test or the tie rule skips them
// (measured against evaluating everything for every lane: shorter for the tree's leaves, and on nested shells too,
// 16.0 against 16.4 ms).  The tie rule (lowest hittable index wins) reads the two indices only where two parameters
// are equal.
__device__ __forceinline__ bool sphere_candidate_with(const CentreTerms ct, double radius, int32_t k, const int32_t* prim_id, bool enabled,
                                                      Best& best) {
    bool better;
    double t;
    {
#pragma clang fp contract(off)
        const double c = ct.oc2 - radius * radius;
        const double discrm = ct.bb - 4.0 * c;
        const double sq = sqrt_discriminant(discrm);
        const double t_smaller = -0.5 * (ct.b + sq);
        const double t_larger = t_smaller + sq;
        const bool near_ok = t_smaller > 0.0;
        const bool hit = !(discrm < 0.0) && (near_ok || t_larger > 1.0e-6);
        t = near_ok ? t_smaller : t_larger;
        better = enabled && hit && (best.k < 0 || t < best.t || (t == best.t && prim_id[k] < prim_id[best.k]));
    }
    best.t = better ? t : best.t;
    best.k = better ? k : best.k;
    return better;
}
__device__ __forceinline__ bool sphere_candidate(const double4 s, int32_t k, const int32_t* prim_id, const Path& p, bool enabled,
                                                 Best& best) {
    return sphere_candidate_with(sphere_centre_terms(s, p), s.w, k, prim_id, enabled, best);
}

// The render kernels' descent works in coordinates RELATIVE to a point of the scene (`base`), in units of `step`:
// a stored value q stands for the coordinate base + q * step, and the affine map is folded into the per-ray factors:
// a box with centre c and half extent h on an axis is crossed between t_c - w and t_c + w, where
// t_c = fma(c, step / d, (base - o) / d) and w = fma(h, |step / d|, eps / |d|), with base - o formed in f64 from the
// path's f64 origin.  So the f32
// roundings scale with distances inside the scene, not with the scene's distance from the coordinate origin, and a
// translated scene is descended exactly like the untranslated one.  GRID = true: q are the 16-bit grid indices of
// Node32 (base = the grid's low corner); GRID = false: q are the f16 values of Node16 (base = centre of the tree's
// box, step = a power of two that brings the half extent to about 1024, inside f16's normal range whatever the
// scene's scale).
// TREE = false: the scene has no tree (a handful of spheres, all tested here): there is no descent to set up.
// LEAN: see hit_part.
template <bool GRID, bool TREE = true, int LEAN = kLeanNone>
__device__ __forceinline__ void trav_begin(const SceneDev& S, const Tables& Tb, const Path& p, Trav& T, uint32_t slot1_addr, int32_t root) {
    T.best.t = 0.0;
    T.best.k = -1;
    T.best.id = 0x7fffffff;
    T.sp = slot1_addr;
    T.top = kLeafSentinel;
    constexpr bool PLAIN = LEAN == kLeanPlain, NODISC = LEAN != kLeanNone;
    for (int32_t i = 0; !NODISC && i < S.n_discs; ++i) {
        double t;
        if (disc_intersect_f64(S.discs[i], p.ox, p.oy, p.oz, p.dx, p.dy, p.dz, &t)) {
            const int32_t k = S.n_spheres + i;  // consider() with the indices read only on a tie
            if (T.best.k < 0 || t < T.best.t || (t == T.best.t && S.prim_id[k] < S.prim_id[T.best.k])) { T.best.t = t; T.best.k = k; }
        }
    }
    // a sphere parameter is always positive: nothing can beat a non-positive disc hit
    const bool spheres_matter = NODISC || !(T.best.k >= 0 && T.best.t <= 0.0);
    // Nested shells (SceneDev::shell_first): of m concentric spheres a segment can only meet first one of four
    // (and which three of them suffice is decided per lane below).
    // Count the shells its origin lies outside of (c > 0 in src/geometry.rs:107; they are the `outside` smallest):
    //   * the largest of those: the ray enters it before any smaller one (t_smaller falls as the radius grows,
    //     rounding included: every operation on the way from the radius to t_smaller is monotonic), and when it is
    //     missed the smaller ones are missed too (so is the discriminant);
    //   * the next smaller one, for an origin ON that largest shell (c > 0 by rounding): its near root may come out
    //     as 0 or below, the far root is then taken (src/geometry.rs:115-120) and the shell inside it comes first;
    //   * the smallest shell the origin lies inside of (c <= 0): the ray leaves that one first (t_larger grows with
    //     the radius);
    //   * the next larger one, when that smallest is refused by the 1e-6 rule of src/geometry.rs:118 (an origin ON it,
    //     leaving).
    // The radii are separated by 1e-4 of themselves and by 4e-6 or more (rt_scene_create), so an origin is on at most
    // one shell, and the separation is far above the rounding of the roots for origins within a thousand radii of the
    // smallest shell; a wave with an origin farther out than that tests every shell.  (tests/test_shell_rule_model.py
    // restates this arithmetic in numpy and checks the rule against testing all shells on rays from origins on, next
    // to and between shells, grazing directions included; the parity tests hold this code to the per-path kernel,
    // which tests every sphere.)
    const int32_t sh0 = S.shell_first, shn = S.shell_count;
    bool shortcut = !PLAIN && shn > 0;
    CentreTerms ct{0.0, 0.0, 0.0};
    if (shortcut) {
        ct = sphere_centre_terms(Tb.sph[S.n_tree + sh0], p);
        shortcut = !__any(!(ct.oc2 <= S.shell_far2));  // also false for a NaN origin
    }
    if (shortcut) {
        // c = oc . oc - r * r is positive exactly when oc . oc > r * r (a difference of two doubles has the sign of the
        // exact difference), so the count needs one comparison per shell against the table of squares
        // The table is ascending and padded with +infinity to kShellTable entries, so the count is a binary search of
        // four probes whatever the number of shells.
        int32_t outside = 0;  // shells with c > 0: the shells [0, outside) do not hold the origin
        double r2_last = -1.0e300;  // the largest of their squares
#pragma unroll
        for (int32_t step = kShellTable / 2; step >= 1; step >>= 1) {
            const double r2 = Tb.shell_r2[outside + step - 1];
            const bool out = ct.oc2 > r2;  // then every entry before it is below oc . oc too
            outside += out ? step : 0;
            r2_last = out ? r2 : r2_last;
        }
        const double c_last = ct.oc2 - r2_last;  // c of the largest of them (huge when there is none)
        // Three of the four in a row: with the origin on the largest "outside" shell (c within 1e-9 of oc . oc: a
        // near root collapses only for c within rounding, ten million times closer) the run starts one shell further
        // in, and the shell after the enclosing one cannot be needed (the enclosing one is a whole gap away, so the
        // 1e-6 rule accepts it); otherwise the near root of the largest outside shell is safe and the run starts there.
        const int32_t first = outside - ((c_last < 1.0e-9 * ct.oc2) ? 2 : 1);
        // Two of those three, by the direction: a segment that moves towards the centre (b < 0) cannot need the last
        // of the run (it meets the enclosing shell's far side at the latest, |b| or more away), one that moves away
        // from it (b > 0) cannot meet the first (a shell its origin is outside of by a margin lies behind it).  Near
        // the tangent (|cos| below 5e-4, or a chord the 1e-6 rule could refuse) a grazed shell may be missed by
        // rounding and the next one matters: a wave with such a lane tests all three.
        const bool steep = ct.bb > 1.6e-11 && ct.bb > 1.0e-6 * ct.oc2;
        const bool two = !__any(!steep);
        const int32_t run_first = first + ((two && ct.b > 0.0) ? 1 : 0);
        auto test_shell = [&](int32_t i) {
            const bool valid = i >= 0 && i < shn;
            const int32_t k = S.n_tree + sh0 + (valid ? i : 0);
            sphere_candidate_with(ct, Tb.sph[k].w, k, Tb.prim_id, spheres_matter && valid, T.best);
        };
        test_shell(run_first);  // written out so that the two tests' reads and arithmetic can overlap
        test_shell(run_first + 1);
        if (!two) test_shell(run_first + 2);
    }
    for (int32_t i = 0; i < S.n_wide; ++i) {
        if (shortcut && i == sh0) { i += shn - 1; continue; }  // the shells are done
        const int32_t k = S.n_tree + i;
        const double4 s = Tb.sph[k];
        if (i == 0 || !((S.wide_same_centre >> i) & 1u)) ct = sphere_centre_terms(s, p);  // wave-uniform: a fact of the scene
        sphere_candidate_with(ct, s.w, k, Tb.prim_id, spheres_matter, T.best);
    }
    T.node = (!TREE || S.n_nodes == 0 || !spheres_matter) ? kLeafSentinel : root;
    if (!TREE) return;
    const double o64[3] = {p.ox, p.oy, p.oz};
    const float df[3] = {float(p.dx), float(p.dy), float(p.dz)};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float d = df[a];
        // 1 / d in units of the scene's parameter scale: every parameter that can matter then lies in [0, 1], and the
        // clamp modifier of the instruction that forms the z entry parameters does the work of max(., 0)
        const float i = ((fabsf(d) >= 1.0e-18f) ? fast_rcp(d) : copysignf(1.0e18f, d)) * S.t_scale;
        const float base = GRID ? S.qbase[a] : S.hbase[a], step = GRID ? S.qstep[a] : S.hstep[a];
        const float rel = float(double(base) - o64[a]);  // base - origin, rounded once
        T.inv[a] = step * i;
        T.A[a] = rel * i;
        T.E[a] = S.eps_rel * fabsf(i);
        if (GRID) {
            T.inv_abs[a] = float2v{T.inv[a], fabsf(T.inv[a])};
            T.A_E[a] = float2v{T.A[a], T.E[a]};
        }
    }
    T.tmax = (T.best.k >= 0) ? float(T.best.t) * (1.0f + 4.0e-7f) * S.t_scale : 3.0e38f;
}

// Entry and exit parameters of the two child boxes of a node, from their centres c and half extents h (node units).
__device__ __forceinline__ void slab2(const Trav& T, const float c0[3], const float c1[3], const float h0[3], const float h1[3],
                                      float* en0, float* ex0, float* en1, float* ex1) {
    float n0[3], f0[3], n1[3], f1[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float ai = fabsf(T.inv[a]);
        const float tc0 = fmaf(c0[a], T.inv[a], T.A[a]), w0 = fmaf(h0[a], ai, T.E[a]);
        const float tc1 = fmaf(c1[a], T.inv[a], T.A[a]), w1 = fmaf(h1[a], ai, T.E[a]);
        n0[a] = tc0 - w0; f0[a] = tc0 + w0;
        n1[a] = tc1 - w1; f1[a] = tc1 + w1;
    }
    // entry on z clamped to [0, 1] like the assembly loops' clamp modifier (1 is beyond every box: see SceneDev::t_scale)
    *en0 = fmaxf(fmaxf(n0[0], n0[1]), fminf(fmaxf(n0[2], 0.0f), 1.0f));
    *ex0 = fminf(fminf(f0[0], f0[1]), fminf(f0[2], T.tmax));
    *en1 = fmaxf(fmaxf(n1[0], n1[1]), fminf(fmaxf(n1[2], 0.0f), 1.0f));
    *ex1 = fminf(fminf(f1[0], f1[1]), fminf(f1[2], T.tmax));
}

// The stack keeps its top in T.top and the rest in LDS (this lane's column: slot `level` at column base +
// level * BLOCK * 4 bytes; T.sp is the byte address of the next free slot).  A traversal starts with T.sp at slot 1
// and top == kLeafSentinel above slot 0, which the kernel fills with kLeafSentinel once and no push ever writes
// (pushes store at slot >= 1): popping the empty stack yields the sentinel, which ends the traversal, without a
// test of sp, and the lane does not pop again before trav_begin resets it.
typedef __attribute__((address_space(3))) int32_t lds_i32;
template <int BLOCK>
__device__ __forceinline__ void pop(Trav& T) {
    T.node = T.top;
    T.sp -= uint32_t(BLOCK * sizeof(int32_t));
    T.top = *(lds_i32*)(uintptr_t)T.sp;
}

template <int BLOCK>
__device__ __forceinline__ void descend(Trav& T, float en0, float ex0, float en1, float ex1, int32_t c0, int32_t c1) {
    const bool h0 = en0 <= ex0, h1 = en1 <= ex1;
    const bool both = h0 && h1, swap = en1 < en0;
    if (both) *(lds_i32*)(uintptr_t)T.sp = T.top;
    if (h0 || h1) {
        T.node = (h0 && !(h1 && swap)) ? c0 : c1;
        T.top = both ? (swap ? c0 : c1) : T.top;
        T.sp += both ? uint32_t(BLOCK * sizeof(int32_t)) : 0u;
    } else {
        pop<BLOCK>(T);
    }
}

// LDS byte address of an object in dynamic shared memory
__device__ __forceinline__ uint32_t lds_address(const void* p) {
    return uint32_t(uintptr_t((__attribute__((address_space(3))) const void*)p));
}

template <bool LDS16, int BLOCK>
__device__ __forceinline__ void inner_step(const SceneDev& S, const uint4* lds_q0, const uint4* lds_q1, Trav& T) {
    float en0, ex0, en1, ex1;
    int32_t c0, c1;
    // dwords: a.x = centres x (child 0 | child 1 << 16), a.y = half extents x, a.z / a.w = the same for y, b.x / b.y for z
    uint4 a, b;
    if (LDS16) {  // an inner link of the staged tree is the LDS address of the node's first half
        const uint32_t idx = (uint32_t(T.node) - lds_address(lds_q0)) >> 4;
        a = lds_q0[idx];
        b = lds_q1[idx];
    }
    else { const uint4* q = reinterpret_cast<const uint4*>(S.nodes32 + T.node); a = q[0]; b = q[1]; }
    const uint32_t cw[3] = {a.x, a.z, b.x}, hw[3] = {a.y, a.w, b.y};
    float ce0[3], ce1[3], he0[3], he1[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (LDS16) {
            const half2_t c = __builtin_bit_cast(half2_t, cw[k]), h = __builtin_bit_cast(half2_t, hw[k]);
            ce0[k] = float(c.x); ce1[k] = float(c.y); he0[k] = float(h.x); he1[k] = float(h.y);
        } else {
            ce0[k] = float(cw[k] & 0xffffu); ce1[k] = float(cw[k] >> 16); he0[k] = float(hw[k] & 0xffffu); he1[k] = float(hw[k] >> 16);
        }
    }
    slab2(T, ce0, ce1, he0, he1, &en0, &ex0, &en1, &ex1);
    c0 = int32_t(b.z);
    c1 = int32_t(b.w);
    descend<BLOCK>(T, en0, ex0, en1, ex1, c0, c1);
}


// The inner-node loop of the LDS-resident f16 tree, written out in gfx950 assembly:
//     while (at least `thr` lanes of the wave stand on an inner node, or some do and none waits at a leaf)
//         those lanes take one inner_step
// It is the same arithmetic as slab2/descend above (f16 bounds fed to v_fma_mix_f32, the lane's stack column in
// LDS with its top in a register).  Every instruction of a step costs about two SIMD cycles whatever its kind
// (tools/microbench/inner_step.hip), so the loop is built for the fewest instructions:
//   * an inner link IS the LDS byte address of the node's first half (the kernel adds the array's address to the
//     links while it stages the nodes) and the second half sits kQ1Offset bytes behind it: a fetch is two ds_read_b128
//     and no address arithmetic;
//   * the entry parameters on z are formed with the clamp modifier (parameters are in units of SceneDev::t_scale, so
//     [0, 1] covers every segment), which is max(., 0) for free;
//   * the lane masks go straight into EXEC where they are needed once (s_and / s_andn2 with EXEC as destination);
//   * the loop's back edge is the conditional branch of the "enough lanes left" test.
// Per step: 2 reads + wait, 18 slab + 2 min + 4 max3 / min3, 3 compares + 3 scalar + 3 selects, 7 for the stack,
// 7 for the test and the masks = 49 (an earlier form of the loop: 56).
// Fixed registers: v112-v127 and s88-s99 are scratch of this block (the 128-bit node halves need named registers,
// inline-asm operands cannot name the dwords of a tuple).  EXEC is narrowed inside and restored before leaving.
#define RT_STR2(x) #x
#define RT_STR(x) RT_STR2(x)
// (The loop is a macro of the block's register base and of the distance to a node's second half, both as string
// literals, because the kernel is built twice: see render_kernel_ws.  rN in the remarks = register base + N.)
#define RT_LDS_LOOP(B, Q1) \
  /* which lanes stand on an inner node, and whether to run at all */ \
        "v_cmp_lt_i32_e32 vcc, -1, %[node]\n\t" \
        "s_mov_b64 s[88:89], exec\n\t" \
        "s_bcnt1_i32_b64 %[cnt], vcc\n\t" \
        "s_cmp_ge_i32 %[cnt], %[thr]\n\t" \
        "s_cbranch_scc1 4f\n\t" \
        "s_cmp_eq_u32 %[cnt], 0\n\t" \
        "s_cbranch_scc1 9f\n\t" \
        "v_cmp_lt_u32_e64 s[90:91], %[sentinel], %[node]\n\t"  /* as unsigned: above the sentinel = a leaf */ \
        "s_cmp_lg_u64 s[90:91], 0\n\t" \
        "s_cbranch_scc1 9f\n\t" \
        "4:\n\t" \
        "s_mov_b64 exec, vcc\n\t" \
        "ds_read_b128 v[" B "+8:" B "+11], %[node]\n\tds_read_b128 v[" B "+12:" B "+15], %[node] offset:" Q1 "\n\t" \
        "1:\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
  /* x: r8 = centres (child 0 | child 1), r9 = half extents */ \
        "v_fma_mix_f32 v[" B "+0], v[" B "+8], %[ix], %[ax] op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t" \
        "v_fma_mix_f32 v[" B "+1], v[" B "+8], %[ix], %[ax] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t" \
        "v_fma_mix_f32 v[" B "+2], v[" B "+9], |%[ix]|, %[ex] op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t" \
        "v_fma_mix_f32 v[" B "+3], v[" B "+9], |%[ix]|, %[ex] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t" \
        "v_pk_add_f32 v[" B "+4:" B "+5], v[" B "+0:" B "+1], v[" B "+2:" B "+3] neg_lo:[0,1] neg_hi:[0,1]\n\t"  /* near parameters */ \
        "v_pk_add_f32 v[" B "+0:" B "+1], v[" B "+0:" B "+1], v[" B "+2:" B "+3]\n\t"  /* far parameters */ \
  /* y: r10 = centres (child 0 | child 1), r11 = half extents */ \
        "v_fma_mix_f32 v[" B "+2], v[" B "+10], %[iy], %[ay] op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t" \
        "v_fma_mix_f32 v[" B "+3], v[" B "+10], %[iy], %[ay] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t" \
        "v_fma_mix_f32 v[" B "+8], v[" B "+11], |%[iy]|, %[ey] op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t" \
        "v_fma_mix_f32 v[" B "+9], v[" B "+11], |%[iy]|, %[ey] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t" \
        "v_pk_add_f32 v[" B "+6:" B "+7], v[" B "+2:" B "+3], v[" B "+8:" B "+9] neg_lo:[0,1] neg_hi:[0,1]\n\t"  /* near parameters */ \
        "v_pk_add_f32 v[" B "+2:" B "+3], v[" B "+2:" B "+3], v[" B "+8:" B "+9]\n\t"  /* far parameters */ \
  /* z: r12 = centres (child 0 | child 1), r13 = half extents */ \
        "v_fma_mix_f32 v[" B "+8], v[" B "+12], %[iz], %[az] op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t" \
        "v_fma_mix_f32 v[" B "+9], v[" B "+12], %[iz], %[az] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t" \
        "v_fma_mix_f32 v[" B "+10], v[" B "+13], |%[iz]|, %[ez] op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t" \
        "v_fma_mix_f32 v[" B "+11], v[" B "+13], |%[iz]|, %[ez] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t" \
        "v_pk_add_f32 v[" B "+12:" B "+13], v[" B "+8:" B "+9], v[" B "+10:" B "+11]\n\t"  /* far parameters */ \
        "v_pk_add_f32 v[" B "+10:" B "+11], v[" B "+8:" B "+9], v[" B "+10:" B "+11] neg_lo:[0,1] neg_hi:[0,1] clamp\n\t"  /* near parameters, clamped to [0, 1]: max(., 0) for free */ \
  /* near x / y / z = r4,117 / r6,119 / r10,123; far = r0,113 / r2,115 / r12,125 (child 0, child 1) */ \
        "v_min_f32_e32 v[" B "+12], v[" B "+12], %[tmax]\n\t" \
        "v_min_f32_e32 v[" B "+13], v[" B "+13], %[tmax]\n\t" \
        "v_max3_f32 v[" B "+4], v[" B "+4], v[" B "+6], v[" B "+10]\n\t"  /* entry parameter of child 0 */ \
        "v_min3_f32 v[" B "+0], v[" B "+0], v[" B "+2], v[" B "+12]\n\t"  /* exit parameter of child 0 */ \
        "v_max3_f32 v[" B "+5], v[" B "+5], v[" B "+7], v[" B "+11]\n\t"  /* entry parameter of child 1 */ \
        "v_min3_f32 v[" B "+1], v[" B "+1], v[" B "+3], v[" B "+13]\n\t"  /* exit parameter of child 1 */ \
  /* r14, r15 = child links c0, c1 */ \
        "v_cmp_le_f32_e64 s[90:91], v[" B "+4], v[" B "+0]\n\t"  /* h0 */ \
        "v_cmp_le_f32_e64 s[92:93], v[" B "+5], v[" B "+1]\n\t"  /* h1 */ \
        "v_cmp_lt_f32_e64 s[94:95], v[" B "+5], v[" B "+4]\n\t"  /* child 1 is entered first */ \
        "s_or_b64 s[98:99], s[90:91], s[92:93]\n\t"  /* any */ \
        "s_and_b64 s[94:95], s[94:95], s[92:93]\n\t" \
        "s_andn2_b64 s[94:95], s[90:91], s[94:95]\n\t"  /* go to c0: h0 and not (h1 and child 1 first) */ \
        "v_cndmask_b32_e64 v[" B "+7], v[" B "+15], v[" B "+14], s[94:95]\n\t" \
        "v_cndmask_b32_e64 v[" B "+6], v[" B "+14], v[" B "+15], s[94:95]\n\t"  /* the other child, pushed where both are hit */ \
        "v_cndmask_b32_e64 %[node], %[top], v[" B "+7], s[98:99]\n\t"  /* no child hit: go on with the stack top */ \
  /* the stack: lanes without a hit child pop, lanes with two push the other one */ \
        "s_andn2_b64 exec, exec, s[98:99]\n\t" \
        "v_subrev_u32_e32 %[sp], %[stride], %[sp]\n\t" \
        "ds_read_b32 %[top], %[sp]\n\t" \
        "s_and_b64 exec, s[90:91], s[92:93]\n\t" \
        "ds_write_b32 %[sp], %[top]\n\t" \
        "v_add_u32_e32 %[sp], %[stride], %[sp]\n\t" \
        "v_mov_b32_e32 %[top], v[" B "+6]\n\t" \
  /* who goes on (lanes that left the loop earlier stand on a leaf or on the sentinel: negative either way) */ \
        "s_mov_b64 exec, s[88:89]\n\t" \
        "v_cmp_lt_i32_e32 vcc, -1, %[node]\n\t" \
        "s_bcnt1_i32_b64 %[cnt], vcc\n\t" \
        "s_mov_b64 exec, vcc\n\t" \
        "ds_read_b128 v[" B "+8:" B "+11], %[node]\n\tds_read_b128 v[" B "+12:" B "+15], %[node] offset:" Q1 "\n\t" \
        "s_cmp_ge_i32 %[cnt], %[thr]\n\t" \
        "s_cbranch_scc1 1b\n\t" \
  /* fewer than `thr` lanes go on: leave when none does, or when some lane of the wave waits at a leaf */ \
        "s_cmp_eq_u32 %[cnt], 0\n\t" \
        "s_cbranch_scc1 9f\n\t" \
        "s_mov_b64 exec, s[88:89]\n\t" \
        "v_cmp_lt_u32_e64 s[90:91], %[sentinel], %[node]\n\t" \
        "s_mov_b64 exec, vcc\n\t" \
        "s_cmp_lg_u64 s[90:91], 0\n\t" \
        "s_cbranch_scc0 1b\n\t" \
        "9:\n\t" \
        "s_mov_b64 exec, s[88:89]\n\t" \
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
template <int BLOCK, bool LOW>
__device__ __forceinline__ void inner_loop_lds16(Trav& T, int32_t thr) {
    static_assert(BLOCK == 512 || BLOCK == 256, "stack column stride");
    // the "s" operands must BE scalar registers: the constraint alone does not move a value the compiler holds in a
    // vector register (it would print v13 into an s_ instruction)
    thr = __builtin_amdgcn_readfirstlane(max(thr, 1));  // with 0 the loop would never see "no lane left"
    int32_t lanes_on_inner;
    static_assert(kQ1Offset == 12288 && kQ1OffsetFive == 8192, "the offsets are spelled out below");
    if constexpr (LOW) {
        asm volatile(RT_LDS_LOOP("80", "8192")
        : [node] "+v"(T.node), [sp] "+v"(T.sp), [top] "+v"(T.top), [cnt] "=&s"(lanes_on_inner)
        : [ix] "v"(T.inv[0]), [iy] "v"(T.inv[1]), [iz] "v"(T.inv[2]), [ax] "v"(T.A[0]), [ay] "v"(T.A[1]),
          [az] "v"(T.A[2]), [ex] "v"(T.E[0]), [ey] "v"(T.E[1]), [ez] "v"(T.E[2]), [tmax] "v"(T.tmax),
          [thr] "s"(thr), [sentinel] "s"(kLeafSentinel), [stride] "s"(uint32_t(BLOCK * sizeof(int32_t)))
        : "memory", "vcc", "scc", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "s88", "s89", "s90", "s91",
          "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99");
    } else {
        asm volatile(RT_LDS_LOOP("112", "12288")
        : [node] "+v"(T.node), [sp] "+v"(T.sp), [top] "+v"(T.top), [cnt] "=&s"(lanes_on_inner)
        : [ix] "v"(T.inv[0]), [iy] "v"(T.inv[1]), [iz] "v"(T.inv[2]), [ax] "v"(T.A[0]), [ay] "v"(T.A[1]),
          [az] "v"(T.A[2]), [ex] "v"(T.E[0]), [ey] "v"(T.E[1]), [ez] "v"(T.E[2]), [tmax] "v"(T.tmax),
          [thr] "s"(thr), [sentinel] "s"(kLeafSentinel), [stride] "s"(uint32_t(BLOCK * sizeof(int32_t)))
        : "memory", "vcc", "scc", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "s88", "s89", "s90", "s91",
          "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99");
    }

}

#define RT_HBM_LOOP(B) \
  /* which lanes stand on an inner node, and whether to run at all */ \
        "v_cmp_lt_i32_e32 vcc, -1, %[node]\n\t" \
        "s_mov_b64 s[88:89], exec\n\t" \
        "s_bcnt1_i32_b64 %[cnt], vcc\n\t" \
        "s_cmp_ge_i32 %[cnt], %[thr]\n\t" \
        "s_cbranch_scc1 4f\n\t" \
        "s_cmp_eq_u32 %[cnt], 0\n\t" \
        "s_cbranch_scc1 9f\n\t" \
        "v_cmp_lt_u32_e64 s[90:91], %[sentinel], %[node]\n\t"  /* as unsigned: above the sentinel = a leaf */ \
        "s_cmp_lg_u64 s[90:91], 0\n\t" \
        "s_cbranch_scc1 9f\n\t" \
        "4:\n\t" \
        "s_mov_b64 exec, vcc\n\t" \
        "v_lshlrev_b32_e32 v[" B "+0], 5, %[node]\n\t" \
        "global_load_dwordx4 v[" B "+8:" B "+11], v[" B "+0], %[base]\n\t" \
        "global_load_dwordx4 v[" B "+12:" B "+15], v[" B "+0], %[base] offset:16\n\t" \
        "1:\n\t" \
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t" \
  /* x: register 8 of the block = centres (child 0 | child 1), register 9 = half extents */ \
        "v_cvt_f32_u32_sdwa v[" B "+0], v[" B "+8] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\t" \
        "v_cvt_f32_u32_sdwa v[" B "+1], v[" B "+8] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t" \
        "v_cvt_f32_u32_sdwa v[" B "+2], v[" B "+9] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\t" \
        "v_cvt_f32_u32_sdwa v[" B "+3], v[" B "+9] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t" \
        "v_pk_fma_f32 v[" B "+0:" B "+1], v[" B "+0:" B "+1], %[px], %[qx] op_sel_hi:[1,0,0]\n\t"  /* centres * inv + A, both children */ \
        "v_pk_fma_f32 v[" B "+2:" B "+3], v[" B "+2:" B "+3], %[px], %[qx] op_sel:[0,1,1] op_sel_hi:[1,1,1]\n\t"  /* half extents * |inv| + E */ \
        "v_pk_add_f32 v[" B "+4:" B "+5], v[" B "+0:" B "+1], v[" B "+2:" B "+3] neg_lo:[0,1] neg_hi:[0,1]\n\t"  /* near parameters */ \
        "v_pk_add_f32 v[" B "+0:" B "+1], v[" B "+0:" B "+1], v[" B "+2:" B "+3]\n\t"  /* far parameters */ \
  /* y: registers 10 and 11 */ \
        "v_cvt_f32_u32_sdwa v[" B "+2], v[" B "+10] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\t" \
        "v_cvt_f32_u32_sdwa v[" B "+3], v[" B "+10] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t" \
        "v_cvt_f32_u32_sdwa v[" B "+8], v[" B "+11] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\t" \
        "v_cvt_f32_u32_sdwa v[" B "+9], v[" B "+11] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t" \
        "v_pk_fma_f32 v[" B "+2:" B "+3], v[" B "+2:" B "+3], %[py], %[qy] op_sel_hi:[1,0,0]\n\t"  /* centres * inv + A, both children */ \
        "v_pk_fma_f32 v[" B "+8:" B "+9], v[" B "+8:" B "+9], %[py], %[qy] op_sel:[0,1,1] op_sel_hi:[1,1,1]\n\t"  /* half extents * |inv| + E */ \
        "v_pk_add_f32 v[" B "+6:" B "+7], v[" B "+2:" B "+3], v[" B "+8:" B "+9] neg_lo:[0,1] neg_hi:[0,1]\n\t"  /* near parameters */ \
        "v_pk_add_f32 v[" B "+2:" B "+3], v[" B "+2:" B "+3], v[" B "+8:" B "+9]\n\t"  /* far parameters */ \
  /* z: registers 12 and 13 */ \
        "v_cvt_f32_u32_sdwa v[" B "+8], v[" B "+12] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\t" \
        "v_cvt_f32_u32_sdwa v[" B "+9], v[" B "+12] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t" \
        "v_cvt_f32_u32_sdwa v[" B "+10], v[" B "+13] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\t" \
        "v_cvt_f32_u32_sdwa v[" B "+11], v[" B "+13] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t" \
        "v_pk_fma_f32 v[" B "+8:" B "+9], v[" B "+8:" B "+9], %[pz], %[qz] op_sel_hi:[1,0,0]\n\t"  /* centres * inv + A, both children */ \
        "v_pk_fma_f32 v[" B "+10:" B "+11], v[" B "+10:" B "+11], %[pz], %[qz] op_sel:[0,1,1] op_sel_hi:[1,1,1]\n\t"  /* half extents * |inv| + E */ \
        "v_pk_add_f32 v[" B "+12:" B "+13], v[" B "+8:" B "+9], v[" B "+10:" B "+11]\n\t"  /* far parameters */ \
        "v_pk_add_f32 v[" B "+10:" B "+11], v[" B "+8:" B "+9], v[" B "+10:" B "+11] neg_lo:[0,1] neg_hi:[0,1] clamp\n\t"  /* near parameters, clamped to [0, 1]: max(., 0) for free */ \
  /* near x / y / z = registers 4,5 / 6,7 / 10,11; far = 0,1 / 2,3 / 12,13 (child 0, child 1) */ \
        "v_min_f32_e32 v[" B "+12], v[" B "+12], %[tmax]\n\t" \
        "v_min_f32_e32 v[" B "+13], v[" B "+13], %[tmax]\n\t" \
        "v_max3_f32 v[" B "+4], v[" B "+4], v[" B "+6], v[" B "+10]\n\t"  /* entry parameter of child 0 */ \
        "v_min3_f32 v[" B "+0], v[" B "+0], v[" B "+2], v[" B "+12]\n\t"  /* exit parameter of child 0 */ \
        "v_max3_f32 v[" B "+5], v[" B "+5], v[" B "+7], v[" B "+11]\n\t"  /* entry parameter of child 1 */ \
        "v_min3_f32 v[" B "+1], v[" B "+1], v[" B "+3], v[" B "+13]\n\t"  /* exit parameter of child 1 */ \
  /* registers 14, 15 = child links c0, c1 */ \
        "v_cmp_le_f32_e64 s[90:91], v[" B "+4], v[" B "+0]\n\t"  /* h0 */ \
        "v_cmp_le_f32_e64 s[92:93], v[" B "+5], v[" B "+1]\n\t"  /* h1 */ \
        "v_cmp_lt_f32_e64 s[94:95], v[" B "+5], v[" B "+4]\n\t"  /* child 1 is entered first */ \
        "s_and_b64 s[96:97], s[90:91], s[92:93]\n\t"  /* both */ \
        "s_or_b64 s[98:99], s[90:91], s[92:93]\n\t"  /* any */ \
        "s_and_b64 s[94:95], s[94:95], s[92:93]\n\t" \
        "s_andn2_b64 s[94:95], s[90:91], s[94:95]\n\t"  /* go to c0: h0 and not (h1 and child 1 first) */ \
        "s_andn2_b64 s[98:99], exec, s[98:99]\n\t"  /* none */ \
        "v_cndmask_b32_e64 %[node], v[" B "+15], v[" B "+14], s[94:95]\n\t" \
        "v_cndmask_b32_e64 v[" B "+6], v[" B "+14], v[" B "+15], s[94:95]\n\t"  /* the other child, pushed where both are hit */ \
        "v_cndmask_b32_e64 %[node], %[node], %[top], s[98:99]\n\t"  /* no child hit: go on with the stack top */ \
  /* every lane's next node is known: decide whether the loop goes on, and if so start the next fetch before */ \
  /* the stack is brought up to date */ \
        "v_cmp_lt_i32_e32 vcc, -1, %[node]\n\t" \
        "s_bcnt1_i32_b64 %[cnt], vcc\n\t" \
        "s_cmp_ge_i32 %[cnt], %[thr]\n\t" \
        "s_cbranch_scc0 5f\n\t" \
        "6:\n\t" \
        "s_mov_b64 exec, vcc\n\t" \
        "v_lshlrev_b32_e32 v[" B "+0], 5, %[node]\n\t" \
        "global_load_dwordx4 v[" B "+8:" B "+11], v[" B "+0], %[base]\n\t" \
        "global_load_dwordx4 v[" B "+12:" B "+15], v[" B "+0], %[base] offset:16\n\t" \
        "s_mov_b64 exec, s[96:97]\n\t"  /* push */ \
        "ds_write_b32 %[sp], %[top]\n\t" \
        "v_add_u32_e32 %[sp], %[stride], %[sp]\n\t" \
        "v_mov_b32_e32 %[top], v[" B "+6]\n\t" \
        "s_mov_b64 exec, s[98:99]\n\t"  /* pop */ \
        "v_subrev_u32_e32 %[sp], %[stride], %[sp]\n\t" \
        "ds_read_b32 %[top], %[sp]\n\t" \
        "s_mov_b64 exec, vcc\n\t" \
        "s_branch 1b\n\t" \
        "5:\n\t"  /* fewer than `thr` lanes go on: leave when none does, or when some lane of the wave waits at a leaf */ \
        "s_cmp_eq_u32 %[cnt], 0\n\t" \
        "s_cbranch_scc1 7f\n\t" \
        "s_mov_b64 exec, s[88:89]\n\t" \
        "v_cmp_lt_u32_e64 s[90:91], %[sentinel], %[node]\n\t" \
        "s_cmp_lg_u64 s[90:91], 0\n\t" \
        "s_cbranch_scc0 6b\n\t" \
        "7:\n\t" \
        "s_mov_b64 exec, s[96:97]\n\t"  /* push */ \
        "ds_write_b32 %[sp], %[top]\n\t" \
        "v_add_u32_e32 %[sp], %[stride], %[sp]\n\t" \
        "v_mov_b32_e32 %[top], v[" B "+6]\n\t" \
        "s_mov_b64 exec, s[98:99]\n\t"  /* pop */ \
        "v_subrev_u32_e32 %[sp], %[stride], %[sp]\n\t" \
        "ds_read_b32 %[top], %[sp]\n\t" \
        "9:\n\t" \
        "s_mov_b64 exec, s[88:89]\n\t" \
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
// The same loop for the tree in HBM (Node32: boxes as 16-bit grid indices, the grid folded into T.inv / T.A / T.B
// by trav_begin<true>): two 16-byte global loads per node, 12 conversions + 6 packed FMAs (both children of an axis
// at once; the pairs {inv, |inv|} and {A, E} are broadcast by op_sel) instead of the 12 mixed FMAs.
// The node's byte offset (index * 32) is formed in 32 bits: rt_scene_create admits at most 2^26 spheres in a tree.
// LOW: the block's sixteen vector registers are v80-v95 instead of v112-v127, for the build of the kernel that runs five
// waves per SIMD (96 registers).
template <int BLOCK, bool LOW>
__device__ __forceinline__ void inner_loop_hbm(Trav& T, const Node32* nodes32, int32_t thr) {
    static_assert(BLOCK == 512 || BLOCK == 256, "stack column stride");
    // scalar registers for the "s" operands, as in inner_loop_lds16
    thr = __builtin_amdgcn_readfirstlane(max(thr, 1));
    const uint64_t base_bits = reinterpret_cast<uint64_t>(nodes32);
    const uint64_t base = uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(int32_t(uint32_t(base_bits))))) |
                          (uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(int32_t(uint32_t(base_bits >> 32))))) << 32);
    int32_t lanes_on_inner;
    if constexpr (LOW) {
        asm volatile(RT_HBM_LOOP("80")
        : [node] "+v"(T.node), [sp] "+v"(T.sp), [top] "+v"(T.top), [cnt] "=&s"(lanes_on_inner)
        : [px] "v"(T.inv_abs[0]), [py] "v"(T.inv_abs[1]), [pz] "v"(T.inv_abs[2]), [qx] "v"(T.A_E[0]), [qy] "v"(T.A_E[1]),
          [qz] "v"(T.A_E[2]), [tmax] "v"(T.tmax),
          [base] "s"(base), [thr] "s"(thr), [sentinel] "s"(kLeafSentinel),
          [stride] "s"(uint32_t(BLOCK * sizeof(int32_t)))
        : "memory", "vcc", "scc", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "s88", "s89", "s90", "s91",
          "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99");
    } else {
        asm volatile(RT_HBM_LOOP("112")
        : [node] "+v"(T.node), [sp] "+v"(T.sp), [top] "+v"(T.top), [cnt] "=&s"(lanes_on_inner)
        : [px] "v"(T.inv_abs[0]), [py] "v"(T.inv_abs[1]), [pz] "v"(T.inv_abs[2]), [qx] "v"(T.A_E[0]), [qy] "v"(T.A_E[1]),
          [qz] "v"(T.A_E[2]), [tmax] "v"(T.tmax),
          [base] "s"(base), [thr] "s"(thr), [sentinel] "s"(kLeafSentinel),
          [stride] "s"(uint32_t(BLOCK * sizeof(int32_t)))
        : "memory", "vcc", "scc", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "s88", "s89", "s90", "s91",
          "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99");
    }

}

template <bool LDS16, int BLOCK>
__device__ __forceinline__ void leaf_step(const Tables& Tb, const Path& p, Trav& T, float t_scale) {
    const int32_t k = ~T.node;
    const bool better = sphere_candidate(Tb.sph[k], k, Tb.prim_id, p, true, T.best);
    T.tmax = better ? float(T.best.t) * (1.0f + 4.0e-7f) * t_scale : T.tmax;
    pop<BLOCK>(T);
}

// MODE selects a diagnostic build of the same kernel:
//   kModeStats   counts, per phase, wave-level turns and the lanes that took part, into the eight 64-bit words at
//                work_counter + 8 = {inner turns*64, inner lanes, leaf turns*64, leaf lanes, shade turns*64, shading
//                lanes, lanes that regenerated, descent-loop turns*64}; its inner-node loop is the compiled one
//   kModeDump    the production code (assembly loops included) that also stores every finished path's colour to
//                dump[(pixel * spp + sample - sample_begin) * 3]: what the parity tests compare path by path
//   kModeTiming  the production code with a cycle stamp around every phase; one lane per wave adds, to the same
//                eight words, {shade, descent-loop header, inner loop, leaf turn, whole kernel} cycles, waves, and the
//                two parts of the shade time spent {closing segments (hit_part), ending paths (flush, hand-out)}
constexpr int kModeProduction = 0, kModeStats = 1, kModeDump = 2, kModeTiming = 3;

__device__ __forceinline__ uint64_t cycle_stamp() { return __builtin_amdgcn_s_memtime(); }
// TREE = false is the kernel of scenes without a tree: the shade phase alone (every sphere is tested per segment).
// FIVE = true: built for 96 vector registers, five waves per SIMD (256-thread workgroups, five to a CU); chosen when five
// workgroups fit the CU's LDS (use_five_waves).  For a tree in LDS that build keeps only the nodes there (each
// workgroup has its own copy: 8 KB + 16 bytes a node, next to its stacks) and reads spheres and material tables
// through L1 like the HBM-tree kernel does; with the fifth wave the scalar and LDS instructions of one wave issue
// beside the vector instructions of another more often (bench scene: 10.32 -> 9.9 ms).
// LEAN: built for scenes known not to hold discs (kLeanNoDiscs) or anything but plain spheres (kLeanPlain: what most
// renders are, the bench scene included), see hit_part and trav_begin.
template <bool LDS16, int BLOCK, int MODE, bool TREE = true, bool FIVE = false, int LEAN = kLeanNone>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(FIVE ? 5 : 4, TREE ? (FIVE ? 5 : 4) : 8))) void render_kernel_ws(SceneDev S, CamDev C, RenderParams P,
                                                          const Node16* __restrict__ nodes16, uint32_t stack_depth,
                                                          float* __restrict__ sum_rgb, float* __restrict__ partials,
                                                          uint32_t* __restrict__ work_counter, float* __restrict__ dump) {
    constexpr bool STATS = MODE == kModeStats, TIMING = MODE == kModeTiming, DUMP = MODE == kModeDump;
    uint32_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t cyc[7] = {0, 0, 0, 0, 0, 0, 0};
    const uint64_t t_begin = TIMING ? cycle_stamp() : 0;
    extern __shared__ uint4 lds_raw[];
    // LDS layout (16-byte units): [q0: n_nodes][q1: n_nodes][spheres: 2*n_spheres][mat: n_prims]
    //                             [kind: n_prims/4][prim_id: n_prims/4][stack: stack_depth*BLOCK/4][hand-out state]
    // (q1 begins kQ1Offset bytes after q0 whatever n_nodes is)
    const int32_t n_prims = S.n_spheres + S.n_discs, quads = (n_prims + 3) / 4;
    uint4* lds_q0 = lds_raw;
    constexpr bool TABLES = LDS16 && !(TREE && FIVE);  // spheres and material tables staged in LDS too
    uint4* lds_q1 = lds_raw + (LDS16 ? (FIVE ? kQ1OffsetFive : kQ1Offset) / 16 : 0);  // second halves at a fixed distance from the first (see inner_loop_lds16)
    uint4* lds_sph = lds_q1 + (LDS16 ? S.n_nodes : 0);
    uint4* lds_mat = lds_sph + (TABLES ? 2 * S.n_spheres : 0);
    uint4* lds_kind = lds_mat + (TABLES ? n_prims : 0);
    uint4* lds_id = lds_kind + (TABLES ? quads : 0);
    uint4* lds_shell = lds_id + (TABLES ? quads : 0);
    int32_t* stk0 = reinterpret_cast<int32_t*>(lds_shell + ((TABLES && S.shell_count > 0) ? kShellTable / 2 : 0));
    int32_t* stk = stk0 + threadIdx.x;
    lds_u32* wave_state = (lds_u32*)(reinterpret_cast<uint32_t*>(stk0 + stack_depth * BLOCK) + 4 * (threadIdx.x >> 6));
    wave_state[0] = blockIdx.x % kCounters; wave_state[1] = 64; wave_state[2] = 0; wave_state[3] = 0;
    stk[0] = kLeafSentinel;
    CamDev* cam_slot = reinterpret_cast<CamDev*>(reinterpret_cast<uint32_t*>(stk0 + stack_depth * BLOCK) + 4 * (BLOCK / 64));
    if (threadIdx.x == 0) *cam_slot = C;
    lds_cam& Cl = *(lds_cam*)cam_slot;
    if (LDS16) {
        const uint4* src = reinterpret_cast<const uint4*>(nodes16);
        // inner links arrive as 16 * node index and become LDS addresses here
        const uint32_t q0_base = lds_address(lds_q0);
        for (int32_t i = threadIdx.x; i < S.n_nodes; i += BLOCK) {
            uint4 b = src[2 * i + 1];
            if (int32_t(b.z) >= 0) b.z += q0_base;
            if (int32_t(b.w) >= 0) b.w += q0_base;
            lds_q0[i] = src[2 * i];
            lds_q1[i] = b;
        }
    }
    if (TABLES) {
        const uint4* ssrc = reinterpret_cast<const uint4*>(S.sph);
        for (int32_t i = threadIdx.x; i < 2 * S.n_spheres; i += BLOCK) lds_sph[i] = ssrc[i];
        const uint4* msrc = reinterpret_cast<const uint4*>(S.mat);
        for (int32_t i = threadIdx.x; i < n_prims; i += BLOCK) lds_mat[i] = msrc[i];
        // the int tables are copied word by word: their length need not be a multiple of four
        int32_t* kdst = reinterpret_cast<int32_t*>(lds_kind);
        int32_t* idst = reinterpret_cast<int32_t*>(lds_id);
        for (int32_t i = threadIdx.x; i < n_prims; i += BLOCK) { kdst[i] = S.kind[i]; idst[i] = S.prim_id[i]; }
        double* rdst = reinterpret_cast<double*>(lds_shell);
        if (S.shell_count > 0 && threadIdx.x < kShellTable) rdst[threadIdx.x] = S.shell_r2[threadIdx.x];
    }
    __syncthreads();
    const Tables Tb = TABLES ? Tables{reinterpret_cast<const double4*>(lds_sph), reinterpret_cast<const float4*>(lds_mat),
                                     reinterpret_cast<const int32_t*>(lds_kind), reinterpret_cast<const int32_t*>(lds_id),
                                     reinterpret_cast<const double*>(lds_shell)}
                            : global_tables(S);

    Path p;
    p.ox = p.oy = p.oz = 0.0;
    p.dx = p.dy = p.dz = 0.0;
    p.tr = p.tg = p.tb = 0.0f;
    p.pixel = 0; p.sample = 0; p.bounce = 0;
    Trav T;
    T.node = kLeafSentinel;
    T.sp = 0;
    T.top = kLeafSentinel;
    T.tmax = 0.0f;
    T.best.t = 0.0; T.best.k = -1; T.best.id = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) { T.inv[a] = 0.0f; T.A[a] = 0.0f; T.E[a] = 0.0f; T.inv_abs[a] = float2v{0.0f, 0.0f}; T.A_E[a] = float2v{0.0f, 0.0f}; }
    bool have_pixel = false, alive = false, done = false;
    int32_t pending = kLeafSentinel;  // RT_RENDER_DEFER_LEAF: the leaf this lane has put off (never set otherwise)
    uint32_t next_sample = 0, item_end = 0, chunk = 0, pix_xy = 0;
    float ar = 0.0f, ag = 0.0f, ab = 0.0f;
    const int trav_threshold = int(P.trav_threshold), inner_threshold = int(P.inner_threshold);
    const int leave_threshold = int(P.leave_threshold);
    const uint32_t q0_addr = lds_address(lds_q0);
    const uint32_t slot1_addr = lds_address(stk) + uint32_t(BLOCK * sizeof(int32_t));

    for (;;) {
        // ---- shade phase: every lane that is not descending.  One Philox block per lane and turn:
        // a path that scatters on draws (pixel, sample, bounce, 0), a path that ends makes room for
        // the pixel's next sample, whose camera ray draws (pixel, sample, 0, 0).
        if (STATS) st[4]++;
        const uint64_t t_shade = TIMING ? cycle_stamp() : 0;
        if (T.node == kLeafSentinel && pending == kLeafSentinel && !done) {
            if (STATS) { st[5]++; if (!alive) st[6]++; }
            Surf sf;
            bool fresh = false;
            if (alive) {
                float Lr, Lg, Lb;
                if (!hit_part<LEAN>(S, Tb, P, p, T.best, sf, Lr, Lg, Lb)) {
                    ar += Lr; ag += Lg; ab += Lb;
                    alive = false;
                    if (DUMP) {
                        float* o = dump + 3 * (size_t(p.pixel) * (P.sample_end - P.sample_begin) + (p.sample - P.sample_begin));
                        o[0] = Lr; o[1] = Lg; o[2] = Lb;
                    }
                }
            }
            const uint64_t t_end = TIMING ? cycle_stamp() : 0;
            if (TIMING) cyc[5] += t_end - t_shade;
            if (!alive) {
                if (have_pixel && next_sample >= item_end) {
                    flush_item(P, sum_rgb, partials, p.pixel, chunk, ar, ag, ab);
                    have_pixel = false;
                }
                if (!have_pixel) {
                    bool need = true;
                    claim_pixels(need, done, p.pixel, pix_xy, chunk, wave_state, work_counter, P, C);
                    if (!done) {
                        have_pixel = true;
                        next_sample = P.sample_begin + chunk * P.chunk_spp;
                        item_end = min(next_sample + P.chunk_spp, P.sample_end);
                        ar = ag = ab = 0.0f;
                    }
                }
                if (have_pixel && next_sample < item_end) {
                    if (P.depth == 0) {
                        ar += 1.0f; ag += 1.0f; ab += 1.0f;
                        if (DUMP) {
                            float* o = dump + 3 * (size_t(p.pixel) * (P.sample_end - P.sample_begin) + (next_sample - P.sample_begin));
                            o[0] = 1.0f; o[1] = 1.0f; o[2] = 1.0f;
                        }
                    } else {
                        p.sample = next_sample;
                        fresh = true;
                        alive = true;
                    }
                    ++next_sample;
                }
            }
            if (TIMING) cyc[6] += cycle_stamp() - t_end;
            if (alive) {
                const uint4 r = philox4x32_10(p.pixel, p.sample, fresh ? 0u : p.bounce, 0u, P.key0, P.key1);
                // the new segment's origin and raw direction from either source, then ONE copy of Ray::new's
                // normalisation (the same arithmetic as start_path_from / scatter_part)
                float d[3];
                double ox, oy, oz;
                if (fresh) {
                    float o[3];
                    camera_ray_raw(Cl, pix_xy & 0xffffu, pix_xy >> 16, r, o, d);
                    ox = double(o[0]); oy = double(o[1]); oz = double(o[2]);
                    p.tr = p.tg = p.tb = 1.0f;
                    p.bounce = 1;
                } else {
                    alive = scatter_raw(P, p, sf, r, d[0], d[1], d[2]);  // false: the bounded fuzz loop gave up, the path contributes nothing
                    ox = sf.hx; oy = sf.hy; oz = sf.hz;
                    ++p.bounce;
                }
                if (alive) {
                    normalize3(d[0], d[1], d[2]);
                    p.ox = ox; p.oy = oy; p.oz = oz;
                    unit_f64(d[0], d[1], d[2], &p.dx, &p.dy, &p.dz);
                    trav_begin<!LDS16, TREE, LEAN>(S, Tb, p, T, slot1_addr, LDS16 ? int32_t(q0_addr) : 0);
                }
            }
        }
        if (TIMING) cyc[0] += cycle_stamp() - t_shade;
        if (__all(done)) break;

        // ---- descent phase
        // RT_RENDER_DEFER_LEAF (statistics build only; an experiment kept for its numbers): a lane that reaches a leaf
        // keeps it as `pending` and goes on with its stack; it parks only at a second leaf.  A leaf turn tests the
        // pending spheres, so more lanes take part in it and fewer leave the inner-node steps early; the price is that
        // the nearest-hit bound of a deferred sphere arrives late and culls less.  The statistics of such a launch
        // measure that trade (bench scene: lanes per leaf turn 37 -> 48 %, per inner step 70.6 -> 72.8 %, 2.6 % more
        // inner steps, 8 % more sphere tests, 4 % less time in this build).  The production loops do not do it: the
        // test "is the next node a leaf, and is the slot free" costs their inner step six instructions, which is
        // what the fuller turns give back; putting leaves off only when the inner loop pauses was slower (+5 %).
        const bool defer = STATS && (P.flags & RT_RENDER_DEFER_LEAF);
        for (; TREE;) {
            const uint64_t t_head = TIMING ? cycle_stamp() : 0;
            const bool descending = T.node != kLeafSentinel || (defer && pending != kLeafSentinel);
            const int n_desc = int(__popcll(__ballot(descending)));
            const bool someone_waits = __any(!descending && !done);
            if (n_desc == 0 || (n_desc < trav_threshold && someone_waits)) break;
            if (STATS) st[7]++;
            // inner steps until fewer than inner_threshold lanes are on inner nodes, or leave_threshold lanes have
            // left them (for a leaf or for good) since this run of steps began; a run never ends while lanes are on
            // inner nodes and none waits at a leaf
            const int run_threshold = max(inner_threshold, int(__popcll(__ballot(T.node >= 0))) - leave_threshold);
            const uint64_t t_inner = TIMING ? cycle_stamp() : 0;
            if (TIMING) cyc[1] += t_inner - t_head;
            if (LDS16 && !STATS) {
                inner_loop_lds16<BLOCK, FIVE>(T, run_threshold);
            } else if (!STATS) {
                inner_loop_hbm<BLOCK, FIVE>(T, S.nodes32, run_threshold);
            } else {
                for (;;) {
                    const int n_inner = int(__popcll(__ballot(T.node >= 0)));
                    const bool waits = T.node < 0 && (T.node != kLeafSentinel || (defer && pending != kLeafSentinel));
                    if (n_inner < run_threshold && (n_inner == 0 || __any(waits))) break;
                    if (STATS) { st[0]++; if (T.node >= 0) st[1]++; }
                    if (T.node >= 0) {
                        inner_step<LDS16, BLOCK>(S, lds_q0, lds_q1, T);
                        if (defer && T.node < 0 && T.node != kLeafSentinel && pending == kLeafSentinel) { pending = T.node; pop<BLOCK>(T); }
                    }
                }
            }
            const uint64_t t_leaf = TIMING ? cycle_stamp() : 0;
            if (TIMING) cyc[2] += t_leaf - t_inner;
            if (defer) {
                const bool test = pending != kLeafSentinel;
                if (__any(test)) { st[2]++; if (test) st[3]++; }
                if (test) {
                    const int32_t k = ~pending;
                    const bool better = sphere_candidate(Tb.sph[k], k, Tb.prim_id, p, true, T.best);
                    T.tmax = better ? float(T.best.t) * (1.0f + 4.0e-7f) * S.t_scale : T.tmax;
                    pending = kLeafSentinel;
                }
                if (T.node < 0 && T.node != kLeafSentinel) { pending = T.node; pop<BLOCK>(T); }
            } else {
                if (STATS && __any(T.node < 0 && T.node != kLeafSentinel)) { st[2]++; if (T.node < 0 && T.node != kLeafSentinel) st[3]++; }
                if (T.node < 0 && T.node != kLeafSentinel) leaf_step<LDS16, BLOCK>(Tb, p, T, S.t_scale);
            }
            if (TIMING) cyc[3] += cycle_stamp() - t_leaf;
        }
    }
    if (STATS) {
        unsigned long long* out = reinterpret_cast<unsigned long long*>(work_counter + 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(out + i, (unsigned long long)st[i]);
    }
    if (TIMING && (threadIdx.x & 63) == 0) {
        cyc[4] = cycle_stamp() - t_begin;
        unsigned long long* out = reinterpret_cast<unsigned long long*>(work_counter + 8);
#pragma unroll
        for (int i = 0; i < 5; ++i) atomicAdd(out + i, (unsigned long long)cyc[i]);
        atomicAdd(out + 5, 1ull);
        atomicAdd(out + 6, (unsigned long long)cyc[5]);
        atomicAdd(out + 7, (unsigned long long)cyc[6]);
    }
}

// One thread per listed path: the colour accel_raytrace returns for it.
__global__ __launch_bounds__(kBlock) void trace_paths_kernel(SceneDev S, CamDev C, RenderParams P,
                                                             const uint32_t* __restrict__ pixel,
                                                             const uint32_t* __restrict__ sample, uint32_t n,
                                                             float* __restrict__ rgb) {
    extern __shared__ int32_t stack_lds[];
    int32_t* stk = stack_lds + threadIdx.x;
    const uint32_t g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= n) return;
    float Lr = 1.0f, Lg = 1.0f, Lb = 1.0f;
    if (P.depth > 0) {
        Path p;
        start_path(C, P, pixel[g], sample[g], p);
        while (advance(S, P, p, stk, Lr, Lg, Lb)) {}
    }
    rgb[3 * size_t(g)] = Lr; rgb[3 * size_t(g) + 1] = Lg; rgb[3 * size_t(g) + 2] = Lb;
}

// One thread per listed path: the primitive ID and parameter of every segment (id -1 = the miss that ends it).
__global__ __launch_bounds__(kBlock) void path_segments_kernel(SceneDev S, CamDev C, RenderParams P,
                                                               const uint32_t* __restrict__ pixel,
                                                               const uint32_t* __restrict__ sample, uint32_t n,
                                                               uint32_t max_segments, int32_t* __restrict__ seg_id,
                                                               double* __restrict__ seg_t, uint32_t* __restrict__ n_seg) {
    extern __shared__ int32_t stack_lds[];
    int32_t* stk = stack_lds + threadIdx.x;
    const uint32_t g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= n) return;
    uint32_t count = 0;
    if (P.depth > 0) {
        Path p;
        start_path(C, P, pixel[g], sample[g], p);
        float Lr, Lg, Lb;
        bool go = true;
        while (go) {
            int32_t id;
            double t;
            go = advance(S, P, p, stk, Lr, Lg, Lb, &id, &t);
            if (count < max_segments) { seg_id[size_t(g) * max_segments + count] = id; seg_t[size_t(g) * max_segments + count] = t; }
            ++count;
        }
    }
    n_seg[g] = count;
}

__global__ __launch_bounds__(kBlock) void generate_primary_kernel(CamDev C, RenderParams P, uint32_t pixel_begin,
                                                                  uint32_t ns, uint32_t n, float* __restrict__ orig,
                                                                  float* __restrict__ dir) {
    const uint32_t g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= n) return;
    float o[3], d[3];
    camera_ray(C, P, pixel_begin + g / ns, P.sample_begin + g % ns, o, d);
#pragma unroll
    for (int a = 0; a < 3; ++a) { orig[3 * size_t(g) + a] = o[a]; dir[3 * size_t(g) + a] = d[a]; }
}

__global__ __launch_bounds__(kBlock) void trace_primary_kernel(SceneDev S, const float* __restrict__ orig,
                                                               const float* __restrict__ dir, uint32_t n,
                                                               int32_t* __restrict__ id, double* __restrict__ t,
                                                               double* __restrict__ normal) {
    extern __shared__ int32_t stack_lds[];
    int32_t* stk = stack_lds + threadIdx.x;
    const uint32_t g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= n) return;
    const double ox = double(orig[3 * size_t(g)]), oy = double(orig[3 * size_t(g) + 1]), oz = double(orig[3 * size_t(g) + 2]);
    const float dx = dir[3 * size_t(g)], dy = dir[3 * size_t(g) + 1], dz = dir[3 * size_t(g) + 2];
    const Best hit = closest_hit<true>(S, ox, oy, oz, double(dx), double(dy), double(dz), stk);
    if (hit.k < 0) { id[g] = -1; return; }
    id[g] = hit.id;
    t[g] = hit.t;
    double nx, ny, nz;
    if (hit.k < S.n_spheres) {
#pragma clang fp contract(off)
        const double4 s = S.sph[hit.k];
        nx = ((ox + hit.t * double(dx)) - s.x) / s.w;  // src/geometry.rs:123-125 at src/ray.rs:12-14
        ny = ((oy + hit.t * double(dy)) - s.y) / s.w;
        nz = ((oz + hit.t * double(dz)) - s.z) / s.w;
    } else {
        const DiscDev& dd = S.discs[hit.k - S.n_spheres];
        nx = dd.nx; ny = dd.ny; nz = dd.nz;
    }
    normal[3 * size_t(g)] = nx; normal[3 * size_t(g) + 1] = ny; normal[3 * size_t(g) + 2] = nz;
}

// image = (old image when accumulating) + partial sums in chunk order
__global__ __launch_bounds__(kBlock) void combine_kernel(const float* __restrict__ partials, uint32_t n_chunks, size_t n_channels,
                                                         uint32_t accumulate, float* __restrict__ sum_rgb) {
    const size_t g = size_t(blockIdx.x) * kBlock + threadIdx.x;
    if (g >= n_channels) return;
    float acc = accumulate ? sum_rgb[g] : 0.0f;
    for (uint32_t c = 0; c < n_chunks; ++c) acc += partials[size_t(c) * n_channels + g];
    sum_rgb[g] = acc;
}

// (1/spp)*sum then color_to_ppm (src/lib.rs:148-149, :186-192), in f64 like the reference.
__global__ __launch_bounds__(kBlock) void finalize_kernel(const float* __restrict__ sum_rgb, size_t n_channels,
                                                          double inv_spp, uint8_t* __restrict__ rgb8) {
    const size_t g = size_t(blockIdx.x) * kBlock + threadIdx.x;
    if (g >= n_channels) return;
    const double x = 255.0 * __dsqrt_rn(inv_spp * double(sum_rgb[g]));
    uint8_t q = 0;  // Rust `as u8`: NaN -> 0, saturating, truncating
    if (x > 0.0) q = (x >= 255.0) ? uint8_t(255) : uint8_t(int(x));
    rgb8[g] = q;
}

// ---------------------------------------------------------------- wavefront renderer (RT_RENDER_KERNEL_WAVEFRONT)
// The same path tracer as a pipeline of kernels that hand rays to one another through queues in HBM, with the rays
// compacted by the material they hit: generate -> for every bounce { trace: nearest hit of every queued ray, the ray's
// slot appended to the queue of its hit's material class; shade: one wave works on rays of ONE class (miss or emitter /
// diffuse or texture map / metal / dielectric), ends the path or scatters it and appends the survivor to the next
// bounce's ray queue } -> accumulate.  A path lives in one slot of a pool for its whole life (slot = pixel in batch *
// samples + sample), a queue holds slot numbers, and a frame larger than the pool is rendered in batches of whole
// pixels.  Every path is traced with the arithmetic of the persistent kernel (same Philox counters, same functions),
// and a pixel's samples are added in the same order and chunk grouping, so the image equals render_kernel_ws's bit
// for bit.  Measured against it on the bench scene the pipeline is four times slower (each segment writes and reads
// its ray, hit and throughput through HBM): it is kept as the measured alternative, not as the default.
struct WfState {
    double* ray;       // 6 per slot: origin, unit direction
    float* thr;        // 3 per slot: throughput
    double* hit_t;     // per slot: parameter of the nearest hit of the segment being shaded
    int32_t* hit_k;    // per slot: its table index, -1 = miss
    float* colour;     // 3 per slot: what the finished path contributes
    uint32_t* rayq;    // kWfSub sub-queues of sub_cap slots: the rays whose next segment is to be traced
    uint32_t* matq;    // kWfClasses x kWfSub sub-queues: the traced rays by class of the material they hit
    uint32_t* counts;  // queue lengths, see wf_count
    uint32_t sub_cap;
};
constexpr int kWfClasses = 4;  // 0 miss / emitter (the path ends), 1 diffuse / texture map, 2 metal, 3 dielectric
// A queue is kWfSub sub-queues with a length word each, every word on a cache line of its own: a workgroup appends
// to sub-queue (its number mod kWfSub).  One length word per queue was the bottleneck of the whole pipeline (a word
// serves about 90 atomics per microsecond; a 1080p / 100 spp frame makes 13 M wave-level appends per queue).
constexpr int kWfSub = 16;
constexpr uint32_t kWfMaxSlots = 1u << 25;
constexpr uint32_t kWfQueuedBounces = 10;  // bounces that go through the queues; wf_finish_kernel takes what is left
// A sub-queue receives what the workgroups of one residue class produce, at most 1 / kWfSub of the entries plus one
// 64-entry block per wave of the grid: room for that share and 65536 entries more.
constexpr uint32_t kWfSubMargin = 65536;
constexpr uint32_t kWfCountWords = 32;  // words between two length words (128 bytes)
constexpr uint32_t kWfCountsPerBounce = (1 + kWfClasses) * kWfSub * kWfCountWords;
// length word of sub-queue `sub` of queue q (0 = rays to trace, 1 + c = class c to shade) of bounce b; the very last
// word block of the array (one past the last bounce that is queued) is an overflow flag
__device__ __forceinline__ uint32_t* wf_count(const WfState& W, uint32_t bounce, uint32_t q, uint32_t sub) {
    return W.counts + size_t(bounce) * kWfCountsPerBounce + (q * kWfSub + sub) * kWfCountWords;
}

// position of each asking lane in a sub-queue: one atomic per wave
__device__ __forceinline__ uint32_t wf_append(uint32_t* count, bool ask) {
    const unsigned long long m = __ballot(ask);
    if (m == 0) return 0;
    const int first = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (int(__lane_id()) == first) base = atomicAdd(count, uint32_t(__popcll(m)));
    base = uint32_t(__builtin_amdgcn_readlane(int32_t(base), first));
    return base + __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
}

// The sub-queues of a bounce's input, each rounded up to whole 64-entry blocks, laid end to end: ends[i] = end of
// segment i in that index space (n_seg <= kWfClasses * kWfSub).  Called by the whole workgroup.
__device__ __forceinline__ void wf_segments(const WfState& W, uint32_t bounce, uint32_t q_first, uint32_t n_seg, uint32_t* len, uint32_t* ends) {
    if (threadIdx.x < n_seg) len[threadIdx.x] = min(*wf_count(W, bounce, q_first + threadIdx.x / kWfSub, threadIdx.x % kWfSub), W.sub_cap);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t i = 0; i < n_seg; ++i) { total += (len[i] + 63u) & ~63u; ends[i] = total; }
    }
    __syncthreads();
}
// segment of the 64-entry block that begins at g0 (g0 < ends[n_seg - 1])
__device__ __forceinline__ uint32_t wf_segment_of(const uint32_t* ends, uint32_t n_seg, uint32_t g0) {
    uint32_t lo = 0, hi = n_seg - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        if (g0 < ends[mid]) hi = mid; else lo = mid + 1;
    }
    return lo;
}

__global__ __launch_bounds__(kBlock) void wf_generate_kernel(CamDev C, RenderParams P, WfState W, uint32_t pixel0, uint32_t n_paths) {
    const uint32_t spp = P.sample_end - P.sample_begin;
    const uint32_t per = (((n_paths + kWfSub - 1) / kWfSub) + 63u) & ~63u;  // paths per sub-queue
    for (uint32_t g = blockIdx.x * kBlock + threadIdx.x; g < n_paths; g += gridDim.x * kBlock) {
        if (g < uint32_t(kWfSub)) *wf_count(W, 0, 0, g) = (P.depth == 0 || g * per >= n_paths) ? 0u : min(per, n_paths - g * per);
        if (P.depth == 0) { W.colour[3 * size_t(g)] = 1.0f; W.colour[3 * size_t(g) + 1] = 1.0f; W.colour[3 * size_t(g) + 2] = 1.0f; continue; }
        Path p;
        start_path(C, P, pixel0 + g / spp, P.sample_begin + g % spp, p);
        double* r = W.ray + 6 * size_t(g);
        r[0] = p.ox; r[1] = p.oy; r[2] = p.oz; r[3] = p.dx; r[4] = p.dy; r[5] = p.dz;
        W.thr[3 * size_t(g)] = 1.0f; W.thr[3 * size_t(g) + 1] = 1.0f; W.thr[3 * size_t(g) + 2] = 1.0f;
        W.rayq[size_t(g / per) * W.sub_cap + g % per] = g;
    }
}

// Nearest hit of every queued ray: the descent of render_kernel_ws (assembly inner loops, f64 sphere test) on 64 rays
// per wave at a time, then the compaction: the ray's slot goes to the queue of its hit's material class.
template <bool LDS16>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(5, 5))) void wf_trace_kernel(SceneDev S, RenderParams P, WfState W,
                                                                                                 const Node16* __restrict__ nodes16, uint32_t bounce) {
    extern __shared__ uint4 lds_raw[];
    uint4* lds_q0 = lds_raw;
    uint4* lds_q1 = lds_raw + (LDS16 ? kQ1OffsetFive / 16 : 0);
    int32_t* stk0 = reinterpret_cast<int32_t*>(lds_q1 + (LDS16 ? S.n_nodes : 0));
    int32_t* stk = stk0 + threadIdx.x;
    stk[0] = kLeafSentinel;
    __shared__ uint32_t seg_len[kWfSub], seg_end[kWfSub];
    if (LDS16) {
        const uint4* src = reinterpret_cast<const uint4*>(nodes16);
        const uint32_t q0_base = lds_address(lds_q0);
        for (int32_t i = threadIdx.x; i < S.n_nodes; i += kBlock) {
            uint4 b = src[2 * i + 1];
            if (int32_t(b.z) >= 0) b.z += q0_base;
            if (int32_t(b.w) >= 0) b.w += q0_base;
            lds_q0[i] = src[2 * i];
            lds_q1[i] = b;
        }
    }
    wf_segments(W, bounce, 0, kWfSub, seg_len, seg_end);
    const Tables Tb = global_tables(S);
    const uint32_t q0_addr = lds_address(lds_q0);
    const uint32_t slot1_addr = lds_address(stk) + uint32_t(kBlock * sizeof(int32_t));
    const uint32_t total = seg_end[kWfSub - 1];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waves = gridDim.x * (kBlock / 64);
    const uint32_t my_sub = blockIdx.x % kWfSub;
    for (uint32_t g0 = (blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 64u; g0 < total; g0 += waves * 64u) {
        const uint32_t seg = wf_segment_of(seg_end, kWfSub, g0);
        const uint32_t j = g0 - (seg ? seg_end[seg - 1] : 0u) + lane;
        const bool active = j < seg_len[seg];
        const uint32_t slot = active ? W.rayq[size_t(seg) * W.sub_cap + j] : 0u;
        const double* r = W.ray + 6 * size_t(slot);
        Path p;
        p.ox = r[0]; p.oy = r[1]; p.oz = r[2]; p.dx = r[3]; p.dy = r[4]; p.dz = r[5];
        p.tr = p.tg = p.tb = 0.0f; p.pixel = p.sample = p.bounce = 0;
        Trav T;
        trav_begin<!LDS16, true>(S, Tb, p, T, slot1_addr, LDS16 ? int32_t(q0_addr) : 0);
        if (!active) T.node = kLeafSentinel;
        while (__any(T.node != kLeafSentinel)) {
            if constexpr (LDS16) inner_loop_lds16<kBlock, true>(T, int32_t(P.inner_threshold));
            else inner_loop_hbm<kBlock, true>(T, S.nodes32, int32_t(P.inner_threshold));
            if (T.node < 0 && T.node != kLeafSentinel) leaf_step<LDS16, kBlock>(Tb, p, T, S.t_scale);
        }
        int32_t cls = 0;
        if (active) {
            W.hit_t[slot] = T.best.t;
            W.hit_k[slot] = T.best.k;
            if (T.best.k >= 0) {
                const int32_t kind = Tb.kind[T.best.k] & 0xff;
                cls = kind == RT_MAT_METAL ? 2 : kind == RT_MAT_DIELECTRIC ? 3 : kind == RT_MAT_EMITTER ? 0 : 1;
            }
        }
#pragma unroll
        for (int c = 0; c < kWfClasses; ++c) {
            const bool mine = active && cls == c;
            const uint32_t pos = wf_append(wf_count(W, bounce, 1 + c, my_sub), mine);
            if (mine) {
                if (pos < W.sub_cap) W.matq[(size_t(c) * kWfSub + my_sub) * W.sub_cap + pos] = slot;
                else *wf_count(W, min(P.depth, kWfQueuedBounces) + 1, 0, 0) = 1u;  // cannot happen by the sizing of the sub-queues; never write past one
            }
        }
    }
}

// One bounce of shading.  The class queues are walked one after the other, each sub-queue rounded up to whole waves,
// so the lanes of a wave work on hits of one material class.
__global__ __launch_bounds__(kBlock) void wf_shade_kernel(SceneDev S, RenderParams P, WfState W, uint32_t pixel0, uint32_t bounce) {
    __shared__ uint32_t seg_len[kWfClasses * kWfSub], seg_end[kWfClasses * kWfSub];
    constexpr uint32_t n_seg = kWfClasses * kWfSub;
    static_assert(n_seg <= uint32_t(kBlock), "one thread per segment");
    wf_segments(W, bounce, 1, n_seg, seg_len, seg_end);
    const Tables Tb = global_tables(S);
    const uint32_t spp = P.sample_end - P.sample_begin;
    const uint32_t total = seg_end[n_seg - 1];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waves = gridDim.x * (kBlock / 64);
    const uint32_t my_sub = blockIdx.x % kWfSub;
    for (uint32_t g0 = (blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 64u; g0 < total; g0 += waves * 64u) {
        const uint32_t seg = wf_segment_of(seg_end, n_seg, g0);
        const uint32_t j = g0 - (seg ? seg_end[seg - 1] : 0u) + lane;
        bool alive = false;
        uint32_t slot = 0;
        if (j < seg_len[seg]) {
            slot = W.matq[size_t(seg) * W.sub_cap + j];
            double* r = W.ray + 6 * size_t(slot);
            float* th = W.thr + 3 * size_t(slot);
            Path p;
            p.ox = r[0]; p.oy = r[1]; p.oz = r[2]; p.dx = r[3]; p.dy = r[4]; p.dz = r[5];
            p.tr = th[0]; p.tg = th[1]; p.tb = th[2];
            p.pixel = pixel0 + slot / spp; p.sample = P.sample_begin + slot % spp; p.bounce = bounce + 1;
            Best hit;
            hit.t = W.hit_t[slot]; hit.k = W.hit_k[slot]; hit.id = 0;
            Surf sf;
            float Lr, Lg, Lb;
            alive = hit_part(S, Tb, P, p, hit, sf, Lr, Lg, Lb);
            float d[3] = {0.0f, 0.0f, 0.0f};
            if (alive) {
                const uint4 rnd = philox4x32_10(p.pixel, p.sample, p.bounce, 0u, P.key0, P.key1);
                alive = scatter_raw(P, p, sf, rnd, d[0], d[1], d[2]);
                if (!alive) { Lr = Lg = Lb = 0.0f; }  // the bounded fuzz loop gave up: the path contributes nothing
            }
            if (alive) {
                normalize3(d[0], d[1], d[2]);
                double dx, dy, dz;
                unit_f64(d[0], d[1], d[2], &dx, &dy, &dz);
                r[0] = sf.hx; r[1] = sf.hy; r[2] = sf.hz; r[3] = dx; r[4] = dy; r[5] = dz;
                th[0] = p.tr; th[1] = p.tg; th[2] = p.tb;
            } else {
                float* o = W.colour + 3 * size_t(slot);
                o[0] = Lr; o[1] = Lg; o[2] = Lb;
            }
        }
        const uint32_t pos = wf_append(wf_count(W, bounce + 1, 0, my_sub), alive);
        if (alive) {
            if (pos < W.sub_cap) W.rayq[size_t(my_sub) * W.sub_cap + pos] = slot;
            else *wf_count(W, min(P.depth, kWfQueuedBounces) + 1, 0, 0) = 1u;
        }
    }
}

// The last stage of the pipeline: what is still alive after `bounce` bounces (a fraction of a percent of the paths
// from the tenth bounce on) is followed to its end by one lane per path, the way trace_paths_kernel follows a path —
// forty more pairs of nearly empty launches per batch cost more than this one.
__global__ __launch_bounds__(kBlock) void wf_finish_kernel(SceneDev S, RenderParams P, WfState W, uint32_t pixel0, uint32_t bounce) {
    extern __shared__ int32_t stack_lds[];
    int32_t* stk = stack_lds + threadIdx.x;
    __shared__ uint32_t seg_len[kWfSub], seg_end[kWfSub];
    wf_segments(W, bounce, 0, kWfSub, seg_len, seg_end);
    const uint32_t spp = P.sample_end - P.sample_begin;
    const uint32_t total = seg_end[kWfSub - 1];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waves = gridDim.x * (kBlock / 64);
    for (uint32_t g0 = (blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 64u; g0 < total; g0 += waves * 64u) {
        const uint32_t seg = wf_segment_of(seg_end, kWfSub, g0);
        const uint32_t j = g0 - (seg ? seg_end[seg - 1] : 0u) + lane;
        if (j >= seg_len[seg]) continue;
        const uint32_t slot = W.rayq[size_t(seg) * W.sub_cap + j];
        const double* r = W.ray + 6 * size_t(slot);
        const float* th = W.thr + 3 * size_t(slot);
        Path p;
        p.ox = r[0]; p.oy = r[1]; p.oz = r[2]; p.dx = r[3]; p.dy = r[4]; p.dz = r[5];
        p.tr = th[0]; p.tg = th[1]; p.tb = th[2];
        p.pixel = pixel0 + slot / spp; p.sample = P.sample_begin + slot % spp; p.bounce = bounce + 1;
        float Lr, Lg, Lb;
        while (advance(S, P, p, stk, Lr, Lg, Lb)) {}
        float* o = W.colour + 3 * size_t(slot);
        o[0] = Lr; o[1] = Lg; o[2] = Lb;
    }
}

// A pixel's colours added in sample order, chunk by chunk, into the places render_kernel_ws's flush_item writes.
__global__ __launch_bounds__(kBlock) void wf_accumulate_kernel(RenderParams P, WfState W, uint32_t pixel0, uint32_t n_batch_pixels,
                                                               float* __restrict__ sum_rgb, float* __restrict__ partials) {
    const uint32_t spp = P.sample_end - P.sample_begin;
    const uint32_t g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= n_batch_pixels * P.n_chunks) return;
    const uint32_t chunk = g / n_batch_pixels, pl = g - chunk * n_batch_pixels;
    const uint32_t s0 = chunk * P.chunk_spp, s1 = min(s0 + P.chunk_spp, spp);
    float ar = 0.0f, ag = 0.0f, ab = 0.0f;
    for (uint32_t s = s0; s < s1; ++s) {
        const float* c = W.colour + 3 * (size_t(pl) * spp + s);
        ar += c[0]; ag += c[1]; ab += c[2];
    }
    flush_item(P, sum_rgb, partials, pixel0 + pl, chunk, ar, ag, ab);
}

// Scratch of rt_render_device_noalloc: the first 2048 bytes are the counter block (the words of the diagnostic
// builds and the work counters), the rest holds the partial sums of chunked renders.  One copy per GPU the library runs on, for the life of
// the process.  In host code this name is only a handle for hipGetSymbolAddress, never a device address.
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];

}  // namespace rt

// =====================================================================
// host side of the device ABI
// =====================================================================
using namespace rt;

struct rt_scene {
    int device = 0;
    rt_scene_info info{};
    SceneDev dev{};
    std::vector<void*> allocations;
    uint32_t* d_work_counter = nullptr;
    float* d_partials = nullptr;        // partial-sum slots of chunked renders, grown on demand
    bool plain = false;                 // spheres of the three common materials only, no nested shells: the production kernel's lean build
    bool no_discs = false;              // no disc in the scene (the lean build of the kernel for scenes without a tree)
    unsigned char* d_wf = nullptr;      // path pool and queues of the wavefront renderer (RT_RENDER_KERNEL_WAVEFRONT), made on first use
    size_t wf_slots = 0;
    uint32_t* d_wf_counts = nullptr;    // its queue lengths, eight words per bounce
    size_t wf_counts_bytes = 0;
    float* d_frame_sum = nullptr;       // frame buffers of rt_render (host-buffer entry point), kept between calls
    uint8_t* d_frame_rgb = nullptr;
    size_t frame_pixels = 0;
    size_t partials_bytes = 0;
    const Node16* d_nodes16 = nullptr;  // f16 copy of the nodes for the LDS-resident kernel (null when it has none)
    bool lds_scene = false;             // the render kernel stages nodes and primitive tables in LDS
    size_t lds16_scene_bytes = 0;
    size_t lds16_five_bytes = 0;  // LDS the five-wave build stages for this scene (nodes only), 0 = the tree has too many nodes for it
    uint32_t stack_depth = 0;
    int n_cu = 0;
    double bound_abs = 0.0;  // largest |coordinate| of any box corner
    double bound_rel = 0.0;  // largest distance, per axis, of any box corner from the reference points hbase / qbase
    hipEvent_t ev_done = nullptr;  // recorded after every render of this handle: the next render waits for it, on any stream
    // workgroups per CU of each render kernel variant for this scene's LDS footprint, 0 = not asked yet
    // ([0] lock-step kernel, [1 + mode] wave-scheduled kernel with the tree in HBM, [5 + mode] with the scene in LDS)
    int per_cu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int five_waves[2] = {-1, -1};  // [tree in HBM, tree in LDS]: 1 = the kernel built for five waves per SIMD is used, 0 = the four-wave one, -1 = not decided yet
};

namespace {

// IEEE half bits of the smallest half >= x; overflow gives +inf.
uint16_t half_bits(_Float16 h) { uint16_t b; std::memcpy(&b, &h, 2); return b; }
_Float16 half_from_bits(uint16_t b) { _Float16 h; std::memcpy(&h, &b, 2); return h; }
uint16_t half_round_up(double x) {
    _Float16 h = _Float16(x);
    if (double(h) < x) {
        uint16_t b = half_bits(h);
        if ((b & 0x7fffu) == 0) b = 0x0001u;
        else if (b & 0x8000u) b = uint16_t(b - 1);    // negative (incl. -inf -> -65504)
        else b = uint16_t(b + 1);
        h = half_from_bits(b);
    }
    return half_bits(h);
}

// ---- scene tables made on the device (the tree of RT_BUILDER_LBVH_DEVICE never visits the host)
// Node -> Node32 with the arithmetic of the host encoder in rt_scene_create_ex: indices chosen in f64 against the
// f32 base and step, bounds rounded outward, then centre and half extent in grid units.
__device__ __forceinline__ uint32_t grid_down(float v, float base, float step) {
    if (!(v < 3.0e38f)) return 65535u;
    double q = floor((double(v) - double(base)) / double(step));
    q = fmin(fmax(q, 0.0), 65535.0);
    while (q > 0.0 && double(base) + q * double(step) > double(v)) q -= 1.0;
    return uint32_t(q);
}
__device__ __forceinline__ uint32_t grid_up(float v, float base, float step) {
    if (!(v < 3.0e38f)) return 65535u;
    double q = ceil((double(v) - double(base)) / double(step));
    q = fmin(fmax(q, 0.0), 65535.0);
    while (q < 65535.0 && double(base) + q * double(step) < double(v)) q += 1.0;
    return uint32_t(q);
}
__global__ __launch_bounds__(kBlock) void encode32_kernel(const Node* __restrict__ nodes, uint32_t n, float3 base, float3 step,
                                                          Node32* __restrict__ out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const Node nd = nodes[i];
    const float b[3] = {base.x, base.y, base.z}, st[3] = {step.x, step.y, step.z};
    Node32 r;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const uint32_t l0 = grid_down(nd.lo0[a], b[a], st[a]), u0 = grid_up(nd.hi0[a], b[a], st[a]);
        const uint32_t l1 = grid_down(nd.lo1[a], b[a], st[a]), u1 = grid_up(nd.hi1[a], b[a], st[a]);
        const uint32_t c0 = (l0 + u0) / 2, c1 = (l1 + u1) / 2;
        r.q[4 * a] = uint16_t(c0); r.q[4 * a + 1] = uint16_t(c1);
        r.q[4 * a + 2] = uint16_t(max(u0 - c0, c0 - l0)); r.q[4 * a + 3] = uint16_t(max(u1 - c1, c1 - l1));
    }
    r.child0 = nd.child0; r.child1 = nd.child1;
    out[i] = r;
}
// out[k] = in[order[k]] for the spheres of the tree (leaf order), the rest of the table as it is
template <class T>
__global__ __launch_bounds__(kBlock) void gather_kernel(const T* __restrict__ in, const uint32_t* __restrict__ order, uint32_t n_tree,
                                                        uint32_t n_total, T* __restrict__ out) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= n_total) return;
    out[k] = in[k < n_tree ? order[k] : k];
}

template <class T>
int upload(rt_scene* sc, const std::vector<T>& host, const T** dev_ptr) {
    *dev_ptr = nullptr;
    if (host.empty()) return RT_OK;
    void* p = nullptr;
    RT_HIP(hipMalloc(&p, host.size() * sizeof(T)));
    sc->allocations.push_back(p);
    sc->info.device_bytes += host.size() * sizeof(T);
    RT_HIP(hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *dev_ptr = static_cast<const T*>(p);
    return RT_OK;
}

void normalize3d(double v[3]) {
    double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    for (int i = 0; i < 3; ++i) v[i] = v[i] / n;
}

CamDev make_cam(const rt_camera& c) {
    CamDev d;
    for (int a = 0; a < 3; ++a) {
        d.lookfrom[a] = float(c.lookfrom[a]);
        d.lookat[a] = float(c.lookat[a]);
        d.h[a] = float(c.horiz_arm[a]);
        d.v[a] = float(c.vert_arm[a]);
    }
    d.aperture = float(c.aperture);
    d.span = float(c.inv_focal_length * c.focal_distance);  // src/camera.rs:67-68
    d.inv_w = float(1.0 / double(c.horiz_res));
    d.inv_h = float(1.0 / double(c.vert_res));
    d.W = c.horiz_res;
    d.H = c.vert_res;
    return d;
}

// eps for a batch of rays whose origins stay within `origin_abs` of zero
float descent_eps(const rt_scene* sc, double origin_abs) {
    double extent = 2.0 * std::max(sc->bound_abs, origin_abs);
    return float(extent * (1.0 / 2097152.0));  // 2^-21 of the largest distance between a box plane and an origin
}

// LDS of the 256-thread kernels: the traversal stack plus four words of pixel hand-out state per wave
size_t stack_bytes(const rt_scene* sc) { return size_t(std::max<uint32_t>(sc->stack_depth, 1u)) * kBlock * sizeof(int32_t) + 64 + kCamLdsBytes; }  // 16 bytes of hand-out state per wave, the camera

// LDS of one workgroup of the LDS-resident kernel: scene tables, the stack columns of kBlockLds lanes, hand-out state
size_t lds16_launch_bytes(size_t scene_bytes, uint32_t stack_depth, int block = kBlockLds) {
    return scene_bytes + size_t(std::max<uint32_t>(stack_depth, 1u)) * size_t(block) * sizeof(int32_t) + 16 * size_t(block / 64) + kCamLdsBytes;
}
constexpr size_t kLdsPerCu = 160 * 1024;

// Restores the calling thread's current HIP device when an entry point returns.
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

typedef void (*ws_kernel_t)(SceneDev, CamDev, RenderParams, const Node16*, uint32_t, float*, float*, uint32_t*, float*);
ws_kernel_t ws_kernel(bool lds16, int mode, bool tree = true, bool five = false, bool plain = false, bool no_discs = false) {
    // the production builds for scenes known not to hold discs, or anything but plain spheres (the statistics and dump
    // builds keep every branch)
    if (plain && tree && mode == kModeProduction) {
        if (lds16) return five ? render_kernel_ws<true, kBlock, kModeProduction, true, true, kLeanPlain> : render_kernel_ws<true, kBlockLds, kModeProduction, true, false, kLeanPlain>;
        return five ? render_kernel_ws<false, kBlock, kModeProduction, true, true, kLeanPlain> : render_kernel_ws<false, kBlock, kModeProduction, true, false, kLeanPlain>;
    }
    if (no_discs && !plain && tree && mode == kModeProduction) {  // e.g. spheres with a texture map or an emitter among them
        if (lds16) return five ? render_kernel_ws<true, kBlock, kModeProduction, true, true, kLeanNoDiscs> : render_kernel_ws<true, kBlockLds, kModeProduction, true, false, kLeanNoDiscs>;
        return five ? render_kernel_ws<false, kBlock, kModeProduction, true, true, kLeanNoDiscs> : render_kernel_ws<false, kBlock, kModeProduction, true, false, kLeanNoDiscs>;
    }
    if (plain && tree && mode == kModeTiming) {  // the cycle stamps go around the production build's own phases
        if (lds16) return five ? render_kernel_ws<true, kBlock, kModeTiming, true, true, kLeanPlain> : render_kernel_ws<true, kBlockLds, kModeTiming, true, false, kLeanPlain>;
        return five ? render_kernel_ws<false, kBlock, kModeTiming, true, true, kLeanPlain> : render_kernel_ws<false, kBlock, kModeTiming, true, false, kLeanPlain>;
    }
    if (lds16 && tree && five) {
        switch (mode) {
        case kModeStats: return render_kernel_ws<true, kBlock, kModeStats, true, true>;
        case kModeDump: return render_kernel_ws<true, kBlock, kModeDump, true, true>;
        case kModeTiming: return render_kernel_ws<true, kBlock, kModeTiming, true, true>;
        default: return render_kernel_ws<true, kBlock, kModeProduction, true, true>;
        }
    }
    if (!lds16 && five) {
        switch (mode) {
        case kModeStats: return render_kernel_ws<false, kBlock, kModeStats, true, true>;
        case kModeDump: return render_kernel_ws<false, kBlock, kModeDump, true, true>;
        case kModeTiming: return render_kernel_ws<false, kBlock, kModeTiming, true, true>;
        default: return render_kernel_ws<false, kBlock, kModeProduction, true, true>;
        }
    }
    if (lds16 && !tree && no_discs && mode == kModeProduction) return render_kernel_ws<true, kBlock, kModeProduction, false, false, kLeanNoDiscs>;
    if (lds16 && !tree && no_discs && mode == kModeTiming) return render_kernel_ws<true, kBlock, kModeTiming, false, false, kLeanNoDiscs>;
    if (lds16 && !tree) {
        switch (mode) {
        case kModeStats: return render_kernel_ws<true, kBlock, kModeStats, false>;
        case kModeDump: return render_kernel_ws<true, kBlock, kModeDump, false>;
        case kModeTiming: return render_kernel_ws<true, kBlock, kModeTiming, false>;
        default: return render_kernel_ws<true, kBlock, kModeProduction, false>;
        }
    }
    if (lds16) {
        switch (mode) {
        case kModeStats: return render_kernel_ws<true, kBlockLds, kModeStats>;
        case kModeDump: return render_kernel_ws<true, kBlockLds, kModeDump>;
        case kModeTiming: return render_kernel_ws<true, kBlockLds, kModeTiming>;
        default: return render_kernel_ws<true, kBlockLds, kModeProduction>;
        }
    }
    switch (mode) {
    case kModeStats: return render_kernel_ws<false, kBlock, kModeStats>;
    case kModeDump: return render_kernel_ws<false, kBlock, kModeDump>;
    case kModeTiming: return render_kernel_ws<false, kBlock, kModeTiming>;
    default: return render_kernel_ws<false, kBlock, kModeProduction>;
    }
}

// Workgroups per CU of one kernel variant for this scene (asked once per scene and variant, then remembered;
// the production variant is asked when the scene is created, so a render call makes no query).
// Whether the scene runs the build of its kernel made for five waves per SIMD (96 registers; 1 M spheres: 24.9 ->
// 24.0 ms, bench scene 10.3 -> 9.9 ms).  It does when five of its 256-thread workgroups fit the CU's LDS: their
// traversal stacks for a tree in HBM, stacks and a copy of the nodes each for a tree in LDS.  A deeper or larger tree
// keeps the four-wave build, which has all its registers (and, in LDS, all its tables).
int use_five_waves(rt_scene* sc, bool lds16, bool* five) {
    int& state = sc->five_waves[lds16 ? 1 : 0];
    if (state < 0) {
        int per_cu = 0;
        if (!lds16) {
            RT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ws_kernel(false, kModeProduction, true, true, sc->plain, sc->no_discs), kBlock, stack_bytes(sc)));
        } else if (sc->lds16_five_bytes != 0) {
            const ws_kernel_t k = ws_kernel(true, kModeProduction, true, true, sc->plain, sc->no_discs);
            RT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, int(kLdsPerCu / 2)));
            RT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, kBlock, lds16_launch_bytes(sc->lds16_five_bytes, sc->stack_depth, kBlock)));
        }
        if (std::getenv("RT_CREATE_LAPS")) std::fprintf(stderr, "[use_five_waves] lds16 %d: %d workgroups per CU with %zu bytes of LDS\n", int(lds16), per_cu, lds16 ? lds16_launch_bytes(sc->lds16_five_bytes, sc->stack_depth, kBlock) : stack_bytes(sc));
        state = per_cu >= 5 ? 1 : 0;
        if (const char* e = std::getenv("RT_FIVE_WAVES")) state = (std::atoi(e) && (!lds16 || sc->lds16_five_bytes != 0)) ? 1 : 0;  // tuning aid
    }
    *five = state == 1;
    return RT_OK;
}

// Kernel, workgroup size and LDS bytes of a launch of the LDS-resident kernel for this scene.
struct LdsLaunch {
    ws_kernel_t kernel;
    int block;
    size_t lds;
};
int lds16_launch(rt_scene* sc, int mode, LdsLaunch* out) {
    const bool tree = sc->dev.n_nodes != 0;
    bool five = false;
    if (tree) { if (int rc = use_five_waves(sc, true, &five)) return rc; }
    out->kernel = ws_kernel(true, mode, tree, five, sc->plain, sc->no_discs);
    // a scene without a tree runs the shade-only kernel: fewer registers, so 256-thread workgroups fill the SIMDs with
    // five waves or more instead of four
    out->block = (tree && !five) ? kBlockLds : kBlock;
    out->lds = five ? lds16_launch_bytes(sc->lds16_five_bytes, sc->stack_depth, kBlock) : lds16_launch_bytes(sc->lds16_scene_bytes, sc->stack_depth);
    return RT_OK;
}

// The work-item schedule of the scene's production kernel (make_params).  The choice of the build was made when the
// scene was created (rt_scene_create asks for the production kernel's occupancy), so this only reads it.
enum { kScheduleLds = 0, kScheduleHbm = 1, kScheduleLdsFive = 2, kScheduleHbmBig = 3 };
int schedule_of(const rt_scene* sc) {
    // a scene that does not fit the 256 MB Infinity Cache: see make_params
    if (!sc->lds_scene) return sc->info.device_bytes > (size_t(256) << 20) ? kScheduleHbmBig : kScheduleHbm;
    // (a scene without a tree runs the shade-only kernel at five or six waves and shares the five-wave schedule:
    // nested shells, chunks of 12 / 8 / 6 samples: 10.43 / 10.22 / 10.14 ms)
    return (sc->dev.n_nodes == 0 || sc->five_waves[1] == 1) ? kScheduleLdsFive : kScheduleLds;
}

int resident_blocks(rt_scene* sc, bool lockstep, bool lds16, int mode, int* out) {
    int& slot = sc->per_cu[lockstep ? 0 : 1 + (lds16 ? 4 : 0) + mode];
    if (slot == 0) {
        int per_cu = 0;
        if (lockstep) {
            RT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, render_kernel, kBlock, stack_bytes(sc)));
        } else if (lds16) {
            LdsLaunch L;
            if (int rc = lds16_launch(sc, mode, &L)) return rc;
            RT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(L.kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(kLdsPerCu / 2)));
            RT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, L.kernel, L.block, L.lds));
        } else {
            bool five = false;
            if (int rc = use_five_waves(sc, false, &five)) return rc;
            RT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ws_kernel(false, mode, true, five, sc->plain, sc->no_discs), kBlock, stack_bytes(sc)));
        }
        slot = per_cu < 1 ? 1 : per_cu;
    }
    *out = slot;
    return RT_OK;
}

int check_render_args(const rt_scene* scene, const rt_camera* cam, const rt_render_opts* opts, const char* who) {
    if (!scene || !cam || !opts) return fail(RT_ERR_INVALID_ARGUMENT, "%s: null argument", who);
    if (cam->horiz_res == 0 || cam->vert_res == 0) return fail(RT_ERR_INVALID_ARGUMENT, "%s: zero resolution", who);
    if (cam->horiz_res > 65535u || cam->vert_res > 65535u) return fail(RT_ERR_LIMIT, "%s: a frame side above 65535 pixels", who);
    if (uint64_t(cam->horiz_res) * cam->vert_res > 0x7fffffffull / 64)
        return fail(RT_ERR_LIMIT, "%s: more than 2^25 pixels", who);
    if (opts->sample_end <= opts->sample_begin)
        return fail(RT_ERR_INVALID_ARGUMENT, "%s: empty sample range [%u, %u)", who, opts->sample_begin, opts->sample_end);
    return RT_OK;
}

// tree_in_hbm: the render kernel will gather its nodes from memory (see chunk rule below)
// Most chunks a render of this frame may be cut into: the partial sums (one frame-sized slot per chunk) stay within
// RT_PARTIALS_BUDGET_BYTES, and never more than 64 slots
uint32_t chunk_cap(const rt_camera& cam) {
    const uint64_t slot = uint64_t(cam.horiz_res) * cam.vert_res * 3 * sizeof(float);
    return uint32_t(std::min<uint64_t>(64, std::max<uint64_t>(1, uint64_t(RT_PARTIALS_BUDGET_BYTES) / std::max<uint64_t>(slot, 1))));
}

// schedule: how the library chooses the samples per work item for a scene (see below)
RenderParams make_params(const rt_camera& cam, const rt_render_opts& o, int schedule, uint32_t max_chunks = 0) {
    const bool tree_in_hbm = schedule == kScheduleHbm || schedule == kScheduleHbmBig;
    if (max_chunks == 0) max_chunks = chunk_cap(cam);
    RenderParams P;
    P.sample_begin = o.sample_begin;
    P.sample_end = o.sample_end;
    P.depth = o.depth;
    P.flags = o.flags;
    P.key0 = uint32_t(o.seed);
    P.key1 = uint32_t(o.seed >> 32);
    P.n_pixels = cam.horiz_res * cam.vert_res;
    P.tiles_x = (cam.horiz_res + 7) / 8;
    P.tiles_y = (cam.vert_res + 7) / 8;
    // Scene in LDS: enough items to keep every lane of the chip busy to the end, about 70 tile items per resident
    // wave (measured at 1080p/100 spp with eight work counters: chunks of 17 samples 10.94 ms, 12 samples 10.82 ms,
    // 9 samples 10.93 ms, 4 samples 11.1 ms; at 2160p/125 spp flat from 63 down to 21).
    // Tree in HBM: chunks of two samples.  Lanes that draw pixels together get neighbours of one tile, and for as long
    // as they stay together their camera rays walk the same nodes, so a wave's 64 gathers fall into few cache lines;
    // every chunk boundary brings the wave's lanes together again (1 M spheres, 1080p/64 spp: chunks of 11 samples
    // 28.9 ms, 4 samples 26.3 ms, 2 samples 25.9 ms, 1 sample 26.2 ms; 7.8 M spheres/16 spp: 16 samples 15.0 ms,
    // 4 samples 10.0 ms, 1 sample 8.5 ms).
    const uint32_t spp = o.sample_end - o.sample_begin;
    if (o.samples_per_item) {
        P.chunk_spp = std::min<uint32_t>(o.samples_per_item, spp);
    } else if (tree_in_hbm) {
        const uint32_t cap = std::max<uint32_t>(max_chunks, 1u);
        // (a scene larger than the Infinity Cache does best with single samples: 7.8 M spheres, chunks of 1 / 2 / 4:
        // 7.93 / 8.30 / 9.41 ms; 1 M spheres, 1 / 2 / 3 / 4: 24.6 / 24.0 / 24.2 / 24.5)
        const uint32_t least = schedule == kScheduleHbmBig ? 1u : 2u;
        P.chunk_spp = std::min<uint32_t>(spp, std::max<uint32_t>(least, (spp + cap - 1) / cap));
    } else {
        // (the five-wave build has a quarter more waves to feed and does best with half as many items again: bench
        // scene, chunks of 15 / 12 / 10 / 8 / 5 samples: 10.44 / 10.24 / 9.97 / 9.90 / 9.95 ms; 2160p/125 spp: 63 / 32 /
        // 21 / 11 samples 47.3 / 46.9 / 46.9 / 47.5 ms)
        const uint64_t want_items = schedule == kScheduleLdsFive ? 27000000 : 18000000, px = std::max<uint32_t>(P.n_pixels, 1);  // pixel chunks = 64 x tile items
        uint32_t chunks = uint32_t(std::min<uint64_t>(std::max<uint64_t>(1, (want_items + px / 2) / px), std::max<uint32_t>(max_chunks, 1u)));
        chunks = std::max<uint32_t>(1, std::min<uint32_t>(chunks, spp));
        P.chunk_spp = (spp + chunks - 1) / chunks;
    }
    P.n_chunks = (spp + P.chunk_spp - 1) / P.chunk_spp;
    // Tree in HBM: counter c owns the c-th eighth of the items, a band of the frame.  Workgroups go round the eight
    // XCDs in launch order and start with counter (workgroup number mod 8), so an XCD's waves work on one band while it
    // lasts and that band's part of the tree stays in the XCD's own L2 (7.8 M spheres: 8.9 -> 8.6 ms; 1 M: 26.0 -> 25.7).
    // Scene in LDS: item i belongs to counter i mod 8, which spreads cheap and dear regions over all counters.
    P.band_items = tree_in_hbm ? (P.tiles_x * P.tiles_y * P.n_chunks + kCounters - 1) / kCounters : 0;
    if (const char* e = std::getenv("RT_BANDS")) {  // tuning aid
        P.band_items = std::atoi(e) ? (P.tiles_x * P.tiles_y * P.n_chunks + kCounters - 1) / kCounters : 0;
    }
    P.trav_threshold = kTravThreshold;
    P.inner_threshold = kInnerThreshold;
    P.leave_threshold = kLeaveThreshold;
    if (const char* e = std::getenv("RT_TRAV_THRESHOLD")) P.trav_threshold = uint32_t(std::atoi(e));    // tuning aid
    if (const char* e = std::getenv("RT_LEAVE_THRESHOLD")) P.leave_threshold = uint32_t(std::max(1, std::atoi(e)));
    if (const char* e = std::getenv("RT_INNER_THRESHOLD")) P.inner_threshold = uint32_t(std::max(1, std::atoi(e)));
    return P;
}

// eps of the render kernels, whose descent works relative to hbase / qbase: 2^-20 of the largest distance, on any
// axis, between a box plane and a path origin (origins lie on surfaces or on the camera's lens)
float descent_eps_rel(const rt_scene* sc, const rt_camera& c) {
    double m = sc->bound_rel;
    for (int a = 0; a < 3; ++a) {
        m = std::max(m, std::fabs(c.lookfrom[a] - double(sc->dev.hbase[a])) + std::fabs(c.aperture));
        m = std::max(m, std::fabs(c.lookfrom[a] - double(sc->dev.qbase[a])) + std::fabs(c.aperture));
    }
    return float(2.0 * m * (1.0 / 1048576.0));  // 2^-20: the centre / half-extent form takes two more roundings per axis
}

// 1 / (a length no segment reaches): with m = the largest distance, on any axis, of a box plane or a path origin
// from the reference point (descent_eps_rel's m = eps_rel * 2^19), two such points are at most 2 * sqrt(3) * m apart
float descent_t_scale(float eps_rel) {
    const double m = double(eps_rel) * 524288.0;
    return float(1.0 / std::max(4.0 * m, 1.0e-30));
}

double camera_abs(const rt_camera& c) {
    double m = 0.0;
    for (int a = 0; a < 3; ++a) m = std::max(m, std::fabs(c.lookfrom[a]) + std::fabs(c.aperture));
    return m;
}

}  // namespace

static void rt_multi_cache_forget(const rt_scene* scene);  // rt_render_multi's buffers for a handle list naming this scene

extern "C" {

int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void rt_scene_destroy(rt_scene* scene) {
    if (!scene) return;
    DeviceGuard guard;
    rt_multi_cache_forget(scene);
    (void)hipSetDevice(scene->device);
    if (scene->ev_done) { (void)hipEventSynchronize(scene->ev_done); (void)hipEventDestroy(scene->ev_done); }
    if (scene->d_partials) (void)hipFree(scene->d_partials);
    if (scene->d_wf) (void)hipFree(scene->d_wf);
    if (scene->d_wf_counts) (void)hipFree(scene->d_wf_counts);
    if (scene->d_frame_sum) (void)hipFree(scene->d_frame_sum);
    if (scene->d_frame_rgb) (void)hipFree(scene->d_frame_rgb);
    for (void* p : scene->allocations) (void)hipFree(p);
    delete scene;
}

int rt_scene_create(const rt_hittable* hittables, size_t n_hittables, const rt_texture* textures, size_t n_textures,
                    int device, rt_scene** out) {
    return rt_scene_create_ex(hittables, n_hittables, textures, n_textures, device, nullptr, out);
}

int rt_scene_create_ex(const rt_hittable* hittables, size_t n_hittables, const rt_texture* textures, size_t n_textures,
                       int device, const rt_scene_opts* scene_opts, rt_scene** out) {
    if (!out) return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: null out pointer");
    const uint32_t builder = scene_opts ? scene_opts->builder : uint32_t(RT_BUILDER_DEFAULT);
    if (builder > RT_BUILDER_LBVH_DEVICE) return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_create_ex: unknown builder %u", builder);
    *out = nullptr;
    if (!hittables && n_hittables) return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: null hittables");
    if (!textures && n_textures) return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: null textures");
    if (n_hittables > (size_t(1) << 30)) return fail(RT_ERR_LIMIT, "rt_scene_create: more than 2^30 hittables");
    auto t0 = std::chrono::steady_clock::now();
    const bool laps = std::getenv("RT_CREATE_LAPS") != nullptr;  // measurement aid: time since entry at each stage, on stderr
    auto lap = [&](const char* what) {
        if (laps) std::fprintf(stderr, "[rt_scene_create] %-18s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    };

    // validate and split by shape (before any device work: a bad scene is reported as such on any machine)
    std::vector<uint32_t> sphere_ids, disc_ids;
    for (size_t i = 0; i < n_hittables; ++i) {
        const rt_hittable& h = hittables[i];
        if (h.shape == RT_SHAPE_SPHERE) sphere_ids.push_back(uint32_t(i));
        else if (h.shape == RT_SHAPE_DISC) disc_ids.push_back(uint32_t(i));
        else return fail(RT_ERR_UNSUPPORTED, "rt_scene_create: hittable %zu has shape %d", i, h.shape);
        // The reference computes on with NaN or infinite geometry and paints garbage; here the f32 descent needs finite
        // box planes well inside the f32 range, so such a scene is refused.
        const double lim = 1.0e15;
        bool finite = std::isfinite(h.radius) && std::fabs(h.radius) <= lim;
        for (int a = 0; a < 3; ++a) finite = finite && std::isfinite(h.centre[a]) && std::fabs(h.centre[a]) <= lim;
        if (h.shape == RT_SHAPE_DISC)
            for (int a = 0; a < 3; ++a) finite = finite && std::isfinite(h.normal[a]);
        if (!finite) return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: hittable %zu has non-finite or out-of-range (> 1e15) geometry", i);
        if (h.material < RT_MAT_DIFFUSE || h.material > RT_MAT_EMITTER)
            return fail(RT_ERR_UNSUPPORTED, "rt_scene_create: hittable %zu has material %d", i, h.material);
        if (h.material == RT_MAT_TEXTUREMAP && (h.texture < 0 || size_t(h.texture) >= n_textures))
            return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: hittable %zu names texture %d of %zu", i, h.texture, n_textures);
    }
    for (size_t i = 0; i < n_textures; ++i)
        if (!textures[i].rgb8 || textures[i].width == 0 || textures[i].height == 0)
            return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: texture %zu is empty", i);

    lap("validated");
    int n_dev = rt_device_count();
    if (n_dev <= 0) return fail(RT_ERR_DEVICE, "rt_scene_create: no HIP device is visible");
    if (device < 0 || device >= n_dev) return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: device %d of %d", device, n_dev);

    DeviceGuard guard;
    rt_scene* sc = new rt_scene;
    sc->device = device;
    auto bail = [&](int code) { rt_scene_destroy(sc); return code; };
    if (hipSetDevice(device) != hipSuccess) return bail(fail(RT_ERR_DEVICE, "rt_scene_create: hipSetDevice(%d) failed", device));
    {   // the device's architecture and CU count, asked once per device (the query takes milliseconds)
        static std::mutex prop_mutex;
        static std::vector<std::pair<std::string, int>> known;  // [device] = {architecture, CUs}
        std::lock_guard<std::mutex> lock(prop_mutex);
        if (known.size() <= size_t(device)) known.resize(size_t(device) + 1);
        if (known[size_t(device)].first.empty()) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, device) != hipSuccess) return bail(fail(RT_ERR_DEVICE, "rt_scene_create: hipGetDeviceProperties failed"));
            known[size_t(device)] = {std::string(prop.gcnArchName), prop.multiProcessorCount};
        }
        const std::string& arch = known[size_t(device)].first;
        if (std::strncmp(arch.c_str(), "gfx950", 6) != 0)
            return bail(fail(RT_ERR_DEVICE, "rt_scene_create: device %d is %s, this library is built for gfx950 only", device, arch.c_str()));
        sc->n_cu = known[size_t(device)].second;
    }

    lap("device set");
    // Wide spheres: a sphere whose box covers at least half of the box around all the others (the ground sphere of
    // src/scenegen.rs:18-21 is the example) would be reached by nearly every ray whatever the tree looks like, and
    // its huge box would sit in the root.  Up to kMaxWide of them stay out of the tree and are tested once per
    // segment before the descent.  The result of a segment does not depend on this choice.
    std::vector<uint32_t> tree_ids = sphere_ids, wide_ids;
    auto box_of = [&](uint32_t id, double lo[3], double hi[3]) {
        const rt_hittable& h = hittables[id];
        for (int a = 0; a < 3; ++a) { lo[a] = h.centre[a] - std::fabs(h.radius); hi[a] = h.centre[a] + std::fabs(h.radius); }
    };
    auto half_area = [](const double lo[3], const double hi[3]) {
        const double x = std::max(hi[0] - lo[0], 0.0), y = std::max(hi[1] - lo[1], 0.0), z = std::max(hi[2] - lo[2], 0.0);
        return x * y + y * z + z * x;
    };
    // Nested shells: four or more spheres with one centre, each radius above the one before by 1e-4 of it and by 4e-6
    // (four times the reference's self-hit guard, so that of two neighbours at most one can be refused by it), the
    // smallest above zero.  `group` comes sorted by |radius|.
    auto well_separated = [&](const std::vector<uint32_t>& group) {
        bool separated = group.size() >= 4 && std::fabs(hittables[group[0]].radius) > 0.0;
        for (size_t g = 1; g < group.size() && separated; ++g) {
            const double a = std::fabs(hittables[group[g - 1]].radius), b = std::fabs(hittables[group[g]].radius);
            separated = b >= a * (1.0 + 1.0e-4) && b - a >= 4.0e-6;
        }
        return separated;
    };
    // the spheres of `ids` that share the centre of ids[i], from position i on, smallest |radius| first
    auto same_centre_from = [&](const std::vector<uint32_t>& ids, size_t i, std::vector<char>* used) {
        const double* ci = hittables[ids[i]].centre;
        std::vector<uint32_t> group;
        for (size_t j = i; j < ids.size(); ++j) {
            const double* cj = hittables[ids[j]].centre;
            if ((*used)[j] || cj[0] != ci[0] || cj[1] != ci[1] || cj[2] != ci[2]) continue;
            (*used)[j] = 1;
            group.push_back(ids[j]);
        }
        std::stable_sort(group.begin(), group.end(), [&](uint32_t a, uint32_t b) { return std::fabs(hittables[a].radius) < std::fabs(hittables[b].radius); });
        return group;
    };
    // A handful of spheres (the bundled scene.json) is cheaper to test one after the other while shading, where nearly
    // all lanes of a wave take part, than through a descent whose inner and leaf turns run half empty: such a scene
    // has no tree, like the reference's own linear path (src/lib.rs:46-49).  So is a scene of up to a dozen spheres
    // around a group of nested shells: a tree cannot tell concentric spheres apart, and the kernels test only two or
    // three of the group per segment (trav_begin).  A dozen spheres that lie apart are better off with a tree (a third
    // faster at twelve, level at five).
    bool no_tree = sphere_ids.size() <= size_t(kTinyScene);
    if (!no_tree && sphere_ids.size() <= size_t(kSmallScene)) {
        std::vector<char> used(sphere_ids.size(), 0);
        for (size_t i = 0; i < sphere_ids.size() && !no_tree; ++i)
            if (!used[i]) no_tree = well_separated(same_centre_from(sphere_ids, i, &used));
    }
    if (no_tree) { wide_ids = sphere_ids; tree_ids.clear(); }
    if (tree_ids.size() >= 2) {
        // Candidates are taken largest box first, each judged against the box around all spheres still in the tree.
        // Two passes over the scene whatever the outcome: the kMaxWide + 1 largest boxes, then the box around
        // everything else; the box "around all the others" of a candidate is that box joined with the other candidates'.
        const size_t n_cand = std::min<size_t>(size_t(kMaxWide) + 1, tree_ids.size());
        std::vector<std::pair<double, size_t>> cand;  // (area, position in tree_ids), largest first, earlier position first on ties
        for (size_t i = 0; i < tree_ids.size(); ++i) {
            double lo[3], hi[3];
            box_of(tree_ids[i], lo, hi);
            const double area = half_area(lo, hi);
            if (cand.size() < n_cand || area > cand.back().first) {
                auto pos = std::upper_bound(cand.begin(), cand.end(), area, [](double a, const std::pair<double, size_t>& c) { return a > c.first; });
                cand.insert(pos, {area, i});
                if (cand.size() > n_cand) cand.pop_back();
            }
        }
        std::vector<char> is_cand(tree_ids.size(), 0);
        for (const auto& c : cand) is_cand[c.second] = 1;
        double rlo[3] = {1e300, 1e300, 1e300}, rhi[3] = {-1e300, -1e300, -1e300};  // box around the non-candidates
        for (size_t i = 0; i < tree_ids.size(); ++i) {
            if (is_cand[i]) continue;
            double lo[3], hi[3];
            box_of(tree_ids[i], lo, hi);
            for (int a = 0; a < 3; ++a) { rlo[a] = std::min(rlo[a], lo[a]); rhi[a] = std::max(rhi[a], hi[a]); }
        }
        std::vector<size_t> taken;  // positions in tree_ids that become wide
        for (size_t c = 0; c < cand.size() && taken.size() < size_t(kMaxWide) && tree_ids.size() - taken.size() >= 2; ++c) {
            double ulo[3] = {rlo[0], rlo[1], rlo[2]}, uhi[3] = {rhi[0], rhi[1], rhi[2]};
            for (size_t o = c + 1; o < cand.size(); ++o) {  // the candidates not yet judged are still in the tree
                double lo[3], hi[3];
                box_of(tree_ids[cand[o].second], lo, hi);
                for (int a = 0; a < 3; ++a) { ulo[a] = std::min(ulo[a], lo[a]); uhi[a] = std::max(uhi[a], hi[a]); }
            }
            double blo[3], bhi[3], ilo[3], ihi[3];
            box_of(tree_ids[cand[c].second], blo, bhi);
            for (int a = 0; a < 3; ++a) { ilo[a] = std::max(blo[a], ulo[a]); ihi[a] = std::min(bhi[a], uhi[a]); }
            const bool overlaps = ilo[0] <= ihi[0] && ilo[1] <= ihi[1] && ilo[2] <= ihi[2];
            if (!overlaps || !(half_area(ilo, ihi) >= 0.5 * half_area(ulo, uhi))) break;
            taken.push_back(cand[c].second);
        }
        if (!taken.empty()) {
            std::vector<char> gone(tree_ids.size(), 0);
            for (size_t t : taken) { gone[t] = 1; wide_ids.push_back(tree_ids[t]); }
            size_t w = 0;
            for (size_t i = 0; i < tree_ids.size(); ++i)
                if (!gone[i]) tree_ids[w++] = tree_ids[i];
            tree_ids.resize(w);
        }
    }
    // The node encoders below store a box as centre and half extent, in which the absent second child of a
    // single-sphere tree (a far-away point box, rt_bvh.cpp) would be a box that a ray can enter.  The rules above
    // leave either no tree or one of at least two spheres; this keeps it so whatever happens to them later.
    if (tree_ids.size() == 1) { wide_ids.push_back(tree_ids[0]); tree_ids.clear(); }
    if (wide_ids.size() > size_t(kSmallScene))
        return bail(fail(RT_ERR_LIMIT, "rt_scene_create: more spheres outside the tree than the kernels test per segment"));
    std::sort(wide_ids.begin(), wide_ids.end());
    // wide spheres with the same centre next to one another (groups by their lowest ID, smallest |radius| first within
    // a group): the render kernels compute the centre-dependent half of the sphere test once per group, and of a
    // group of well separated concentric spheres (nested shells) they test only the three a segment can meet first
    uint32_t wide_same_centre = 0;
    int32_t shell_first = 0, shell_count = 0;
    double shell_rmin2 = 0.0;
    std::vector<double> shell_r2;
    {
        std::vector<uint32_t> grouped;
        std::vector<char> used(wide_ids.size(), 0);
        for (size_t i = 0; i < wide_ids.size(); ++i) {
            if (used[i]) continue;
            const std::vector<uint32_t> group = same_centre_from(wide_ids, i, &used);
            if (well_separated(group) && int32_t(group.size()) > shell_count) {
                shell_first = int32_t(grouped.size());
                shell_count = int32_t(group.size());
                shell_rmin2 = hittables[group.front()].radius * hittables[group.front()].radius;  // the smallest shell sets the reach
                shell_r2.clear();
                for (uint32_t id : group) shell_r2.push_back(hittables[id].radius * hittables[id].radius);
                shell_r2.resize(size_t(kShellTable), HUGE_VAL);  // padding the kernels' binary search never counts
            }
            for (size_t g = 0; g < group.size(); ++g) {
                if (g > 0 && grouped.size() < 32) wide_same_centre |= 1u << grouped.size();
                grouped.push_back(group[g]);
            }
        }
        wide_ids.swap(grouped);
    }
    // the assembly descent forms a node's byte offset (index * 32) in 32 bits
    if (tree_ids.size() > (size_t(1) << 26)) return bail(fail(RT_ERR_LIMIT, "rt_scene_create: more than 2^26 spheres in the tree"));

    lap("wide chosen");
    // BVH over the other spheres
    std::vector<double> sph4(4 * tree_ids.size());
    for (size_t i = 0; i < tree_ids.size(); ++i) {
        const rt_hittable& h = hittables[tree_ids[i]];
        sph4[4 * i] = h.centre[0]; sph4[4 * i + 1] = h.centre[1]; sph4[4 * i + 2] = h.centre[2]; sph4[4 * i + 3] = h.radius;
    }
    lap("sph4 filled");
    uint32_t lg = 0;
    while ((size_t(1) << lg) < tree_ids.size()) ++lg;
    BuiltBvh bvh;
    // RT_BUILDER_LBVH_DEVICE on a scene that cannot sit in LDS anyway: the tree, its quantised copy and the sphere
    // tables are made on the device and stay there (bvh.nodes / bvh.leaf_order stay empty); smaller scenes come back
    // to the host, which also encodes the f16 copy for the LDS-resident kernel.
    constexpr size_t kResidentMin = 4096;
    const bool resident = builder == RT_BUILDER_LBVH_DEVICE && tree_ids.size() >= kResidentMin;
    double* d_build_sph = nullptr;    // the tree's spheres in tree_ids order
    Node* d_tree_nodes = nullptr;
    uint32_t* d_leaf_order = nullptr;  // leaf position -> index into tree_ids
    if (resident) {
        if (int brc = build_lbvh_resident(sph4.data(), tree_ids.size(), &d_build_sph, &d_tree_nodes, &d_leaf_order, &bvh.depth,
                                          bvh.scene_lo, bvh.scene_hi))
            return bail(brc);
        for (void* p : {static_cast<void*>(d_build_sph), static_cast<void*>(d_tree_nodes), static_cast<void*>(d_leaf_order)})
            sc->allocations.push_back(p);  // from here on rt_scene_destroy frees them
        sc->info.device_bytes += (tree_ids.size() - 1) * sizeof(Node);
    } else if (builder == RT_BUILDER_LBVH_DEVICE) {
        if (int brc = build_lbvh_device(sph4.data(), tree_ids.size(), &bvh)) return bail(brc);
    } else {
        build_bvh(sph4.data(), tree_ids.size(), lg + 7, &bvh);
    }
    lap("tree built");
    const size_t n_nodes = resident ? tree_ids.size() - 1 : bvh.nodes.size();
    sc->stack_depth = bvh.depth + 2;  // one pending entry per level, plus slot 0 (the sentinel below every stack)
    // bound = largest |coordinate| of any surface point: path origins lie on surfaces (the wide spheres' too), and the
    // origin widening of the f32 descent is sized from it
    double bound = 0.0;
    for (int a = 0; a < 3; ++a) bound = std::max(bound, std::max(std::fabs(bvh.scene_lo[a]), std::fabs(bvh.scene_hi[a])));
    for (uint32_t id : wide_ids)
        for (int a = 0; a < 3; ++a) bound = std::max(bound, std::fabs(hittables[id].centre[a]) + std::fabs(hittables[id].radius));

    // SoA primitive tables: spheres in leaf order, then discs in scene order
    const size_t n_s = sphere_ids.size(), n_d = disc_ids.size();
    std::vector<double4> h_sph(n_s);
    std::vector<DiscDev> h_disc(n_d);
    std::vector<float4> h_mat(n_s + n_d);
    std::vector<int32_t> h_kind(n_s + n_d), h_id(n_s + n_d);
    std::vector<TexMatDev> h_texmat;
    auto fill_material = [&](size_t k, uint32_t id) {
        const rt_hittable& h = hittables[id];
        float w = 0.0f;
        if (h.material == RT_MAT_METAL) w = float(h.fuzz);
        if (h.material == RT_MAT_DIELECTRIC) w = float(h.refractive_index);
        h_mat[k] = make_float4(float(h.albedo[0]), float(h.albedo[1]), float(h.albedo[2]), w);
        int32_t word = h.material;
        if (h.material == RT_MAT_TEXTUREMAP) {
            TexMatDev tm;
            double up[3] = {h.orient_up[0], h.orient_up[1], h.orient_up[2]};
            double ar[3] = {h.orient_around[0], h.orient_around[1], h.orient_around[2]};
            normalize3d(up);  // src/materials.rs:64-68: both are normalised, the pair is NOT re-orthogonalised
            normalize3d(ar);
            double a1[3] = {up[1] * ar[2] - up[2] * ar[1], up[2] * ar[0] - up[0] * ar[2], up[0] * ar[1] - up[1] * ar[0]};
            for (int a = 0; a < 3; ++a) { tm.up[a] = float(up[a]); tm.a0[a] = float(ar[a]); tm.a1[a] = float(a1[a]); }
            tm.tex = h.texture;
            word |= int32_t(h_texmat.size()) << 8;
            h_texmat.push_back(tm);
        }
        h_kind[k] = word;
        h_id[k] = int32_t(id);
    };
    const size_t n_tree = tree_ids.size(), n_wide = wide_ids.size();
    for (size_t k = 0; k < n_s; ++k) {
        // (a tree that stays on the device: build order here, the device puts the tables into leaf order below)
        const uint32_t id = k < n_tree ? tree_ids[resident ? k : bvh.leaf_order[k]] : wide_ids[k - n_tree];
        const rt_hittable& h = hittables[id];
        h_sph[k] = make_double4(h.centre[0], h.centre[1], h.centre[2], h.radius);
        fill_material(k, id);
    }
    for (size_t i = 0; i < n_d; ++i) {
        const rt_hittable& h = hittables[disc_ids[i]];
        DiscDev d;
        d.cx = h.centre[0]; d.cy = h.centre[1]; d.cz = h.centre[2];
        d.nx = h.normal[0]; d.ny = h.normal[1]; d.nz = h.normal[2];
        d.r = h.radius; d.pad = 0.0;
        h_disc[i] = d;
        for (int a = 0; a < 3; ++a) bound = std::max(bound, std::fabs(h.centre[a]) + std::fabs(h.radius));
        fill_material(n_s + i, disc_ids[i]);
    }
    sc->bound_abs = bound;

    lap("tables filled");
    // textures as RGBA8 so one texel is one aligned 4-byte load
    std::vector<TexImgDev> h_teximg(n_textures);
    for (size_t i = 0; i < n_textures; ++i) {
        const size_t n_px = size_t(textures[i].width) * textures[i].height;
        std::vector<uchar4> rgba(n_px);
        for (size_t p = 0; p < n_px; ++p)
            rgba[p] = make_uchar4(textures[i].rgb8[3 * p], textures[i].rgb8[3 * p + 1], textures[i].rgb8[3 * p + 2], 255);
        const uchar4* d_px = nullptr;
        int rc = upload(sc, rgba, &d_px);
        if (rc != RT_OK) return bail(rc);
        h_teximg[i] = TexImgDev{d_px, textures[i].width, textures[i].height};
    }

    int rc;
    lap("textures");
    // box around every finite node box (the f32 boxes are rounded outward, so it may exceed the spheres' box by an ulp)
    double glo[3], ghi[3];
    for (int a = 0; a < 3; ++a) { glo[a] = bvh.scene_lo[a]; ghi[a] = bvh.scene_hi[a]; }
    auto take_node = [&](const Node& nd) {
        for (int a = 0; a < 3; ++a) {
            if (nd.lo0[a] < 3.0e38f) { glo[a] = std::min(glo[a], double(nd.lo0[a])); ghi[a] = std::max(ghi[a], double(nd.hi0[a])); }
            if (nd.lo1[a] < 3.0e38f) { glo[a] = std::min(glo[a], double(nd.lo1[a])); ghi[a] = std::max(ghi[a], double(nd.hi1[a])); }
        }
    };
    if (resident) {  // the root's two boxes hold every box below them (outward rounding is monotonic)
        Node root;
        if (hipMemcpy(&root, d_tree_nodes, sizeof(Node), hipMemcpyDeviceToHost) != hipSuccess)
            return bail(fail(RT_ERR_DEVICE, "rt_scene_create: download of the root node failed"));
        take_node(root);
    }
    for (const Node& nd : bvh.nodes) {
        if (!(nd.lo0[0] < 3.0e38f) || !(nd.lo1[0] < 3.0e38f))  // see the single-sphere rule above
            return bail(fail(RT_ERR_LIMIT, "rt_scene_create: a tree node with an absent child reached the node encoders"));
        take_node(nd);
    }
    // f16 copy of the nodes for the LDS-resident kernel: scenes that leave room for two workgroups per CU.  The
    // values are (plane - hbase) / hstep with hbase = the centre of the box above and hstep a power of two that puts
    // the half extent in [512, 1024): f16 then resolves 2^-11 of a plane's distance from the scene's centre, whatever
    // the scene's position and scale (taken as absolute coordinates, half floats have an ulp of 8 at 1e4 and end at 65504).
    for (int a = 0; a < 3; ++a) {
        const float centre = float(0.5 * (glo[a] + ghi[a]));
        const double half = std::max(std::max(ghi[a] - double(centre), double(centre) - glo[a]), 1.0e-30);
        int e = 0;
        (void)std::frexp(half, &e);  // half = m * 2^e, m in [0.5, 1)
        sc->dev.hbase[a] = centre;
        sc->dev.hstep[a] = std::ldexp(1.0f, e - 10);  // half / hstep = m * 2^10 in [512, 1024)
    }
    {
        const size_t n_prims = n_s + n_d;
        // node halves: the first array takes kQ1Offset bytes whatever it holds (see inner_loop_lds16)
        const size_t scene_bytes = kQ1Offset + 16 * n_nodes + 32 * n_s + 16 * n_prims + 2 * 16 * ((n_prims + 3) / 4) +
                                   (shell_count > 0 ? 8 * size_t(kShellTable) : 0);
        if (n_s > 0 && !resident && 16 * n_nodes <= kQ1Offset && lds16_launch_bytes(scene_bytes, sc->stack_depth) <= kLdsPerCu / 2) {
            std::vector<Node16> n16(bvh.nodes.size());
            auto rel = [&](float plane, int a) { return (double(plane) - double(sc->dev.hbase[a])) / double(sc->dev.hstep[a]); };  // exact
            // centre = the half float nearest to the box's middle, half extent = the smallest half float that still
            // reaches both faces from that centre
            auto encode = [&](float lo, float hi, int a, uint16_t* c_out, uint16_t* h_out) {
                if (!(lo < 3.0e38f)) { *c_out = half_bits(_Float16(65504.0f)); *h_out = 0; return; }  // far-away point box
                const double l = rel(lo, a), u = rel(hi, a);
                const _Float16 c = _Float16(0.5 * (l + u));
                *c_out = half_bits(c);
                *h_out = half_round_up(std::max(u - double(c), double(c) - l));
            };
            for (size_t i = 0; i < bvh.nodes.size(); ++i) {
                const Node& nd = bvh.nodes[i];
                for (int a = 0; a < 3; ++a) {
                    encode(nd.lo0[a], nd.hi0[a], a, &n16[i].h[4 * a], &n16[i].h[4 * a + 2]);
                    encode(nd.lo1[a], nd.hi1[a], a, &n16[i].h[4 * a + 1], &n16[i].h[4 * a + 3]);
                }
                // an inner link is 16 * node index: the byte offset of the node's first half in the staged array
            n16[i].child0 = nd.child0 >= 0 ? nd.child0 * 16 : nd.child0; n16[i].child1 = nd.child1 >= 0 ? nd.child1 * 16 : nd.child1;
            }
            if ((rc = upload(sc, n16, &sc->d_nodes16)) != RT_OK) return bail(rc);
            sc->lds16_scene_bytes = scene_bytes;
            sc->lds_scene = true;
            if (n_nodes != 0 && 16 * n_nodes <= kQ1OffsetFive) sc->lds16_five_bytes = kQ1OffsetFive + 16 * n_nodes;
        }
    }
    {
        // 16-bit grid over the box of the tree's spheres; base and step are f32 values first, the indices are then
        // chosen in f64 against exactly those values, so that base + q * step never cuts into a box
        float base[3], step[3];
        for (int a = 0; a < 3; ++a) {
            base[a] = std::nextafter(float(glo[a]), -INFINITY);
            step[a] = std::nextafter(float(std::max(ghi[a] - double(base[a]), 1.0e-30) / 65535.0), INFINITY);
            sc->dev.qbase[a] = base[a];
            sc->dev.qstep[a] = step[a];
        }
        auto q_down = [&](float v, int a) -> uint16_t {
            if (!(v < 3.0e38f)) return 65535;  // the far-away point box of a single-sphere tree becomes the grid's far corner
            double q = std::floor((double(v) - double(base[a])) / double(step[a]));
            q = std::min(std::max(q, 0.0), 65535.0);
            while (q > 0.0 && double(base[a]) + q * double(step[a]) > double(v)) q -= 1.0;
            return uint16_t(q);
        };
        auto q_up = [&](float v, int a) -> uint16_t {
            if (!(v < 3.0e38f)) return 65535;
            double q = std::ceil((double(v) - double(base[a])) / double(step[a]));
            q = std::min(std::max(q, 0.0), 65535.0);
            while (q < 65535.0 && double(base[a]) + q * double(step[a]) < double(v)) q += 1.0;
            return uint16_t(q);
        };
        // large trees are quantised by the device kernel whichever builder made them (the host loop below takes
        // 40 ms per million nodes); the f32 nodes are on the device already or go there now
        const bool encode_on_device = resident || n_nodes >= kResidentMin;
        if (encode_on_device && !resident) {
            if ((rc = upload(sc, bvh.nodes, &sc->dev.nodes)) != RT_OK) return bail(rc);
            d_tree_nodes = const_cast<Node*>(sc->dev.nodes);
        }
        if (encode_on_device) {
            void* p = nullptr;
            if (hipMalloc(&p, n_nodes * sizeof(Node32)) != hipSuccess) return bail(fail(RT_ERR_DEVICE, "rt_scene_create: hipMalloc of the quantised nodes failed"));
            sc->allocations.push_back(p);
            sc->info.device_bytes += n_nodes * sizeof(Node32);
            hipLaunchKernelGGL(encode32_kernel, dim3(uint32_t((n_nodes + kBlock - 1) / kBlock)), dim3(kBlock), 0, nullptr, d_tree_nodes,
                               uint32_t(n_nodes), make_float3(base[0], base[1], base[2]), make_float3(step[0], step[1], step[2]),
                               static_cast<Node32*>(p));
            if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess)
                return bail(fail(RT_ERR_DEVICE, "rt_scene_create: the node encoder failed"));
            sc->dev.nodes32 = static_cast<const Node32*>(p);
        }
        std::vector<Node32> n32(encode_on_device ? 0 : bvh.nodes.size());
        for (size_t i = 0; i < n32.size(); ++i) {
            const Node& nd = bvh.nodes[i];
            for (int a = 0; a < 3; ++a) {
                // grid bounds (rounded outward), then centre and half extent in grid units: the centre is a grid
                // point, the half extent reaches the farther bound
                const uint32_t l0 = q_down(nd.lo0[a], a), u0 = q_up(nd.hi0[a], a), l1 = q_down(nd.lo1[a], a), u1 = q_up(nd.hi1[a], a);
                const uint32_t c0 = (l0 + u0) / 2, c1 = (l1 + u1) / 2;
                n32[i].q[4 * a] = uint16_t(c0); n32[i].q[4 * a + 1] = uint16_t(c1);
                n32[i].q[4 * a + 2] = uint16_t(std::max(u0 - c0, c0 - l0)); n32[i].q[4 * a + 3] = uint16_t(std::max(u1 - c1, c1 - l1));
            }
            n32[i].child0 = nd.child0; n32[i].child1 = nd.child1;
        }
        if (!encode_on_device && (rc = upload(sc, n32, &sc->dev.nodes32)) != RT_OK) return bail(rc);
    }
    lap("nodes encoded");
    if (resident) {
        sc->dev.nodes = d_tree_nodes;
        // the four sphere tables: uploaded in build order, put into leaf order by the device
        auto in_leaf_order = [&](auto& host, auto** final_ptr) -> int {
            using T = typename std::remove_reference<decltype(host)>::type::value_type;
            const T* staged = nullptr;
            int r = upload(sc, host, &staged);
            if (r != RT_OK) return r;
            void* p = nullptr;
            RT_HIP(hipMalloc(&p, host.size() * sizeof(T)));
            sc->allocations.push_back(p);
            hipLaunchKernelGGL(gather_kernel<T>, dim3(uint32_t((host.size() + kBlock - 1) / kBlock)), dim3(kBlock), 0, nullptr, staged,
                               d_leaf_order, uint32_t(n_tree), uint32_t(host.size()), static_cast<T*>(p));
            RT_HIP(hipGetLastError());
            RT_HIP(hipDeviceSynchronize());
            // the staged copy was the last allocation but one
            sc->allocations.erase(sc->allocations.end() - 2);
            (void)hipFree(const_cast<T*>(staged));
            *final_ptr = static_cast<const T*>(p);
            return RT_OK;
        };
        if ((rc = in_leaf_order(h_sph, &sc->dev.sph)) != RT_OK) return bail(rc);
        if ((rc = in_leaf_order(h_mat, &sc->dev.mat)) != RT_OK) return bail(rc);
        if ((rc = in_leaf_order(h_kind, &sc->dev.kind)) != RT_OK) return bail(rc);
        if ((rc = in_leaf_order(h_id, &sc->dev.prim_id)) != RT_OK) return bail(rc);
        // build-time buffers: the spheres in build order and the leaf order
        for (void* p : {static_cast<void*>(d_build_sph), static_cast<void*>(d_leaf_order)}) {
            sc->allocations.erase(std::find(sc->allocations.begin(), sc->allocations.end(), p));
            (void)hipFree(p);
        }
    } else {
        if (!sc->dev.nodes && (rc = upload(sc, bvh.nodes, &sc->dev.nodes)) != RT_OK) return bail(rc);
        if ((rc = upload(sc, h_sph, &sc->dev.sph)) != RT_OK) return bail(rc);
        if ((rc = upload(sc, h_mat, &sc->dev.mat)) != RT_OK) return bail(rc);
        if ((rc = upload(sc, h_kind, &sc->dev.kind)) != RT_OK) return bail(rc);
        if ((rc = upload(sc, h_id, &sc->dev.prim_id)) != RT_OK) return bail(rc);
    }
    lap("tables on device");
    if ((rc = upload(sc, h_disc, &sc->dev.discs)) != RT_OK) return bail(rc);
    if ((rc = upload(sc, h_texmat, &sc->dev.texmat)) != RT_OK) return bail(rc);
    if ((rc = upload(sc, h_teximg, &sc->dev.teximg)) != RT_OK) return bail(rc);
    sc->dev.n_nodes = int32_t(n_nodes);
    sc->dev.n_spheres = int32_t(n_s);
    sc->dev.n_tree = int32_t(n_tree);
    sc->dev.n_wide = int32_t(n_wide);
    sc->dev.wide_same_centre = wide_same_centre;
    sc->dev.shell_first = shell_first;
    sc->dev.shell_count = shell_count;
    sc->dev.shell_far2 = 1.0e6 * shell_rmin2;
    if ((rc = upload(sc, shell_r2, &sc->dev.shell_r2)) != RT_OK) return bail(rc);
    sc->dev.n_discs = int32_t(n_d);
    sc->plain = n_d == 0 && shell_count == 0 && h_texmat.empty();
    for (size_t i = 0; i < n_hittables && sc->plain; ++i) sc->plain = hittables[i].material <= RT_MAT_DIELECTRIC;
    sc->no_discs = n_d == 0;
    if (std::getenv("RT_NO_PLAIN")) sc->plain = sc->no_discs = false;  // measurement aid: the build with every branch
    sc->dev.eps = 0.0f;
    sc->dev.eps_rel = 0.0f;
    {   // every surface's box against the two reference points of the relative descent
        double rel = 0.0;
        auto take = [&](const double c[3], double r) {
            for (int a = 0; a < 3; ++a)
                for (float base : {sc->dev.hbase[a], sc->dev.qbase[a]})
                    rel = std::max(rel, std::fabs(c[a] - double(base)) + std::fabs(r));
        };
        for (size_t i = 0; i < n_hittables; ++i) take(hittables[i].centre, hittables[i].radius);
        sc->bound_rel = rel;
    }
    {
        void* p = nullptr;
        if (hipMalloc(&p, kCounterBlockBytes) != hipSuccess) return bail(fail(RT_ERR_DEVICE, "rt_scene_create: hipMalloc of the work counter failed"));
        sc->allocations.push_back(p);
        sc->d_work_counter = static_cast<uint32_t*>(p);
    }
    if (stack_bytes(sc) > 64 * 1024)
        return bail(fail(RT_ERR_LIMIT, "rt_scene_create: BVH depth %u needs a %zu-byte LDS stack", bvh.depth, stack_bytes(sc)));
    if (hipEventCreateWithFlags(&sc->ev_done, hipEventDisableTiming) != hipSuccess)
        return bail(fail(RT_ERR_DEVICE, "rt_scene_create: hipEventCreate failed"));
    {   // the production kernel's residency, asked here so that no render call has to
        int per_cu = 0;
        if ((rc = resident_blocks(sc, false, sc->lds_scene, kModeProduction, &per_cu)) != RT_OK) return bail(rc);
    }

    lap("rest");
    sc->info.n_hittables = n_hittables;
    sc->info.n_spheres = n_s;
    sc->info.n_discs = n_d;
    sc->info.n_nodes = n_nodes;
    sc->info.bvh_depth = bvh.depth;
    sc->info.stack_depth = sc->stack_depth;
    sc->info.device = device;
    sc->info.lds_resident = sc->lds_scene ? 1 : 0;
    sc->info.n_wide = uint32_t(n_wide);
    sc->info.n_shells = uint32_t(sc->dev.shell_count);
    sc->info.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    *out = sc;
    return RT_OK;
}

int rt_scene_get_info(const rt_scene* scene, rt_scene_info* out) {
    if (!scene || !out) return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_get_info: null argument");
    *out = scene->info;
    return RT_OK;
}

namespace {

// Grows the scene's partial-sum buffer to `need` bytes.  Waits for the handle's last render first: it may still be
// writing the old buffer.
int reserve_partials(rt_scene* scene, size_t need) {
    if (need <= scene->partials_bytes) return RT_OK;
    RT_HIP(hipEventSynchronize(scene->ev_done));
    if (scene->d_partials) { (void)hipFree(scene->d_partials); scene->d_partials = nullptr; scene->partials_bytes = 0; }
    RT_HIP(hipMalloc(reinterpret_cast<void**>(&scene->d_partials), need));
    scene->partials_bytes = need;
    return RT_OK;
}

size_t partials_needed(const RenderParams& P) {
    return P.n_chunks > 1 ? size_t(P.n_chunks) * P.n_pixels * 3 * sizeof(float) : 0;
}

// RT_RENDER_KERNEL_WAVEFRONT: the render as a pipeline of kernels with queues in HBM (see wf_trace_kernel).  The pool
// of path slots belongs to the scene and is made on first use (104 bytes per slot, at most 2^25 slots = 3.5 GB); a
// frame with more paths is rendered in batches of whole pixels.  Per batch: one generate, min(depth, 10) x (trace +
// shade), one finish for the paths that live longer, one accumulate; nothing comes back to the host in between.
int launch_wavefront(rt_scene* scene, const SceneDev& S, const CamDev& C, const RenderParams& P, float* d_sum_rgb, hipStream_t st,
                     float* d_partials) {
    const uint32_t spp = P.sample_end - P.sample_begin;
    uint32_t max_slots = kWfMaxSlots;
    if (const char* e = std::getenv("RT_WF_SLOTS")) max_slots = uint32_t(std::min<long long>(std::max<long long>(std::atoll(e), 1), kWfMaxSlots));  // test aid: small pools, many batches
    if (spp > max_slots) return fail(RT_ERR_LIMIT, "wavefront render: %u samples per pixel exceed the pool of %u path slots", spp, max_slots);
    const uint64_t n_paths = uint64_t(P.n_pixels) * spp;
    const size_t slots = size_t(std::min<uint64_t>(n_paths, (max_slots / spp) * uint64_t(spp)));
    const size_t count_bytes = (size_t(std::min<uint32_t>(P.depth, kWfQueuedBounces)) + 2) * kWfCountsPerBounce * sizeof(uint32_t);
    const size_t sub_cap = (slots + kWfSub - 1) / kWfSub + kWfSubMargin;
    const size_t pool_bytes = slots * (48 + 8 + 12 + 12 + 4) + (1 + kWfClasses) * kWfSub * sub_cap * sizeof(uint32_t) + 256;
    if (scene->wf_slots < slots || !scene->d_wf) {
        RT_HIP(hipEventSynchronize(scene->ev_done));
        if (scene->d_wf) { (void)hipFree(scene->d_wf); scene->d_wf = nullptr; scene->wf_slots = 0; }
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&scene->d_wf), pool_bytes));
        scene->wf_slots = slots;
    }
    if (count_bytes > scene->wf_counts_bytes) {
        RT_HIP(hipEventSynchronize(scene->ev_done));
        if (scene->d_wf_counts) { (void)hipFree(scene->d_wf_counts); scene->d_wf_counts = nullptr; scene->wf_counts_bytes = 0; }
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&scene->d_wf_counts), count_bytes));
        scene->wf_counts_bytes = count_bytes;
    }
    WfState W;
    unsigned char* base = scene->d_wf;
    const size_t cap = scene->wf_slots;
    const size_t pool_sub_cap = (cap + kWfSub - 1) / kWfSub + kWfSubMargin;  // of the pool as it was made (it may be larger than this frame needs)
    W.ray = reinterpret_cast<double*>(base); base += cap * 48;
    W.hit_t = reinterpret_cast<double*>(base); base += cap * 8;
    W.thr = reinterpret_cast<float*>(base); base += cap * 12;
    W.colour = reinterpret_cast<float*>(base); base += cap * 12;
    W.hit_k = reinterpret_cast<int32_t*>(base); base += cap * 4;
    W.rayq = reinterpret_cast<uint32_t*>(base); base += size_t(kWfSub) * pool_sub_cap * 4;
    W.matq = reinterpret_cast<uint32_t*>(base);
    W.counts = scene->d_wf_counts;
    W.sub_cap = uint32_t(pool_sub_cap);
    (void)sub_cap;
    const bool lds16 = scene->lds_scene && scene->dev.n_nodes != 0 && scene->lds16_five_bytes != 0;
    const size_t trace_lds = lds16 ? lds16_launch_bytes(scene->lds16_five_bytes, scene->stack_depth, kBlock) - kCamLdsBytes : stack_bytes(scene);
    if (lds16) RT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wf_trace_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(kLdsPerCu / 2)));
    const uint32_t batch_pixels = uint32_t(slots / spp);
    for (uint32_t pixel0 = 0; pixel0 < P.n_pixels; pixel0 += batch_pixels) {
        const uint32_t npix = std::min<uint32_t>(batch_pixels, P.n_pixels - pixel0);
        const uint32_t n = npix * spp;
        // (grids are multiples of kWfSub, so that every sub-queue gets the output of the same number of workgroups)
        auto grid_of = [&](uint64_t per_cu) {
            const uint64_t g = std::min<uint64_t>((uint64_t(n) + kBlock - 1) / kBlock, uint64_t(scene->n_cu) * per_cu);
            return uint32_t((g + kWfSub - 1) / kWfSub * kWfSub);
        };
        const uint32_t full = grid_of(8), trace_grid = grid_of(5);
        RT_HIP(hipMemsetAsync(W.counts, 0, count_bytes, st));
        hipLaunchKernelGGL(wf_generate_kernel, dim3(full), dim3(kBlock), 0, st, C, P, W, pixel0, n);
        const uint32_t queued_bounces = std::min<uint32_t>(P.depth, kWfQueuedBounces);
        for (uint32_t b = 0; b < queued_bounces; ++b) {
            if (lds16) hipLaunchKernelGGL(wf_trace_kernel<true>, dim3(trace_grid), dim3(kBlock), trace_lds, st, S, P, W, scene->d_nodes16, b);
            else hipLaunchKernelGGL(wf_trace_kernel<false>, dim3(trace_grid), dim3(kBlock), trace_lds, st, S, P, W, static_cast<const Node16*>(nullptr), b);
            hipLaunchKernelGGL(wf_shade_kernel, dim3(full), dim3(kBlock), 0, st, S, P, W, pixel0, b);
        }
        if (queued_bounces < P.depth)
            hipLaunchKernelGGL(wf_finish_kernel, dim3(trace_grid), dim3(kBlock), stack_bytes(scene), st, S, P, W, pixel0, queued_bounces);
        const uint32_t n_acc = npix * P.n_chunks;
        hipLaunchKernelGGL(wf_accumulate_kernel, dim3((n_acc + kBlock - 1) / kBlock), dim3(kBlock), 0, st, P, W, pixel0, npix, d_sum_rgb, d_partials);
        RT_HIP(hipGetLastError());
    }
    return RT_OK;
}

// The body of rt_render_device.  work_counter / partials: the scene's own, or the caller's (the entry point that
// allocates nothing brings its own).  d_dump: per-path colours of the kModeDump build, else null.
int launch_render(rt_scene* scene, const rt_camera* cam, const rt_render_opts* opts, const RenderParams& P, float* d_sum_rgb,
                  hipStream_t st, uint32_t* d_work_counter, float* d_partials, float* d_dump) {
    SceneDev S = scene->dev;
    S.eps = descent_eps(scene, camera_abs(*cam));
    S.eps_rel = descent_eps_rel(scene, *cam);
    S.t_scale = descent_t_scale(S.eps_rel);
    const CamDev C = make_cam(*cam);
    const uint64_t n_work = uint64_t(P.tiles_x) * P.tiles_y * 64u * P.n_chunks;
    // one render of a handle at a time on the device: they share the work counter and the partial sums.  A stream
    // that is being captured into a graph takes neither the wait nor the record: the event would become part of the
    // capture, and the graph's own order (its nodes follow one another on the captured stream) stands in for it;
    // keeping replays of the graph apart from other renders of the same handle is then the caller's business.
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (st && hipStreamIsCapturing(st, &capture) != hipSuccess) { (void)hipGetLastError(); capture = hipStreamCaptureStatusNone; }
    const bool capturing = capture == hipStreamCaptureStatusActive;
    if (!capturing) RT_HIP(hipStreamWaitEvent(st, scene->ev_done, 0));
    RT_HIP(hipMemsetAsync(d_work_counter, 0, kCounterBlockBytes, st));
    auto grid_for = [&](int per_cu, int block) {
        uint64_t blocks = uint64_t(scene->n_cu) * uint64_t(per_cu < 1 ? 1 : per_cu);
        return dim3(uint32_t(std::min<uint64_t>(blocks, (n_work + block - 1) / block)));
    };
    int mode = kModeProduction;
    if (d_dump) mode = kModeDump;
    else if (opts->flags & RT_RENDER_STATS) mode = kModeStats;
    else if (opts->flags & RT_RENDER_TIMING) mode = kModeTiming;
    const uint32_t depth_slots = std::max<uint32_t>(scene->stack_depth, 1u);
    int per_cu = 0, rc;
    if ((opts->flags & RT_RENDER_KERNEL_WAVEFRONT) && mode == kModeProduction) {
        if (d_work_counter != scene->d_work_counter)
            return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_device_noalloc: the wavefront renderer keeps its path pool in memory of its own");
        if (capturing) return fail(RT_ERR_INVALID_ARGUMENT, "wavefront render: the path pool cannot be made on a capturing stream");
        if ((rc = launch_wavefront(scene, S, C, P, d_sum_rgb, st, d_partials)) != RT_OK) return rc;
    } else if ((opts->flags & RT_RENDER_KERNEL_LOCKSTEP) && !d_dump) {
        if ((rc = resident_blocks(scene, true, false, 0, &per_cu)) != RT_OK) return rc;
        hipLaunchKernelGGL(render_kernel, grid_for(per_cu, kBlock), dim3(kBlock), stack_bytes(scene), st, S, C, P, depth_slots,
                           d_sum_rgb, d_partials, d_work_counter);
    } else if (scene->lds_scene && !(opts->flags & RT_RENDER_KERNEL_NO_LDS_SCENE)) {
        if ((rc = resident_blocks(scene, false, true, mode, &per_cu)) != RT_OK) return rc;
        LdsLaunch L;
        if ((rc = lds16_launch(scene, mode, &L)) != RT_OK) return rc;
        hipLaunchKernelGGL(L.kernel, grid_for(per_cu, L.block), dim3(L.block), L.lds, st, S, C, P, scene->d_nodes16, depth_slots,
                           d_sum_rgb, d_partials, d_work_counter, d_dump);
    } else {
        if ((rc = resident_blocks(scene, false, false, mode, &per_cu)) != RT_OK) return rc;
        bool five = false;
        if ((rc = use_five_waves(scene, false, &five)) != RT_OK) return rc;
        hipLaunchKernelGGL(ws_kernel(false, mode, true, five, scene->plain, scene->no_discs), grid_for(per_cu, kBlock), dim3(kBlock), stack_bytes(scene), st, S, C, P,
                           static_cast<const Node16*>(nullptr), depth_slots, d_sum_rgb, d_partials, d_work_counter, d_dump);
    }
    RT_HIP(hipGetLastError());
    if (P.n_chunks > 1) {
        const size_t n = size_t(P.n_pixels) * 3;
        hipLaunchKernelGGL(combine_kernel, dim3(uint32_t((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, d_partials, P.n_chunks, n,
                           uint32_t(opts->flags & RT_RENDER_ACCUMULATE), d_sum_rgb);
        RT_HIP(hipGetLastError());
    }
    if (!capturing) RT_HIP(hipEventRecord(scene->ev_done, st));
    return RT_OK;
}

}  // namespace

int rt_scene_reserve(rt_scene* scene, const rt_camera* cam, const rt_render_opts* opts) {
    int rc = check_render_args(scene, cam, opts, "rt_scene_reserve");
    if (rc != RT_OK) return rc;
    DeviceGuard guard;
    RT_HIP(hipSetDevice(scene->device));
    return reserve_partials(scene, partials_needed(make_params(*cam, *opts, schedule_of(scene))));
}

int rt_render_device(rt_scene* scene, const rt_camera* cam, const rt_render_opts* opts, float* d_sum_rgb, void* stream) {
    int rc = check_render_args(scene, cam, opts, "rt_render_device");
    if (rc != RT_OK) return rc;
    if (!d_sum_rgb) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_device: null output buffer");
    DeviceGuard guard;
    RT_HIP(hipSetDevice(scene->device));
    // grown on first use unless rt_scene_reserve has sized it (the one case in which this call waits and allocates)
    if ((rc = reserve_partials(scene, partials_needed(make_params(*cam, *opts, schedule_of(scene))))) != RT_OK) return rc;
    return launch_render(scene, cam, opts, make_params(*cam, *opts, schedule_of(scene)), d_sum_rgb, static_cast<hipStream_t>(stream), scene->d_work_counter,
                         scene->d_partials, nullptr);
}

namespace {
// Device address of `scratch` on each device, asked once per device with that device current.
constexpr int kMaxDevices = 64;
unsigned char* g_scratch_ptr[kMaxDevices] = {};
std::mutex g_scratch_mutex;

int scratch_of_current_device(int device, unsigned char** out) {
    if (device < 0 || device >= kMaxDevices) return fail(RT_ERR_LIMIT, "rt_render_device_noalloc: device %d beyond %d", device, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(rt::scratch));
        if (e != hipSuccess || !p) return fail(RT_ERR_DEVICE, "rt_render_device_noalloc: hipGetSymbolAddress failed: %s", hipGetErrorString(e));
        g_scratch_ptr[device] = static_cast<unsigned char*>(p);
    }
    *out = g_scratch_ptr[device];
    return RT_OK;
}
}  // namespace

int rt_render_device_noalloc(rt_scene* scene, const rt_camera* cam, const rt_render_opts* opts, float* d_sum_rgb, void* stream) {
    int rc = check_render_args(scene, cam, opts, "rt_render_device_noalloc");
    if (rc != RT_OK) return rc;
    if (!d_sum_rgb) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_device_noalloc: null output buffer");
    if (opts->flags & (RT_RENDER_STATS | RT_RENDER_TIMING))
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_device_noalloc: the diagnostic builds report through the scene's own counter");
    DeviceGuard guard;
    RT_HIP(hipSetDevice(scene->device));
    unsigned char* base = nullptr;
    if ((rc = scratch_of_current_device(scene->device, &base)) != RT_OK) return rc;
    static_assert(rt::kCounterBlockBytes <= 2048, "counter block inside the scratch header");
    constexpr size_t kHeader = 2048, kPartialBytes = size_t(RT_NOALLOC_SCRATCH_BYTES) - kHeader;
    const size_t slot = size_t(cam->horiz_res) * cam->vert_res * 3 * sizeof(float);
    RenderParams P = make_params(*cam, *opts, schedule_of(scene));
    if (partials_needed(P) > kPartialBytes) {
        if (opts->samples_per_item)
            return fail(RT_ERR_LIMIT, "rt_render_device_noalloc: %u chunks of a %u x %u frame need %zu bytes of partial sums, the fixed scratch holds %zu",
                        P.n_chunks, cam->horiz_res, cam->vert_res, partials_needed(P), kPartialBytes);
        // chunk size left to the library: as many chunks as the scratch holds (one chunk needs no partial sums at all)
        P = make_params(*cam, *opts, schedule_of(scene), uint32_t(std::max<size_t>(kPartialBytes / slot, 1)));
    }
    return launch_render(scene, cam, opts, P, d_sum_rgb, static_cast<hipStream_t>(stream), reinterpret_cast<uint32_t*>(base),
                         reinterpret_cast<float*>(base + kHeader), nullptr);
}

int rt_debug_render_paths(rt_scene* scene, const rt_camera* cam, const rt_render_opts* opts, float* rgb) {
    int rc = check_render_args(scene, cam, opts, "rt_debug_render_paths");
    if (rc != RT_OK) return rc;
    if (!rgb) return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_render_paths: null output buffer");
    if (opts->flags & RT_RENDER_ACCUMULATE) return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_render_paths: RT_RENDER_ACCUMULATE has no meaning here");
    DeviceGuard guard;
    RT_HIP(hipSetDevice(scene->device));
    const size_t n_pixels = size_t(cam->horiz_res) * cam->vert_res;
    const size_t n_paths = n_pixels * (opts->sample_end - opts->sample_begin);
    if (n_paths > (size_t(1) << 28)) return fail(RT_ERR_LIMIT, "rt_debug_render_paths: more than 2^28 paths in one call");
    if ((rc = reserve_partials(scene, partials_needed(make_params(*cam, *opts, schedule_of(scene))))) != RT_OK) return rc;
    float *d_sum = nullptr, *d_dump = nullptr;
    RT_HIP(hipMalloc(reinterpret_cast<void**>(&d_sum), n_pixels * 3 * sizeof(float)));
    if (hipMalloc(reinterpret_cast<void**>(&d_dump), n_paths * 3 * sizeof(float)) != hipSuccess) {
        (void)hipFree(d_sum);
        return fail(RT_ERR_DEVICE, "rt_debug_render_paths: hipMalloc failed");
    }
    if (hipMemset(d_dump, 0, n_paths * 3 * sizeof(float)) != hipSuccess) rc = fail(RT_ERR_DEVICE, "rt_debug_render_paths: hipMemset failed");
    if (rc == RT_OK) rc = launch_render(scene, cam, opts, make_params(*cam, *opts, schedule_of(scene)), d_sum, nullptr, scene->d_work_counter, scene->d_partials, d_dump);
    if (rc == RT_OK && hipDeviceSynchronize() != hipSuccess) rc = fail(RT_ERR_DEVICE, "rt_debug_render_paths: kernel execution failed");
    if (rc == RT_OK && hipMemcpy(rgb, d_dump, n_paths * 3 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(RT_ERR_DEVICE, "rt_debug_render_paths: download failed");
    (void)hipFree(d_sum);
    (void)hipFree(d_dump);
    return rc;
}

int rt_render_chunk(const rt_scene* scene, const rt_camera* cam, const rt_render_opts* opts, uint32_t* samples_per_item) {
    if (!scene || !cam || !opts || !samples_per_item) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_chunk: null argument");
    if (opts->sample_end <= opts->sample_begin) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_chunk: empty sample range");
    *samples_per_item = make_params(*cam, *opts, schedule_of(scene)).chunk_spp;
    return RT_OK;
}

int rt_debug_render_stats(rt_scene* scene, uint64_t out[8]) {
    if (!scene || !out) return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_render_stats: null argument");
    DeviceGuard guard;
    RT_HIP(hipSetDevice(scene->device));
    RT_HIP(hipDeviceSynchronize());
    RT_HIP(hipMemcpy(out, scene->d_work_counter + 8, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_debug_render_launch(rt_scene* scene, uint32_t* block_threads, uint32_t* workgroups_per_cu, uint32_t* waves_per_simd) {
    if (!scene || !block_threads || !workgroups_per_cu || !waves_per_simd) return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_render_launch: null argument");
    DeviceGuard guard;
    RT_HIP(hipSetDevice(scene->device));
    const bool lds16 = scene->lds_scene;
    int block = kBlock;
    if (lds16) {
        LdsLaunch L;
        if (int rc = lds16_launch(scene, kModeProduction, &L)) return rc;
        block = L.block;
    }
    int per_cu = 0;
    if (int rc = resident_blocks(scene, false, lds16, kModeProduction, &per_cu)) return rc;
    *block_threads = uint32_t(block);
    *workgroups_per_cu = uint32_t(per_cu);
    *waves_per_simd = uint32_t(per_cu * (block / 64) / 4);
    return RT_OK;
}

int rt_finalize_device(const rt_scene* scene, const float* d_sum_rgb, size_t n_pixels, uint32_t spp_total, uint8_t* d_rgb8,
                       void* stream) {
    if (!scene || !d_sum_rgb || !d_rgb8) return fail(RT_ERR_INVALID_ARGUMENT, "rt_finalize_device: null argument");
    if (spp_total == 0) return fail(RT_ERR_INVALID_ARGUMENT, "rt_finalize_device: spp_total is zero");
    DeviceGuard guard;
    RT_HIP(hipSetDevice(scene->device));
    const size_t n = n_pixels * 3;
    if (n == 0) return RT_OK;
    hipLaunchKernelGGL(finalize_kernel, dim3(uint32_t((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), d_sum_rgb, n, 1.0 / double(spp_total), d_rgb8);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

int rt_render(rt_scene* scene, const rt_camera* cam, const rt_render_opts* opts, float* sum_rgb, uint8_t* rgb8) {
    int rc = check_render_args(scene, cam, opts, "rt_render");
    if (rc != RT_OK) return rc;
    if ((opts->flags & RT_RENDER_ACCUMULATE) && !sum_rgb) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render: RT_RENDER_ACCUMULATE needs sum_rgb");
    DeviceGuard guard;
    RT_HIP(hipSetDevice(scene->device));
    const size_t n_pixels = size_t(cam->horiz_res) * cam->vert_res;
    if (n_pixels > scene->frame_pixels) {  // the handle keeps its frame buffers: a second call of the same size allocates nothing
        RT_HIP(hipEventSynchronize(scene->ev_done));
        if (scene->d_frame_sum) (void)hipFree(scene->d_frame_sum);
        if (scene->d_frame_rgb) (void)hipFree(scene->d_frame_rgb);
        scene->d_frame_sum = nullptr; scene->d_frame_rgb = nullptr; scene->frame_pixels = 0;
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&scene->d_frame_sum), n_pixels * 3 * sizeof(float)));
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&scene->d_frame_rgb), n_pixels * 3));
        scene->frame_pixels = n_pixels;
    }
    float* d_sum = scene->d_frame_sum;
    uint8_t* d_rgb = scene->d_frame_rgb;
    if (opts->flags & RT_RENDER_ACCUMULATE)
        if (hipMemcpy(d_sum, sum_rgb, n_pixels * 3 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return fail(RT_ERR_DEVICE, "rt_render: upload of the running sum failed");
    rc = rt_render_device(scene, cam, opts, d_sum, nullptr);
    if (rc == RT_OK && rgb8) {
        // a running sum handed in with RT_RENDER_ACCUMULATE is taken to hold samples [0, sample_begin)
        const uint32_t in_sum = (opts->flags & RT_RENDER_ACCUMULATE) ? opts->sample_end : opts->sample_end - opts->sample_begin;
        rc = rt_finalize_device(scene, d_sum, n_pixels, in_sum, d_rgb, nullptr);
    }
    if (rc == RT_OK && hipDeviceSynchronize() != hipSuccess) rc = fail(RT_ERR_DEVICE, "rt_render: kernel execution failed: %s", hipGetErrorString(hipGetLastError()));
    if (rc == RT_OK && sum_rgb && hipMemcpy(sum_rgb, d_sum, n_pixels * 3 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(RT_ERR_DEVICE, "rt_render: download failed");
    if (rc == RT_OK && rgb8 && hipMemcpy(rgb8, d_rgb, n_pixels * 3, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(RT_ERR_DEVICE, "rt_render: download failed");
    return rc;
}

namespace {

// RCCL, bound at the first multi-device render: the single-device paths then carry no dependency on it.
// Only the five calls below are used; types follow rccl.h (ncclComm_t is an opaque pointer, ncclFloat32 = 7, ncclSum = 0).
struct Rccl {
    void* so = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*Reduce)(const void* send, void* recv, size_t count, int dtype, int op, int root, void* comm, hipStream_t st) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::vector<int> devices;   // the device list of `comms` (one communicator set is kept: renders reuse it)
    std::vector<void*> comms;
    bool tried = false;

    bool load() {
        if (tried) return so != nullptr;
        tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (so) break;
        }
        if (!so) return false;
        auto sym = [&](const char* n) { return dlsym(so, n); };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        Reduce = reinterpret_cast<decltype(Reduce)>(sym("ncclReduce"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !Reduce || !GroupStart || !GroupEnd || !GetErrorString) { dlclose(so); so = nullptr; }
        return so != nullptr;
    }
    // communicators over exactly these devices, in this order (rank g = devs[g])
    int communicators(const std::vector<int>& devs) {
        if (devs == devices && !comms.empty()) return RT_OK;
        for (void* c : comms) (void)CommDestroy(c);
        comms.assign(devs.size(), nullptr);
        devices.clear();
        const int r = CommInitAll(comms.data(), int(devs.size()), devs.data());
        if (r != 0) { comms.clear(); return fail(RT_ERR_DEVICE, "rt_render_multi: ncclCommInitAll failed: %s", GetErrorString(r)); }
        devices = devs;
        return RT_OK;
    }
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

}  // namespace

namespace {

// What a multi-device render needs besides the scenes: one stream and one frame buffer per handle, the landing slots
// and the 8-bit frame on the first device, one event per handle.  Kept between calls for the same handle list and
// frame size (a second call allocates nothing); replaced when either changes, dropped when one of its scenes is destroyed.
struct MultiCache {
    std::vector<rt_scene*> scenes;
    size_t n_pixels = 0;
    std::vector<float*> d_sum;
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> landed;
    float* d_land = nullptr;   // n_scenes - 1 slots on the first device (peer path)
    uint8_t* d_rgb = nullptr;  // on the first device
    void release() {
        for (size_t g = 0; g < scenes.size(); ++g) {
            if (hipSetDevice(scenes[g]->device) != hipSuccess) continue;
            if (g < streams.size() && streams[g]) { (void)hipStreamSynchronize(streams[g]); (void)hipStreamDestroy(streams[g]); }
            if (g < landed.size() && landed[g]) (void)hipEventDestroy(landed[g]);
            if (g < d_sum.size() && d_sum[g]) (void)hipFree(d_sum[g]);
        }
        if (!scenes.empty() && hipSetDevice(scenes[0]->device) == hipSuccess) {
            if (d_land) (void)hipFree(d_land);
            if (d_rgb) (void)hipFree(d_rgb);
        }
        scenes.clear(); d_sum.clear(); streams.clear(); landed.clear();
        d_land = nullptr; d_rgb = nullptr; n_pixels = 0;
    }
};
MultiCache g_multi;
std::mutex g_multi_mutex;

}  // namespace

// called by rt_scene_destroy: the cache must not outlive a scene it names
static void rt_multi_cache_forget(const rt_scene* scene) {
    std::lock_guard<std::mutex> lock(g_multi_mutex);
    for (const rt_scene* s : g_multi.scenes)
        if (s == scene) { g_multi.release(); return; }
}

int rt_render_multi(rt_scene* const* scenes, size_t n_scenes, const rt_camera* cam, const rt_render_opts* opts, float* sum_rgb,
                    uint8_t* rgb8) {
    if (!scenes || n_scenes == 0) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_multi: no scene handles");
    if (n_scenes > 64) return fail(RT_ERR_LIMIT, "rt_render_multi: more than 64 scene handles");
    for (size_t g = 0; g < n_scenes; ++g) {
        int rc = check_render_args(scenes[g], cam, opts, "rt_render_multi");
        if (rc != RT_OK) return rc;
        for (size_t k = 0; k < g; ++k)
            if (scenes[k] == scenes[g]) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_multi: handle %zu is handle %zu again", g, k);
    }
    if (opts->flags & RT_RENDER_ACCUMULATE) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_multi: RT_RENDER_ACCUMULATE is not supported here");
    const uint32_t spp = opts->sample_end - opts->sample_begin;
    if (spp < n_scenes) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_multi: %u samples for %zu devices", spp, n_scenes);
    const size_t n_pixels = size_t(cam->horiz_res) * cam->vert_res, n_channels = n_pixels * 3, bytes = n_channels * sizeof(float);
    DeviceGuard guard;
    const int dev0 = scenes[0]->device;
    std::vector<int> devs(n_scenes);
    bool distinct = true;
    for (size_t g = 0; g < n_scenes; ++g) {
        devs[g] = scenes[g]->device;
        for (size_t k = 0; k < g; ++k) distinct = distinct && devs[k] != devs[g];
    }
    // How the partial images meet on the first device.  "rccl": one ncclReduce(sum) to rank 0 over xGMI (needs one
    // handle per DIFFERENT device; the library picks the order of the additions).  "peer": every device copies its
    // image into a landing slot of its own on the first device, all copies in flight at once, and one kernel adds the
    // slots in handle order.  Default: rccl where it applies, loads and sets up its communicators, else peer.
    const char* how = std::getenv("RT_MULTI_REDUCE");
    const bool force_rccl = how && std::strcmp(how, "rccl") == 0, force_peer = how && std::strcmp(how, "peer") == 0;
    if (how && !force_rccl && !force_peer) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_multi: RT_MULTI_REDUCE is '%s' (rccl or peer)", how);
    if (force_rccl && !distinct) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_multi: RT_MULTI_REDUCE=rccl needs every handle on a different device");
    bool use_rccl = distinct && !force_peer && (n_scenes > 1 || force_rccl);
    std::unique_lock<std::mutex> rccl_lock(g_rccl_mutex, std::defer_lock);
    if (use_rccl) {
        rccl_lock.lock();
        if (!g_rccl.load()) {
            if (force_rccl) return fail(RT_ERR_DEVICE, "rt_render_multi: librccl.so could not be loaded: %s", dlerror());
            use_rccl = false;
        } else if (int crc = g_rccl.communicators(devs)) {
            // a machine on which RCCL cannot set up these communicators still renders: the peer copies need nothing
            // but HIP.  Only a caller that asked for RCCL by name gets the error (rt_last_error has RCCL's own text).
            if (force_rccl) return crc;
            use_rccl = false;
        }
        if (!use_rccl) rccl_lock.unlock();
    }

    std::lock_guard<std::mutex> cache_lock(g_multi_mutex);
    int rc = RT_OK;
    auto chk = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == RT_OK) rc = fail(RT_ERR_DEVICE, "rt_render_multi: %s failed: %s", what, hipGetErrorString(e));
    };
    MultiCache& mc = g_multi;
    const std::vector<rt_scene*> key(scenes, scenes + n_scenes);
    if (mc.scenes != key || mc.n_pixels != n_pixels) {
        mc.release();
        mc.scenes = key;
        mc.n_pixels = n_pixels;
        mc.d_sum.assign(n_scenes, nullptr);
        mc.streams.assign(n_scenes, nullptr);
        mc.landed.assign(n_scenes, nullptr);
        for (size_t g = 0; g < n_scenes && rc == RT_OK; ++g) {
            chk(hipSetDevice(devs[g]), "hipSetDevice");
            chk(hipStreamCreateWithFlags(&mc.streams[g], hipStreamNonBlocking), "hipStreamCreate");
            chk(hipMalloc(reinterpret_cast<void**>(&mc.d_sum[g]), bytes), "hipMalloc");
            chk(hipEventCreateWithFlags(&mc.landed[g], hipEventDisableTiming), "hipEventCreate");
            if (rc == RT_OK && devs[g] != dev0) {  // direct xGMI access where the pair allows it; the copies work either way
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, devs[g], dev0) == hipSuccess && can) {
                    const hipError_t e = hipDeviceEnablePeerAccess(dev0, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) chk(e, "hipDeviceEnablePeerAccess");
                    (void)hipGetLastError();
                }
            }
        }
        if (rc == RT_OK) {
            chk(hipSetDevice(dev0), "hipSetDevice");
            if (n_scenes > 1) chk(hipMalloc(reinterpret_cast<void**>(&mc.d_land), bytes * (n_scenes - 1)), "hipMalloc");
            chk(hipMalloc(reinterpret_cast<void**>(&mc.d_rgb), n_pixels * 3), "hipMalloc");
        }
        // the partial sums of chunked renders, so that the render calls below find them in place
        for (size_t g = 0; g < n_scenes && rc == RT_OK; ++g) {
            rt_render_opts o = *opts;
            o.sample_begin = opts->sample_begin + uint32_t(uint64_t(spp) * g / n_scenes);
            o.sample_end = opts->sample_begin + uint32_t(uint64_t(spp) * (g + 1) / n_scenes);
            rc = rt_scene_reserve(scenes[g], cam, &o);
        }
        if (rc != RT_OK) { mc.release(); return rc; }
    }
    // every device traces its share of the sample indices at the same time, each on a stream of its own; on the peer
    // path its copy to the first device follows on the same stream
    for (size_t g = 0; g < n_scenes && rc == RT_OK; ++g) {
        rt_render_opts o = *opts;
        o.sample_begin = opts->sample_begin + uint32_t(uint64_t(spp) * g / n_scenes);
        o.sample_end = opts->sample_begin + uint32_t(uint64_t(spp) * (g + 1) / n_scenes);
        rc = rt_render_device(scenes[g], cam, &o, mc.d_sum[g], mc.streams[g]);
        if (rc != RT_OK || use_rccl || g == 0) continue;
        chk(hipSetDevice(devs[g]), "hipSetDevice");
        chk(hipMemcpyPeerAsync(mc.d_land + (g - 1) * n_channels, dev0, mc.d_sum[g], devs[g], bytes, mc.streams[g]), "hipMemcpyPeerAsync");
        chk(hipEventRecord(mc.landed[g], mc.streams[g]), "hipEventRecord");
    }
    // the one exchange
    if (rc == RT_OK && use_rccl) {
        int r = g_rccl.GroupStart();
        for (size_t g = 0; g < n_scenes && r == 0; ++g) {
            chk(hipSetDevice(devs[g]), "hipSetDevice");
            r = g_rccl.Reduce(mc.d_sum[g], mc.d_sum[g], n_channels, /*ncclFloat32*/ 7, /*ncclSum*/ 0, 0, g_rccl.comms[g], mc.streams[g]);
        }
        const int r_end = g_rccl.GroupEnd();
        if (r == 0) r = r_end;
        if (r != 0 && rc == RT_OK) rc = fail(RT_ERR_DEVICE, "rt_render_multi: ncclReduce failed: %s", g_rccl.GetErrorString(r));
    } else if (rc == RT_OK && n_scenes > 1) {
        chk(hipSetDevice(dev0), "hipSetDevice");
        for (size_t g = 1; g < n_scenes; ++g) chk(hipStreamWaitEvent(mc.streams[0], mc.landed[g], 0), "hipStreamWaitEvent");
        if (rc == RT_OK) {
            hipLaunchKernelGGL(combine_kernel, dim3(uint32_t((n_channels + kBlock - 1) / kBlock)), dim3(kBlock), 0, mc.streams[0], mc.d_land,
                               uint32_t(n_scenes - 1), n_channels, 1u, mc.d_sum[0]);
            chk(hipGetLastError(), "combine_kernel");
        }
    }
    if (rc == RT_OK) chk(hipSetDevice(dev0), "hipSetDevice");
    if (rc == RT_OK && rgb8) rc = rt_finalize_device(scenes[0], mc.d_sum[0], n_pixels, spp, mc.d_rgb, mc.streams[0]);
    // the first device's stream ends with everything the result depends on; the other streams have nothing after
    // their copy (peer) or their part of the reduce (RCCL), and the next call's renders wait for each scene's own event
    if (rc == RT_OK) chk(hipStreamSynchronize(mc.streams[0]), "kernel execution");
    if (rc == RT_OK && sum_rgb) chk(hipMemcpy(sum_rgb, mc.d_sum[0], bytes, hipMemcpyDeviceToHost), "download");
    if (rc == RT_OK && rgb8) chk(hipMemcpy(rgb8, mc.d_rgb, n_pixels * 3, hipMemcpyDeviceToHost), "download");
    if (rc != RT_OK) {  // leave nothing in flight behind an error, and nothing cached that may be half set up
        for (size_t g = 0; g < n_scenes; ++g)
            if (hipSetDevice(devs[g]) == hipSuccess && mc.streams[g]) (void)hipStreamSynchronize(mc.streams[g]);
        mc.release();
    }
    return rc;
}

int rt_trace_primary(rt_scene* scene, const float* orig, const float* dir, size_t n, int32_t* id, double* t, double* normal) {
    if (!scene || !orig || !dir || !id || !t || !normal) return fail(RT_ERR_INVALID_ARGUMENT, "rt_trace_primary: null argument");
    if (n == 0) return RT_OK;
    if (n > 0x7fffffffull - kBlock) return fail(RT_ERR_LIMIT, "rt_trace_primary: more than 2^31 rays in one call");
    DeviceGuard guard;
    RT_HIP(hipSetDevice(scene->device));
    double origin_abs = 0.0;
    for (size_t i = 0; i < 3 * n; ++i) {
        if (!std::isfinite(orig[i]) || !std::isfinite(dir[i])) return fail(RT_ERR_INVALID_ARGUMENT, "rt_trace_primary: non-finite ray component at %zu", i);
        origin_abs = std::max(origin_abs, double(std::fabs(orig[i])));
    }
    float *d_o = nullptr, *d_d = nullptr;
    int32_t* d_id = nullptr;
    double *d_t = nullptr, *d_n = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_o); (void)hipFree(d_d); (void)hipFree(d_id); (void)hipFree(d_t); (void)hipFree(d_n); };
    int rc = RT_OK;
    auto chk = [&](hipError_t e, const char* what) { if (e != hipSuccess && rc == RT_OK) rc = fail(RT_ERR_DEVICE, "rt_trace_primary: %s failed: %s", what, hipGetErrorString(e)); };
    chk(hipMalloc(reinterpret_cast<void**>(&d_o), 3 * n * sizeof(float)), "hipMalloc");
    chk(hipMalloc(reinterpret_cast<void**>(&d_d), 3 * n * sizeof(float)), "hipMalloc");
    chk(hipMalloc(reinterpret_cast<void**>(&d_id), n * sizeof(int32_t)), "hipMalloc");
    chk(hipMalloc(reinterpret_cast<void**>(&d_t), n * sizeof(double)), "hipMalloc");
    chk(hipMalloc(reinterpret_cast<void**>(&d_n), 3 * n * sizeof(double)), "hipMalloc");
    if (rc == RT_OK) {
        chk(hipMemcpy(d_o, orig, 3 * n * sizeof(float), hipMemcpyHostToDevice), "upload");
        chk(hipMemcpy(d_d, dir, 3 * n * sizeof(float), hipMemcpyHostToDevice), "upload");
        chk(hipMemset(d_t, 0, n * sizeof(double)), "memset");
        chk(hipMemset(d_n, 0, 3 * n * sizeof(double)), "memset");
    }
    if (rc == RT_OK) {
        SceneDev S = scene->dev;
        S.eps = descent_eps(scene, origin_abs);
        hipLaunchKernelGGL(trace_primary_kernel, dim3(uint32_t((n + kBlock - 1) / kBlock)), dim3(kBlock), stack_bytes(scene), nullptr,
                           S, d_o, d_d, uint32_t(n), d_id, d_t, d_n);
        chk(hipGetLastError(), "launch");
        chk(hipDeviceSynchronize(), "kernel");
    }
    if (rc == RT_OK) {
        chk(hipMemcpy(id, d_id, n * sizeof(int32_t), hipMemcpyDeviceToHost), "download");
        std::vector<double> ht(n), hn(3 * n);
        chk(hipMemcpy(ht.data(), d_t, n * sizeof(double), hipMemcpyDeviceToHost), "download");
        chk(hipMemcpy(hn.data(), d_n, 3 * n * sizeof(double), hipMemcpyDeviceToHost), "download");
        if (rc == RT_OK)
            for (size_t k = 0; k < n; ++k)
                if (id[k] >= 0) { t[k] = ht[k]; for (int a = 0; a < 3; ++a) normal[3 * k + a] = hn[3 * k + a]; }
    }
    cleanup();
    return rc;
}

int rt_generate_primary(rt_scene* scene, const rt_camera* cam, uint64_t seed, uint32_t pixel_begin, uint32_t pixel_end,
                        uint32_t sample_begin, uint32_t sample_end, float* orig, float* dir) {
    if (!scene || !cam || !orig || !dir) return fail(RT_ERR_INVALID_ARGUMENT, "rt_generate_primary: null argument");
    if (cam->horiz_res == 0 || cam->vert_res == 0) return fail(RT_ERR_INVALID_ARGUMENT, "rt_generate_primary: zero resolution");
    if (pixel_end < pixel_begin || uint64_t(pixel_end) > uint64_t(cam->horiz_res) * cam->vert_res || sample_end <= sample_begin)
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_generate_primary: bad pixel or sample range");
    const uint64_t ns = sample_end - sample_begin, n = uint64_t(pixel_end - pixel_begin) * ns;
    if (n == 0) return RT_OK;
    if (n > 0x7fffffffull - kBlock) return fail(RT_ERR_LIMIT, "rt_generate_primary: more than 2^31 rays in one call");
    DeviceGuard guard;
    RT_HIP(hipSetDevice(scene->device));
    float *d_o = nullptr, *d_d = nullptr;
    RT_HIP(hipMalloc(reinterpret_cast<void**>(&d_o), 3 * n * sizeof(float)));
    if (hipMalloc(reinterpret_cast<void**>(&d_d), 3 * n * sizeof(float)) != hipSuccess) { (void)hipFree(d_o); return fail(RT_ERR_DEVICE, "rt_generate_primary: hipMalloc failed"); }
    rt_render_opts o{};
    o.sample_begin = sample_begin; o.sample_end = sample_end; o.depth = 1; o.flags = 0; o.seed = seed;
    const RenderParams P = make_params(*cam, o, kScheduleLds);
    hipLaunchKernelGGL(generate_primary_kernel, dim3(uint32_t((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, nullptr, make_cam(*cam), P,
                       pixel_begin, uint32_t(ns), uint32_t(n), d_o, d_d);
    int rc = RT_OK;
    if (hipDeviceSynchronize() != hipSuccess) rc = fail(RT_ERR_DEVICE, "rt_generate_primary: kernel failed");
    if (rc == RT_OK && (hipMemcpy(orig, d_o, 3 * n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess ||
                        hipMemcpy(dir, d_d, 3 * n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess))
        rc = fail(RT_ERR_DEVICE, "rt_generate_primary: download failed");
    (void)hipFree(d_o);
    (void)hipFree(d_d);
    return rc;
}

int rt_trace_paths(rt_scene* scene, const rt_camera* cam, const rt_render_opts* opts, const uint32_t* pixel,
                   const uint32_t* sample, size_t n, float* rgb) {
    return rt_trace_path_segments(scene, cam, opts, pixel, sample, n, rgb, 0, nullptr, nullptr, nullptr);
}

int rt_trace_path_segments(rt_scene* scene, const rt_camera* cam, const rt_render_opts* opts, const uint32_t* pixel,
                           const uint32_t* sample, size_t n, float* rgb, uint32_t max_segments, int32_t* seg_id,
                           double* seg_t, uint32_t* n_seg) {
    if (!scene || !cam || !opts || !pixel || !sample) return fail(RT_ERR_INVALID_ARGUMENT, "rt_trace_paths: null argument");
    if (!rgb && !n_seg) return fail(RT_ERR_INVALID_ARGUMENT, "rt_trace_paths: no output requested");
    if (n_seg && max_segments && (!seg_id || !seg_t)) return fail(RT_ERR_INVALID_ARGUMENT, "rt_trace_path_segments: null segment buffers");
    if (uint64_t(n) * max_segments > (uint64_t(1) << 31)) return fail(RT_ERR_LIMIT, "rt_trace_path_segments: more than 2^31 segment records");
    if (cam->horiz_res == 0 || cam->vert_res == 0) return fail(RT_ERR_INVALID_ARGUMENT, "rt_trace_paths: zero resolution");
    if (n == 0) return RT_OK;
    if (n > 0x7fffffffull - kBlock) return fail(RT_ERR_LIMIT, "rt_trace_paths: more than 2^31 paths in one call");
    const uint64_t n_pixels = uint64_t(cam->horiz_res) * cam->vert_res;
    for (size_t k = 0; k < n; ++k)
        if (pixel[k] >= n_pixels) return fail(RT_ERR_INVALID_ARGUMENT, "rt_trace_paths: pixel %u of %llu", pixel[k], (unsigned long long)n_pixels);
    DeviceGuard guard;
    RT_HIP(hipSetDevice(scene->device));
    uint32_t *d_p = nullptr, *d_s = nullptr, *d_nseg = nullptr;
    float* d_rgb = nullptr;
    int32_t* d_sid = nullptr;
    double* d_st = nullptr;
    const size_t n_rec = std::max<size_t>(size_t(n) * max_segments, 1);
    int rc = RT_OK;
    auto chk = [&](hipError_t e, const char* what) { if (e != hipSuccess && rc == RT_OK) rc = fail(RT_ERR_DEVICE, "rt_trace_paths: %s failed: %s", what, hipGetErrorString(e)); };
    chk(hipMalloc(reinterpret_cast<void**>(&d_p), n * sizeof(uint32_t)), "hipMalloc");
    chk(hipMalloc(reinterpret_cast<void**>(&d_s), n * sizeof(uint32_t)), "hipMalloc");
    chk(hipMalloc(reinterpret_cast<void**>(&d_rgb), 3 * n * sizeof(float)), "hipMalloc");
    if (n_seg) {
        chk(hipMalloc(reinterpret_cast<void**>(&d_nseg), n * sizeof(uint32_t)), "hipMalloc");
        chk(hipMalloc(reinterpret_cast<void**>(&d_sid), n_rec * sizeof(int32_t)), "hipMalloc");
        chk(hipMalloc(reinterpret_cast<void**>(&d_st), n_rec * sizeof(double)), "hipMalloc");
        if (rc == RT_OK) { chk(hipMemset(d_sid, 0xff, n_rec * sizeof(int32_t)), "memset"); chk(hipMemset(d_st, 0, n_rec * sizeof(double)), "memset"); }
    }
    if (rc == RT_OK) {
        chk(hipMemcpy(d_p, pixel, n * sizeof(uint32_t), hipMemcpyHostToDevice), "upload");
        chk(hipMemcpy(d_s, sample, n * sizeof(uint32_t), hipMemcpyHostToDevice), "upload");
    }
    if (rc == RT_OK) {
        SceneDev S = scene->dev;
        S.eps = descent_eps(scene, camera_abs(*cam));
        rt_render_opts o = *opts;
        if (o.sample_end <= o.sample_begin) { o.sample_begin = 0; o.sample_end = 1; }
        const dim3 grid(uint32_t((n + kBlock - 1) / kBlock));
        if (rgb) {
            hipLaunchKernelGGL(trace_paths_kernel, grid, dim3(kBlock), stack_bytes(scene), nullptr, S, make_cam(*cam),
                               make_params(*cam, o, kScheduleLds), d_p, d_s, uint32_t(n), d_rgb);
            chk(hipGetLastError(), "launch");
        }
        if (n_seg) {
            hipLaunchKernelGGL(path_segments_kernel, grid, dim3(kBlock), stack_bytes(scene), nullptr, S, make_cam(*cam),
                               make_params(*cam, o, kScheduleLds), d_p, d_s, uint32_t(n), max_segments, d_sid, d_st, d_nseg);
            chk(hipGetLastError(), "launch");
        }
        chk(hipDeviceSynchronize(), "kernel");
        if (rgb) chk(hipMemcpy(rgb, d_rgb, 3 * n * sizeof(float), hipMemcpyDeviceToHost), "download");
        if (n_seg) {
            chk(hipMemcpy(n_seg, d_nseg, n * sizeof(uint32_t), hipMemcpyDeviceToHost), "download");
            if (max_segments) {
                chk(hipMemcpy(seg_id, d_sid, size_t(n) * max_segments * sizeof(int32_t), hipMemcpyDeviceToHost), "download");
                chk(hipMemcpy(seg_t, d_st, size_t(n) * max_segments * sizeof(double), hipMemcpyDeviceToHost), "download");
            }
        }
    }
    (void)hipFree(d_p);
    (void)hipFree(d_s);
    (void)hipFree(d_rgb);
    (void)hipFree(d_nseg);
    (void)hipFree(d_sid);
    (void)hipFree(d_st);
    return rc;
}

}  // extern "C"
