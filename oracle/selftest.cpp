// selftest.cpp — the reference's inline known-answer tests restated against the
// oracle, plus hand-derived checks for the rules no reference test covers.
// TEST INFRASTRUCTURE ONLY.  Exit status 0 = all passed.
#include "rt_oracle.hpp"

#include <cstdio>

using namespace orc;

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

static BoundingBox box(double a0, double a1, double b0, double b1, double c0, double c1) {
    return BoundingBox{{{a0, a1}, {b0, b1}, {c0, c1}}, -1};
}
static bool same_dims(const BoundingBox& a, const BoundingBox& b) {
    for (int i = 0; i < 3; ++i)
        if (a.dims[i].start != b.dims[i].start || a.dims[i].end != b.dims[i].end) return false;
    return true;
}
static bool veq(const Vec3& a, const Vec3& b) { return a[0] == b[0] && a[1] == b[1] && a[2] == b[2]; }

int main() {
    // vector.rs:129-147
    CHECK(norm(V(3, 4, 0)) == 5.0);
    CHECK(dot(V(1, 0, 0), V(0.5, 0, 0)) == 0.5);
    CHECK(veq(cross(V(1, 0, 0), V(0, 1, 0)), V(0, 0, 1)));
    // ray.rs:21-25
    CHECK(veq(Ray::make(V(0, 0, 0), V(1, 0, 0)).position_at(0.5), V(0.5, 0, 0)));
    // geometry.rs:150-154
    CHECK(veq(sphere_normal_at(V(0, 0, 0), 2.0, V(2, 0, 0)), V(1, 0, 0)));
    // geometry.rs:156-160
    {
        Hittable d; std::memset(&d, 0, sizeof d); d.shape = SHAPE_DISC;
        Vec3 n = normalize(V(0, 0, 1));
        for (int i = 0; i < 3; ++i) d.normal[i] = n[i];
        CHECK(veq(shape_normal_at(d, V(0, 0.5, 0)), V(0, 0, 1)));
    }
    // geometry.rs:162-167
    { double t = -1; CHECK(sphere_intersect(V(0, 0, 0), 2.0, Ray::make(V(0, 0, -3), V(0, 0, 1)), &t)); CHECK(t == 1.0); }
    // geometry.rs:169-174
    { double t; CHECK(!sphere_intersect(V(0, 0, 0), 1.0, Ray::make(V(2, 0, 0), V(1, 0, 0)), &t)); }
    // geometry.rs:176-181
    {
        Ray r = Ray::make(V(1, 0, 3), V(0, 0, -1));
        double t = 0;
        CHECK(disc_intersect(V(0, 0, 0), normalize(V(0, 0, 1)), 2.0, r, &t));
        CHECK(veq(r.position_at(t), V(1, 0, 0)));
    }
    // geometry.rs:183-197 (legacy BoundBox cover = per-axis interval cover)
    {
        BoundingBox c = make_cover_of(box(0, 1, 0, 1, 0, 1), box(-1, 1, 0, 2, -1, 0));
        CHECK(same_dims(c, box(-1, 1, 0, 2, -1, 1)));
    }
    // geometry.rs:199-211 (same formula as make_covering)
    {
        BoundingBox b = make_covering(make_sphere(V(1, 2, 3), 2.0), 0);
        CHECK(same_dims(b, box(-1, 3, 0, 4, 1, 5)));
    }
    // intervals.rs:77-80
    CHECK((Interval{1.0, 2.0}.size()) == 1.0);
    // intervals.rs:82-94
    {
        Interval in1{0, 1}, in2{1, 2}, in3{1, 0}, out;
        CHECK(!intersection(in1, in2, &out));
        CHECK(!intersection(in1, in3, &out));
        CHECK(!intersection(in2, in3, &out));
    }
    // intervals.rs:96-103
    { Interval c = cover(Interval{-1, 1}, Interval{2, 3}); CHECK(c.start == -1.0 && c.end == 3.0); }
    // intervals.rs:105-112
    { Interval a{-1, 1}, b{2, 3}; const Interval& l = get_larger(a, b); CHECK(l.start == -1.0 && l.end == 1.0); }
    // intervals.rs:114-120 (macro == constructor): nothing to restate beyond construction
    { Interval a{0.0, 1.0}, b{0.0, 1.0}; CHECK(a.start == b.start && a.end == b.end); }
    // boundingvolume.rs:272-291: LINE semantics, the 1e-4 divisor, and a miss
    {
        BoundingBox b = box(0, 1, 0, 1, 0, 1);
        CHECK(bbox_check_intersection(b, Ray{V(1.5, 0.5, 0.5), V(1, 0, 0)}));
        CHECK(bbox_check_intersection(b, Ray{V(1.5, 0.5, 0.5), V(-1, 0, 0)}));
        CHECK(!bbox_check_intersection(b, Ray{V(1.5, 1.5, 0.5), V(1, 0, 0)}));
    }
    // boundingvolume.rs:293-305
    CHECK(same_dims(make_cover_of(box(0, 1, 0, 1, 0, 1), box(1, 2, 1, 2, 1, 2)), box(0, 2, 0, 2, 0, 2)));
    // boundingvolume.rs:307-327
    CHECK(bbox_longest_axis(box(0, 1, 0, 1, 0, 1)) == 2);
    CHECK(bbox_longest_axis(box(0, 1, 0, 2, 0, 1)) == 1);
    // boundingvolume.rs:329-337
    { BoundingBox b = box(0, 1, 0, 2, 0, 3); CHECK(b.dims[0].midpoint() == 0.5 && b.dims[1].midpoint() == 1.0 && b.dims[2].midpoint() == 1.5); }
    // boundingvolume.rs:339-362
    {
        BoundingBox b1 = box(0, 1, 0, 2, 0, 3), b2 = box(-1, 1, -1, 1, -1, 1), b3 = box(-2, 5, -2, 2, -1, 3);
        BoundingBox list[3] = {b1, b2, b3};
        sort_on_index(list, 3, 0);
        CHECK(same_dims(list[0], b1));
        sort_on_index(list, 3, 2);
        CHECK(same_dims(list[1], b1));
    }
    // boundingvolume.rs:364-371
    CHECK(same_dims(make_cover_of(bbox_empty(), box(-1, 1, -1, 1, -1, 1)), box(-1, 1, -1, 1, -1, 1)));
    // boundingvolume.rs:373-389
    {
        BoundingBox b1 = box(0, 1, 0, 1, 0, 1);
        BoundingBox list[3] = {b1, box(-1, 1, -1, 1, -1, 1), box(0, 2, 0, 1, 1, 1)};
        BoundingBox total = make_all_covering(list, 3);
        CHECK(bbox_longest_axis(total) == 0);
        sort_on_index(list, 3, bbox_longest_axis(total));
        CHECK(same_dims(list[0], b1));
    }
    // boundingvolume.rs:391-420
    {
        BoundingBox b1 = box(0, 1, 0, 2, -1, 2), b2 = box(-2, 0, -3, 0, -2, 0), b3 = box(-2, 1, 0, 1, 3, 4);
        BoundingBox list[3] = {b1, b2, b3};
        BoundingBox total = make_all_covering(list, 3);
        sort_on_index(list, 3, bbox_longest_axis(total));
        CHECK(same_dims(list[0], b3));
        // split_on_covering: halfway = 1, so the right half is list[1..3]
        BoundingBox* right = list + 1;
        CHECK(same_dims(right[0], b2));
        BoundingBox rc = make_all_covering(right, 2);
        CHECK(bbox_longest_axis(rc) == 1);
        sort_on_index(right, 2, bbox_longest_axis(rc));
        CHECK(same_dims(right[0], b1));
    }
    // boundingvolume.rs:422-449
    {
        BoundingBox b1 = box(0, 1, 0, 2, -1, 2), b2 = box(-2, 0, -3, 0, -2, 0), b3 = box(-2, 1, 0, 1, 3, 4);
        BoundingBox list[3] = {b1, b2, b3};
        std::unique_ptr<CoveringTree> t = make_coveringtree(list, 3);
        CHECK(t->right != nullptr);
        CHECK(same_dims(t->right->cover, make_cover_of(b1, b2)));
    }
    // boundingvolume.rs:451-502
    {
        Scene sc;
        Hittable h1 = make_sphere(V(0, 0, 0), 5.0), h2 = make_sphere(V(0, 0, 2), 1.0), h3 = make_sphere(V(0, 2, 0), 1.0);
        for (Hittable* h : {&h1, &h2, &h3}) { h->material = MAT_DIFFUSE; h->albedo[0] = h->albedo[1] = h->albedo[2] = 1.0; }
        sc.hittables = {h1, h2, h3};
        CHECK(build_tree(sc));
        CHECK(sc.tree_nodes == 5);
        Ray ray{V(-1.5, -0.5, -0.5), V(1, 0, 0)};
        std::vector<Candidate> sub;
        tree_filter(sc.tree.get(), sc.hittables.data(), sub, ray, nullptr);
        int idx; double t;
        if (nearest(sub, false, &idx, &t)) CHECK(std::isfinite(t));
        CHECK(bbox_check_intersection(sc.tree->cover, ray));
        CHECK(!sub.empty());
        // the origin is inside the big sphere: the hit is its far root
        CHECK(nearest(sub, false, &idx, &t) && idx == 0 && t > 6.0 && t < 6.5);
        // the topology-free statement returns the same answer
        std::vector<Candidate> gated;
        candidates(sc, TRAVERSE_GATED_BRUTE, ray, gated, nullptr);
        int idx2; double t2;
        CHECK(nearest(gated, true, &idx2, &t2) && idx2 == idx && t2 == t);
    }
    // boundingvolume.rs:504-513 (scenegen.rs:111-127): one-sphere tree, ray pointing away
    {
        Scene sc;
        Hittable h = make_sphere(V(0, 0, 0), 5.0);
        h.material = MAT_DIFFUSE; h.albedo[0] = 0.4; h.albedo[1] = 0.2; h.albedo[2] = 0.1;
        sc.hittables = {h};
        CHECK(build_tree(sc));
        CHECK(sc.tree_nodes == 1);
        std::vector<Candidate> sub;
        tree_filter(sc.tree.get(), sc.hittables.data(), sub, Ray{V(10, 0, 0), V(1, 0, 0)}, nullptr);
        int idx; double t;
        CHECK(!nearest(sub, false, &idx, &t));
    }
    // camera.rs:95-101
    {
        for (int mode = 0; mode < 2; ++mode)
            for (uint32_t k = 0; k < 1000; ++k) {
                Sampler s(mode, 7, k, 0);
                double u4[4]; s.direct4(0, 0, u4);
                double p[2];
                random_in_disc(s, u4, p);
                CHECK(p[0] * p[0] + p[1] * p[1] <= 1.0 + 1e-12);
                CHECK(p[0] >= 0.0 && p[1] >= 0.0);  // positive quadrant only (camera.rs:81-89)
            }
    }

    // ---- hand-derived checks for rules that no reference test covers ----
    // lib.rs:186-192
    CHECK(channel_to_ppm(1.0) == 255);
    CHECK(channel_to_ppm(0.25) == 127);
    CHECK(channel_to_ppm(0.0) == 0);
    CHECK(channel_to_ppm(4.0) == 255);
    CHECK(channel_to_ppm(-1.0) == 0);  // sqrt(-1) = NaN -> 0
    CHECK(channel_to_ppm(std::numeric_limits<double>::quiet_NaN()) == 0);
    // materials.rs:133-139: normal incidence on n = 1.5 gives r0 = 0.04
    CHECK(schlick(-1.0, 1.5, 0.0399));
    CHECK(!schlick(-1.0, 1.5, 0.0401));
    CHECK(schlick(0.0, 1.5, 0.999));  // grazing: reflectance 1
    // Philox4x32-10 known answers (Random123 kat_vectors)
    {
        uint32_t c0[4] = {0, 0, 0, 0}, k0[2] = {0, 0}, o[4];
        philox4x32_10(c0, k0, o);
        CHECK(o[0] == 0x6627e8d5u && o[1] == 0xe169c58du && o[2] == 0xbc57ac4cu && o[3] == 0x9b00dbd8u);
        uint32_t c1[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, k1[2] = {0xffffffffu, 0xffffffffu};
        philox4x32_10(c1, k1, o);
        CHECK(o[0] == 0x408f276du && o[1] == 0x41c83b0eu && o[2] == 0xa20bc7c6u && o[3] == 0x6d5451fdu);
        uint32_t c2[4] = {0x243f6a88u, 0x85a308d3u, 0x13198a2eu, 0x03707344u}, k2[2] = {0xa4093822u, 0x299f31d0u};
        philox4x32_10(c2, k2, o);
        CHECK(o[0] == 0xd16cfe09u && o[1] == 0x94fdccebu && o[2] == 0x5001e420u && o[3] == 0x24126ea1u);
    }
    // integrator rules on tiny scenes
    {
        Scene sc;
        std::vector<Candidate> scratch;
        // (a) miss: sky gradient times the initial throughput (lib.rs:117-130)
        Sampler s0(SAMPLER_DIRECT, 1, 0, 0);
        Color c = trace(sc, TRAVERSE_LINEAR, Ray{V(0, 0, 0), V(0, 1, 0)}, 10, s0, scratch, nullptr);
        CHECK(c.r == 0.5 && c.g == 0.7 && c.b == 1.0);
        c = trace(sc, TRAVERSE_LINEAR, Ray{V(0, 0, 0), V(0, -1, 0)}, 10, s0, scratch, nullptr);
        CHECK(c.r == 1.0 && c.g == 1.0 && c.b == 1.0);
        // (b) depth 0 returns the initial throughput (lib.rs:134)
        c = trace(sc, TRAVERSE_LINEAR, Ray{V(0, 0, 0), V(0, 1, 0)}, 0, s0, scratch, nullptr);
        CHECK(c.r == 1.0 && c.g == 1.0 && c.b == 1.0);
        // (c) emitter: albedo * |dir.n| and the throughput is dropped (lib.rs:101-104)
        Hittable mirror = make_sphere(V(0, 0, 5), 1.0);
        mirror.material = MAT_METAL; mirror.albedo[0] = 0.5; mirror.albedo[1] = 0.25; mirror.albedo[2] = 0.125; mirror.fuzz = 0.0;
        Hittable lamp = make_sphere(V(0, 0, -5), 1.0);
        lamp.material = MAT_EMITTER; lamp.albedo[0] = 1.0; lamp.albedo[1] = 0.8; lamp.albedo[2] = 0.6;
        sc.hittables = {mirror, lamp};
        Sampler s1(SAMPLER_DIRECT, 1, 0, 0);
        c = trace(sc, TRAVERSE_LINEAR, Ray{V(0, 0, 0), V(0, 0, 1)}, 10, s1, scratch, nullptr);
        CHECK(c.r == 1.0 && c.g == 0.8 && c.b == 0.6);  // not multiplied by the mirror's albedo
        // (d) depth exhaustion inside a closed diffuse shell returns the product of albedos
        Hittable shell = make_sphere(V(0, 0, 0), 10.0);
        shell.material = MAT_DIFFUSE; shell.albedo[0] = 0.5; shell.albedo[1] = 0.5; shell.albedo[2] = 0.5;
        sc.hittables = {shell};
        Sampler s2(SAMPLER_DIRECT, 1, 0, 0);
        c = trace(sc, TRAVERSE_LINEAR, Ray{V(0, 0, 0), V(0, 0, 1)}, 1, s2, scratch, nullptr);
        CHECK(c.r == 0.5 && c.g == 0.5 && c.b == 0.5);
        // (e) a disc behind the origin is accepted with a negative parameter (geometry.rs:133-139)
        double t = 0;
        CHECK(disc_intersect(V(0, 0, -1), V(0, 0, 1), 1.0, Ray{V(0, 0, 0), V(0, 0, 1)}, &t) && t == -1.0);
        // (f) first minimum wins / None sorts last (lib.rs:31-38)
        std::vector<Candidate> cs = {{0, false, 0}, {1, true, 2.0}, {2, true, 2.0}, {3, true, 3.0}};
        int idx; CHECK(nearest(cs, false, &idx, &t) && idx == 1 && t == 2.0);
        std::vector<Candidate> none = {{0, false, 0}};
        CHECK(!nearest(none, false, &idx, &t));
    }
    // dielectric at normal incidence goes straight through when Schlick does not fire
    {
        Hittable g = make_sphere(V(0, 0, 0), 1.0);
        g.material = MAT_DIELECTRIC; g.refractive_index = 1.5;
        Sampler s(SAMPLER_DIRECT, 3, 0, 0);
        bool ok;
        int through = 0, back = 0;
        for (uint32_t k = 0; k < 2000; ++k) {
            Sampler sk(SAMPLER_DIRECT, 3, k, 0);
            Ray out = scatter(g, Ray{V(0, 0, -3), V(0, 0, 1)}, V(0, 0, -1), sk, 1, &ok);
            if (out.dir[2] == 1.0) ++through;
            if (out.dir[2] == -1.0) ++back;
        }
        CHECK(through + back == 2000);
        CHECK(back > 40 && back < 130);  // reflectance 0.04
    }
    // scene generator: counts and the four fixed spheres first (scenegen.rs:16-61)
    {
        std::vector<Hittable> sc = gen_scene(1);
        CHECK(sc.size() == 488);
        CHECK(sc[0].radius == 1000.0 && sc[0].centre[1] == -1000.0 && sc[0].material == MAT_DIFFUSE);
        CHECK(sc[1].material == MAT_DIELECTRIC && sc[1].refractive_index == 1.5);
        CHECK(sc[2].material == MAT_DIFFUSE && sc[2].albedo[0] == 0.4);
        CHECK(sc[3].material == MAT_METAL && sc[3].fuzz == 0.0);
        int nd = 0, nm = 0, ng = 0;
        for (size_t i = 4; i < sc.size(); ++i) {
            CHECK(sc[i].radius == 0.2 && sc[i].centre[1] == 0.2);
            nd += sc[i].material == MAT_DIFFUSE; nm += sc[i].material == MAT_METAL; ng += sc[i].material == MAT_DIELECTRIC;
        }
        CHECK(nd > 340 && nd < 430 && nm > 40 && nm < 110 && ng > 5 && ng < 50);
        // 488 leaves -> 975 nodes, depth ceil(log2 488) + 1 levels
        Scene s; s.hittables = sc;
        CHECK(build_tree(s));
        CHECK(s.tree_nodes == 975);
        CHECK(tree_depth(s.tree.get()) == 10);
    }
    if (failures == 0) std::printf("oracle selftest: all checks passed\n");
    return failures == 0 ? 0 : 1;
}
