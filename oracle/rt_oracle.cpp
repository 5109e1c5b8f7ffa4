// rt_oracle.cpp — C exports of the CPU restatement, for ctypes.
// TEST INFRASTRUCTURE ONLY (see rt_oracle.hpp).  Built by oracle/Makefile into
// oracle/build/librt_oracle.so with -ffp-contract=off.
#include "rt_oracle.hpp"

#include <chrono>
#include <cstdio>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace orc;

extern "C" {

struct orc_scene { Scene sc; };

// Copies hittables and textures (rgb8, row-major, row 0 = top).
orc_scene* orc_scene_create(const Hittable* h, size_t n, const Texture* tex, size_t n_tex) {
    orc_scene* s = new orc_scene;
    s->sc.hittables.assign(h, h + n);
    for (size_t i = 0; i < n_tex; ++i) {
        s->sc.texture_data.emplace_back(tex[i].rgb8, tex[i].rgb8 + size_t(3) * tex[i].width * tex[i].height);
        s->sc.textures.push_back(Texture{nullptr, tex[i].width, tex[i].height});
    }
    for (size_t i = 0; i < n_tex; ++i) s->sc.textures[i].rgb8 = s->sc.texture_data[i].data();
    return s;
}
void orc_scene_destroy(orc_scene* s) { delete s; }

// Builds the reference's pointer tree; returns node count, or -1 when the scene
// holds a non-sphere (the reference hits unimplemented!()).
long orc_scene_build_tree(orc_scene* s) {
    if (!build_tree(s->sc)) return -1;
    return long(s->sc.tree_nodes);
}
int orc_scene_tree_depth(const orc_scene* s) { return tree_depth(s->sc.tree.get()); }
void orc_scene_set_tie_rule(orc_scene* s, int lowest_index) { s->sc.tie_lowest_index = lowest_index != 0; }
// Test aid: see Scene::dir_noise (0 = the reference's behaviour).
void orc_scene_set_dir_noise(orc_scene* s, double amplitude) { s->sc.dir_noise = amplitude; }

void orc_camera_build(const double lookat[3], const double lookfrom[3], double inv_focal_length, double aperture,
                      uint32_t horiz_res, uint32_t vert_res, Camera* out) {
    *out = camera_build(V(lookat), V(lookfrom), inv_focal_length, aperture, horiz_res, vert_res);
}

size_t orc_gen_scene(uint64_t seed, Hittable* out, size_t cap) {
    std::vector<Hittable> v = gen_scene(seed);
    if (out) std::memcpy(out, v.data(), sizeof(Hittable) * std::min(cap, v.size()));
    return v.size();
}

// Nearest hit for n rays given as (orig, dir) doubles; dir is used as given
// (Sphere::intersect never renormalises, geometry.rs:104-121).
// id = -1 on a miss; t and normal are written only for hits.
void orc_trace(const orc_scene* s, int traverse, const double* orig, const double* dir, size_t n, int32_t* id,
               double* t, double* normal, uint64_t* counters /* box, shape */) {
    TraceCounters total;
#pragma omp parallel
    {
        std::vector<Candidate> scratch;
        TraceCounters ctr;
#pragma omp for schedule(dynamic, 256)
        for (long k = 0; k < long(n); ++k) {
            Ray r{V(orig + 3 * k), V(dir + 3 * k)};
            candidates(s->sc, traverse, r, scratch, &ctr);
            int idx; double tt;
            if (nearest(scratch, s->sc.tie_lowest_index || traverse != TRAVERSE_TREE, &idx, &tt)) {
                id[k] = idx;
                t[k] = tt;
                Vec3 nn = shape_normal_at(s->sc.hittables[idx], r.position_at(tt));
                for (int i = 0; i < 3; ++i) normal[3 * k + i] = nn[i];
            } else {
                id[k] = -1;
            }
        }
#pragma omp critical
        { total.box_tests += ctr.box_tests; total.shape_tests += ctr.shape_tests; }
    }
    if (counters) { counters[0] = total.box_tests; counters[1] = total.shape_tests; }
}

// Camera rays for pixels [pixel_begin, pixel_end) (row-major index j*W+i) and
// samples [sample_begin, sample_end); ray r = (pixel - pixel_begin)*ns + (s - sample_begin).
void orc_generate_primary(const Camera* cam, uint64_t seed, int sampler, uint32_t pixel_begin, uint32_t pixel_end,
                          uint32_t sample_begin, uint32_t sample_end, double* orig, double* dir) {
    size_t ns = sample_end - sample_begin;
    for (uint32_t p = pixel_begin; p < pixel_end; ++p)
        for (uint32_t sidx = sample_begin; sidx < sample_end; ++sidx) {
            Sampler smp(sampler, seed, p, sidx);
            Ray r = camera_ray(*cam, p % cam->horiz_res, p / cam->horiz_res, smp);
            size_t k = size_t(p - pixel_begin) * ns + (sidx - sample_begin);
            for (int i = 0; i < 3; ++i) { orig[3 * k + i] = r.orig[i]; dir[3 * k + i] = r.dir[i]; }
        }
}

// lib.rs:137-184 without the file: sum[pixel] = sum over samples
// [sample_begin, sample_end) of the traced colour, added in sample order from
// zero.  rgb8 (optional) = color_to_ppm((1/spp)*sum) with spp = sample_end -
// sample_begin.  threads <= 1 runs the reference's single-threaded loop nest;
// otherwise scanlines are spread over OpenMP threads (per-pixel results do not
// depend on the thread count).  Only rows row_begin, row_begin + row_stride, ...
// below row_end are rendered (a bounded sample of a large frame).  counters: box tests, shape tests, segments, paths.
// Returns wall seconds of the render loop.
double orc_render(const orc_scene* s, const Camera* cam, int traverse, int sampler, uint32_t sample_begin,
                  uint32_t sample_end, uint32_t depth, uint64_t seed, int threads, uint32_t row_begin,
                  uint32_t row_end, uint32_t row_stride, double* sum_rgb, uint8_t* rgb8, uint64_t* counters) {
    const uint32_t W = cam->horiz_res;
    const double inv_spp = 1.0 / double(sample_end - sample_begin);
    TraceCounters total;
    auto t0 = std::chrono::steady_clock::now();
#ifdef _OPENMP
    int nt = threads <= 1 ? 1 : threads;
#else
    int nt = 1; (void)threads;
#endif
#pragma omp parallel num_threads(nt)
    {
        std::vector<Candidate> scratch;
        TraceCounters ctr;
#pragma omp for schedule(dynamic, 1)
        for (long j = long(row_begin); j < long(row_end); j += long(row_stride ? row_stride : 1)) {
            for (uint32_t i = 0; i < W; ++i) {
                uint32_t pixel = uint32_t(j) * W + i;
                Color acc{0.0, 0.0, 0.0};
                for (uint32_t sidx = sample_begin; sidx < sample_end; ++sidx) {
                    Sampler smp(sampler, seed, pixel, sidx);
                    Ray r = camera_ray(*cam, i, uint32_t(j), smp);
                    acc = acc + trace(s->sc, traverse, r, depth, smp, scratch, &ctr);
                    ctr.paths++;
                }
                if (sum_rgb) { sum_rgb[3 * size_t(pixel)] = acc.r; sum_rgb[3 * size_t(pixel) + 1] = acc.g; sum_rgb[3 * size_t(pixel) + 2] = acc.b; }
                if (rgb8) {
                    Color m = inv_spp * acc;  // lib.rs:148/175
                    rgb8[3 * size_t(pixel)] = channel_to_ppm(m.r);
                    rgb8[3 * size_t(pixel) + 1] = channel_to_ppm(m.g);
                    rgb8[3 * size_t(pixel) + 2] = channel_to_ppm(m.b);
                }
            }
        }
#pragma omp critical
        {
            total.box_tests += ctr.box_tests; total.shape_tests += ctr.shape_tests;
            total.segments += ctr.segments; total.paths += ctr.paths;
        }
    }
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (counters) { counters[0] = total.box_tests; counters[1] = total.shape_tests; counters[2] = total.segments; counters[3] = total.paths; }
    return secs;
}

// Radiance of single paths (pixel, sample) — lets a test compare path by path.
// segments (optional) receives the number of segments each path traced.
void orc_trace_paths(const orc_scene* s, const Camera* cam, int traverse, int sampler, const uint32_t* pixel,
                     const uint32_t* sample, size_t n, uint32_t depth, uint64_t seed, double* rgb, uint32_t* segments) {
    // paths are independent (each has its own sampler): threads share nothing but the scene
#pragma omp parallel
    {
        std::vector<Candidate> scratch;
#pragma omp for schedule(dynamic, 8)
        for (size_t k = 0; k < n; ++k) {
            Sampler smp(sampler, seed, pixel[k], sample[k]);
            Ray r = camera_ray(*cam, pixel[k] % cam->horiz_res, pixel[k] / cam->horiz_res, smp);
            TraceCounters ctr;
            Color c = trace(s->sc, traverse, r, depth, smp, scratch, &ctr);
            rgb[3 * k] = c.r; rgb[3 * k + 1] = c.g; rgb[3 * k + 2] = c.b;
            if (segments) segments[k] = uint32_t(ctr.segments);
        }
    }
}

// Segment-by-segment dump of single paths (primitive ID, -1 = closing miss; parameter).
void orc_path_segments(const orc_scene* s, const Camera* cam, int traverse, int sampler, const uint32_t* pixel,
                       const uint32_t* sample, size_t n, uint32_t depth, uint64_t seed, uint32_t max_segments,
                       int32_t* seg_id, double* seg_t, uint32_t* n_seg) {
#pragma omp parallel
    {
        std::vector<Candidate> scratch;
#pragma omp for schedule(dynamic, 8)
        for (size_t k = 0; k < n; ++k) {
            Sampler smp(sampler, seed, pixel[k], sample[k]);
            Ray r = camera_ray(*cam, pixel[k] % cam->horiz_res, pixel[k] / cam->horiz_res, smp);
            SegmentLog log{seg_id + k * max_segments, seg_t + k * max_segments, max_segments, 0};
            trace(s->sc, traverse, r, depth, smp, scratch, nullptr, &log);
            n_seg[k] = log.n;
        }
    }
}

void orc_finalize(const double* sum_rgb, size_t n_channels, double spp, uint8_t* rgb8) {
    double inv = 1.0 / spp;
    for (size_t k = 0; k < n_channels; ++k) rgb8[k] = channel_to_ppm(inv * sum_rgb[k]);
}

// --- small pieces exported one by one for the known-answer tests -----------
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox4x32_10(ctr, key, out); }
// returns 1 and *t on a hit, 0 for None; normalise_dir mirrors Ray::new
int orc_sphere_intersect(const double c[3], double r, const double o[3], const double d[3], int normalise_dir, double* t) {
    Ray ray = normalise_dir ? Ray::make(V(o), V(d)) : Ray{V(o), V(d)};
    return sphere_intersect(V(c), r, ray, t) ? 1 : 0;
}
int orc_disc_intersect(const double c[3], const double nrm[3], double r, int normalise_normal, const double o[3],
                       const double d[3], int normalise_dir, double* t) {
    Ray ray = normalise_dir ? Ray::make(V(o), V(d)) : Ray{V(o), V(d)};
    Vec3 n = normalise_normal ? normalize(V(nrm)) : V(nrm);  // Disc::new normalises (geometry.rs:129-131)
    return disc_intersect(V(c), n, r, ray, t) ? 1 : 0;
}
int orc_bbox_check(const double dims[6], const double o[3], const double d[3]) {
    BoundingBox bb{{{dims[0], dims[1]}, {dims[2], dims[3]}, {dims[4], dims[5]}}, -1};
    return bbox_check_intersection(bb, Ray{V(o), V(d)}) ? 1 : 0;
}
int orc_bbox_longest_axis(const double dims[6]) {
    BoundingBox bb{{{dims[0], dims[1]}, {dims[2], dims[3]}, {dims[4], dims[5]}}, -1};
    return bbox_longest_axis(bb);
}
int orc_interval_intersection(double a0, double a1, double b0, double b1, double out[2]) {
    Interval r;
    if (!intersection(Interval{a0, a1}, Interval{b0, b1}, &r)) return 0;
    out[0] = r.start; out[1] = r.end;
    return 1;
}
void orc_interval_cover(double a0, double a1, double b0, double b1, double out[2]) {
    Interval r = cover(Interval{a0, a1}, Interval{b0, b1});
    out[0] = r.start; out[1] = r.end;
}
// Builds the reference tree over n boxes (6 doubles each) and writes, in
// pre-order, each node's cover (6 doubles) and boxed index (-1 for inner nodes).
size_t orc_tree_over_boxes(const double* dims, size_t n, double* covers, int32_t* boxed, size_t cap) {
    std::vector<BoundingBox> boxes(n);
    for (size_t i = 0; i < n; ++i)
        boxes[i] = BoundingBox{{{dims[6 * i], dims[6 * i + 1]}, {dims[6 * i + 2], dims[6 * i + 3]}, {dims[6 * i + 4], dims[6 * i + 5]}}, int(i)};
    std::unique_ptr<CoveringTree> t = make_coveringtree(boxes.data(), n);
    size_t k = 0;
    struct W { static void go(const CoveringTree* t, double* covers, int32_t* boxed, size_t cap, size_t& k) {
        if (!t) return;
        if (k < cap) {
            for (int i = 0; i < 3; ++i) { covers[6 * k + 2 * i] = t->cover.dims[i].start; covers[6 * k + 2 * i + 1] = t->cover.dims[i].end; }
            boxed[k] = t->cover.boxed;
        }
        ++k;
        go(t->left.get(), covers, boxed, cap, k);
        go(t->right.get(), covers, boxed, cap, k);
    } };
    W::go(t.get(), covers, boxed, cap, k);
    return k;
}
// Sorts n boxes by extent on axis idx (boundingvolume.rs:142-144); writes the permutation.
void orc_sort_on_index(const double* dims, size_t n, int idx, int32_t* order) {
    std::vector<BoundingBox> boxes(n);
    for (size_t i = 0; i < n; ++i)
        boxes[i] = BoundingBox{{{dims[6 * i], dims[6 * i + 1]}, {dims[6 * i + 2], dims[6 * i + 3]}, {dims[6 * i + 4], dims[6 * i + 5]}}, int(i)};
    sort_on_index(boxes.data(), n, idx);
    for (size_t i = 0; i < n; ++i) order[i] = boxes[i].boxed;
}
uint8_t orc_channel_to_ppm(double c) { return channel_to_ppm(c); }
int orc_schlick(double cosine, double r_idx, double u) { return schlick(cosine, r_idx, u) ? 1 : 0; }
// One scatter through a single hittable, with a sampler at (pixel, sample, bounce).
int orc_scatter(const Hittable* h, const double o[3], const double d[3], const double loc[3], int sampler, uint64_t seed,
                uint32_t pixel, uint32_t sample, uint32_t bounce, double out_o[3], double out_d[3]) {
    Sampler smp(sampler, seed, pixel, sample);
    bool ok;
    Ray r = scatter(*h, Ray{V(o), V(d)}, V(loc), smp, bounce, &ok);
    for (int i = 0; i < 3; ++i) { out_o[i] = r.orig[i]; out_d[i] = r.dir[i]; }
    return ok ? 1 : 0;
}
void orc_albedo(const orc_scene* s, int index, const double loc[3], double rgb[3]) {
    Color c = albedo(s->sc.hittables[index], s->sc.textures.data(), V(loc));
    rgb[0] = c.r; rgb[1] = c.g; rgb[2] = c.b;
}
int orc_max_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
size_t orc_sizeof_hittable() { return sizeof(Hittable); }
size_t orc_sizeof_camera() { return sizeof(Camera); }

}  // extern "C"
