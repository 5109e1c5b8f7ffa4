// rt_bvh.cpp — host BVH build: binned SAH, one sphere per leaf, flattened into
// 64-byte two-child nodes in depth-first order.
//
// Replaces make_coveringtree (src/boundingvolume.rs:192-208), whose median split
// on box EXTENT yields almost fully overlapping siblings.  The topology is free
// to differ: a leaf of the reference tree is reached iff its own box passes the
// line test (an ancestor's per-axis interval always contains its descendants'),
// so the nearest hit does not depend on how the boxes are grouped.
//
// A subtree over m spheres has exactly m - 1 inner nodes and m leaves, so the depth-first position of every node
// and leaf is known before the subtree is built: both halves of a split write into their own index ranges, and
// large halves are built on separate threads.  The result does not depend on the number of threads.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <future>
#include <limits>
#include <thread>

#include "rt_internal.h"

namespace rt {
namespace {

struct Box {
    double lo[3], hi[3];
    void reset() {
        for (int a = 0; a < 3; ++a) { lo[a] = std::numeric_limits<double>::infinity(); hi[a] = -lo[a]; }
    }
    void grow(const Box& b) {
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); }
    }
    void grow(const double p[3]) {
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); }
    }
    double half_area() const {
        double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0.0)) return 0.0;
        return dx * dy + dy * dz + dz * dx;
    }
};

float round_down(double x) {
    float f = float(x);
    if (double(f) > x) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
    return f;
}
float round_up(double x) {
    float f = float(x);
    if (double(f) < x) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
    return f;
}

uint32_t ceil_log2(size_t m) {
    uint32_t l = 0;
    while ((size_t(1) << l) < m) ++l;
    return l;
}

constexpr int kBins = 32;

struct Builder {
    const double* sph;
    std::vector<Box> prim_box;
    std::vector<uint32_t> idx;
    uint32_t max_depth;
    BuiltBvh* out;
    std::atomic<uint32_t> depth{0};
    std::atomic<int> threads_left{0};  // extra threads that may still be started
    static constexpr size_t kParallelMin = 16384;  // spheres in a subtree worth a thread of its own

    const double* centroid(uint32_t p) const { return sph + 4 * size_t(p); }

    // Returns the child reference for [begin, end) and its bounds.
    // node_base / leaf_base: where this subtree's inner nodes and leaves start in depth-first order.
    int32_t build(size_t begin, size_t end, uint32_t level, size_t node_base, size_t leaf_base, Box* bounds) {
        const size_t m = end - begin;
        if (m == 1) {
            *bounds = prim_box[idx[begin]];
            out->leaf_order[leaf_base] = idx[begin];
            return ~int32_t(leaf_base);
        }
        uint32_t seen = depth.load(std::memory_order_relaxed);
        while (seen < level && !depth.compare_exchange_weak(seen, level, std::memory_order_relaxed)) {}
        const int32_t me = int32_t(node_base);

        size_t mid = begin;
        const bool forced_median = (max_depth - level) < ceil_log2(m);  // levels left below this one
        bool split_done = false;
        if (!forced_median) {
            Box cb;
            cb.reset();
            for (size_t i = begin; i < end; ++i) cb.grow(centroid(idx[i]));
            double best_cost = std::numeric_limits<double>::infinity();
            int best_axis = -1, best_bin = -1;
            for (int axis = 0; axis < 3; ++axis) {
                const double ext = cb.hi[axis] - cb.lo[axis];
                if (!(ext > 0.0)) continue;
                const double scale = double(kBins) / ext;
                Box bin_box[kBins];
                size_t bin_cnt[kBins];
                for (int b = 0; b < kBins; ++b) { bin_box[b].reset(); bin_cnt[b] = 0; }
                for (size_t i = begin; i < end; ++i) {
                    int b = int((centroid(idx[i])[axis] - cb.lo[axis]) * scale);
                    b = std::min(std::max(b, 0), kBins - 1);
                    bin_box[b].grow(prim_box[idx[i]]);
                    bin_cnt[b]++;
                }
                double right_area[kBins];
                size_t right_cnt[kBins];
                Box acc;
                acc.reset();
                size_t cnt = 0;
                for (int b = kBins - 1; b > 0; --b) {
                    acc.grow(bin_box[b]); cnt += bin_cnt[b];
                    right_area[b] = acc.half_area(); right_cnt[b] = cnt;
                }
                acc.reset();
                cnt = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    acc.grow(bin_box[b]); cnt += bin_cnt[b];
                    if (cnt == 0 || right_cnt[b + 1] == 0) continue;
                    double cost = acc.half_area() * double(cnt) + right_area[b + 1] * double(right_cnt[b + 1]);
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
                }
            }
            if (best_axis >= 0) {
                const double lo = cb.lo[best_axis], scale = double(kBins) / (cb.hi[best_axis] - cb.lo[best_axis]);
                auto it = std::partition(idx.begin() + begin, idx.begin() + end, [&](uint32_t p) {
                    int b = int((centroid(p)[best_axis] - lo) * scale);
                    b = std::min(std::max(b, 0), kBins - 1);
                    return b <= best_bin;
                });
                mid = size_t(it - idx.begin());
                split_done = (mid > begin && mid < end);
            }
        }
        if (!split_done) {  // balanced split on the longest centroid axis (also the depth-limit fallback)
            Box cb;
            cb.reset();
            for (size_t i = begin; i < end; ++i) cb.grow(centroid(idx[i]));
            int axis = 0;
            for (int a = 1; a < 3; ++a)
                if (cb.hi[a] - cb.lo[a] > cb.hi[axis] - cb.lo[axis]) axis = a;
            mid = begin + m / 2;
            std::nth_element(idx.begin() + begin, idx.begin() + mid, idx.begin() + end, [&](uint32_t a, uint32_t b) {
                double ca = centroid(a)[axis], cb2 = centroid(b)[axis];
                return ca < cb2 || (ca == cb2 && a < b);
            });
        }
        Box b0, b1;
        const size_t m0 = mid - begin;  // left half: m0 - 1 inner nodes right after this one, m0 leaves
        int32_t c0, c1;
        if (m >= kParallelMin && threads_left.fetch_sub(1, std::memory_order_relaxed) > 0) {
            auto right = std::async(std::launch::async, [&]() { return build(mid, end, level + 1, node_base + m0, leaf_base + m0, &b1); });
            c0 = build(begin, mid, level + 1, node_base + 1, leaf_base, &b0);
            c1 = right.get();
            threads_left.fetch_add(1, std::memory_order_relaxed);
        } else {
            if (m >= kParallelMin) threads_left.fetch_add(1, std::memory_order_relaxed);
            c0 = build(begin, mid, level + 1, node_base + 1, leaf_base, &b0);
            c1 = build(mid, end, level + 1, node_base + m0, leaf_base + m0, &b1);
        }
        Node& nd = out->nodes[size_t(me)];
        for (int a = 0; a < 3; ++a) {
            nd.lo0[a] = round_down(b0.lo[a]); nd.hi0[a] = round_up(b0.hi[a]);
            nd.lo1[a] = round_down(b1.lo[a]); nd.hi1[a] = round_up(b1.hi[a]);
        }
        nd.child0 = c0; nd.child1 = c1;
        nd.pad[0] = nd.pad[1] = 0;
        *bounds = b0;
        bounds->grow(b1);
        return me;
    }
};

}  // namespace

void build_bvh(const double* spheres, size_t n, uint32_t max_depth, BuiltBvh* out) {
    out->nodes.clear();
    out->leaf_order.clear();
    out->depth = 0;
    for (int a = 0; a < 3; ++a) { out->scene_lo[a] = 0.0; out->scene_hi[a] = 0.0; }
    if (n == 0) return;
    Builder b;
    b.sph = spheres;
    b.max_depth = std::max(max_depth, ceil_log2(n) + 1);
    b.out = out;
    b.prim_box.resize(n);
    b.idx.resize(n);
    Box scene;
    scene.reset();
    for (size_t i = 0; i < n; ++i) {
        const double* s = spheres + 4 * i;
        // src/boundingvolume.rs:246-248 with |r|: a sphere given with a negative radius (the hollow-glass idiom:
        // same surface, inward normal) occupies the same box; taken as given its box would be inside out and no
        // parent would cover it
        const double r = std::fabs(s[3]);
        for (int a = 0; a < 3; ++a) {
            b.prim_box[i].lo[a] = s[a] - r;
            b.prim_box[i].hi[a] = s[a] + r;
        }
        scene.grow(b.prim_box[i]);
        b.idx[i] = uint32_t(i);
    }
    out->nodes.assign(n > 1 ? n - 1 : 0, Node());
    out->leaf_order.assign(n, 0);
    const unsigned hw = std::thread::hardware_concurrency();
    b.threads_left = int(std::min<unsigned>(hw ? hw : 1u, 16u)) - 1;
    Box root;
    int32_t ref = b.build(0, n, 1, 0, 0, &root);
    out->depth = b.depth.load();
    if (ref < 0) {
        // a single sphere: the root's second child is the same leaf behind a far-away point box
        // (entering it would only repeat the same sphere test)
        Node nd;
        for (int a = 0; a < 3; ++a) {
            nd.lo0[a] = round_down(root.lo[a]); nd.hi0[a] = round_up(root.hi[a]);
            nd.lo1[a] = 3.0e38f; nd.hi1[a] = 3.0e38f;
        }
        nd.child0 = ref; nd.child1 = ref;
        nd.pad[0] = nd.pad[1] = 0;
        out->nodes.push_back(nd);
        out->depth = 1;
    }
    for (int a = 0; a < 3; ++a) { out->scene_lo[a] = scene.lo[a]; out->scene_hi[a] = scene.hi[a]; }
}

}  // namespace rt
