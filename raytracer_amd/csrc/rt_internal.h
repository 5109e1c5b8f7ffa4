// rt_internal.h — declarations shared by the host and device halves of the library.
#pragma once
#include <cstdarg>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rt_mi355x.h"

namespace rt {

// Records the message returned by rt_last_error() for the calling thread and returns `code`.
int fail(int code, const char* fmt, ...);

// ---- flattened BVH (host side; uploaded as-is) -----------------------------
// Two-child node that stores the boxes of BOTH children, so one 64-byte fetch
// decides both descents.  child >= 0: index of an inner node; child < 0: leaf,
// ~child = position of the sphere in leaf order.  Bounds are f32, rounded
// outward from the f64 boxes [c - r, c + r] (src/boundingvolume.rs:242-253).
struct alignas(64) Node {
    float lo0[3], hi0[3];
    float lo1[3], hi1[3];
    int32_t child0, child1;
    int32_t pad[2];
};
static_assert(sizeof(Node) == 64, "Node must stay one 64-byte line");

struct BuiltBvh {
    std::vector<Node> nodes;          // inner nodes, depth-first order, root = 0
    std::vector<uint32_t> leaf_order; // leaf position -> index into the sphere list given to build_bvh
    uint32_t depth = 0;               // inner-node levels on the longest root-to-leaf path
    double scene_lo[3] = {0, 0, 0}, scene_hi[3] = {0, 0, 0};
};

// Binned-SAH build over spheres (cx,cy,cz,r as 4 doubles each).  max_depth
// bounds the inner-node levels (and with it the traversal stack).
void build_bvh(const double* spheres, size_t n, uint32_t max_depth, BuiltBvh* out);

// The same contract, built on the current HIP device: Morton codes, radix sort, one thread per inner node
// for the topology, one thread per leaf for the boxes (rt_lbvh.hip).  Returns an rt_status.
int build_lbvh_device(const double* spheres, size_t n, BuiltBvh* out);

// The same build with the result left in device memory (n >= 2): *d_sph_out = the spheres as given (4 doubles each,
// build order), *d_nodes_out = n - 1 nodes in depth-first order, *d_order_out = leaf position -> build order; the
// caller owns the three buffers (hipFree).  scene_lo / scene_hi = box around the spheres.
int build_lbvh_resident(const double* spheres, size_t n, double** d_sph_out, Node** d_nodes_out, uint32_t** d_order_out,
                        uint32_t* depth_out, double scene_lo[3], double scene_hi[3]);

// SplitMix64 used by the seeded scene generators.
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

}  // namespace rt
