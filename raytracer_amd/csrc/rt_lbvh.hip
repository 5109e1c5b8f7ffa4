// rt_lbvh.hip — BVH build on the device for large scenes (linear BVH over Morton-sorted centres).
//
// Replaces make_coveringtree (src/boundingvolume.rs:192-208) like the host SAH builder does, for
// scenes where the host build dominates a short render (1 M spheres: 0.9 s on the host).  The tree
// differs from the host one (no surface-area heuristic), the nearest hit of a ray does not: it is
// the same argmin over the spheres whose boxes the conservative descent reaches.
//
//   1. 63-bit Morton code of every centre: 21 bits per axis on ONE cubic grid over the box around all centres (the
//      same cell size on every axis: a flat scene then spends no splits on its thin axis until the cells get that
//      small, where a grid normalised per axis would cut it into slabs)
//   2. radix sort of (code, sphere index) — hipCUB, stable, so equal codes keep index order
//   3. one thread per inner node finds its range and split by common-prefix lengths of neighbouring
//      keys (key = code then position, so duplicates still split) and records the parent links
//   4. one thread per leaf climbs towards the root; the second thread to reach a node (an atomic
//      flag per node) finds the other child's box already stored, so every node gets both child
//      boxes (f32, rounded outward from the f64 boxes [c - r, c + r] of src/boundingvolume.rs:246-248)
//      and the number of levels below it
//   5. one thread per inner node renumbers it in depth-first order (a node's new index = first leaf of its range
//      + the number of left turns on the way down to it), so that a node's left child is its neighbour in memory
//      as in the host builder's trees
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "rt_internal.h"

namespace rt {
namespace {

#define LBVH_HIP(call)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) { rc = fail(RT_ERR_DEVICE, "%s: %s", #call, hipGetErrorString(e_)); goto done; } \
    } while (0)

typedef unsigned long long key_t;
__device__ __forceinline__ key_t spread21(uint32_t v) {  // 21 bits -> every third bit of 63
    key_t x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

// inv_cell = 2^21 / (the largest extent of the box around the centres): one cubic grid for all three axes
__global__ void morton_kernel(const double4* __restrict__ sph, uint32_t n, double3 lo, double inv_cell,
                              key_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double4 s = sph[i];
    const uint32_t x = uint32_t(fmin(fmax((s.x - lo.x) * inv_cell, 0.0), 2097151.0));
    const uint32_t y = uint32_t(fmin(fmax((s.y - lo.y) * inv_cell, 0.0), 2097151.0));
    const uint32_t z = uint32_t(fmin(fmax((s.z - lo.z) * inv_cell, 0.0), 2097151.0));
    keys[i] = (spread21(x) << 2) | (spread21(y) << 1) | spread21(z);
    vals[i] = i;
}

// length of the common prefix of keys i and j (code first, then position); -1 outside the array
__device__ __forceinline__ int prefix(const key_t* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const key_t a = keys[i], b = keys[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz(uint32_t(i) ^ uint32_t(j));
}

// Inner node i of n - 1 (node 0 is the root); leaves are coded ~position as in Node.
__global__ void hierarchy_kernel(const key_t* __restrict__ keys, int n, Node* __restrict__ nodes,
                                 int32_t* __restrict__ parent_of_inner, int32_t* __restrict__ parent_of_leaf,
                                 int32_t* __restrict__ first_leaf) {
    const int i = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n - 1) return;
    const int d = (prefix(keys, n, i, i + 1) - prefix(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = prefix(keys, n, i, i - d);
    int lmax = 2;
    while (prefix(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (prefix(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = prefix(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (prefix(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int split = i + s * d + min(d, 0);
    const int first = min(i, j), last = max(i, j);
    const int32_t left = (first == split) ? ~split : split;
    const int32_t right = (last == split + 1) ? ~(split + 1) : split + 1;
    nodes[i].child0 = left;
    nodes[i].child1 = right;
    first_leaf[i] = first;
    if (left < 0) parent_of_leaf[~left] = i; else parent_of_inner[left] = i;
    if (right < 0) parent_of_leaf[~right] = i; else parent_of_inner[right] = i;
    if (i == 0) parent_of_inner[0] = -1;
}

// One thread per leaf.  pad[0] / pad[1] of a node receive the inner-node levels below child 0 / child 1.
__global__ void refit_kernel(const double4* __restrict__ sph, const uint32_t* __restrict__ order, int n,
                             Node* __restrict__ nodes, const int32_t* __restrict__ parent_of_inner,
                             const int32_t* __restrict__ parent_of_leaf, uint32_t* __restrict__ arrived) {
    const int k = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (k >= n) return;
    const double4 s = sph[order[k]];
    const double r = fabs(s.w);  // like the host builder: src/boundingvolume.rs:246-248 with |r|
    float lo[3] = {__double2float_rd(s.x - r), __double2float_rd(s.y - r), __double2float_rd(s.z - r)};
    float hi[3] = {__double2float_ru(s.x + r), __double2float_ru(s.y + r), __double2float_ru(s.z + r)};
    int32_t levels = 0;
    int32_t me = ~k;
    int32_t node = parent_of_leaf[k];
    while (node >= 0) {
        Node* nd = nodes + node;
        const bool is_left = nd->child0 == me;
        float* mylo = is_left ? nd->lo0 : nd->lo1;
        float* myhi = is_left ? nd->hi0 : nd->hi1;
        for (int a = 0; a < 3; ++a) { mylo[a] = lo[a]; myhi[a] = hi[a]; }
        nd->pad[is_left ? 0 : 1] = levels;
        __threadfence();
        if (atomicAdd(arrived + node, 1u) == 0u) return;  // the sibling's thread completes this node
        __threadfence();
        const volatile float* olo = is_left ? nd->lo1 : nd->lo0;
        const volatile float* ohi = is_left ? nd->hi1 : nd->hi0;
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], olo[a]); hi[a] = fmaxf(hi[a], ohi[a]); }
        const volatile int32_t* opad = nd->pad;
        levels = 1 + max(levels, opad[is_left ? 1 : 0]);
        me = node;
        node = parent_of_inner[node];
    }
}

// Depth-first (preorder) index of an inner node: its first leaf + the left turns on the path from the root.
// (Leaves before the node hang off that path in subtrees that were left behind at right turns; a subtree of m leaves
// holds m - 1 inner nodes; the ancestors themselves come first too: first - right turns + depth = first + left turns.)
__global__ void relabel_kernel(const Node* __restrict__ nodes, const int32_t* __restrict__ parent_of_inner,
                               const int32_t* __restrict__ first_leaf, int n_inner, Node* __restrict__ out) {
    const int i = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n_inner) return;
    int left_turns = 0;
    for (int node = i, parent = parent_of_inner[i]; parent >= 0; node = parent, parent = parent_of_inner[parent])
        left_turns += nodes[parent].child0 == node ? 1 : 0;
    Node nd = nodes[i];
    if (nd.child0 >= 0) nd.child0 = first_leaf[nd.child0] + left_turns + 1;
    if (nd.child1 >= 0) nd.child1 = first_leaf[nd.child1] + left_turns;
    nd.pad[0] = nd.pad[1] = 0;
    out[first_leaf[i] + left_turns] = nd;
}

}  // namespace

namespace {

// Steps 1-5 on spheres that are already in device memory (n >= 2).  On success *nodes_out (n - 1 nodes, depth-first
// order) and *order_out (leaf position -> index into d_sph) are device buffers the caller frees with hipFree.
int lbvh_on_device(const double4* d_sph, size_t n, const double clo[3], const double chi[3], Node** nodes_out, uint32_t** order_out,
                   uint32_t* depth_out) {
    *nodes_out = nullptr;
    *order_out = nullptr;
    const double3 lo = make_double3(clo[0], clo[1], clo[2]);
    const double extent = std::max(std::max(chi[0] - clo[0], chi[1] - clo[1]), chi[2] - clo[2]);
    const double inv_cell = extent > 0.0 ? 2097152.0 / extent : 0.0;
    int rc = RT_OK;
    const int ni = int(n);
    key_t *d_keys = nullptr, *d_keys2 = nullptr;
    uint32_t *d_vals = nullptr, *d_vals2 = nullptr, *d_arrived = nullptr;
    int32_t *d_pin = nullptr, *d_pleaf = nullptr, *d_first = nullptr;
    Node *d_nodes = nullptr, *d_nodes_dfs = nullptr;
    void* d_tmp = nullptr;
    size_t tmp_bytes = 0;
    const uint32_t block = 256, grid_n = uint32_t((n + block - 1) / block), grid_i = uint32_t((n - 1 + block - 1) / block);
    {
        LBVH_HIP(hipMalloc(&d_keys, n * sizeof(key_t))); LBVH_HIP(hipMalloc(&d_keys2, n * sizeof(key_t)));
        LBVH_HIP(hipMalloc(&d_vals, n * 4)); LBVH_HIP(hipMalloc(&d_vals2, n * 4));
        LBVH_HIP(hipMalloc(&d_pin, n * 4)); LBVH_HIP(hipMalloc(&d_pleaf, n * 4));
        LBVH_HIP(hipMalloc(&d_arrived, n * 4));
        LBVH_HIP(hipMalloc(&d_nodes, (n - 1) * sizeof(Node)));
        LBVH_HIP(hipMalloc(&d_nodes_dfs, (n - 1) * sizeof(Node)));
        LBVH_HIP(hipMalloc(&d_first, n * 4));
        LBVH_HIP(hipMemsetAsync(d_arrived, 0, n * 4, nullptr));
        LBVH_HIP(hipMemsetAsync(d_nodes, 0, (n - 1) * sizeof(Node), nullptr));
        hipLaunchKernelGGL(morton_kernel, dim3(grid_n), dim3(block), 0, nullptr, d_sph, uint32_t(n), lo, inv_cell, d_keys, d_vals);
        LBVH_HIP(hipGetLastError());
        LBVH_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_keys, d_keys2, d_vals, d_vals2, ni, 0, 63, nullptr));
        LBVH_HIP(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16));
        LBVH_HIP(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_keys, d_keys2, d_vals, d_vals2, ni, 0, 63, nullptr));
        hipLaunchKernelGGL(hierarchy_kernel, dim3(grid_i), dim3(block), 0, nullptr, d_keys2, ni, d_nodes, d_pin, d_pleaf, d_first);
        LBVH_HIP(hipGetLastError());
        hipLaunchKernelGGL(refit_kernel, dim3(grid_n), dim3(block), 0, nullptr, d_sph, d_vals2, ni, d_nodes, d_pin, d_pleaf, d_arrived);
        LBVH_HIP(hipGetLastError());
        {
            Node root;  // pad[] of the root = levels below its children; the renumbering clears the pads
            LBVH_HIP(hipMemcpy(&root, d_nodes, sizeof(Node), hipMemcpyDeviceToHost));
            *depth_out = uint32_t(1 + std::max(root.pad[0], root.pad[1]));
        }
        hipLaunchKernelGGL(relabel_kernel, dim3(grid_i), dim3(block), 0, nullptr, d_nodes, d_pin, d_first, ni - 1, d_nodes_dfs);
        LBVH_HIP(hipGetLastError());
        LBVH_HIP(hipDeviceSynchronize());
        *nodes_out = d_nodes_dfs; d_nodes_dfs = nullptr;
        *order_out = d_vals2; d_vals2 = nullptr;
    }
done:
    for (void* p : {static_cast<void*>(d_keys), static_cast<void*>(d_keys2), static_cast<void*>(d_vals),
                    static_cast<void*>(d_vals2), static_cast<void*>(d_pin), static_cast<void*>(d_pleaf),
                    static_cast<void*>(d_arrived), static_cast<void*>(d_nodes), static_cast<void*>(d_nodes_dfs),
                    static_cast<void*>(d_first), d_tmp})
        (void)hipFree(p);
    return rc;
}

// boxes around the centres (the Morton grid) and around the spheres
void sphere_extents(const double* spheres, size_t n, double clo[3], double chi[3], double slo[3], double shi[3]) {
    for (int a = 0; a < 3; ++a) { clo[a] = slo[a] = 1e300; chi[a] = shi[a] = -1e300; }
    for (size_t i = 0; i < n; ++i) {
        const double* s = spheres + 4 * i;
        for (int a = 0; a < 3; ++a) {
            clo[a] = std::min(clo[a], s[a]); chi[a] = std::max(chi[a], s[a]);
            slo[a] = std::min(slo[a], s[a] - std::fabs(s[3])); shi[a] = std::max(shi[a], s[a] + std::fabs(s[3]));
        }
    }
}

}  // namespace

// Same contract as build_bvh (rt_internal.h); runs on the current HIP device, synchronous.
int build_lbvh_device(const double* spheres, size_t n, BuiltBvh* out) {
    out->nodes.clear();
    out->leaf_order.clear();
    out->depth = 0;
    for (int a = 0; a < 3; ++a) { out->scene_lo[a] = 0.0; out->scene_hi[a] = 0.0; }
    if (n < 2) {  // nothing to sort: the host builder's special cases apply
        build_bvh(spheres, n, 1, out);
        return RT_OK;
    }
    if (n > (size_t(1) << 30)) return fail(RT_ERR_LIMIT, "build_lbvh_device: more than 2^30 spheres");
    double clo[3], chi[3];
    sphere_extents(spheres, n, clo, chi, out->scene_lo, out->scene_hi);
    int rc = RT_OK;
    double4* d_sph = nullptr;
    Node* d_nodes = nullptr;
    uint32_t* d_order = nullptr;
    {
        LBVH_HIP(hipMalloc(&d_sph, n * sizeof(double4)));
        LBVH_HIP(hipMemcpy(d_sph, spheres, n * sizeof(double4), hipMemcpyHostToDevice));
        if ((rc = lbvh_on_device(d_sph, n, clo, chi, &d_nodes, &d_order, &out->depth)) != RT_OK) goto done;
        out->nodes.resize(n - 1);
        out->leaf_order.resize(n);
        LBVH_HIP(hipMemcpy(out->nodes.data(), d_nodes, (n - 1) * sizeof(Node), hipMemcpyDeviceToHost));
        LBVH_HIP(hipMemcpy(out->leaf_order.data(), d_order, n * 4, hipMemcpyDeviceToHost));
    }
done:
    (void)hipFree(d_sph);
    (void)hipFree(d_nodes);
    (void)hipFree(d_order);
    return rc;
}

// The tree stays on the device: *d_sph_out (the spheres as uploaded, build order), *d_nodes_out (depth-first order),
// *d_order_out (leaf position -> build order) belong to the caller (hipFree).  n >= 2.
int build_lbvh_resident(const double* spheres, size_t n, double** d_sph_out, Node** d_nodes_out, uint32_t** d_order_out,
                        uint32_t* depth_out, double scene_lo[3], double scene_hi[3]) {
    *d_sph_out = nullptr; *d_nodes_out = nullptr; *d_order_out = nullptr;
    if (n < 2 || n > (size_t(1) << 30)) return fail(RT_ERR_LIMIT, "build_lbvh_resident: %zu spheres", n);
    double clo[3], chi[3];
    sphere_extents(spheres, n, clo, chi, scene_lo, scene_hi);
    int rc = RT_OK;
    double4* d_sph = nullptr;
    {
        LBVH_HIP(hipMalloc(&d_sph, n * sizeof(double4)));
        LBVH_HIP(hipMemcpy(d_sph, spheres, n * sizeof(double4), hipMemcpyHostToDevice));
        if ((rc = lbvh_on_device(d_sph, n, clo, chi, d_nodes_out, d_order_out, depth_out)) != RT_OK) goto done;
        *d_sph_out = reinterpret_cast<double*>(d_sph);
        d_sph = nullptr;
    }
done:
    (void)hipFree(d_sph);
    return rc;
}

}  // namespace rt
