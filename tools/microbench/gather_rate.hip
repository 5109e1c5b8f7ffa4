// gather_rate.hip — how fast a CU can chase pointers through a table of 32-byte nodes (two 16-byte loads per lane and
// step, the access of the HBM-resident tree's descent), by table size, waves per CU and independent chains per lane.
//   hipcc --offload-arch=gfx950 -O2 -o gather_rate gather_rate.hip && ./gather_rate
// Every lane walks its own pseudo-random chain: the next index comes out of the data just loaded (a dependent gather,
// like a child link).  CHAINS independent walks per lane give CHAINS loads in flight per lane.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

struct Node { uint4 a, b; };  // 32 bytes; a.x = next index

template <int CHAINS, int LOADS>
__global__ void walk(const Node* __restrict__ nodes, unsigned mask, int steps, unsigned* sink, unsigned long long* cycles) {
    unsigned idx[CHAINS];
    const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) idx[c] = (g * 2654435761u + c * 0x9E3779B9u) & mask;
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            const uint4 a = nodes[idx[c]].a;
            if (LOADS == 2) { const uint4 b = nodes[idx[c]].b; acc += b.w; }
            idx[c] = a.x & mask;
            acc += a.y;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 0x12345u) *sink = acc;
    if ((threadIdx.x & 63) == 0) atomicAdd(cycles, t1 - t0);
}

// the same walk with only some lanes of every wave taking part: lanes [0, active) when spread == 0, else every
// (64 / active)-th lane — does a gather instruction cost by the lanes that are on?
__global__ void walk_masked(const Node* __restrict__ nodes, unsigned mask, int steps, int active, int spread, unsigned* sink,
                            unsigned long long* cycles) {
    const unsigned g = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    const bool on = spread ? (lane % (64 / active) == 0) : (int(lane) < active);
    unsigned idx = (g * 2654435761u) & mask, acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (on) {
        for (int s = 0; s < steps; ++s) {
            const uint4 a = nodes[idx].a;
            const uint4 b = nodes[idx].b;
            idx = a.x & mask;
            acc += a.y + b.w;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 0x12345u) *sink = acc;
    if (lane == 0) atomicAdd(cycles, t1 - t0);
}

template <int CHAINS, int LOADS>
void run(const Node* d_nodes, unsigned n_nodes, int threads_per_cu, int n_cu, unsigned* d_sink, unsigned long long* d_cyc, const char* label) {
    const int steps = 2000 / CHAINS;
    const int block = threads_per_cu > 1024 ? threads_per_cu / 2 : threads_per_cu, blocks = n_cu * (threads_per_cu / block);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((walk<CHAINS, LOADS>), dim3(blocks), dim3(block), 0, 0, d_nodes, n_nodes - 1, 50, d_sink, d_cyc);  // warm-up
    CHECK(hipMemset(d_cyc, 0, 8));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((walk<CHAINS, LOADS>), dim3(blocks), dim3(block), 0, 0, d_nodes, n_nodes - 1, steps, d_sink, d_cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long cyc = 0;
    CHECK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
    const double visits = double(blocks) * block * steps * CHAINS;
    const double waves = double(blocks) * block / 64;
    std::printf("%-22s %2d waves/CU  %d chain(s)  %d x16B: %7.1f G nodes/s  (%5.2f TB/s of node bytes)  %6.0f cycles per step per wave\n", label,
                threads_per_cu / 64, CHAINS, LOADS, visits / (ms * 1e-3) / 1e9, visits * 16.0 * LOADS / (ms * 1e-3) / 1e12,
                double(cyc) / waves / steps);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    unsigned* d_sink; unsigned long long* d_cyc;
    CHECK(hipMalloc(&d_sink, 4)); CHECK(hipMalloc(&d_cyc, 8));
    for (unsigned log2n : {10u, 17u, 20u, 23u}) {  // 32 KB (L1), 4 MB (one L2), 32 MB (the 1M-sphere tree), 256 MB
        const unsigned n = 1u << log2n;
        std::vector<Node> h(n);
        unsigned long long x = 88172645463325252ull;
        for (unsigned i = 0; i < n; ++i) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            h[i].a = make_uint4(unsigned(x >> 11), unsigned(x), i, 0);
            h[i].b = make_uint4(i, 1, 2, unsigned(x >> 40));
        }
        Node* d;
        CHECK(hipMalloc(&d, size_t(n) * sizeof(Node)));
        CHECK(hipMemcpy(d, h.data(), size_t(n) * sizeof(Node), hipMemcpyHostToDevice));
        char label[64];
        std::snprintf(label, sizeof label, "table %u KB", unsigned(size_t(n) * 32 / 1024));
        for (int tpc : {1024, 2048}) {
            run<1, 2>(d, n, tpc, n_cu, d_sink, d_cyc, label);
            run<2, 2>(d, n, tpc, n_cu, d_sink, d_cyc, label);
            run<4, 2>(d, n, tpc, n_cu, d_sink, d_cyc, label);
        }
        run<1, 1>(d, n, 1024, n_cu, d_sink, d_cyc, label);
        run<4, 1>(d, n, 1024, n_cu, d_sink, d_cyc, label);
        for (int spread = 0; spread < 2; ++spread)
            for (int active : {64, 32, 16, 8}) {
                const int steps = 2000;
                hipEvent_t e0, e1;
                CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
                CHECK(hipMemset(d_cyc, 0, 8));
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(walk_masked, dim3(n_cu), dim3(1024), 0, 0, d, n - 1, steps, active, spread, d_sink, d_cyc);
                CHECK(hipEventRecord(e1));
                CHECK(hipDeviceSynchronize());
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                unsigned long long cyc = 0;
                CHECK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
                std::printf("%-22s 16 waves/CU  %2d of 64 lanes on (%s): %7.1f G nodes/s  %6.0f cycles per step per wave\n", label, active,
                            spread ? "spread" : "packed", double(n_cu) * 16 * active * steps / (ms * 1e-3) / 1e9, double(cyc) / (n_cu * 16.0) / steps);
            }
        CHECK(hipFree(d));
    }
    return 0;
}
