// inner_step.hip — the instruction stream of the render kernel's inner-node loop (inner_loop_lds16 in
// raytracer_amd/csrc/rt_device.hip) replayed on a synthetic tree in LDS in which every lane always finds exactly one
// child, so the loop never leaves: SIMD cycles per step at 1, 2 and 4 waves per SIMD with all 64 lanes on.  Variants
// (second argument) change one thing each, to see what the stream's cost is made of:
//   0  the loop as the render kernel runs it
//   1  without the exit test (v_cmp / s_bcnt1 / s_cmp / s_cbranch)
//   2  without the push / pop bookkeeping
//   3  without the scalar mask logic: the child is chosen by one v_cmp + v_cndmask
//   4  the three compares first and the scalar logic spread between the vector instructions that follow
//   5  the two halves of a node side by side (one address computation, second read at offset 16) instead of in two arrays
//   hipcc --offload-arch=gfx950 -O2 -o inner_step inner_step.hip && ./inner_step
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int kNodes = 512;  // 2 x 8 KB of LDS, like a small scene
constexpr int kSteps = 4096;

#define SLAB                                                                                      \
    "v_fma_mix_f32 v112, v120, %[ix], %[ax] op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"                   \
    "v_fma_mix_f32 v113, v120, %[ix], %[ax] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"                   \
    "v_fma_mix_f32 v114, v121, |%[ix]|, %[ex] op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"                 \
    "v_fma_mix_f32 v115, v121, |%[ix]|, %[ex] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"                 \
    "v_pk_add_f32 v[116:117], v[112:113], v[114:115] neg_lo:[0,1] neg_hi:[0,1]\n\t"                 \
    "v_pk_add_f32 v[112:113], v[112:113], v[114:115]\n\t"                                           \
    "v_fma_mix_f32 v114, v122, %[iy], %[ay] op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"                   \
    "v_fma_mix_f32 v115, v122, %[iy], %[ay] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"                   \
    "v_fma_mix_f32 v120, v123, |%[iy]|, %[ey] op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"                 \
    "v_fma_mix_f32 v121, v123, |%[iy]|, %[ey] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"                 \
    "v_pk_add_f32 v[118:119], v[114:115], v[120:121] neg_lo:[0,1] neg_hi:[0,1]\n\t"                 \
    "v_pk_add_f32 v[114:115], v[114:115], v[120:121]\n\t"                                           \
    "v_fma_mix_f32 v120, v124, %[iz], %[az] op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"                   \
    "v_fma_mix_f32 v121, v124, %[iz], %[az] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"                   \
    "v_fma_mix_f32 v122, v125, |%[iz]|, %[ez] op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"                 \
    "v_fma_mix_f32 v123, v125, |%[iz]|, %[ez] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"                 \
    "v_pk_add_f32 v[124:125], v[120:121], v[122:123]\n\t"                                           \
    "v_pk_add_f32 v[122:123], v[120:121], v[122:123] neg_lo:[0,1] neg_hi:[0,1]\n\t"                 \
    "v_max_f32_e32 v122, 0, v122\n\t"                                                               \
    "v_max_f32_e32 v123, 0, v123\n\t"                                                               \
    "v_min_f32_e32 v124, v124, %[tmax]\n\t"                                                         \
    "v_min_f32_e32 v125, v125, %[tmax]\n\t"                                                         \
    "v_max3_f32 v116, v116, v118, v122\n\t"                                                         \
    "v_min3_f32 v112, v112, v114, v124\n\t"                                                         \
    "v_max3_f32 v117, v117, v119, v123\n\t"                                                         \
    "v_min3_f32 v113, v113, v115, v125\n\t"

#define SELECT_SCALAR                                                                             \
    "v_cmp_le_f32_e64 s[90:91], v116, v112\n\t"                                                     \
    "v_cmp_le_f32_e64 s[92:93], v117, v113\n\t"                                                     \
    "v_cmp_lt_f32_e64 s[94:95], v117, v116\n\t"                                                     \
    "s_and_b64 s[96:97], s[90:91], s[92:93]\n\t"                                                    \
    "s_or_b64 s[98:99], s[90:91], s[92:93]\n\t"                                                     \
    "s_and_b64 s[94:95], s[94:95], s[92:93]\n\t"                                                    \
    "s_andn2_b64 s[94:95], s[90:91], s[94:95]\n\t"                                                  \
    "s_andn2_b64 s[98:99], exec, s[98:99]\n\t"                                                      \
    "v_cndmask_b32_e64 %[node], v127, v126, s[94:95]\n\t"                                           \
    "v_cndmask_b32_e64 v118, v126, v127, s[94:95]\n\t"                                              \
    "v_cndmask_b32_e64 %[node], %[node], %[top], s[98:99]\n\t"

#define SELECT_VECTOR                                                                             \
    "v_cmp_le_f32_e32 vcc, v116, v112\n\t"                                                          \
    "v_cndmask_b32_e32 %[node], v127, v126, vcc\n\t"                                                \
    "s_mov_b64 s[96:97], 0\n\t"                                                                     \
    "s_mov_b64 s[98:99], 0\n\t"

#define EXIT_TEST                                                                                 \
    "v_cmp_lt_i32_e32 vcc, -1, %[node]\n\t"                                                         \
    "s_bcnt1_i32_b64 s87, vcc\n\t"                                                                  \
    "s_cmp_ge_i32 s87, %[thr]\n\t"                                                                  \
    "s_cbranch_scc0 5f\n\t"

#define FETCH                                                                                     \
    "s_mov_b64 exec, s[88:89]\n\t"                                                                  \
    "v_lshl_add_u32 v112, %[node], 4, %[q0]\n\t"                                                    \
    "v_lshl_add_u32 v113, %[node], 4, %[q1]\n\t"                                                    \
    "ds_read_b128 v[120:123], v112\n\t"                                                             \
    "ds_read_b128 v[124:127], v113\n\t"

#define FETCH_SIDE_BY_SIDE                                                                        \
    "s_mov_b64 exec, s[88:89]\n\t"                                                                  \
    "v_lshl_add_u32 v112, %[node], 5, %[q0]\n\t"                                                    \
    "ds_read_b128 v[120:123], v112\n\t"                                                             \
    "ds_read_b128 v[124:127], v112 offset:16\n\t"

#define PUSH_POP                                                                                  \
    "s_mov_b64 exec, s[96:97]\n\t"                                                                  \
    "ds_write_b32 %[sp], %[top]\n\t"                                                                \
    "v_add_u32_e32 %[sp], %[stride], %[sp]\n\t"                                                     \
    "v_mov_b32_e32 %[top], v118\n\t"                                                                \
    "s_mov_b64 exec, s[98:99]\n\t"                                                                  \
    "v_subrev_u32_e32 %[sp], %[stride], %[sp]\n\t"                                                  \
    "ds_read_b32 %[top], %[sp]\n\t"                                                                 \
    "s_mov_b64 exec, s[88:89]\n\t"

// the three compares, then scalar logic interleaved with the vector instructions that do not need it yet
#define SELECT_SPREAD                                                                             \
    "v_cmp_le_f32_e64 s[90:91], v116, v112\n\t"                                                     \
    "v_cmp_le_f32_e64 s[92:93], v117, v113\n\t"                                                     \
    "v_cmp_lt_f32_e64 s[94:95], v117, v116\n\t"                                                     \
    "s_and_b64 s[94:95], s[94:95], s[92:93]\n\t"                                                    \
    "s_or_b64 s[98:99], s[90:91], s[92:93]\n\t"                                                     \
    "s_andn2_b64 s[94:95], s[90:91], s[94:95]\n\t"                                                  \
    "s_and_b64 s[96:97], s[90:91], s[92:93]\n\t"                                                    \
    "v_cndmask_b32_e64 %[node], v127, v126, s[94:95]\n\t"                                           \
    "s_andn2_b64 s[98:99], exec, s[98:99]\n\t"                                                      \
    "v_cndmask_b32_e64 v118, v126, v127, s[94:95]\n\t"                                              \
    "v_cndmask_b32_e64 %[node], %[node], %[top], s[98:99]\n\t"

#define OPERANDS                                                                                                        \
    : [node] "+v"(node), [sp] "+v"(sp), [top] "+v"(top)                                                                 \
    : [ix] "v"(ix), [iy] "v"(iy), [iz] "v"(iz), [ax] "v"(ax), [ay] "v"(ay), [az] "v"(az), [ex] "v"(ex), [ey] "v"(ey),   \
      [ez] "v"(ez), [tmax] "v"(tmax), [q0] "s"(q0), [q1] "s"(q1), [thr] "s"(thr), [stride] "s"(stride), [n] "s"(steps)  \
    : "memory", "vcc", "scc", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122",   \
      "v123", "v124", "v125", "v126", "v127", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95",     \
      "s96", "s97", "s98", "s99"

#define HEAD "s_mov_b64 s[88:89], exec\n\ts_mov_b32 s86, %[n]\n\t" FETCH "1:\n\ts_waitcnt lgkmcnt(0)\n\t"
#define COUNT "s_sub_u32 s86, s86, 1\n\ts_cmp_lg_u32 s86, 0\n\ts_cbranch_scc1 1b\n\t5:\n\ts_mov_b64 exec, s[88:89]\n\ts_waitcnt lgkmcnt(0)\n\t"

template <int VARIANT>
__global__ void k_inner(unsigned long long* out, const uint4* tree, float seed) {
    extern __shared__ uint4 lds[];
    if (VARIANT == 5) {
        for (int i = threadIdx.x; i < kNodes; i += blockDim.x) { lds[2 * i] = tree[i]; lds[2 * i + 1] = tree[kNodes + i]; }
    } else {
        for (int i = threadIdx.x; i < 2 * kNodes; i += blockDim.x) lds[i] = tree[i];
    }
    __syncthreads();
    const uint32_t q0 = 0, q1 = kNodes * 16;
    const uint32_t stride = blockDim.x * 4;
    uint32_t sp = 2 * kNodes * 16 + threadIdx.x * 4 + stride;
    int32_t node = int32_t((threadIdx.x * 2654435761u) >> 23) & (kNodes - 1), top = 0x80000000;
    const float ix = 1.0f + seed * 0.001f * threadIdx.x, iy = -1.5f, iz = 0.75f, ax = 0.5f, ay = 0.25f, az = -0.5f;
    const float ex = 1e-3f, ey = 1e-3f, ez = 1e-3f, tmax = 3.0e38f;
    const int32_t thr = 1, steps = kSteps;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (VARIANT == 0) asm volatile(HEAD SLAB SELECT_SCALAR EXIT_TEST FETCH PUSH_POP COUNT OPERANDS);
    if (VARIANT == 1) asm volatile(HEAD SLAB SELECT_SCALAR FETCH PUSH_POP COUNT OPERANDS);
    if (VARIANT == 2) asm volatile(HEAD SLAB SELECT_SCALAR EXIT_TEST FETCH COUNT OPERANDS);
    if (VARIANT == 3) asm volatile(HEAD SLAB SELECT_VECTOR EXIT_TEST FETCH PUSH_POP COUNT OPERANDS);
    if (VARIANT == 4) asm volatile(HEAD SLAB SELECT_SPREAD EXIT_TEST FETCH PUSH_POP COUNT OPERANDS);
    if (VARIANT == 5)
        asm volatile("s_mov_b64 s[88:89], exec\n\ts_mov_b32 s86, %[n]\n\t" FETCH_SIDE_BY_SIDE "1:\n\ts_waitcnt lgkmcnt(0)\n\t" SLAB SELECT_SCALAR EXIT_TEST
                         FETCH_SIDE_BY_SIDE PUSH_POP COUNT OPERANDS);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (node == 0x12345678 && sp == 7u && top == 3) out[1] = 1;
    if ((threadIdx.x & 63) == 0) atomicAdd(out, t1 - t0);
}

static uint16_t half_bits(float f) {  // values used here are exactly representable
    _Float16 h = (_Float16)f;
    uint16_t b;
    __builtin_memcpy(&b, &h, 2);
    return b;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    // node i: child 0 = a box every ray meets (centre 0, half extent 30000), child 1 = an empty box (negative half
    // extent); links to pseudo-random inner nodes, never a leaf
    std::vector<uint32_t> tree(2 * kNodes * 4);
    const uint32_t c = uint32_t(half_bits(0.0f)) | (uint32_t(half_bits(0.0f)) << 16);
    const uint32_t h = uint32_t(half_bits(30000.0f)) | (uint32_t(half_bits(-30000.0f)) << 16);
    for (int i = 0; i < kNodes; ++i) {
        uint32_t* q0 = &tree[size_t(i) * 4];
        uint32_t* q1 = &tree[size_t(kNodes + i) * 4];
        q0[0] = c; q0[1] = h; q0[2] = c; q0[3] = h;
        q1[0] = c; q1[1] = h;
        q1[2] = (uint32_t(i) * 197u + 31u) & (kNodes - 1);
        q1[3] = (uint32_t(i) * 89u + 7u) & (kNodes - 1);
    }
    uint4* d_tree;
    unsigned long long* d_out;
    CHECK(hipMalloc(&d_tree, tree.size() * 4));
    CHECK(hipMemcpy(d_tree, tree.data(), tree.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_out, 16));
    const char* names[6] = {"loop as run", "no exit test", "no push/pop", "vector select", "scalar logic spread", "node halves side by side"};
    std::printf("%-22s %14s %14s %14s   (SIMD cycles per step of one wave, 64 lanes on)\n", "variant", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD");
    for (int v = 0; v < 6; ++v) {
        double cyc[3];
        int slot = 0;
        for (int threads : {256, 512, 1024}) {
            const size_t lds_bytes = size_t(2 * kNodes) * 16 + size_t(threads) * 4 * 8;
            CHECK(hipMemset(d_out, 0, 16));
            switch (v) {
                case 0: hipLaunchKernelGGL(k_inner<0>, dim3(n_cu), dim3(threads), lds_bytes, 0, d_out, d_tree, 1.5f); break;
                case 1: hipLaunchKernelGGL(k_inner<1>, dim3(n_cu), dim3(threads), lds_bytes, 0, d_out, d_tree, 1.5f); break;
                case 2: hipLaunchKernelGGL(k_inner<2>, dim3(n_cu), dim3(threads), lds_bytes, 0, d_out, d_tree, 1.5f); break;
                case 3: hipLaunchKernelGGL(k_inner<3>, dim3(n_cu), dim3(threads), lds_bytes, 0, d_out, d_tree, 1.5f); break;
                case 4: hipLaunchKernelGGL(k_inner<4>, dim3(n_cu), dim3(threads), lds_bytes, 0, d_out, d_tree, 1.5f); break;
                default: hipLaunchKernelGGL(k_inner<5>, dim3(n_cu), dim3(threads), lds_bytes, 0, d_out, d_tree, 1.5f); break;
            }
            CHECK(hipDeviceSynchronize());
            unsigned long long hh[2];
            CHECK(hipMemcpy(hh, d_out, 16, hipMemcpyDeviceToHost));
            const double waves = double(n_cu) * threads / 64.0;
            cyc[slot++] = double(hh[0]) / waves / kSteps / (threads / 256.0);
        }
        std::printf("%-22s %14.1f %14.1f %14.1f\n", names[v], cyc[0], cyc[1], cyc[2]);
    }
    CHECK(hipFree(d_tree));
    CHECK(hipFree(d_out));
    return 0;
}
