// instr_cost.hip — issue cost of single gfx950 vector instructions, measured the way the render kernel meets them:
// independent instructions of one kind, 1 or 4 waves per SIMD, cycles from s_memtime.
//   hipcc --offload-arch=gfx950 -O2 -o instr_cost instr_cost.hip && ./instr_cost
// Prints, per instruction: cycles per instruction seen by one wave, and SIMD cycles per wave-instruction
// (= the former divided by the waves sharing the SIMD), for 1 and 4 waves per SIMD.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int kIters = 256, kUnroll = 16;

#define REP16(X) X X X X X X X X X X X X X X X X

// 32-bit destination, three 32-bit sources
#define KERNEL_V32(NAME, TEXT)                                                                             \
    __global__ void NAME(unsigned long long* out, float seed) {                                             \
        float a = seed + threadIdx.x, b = seed * 0.5f, c = 1.0f + seed;                                     \
        float r0 = a, r1 = b, r2 = c, r3 = a + 1, r4 = b + 1, r5 = c + 1, r6 = a + 2, r7 = b + 2;           \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                               \
        for (int i = 0; i < kIters; ++i) {                                                                  \
            asm volatile(TEXT(%0) TEXT(%1) TEXT(%2) TEXT(%3) TEXT(%4) TEXT(%5) TEXT(%6) TEXT(%7)            \
                         TEXT(%0) TEXT(%1) TEXT(%2) TEXT(%3) TEXT(%4) TEXT(%5) TEXT(%6) TEXT(%7)            \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)   \
                         : "v"(b), "v"(c));                                                                 \
        }                                                                                                   \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                               \
        if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678f) out[1] = 1;                                \
        if ((threadIdx.x & 63) == 0) atomicAdd(out, t1 - t0);                                               \
    }
// 64-bit destination and sources
#define KERNEL_V64(NAME, TEXT)                                                                             \
    __global__ void NAME(unsigned long long* out, float seed) {                                             \
        double a = seed + threadIdx.x, b = seed * 0.5, c = 1.0 + seed;                                      \
        double r0 = a, r1 = b, r2 = c, r3 = a + 1, r4 = b + 1, r5 = c + 1, r6 = a + 2, r7 = b + 2;          \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                               \
        for (int i = 0; i < kIters; ++i) {                                                                  \
            asm volatile(TEXT(%0) TEXT(%1) TEXT(%2) TEXT(%3) TEXT(%4) TEXT(%5) TEXT(%6) TEXT(%7)            \
                         TEXT(%0) TEXT(%1) TEXT(%2) TEXT(%3) TEXT(%4) TEXT(%5) TEXT(%6) TEXT(%7)            \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)   \
                         : "v"(b), "v"(c));                                                                 \
        }                                                                                                   \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                               \
        if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678) out[1] = 1;                                 \
        if ((threadIdx.x & 63) == 0) atomicAdd(out, t1 - t0);                                               \
    }

#define T_FMA32(R) "v_fma_f32 " #R ", " #R ", %8, %9\n\t"
#define T_FMAMIX(R) "v_fma_mix_f32 " #R ", " #R ", %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
#define T_DOT2(R) "v_dot2_f32_f16 " #R ", " #R ", %8, %9 neg_hi:[0,1,0] clamp\n\t"
#define T_MIN32(R) "v_min_f32 " #R ", " #R ", %8\n\t"
#define T_MAX3(R) "v_max3_f32 " #R ", " #R ", %8, %9\n\t"
#define T_CNDMASK(R) "v_cndmask_b32 " #R ", " #R ", %8, vcc\n\t"
#define T_XOR(R) "v_xor_b32 " #R ", " #R ", %8\n\t"
#define T_MULHI(R) "v_mul_hi_u32 " #R ", " #R ", %8\n\t"
#define T_MULLO(R) "v_mul_lo_u32 " #R ", " #R ", %8\n\t"
#define T_MUL24(R) "v_mul_u32_u24 " #R ", " #R ", %8\n\t"
#define T_MAD24(R) "v_mad_u32_u24 " #R ", " #R ", %8, %9\n\t"
#define T_RCP(R) "v_rcp_f32 " #R ", " #R "\n\t"
#define T_SQRT(R) "v_sqrt_f32 " #R ", " #R "\n\t"
#define T_RSQ(R) "v_rsq_f32 " #R ", " #R "\n\t"
#define T_SIN(R) "v_sin_f32 " #R ", " #R "\n\t"
#define T_CVTU(R) "v_cvt_f32_u32 " #R ", " #R "\n\t"
#define T_LSHLADD(R) "v_lshl_add_u32 " #R ", " #R ", 4, %8\n\t"
#define T_FMA64(R) "v_fma_f64 " #R ", " #R ", %8, %9\n\t"
#define T_MUL64(R) "v_mul_f64 " #R ", " #R ", %8\n\t"
#define T_ADD64(R) "v_add_f64 " #R ", " #R ", %8\n\t"
#define T_RSQ64(R) "v_rsq_f64 " #R ", " #R "\n\t"
#define T_SQRT64(R) "v_sqrt_f64 " #R ", " #R "\n\t"
#define T_RCP64(R) "v_rcp_f64 " #R ", " #R "\n\t"
#define T_LDEXP64(R) "v_ldexp_f64 " #R ", " #R ", 3\n\t"
#define T_MAD64(R) "v_mad_u64_u32 " #R ", vcc, %8, %9, " #R "\n\t"
#define T_PKFMA(R) "v_pk_fma_f32 " #R ", " #R ", %8, %9\n\t"
#define T_PKADDC(R) "v_pk_add_f32 " #R ", " #R ", %8 neg_lo:[0,1] neg_hi:[0,1] clamp\n\t"
#define T_PKMUL(R) "v_pk_mul_f32 " #R ", " #R ", %8\n\t"
#define T_CMP64(R) "v_cmp_lt_f64 vcc, " #R ", %8\n\t"
#define T_CVT3264(R) "v_cvt_f64_f32 " #R ", %L8\n\t"

KERNEL_V32(k_fma32, T_FMA32)
KERNEL_V32(k_fmamix, T_FMAMIX)
KERNEL_V32(k_dot2, T_DOT2)
KERNEL_V32(k_min32, T_MIN32)
KERNEL_V32(k_max3, T_MAX3)
KERNEL_V32(k_cndmask, T_CNDMASK)
KERNEL_V32(k_xor, T_XOR)
KERNEL_V32(k_mulhi, T_MULHI)
KERNEL_V32(k_mullo, T_MULLO)
KERNEL_V32(k_mul24, T_MUL24)
KERNEL_V32(k_mad24, T_MAD24)
KERNEL_V32(k_rcp, T_RCP)
KERNEL_V32(k_sqrt, T_SQRT)
KERNEL_V32(k_rsq, T_RSQ)
KERNEL_V32(k_sin, T_SIN)
KERNEL_V32(k_cvtu, T_CVTU)
KERNEL_V32(k_lshladd, T_LSHLADD)
KERNEL_V64(k_fma64, T_FMA64)
KERNEL_V64(k_mul64, T_MUL64)
KERNEL_V64(k_add64, T_ADD64)
KERNEL_V64(k_rsq64, T_RSQ64)
KERNEL_V64(k_sqrt64, T_SQRT64)
KERNEL_V64(k_rcp64, T_RCP64)
KERNEL_V64(k_ldexp64, T_LDEXP64)
KERNEL_V64(k_pkfma, T_PKFMA)
KERNEL_V64(k_pkmul, T_PKMUL)
KERNEL_V64(k_pkaddc, T_PKADDC)
KERNEL_V64(k_cmp64, T_CMP64)

// the same instruction as ONE dependent chain (every instruction reads the result of the one before)
#define KERNEL_DEP32(NAME, TEXT)                                                                           \
    __global__ void NAME(unsigned long long* out, float seed) {                                             \
        float b = seed * 0.5f, c = 1.0f + seed, r0 = seed + threadIdx.x;                                    \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                               \
        for (int i = 0; i < kIters; ++i) {                                                                  \
            asm volatile(REP16(TEXT(%0)) : "+v"(r0) : "v"(b), "v"(c), "v"(c), "v"(c), "v"(c), "v"(c), "v"(c), "v"(b), "v"(c) : "vcc"); \
        }                                                                                                   \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                               \
        if (r0 == 12345.678f) out[1] = 1;                                                                   \
        if ((threadIdx.x & 63) == 0) atomicAdd(out, t1 - t0);                                               \
    }
#define KERNEL_DEP64(NAME, TEXT)                                                                           \
    __global__ void NAME(unsigned long long* out, float seed) {                                             \
        double b = seed * 0.5, c = 1.0 + seed, r0 = seed + threadIdx.x;                                     \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                               \
        for (int i = 0; i < kIters; ++i) {                                                                  \
            asm volatile(REP16(TEXT(%0)) : "+v"(r0) : "v"(b), "v"(c), "v"(c), "v"(c), "v"(c), "v"(c), "v"(c), "v"(b), "v"(c) : "vcc"); \
        }                                                                                                   \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                               \
        if (r0 == 12345.678) out[1] = 1;                                                                    \
        if ((threadIdx.x & 63) == 0) atomicAdd(out, t1 - t0);                                               \
    }
#define T_CND64S(R) "v_cndmask_b32_e64 " #R ", " #R ", %8, s[20:21]\n\t"
#define T_CMPCND(R) "v_cmp_lt_f32 vcc, " #R ", %8\n\tv_cndmask_b32 " #R ", " #R ", %9, vcc\n\t"
#define T_CMPSCND(R) "v_cmp_lt_f32_e64 s[20:21], " #R ", %8\n\tv_cndmask_b32_e64 " #R ", " #R ", %9, s[20:21]\n\t"
#define T_ADDU(R) "v_add_u32 " #R ", " #R ", %8\n\t"
#define T_CVT6432(R) "v_cvt_f32_f64 " #R ", %8\n\t"
KERNEL_DEP32(d_fma32, T_FMA32)
KERNEL_DEP32(d_fmamix, T_FMAMIX)
KERNEL_DEP32(d_min32, T_MIN32)
KERNEL_DEP32(d_xor, T_XOR)
KERNEL_DEP32(d_addu, T_ADDU)
KERNEL_DEP32(d_mulhi, T_MULHI)
KERNEL_DEP32(d_rcp, T_RCP)
KERNEL_DEP32(d_cnd_vcc, T_CNDMASK)
KERNEL_DEP32(d_cmp_cnd, T_CMPCND)
KERNEL_DEP64(d_fma64, T_FMA64)
KERNEL_DEP64(d_add64, T_ADD64)
KERNEL_DEP64(d_rsq64, T_RSQ64)
__global__ void k_cnd_sgpr(unsigned long long* out, float seed) {
    float b = seed * 0.5f, c = 1.0f + seed;
    float r0 = b, r1 = b, r2 = c, r3 = b + 1, r4 = b + 1, r5 = c + 1, r6 = b + 2, r7 = b + 2;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < kIters; ++i) {
        asm volatile("s_mov_b64 s[20:21], 0x5555\n\t" T_CND64S(%0) T_CND64S(%1) T_CND64S(%2) T_CND64S(%3) T_CND64S(%4) T_CND64S(%5) T_CND64S(%6) T_CND64S(%7)
                     T_CND64S(%0) T_CND64S(%1) T_CND64S(%2) T_CND64S(%3) T_CND64S(%4) T_CND64S(%5) T_CND64S(%6) T_CND64S(%7)
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                     : "v"(b), "v"(c)
                     : "s20", "s21");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678f) out[1] = 1;
    if ((threadIdx.x & 63) == 0) atomicAdd(out, t1 - t0);
}
__global__ void k_cmps_cnd(unsigned long long* out, float seed) {
    float b = seed * 0.5f, c = 1.0f + seed;
    float r0 = b, r1 = b, r2 = c, r3 = b + 1, r4 = b + 1, r5 = c + 1, r6 = b + 2, r7 = b + 2;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < kIters; ++i) {
        asm volatile(T_CMPSCND(%0) T_CMPSCND(%1) T_CMPSCND(%2) T_CMPSCND(%3) T_CMPSCND(%4) T_CMPSCND(%5) T_CMPSCND(%6) T_CMPSCND(%7)
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                     : "v"(b), "v"(c)
                     : "s20", "s21");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678f) out[1] = 1;
    if ((threadIdx.x & 63) == 0) atomicAdd(out, t1 - t0);
}
// salu: independent s_add / s_and pairs, and a v_cmp -> s_and -> s_cbranch ladder like the descent loop's exit test
__global__ void k_salu(unsigned long long* out, float seed) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < kIters; ++i) {
        asm volatile(REP16("s_add_u32 s20, s20, 1\n\ts_and_b64 s[22:23], s[22:23], exec\n\t") ::: "s20", "s22", "s23", "scc");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (seed == 12345.0f) out[1] = 1;
    if ((threadIdx.x & 63) == 0) atomicAdd(out, (t1 - t0) / 2);
}
__global__ void k_valu_salu_mix(unsigned long long* out, float seed) {
    float r0 = seed, r1 = seed + 1, b = seed * 0.5f, c = seed + 2;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < kIters; ++i) {
        asm volatile(REP16("v_fma_f32 %0, %0, %2, %3\n\ts_add_u32 s20, s20, 1\n\tv_fma_f32 %1, %1, %2, %3\n\ts_and_b64 s[22:23], s[22:23], exec\n\t")
                     : "+v"(r0), "+v"(r1) : "v"(b), "v"(c) : "s20", "s22", "s23", "scc");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (r0 + r1 == 12345.0f) out[1] = 1;
    if ((threadIdx.x & 63) == 0) atomicAdd(out, (t1 - t0) / 4);
}

// v_mad_u64_u32: 64-bit destination, 32-bit sources
__global__ void k_mad64(unsigned long long* out, float seed) {
    unsigned long long r0 = threadIdx.x, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7;
    unsigned int b = (unsigned int)(seed * 1000.0f) | 1u, c = 0xD2511F53u;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < kIters; ++i) {
        asm volatile(T_MAD64(%0) T_MAD64(%1) T_MAD64(%2) T_MAD64(%3) T_MAD64(%4) T_MAD64(%5) T_MAD64(%6) T_MAD64(%7)
                     T_MAD64(%0) T_MAD64(%1) T_MAD64(%2) T_MAD64(%3) T_MAD64(%4) T_MAD64(%5) T_MAD64(%6) T_MAD64(%7)
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                     : "v"(b), "v"(c)
                     : "vcc");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345) out[1] = 1;
    if ((threadIdx.x & 63) == 0) atomicAdd(out, t1 - t0);
}

// ds_read_b128 at random 16-byte slots (the node fetch of the LDS-resident tree): dependent chain = latency, independent = rate
__global__ void k_lds_random(unsigned long long* out, float seed, int dependent) {
    __shared__ uint4 nodes[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) nodes[i] = make_uint4(i * 2654435761u, i * 40503u, i, i ^ 0x5bd1e995u);
    __syncthreads();
    unsigned int idx = (threadIdx.x * 2654435761u + (unsigned int)seed) & 2047u;
    unsigned int acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (dependent) {
        for (int i = 0; i < kIters * 4; ++i) { uint4 v = nodes[idx]; idx = v.x & 2047u; acc += v.y; }
    } else {
        for (int i = 0; i < kIters; ++i) {
            uint4 a = nodes[idx], b = nodes[(idx + 977u) & 2047u], c = nodes[(idx + 1311u) & 2047u], d = nodes[(idx + 1999u) & 2047u];
            acc += a.y + b.y + c.y + d.y;
            idx = (idx * 5u + 1u) & 2047u;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 0x12345678u) out[1] = 1;
    if ((threadIdx.x & 63) == 0) atomicAdd(out, t1 - t0);
}

template <class F>
double run(F launch, int threads, unsigned long long* d_out, int n_cu) {
    CHECK(hipMemset(d_out, 0, 16));
    launch(dim3(n_cu), dim3(threads));
    CHECK(hipDeviceSynchronize());
    unsigned long long h[2];
    CHECK(hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost));
    const double waves = double(n_cu) * threads / 64.0;
    return double(h[0]) / waves;  // cycles per wave
}

int main() {
    int n_cu = 0;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    n_cu = prop.multiProcessorCount;
    unsigned long long* d_out;
    CHECK(hipMalloc(&d_out, 16));
    std::printf("%-16s %12s %12s %12s %12s\n", "instruction", "1w cyc/inst", "1w SIMD cyc", "4w cyc/inst", "4w SIMD cyc");
    const double n_inst = double(kIters) * kUnroll;
#define ROW(NAME, K)                                                                                                   \
    {                                                                                                                  \
        double a = run([&](dim3 g, dim3 b) { hipLaunchKernelGGL(K, g, b, 0, 0, d_out, 1.5f); }, 256, d_out, n_cu) / n_inst;   \
        double c = run([&](dim3 g, dim3 b) { hipLaunchKernelGGL(K, g, b, 0, 0, d_out, 1.5f); }, 1024, d_out, n_cu) / n_inst;  \
        std::printf("%-16s %12.2f %12.2f %12.2f %12.2f\n", NAME, a, a, c, c / 4.0);                                    \
    }
    ROW("v_fma_f32", k_fma32)
    ROW("v_fma_mix_f32", k_fmamix)
    ROW("v_dot2_f32_f16", k_dot2)
    ROW("v_min_f32", k_min32)
    ROW("v_max3_f32", k_max3)
    ROW("v_cndmask_b32", k_cndmask)
    ROW("v_xor_b32", k_xor)
    ROW("v_lshl_add_u32", k_lshladd)
    ROW("v_cvt_f32_u32", k_cvtu)
    ROW("v_mul_hi_u32", k_mulhi)
    ROW("v_mul_lo_u32", k_mullo)
    ROW("v_mul_u32_u24", k_mul24)
    ROW("v_mad_u32_u24", k_mad24)
    ROW("v_mad_u64_u32", k_mad64)
    ROW("v_rcp_f32", k_rcp)
    ROW("v_sqrt_f32", k_sqrt)
    ROW("v_rsq_f32", k_rsq)
    ROW("v_sin_f32", k_sin)
    ROW("v_pk_fma_f32", k_pkfma)
    ROW("v_pk_mul_f32", k_pkmul)
    ROW("v_pk_add_f32 clamp", k_pkaddc)
    ROW("v_fma_f64", k_fma64)
    ROW("v_mul_f64", k_mul64)
    ROW("v_add_f64", k_add64)
    ROW("v_cmp_lt_f64", k_cmp64)
    ROW("v_ldexp_f64", k_ldexp64)
    ROW("v_rsq_f64", k_rsq64)
    ROW("v_sqrt_f64", k_sqrt64)
    ROW("v_rcp_f64", k_rcp64)
    std::printf("-- one dependent chain per wave\n");
    ROW("dep v_fma_f32", d_fma32)
    ROW("dep v_fma_mix", d_fmamix)
    ROW("dep v_min_f32", d_min32)
    ROW("dep v_xor_b32", d_xor)
    ROW("dep v_add_u32", d_addu)
    ROW("dep v_mul_hi_u32", d_mulhi)
    ROW("dep v_rcp_f32", d_rcp)
    ROW("dep cndmask vcc", d_cnd_vcc)
    ROW("dep cmp+cnd (x2)", d_cmp_cnd)
    ROW("dep v_fma_f64", d_fma64)
    ROW("dep v_add_f64", d_add64)
    ROW("dep v_rsq_f64", d_rsq64)
    std::printf("-- others\n");
    ROW("cndmask sgpr", k_cnd_sgpr)
    ROW("cmp_e64+cnd (x2)", k_cmps_cnd)
    ROW("salu", k_salu)
    ROW("valu+salu mix", k_valu_salu_mix)
    for (int dep = 1; dep >= 0; --dep)
        for (int threads : {256, 512, 1024}) {
            double cyc = run([&](dim3 g, dim3 b) { hipLaunchKernelGGL(k_lds_random, g, b, 0, 0, d_out, 3.0f, dep); }, threads, d_out, n_cu);
            std::printf("ds_read_b128 random, %s, %d waves/CU: %.1f cycles per read per wave\n", dep ? "dependent chain" : "4 in flight",
                        threads / 64, cyc / (kIters * 4.0));
        }
    CHECK(hipFree(d_out));
    return 0;
}
