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
//   6  the ray parameter scaled so that the range of interest is [0, 1]: the clamp modifier of the z additions does the
//      work of the four v_max / v_min instructions
//   7  each (centre, half extent) pair of a child and axis in one dword, near and far parameter each by ONE
//      v_dot2_f32_f16 against the packed pair (step / d, |step / d|) as halves; clamp as in 6: 12 + 4 + 3 arithmetic
//      instructions per step instead of 18 + 8 + 3
//   8  a 4-child node (four 16-byte reads: 4 x 3 (centre, half extent) dwords + four 16-bit links) with the arithmetic
//      of 7: nearest hit child next, the other hit children pushed in storage order
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

#define SLAB_NEAR                                                                                      \
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
    "v_pk_add_f32 v[122:123], v[120:121], v[122:123] neg_lo:[0,1] neg_hi:[0,1] clamp\n\t"           \
    "v_min_f32_e32 v124, v124, %[tmax]\n\t"                                                         \
    "v_min_f32_e32 v125, v125, %[tmax]\n\t"                                                         \
    "v_max3_f32 v116, v116, v118, v122\n\t"                                                         \
    "v_min3_f32 v112, v112, v114, v124\n\t"                                                         \
    "v_max3_f32 v117, v117, v119, v123\n\t"                                                         \
    "v_min3_f32 v113, v113, v115, v125\n\t"


// variants 9 / 10: the step after the restructuring.  The masks are used where they are made (s_and / s_andn2 write
// EXEC directly), the stack is brought up to date BEFORE the next fetch (9) or after it (10, which has to keep the old
// EXEC in s[84:85]), the second half of a node is read at a constant offset from the first (one address computation),
// and the loop's back edge is the conditional branch of the exit test.
#define SELECT_ANY                                                                                \
    "v_cmp_le_f32_e64 s[90:91], v116, v112\n\t"                                                     \
    "v_cmp_le_f32_e64 s[92:93], v117, v113\n\t"                                                     \
    "v_cmp_lt_f32_e64 s[94:95], v117, v116\n\t"                                                     \
    "s_or_b64 s[98:99], s[90:91], s[92:93]\n\t"                                                     \
    "s_and_b64 s[94:95], s[94:95], s[92:93]\n\t"                                                    \
    "s_andn2_b64 s[94:95], s[90:91], s[94:95]\n\t"                                                  \
    "v_cndmask_b32_e64 v119, v127, v126, s[94:95]\n\t"                                              \
    "v_cndmask_b32_e64 v118, v126, v127, s[94:95]\n\t"                                              \
    "v_cndmask_b32_e64 %[node], %[top], v119, s[98:99]\n\t"
#define STACK_FIRST                                                                               \
    "s_andn2_b64 exec, exec, s[98:99]\n\t"                                                          \
    "v_subrev_u32_e32 %[sp], %[stride], %[sp]\n\t"                                                  \
    "ds_read_b32 %[top], %[sp]\n\t"                                                                 \
    "s_and_b64 exec, s[90:91], s[92:93]\n\t"                                                        \
    "ds_write_b32 %[sp], %[top]\n\t"                                                                \
    "v_add_u32_e32 %[sp], %[stride], %[sp]\n\t"                                                     \
    "v_mov_b32_e32 %[top], v118\n\t"
#define TEST_FETCH_LOOP                                                                           \
    "s_mov_b64 exec, s[88:89]\n\t"                                                                  \
    "v_cmp_lt_i32_e32 vcc, -1, %[node]\n\t"                                                         \
    "s_bcnt1_i32_b64 s87, vcc\n\t"                                                                  \
    "s_mov_b64 exec, vcc\n\t"                                                                       \
    "v_lshl_add_u32 v112, %[node], 4, %[q0]\n\t"                                                    \
    "ds_read_b128 v[120:123], v112\n\t"                                                             \
    "ds_read_b128 v[124:127], v112 offset:8192\n\t"                                                 \
    "s_sub_u32 s86, s86, 1\n\t"                                                                     \
    "s_cmp_ge_i32 s87, %[thr]\n\t"                                                                  \
    "s_cbranch_scc0 5f\n\t"                                                                         \
    "s_cmp_lg_u32 s86, 0\n\t"                                                                       \
    "s_cbranch_scc1 1b\n\t"                                                                         \
    "5:\n\ts_mov_b64 exec, s[88:89]\n\ts_waitcnt lgkmcnt(0)\n\t"
// variant 6: z additions with clamp (parameters normalised to [0, 1]), no v_max / v_min
#define SLAB_CLAMP                                                                                \
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
    "v_pk_add_f32 v[124:125], v[120:121], v[122:123] clamp\n\t"                                     \
    "v_pk_add_f32 v[122:123], v[120:121], v[122:123] neg_lo:[0,1] neg_hi:[0,1] clamp\n\t"           \
    "v_max3_f32 v116, v116, v118, v122\n\t"                                                         \
    "v_min3_f32 v112, v112, v114, v124\n\t"                                                         \
    "v_max3_f32 v117, v117, v119, v123\n\t"                                                         \
    "v_min3_f32 v113, v113, v115, v125\n\t"

// variant 7: v120 / v121 = (centre, half extent) of child 0 / child 1 on x, v122 / v123 on y, v124 / v125 on z;
// %[ix] = halves (step / d, |step / d|), %[ax] = A - E (near addend), %[ex] = A + E (far addend)
#define SLAB_DOT2                                                                                 \
    "v_dot2_f32_f16 v116, v120, %[ix], %[ax] neg_hi:[0,1,0]\n\t"                                    \
    "v_dot2_f32_f16 v112, v120, %[ix], %[ex]\n\t"                                                   \
    "v_dot2_f32_f16 v117, v121, %[ix], %[ax] neg_hi:[0,1,0]\n\t"                                    \
    "v_dot2_f32_f16 v113, v121, %[ix], %[ex]\n\t"                                                   \
    "v_dot2_f32_f16 v118, v122, %[iy], %[ay] neg_hi:[0,1,0]\n\t"                                    \
    "v_dot2_f32_f16 v114, v122, %[iy], %[ey]\n\t"                                                   \
    "v_dot2_f32_f16 v119, v123, %[iy], %[ay] neg_hi:[0,1,0]\n\t"                                    \
    "v_dot2_f32_f16 v115, v123, %[iy], %[ey]\n\t"                                                   \
    "v_dot2_f32_f16 v122, v124, %[iz], %[az] neg_hi:[0,1,0] clamp\n\t"                              \
    "v_dot2_f32_f16 v120, v124, %[iz], %[ez] clamp\n\t"                                             \
    "v_dot2_f32_f16 v123, v125, %[iz], %[az] neg_hi:[0,1,0] clamp\n\t"                              \
    "v_dot2_f32_f16 v121, v125, %[iz], %[ez] clamp\n\t"                                             \
    "v_max3_f32 v116, v116, v118, v122\n\t"                                                         \
    "v_min3_f32 v112, v112, v114, v120\n\t"                                                         \
    "v_max3_f32 v117, v117, v119, v123\n\t"                                                         \
    "v_min3_f32 v113, v113, v115, v121\n\t"

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


// variant 8: the 4-child node.  v96-v99 = x (centre, half extent) of children 0-3, v100-v103 = y, v104-v107 = z,
// v108 = link 0 | link 1 << 16, v109 = link 2 | link 3 << 16 (16-bit signed links).  %[q0] .. = the four arrays.
#define FETCH4                                                                                    \
    "s_mov_b64 exec, s[88:89]\n\t"                                                                  \
    "v_lshl_add_u32 v112, %[node], 4, %[q0]\n\t"                                                    \
    "v_lshl_add_u32 v113, %[node], 4, %[q1]\n\t"                                                    \
    "v_lshl_add_u32 v114, %[node], 4, %[q2]\n\t"                                                    \
    "v_lshl_add_u32 v115, %[node], 4, %[q3]\n\t"                                                    \
    "ds_read_b128 v[96:99], v112\n\t"                                                               \
    "ds_read_b128 v[100:103], v113\n\t"                                                             \
    "ds_read_b128 v[104:107], v114\n\t"                                                             \
    "ds_read_b64 v[108:109], v115\n\t"
#define DOT4(N0, F0, SRC, I, A, E, MOD)                                                           \
    "v_dot2_f32_f16 v" #N0 ", v" #SRC ", " I ", " A " neg_hi:[0,1,0]" MOD "\n\t"                     \
    "v_dot2_f32_f16 v" #F0 ", v" #SRC ", " I ", " E MOD "\n\t"
#define SLAB4                                                                                     \
    DOT4(112, 116, 96, "%[ix]", "%[ax]", "%[ex]", "") DOT4(113, 117, 97, "%[ix]", "%[ax]", "%[ex]", "")          \
    DOT4(114, 118, 98, "%[ix]", "%[ax]", "%[ex]", "") DOT4(115, 119, 99, "%[ix]", "%[ax]", "%[ex]", "")          \
    DOT4(120, 124, 100, "%[iy]", "%[ay]", "%[ey]", "") DOT4(121, 125, 101, "%[iy]", "%[ay]", "%[ey]", "")        \
    DOT4(122, 126, 102, "%[iy]", "%[ay]", "%[ey]", "") DOT4(123, 127, 103, "%[iy]", "%[ay]", "%[ey]", "")        \
    DOT4(96, 100, 104, "%[iz]", "%[az]", "%[ez]", " clamp") DOT4(97, 101, 105, "%[iz]", "%[az]", "%[ez]", " clamp")  \
    DOT4(98, 102, 106, "%[iz]", "%[az]", "%[ez]", " clamp") DOT4(99, 103, 107, "%[iz]", "%[az]", "%[ez]", " clamp")  \
    "v_max3_f32 v112, v112, v120, v96\n\t"                                                          \
    "v_max3_f32 v113, v113, v121, v97\n\t"                                                          \
    "v_max3_f32 v114, v114, v122, v98\n\t"                                                          \
    "v_max3_f32 v115, v115, v123, v99\n\t"                                                          \
    "v_min3_f32 v116, v116, v124, v100\n\t"                                                         \
    "v_min3_f32 v117, v117, v125, v101\n\t"                                                         \
    "v_min3_f32 v118, v118, v126, v102\n\t"                                                         \
    "v_min3_f32 v119, v119, v127, v103\n\t"
// hit masks s[70:77], entry parameters of the hit children (others +inf) v120-v123, the smallest v124, "is the
// nearest" masks s[78:85] (first of equals), links v96-v99, next node, masks of the children to push in s[70:77]
#define SELECT4                                                                                   \
    "v_cmp_lt_f32_e64 s[70:71], v112, v116\n\t"                                                     \
    "v_cmp_lt_f32_e64 s[72:73], v113, v117\n\t"                                                     \
    "v_cmp_lt_f32_e64 s[74:75], v114, v118\n\t"                                                     \
    "v_cmp_lt_f32_e64 s[76:77], v115, v119\n\t"                                                     \
    "v_cndmask_b32_e64 v120, %[inf], v112, s[70:71]\n\t"                                            \
    "v_cndmask_b32_e64 v121, %[inf], v113, s[72:73]\n\t"                                            \
    "v_cndmask_b32_e64 v122, %[inf], v114, s[74:75]\n\t"                                            \
    "v_cndmask_b32_e64 v123, %[inf], v115, s[76:77]\n\t"                                            \
    "v_min3_f32 v124, v120, v121, v122\n\t"                                                         \
    "v_min_f32_e32 v124, v124, v123\n\t"                                                            \
    "v_bfe_i32 v96, v108, 0, 16\n\t"                                                                \
    "v_ashrrev_i32_e32 v97, 16, v108\n\t"                                                           \
    "v_bfe_i32 v98, v109, 0, 16\n\t"                                                                \
    "v_ashrrev_i32_e32 v99, 16, v109\n\t"                                                           \
    "v_cmp_eq_f32_e64 s[78:79], v120, v124\n\t"                                                     \
    "v_cmp_eq_f32_e64 s[80:81], v121, v124\n\t"                                                     \
    "v_cmp_eq_f32_e64 s[82:83], v122, v124\n\t"                                                     \
    "s_and_b64 s[78:79], s[78:79], s[70:71]\n\t"                                                    \
    "s_andn2_b64 s[80:81], s[80:81], s[78:79]\n\t"                                                  \
    "s_and_b64 s[80:81], s[80:81], s[72:73]\n\t"                                                    \
    "s_or_b64 s[84:85], s[78:79], s[80:81]\n\t"                                                     \
    "s_andn2_b64 s[82:83], s[82:83], s[84:85]\n\t"                                                  \
    "s_and_b64 s[82:83], s[82:83], s[74:75]\n\t"                                                    \
    "s_or_b64 s[84:85], s[84:85], s[82:83]\n\t"                                                     \
    "v_cndmask_b32_e64 %[node], v99, v98, s[82:83]\n\t"                                             \
    "v_cndmask_b32_e64 %[node], %[node], v97, s[80:81]\n\t"                                         \
    "v_cndmask_b32_e64 %[node], %[node], v96, s[78:79]\n\t"                                         \
    "s_or_b64 s[84:85], s[84:85], s[76:77]\n\t"                                                     \
    "s_andn2_b64 s[98:99], exec, s[84:85]\n\t"                                                      \
    "v_cndmask_b32_e64 %[node], %[node], %[top], s[98:99]\n\t"                                      \
    "s_andn2_b64 s[70:71], s[70:71], s[78:79]\n\t"                                                  \
    "s_andn2_b64 s[72:73], s[72:73], s[80:81]\n\t"                                                  \
    "s_andn2_b64 s[74:75], s[74:75], s[82:83]\n\t"                                                  \
    "s_or_b64 s[84:85], s[78:79], s[80:81]\n\t"                                                     \
    "s_or_b64 s[84:85], s[84:85], s[82:83]\n\t"                                                     \
    "s_and_b64 s[76:77], s[76:77], s[84:85]\n\t"
#define PUSH1(MASK, LINK)                                                                         \
    "s_mov_b64 exec, " MASK "\n\t"                                                                  \
    "ds_write_b32 %[sp], %[top]\n\t"                                                                \
    "v_add_u32_e32 %[sp], %[stride], %[sp]\n\t"                                                     \
    "v_mov_b32_e32 %[top], " LINK "\n\t"
#define PUSH_POP4                                                                                 \
    PUSH1("s[76:77]", "v99") PUSH1("s[74:75]", "v98") PUSH1("s[72:73]", "v97") PUSH1("s[70:71]", "v96")  \
    "s_mov_b64 exec, s[98:99]\n\t"                                                                  \
    "v_subrev_u32_e32 %[sp], %[stride], %[sp]\n\t"                                                  \
    "ds_read_b32 %[top], %[sp]\n\t"                                                                 \
    "s_mov_b64 exec, s[88:89]\n\t"
// the pushes of variant 8 need the links (v96-v99) after the next node's reads were issued into the same registers,
// so its fetch comes after the stack bookkeeping
#define OPERANDS4                                                                                                       \
    : [node] "+v"(node), [sp] "+v"(sp), [top] "+v"(top)                                                                 \
    : [ix] "v"(ix), [iy] "v"(iy), [iz] "v"(iz), [ax] "v"(ax), [ay] "v"(ay), [az] "v"(az), [ex] "v"(ex), [ey] "v"(ey),   \
      [ez] "v"(ez), [inf] "v"(inf), [q0] "s"(q0), [q1] "s"(q1), [q2] "s"(q2), [q3] "s"(q3), [thr] "s"(thr),             \
      [stride] "s"(stride), [n] "s"(steps)                                                                              \
    : "memory", "vcc", "scc", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106",       \
      "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120",   \
      "v121", "v122", "v123", "v124", "v125", "v126", "v127", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77",   \
      "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s98", "s99"
#define HEAD4 "s_mov_b64 s[88:89], exec\n\ts_mov_b32 s86, %[n]\n\t" FETCH4 "1:\n\ts_waitcnt lgkmcnt(0)\n\t"

#define OPERANDS                                                                                                        \
    : [node] "+v"(node), [sp] "+v"(sp), [top] "+v"(top)                                                                 \
    : [ix] "v"(ix), [iy] "v"(iy), [iz] "v"(iz), [ax] "v"(ax), [ay] "v"(ay), [az] "v"(az), [ex] "v"(ex), [ey] "v"(ey),   \
      [ez] "v"(ez), [tmax] "v"(tmax), [q0] "s"(q0), [q1] "s"(q1), [thr] "s"(thr), [stride] "s"(stride), [n] "s"(steps)  \
    : "memory", "vcc", "scc", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122",   \
      "v123", "v124", "v125", "v126", "v127", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95",     \
      "s96", "s97", "s98", "s99"

#define HEAD "s_mov_b64 s[88:89], exec\n\ts_mov_b32 s86, %[n]\n\t" FETCH "1:\n\ts_waitcnt lgkmcnt(0)\n\t"
#define COUNT "s_sub_u32 s86, s86, 1\n\ts_cmp_lg_u32 s86, 0\n\ts_cbranch_scc1 1b\n\t5:\n\ts_mov_b64 exec, s[88:89]\n\ts_waitcnt lgkmcnt(0)\n\t"

__device__ __forceinline__ float pack_halves(float lo, float hi) {
    typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
    const half2_t h = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(float, h);
}

template <int VARIANT>
__global__ void k_inner(unsigned long long* out, const uint4* tree, float seed, int coherent) {
    extern __shared__ uint4 lds[];
    constexpr int kArrays = VARIANT == 8 ? 4 : 2;
    if (VARIANT == 5) {
        for (int i = threadIdx.x; i < kNodes; i += blockDim.x) { lds[2 * i] = tree[i]; lds[2 * i + 1] = tree[kNodes + i]; }
    } else {
        for (int i = threadIdx.x; i < kArrays * kNodes; i += blockDim.x) lds[i] = tree[i];
    }
    __syncthreads();
    const uint32_t q0 = 0, q1 = kNodes * 16, q2 = 2 * kNodes * 16, q3 = 3 * kNodes * 16;
    const uint32_t stride = blockDim.x * 4;
    uint32_t sp = kArrays * kNodes * 16 + threadIdx.x * 4 + stride;
    // coherent: the lanes of a wave walk the same nodes (reads are broadcasts, no bank conflicts): the issue cost alone
    int32_t node = int32_t(((coherent ? threadIdx.x >> 6 : threadIdx.x) * 2654435761u) >> 23) & (kNodes - 1), top = 0x80000000;
    const float ixf = 1.0f + seed * 0.001f * threadIdx.x, iyf = -1.5f, izf = 0.75f;
    const bool packed = VARIANT >= 7;  // (step / d, |step / d|) as two halves
    const float ix = packed ? pack_halves(ixf, fabsf(ixf)) : ixf, iy = packed ? pack_halves(iyf, fabsf(iyf)) : iyf,
                iz = packed ? pack_halves(izf, fabsf(izf)) : izf;
    const float ax = 0.5f, ay = 0.25f, az = -0.5f;
    const float ex = packed ? 0.501f : 1e-3f, ey = packed ? 0.251f : 1e-3f, ez = packed ? -0.499f : 1e-3f, tmax = 3.0e38f;
    const float inf = __builtin_inff();
    const int32_t thr = 1, steps = kSteps;
    (void)q2; (void)q3; (void)inf; (void)tmax;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (VARIANT == 0) asm volatile(HEAD SLAB SELECT_SCALAR EXIT_TEST FETCH PUSH_POP COUNT OPERANDS);
    if (VARIANT == 1) asm volatile(HEAD SLAB SELECT_SCALAR FETCH PUSH_POP COUNT OPERANDS);
    if (VARIANT == 2) asm volatile(HEAD SLAB SELECT_SCALAR EXIT_TEST FETCH COUNT OPERANDS);
    if (VARIANT == 3) asm volatile(HEAD SLAB SELECT_VECTOR EXIT_TEST FETCH PUSH_POP COUNT OPERANDS);
    if (VARIANT == 4) asm volatile(HEAD SLAB SELECT_SPREAD EXIT_TEST FETCH PUSH_POP COUNT OPERANDS);
    if (VARIANT == 6) asm volatile(HEAD SLAB_CLAMP SELECT_SCALAR EXIT_TEST FETCH PUSH_POP COUNT OPERANDS);
    if (VARIANT == 7) asm volatile(HEAD SLAB_DOT2 SELECT_SCALAR EXIT_TEST FETCH PUSH_POP COUNT OPERANDS);
    if (VARIANT == 8) asm volatile(HEAD4 SLAB4 SELECT4 EXIT_TEST PUSH_POP4 FETCH4 COUNT OPERANDS4);
    if (VARIANT == 9) asm volatile(HEAD SLAB_NEAR SELECT_SCALAR EXIT_TEST FETCH PUSH_POP COUNT OPERANDS);
    if (VARIANT == 10)
        asm volatile("s_mov_b64 s[88:89], exec\n\ts_mov_b32 s86, %[n]\n\tv_lshl_add_u32 v112, %[node], 4, %[q0]\n\tds_read_b128 v[120:123], v112\n\t"
                     "ds_read_b128 v[124:127], v112 offset:8192\n\t1:\n\ts_waitcnt lgkmcnt(0)\n\t" SLAB_NEAR SELECT_ANY STACK_FIRST TEST_FETCH_LOOP OPERANDS);
    if (VARIANT == 5)
        asm volatile("s_mov_b64 s[88:89], exec\n\ts_mov_b32 s86, %[n]\n\t" FETCH_SIDE_BY_SIDE "1:\n\ts_waitcnt lgkmcnt(0)\n\t" SLAB SELECT_SCALAR EXIT_TEST
                         FETCH_SIDE_BY_SIDE PUSH_POP COUNT OPERANDS);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (node == 0x12345678 && sp == 7u && top == 3) out[1] = 1;
    if ((threadIdx.x & 63) == 0) atomicAdd(out, t1 - t0);
    if (seed == -1.0f) out[2 + threadIdx.x] = (unsigned long long)node;  // keeps the loop's result alive
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
    // Layout A (variants 0-6): node i = two 16-byte halves; child 0 = a box every ray meets (centre 0, half extent
    // 30000), child 1 = an empty box (negative half extent); links to pseudo-random inner nodes, never a leaf.
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
    // Layout B (variant 7): one dword per child and axis = (centre, half extent); same two boxes.
    const uint32_t hit = uint32_t(half_bits(0.0f)) | (uint32_t(half_bits(30000.0f)) << 16);
    const uint32_t miss = uint32_t(half_bits(0.0f)) | (uint32_t(half_bits(-30000.0f)) << 16);
    std::vector<uint32_t> tree_b(2 * kNodes * 4);
    for (int i = 0; i < kNodes; ++i) {
        uint32_t* q0 = &tree_b[size_t(i) * 4];
        uint32_t* q1 = &tree_b[size_t(kNodes + i) * 4];
        q0[0] = hit; q0[1] = miss; q0[2] = hit; q0[3] = miss;
        q1[0] = hit; q1[1] = miss;
        q1[2] = (uint32_t(i) * 197u + 31u) & (kNodes - 1);
        q1[3] = (uint32_t(i) * 89u + 7u) & (kNodes - 1);
    }
    // Layout C (variant 8): four children, child 0 the box every ray meets, the other three empty; 16-bit links.
    std::vector<uint32_t> tree_c(4 * kNodes * 4);
    for (int i = 0; i < kNodes; ++i) {
        for (int a = 0; a < 3; ++a) {
            uint32_t* q = &tree_c[(size_t(a) * kNodes + i) * 4];
            q[0] = hit; q[1] = miss; q[2] = miss; q[3] = miss;
        }
        uint32_t* q3 = &tree_c[(size_t(3) * kNodes + i) * 4];
        const uint32_t l0 = (uint32_t(i) * 197u + 31u) & (kNodes - 1), l1 = (uint32_t(i) * 89u + 7u) & (kNodes - 1);
        q3[0] = l0 | (l1 << 16); q3[1] = l1 | (l0 << 16); q3[2] = 0; q3[3] = 0;
    }
    uint4 *d_tree, *d_tree_b, *d_tree_c;
    unsigned long long* d_out;
    CHECK(hipMalloc(&d_tree, tree.size() * 4));
    CHECK(hipMemcpy(d_tree, tree.data(), tree.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_tree_b, tree_b.size() * 4));
    CHECK(hipMemcpy(d_tree_b, tree_b.data(), tree_b.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_tree_c, tree_c.size() * 4));
    CHECK(hipMemcpy(d_tree_c, tree_c.data(), tree_c.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_out, 16 + 1024 * 8));
    const char* names[11] = {"earlier loop (56 instr.)", "no exit test", "no push/pop", "vector select", "scalar logic spread", "node halves side by side",
                            "clamp for v_max/v_min", "dot2 + clamp", "4-child node, dot2", "earlier loop, LDS links", "loop as run (49 instr.)"};
    const int boxes[11] = {2, 2, 2, 2, 2, 2, 2, 2, 4, 2, 2};
    std::printf("%-24s %14s %14s %14s %16s   (SIMD cycles per step of one wave, 64 lanes on)\n", "variant", "1 wave/SIMD", "2 waves/SIMD",
                "4 waves/SIMD", "per box, 4 w/SIMD");
    for (int coherent = 0; coherent < 2; ++coherent) {
    std::printf("-- %s\n", coherent ? "every lane of a wave on the same node (broadcast reads)" : "every lane on a node of its own (random 16-byte slots)");
    for (int v = 0; v < 11; ++v) {
        double cyc[3];
        int slot = 0;
        for (int threads : {256, 512, 1024}) {
            const size_t lds_bytes = size_t(v == 8 ? 4 : 2) * kNodes * 16 + size_t(threads) * 4 * 8;
            CHECK(hipMemset(d_out, 0, 16));
            switch (v) {
                case 0: hipLaunchKernelGGL(k_inner<0>, dim3(n_cu), dim3(threads), lds_bytes, 0, d_out, d_tree, 1.5f, coherent); break;
                case 1: hipLaunchKernelGGL(k_inner<1>, dim3(n_cu), dim3(threads), lds_bytes, 0, d_out, d_tree, 1.5f, coherent); break;
                case 2: hipLaunchKernelGGL(k_inner<2>, dim3(n_cu), dim3(threads), lds_bytes, 0, d_out, d_tree, 1.5f, coherent); break;
                case 3: hipLaunchKernelGGL(k_inner<3>, dim3(n_cu), dim3(threads), lds_bytes, 0, d_out, d_tree, 1.5f, coherent); break;
                case 4: hipLaunchKernelGGL(k_inner<4>, dim3(n_cu), dim3(threads), lds_bytes, 0, d_out, d_tree, 1.5f, coherent); break;
                case 5: hipLaunchKernelGGL(k_inner<5>, dim3(n_cu), dim3(threads), lds_bytes, 0, d_out, d_tree, 1.5f, coherent); break;
                case 6: hipLaunchKernelGGL(k_inner<6>, dim3(n_cu), dim3(threads), lds_bytes, 0, d_out, d_tree, 1.5f, coherent); break;
                case 7: hipLaunchKernelGGL(k_inner<7>, dim3(n_cu), dim3(threads), lds_bytes, 0, d_out, d_tree_b, 1.5f, coherent); break;
                case 9: hipLaunchKernelGGL(k_inner<9>, dim3(n_cu), dim3(threads), lds_bytes, 0, d_out, d_tree, 1.5f, coherent); break;
                case 10: hipLaunchKernelGGL(k_inner<10>, dim3(n_cu), dim3(threads), lds_bytes, 0, d_out, d_tree, 1.5f, coherent); break;
                default: hipLaunchKernelGGL(k_inner<8>, dim3(n_cu), dim3(threads), lds_bytes, 0, d_out, d_tree_c, 1.5f, coherent); break;
            }
            CHECK(hipDeviceSynchronize());
            unsigned long long hh[2];
            CHECK(hipMemcpy(hh, d_out, 16, hipMemcpyDeviceToHost));
            const double waves = double(n_cu) * threads / 64.0;
            cyc[slot++] = double(hh[0]) / waves / kSteps / (threads / 256.0);
        }
        std::printf("%-24s %14.1f %14.1f %14.1f %16.1f\n", names[v], cyc[0], cyc[1], cyc[2], cyc[2] / boxes[v]);
    }
    }
    CHECK(hipFree(d_tree));
    CHECK(hipFree(d_tree_b));
    CHECK(hipFree(d_tree_c));
    CHECK(hipFree(d_out));
    return 0;
}
