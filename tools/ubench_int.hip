// Integer-VALU issue-rate microbenchmark for gfx950 (round 2 rewrite).
//
// Question it answers: what is the VALU-issue ceiling for the instruction classes the Goldilocks / Poseidon
// kernels are made of, in SHADER CYCLES per wavefront-instruction per SIMD, with the clock MEASURED rather
// than assumed?  Method:
//   * every wave brackets its loop with s_memtime (shader-clock ticks) and s_memrealtime (constant 100 MHz),
//     so the effective clock = d(memtime) / d(realtime) * 100 MHz comes out of the run itself;
//   * W waves per SIMD are made resident (grid = 256 CUs x W blocks of 256 threads), W swept over 1..8:
//     throughput cycles per instruction = (cycles a wave needed) / (W x instructions per wave);
//   * each class runs as 8 independent dependency chains per lane, so at W >= 2 latency is hidden and the
//     figure is the issue rate; the W = 1 column shows the single-wave (latency-exposed) rate;
//   * besides single opcodes, the REAL sequences are timed (gl_mul_nc, x^7 S-box, dense MDS layer, one whole
//     permutation) straight from vectorx_amd/csrc — their cycles divided by their instruction counts give the
//     mix-weighted ceiling bench.py prices hash_leaves_colmajor_kernel against.
// Build + run (GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vectorx_amd/csrc tools/ubench_int.hip -o gpurun_out/ubench_int
//   ./gpurun_out/ubench_int > gpurun_out/ubench_int.md
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "poseidon.hip.h"

#define NCHAIN 8
#define BODY_REPEAT 4   // instructions per chain per loop iteration

struct WaveStamp {
  uint64_t cyc, rt;
};

__device__ __forceinline__ uint64_t memtime() {
  uint64_t t;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t));
  return t;
}
__device__ __forceinline__ uint64_t memrealtime() {
  uint64_t t;
  asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t));
  return t;
}

// One asm statement holds the whole loop body (4 repetitions x 8 chains), so the compiler cannot put hazard nops between
// the instructions under test (hipcc pads consecutive inline-asm statements that touch the same registers with s_nop 0).
// Operands: %0..%7 = a[0..7] (32-bit), %8..%15 = w[0..7] (64-bit), %16 = b, %17 = c (32-bit).
#define A0 "0"
#define A1 "1"
#define A2 "2"
#define A3 "3"
#define A4 "4"
#define A5 "5"
#define A6 "6"
#define A7 "7"
#define W0 "8"
#define W1 "9"
#define W2 "10"
#define W3 "11"
#define W4 "12"
#define W5 "13"
#define W6 "14"
#define W7 "15"
#define REP8(X) X(0, 1) X(1, 2) X(2, 3) X(3, 4) X(4, 5) X(5, 6) X(6, 7) X(7, 0)
#define REP4P(X) X(0, 1) X(2, 3) X(4, 5) X(6, 7)
#define BODY(X) REP8(X) REP8(X) REP8(X) REP8(X)
#define BODYP(X) REP4P(X) REP4P(X) REP4P(X) REP4P(X)
#define RUN(TEXT)                                                                                                              \
  asm volatile(TEXT : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(w[0]), \
               "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7])                                \
               : "v"(b), "v"(c)                                                                                                \
               : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27")

#define T_MAD_SGPR(i, j) "v_mad_u64_u32 %" W##i ", s[20:21], %" A##i ", %16, %" W##i "\n"
#define T_MAD_VCC(i, j) "v_mad_u64_u32 %" W##i ", vcc, %" A##i ", %16, %" W##i "\n"
#define T_MAD_NOACC(i, j) "v_mad_u64_u32 %" W##i ", vcc, %" A##i ", %16, 0\n"
#define T_MAD_CONST(i, j) "v_mad_u64_u32 %" W##i ", vcc, %" A##i ", 41, %" W##i "\n"
#define T_MUL_LO(i, j) "v_mul_lo_u32 %" A##i ", %" A##i ", %16\n"
#define T_MUL_HI(i, j) "v_mul_hi_u32 %" A##i ", %" A##i ", %16\n"
#define T_ADD_U32(i, j) "v_add_u32_e32 %" A##i ", %" A##i ", %16\n"
#define T_ADDCO_VCC(i, j) "v_add_co_u32_e32 %" A##i ", vcc, %" A##i ", %16\n"
#define T_ADDC_VCC_PAIR(i, j) "v_add_co_u32_e32 %" A##i ", vcc, %" A##i ", %16\n v_addc_co_u32_e32 %" A##j ", vcc, %" A##j ", %17, vcc\n"
#define T_ADDCO_SGPR(i, j) "v_add_co_u32_e64 %" A##i ", s[20:21], %" A##i ", %16\n"
#define T_ADDC_SGPR_ADJ(i, j) "v_add_co_u32_e64 %" A##i ", s[20:21], %" A##i ", %16\n v_addc_co_u32_e64 %" A##j ", s[20:21], %" A##j ", %17, s[20:21]\n"
#define T_ADDC_SGPR_NOP(i, j) "v_add_co_u32_e64 %" A##i ", s[20:21], %" A##i ", %16\n s_nop 0\n v_addc_co_u32_e64 %" A##j ", s[20:21], %" A##j ", %17, s[20:21]\n"
#define T_ADDC_SGPR_NOP1(i, j) "v_add_co_u32_e64 %" A##i ", s[20:21], %" A##i ", %16\n s_nop 1\n v_addc_co_u32_e64 %" A##j ", s[20:21], %" A##j ", %17, s[20:21]\n"
#define T_ADDC_SGPR_ILV(i, j)                                                                                             \
  "v_add_co_u32_e64 %" A##i ", s[20:21], %" A##i ", %16\n v_add_co_u32_e64 %" A##j ", s[22:23], %" A##j ", %16\n"         \
  " v_add_u32_e32 %" A##i ", %" A##i ", %17\n"                                                                            \
  " v_addc_co_u32_e64 %" A##i ", s[20:21], %" A##i ", %17, s[20:21]\n v_addc_co_u32_e64 %" A##j ", s[22:23], %" A##j ", %17, s[22:23]\n"
#define T_SUBB_SGPR(i, j) "v_subb_co_u32_e64 %" A##i ", s[20:21], %" A##i ", %16, s[22:23]\n"
#define T_SUBB_VCC(i, j) "v_subb_co_u32_e32 %" A##i ", vcc, %" A##i ", %16, vcc\n"
#define T_CNDMASK_SGPR(i, j) "v_cndmask_b32_e64 %" A##i ", %" A##i ", %16, s[22:23]\n"
#define T_CNDMASK_CONST(i, j) "v_cndmask_b32_e64 %" A##i ", 0, -1, s[22:23]\n"
#define T_CNDMASK_VCC(i, j) "v_cndmask_b32_e32 %" A##i ", %" A##i ", %16, vcc\n"
#define T_CMP_CNDMASK_VCC(i, j) "v_cmp_lt_u32_e32 vcc, %" A##i ", %16\n v_cndmask_b32_e32 %" A##i ", %" A##i ", %17, vcc\n"
#define T_CNDMASK_VCC64(i, j) "v_cndmask_b32_e64 %" A##i ", %" A##i ", %16, vcc\n"
#define T_MOV_B64(i, j) "v_mov_b64 %" W##i ", %" W##j "\n"
#define T_LSHL_ADD_U64(i, j) "v_lshl_add_u64 %" W##i ", %" W##i ", 0, %" W##j "\n"
#define T_LSHLREV_B64(i, j) "v_lshlrev_b64 %" W##i ", 3, %" W##i "\n"
#define T_LSHRREV_B32(i, j) "v_lshrrev_b32_e32 %" A##i ", 1, %" A##i "\n"
#define T_AND_B32(i, j) "v_and_b32_e32 %" A##i ", %" A##i ", %16\n"
#define T_MOV_B32(i, j) "v_mov_b32_e32 %" A##i ", %16\n"
#define T_CMP_LT_U64(i, j) "v_cmp_lt_u64_e32 vcc, %" W##i ", %" W##j "\n"
#define T_ADD_NOP(i, j) "v_add_u32_e32 %" A##i ", %" A##i ", %16\n s_nop 0\n"
#define T_MAD_NOP(i, j) "v_mad_u64_u32 %" W##i ", s[20:21], %" A##i ", %16, %" W##i "\n s_nop 0\n"
#define T_ADD3_U32(i, j) "v_add3_u32 %" A##i ", %" A##i ", %16, %17\n"
#define T_MAD_ADD(i, j) "v_mad_u64_u32 %" W##i ", vcc, %" A##i ", %16, %" W##i "\n v_add_u32_e32 %" A##i ", %" A##i ", %17\n"
#define T_MAD_ADD2(i, j) "v_mad_u64_u32 %" W##i ", vcc, %" A##i ", %16, %" W##i "\n v_add_u32_e32 %" A##i ", %" A##i ", %17\n v_and_b32_e32 %" A##j ", %" A##j ", %16\n"

template <int OP>
__global__ __launch_bounds__(256) void k_op(WaveStamp* stamps, uint64_t* sink, int iters, uint32_t seed) {
  uint32_t a[NCHAIN], b = seed | 1, c = seed * 3 + 7;
  uint64_t w[NCHAIN];
#pragma unroll
  for (int i = 0; i < NCHAIN; ++i) {
    a[i] = threadIdx.x * 17 + i + seed;
    w[i] = ((uint64_t)a[i] << 32) | (a[i] * 5);
  }
  const uint64_t t0 = memtime(), r0 = memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (OP == 0) RUN(BODY(T_MAD_SGPR));
    if (OP == 1) RUN(BODY(T_MAD_VCC));
    if (OP == 2) RUN(BODY(T_MAD_NOACC));
    if (OP == 3) RUN(BODY(T_MAD_CONST));
    if (OP == 4) RUN(BODY(T_MUL_LO));
    if (OP == 5) RUN(BODY(T_MUL_HI));
    if (OP == 6) RUN(BODY(T_ADD_U32));
    if (OP == 7) RUN(BODY(T_ADDCO_VCC));
    if (OP == 8) RUN(BODYP(T_ADDC_VCC_PAIR));
    if (OP == 9) RUN(BODY(T_ADDCO_SGPR));
    if (OP == 10) RUN(BODYP(T_ADDC_SGPR_ADJ));
    if (OP == 11) RUN(BODYP(T_ADDC_SGPR_NOP));
    if (OP == 12) RUN(BODYP(T_ADDC_SGPR_NOP1));
    if (OP == 13) RUN(BODYP(T_ADDC_SGPR_ILV));
    if (OP == 14) RUN(BODY(T_SUBB_SGPR));
    if (OP == 15) RUN(BODY(T_SUBB_VCC));
    if (OP == 16) RUN(BODY(T_CNDMASK_SGPR));
    if (OP == 17) RUN(BODY(T_CNDMASK_CONST));
    if (OP == 18) RUN(BODY(T_CNDMASK_VCC));
    if (OP == 19) RUN(BODY(T_LSHL_ADD_U64));
    if (OP == 20) RUN(BODY(T_LSHLREV_B64));
    if (OP == 21) RUN(BODY(T_LSHRREV_B32));
    if (OP == 22) RUN(BODY(T_AND_B32));
    if (OP == 23) RUN(BODY(T_MOV_B32));
    if (OP == 24) RUN(BODY(T_CMP_LT_U64));
    if (OP == 25) RUN(BODY(T_ADD_NOP));
    if (OP == 26) RUN(BODY(T_MAD_NOP));
    if (OP == 27) RUN(BODY(T_ADD3_U32));
    if (OP == 28) RUN(BODY(T_MAD_ADD));
    if (OP == 29) RUN(BODY(T_MAD_ADD2));
    if (OP == 30) RUN(BODY(T_CMP_CNDMASK_VCC));
    if (OP == 31) RUN(BODY(T_CNDMASK_VCC64));
    if (OP == 32) RUN(BODY(T_MOV_B64));
  }
  const uint64_t t1 = memtime(), r1 = memrealtime();
  uint64_t s = b;
#pragma unroll
  for (int i = 0; i < NCHAIN; ++i) s += a[i] + w[i];
  if (s == 0x1234567) sink[0] = s;
  if ((threadIdx.x & 63) == 0) {
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    stamps[wave].cyc = t1 - t0;
    stamps[wave].rt = r1 - r0;
  }
}

// ---- candidate: the fused multiply-reduce as ONE asm block, carries through VCC wherever the chain is adjacent (the
// implicit-VCC VOP2 forms need no hazard nops), explicit SGPR pairs only where the consumer sits >= 2 instructions later.
// AMDGPU inline asm has no sub-register modifiers, so the 64-bit temporaries whose halves are addressed separately
// live in FIXED registers (clobbered): P00 = v[100:101], P01 = v[102:103], P11 = v[104:105], P10 = v[106:107],
// T = v[108:109], D = v[110:111]; carries in s[40:47].
__device__ __forceinline__ u64 gl_mul_nc_v2(u64 a, u64 b) {
  u64 r;
  asm("v_mad_u64_u32 v[100:101], vcc, %[a0], %[b0], 0\n"
      "v_mad_u64_u32 v[102:103], vcc, %[a0], %[b1], 0\n"
      "v_mad_u64_u32 v[104:105], vcc, %[a1], %[b1], 0\n"
      "v_mad_u64_u32 v[106:107], s[40:41], %[a1], %[b0], v[102:103]\n"
      "v_add_co_u32_e32 v101, vcc, v101, v106\n"
      "v_addc_co_u32_e32 v104, vcc, v104, v107, vcc\n"
      "v_addc_co_u32_e32 v105, vcc, 0, v105, vcc\n"
      "v_mad_u64_u32 v[108:109], s[42:43], v104, -1, v[100:101]\n"
      "v_subb_co_u32_e64 v108, vcc, v108, v105, s[40:41]\n"
      "v_subbrev_co_u32_e32 v109, vcc, 0, v109, vcc\n"
      "s_andn2_b64 s[44:45], s[42:43], vcc\n"
      "s_andn2_b64 s[46:47], vcc, s[42:43]\n"
      "v_cndmask_b32_e64 v110, 0, -1, s[44:45]\n"
      "v_cndmask_b32_e64 v111, 0, -1, s[46:47]\n"
      "v_cndmask_b32_e64 v110, v110, 1, s[46:47]\n"
      "v_lshl_add_u64 %[r], v[108:109], 0, v[110:111]\n"
      : [r] "=v"(r)
      : [a0] "v"((u32)a), [a1] "v"((u32)(a >> 32)), [b0] "v"((u32)b), [b1] "v"((u32)(b >> 32))
      : "vcc", "scc", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "s40", "s41", "s42",
        "s43", "s44", "s45", "s46", "s47");
  return r;
}
// (A "v3" with the schoolbook (x >> 32) addends expressed as register pairs overlapping the previous product's high half
// — 11 instructions — does not assemble: gfx950 requires 64-bit VGPR tuples to be even-aligned.)

// ---- round 3 experiment: THREE independent multiply-reduces interleaved instruction by instruction in ONE asm
// block.  Stream A keeps its adjacent carries in VCC (as gl_mul_nc_v2); streams B and C carry through their own SGPR pairs
// (VOP3 forms), which is legal here because an SGPR written by a VALU instruction may be read by a VALU instruction only two
// issue slots later (gfx940+ hazard) — and in the interleaved order the consumer of a B carry sits exactly two slots after
// its producer (one A and one C instruction in between).  Fixed temporaries: A v[100:105], B v[106:111], C v[112:117] (+ one
// output-side pair each: v[118:123]); SGPRs s[40:63].
#define MUL3_LINE_A(x) x "\n"
__device__ __forceinline__ void gl_mul_nc_x3(u64 a0, u64 b0, u64 a1, u64 b1, u64 a2, u64 b2, u64& r0, u64& r1, u64& r2) {
  asm(// p00
      "v_mad_u64_u32 v[100:101], vcc, %[a0l], %[b0l], 0\n"
      "v_mad_u64_u32 v[106:107], s[48:49], %[a1l], %[b1l], 0\n"
      "v_mad_u64_u32 v[112:113], s[56:57], %[a2l], %[b2l], 0\n"
      // p01
      "v_mad_u64_u32 v[102:103], vcc, %[a0l], %[b0h], 0\n"
      "v_mad_u64_u32 v[108:109], s[48:49], %[a1l], %[b1h], 0\n"
      "v_mad_u64_u32 v[114:115], s[56:57], %[a2l], %[b2h], 0\n"
      // p11
      "v_mad_u64_u32 v[104:105], vcc, %[a0h], %[b0h], 0\n"
      "v_mad_u64_u32 v[110:111], s[48:49], %[a1h], %[b1h], 0\n"
      "v_mad_u64_u32 v[116:117], s[56:57], %[a2h], %[b2h], 0\n"
      // p10 = a1 b0 + p01, carry cM
      "v_mad_u64_u32 v[102:103], s[40:41], %[a0h], %[b0l], v[102:103]\n"
      "v_mad_u64_u32 v[108:109], s[50:51], %[a1h], %[b1l], v[108:109]\n"
      "v_mad_u64_u32 v[114:115], s[58:59], %[a2h], %[b2l], v[114:115]\n"
      // lo64.hi = p00.hi + p10.lo
      "v_add_co_u32_e64 v101, s[42:43], v101, v102\n"
      "v_add_co_u32_e64 v107, s[52:53], v107, v108\n"
      "v_add_co_u32_e64 v113, s[60:61], v113, v114\n"
      // hl
      "v_addc_co_u32_e64 v104, s[42:43], v104, v103, s[42:43]\n"
      "v_addc_co_u32_e64 v110, s[52:53], v110, v109, s[52:53]\n"
      "v_addc_co_u32_e64 v116, s[60:61], v116, v115, s[60:61]\n"
      // hh
      "v_addc_co_u32_e64 v105, s[42:43], 0, v105, s[42:43]\n"
      "v_addc_co_u32_e64 v111, s[52:53], 0, v111, s[52:53]\n"
      "v_addc_co_u32_e64 v117, s[60:61], 0, v117, s[60:61]\n"
      // T = hl * EPS + lo64, carry cT
      "v_mad_u64_u32 v[100:101], s[44:45], v104, -1, v[100:101]\n"
      "v_mad_u64_u32 v[106:107], s[54:55], v110, -1, v[106:107]\n"
      "v_mad_u64_u32 v[112:113], s[62:63], v116, -1, v[112:113]\n"
      // u = T - hh - cM
      "v_subb_co_u32_e64 v100, s[40:41], v100, v105, s[40:41]\n"
      "v_subb_co_u32_e64 v106, s[50:51], v106, v111, s[50:51]\n"
      "v_subb_co_u32_e64 v112, s[58:59], v112, v117, s[58:59]\n"
      "v_subbrev_co_u32_e64 v101, s[40:41], 0, v101, s[40:41]\n"
      "v_subbrev_co_u32_e64 v107, s[50:51], 0, v107, s[50:51]\n"
      "v_subbrev_co_u32_e64 v113, s[58:59], 0, v113, s[58:59]\n"
      // masks: m1 = cT & ~bw (+EPS), m2 = bw & ~cT (-EPS)
      "s_andn2_b64 s[42:43], s[44:45], s[40:41]\n"
      "s_andn2_b64 s[40:41], s[40:41], s[44:45]\n"
      "s_andn2_b64 s[52:53], s[54:55], s[50:51]\n"
      "s_andn2_b64 s[50:51], s[50:51], s[54:55]\n"
      "s_andn2_b64 s[60:61], s[62:63], s[58:59]\n"
      "s_andn2_b64 s[58:59], s[58:59], s[62:63]\n"
      "v_cndmask_b32_e64 v118, 0, -1, s[42:43]\n"
      "v_cndmask_b32_e64 v120, 0, -1, s[52:53]\n"
      "v_cndmask_b32_e64 v122, 0, -1, s[60:61]\n"
      "v_cndmask_b32_e64 v119, 0, -1, s[40:41]\n"
      "v_cndmask_b32_e64 v121, 0, -1, s[50:51]\n"
      "v_cndmask_b32_e64 v123, 0, -1, s[58:59]\n"
      "v_cndmask_b32_e64 v118, v118, 1, s[40:41]\n"
      "v_cndmask_b32_e64 v120, v120, 1, s[50:51]\n"
      "v_cndmask_b32_e64 v122, v122, 1, s[58:59]\n"
      "v_lshl_add_u64 %[r0], v[100:101], 0, v[118:119]\n"
      "v_lshl_add_u64 %[r1], v[106:107], 0, v[120:121]\n"
      "v_lshl_add_u64 %[r2], v[112:113], 0, v[122:123]\n"
      : [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=v"(r2)
      : [a0l] "v"((u32)a0), [a0h] "v"((u32)(a0 >> 32)), [b0l] "v"((u32)b0), [b0h] "v"((u32)(b0 >> 32)), [a1l] "v"((u32)a1), [a1h] "v"((u32)(a1 >> 32)),
        [b1l] "v"((u32)b1), [b1h] "v"((u32)(b1 >> 32)), [a2l] "v"((u32)a2), [a2h] "v"((u32)(a2 >> 32)), [b2l] "v"((u32)b2), [b2h] "v"((u32)(b2 >> 32))
      : "vcc", "scc", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116",
        "v117", "v118", "v119", "v120", "v121", "v122", "v123", "s40", "s41", "s42", "s43", "s44", "s45", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55",
        "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63");
}
GLD void poseidon_sbox_x3(u64& x, u64& y, u64& z) {
  u64 x2, y2, z2, x4, y4, z4, x3, y3, z3;
  gl_mul_nc_x3(x, x, y, y, z, z, x2, y2, z2);
  gl_mul_nc_x3(x2, x2, y2, y2, z2, z2, x4, y4, z4);
  gl_mul_nc_x3(x, x2, y, y2, z, z2, x3, y3, z3);
  gl_mul_nc_x3(x3, x4, y3, y4, z3, z4, x, y, z);
}
__global__ void k_check_mul3(const u64* a, const u64* b, u64* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (3 * i + 2 >= n) return;
  u64 r0, r1, r2;
  gl_mul_nc_x3(a[3 * i], b[3 * i], a[3 * i + 1], b[3 * i + 1], a[3 * i + 2], b[3 * i + 2], r0, r1, r2);
  out[3 * i] = gl_canon(r0), out[3 * i + 1] = gl_canon(r1), out[3 * i + 2] = gl_canon(r2);
}
GLD u64 poseidon_sbox_nc_v2(u64 x) {
  const u64 x2 = gl_mul_nc_v2(x, x), x4 = gl_mul_nc_v2(x2, x2), x3 = gl_mul_nc_v2(x, x2);
  return gl_mul_nc_v2(x3, x4);
}
__global__ void k_check_mul(const u64* a, const u64* b, u64* out_ref, u64* out_v2, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out_ref[i] = gl_canon(gl_mul_nc(a[i], b[i]));
  out_v2[i] = gl_canon(gl_mul_nc_v2(a[i], b[i]));
}

// ---- real sequences from the product headers ------------------------------------------------------------------
enum Seq { SEQ_MUL, SEQ_SBOX, SEQ_MDS, SEQ_PERM, SEQ_ADD_NC, SEQ_MUL_V2, SEQ_SBOX_V2, SEQ_MUL_V3, SEQ_SBOX_V3, SEQ_MDS_RC, SEQ_BLOCK3, SEQ_MUL_X3, SEQ_SBOX_X3, SEQ_SBOX_FX, NUM_SEQS };
template <int SEQ>
__global__ __launch_bounds__(256, 4) void k_seq(WaveStamp* stamps, uint64_t* sink, int iters, uint64_t seed) {
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = seed * (threadIdx.x + 3 + i) + i;
  const uint64_t t0 = memtime(), r0 = memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (SEQ == SEQ_MUL) {
#pragma unroll
      for (int i = 0; i < 12; ++i) s[i] = gl_mul_nc(s[i], s[(i + 5) % 12]);
    }
    if (SEQ == SEQ_SBOX) {
#pragma unroll
      for (int i = 0; i < 12; ++i) s[i] = poseidon_sbox_nc(s[i]);
    }
    if (SEQ == SEQ_MUL_V2) {
#pragma unroll
      for (int i = 0; i < 12; ++i) s[i] = gl_mul_nc_v2(s[i], s[(i + 5) % 12]);
    }
    if (SEQ == SEQ_SBOX_V2) {
#pragma unroll
      for (int i = 0; i < 12; ++i) s[i] = poseidon_sbox_nc_v2(s[i]);
    }
    if (SEQ == SEQ_MUL_X3) {
#pragma unroll
      for (int i = 0; i < 12; i += 3) gl_mul_nc_x3(s[i], s[(i + 5) % 12], s[i + 1], s[(i + 6) % 12], s[i + 2], s[(i + 7) % 12], s[i], s[i + 1], s[i + 2]);
    }
    if (SEQ == SEQ_SBOX_X3) {
#pragma unroll
      for (int i = 0; i < 12; i += 3) poseidon_sbox_x3(s[i], s[i + 1], s[i + 2]);
    }
    if (SEQ == SEQ_SBOX_FX) {
#pragma unroll
      for (int i = 0; i < 12; ++i) s[i] = poseidon_sbox_fx(s[i]);
    }
    if (SEQ == SEQ_MDS_RC) poseidon_mds_rc_nc(s, it & 15);
    if (SEQ == SEQ_BLOCK3) poseidon_partial_block_nc<POSEIDON_BLOCK_B>(s, POSEIDON_BLK.kappa[it & 3], POSEIDON_BLK.K[it & 3]);
    if (SEQ == SEQ_MDS) poseidon_mds_nc(s);
    if (SEQ == SEQ_PERM) poseidon_permute_nc(s);
    if (SEQ == SEQ_ADD_NC) {
#pragma unroll
      for (int i = 0; i < 12; ++i) s[i] = gl_add_nc_c(s[i], POSEIDON_RC[i]);
    }
  }
  const uint64_t t1 = memtime(), r1 = memrealtime();
  u64 x = 0;
#pragma unroll
  for (int i = 0; i < 12; ++i) x ^= s[i];
  if (x == 0x1234567) sink[0] = x;
  if ((threadIdx.x & 63) == 0) {
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    stamps[wave].cyc = t1 - t0;
    stamps[wave].rt = r1 - r0;
  }
}

struct Result {
  double cyc_per_unit;  // shader cycles per unit per SIMD (throughput)
  double ghz;           // effective clock from memtime / memrealtime
  double wall_ms;
  double units_per_s;   // chip-wide
};

template <class Launch>
Result measure(Launch&& launch, int waves_per_simd, double units_per_wave) {
  const int blocks = 256 * waves_per_simd, threads = 256;
  const size_t nwaves = (size_t)blocks * 4;
  WaveStamp* d_st;
  uint64_t* d_sink;
  hipMalloc(&d_st, nwaves * sizeof(WaveStamp));
  hipMalloc(&d_sink, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch(blocks, threads, d_st, d_sink);  // warm-up
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch(blocks, threads, d_st, d_sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<WaveStamp> st(nwaves);
  hipMemcpy(st.data(), d_st, nwaves * sizeof(WaveStamp), hipMemcpyDeviceToHost);
  std::vector<double> cyc(nwaves), ghz(nwaves);
  for (size_t i = 0; i < nwaves; ++i) {
    cyc[i] = (double)st[i].cyc;
    ghz[i] = st[i].rt ? (double)st[i].cyc / ((double)st[i].rt / 100e6) / 1e9 : 0;
  }
  std::sort(cyc.begin(), cyc.end());
  std::sort(ghz.begin(), ghz.end());
  Result r;
  r.cyc_per_unit = cyc[nwaves / 2] / (waves_per_simd * units_per_wave);  // median wave
  r.ghz = ghz[nwaves / 2];
  r.wall_ms = ms;
  r.units_per_s = (double)nwaves * units_per_wave / (ms * 1e-3);
  hipFree(d_st);
  hipFree(d_sink);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return r;
}

static const int SWEEP[] = {1, 2, 3, 4, 6, 8};
static const int NSWEEP = 6;

// valu_per_iter: VALU instructions one wave issues per loop iteration
template <int OP>
void run_op(const char* name, int valu_per_iter) {
  const int iters = 2048;
  printf("| `%s` |", name);
  double ghz = 0, rate8 = 0;
  for (int k = 0; k < NSWEEP; ++k) {
    const int W = SWEEP[k];
    const double per_wave = (double)iters * valu_per_iter;
    Result r = measure([&](int b, int t, WaveStamp* st, uint64_t* sk) { hipLaunchKernelGGL(k_op<OP>, dim3(b), dim3(t), 0, 0, st, sk, iters, 3u + k); }, W, per_wave);
    printf(" %.2f |", r.cyc_per_unit);
    ghz = r.ghz;
    rate8 = r.units_per_s;
  }
  printf(" %.2f | %.3e | %.2f |\n", ghz, rate8, 1024.0 * ghz * 1e9 / rate8);
  fflush(stdout);
}

template <int SEQ>
void run_seq(const char* name, double units_per_iter, int iters, const char* unit) {
  printf("| %s |", name);
  double ghz = 0, rate = 0;
  const int sw[] = {1, 2, 3, 4};
  for (int k = 0; k < 4; ++k) {
    const int W = sw[k];
    Result r = measure([&](int b, int t, WaveStamp* st, uint64_t* sk) { hipLaunchKernelGGL(k_seq<SEQ>, dim3(b), dim3(t), 0, 0, st, sk, iters, 0x9E3779B97F4A7C15ull + k); }, W,
                       (double)iters * units_per_iter);
    printf(" %.1f |", r.cyc_per_unit);
    ghz = r.ghz;
    rate = r.units_per_s * 64;  // per lane
  }
  printf(" %.2f | %.3e %s/s | %.1f |\n", ghz, rate, unit, 1024.0 * ghz * 1e9 / (rate / 64));
  fflush(stdout);
}

static int check_mul_x3() {
  const size_t n = 3 << 18;
  std::vector<u64> a(n), b(n), r(n);
  u64 x = 0x13198A2E03707344ull;
  auto next = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  const u64 edge[] = {0, 1, GL_P - 1, GL_P, GL_P + 1, ~0ull, 0xFFFFFFFFull, 0x100000000ull, 0xFFFFFFFF00000000ull, 0x8000000000000000ull, GL_EPS - 1, 0xFFFFFFFEFFFFFFFFull};
  const int ne = sizeof(edge) / sizeof(edge[0]);
  for (size_t i = 0; i < n; ++i) {
    a[i] = next();
    b[i] = next();
    if (i < (size_t)ne * ne) { a[i] = edge[i / ne]; b[i] = edge[i % ne]; }
  }
  u64 *da, *db, *d0;
  hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&d0, n * 8);
  hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_check_mul3, dim3((n / 3 + 255) / 256), dim3(256), 0, 0, da, db, d0, n);
  hipMemcpy(r.data(), d0, n * 8, hipMemcpyDeviceToHost);
  size_t bad = 0;
  for (size_t i = 0; i < n; ++i) {
    const unsigned __int128 P = (unsigned __int128)a[i] * b[i];
    if (r[i] != (u64)(P % GL_P)) ++bad;
  }
  printf("gl_mul_nc_x3 (three multiplies interleaved in one asm block: stream A on VCC, B and C on their own SGPR carry pairs, consumer two issue slots "
         "after its producer) vs 128-bit host arithmetic on %zu operand pairs incl. %d edge pairs: %zu mismatches\n\n", n, ne * ne, bad);
  hipFree(da); hipFree(db); hipFree(d0);
  return bad != 0;
}

static int check_mul_v2() {
  const size_t n = 1 << 20;
  std::vector<u64> a(n), b(n), r0(n), r1(n);
  u64 x = 0x243F6A8885A308D3ull;
  auto next = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  const u64 edge[] = {0, 1, GL_P - 1, GL_P, GL_P + 1, ~0ull, 0xFFFFFFFFull, 0x100000000ull, 0xFFFFFFFF00000000ull, 0x8000000000000000ull, GL_EPS - 1, 0xFFFFFFFEFFFFFFFFull};
  const int ne = sizeof(edge) / sizeof(edge[0]);
  for (size_t i = 0; i < n; ++i) {
    a[i] = next();
    b[i] = next();
    if (i < (size_t)ne * ne) { a[i] = edge[i / ne]; b[i] = edge[i % ne]; }
  }
  u64 *da, *db, *d0, *d1;
  hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8);
  hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_check_mul, dim3(n / 256), dim3(256), 0, 0, da, db, d0, d1, n);
  hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost);
  size_t bad = 0, badref = 0;
  for (size_t i = 0; i < n; ++i) {
    const unsigned __int128 P = (unsigned __int128)a[i] * b[i];
    const u64 want = (u64)(P % GL_P);
    if (r0[i] != want) ++badref;
    if (r1[i] != want) ++bad;
  }
  printf("\ngl_mul_nc_v2 (single asm block, carries through VCC) vs 128-bit host arithmetic on %zu operand pairs incl. %d edge pairs: %zu mismatches (product gl_mul_nc: %zu)\n\n", n, ne * ne, bad, badref);
  hipFree(da); hipFree(db); hipFree(d0); hipFree(d1);
  return bad != 0;
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  printf("# gfx950 integer-VALU issue rates (tools/ubench_int.hip)\n\n");
  printf("device: %s, %d CUs, clockRate %d kHz; 256-thread blocks, grid = 256 x W (W waves per SIMD resident); cycles = s_memtime ticks of the MEDIAN wave / "
         "(W x VALU instructions per wave) = shader cycles per wavefront-instruction per SIMD; GHz = median over waves of d(s_memtime)/d(s_memrealtime @100 MHz) "
         "at W = 8; last column = chip-wide wavefront-instructions/s at W = 8 from HIP-event wall time.  Each loop body is ONE asm statement: 32 instructions on 8 "
         "independent chains.\n\n",
         prop.name, prop.multiProcessorCount, prop.clockRate);
  printf("## single opcodes and short patterns\n\n");
  printf("| instruction(s) | W=1 | W=2 | W=3 | W=4 | W=6 | W=8 | GHz | wave-inst/s (W=8) | cycles = 1024 SIMDs x GHz / rate |\n|---|---|---|---|---|---|---|---|---|---|\n");
  run_op<0>("v_mad_u64_u32 v,s[..],a,b,acc", 32);
  run_op<1>("v_mad_u64_u32 v,vcc,a,b,acc", 32);
  run_op<2>("v_mad_u64_u32 v,vcc,a,b,0", 32);
  run_op<3>("v_mad_u64_u32 v,vcc,a,41,acc (inline constant)", 32);
  run_op<4>("v_mul_lo_u32", 32);
  run_op<5>("v_mul_hi_u32", 32);
  run_op<6>("v_add_u32_e32", 32);
  run_op<7>("v_add_co_u32_e32 (vcc out)", 32);
  run_op<8>("v_add_co_u32_e32 ; v_addc_co_u32_e32 (vcc chain)", 32);
  run_op<9>("v_add_co_u32_e64 (sgpr-pair carry out)", 32);
  run_op<10>("v_add_co_u32_e64 s ; v_addc_co_u32_e64 s (adjacent, NO nop: results may be wrong, timing only)", 32);
  run_op<11>("v_add_co_u32_e64 s ; s_nop 0 ; v_addc_co_u32_e64 s (per VALU)", 32);
  run_op<12>("v_add_co_u32_e64 s ; s_nop 1 ; v_addc_co_u32_e64 s (per VALU)", 32);
  run_op<13>("2 sgpr carry chains interleaved + 1 filler, no nop (per VALU)", 80);
  run_op<14>("v_subb_co_u32_e64 (sgpr in, sgpr out)", 32);
  run_op<15>("v_subb_co_u32_e32 (vcc in/out)", 32);
  run_op<16>("v_cndmask_b32_e64 v,v,v,s", 32);
  run_op<17>("v_cndmask_b32_e64 v,0,-1,s", 32);
  run_op<18>("v_cndmask_b32_e32 (vcc)", 32);
  run_op<30>("v_cmp_lt_u32_e32 vcc ; v_cndmask_b32_e32 vcc (per VALU)", 64);
  run_op<31>("v_cndmask_b32_e64 v,v,v,vcc", 32);
  run_op<32>("v_mov_b64", 32);
  run_op<19>("v_lshl_add_u64", 32);
  run_op<20>("v_lshlrev_b64", 32);
  run_op<21>("v_lshrrev_b32_e32", 32);
  run_op<22>("v_and_b32_e32", 32);
  run_op<23>("v_mov_b32_e32", 32);
  run_op<24>("v_cmp_lt_u64_e32", 32);
  run_op<27>("v_add3_u32", 32);
  run_op<25>("v_add_u32_e32 ; s_nop 0 (per VALU)", 32);
  run_op<26>("v_mad_u64_u32 ; s_nop 0 (per VALU)", 32);
  run_op<28>("v_mad_u64_u32 ; v_add_u32_e32 alternating (per VALU)", 64);
  run_op<29>("v_mad_u64_u32 ; v_add_u32 ; v_and_b32 (per VALU)", 96);
  if (check_mul_v2()) printf("**gl_mul_nc_v2 IS WRONG**\n");
  if (check_mul_x3()) printf("**gl_mul_nc_x3 IS WRONG**\n");
  printf("## real sequences (vectorx_amd/csrc as hipcc compiles them, hazard nops included): shader cycles per unit per SIMD\n\n");
  printf("| sequence (unit) | W=1 | W=2 | W=3 | W=4 | GHz | lane-units/s (W=4) | cycles/unit/SIMD from wall time |\n|---|---|---|---|---|---|---|---|\n");
  run_seq<SEQ_ADD_NC>("gl_add_nc_c (per add, wave-wide)", 12, 2048, "adds");
  run_seq<SEQ_MUL>("gl_mul_nc (per multiply)", 12, 1024, "muls");
  run_seq<SEQ_MUL_V2>("gl_mul_nc_v2 = one asm block (per multiply)", 12, 1024, "muls");
  run_seq<SEQ_SBOX>("poseidon_sbox_nc x^7 (per S-box)", 12, 512, "sboxes");
  run_seq<SEQ_MDS_RC>("poseidon_mds_rc_nc, constants folded (per 12x12 layer)", 1, 1024, "layers");
  run_seq<SEQ_BLOCK3>("poseidon_partial_block_nc<4> (per block of 4 partial rounds)", 1, 512, "blocks");
  run_seq<SEQ_SBOX_V2>("x^7 on gl_mul_nc_v2 (per S-box)", 12, 512, "sboxes");
  run_seq<SEQ_SBOX_FX>("x^7 on gl_mul_nc_fx = the product's S-box (per S-box)", 12, 512, "sboxes");
  run_seq<SEQ_MUL_X3>("gl_mul_nc_x3 = 3 multiplies interleaved in one asm block (per multiply)", 12, 1024, "muls");
  run_seq<SEQ_SBOX_X3>("x^7 on gl_mul_nc_x3, three lanes at a time (per S-box)", 12, 512, "sboxes");
  run_seq<SEQ_MDS>("poseidon_mds_nc (per 12x12 layer)", 1, 1024, "layers");
  run_seq<SEQ_PERM>("poseidon_permute_nc (per permutation)", 1, 64, "perms");
  return 0;
}
