// Poseidon-Goldilocks permutation (width 12, x^7, 8 full + 22 partial rounds) for gfx950.
// Replaces plonky2::hash::poseidon / poseidon_goldilocks and the AVX2/NEON hand-scheduled forms
// (plonky2 v0.2.0, un-vendored: /root/reference/Cargo.lock:4848-4905; algorithm per SURVEY.md A.2).
//
// This path is bound by the integer ALU, not HBM.  Measured (tools/ubench_int.hip, profiles/r02_ubench_int.md): on
// gfx950 v_mad_u64_u32, v_mul_*, EVERY carry-producing add/sub (VOP2 forms included), v_cndmask_b32_e64, v_lshl_add_u64,
// 64-bit shifts and compares all issue at ~4.4 shader cycles per wavefront per SIMD; only plain 32-bit VOP1/VOP2 ops
// reach ~2.5, and only in runs.  So time = instruction count, and the design minimises the INSTRUCTION COUNT:
//  * lanes are kept as arbitrary u64 representatives (not canonical) between operations; one
//    conditional subtraction at the very end canonicalises;
//  * every linear layer works on the 32-bit halves of a lane with SMALL INTEGER multipliers (one v_mad_u64_u32 per
//    half per term into plain 64-bit accumulators — no carries) and folds once per output (5 instructions);
//  * round constants are never added on their own: they are the initial values of the accumulators of the linear
//    layer in front of them;
//  * the 22 partial rounds run in blocks of 4 (hashing: [dense layer of full round 3 + 3] + 4·4 + 3, POSEIDON_SCHED_H; gate evaluation: 4·5 + 2) through integer powers of the MDS matrix (below) — upstream's "fast"
//    sparse form needs full-size constants, i.e. six multiply-adds per term instead of two.
// The 12-lane state lives in VGPRs; constants sit in constant memory and, because the loops are unrolled, are fetched
// with scalar loads shared by the whole wavefront.
#pragma once
#include "goldilocks.hip.h"
#include "poseidon_constants.h"

__constant__ u64 POSEIDON_RC[VX_POSEIDON_N_ROUND_CONSTANTS] = VX_POSEIDON_ROUND_CONSTANTS_INIT;

#define POSEIDON_WIDTH 12
#define POSEIDON_RATE 8

// ---- non-canonical ("nc") arithmetic: values are any u64 congruent to the field element ----------
// a: any u64, b: CANONICAL (< p)  ->  a + b (nc).  One carry fix suffices because b <= 2^64 - 2^32.
GLD u64 gl_add_nc_c(u64 a, u64 b) {
  u64 s = a + b;
  if (s < a) s += GL_EPS;
  return s;
}
// a: any u64, b: CANONICAL  ->  a - b (nc).  On a borrow the wrapped difference is >= 2^64 - (p-1) = EPS, so one fix suffices.
GLD u64 gl_sub_nc_c(u64 a, u64 b) {
  u64 d = a - b;
  if (a < b) d -= GL_EPS;
  return d;
}
// The fused multiply-reduce of goldilocks.hip.h (gl_mul_nc) as ONE asm block with the adjacent carries chained through
// VCC (implicit-VCC VOP2 forms): same 16 VALU + 2 SALU instructions, but no hazard nops between dependent statements
// and measurably faster — 61.3 instead of 66.3 shader cycles per multiply per SIMD (profiles/r02_ubench_int.md).
// AMDGPU inline asm cannot name the halves of a 64-bit operand, so the three temporaries whose halves are used
// separately live in FIXED registers v[118:123] (clobbered; free for the compiler outside the block).  Poseidon-only:
// kernels with a tight VGPR budget (the NTT passes run at 40 VGPRs) keep gl_mul_nc.
#ifndef POSEIDON_NO_FX
GLD u64 gl_mul_nc_fx(u64 a, u64 b) {
  u64 r, p01, sM, sT, m1, m2;
  asm("v_mad_u64_u32 v[118:119], vcc, %[a0], %[b0], 0\n"               // p00
      "v_mad_u64_u32 %[p01], vcc, %[a0], %[b1], 0\n"                     // p01
      "v_mad_u64_u32 v[120:121], vcc, %[a1], %[b1], 0\n"                 // p11
      "v_mad_u64_u32 v[122:123], %[sM], %[a1], %[b0], %[p01]\n"          // p10 = a1 b0 + p01, carry cM (worth 2^96 = -1)
      "v_add_co_u32_e32 v119, vcc, v119, v122\n"                         // lo64 = (p00_lo, p00_hi + p10_lo)
      "v_addc_co_u32_e32 v120, vcc, v120, v123, vcc\n"                   // hl
      "v_addc_co_u32_e32 v121, vcc, 0, v121, vcc\n"                      // hh
      "v_mad_u64_u32 v[118:119], %[sT], v120, -1, v[118:119]\n"          // T = hl * EPS + lo64, carry cT
      "v_subb_co_u32_e64 v118, vcc, v118, v121, %[sM]\n"                 // u = T - hh - cM
      "v_subbrev_co_u32_e32 v119, vcc, 0, v119, vcc\n"                   // borrow bw in vcc
      "s_andn2_b64 %[m1], %[sT], vcc\n"                                  // cT & ~bw : + EPS
      "s_andn2_b64 %[m2], vcc, %[sT]\n"                                  // bw & ~cT : - EPS
      "v_cndmask_b32_e64 v122, 0, -1, %[m1]\n"
      "v_cndmask_b32_e64 v123, 0, -1, %[m2]\n"
      "v_cndmask_b32_e64 v122, v122, 1, %[m2]\n"
      "v_lshl_add_u64 %[r], v[118:119], 0, v[122:123]\n"
      : [r] "=v"(r), [p01] "=&v"(p01), [sM] "=&s"(sM), [sT] "=&s"(sT), [m1] "=&s"(m1), [m2] "=&s"(m2)
      : [a0] "v"((u32)a), [a1] "v"((u32)(a >> 32)), [b0] "v"((u32)b), [b1] "v"((u32)(b >> 32))
      : "vcc", "scc", "v118", "v119", "v120", "v121", "v122", "v123");
  return r;
}
#else
GLD u64 gl_mul_nc_fx(u64 a, u64 b) { return gl_mul_nc(a, b); }
#endif
GLD u64 poseidon_sbox_nc(u64 x) {  // generic form (quotient kernel's PoseidonGate, lane-cooperative permutation)
  const u64 x2 = gl_mul_nc(x, x), x4 = gl_mul_nc(x2, x2), x3 = gl_mul_nc(x, x2);
  return gl_mul_nc(x3, x4);
}
GLD u64 poseidon_sbox_fx(u64 x) {  // one-thread-per-state permutation only (kernels with >= 124 VGPRs to their name)
  const u64 x2 = gl_mul_nc_fx(x, x), x4 = gl_mul_nc_fx(x2, x2), x3 = gl_mul_nc_fx(x, x2);
  return gl_mul_nc_fx(x3, x4);
}
GLD u64 poseidon_sbox(u64 x) { return gl_canon(poseidon_sbox_nc(x)); }

// al + ah * 2^32 (mod p) for al, ah < 2^42 — the fold at the end of an MDS row — as SOME u64 representative.
//   ah * 2^32 = ah_lo * 2^32 + ah_hi * 2^64 = ah_lo * 2^32 + ah_hi * EPS  (mod p);  X = ah_hi * EPS + al < 2^43 cannot
//   overflow (one v_mad_u64_u32);  adding ah_lo * 2^32 only touches the high dword; if that carries, the wrapped
//   value is < 2^43 and the 2^64 it lost is worth EPS, which cannot overflow again.  5 VALU instructions, against
//   ~15 for the generic 128-bit route (shift, add, carry detect, gl_reduce128_nc).
GLD u64 mds_fold_nc(u64 al, u64 ah) {
  const u32 eps = 0xFFFFFFFFu;
  u64 X, cX;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(X), "=s"(cX) : "v"((u32)(ah >> 32)), "v"(eps), "v"(al));
  u32 h, d, l2, h2;
  u64 c, c2, c3;
  asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(h), "=s"(c) : "v"((u32)(X >> 32)), "v"((u32)ah));
  asm("v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(d) : "s"(c));
  asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(l2), "=s"(c2) : "v"((u32)X), "v"(d));
  asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(h2), "=s"(c3) : "v"(h), "s"(c2));
  return gl_pack(l2, h2);
}

// The same fold for accumulators of ANY size below 2^64 (the final layer of a block of 4 partial rounds reaches 2^63.8):
//   X = ah_hi * EPS + al may carry (cX), adding ah_lo * 2^32 may carry (c2); each lost 2^64 is worth EPS, so
//   Y = (cX + c2) * EPS + wrapped value — one more v_mad_u64_u32 — whose own carry is fixed by a last + EPS (the wrapped
//   Y is < 2^34 then, so that cannot overflow).  8 VALU instructions.
GLD u64 mds_fold_wide_nc(u64 al, u64 ah) {
  const u32 eps = 0xFFFFFFFFu;
  u64 X, cX, c2, cj, Y, c3;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(X), "=s"(cX) : "v"((u32)(ah >> 32)), "v"(eps), "v"(al));
  u32 h, k, d;
  asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(h), "=s"(c2) : "v"((u32)(X >> 32)), "v"((u32)ah));
  asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(k) : "s"(c2));
  asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(k), "=s"(cj) : "v"(k), "s"(cX));
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(Y), "=s"(c3) : "v"(k), "v"(eps), "v"(gl_pack((u32)X, h)));
  asm("v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(d) : "s"(c3));
  return Y + (u64)d;
}

// Dense MDS layer: out[r] = sum_i CIRC[i] * v[(i+r) % 12] + DIAG[r]*v[r]; entries < 2^6, so the low and
// high 32-bit halves are accumulated separately (< 2^42 each) and folded once:  lo + hi*2^32 (mod p).
GLD void poseidon_mds_nc(u64 (&s)[12]) {
  const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  u32 lo[12], hi[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    lo[i] = (u32)s[i];
    hi[i] = (u32)(s[i] >> 32);
  }
#pragma unroll
  for (int r = 0; r < 12; ++r) {
    u64 al = 0, ah = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      al += (u64)C[i] * lo[(i + r) % 12];
      ah += (u64)C[i] * hi[(i + r) % 12];
    }
    if (r == 0) {
      al += (u64)8 * lo[0];
      ah += (u64)8 * hi[0];
    }
    s[r] = mds_fold_nc(al, ah);  // al, ah < 2^42
  }
}
// canonical-in / canonical-out wrapper used by the quotient kernel's PoseidonGate evaluation
GLD void poseidon_mds(u64 (&s)[12]) {
  poseidon_mds_nc(s);
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = gl_canon(s[i]);
}

// ---- dot products with full-size constants, carry-free -----------------------------------------------------------
// sum_i a_i * b_i with a_i any u64 and b_i a 64-bit constant.  a = a0 + 2^32 a1, so a*b = a0*b + a1*b' with the
// second constant b' = 2^32 b mod p; both constants are pre-split into limbs of 22 + 22 + 20 bits.  Limb k of b
// meets a0 and limb k of b' meets a1 in the SAME accumulator S_k: <= 26 products below 2^54 each, so a plain 64-bit
// register and ONE v_mad_u64_u32 per product suffice — no carries anywhere — and the result is
// S_0 + 2^22 S_1 + 2^44 S_2 < 2^104, recombined and reduced once per dot product.  6 multiply-adds per term instead
// of a 64x64->128 product (4 multiplies + 4 pair-forming moves + 1 add) followed by a 5-instruction carry chain.
struct Limbs3x2 {
  u32 lo[3];  // limbs of b      (multiply the low  half of the state element)
  u32 hi[3];  // limbs of 2^32 b (multiply the high half)
};
template <int R, int C>
struct Limbs3Table {
  Limbs3x2 v[R][C];
};
constexpr u64 gl_mul_2_32_const(u64 b) { return (u64)((((unsigned __int128)b) << 32) % (unsigned __int128)GL_P); }
template <int R, int C>
constexpr Limbs3Table<R, C> make_limbs3(const u64 (&raw)[R][C]) {
  Limbs3Table<R, C> t{};
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c) {
      const u64 b = raw[r][c] % GL_P, bh = gl_mul_2_32_const(b);
      t.v[r][c].lo[0] = (u32)(b & 0x3FFFFFu);
      t.v[r][c].lo[1] = (u32)((b >> 22) & 0x3FFFFFu);
      t.v[r][c].lo[2] = (u32)(b >> 44);
      t.v[r][c].hi[0] = (u32)(bh & 0x3FFFFFu);
      t.v[r][c].hi[1] = (u32)((bh >> 22) & 0x3FFFFFu);
      t.v[r][c].hi[2] = (u32)(bh >> 44);
    }
  return t;
}
constexpr Limbs3x2 make_limbs3x2(u64 b) {
  Limbs3x2 t{};
  const u64 bh = gl_mul_2_32_const(b);
  t.lo[0] = (u32)(b & 0x3FFFFFu), t.lo[1] = (u32)((b >> 22) & 0x3FFFFFu), t.lo[2] = (u32)(b >> 44);
  t.hi[0] = (u32)(bh & 0x3FFFFFu), t.hi[1] = (u32)((bh >> 22) & 0x3FFFFFu), t.hi[2] = (u32)(bh >> 44);
  return t;
}
struct dot3 {
  u64 s0, s1, s2;
};
GLD void dot3_mac(dot3& D, u64 a, const Limbs3x2& b) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32);
  D.s0 += (u64)a0 * b.lo[0];
  D.s1 += (u64)a0 * b.lo[1];
  D.s2 += (u64)a0 * b.lo[2];
  D.s0 += (u64)a1 * b.hi[0];
  D.s1 += (u64)a1 * b.hi[1];
  D.s2 += (u64)a1 * b.hi[2];
}
GLD u64 dot3_reduce_nc(const dot3& D) {
  typedef unsigned __int128 u128;
  const u128 V = (u128)D.s0 + ((u128)D.s1 << 22) + ((u128)D.s2 << 44);  // < 2^104
  return gl_reduce128_nc((u64)V, (u64)(V >> 64));
}
GLD u64 dot3_reduce_add_nc(const dot3& D, u64 addend) {
  typedef unsigned __int128 u128;
  const u128 V = (u128)D.s0 + ((u128)D.s1 << 22) + ((u128)D.s2 << 44) + (u128)addend;
  return gl_reduce128_nc((u64)V, (u64)(V >> 64));
}

// ---- the 22 partial rounds in INTEGER-POWER BLOCKS ---------------------------------------------------------------
// Only lane 0 is non-linear in a partial round, and the MDS entries are tiny (< 2^6), so a block of B rounds is a
// handful of dot products with SMALL INTEGER coefficients — 32-bit multipliers, two v_mad_u64_u32 per term on the
// 32-bit halves of a lane, no carries, one 5-instruction fold per dot product — instead of B dense layers (or the
// "fast" sparse form, whose full-size constants cost six multiply-adds per term):
//   M = MDS (integers), Q = M with row 0 zeroed;  the block starts from u (round constants of its first round already
//   added), w = (y_0, u_1 .. u_11) with y_0 = u_0^7;
//   S-box input of round j:   x_j = row0(M) Q^(j-1) . w  +  sum_{1<=i<j} y_i (row0(M) Q^(j-1-i))[0]  + kappa_j
//   state after the block:    u'  = M Q^(B-1) w  +  sum_{1<=i<B} y_i (M Q^(B-1-i)) e_0  +  K
// (lane 0 is REPLACED by the S-box output before each layer, hence Q: no subtraction appears).  kappa_j and K collect the
// round constants (K also those of the round that follows the block) and ride in as the accumulators' initial values.
// Schedule (PoseidonGate evaluation; the hashing permutation uses POSEIDON_SCHED_H below): 5 blocks of 4 + 1 block of 2.  B = 4 is the longest block whose coefficients still fit 32-bit multipliers
// (entries of M Q^3 < 2^28.3) and whose accumulators still fit 64 bits (row sums incl. the y terms < 2^31.72 — checked by
// static_assert below); the final layer's accumulators then reach 2^63.8 and are folded by mds_fold_wide_nc.  B = 5 would
// need 37-bit coefficients.
// Equivalence to the naive rounds is exact integer algebra mod p; tests: the permutation KATs, 2.4 M iterated
// permutations against the oracle, every Merkle / proof parity test.
#define POSEIDON_NBLOCKS 6
#define POSEIDON_BLOCK_B 4
constexpr int POSEIDON_SCHED[POSEIDON_NBLOCKS] = {4, 4, 4, 4, 4, 2};
struct PoseidonIntBlock {
  u32 A[4][12];   // A[j][i]: coefficient of w_i in x_j   (1 <= j < B)
  u32 b[4][4];    // b[j][i]: coefficient of y_i in x_j   (1 <= i < j)
  u32 C[12][12];  // M Q^(B-1)
  u32 c[4][12];   // c[i][r]: coefficient of y_i in u'_r  (1 <= i < B)
  u64 max_row_sum;  // max over rows of sum_i C[r][i] + sum_i c[i][r]: bounds the accumulators
};
struct PoseidonMat {
  u64 m[12][12];
};
constexpr PoseidonMat poseidon_mds_int() {
  const u64 CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  PoseidonMat M{};
  for (int r = 0; r < 12; ++r)
    for (int i = 0; i < 12; ++i) M.m[r][(i + r) % 12] += CIRC[i];
  M.m[0][0] += 8;
  return M;
}
constexpr PoseidonMat poseidon_mat_mul(const PoseidonMat& X, const PoseidonMat& Y) {  // exact integers (entries stay < 2^40)
  PoseidonMat R{};
  for (int i = 0; i < 12; ++i)
    for (int j = 0; j < 12; ++j)
      for (int t = 0; t < 12; ++t) R.m[i][j] += X.m[i][t] * Y.m[t][j];
  return R;
}
constexpr PoseidonMat poseidon_q_pow(int k) {
  PoseidonMat Q = poseidon_mds_int(), R{};
  for (int j = 0; j < 12; ++j) Q.m[0][j] = 0;
  for (int i = 0; i < 12; ++i) R.m[i][i] = 1;
  for (int t = 0; t < k; ++t) R = poseidon_mat_mul(Q, R);
  return R;
}
constexpr PoseidonIntBlock make_int_block(int B) {
  PoseidonIntBlock T{};
  const PoseidonMat M = poseidon_mds_int();
  for (int j = 1; j < B; ++j) {
    const PoseidonMat X = poseidon_mat_mul(M, poseidon_q_pow(j - 1));
    for (int i = 0; i < 12; ++i) T.A[j][i] = (u32)X.m[0][i];
    for (int i = 1; i < j; ++i) T.b[j][i] = (u32)poseidon_mat_mul(M, poseidon_q_pow(j - 1 - i)).m[0][0];
  }
  const PoseidonMat CC = poseidon_mat_mul(M, poseidon_q_pow(B - 1));
  for (int r = 0; r < 12; ++r)
    for (int i = 0; i < 12; ++i) T.C[r][i] = (u32)CC.m[r][i];
  for (int i = 1; i < B; ++i) {
    const PoseidonMat X = poseidon_mat_mul(M, poseidon_q_pow(B - 1 - i));
    for (int r = 0; r < 12; ++r) T.c[i][r] = (u32)X.m[r][0];
  }
  T.max_row_sum = 0;
  for (int r = 0; r < 12; ++r) {
    u64 sum = 0;
    for (int i = 0; i < 12; ++i) sum += CC.m[r][i];
    for (int i = 1; i < B; ++i) sum += T.c[i][r];
    if (sum > T.max_row_sum) T.max_row_sum = sum;
  }
  for (int r = 0; r < 12; ++r)
    for (int i = 0; i < 12; ++i)
      if (CC.m[r][i] >> 32) T.max_row_sum = ~(u64)0;  // a coefficient that does not fit a 32-bit multiplier: unusable block length
  return T;
}
// every accumulator is sum_i c_i * (32-bit half) + (32-bit constant half) <= max_row_sum * (2^32 - 1) + 2^32 - 1 < 2^64
static_assert(make_int_block(POSEIDON_BLOCK_B).max_row_sum < ((u64)1 << 32) - 1, "block too long: the accumulators would overflow 64 bits");
static_assert(make_int_block(2).max_row_sum < ((u64)1 << 25), "block of 2: simple fold range");
// Round constants of the blocks: kappa[blk][j] (S-box input of the block's round j >= 1) and K[blk][r] (state after the
// block, INCLUDING the constants of the round that follows it).
struct PoseidonBlockConsts {
  u64 kappa[POSEIDON_NBLOCKS][4];
  u64 K[POSEIDON_NBLOCKS][12];
  u64 Ksplit[POSEIDON_NBLOCKS][12][2];  // {low dword, high dword} of K: accumulator seeds for the asm MDS rows
};
constexpr u64 gl_mulmod_const(u64 a, u64 b) { return (u64)(((unsigned __int128)a * b) % (unsigned __int128)GL_P); }
constexpr u64 POSEIDON_RC_RAW[VX_POSEIDON_N_ROUND_CONSTANTS] = VX_POSEIDON_ROUND_CONSTANTS_INIT;
constexpr u64 gl_addmod_const(u64 a, u64 b) { return (u64)(((unsigned __int128)a + b) % (unsigned __int128)GL_P); }
// `sched` / `r0`: the block lengths and the round the first block starts in (4 = the first partial round; 3 for the hashing schedule
// below, whose first block opens with the dense layer of full round 3)
constexpr PoseidonBlockConsts make_block_consts_g(const int (&sched)[POSEIDON_NBLOCKS], int r0) {
  PoseidonBlockConsts R{};
  const PoseidonMat M = poseidon_mds_int();
  for (int blk = 0; blk < POSEIDON_NBLOCKS; ++blk) {
    const int B = sched[blk];
    u64 cv[12] = {};  // constant part of w^(j) (the block's first-round constants are already inside u)
    for (int j = 1; j <= B; ++j) {
      u64 t[12] = {};
      for (int i = 0; i < 12; ++i) {
        u64 acc = 0;
        for (int q = 0; q < 12; ++q) acc = gl_addmod_const(acc, gl_mulmod_const(M.m[i][q], cv[q]));
        t[i] = gl_addmod_const(acc, POSEIDON_RC_RAW[(r0 + j) * 12 + i] % GL_P);  // r0 + B <= 26: always a real round
      }
      if (j < B) {
        R.kappa[blk][j] = t[0];
        cv[0] = 0;
        for (int i = 1; i < 12; ++i) cv[i] = t[i];
      } else {
        for (int i = 0; i < 12; ++i) {
          R.K[blk][i] = t[i];
          R.Ksplit[blk][i][0] = t[i] & 0xFFFFFFFFull;
          R.Ksplit[blk][i][1] = t[i] >> 32;
        }
      }
    }
    r0 += B;
  }
  return R;
}
static_assert(POSEIDON_SCHED[0] + POSEIDON_SCHED[1] + POSEIDON_SCHED[2] + POSEIDON_SCHED[3] + POSEIDON_SCHED[4] + POSEIDON_SCHED[5] == 22,
              "the blocks must cover the 22 partial rounds");
constexpr PoseidonBlockConsts make_block_consts() { return make_block_consts_g(POSEIDON_SCHED, 4); }
__constant__ PoseidonBlockConsts POSEIDON_BLK = make_block_consts();
// THE HASHING SCHEDULE (round 4): the dense layer that ends full round 3 is itself "a partial round whose lane 0 has already been
// through its S-box" — so it opens the first block instead of standing alone: block 0 = that layer + partial rounds 4, 5, 6 (a block of
// 4 whose first S-box is the identity, same coefficient matrices M Q^0..3), then 4 + 4 + 4 + 4 + 3 for rounds 7 .. 25.  One dense
// layer and its twelve folds (348 instructions) leave the permutation, and the odd block is a 3 instead of a 2: 3494 -> 3201 linear-layer
// instructions over the 22 partial rounds + that layer, 2.3 % of the permutation.  The PoseidonGate evaluation (plonk_kernels.hip.h) keeps
// the schedule above: its wires sit between the layer and the first partial round's S-box either way, and it is 4 % of a proof.
constexpr int POSEIDON_SCHED_H[POSEIDON_NBLOCKS] = {4, 4, 4, 4, 4, 3};
static_assert(POSEIDON_SCHED_H[0] + POSEIDON_SCHED_H[1] + POSEIDON_SCHED_H[2] + POSEIDON_SCHED_H[3] + POSEIDON_SCHED_H[4] + POSEIDON_SCHED_H[5] == 23,
              "the hashing blocks must cover the dense layer of round 3 and the 22 partial rounds");
static_assert(make_int_block(3).max_row_sum < ((u64)1 << 25), "block of 3: simple fold range (accumulators < 2^57)");
__constant__ PoseidonBlockConsts POSEIDON_BLK_H = make_block_consts_g(POSEIDON_SCHED_H, 3);
// round constants with one all-zero round appended, so "the constants of the next round" exists after round 29 too; stored
// pre-split — {low dword, high dword} as two u64 — because each half seeds its own 64-bit accumulator: a scalar load puts
// it into an SGPR pair that is the 64-bit ADDEND of the row's first v_mad_u64_u32 (no instruction spent on the constant)
struct PoseidonRcExt {
  u64 v[31 * 12];
  u64 split[31 * 12][2];
};
constexpr PoseidonRcExt make_rc_ext() {
  PoseidonRcExt R{};
  for (int i = 0; i < 360; ++i) {
    R.v[i] = POSEIDON_RC_RAW[i] % GL_P;
    R.split[i][0] = R.v[i] & 0xFFFFFFFFull;
    R.split[i][1] = R.v[i] >> 32;
  }
  return R;
}
__constant__ PoseidonRcExt POSEIDON_RC_EXT = make_rc_ext();

GLD void poseidon_mac32(u64& al, u64& ah, u64 x, u32 c) {
  al += (u64)(u32)x * c;
  ah += (u64)(u32)(x >> 32) * c;
}
// Dense MDS layer with the NEXT round's constants folded in: each row is one asm block (poseidon_mds_asm.inc, generated) —
// 24 v_mad_u64_u32 whose first pair takes the pre-split constant as its 64-bit addend straight from an SGPR pair, and
// the 5-instruction fold.  (Left to hipcc, the row sum is re-associated: the constant costs a v_lshl_add_u64 per accumulator
// and the x16 / x2 terms become shift-adds with pair-forming moves.)  `round_next` indexes POSEIDON_RC_EXT (30 = all-zero).
#include "poseidon_mds_asm.inc"
GLD void poseidon_mds_rc_nc(u64 (&s)[12], int round_next) {
  u32 lo[12], hi[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    lo[i] = (u32)s[i];
    hi[i] = (u32)(s[i] >> 32);
  }
  poseidon_mds_rows_asm(s, lo, hi, POSEIDON_RC_EXT.split + round_next * 12);
}
// One block of B partial rounds (see above).  `sbox(x)` is the S-box; `lane0(j, x)` is handed the S-box INPUT x of the
// block's round j and returns the value that actually goes through the S-box: the identity for the permutation; the
// quotient kernel's PoseidonGate passes the wire the gate constrains to equal x (and pushes x - wire as a constraint) —
// the recurrences are linear in everything but the S-box outputs, so that is exactly the gate's eval_unfiltered.
// FIRST_ID: the block opens with a dense layer whose input lane 0 is used as it is (the hashing schedule's block 0: every lane of `s` has
// just been through the S-boxes of full round 3).
template <int B, bool FIRST_ID = false, class SBOX, class LANE0>
GLD void poseidon_partial_block_g(u64 (&s)[12], const u64* __restrict__ kappa, const u64* __restrict__ K, SBOX&& sbox, LANE0&& lane0) {
  constexpr PoseidonIntBlock T = make_int_block(B);
  u64 y[B];
  if constexpr (FIRST_ID)
    y[0] = s[0];
  else
    y[0] = sbox(lane0(0, s[0]));
  s[0] = y[0];  // s is now w = (y_0, u_1 .. u_11)
#pragma unroll
  for (int j = 1; j < B; ++j) {
    const u64 k = kappa[j];
    u64 al = (u32)k, ah = k >> 32;
#pragma unroll
    for (int i = 0; i < 12; ++i) poseidon_mac32(al, ah, s[i], T.A[j][i]);
#pragma unroll
    for (int i = 1; i < j; ++i) poseidon_mac32(al, ah, y[i], T.b[j][i]);
    y[j] = sbox(lane0(j, mds_fold_nc(al, ah)));
  }
  u64 out[12];
#pragma unroll
  for (int r = 0; r < 12; ++r) {
    const u64 k = K[r];
    u64 al = (u32)k, ah = k >> 32;
#pragma unroll
    for (int i = 0; i < 12; ++i) poseidon_mac32(al, ah, s[i], T.C[r][i]);
#pragma unroll
    for (int i = 1; i < B; ++i) poseidon_mac32(al, ah, y[i], T.c[i][r]);
    // B <= 3: al, ah < 2^57 and the 5-instruction fold applies; B = 4: up to 2^63.8, the fold must survive two carries
    out[r] = B >= 4 ? mds_fold_wide_nc(al, ah) : mds_fold_nc(al, ah);
  }
#pragma unroll
  for (int r = 0; r < 12; ++r) s[r] = out[r];
}
template <int B, bool FIRST_ID = false>
GLD void poseidon_partial_block_nc(u64 (&s)[12], const u64* __restrict__ kappa, const u64* __restrict__ K) {
  auto sb = [](u64 x) { return poseidon_sbox_fx(x); };
  auto id = [](int, u64 x) { return x; };
  poseidon_partial_block_g<B, FIRST_ID>(s, kappa, K, sb, id);
}
// full round 3's dense layer + the 22 partial rounds, hashing schedule; `s` = the state right after the S-boxes of round 3
GLD void poseidon_partial_rounds_h_nc(u64 (&s)[12]) {
  poseidon_partial_block_nc<POSEIDON_BLOCK_B, true>(s, POSEIDON_BLK_H.kappa[0], POSEIDON_BLK_H.K[0]);
#pragma unroll 1
  for (int blk = 1; blk < POSEIDON_NBLOCKS - 1; ++blk) poseidon_partial_block_nc<POSEIDON_BLOCK_B>(s, POSEIDON_BLK_H.kappa[blk], POSEIDON_BLK_H.K[blk]);
  poseidon_partial_block_nc<3>(s, POSEIDON_BLK_H.kappa[POSEIDON_NBLOCKS - 1], POSEIDON_BLK_H.K[POSEIDON_NBLOCKS - 1]);
}

// The permutation WITHOUT its last dense layer: everything up to and including the S-boxes of round 29.  A sponge decides which rows of
// that layer it needs (below); poseidon_permute_nc = this + all twelve.
GLD void poseidon_permute_body_nc(u64 (&s)[12]) {
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = gl_add_nc_c(s[i], POSEIDON_RC_EXT.v[i]);
#pragma unroll 1
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = poseidon_sbox_fx(s[i]);
    poseidon_mds_rc_nc(s, r + 1);
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = poseidon_sbox_fx(s[i]);   // round 3; its dense layer opens the first block
  poseidon_partial_rounds_h_nc(s);
#pragma unroll 1
  for (int r = 26; r < 29; ++r) {
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = poseidon_sbox_fx(s[i]);
    poseidon_mds_rc_nc(s, r + 1);
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = poseidon_sbox_fx(s[i]);   // round 29
}
// Permutation on arbitrary-u64 lanes; outputs are arbitrary u64 representatives (NOT canonical).
GLD void poseidon_permute_nc(u64 (&s)[12]) {
  poseidon_permute_body_nc(s);
  poseidon_mds_rc_nc(s, 30);
}
// THE OVERWRITE-MODE SPONGE ONLY EVER READS PART OF A PERMUTATION'S OUTPUT (round 4): when the next chunk is full its eight words
// REPLACE lanes 0..7, so of the last dense layer only the capacity rows 8..11 are live — 8 rows x 29 instructions less, on every
// permutation of a leaf but its last two; the last permutation hands out lanes 0..3 only (rows4 below, as in two_to_one).
// s[0..7] are garbage afterwards.
GLD void poseidon_last_layer_capacity_rows_nc(u64 (&s)[12]) {
  u32 lo[12], hi[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    lo[i] = (u32)s[i];
    hi[i] = (u32)(s[i] >> 32);
  }
  const u64(*__restrict__ rc)[2] = POSEIDON_RC_EXT.split + 30 * 12;   // the all-zero "round 30"
  s[8] = poseidon_mds_row8_asm(lo, hi, rc[8][0], rc[8][1]);
  s[9] = poseidon_mds_row9_asm(lo, hi, rc[9][0], rc[9][1]);
  s[10] = poseidon_mds_row10_asm(lo, hi, rc[10][0], rc[10][1]);
  s[11] = poseidon_mds_row11_asm(lo, hi, rc[11][0], rc[11][1]);
}

// The permutation of PoseidonHash::two_to_one — lanes 8..11 enter as ZERO and only lanes 0..3 leave — with the two things that
// buys: in round 0 the four capacity lanes hold their round constants, so their S-box outputs are the compile-time constants
// rc^7 (4 of 12 S-boxes gone), and the last dense layer only needs its first four rows (8 of 12 rows gone): 488 of the
// 12,800 instructions, on the 5·10^7 node permutations of a proof.  s[8..11] are ignored on entry; s[4..11] are garbage on exit.
constexpr u64 gl_pow7_const(u64 x) {
  const u64 x2 = gl_mulmod_const(x, x), x4 = gl_mulmod_const(x2, x2), x3 = gl_mulmod_const(x2, x);
  return gl_mulmod_const(x3, x4);
}
struct PoseidonCapSbox {
  u64 v[4];
};
constexpr PoseidonCapSbox make_cap_sbox() {
  PoseidonCapSbox r{};
  for (int i = 0; i < 4; ++i) r.v[i] = gl_pow7_const(POSEIDON_RC_RAW[8 + i] % GL_P);
  return r;
}
__constant__ PoseidonCapSbox POSEIDON_CAP_SBOX = make_cap_sbox();
GLD void poseidon_mds_rows4_rc_nc(u64 (&s)[12], int round_next) {
  u32 lo[12], hi[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    lo[i] = (u32)s[i];
    hi[i] = (u32)(s[i] >> 32);
  }
  const u64(*__restrict__ rc)[2] = POSEIDON_RC_EXT.split + round_next * 12;
  s[0] = poseidon_mds_row0_asm(lo, hi, rc[0][0], rc[0][1]);
  s[1] = poseidon_mds_row1_asm(lo, hi, rc[1][0], rc[1][1]);
  s[2] = poseidon_mds_row2_asm(lo, hi, rc[2][0], rc[2][1]);
  s[3] = poseidon_mds_row3_asm(lo, hi, rc[3][0], rc[3][1]);
}
// One absorption step of hash_n_to_m_no_pad's overwrite-mode sponge after the chunk has been written into lanes 0..7: the permutation
// with exactly the rows of its last layer that stay live.  `next_full`: another chunk of eight follows; `last`: this was the final
// chunk (only the digest lanes 0..3 leave).  Both wave-uniform.
GLD void poseidon_sponge_step_nc(u64 (&s)[12], bool next_full, bool last) {
  poseidon_permute_body_nc(s);
  if (next_full)
    poseidon_last_layer_capacity_rows_nc(s);
  else if (last)
    poseidon_mds_rows4_rc_nc(s, 30);
  else
    poseidon_mds_rc_nc(s, 30);
}
// (Round 3 tried two savings inside the leaf sponge — first permutation of a leaf: capacity lanes zero, last one: four lanes out —
// under wave-uniform flags measured NO gain on hash_leaves_colmajor_kernel, 125.65 / 125.98 ms against 125.70 / 125.95: the
// branches cost what the 488 instructions per leaf save.  Not kept; profiles/r03_hash_experiments.md.)
GLD void poseidon_two_to_one_permute_nc(u64 (&s)[12]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = gl_add_nc_c(s[i], POSEIDON_RC_EXT.v[i]);
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = poseidon_sbox_fx(s[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) s[8 + i] = POSEIDON_CAP_SBOX.v[i];   // (0 + rc)^7
  poseidon_mds_rc_nc(s, 1);
#pragma unroll 1
  for (int r = 1; r < 3; ++r) {
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = poseidon_sbox_fx(s[i]);
    poseidon_mds_rc_nc(s, r + 1);
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = poseidon_sbox_fx(s[i]);   // round 3
  poseidon_partial_rounds_h_nc(s);
#pragma unroll 1
  for (int r = 26; r < 29; ++r) {
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = poseidon_sbox_fx(s[i]);
    poseidon_mds_rc_nc(s, r + 1);
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = poseidon_sbox_fx(s[i]);
  poseidon_mds_rows4_rc_nc(s, 30);
}

GLD void poseidon_permute(u64 (&s)[12]) {
  poseidon_permute_nc(s);
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = gl_canon(s[i]);
}

// PoseidonHash::two_to_one (hashing.rs): state = [l0..l3, r0..r3, 0,0,0,0] -> permute -> [0..4]
GLD void poseidon_two_to_one(const u64* l, const u64* r, u64* out) {
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s[i] = l[i];
    s[4 + i] = r[i];
    s[8 + i] = 0;
  }
  poseidon_permute_nc(s);
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = gl_canon(s[i]);
}

// ------------------------------------------------------------------------------------------------
// Lane-cooperative permutation for LATENCY-bound launches (the small levels of a Merkle tree, small FRI layers):
// 16 lanes per state, lane g < 12 holds s[g] (lanes 12..15 of a group shadow lanes 0..3 and are ignored), so one
// wavefront carries 4 states and a permutation is ~30 x (S-box + 12 cross-lane multiply-adds) = ~4k instructions
// deep instead of ~22k.  Below ~16k states a launch of the one-thread-per-state form costs one full permutation
// latency (~60 us) whatever its size; this form costs ~1/5 of that.  Plain (non-"fast") partial rounds: every lane
// adds its round constant, lane 0 takes the S-box, then the same cross-lane MDS as in the full rounds.
// ------------------------------------------------------------------------------------------------
GLD u64 poseidon_coop_mds_nc(u64 v, int g, int group_base) {
  const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  const u32 lo = (u32)v, hi = (u32)(v >> 32);
  u64 al = 0, ah = 0;
  int idx = g < 12 ? g : g - 12;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const int src = group_base + idx;
    al += (u64)C[i] * (u32)__shfl((int)lo, src, 64);
    ah += (u64)C[i] * (u32)__shfl((int)hi, src, 64);
    idx = idx == 11 ? 0 : idx + 1;
  }
  if (g == 0) {
    al += (u64)8 * lo;
    ah += (u64)8 * hi;
  }
  return mds_fold_nc(al, ah);
}
// v: this lane's state element (any u64 representative); returns the permuted element (NOT canonical).
GLD u64 poseidon_permute_coop_nc(u64 v, int g, int group_base) {
  const int gi = g < 12 ? g : g - 12;
#pragma unroll 1
  for (int r = 0; r < 30; ++r) {
    v = gl_add_nc_c(v, POSEIDON_RC[r * 12 + gi]);
    const bool full = r < 4 || r >= 26;
    if (full || g == 0) v = poseidon_sbox_nc(v);
    v = poseidon_coop_mds_nc(v, g, group_base);
  }
  return v;
}

// The same permutation with the round constants in LDS (`rc`: the 360 constants, copied there once per workgroup) and the next
// round's constant fetched while the current round computes: in the fused tree-top kernel (merkle.hip.h) a workgroup walks up to
// 11 levels = 11 dependent permutations, and 30 per-lane constant loads from global memory per permutation were pure latency.
GLD u64 poseidon_permute_coop_lds_nc(u64 v, int g, int group_base, const u64* rc) {
  const int gi = g < 12 ? g : g - 12;
  u64 k = rc[gi];
#pragma unroll 1
  for (int r = 0; r < 30; ++r) {
    const u64 kn = rc[(r < 29 ? r + 1 : r) * 12 + gi];
    v = gl_add_nc_c(v, k);
    const bool full = r < 4 || r >= 26;
    if (full || g == 0) v = poseidon_sbox_nc(v);
    v = poseidon_coop_mds_nc(v, g, group_base);
    k = kn;
  }
  return v;
}
