// Poseidon-Goldilocks permutation (width 12, x^7, 8 full + 22 partial rounds) for gfx950.
// Replaces plonky2::hash::poseidon / poseidon_goldilocks and the AVX2/NEON hand-scheduled forms
// (plonky2 v0.2.0, un-vendored: /root/reference/Cargo.lock:4848-4905; algorithm per SURVEY.md A.2).
//
// This path is bound by the integer ALU, not HBM (measured: tools/ubench_int.hip — on gfx950 every
// multiply-class / VOP3 instruction, v_mad_u64_u32 included, issues at ~4.4-5.2 cycles per wavefront, a
// plain v_add_u32 at ~2.8), so the design minimises the INSTRUCTION COUNT per permutation:
//  * lanes are kept as arbitrary u64 representatives (not canonical) between operations; one
//    conditional subtraction at the very end canonicalises;
//  * the 22 partial rounds use the "fast" sparse form (one dense 11x11 matrix up front, then per round a
//    12-term dot product and 11 multiply-adds instead of a dense 12x12 MDS) — constants re-derived and
//    proven equivalent in tools/gen_poseidon_fast_constants.py (the role of upstream's FAST_PARTIAL_*);
//  * the dense MDS of the 8 full rounds multiplies the low/high 32-bit halves by the <2^6 circulant
//    entries with v_mad_u64_u32 (a 32x32+64 multiply-accumulate in ONE instruction) and folds once.
// The 12-lane state lives in VGPRs; round constants sit in constant memory and, because the loops
// are unrolled, are fetched with scalar loads shared by the whole wavefront.
#pragma once
#include "goldilocks.hip.h"
#include "poseidon_constants.h"
#include "poseidon_fast_constants.h"

__constant__ u64 POSEIDON_RC[VX_POSEIDON_N_ROUND_CONSTANTS] = VX_POSEIDON_ROUND_CONSTANTS_INIT;
__constant__ u64 POSEIDON_FAST_FIRST[12] = VX_FAST_PARTIAL_FIRST_ROUND_CONSTANT_INIT;
__constant__ u64 POSEIDON_FAST_K[22] = VX_FAST_PARTIAL_ROUND_CONSTANTS_INIT;
__constant__ u64 POSEIDON_FAST_INIT[11][11] = VX_FAST_PARTIAL_INITIAL_MATRIX_INIT;
__constant__ u64 POSEIDON_FAST_W_HATS[22][11] = VX_FAST_PARTIAL_W_HATS_INIT;
__constant__ u64 POSEIDON_FAST_VS[22][11] = VX_FAST_PARTIAL_VS_INIT;

#define POSEIDON_WIDTH 12
#define POSEIDON_RATE 8

// ---- non-canonical ("nc") arithmetic: values are any u64 congruent to the field element ----------
GLD void gl_mul128(u64 a, u64 b, u64& lo, u64& hi) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  const u64 p00 = (u64)a0 * b0;
  const u64 p01 = (u64)a0 * b1 + (p00 >> 32);
  const u64 p10 = (u64)a1 * b0 + (u32)p01;
  hi = (u64)a1 * b1 + (p01 >> 32) + (p10 >> 32);
  lo = (p10 << 32) | (u32)p00;
}
// hi*2^64 + lo  ->  some u64 congruent mod p
GLD u64 gl_reduce128_nc(u64 lo, u64 hi) {
  const u64 hh = hi >> 32, hl = hi & GL_EPS;
  u64 t = lo - hh;
  if (lo < hh) t -= GL_EPS;
  const u64 m = (hl << 32) - hl;
  u64 r = t + m;
  if (r < m) r += GL_EPS;
  return r;
}
GLD u64 gl_mul_nc(u64 a, u64 b) {
  u64 lo, hi;
  gl_mul128(a, b, lo, hi);
  return gl_reduce128_nc(lo, hi);
}
// a: any u64, b: CANONICAL (< p)  ->  a + b (nc).  One carry fix suffices because b <= 2^64 - 2^32.
GLD u64 gl_add_nc_c(u64 a, u64 b) {
  u64 s = a + b;
  if (s < a) s += GL_EPS;
  return s;
}
// a*b + c (all nc)
GLD u64 gl_mad_nc(u64 a, u64 b, u64 c) {
  u64 lo, hi;
  gl_mul128(a, b, lo, hi);
  const u64 l2 = lo + c;
  hi += (l2 < lo);  // hi <= 2^64 - 2: no overflow
  return gl_reduce128_nc(l2, hi);
}

GLD u64 poseidon_sbox_nc(u64 x) {
  const u64 x2 = gl_mul_nc(x, x), x4 = gl_mul_nc(x2, x2), x3 = gl_mul_nc(x, x2);
  return gl_mul_nc(x3, x4);
}
GLD u64 poseidon_sbox(u64 x) { return gl_canon(poseidon_sbox_nc(x)); }

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
    // value = al + ah*2^32, al, ah < 2^42
    const u64 l = al + (ah << 32);
    const u64 h = (ah >> 32) + (l < al ? 1 : 0);
    s[r] = gl_reduce128_nc(l, h);
  }
}
// canonical-in / canonical-out wrapper used by the quotient kernel's PoseidonGate evaluation
GLD void poseidon_mds(u64 (&s)[12]) {
  poseidon_mds_nc(s);
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = gl_canon(s[i]);
}

// 192-bit accumulator for dot products with full-size constants
struct acc192 {
  u64 lo, hi;
  u32 c;
};
GLD void acc_mul_add(acc192& A, u64 a, u64 b) {
  u64 lo, hi;
  gl_mul128(a, b, lo, hi);
  const u64 l2 = A.lo + lo;
  const u64 carry = l2 < lo;
  const u64 h1 = A.hi + hi;
  const u32 c1 = h1 < hi;
  const u64 h2 = h1 + carry;
  const u32 c2 = h2 < carry;
  A.lo = l2;
  A.hi = h2;
  A.c += c1 + c2;
}
// lo + hi*2^64 + c*2^128,  2^128 = -2^32 (mod p)
GLD u64 acc_reduce_nc(const acc192& A) {
  u64 r = gl_reduce128_nc(A.lo, A.hi);
  const u64 sub = (u64)A.c << 32;  // < 2^37
  const u64 d = r - sub;
  return r < sub ? d - GL_EPS : d;  // wrapped d + p
}

// Permutation on arbitrary-u64 lanes; outputs are arbitrary u64 representatives (NOT canonical).
GLD void poseidon_permute_nc(u64 (&s)[12]) {
#pragma unroll 1
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = poseidon_sbox_nc(gl_add_nc_c(s[i], POSEIDON_RC[r * 12 + i]));
    poseidon_mds_nc(s);
  }
  // ---- partial rounds, fast form ----
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = gl_add_nc_c(s[i], POSEIDON_FAST_FIRST[i]);
  {
    u64 t[11];
#pragma unroll 1
    for (int r = 0; r < 11; ++r) {
      acc192 A = {0, 0, 0};
#pragma unroll
      for (int c = 0; c < 11; ++c) acc_mul_add(A, s[1 + c], POSEIDON_FAST_INIT[r][c]);
      t[r] = acc_reduce_nc(A);
    }
#pragma unroll
    for (int r = 0; r < 11; ++r) s[1 + r] = t[r];
  }
#pragma unroll 1
  for (int r = 0; r < 22; ++r) {
    const u64 s0 = gl_add_nc_c(poseidon_sbox_nc(s[0]), POSEIDON_FAST_K[r]);
    acc192 A = {0, 0, 0};
    acc_mul_add(A, s0, 25);  // M[0][0] = CIRC[0] + DIAG[0]
#pragma unroll
    for (int i = 0; i < 11; ++i) acc_mul_add(A, s[1 + i], POSEIDON_FAST_W_HATS[r][i]);
#pragma unroll
    for (int i = 0; i < 11; ++i) s[1 + i] = gl_mad_nc(s0, POSEIDON_FAST_VS[r][i], s[1 + i]);
    s[0] = acc_reduce_nc(A);
  }
#pragma unroll 1
  for (int r = 26; r < 30; ++r) {
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = poseidon_sbox_nc(gl_add_nc_c(s[i], POSEIDON_RC[r * 12 + i]));
    poseidon_mds_nc(s);
  }
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
