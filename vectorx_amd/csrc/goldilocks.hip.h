// Goldilocks field F_p, p = 2^64 - 2^32 + 1, and its quadratic extension F_p[X]/(X^2-7), for gfx950.
// Replaces plonky2_field::goldilocks_field / goldilocks_extensions (plonky2 v0.2.0, un-vendored git
// dependency named at /root/reference/Cargo.lock:4848-4905; algorithm per SURVEY.md A.1).
// CDNA4 has no 64x64->128 multiply: a field mul is a 4-limb v_mad_u64_u32 product plus the
// 2^64 = 2^32-1, 2^96 = -1 fold.  All functions take and return CANONICAL values (< p).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint64_t u64;
typedef uint32_t u32;

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL

#define GLD __device__ __forceinline__


// ---- instruction-count-minimal building blocks (gfx950: every VOP3 / multiply-class instruction costs
// about the same issue time, so fewer instructions is the only lever; see tools/ubench_int.hip).
// Carries travel in SGPR pairs (lane masks) produced / consumed by the *_co instructions, which hipcc does
// not use on its own for 64-bit compare-and-fix sequences.
GLD u64 gl_pack(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }
// x + d (mod 2^64) for a 32-bit d held in one VGPR: v_mad_u64_u32 with the inline multiplier 1 takes the 32-bit
// operand as it is — the 64-bit add forms need it widened into an aligned register pair (a zero and often a move)
GLD u64 gl_add32(u64 x, u32 d) {
  u64 r, c;
  asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(r), "=s"(c) : "v"(d), "v"(x));
  return r;
}

// x - p = x + EPS (mod 2^64): one compare, one select of the 32-bit addend, one 64-bit add
GLD u64 gl_canon(u64 x) {
  u32 d0;
  asm("v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(d0) : "s"(__builtin_amdgcn_uicmpl(x, 0xFFFFFFFF00000001ULL, 35 /* ICMP_UGE */)));
  return gl_add32(x, d0);
}

// a, b canonical  ->  (a + b) mod p, canonical.   S = a+b in [0, 2p-2];  S >= p  <=>  a carry out of
// (a+b) or of (a+b)+EPS;  in that case the answer is (S + EPS) mod 2^64.
GLD u64 gl_add(u64 a, u64 b) {
  u32 s0, s1, d0;
  u64 c0, c1;
  asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(s0), "=s"(c0) : "v"((u32)a), "v"((u32)b));
  asm("v_addc_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(s1), "=s"(c1) : "v"((u32)(a >> 32)), "v"((u32)(b >> 32)), "s"(c0));
  const u64 s = gl_pack(s0, s1);
  // S >= p  <=>  carry out, or no carry and s >= p; either way the answer is s + EPS (mod 2^64): 5 VALU
  const u64 m = c1 | __builtin_amdgcn_uicmpl(s, (u64)GL_P, 35 /* ICMP_UGE */);
  asm("v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(d0) : "s"(m));
  return gl_add32(s, d0);
}
// a, b canonical -> (a - b) mod p, canonical: on a borrow the wrapped difference plus p (= minus EPS mod 2^64) is the
// answer; the borrow of the subtraction itself selects it (5 VALU; hipcc's form re-compares the operands: 6)
GLD u64 gl_sub(u64 a, u64 b) {
  u32 d0, d1, m, r0, r1;
  u64 b0, bw, b2, b3;
  asm("v_sub_co_u32_e64 %0, %1, %2, %3" : "=v"(d0), "=s"(b0) : "v"((u32)a), "v"((u32)b));
  asm("v_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(d1), "=s"(bw) : "v"((u32)(a >> 32)), "v"((u32)(b >> 32)), "s"(b0));
  asm("v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(m) : "s"(bw));
  asm("v_sub_co_u32_e64 %0, %1, %2, %3" : "=v"(r0), "=s"(b2) : "v"(d0), "v"(m));
  asm("v_subbrev_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(r1), "=s"(b3) : "v"(d1), "s"(b2));
  return gl_pack(r0, r1);
}
GLD u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }
GLD u64 gl_dbl(u64 a) { return gl_add(a, a); }

// 64 x 64 -> 128 with four v_mad_u64_u32; the 32-bit addends are small enough that no step can carry.
GLD void gl_mul128(u64 a, u64 b, u64& lo, u64& hi) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  const u64 p00 = (u64)a0 * b0;
  const u64 p01 = (u64)a0 * b1 + (p00 >> 32);
  const u64 p10 = (u64)a1 * b0 + (u32)p01;
  hi = (u64)a1 * b1 + (p01 >> 32) + (p10 >> 32);
  lo = (p10 << 32) | (u32)p00;
}
// hi*2^64 + lo  ->  SOME u64 congruent to it mod p (not necessarily canonical).
//   x = lo + hl*EPS - hh   (hi = hh*2^32 + hl, 2^64 = EPS, 2^96 = -1)
//   (T, cT) = hl*EPS + lo  in one v_mad_u64_u32 with carry-out;  u = T - hh with borrow bw;
//   result = u + (cT - bw)*EPS  (mod 2^64): the four (cT, bw) cases are exact — when both are set the wrapped
//   u already equals T + 2^64 - hh, a valid representative because 2^64 = EPS (mod p).
// CANON: the canonical value costs ONE more instruction, not a separate conditional subtraction: with a carry
// (cT) the corrected value is already < p (T_wrapped < (2^32-1)^2, so T_wrapped - hh + EPS < p when there is no
// borrow, and T_wrapped + EPS - hh < EPS when there is); with a borrow alone it is in (p - 2^32, p); only the
// no-carry-no-borrow case can leave u >= p, and subtracting p is adding EPS (mod 2^64) — the same addend as the
// carry case, so the compare just widens that lane mask.
template <bool CANON>
GLD u64 gl_fix_reduce(u32 u0, u32 u1, u64 cT, u64 bw) {
  const u64 u = gl_pack(u0, u1);
  u64 m1 = cT & ~bw;
  const u64 m2 = bw & ~cT;  // +EPS  /  -EPS (= + 0xFFFFFFFF00000001)
  if (CANON) m1 = cT | (__builtin_amdgcn_uicmpl(u, (u64)GL_P, 35 /* ICMP_UGE */) & ~bw);
  u32 d0, d1;
  asm("v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(d0) : "s"(m1));
  asm("v_cndmask_b32_e64 %0, %1, 1, %2" : "=v"(d0) : "v"(d0), "s"(m2));
  asm("v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(d1) : "s"(m2));
  const u64 r = gl_add32(u, d0);
  u32 r1;  // the high dword of the addend only touches the high dword of the sum: a plain 32-bit add
  asm("v_add_u32_e32 %0, %1, %2" : "=v"(r1) : "v"((u32)(r >> 32)), "v"(d1));
  return gl_pack((u32)r, r1);
}
template <bool CANON>
GLD u64 gl_reduce128_t(u64 lo, u64 hi) {
  const u32 hh = (u32)(hi >> 32), hl = (u32)hi;
  u64 T, cT, b0, bw;
  const u32 eps = 0xFFFFFFFFu;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(T), "=s"(cT) : "v"(hl), "v"(eps), "v"(lo));
  u32 u0, u1;
  asm("v_sub_co_u32_e64 %0, %1, %2, %3" : "=v"(u0), "=s"(b0) : "v"((u32)T), "v"(hh));
  asm("v_subbrev_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(u1), "=s"(bw) : "v"((u32)(T >> 32)), "s"(b0));
  return gl_fix_reduce<CANON>(u0, u1, cT, bw);
}
GLD u64 gl_reduce128_nc(u64 lo, u64 hi) { return gl_reduce128_t<false>(lo, hi); }
// x = hi*2^64 + lo  ->  x mod p (canonical)
GLD u64 gl_reduce128(u64 lo, u64 hi) { return gl_reduce128_t<true>(lo, hi); }
// Fused multiply(-add)-reduce, 16 (18) VALU instructions, no register-pair shuffling:
//   a*b (+c) = p00 + (p10 + cM*2^64)*2^32 + p11*2^64   with p10 = a1*b0 + a0*b1 (carry-out cM),
//   lo64 = p00 + (p10 mod 2^32)*2^32,  hi64 = p11 + (p10 >> 32) + carries = hh*2^32 + hl,
//   result = lo64 + hl*EPS - (hh + cM)       (2^64 = EPS, 2^96 = -1 mod p; cM rides in as the borrow-in
//   of the final subtraction), fixed up exactly as in gl_reduce128_nc.
template <bool WITH_ADDEND, bool CANON>
GLD u64 gl_mulmad_t(u64 a, u64 b, u64 c) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u64 p00, c0 = 0;
  if (WITH_ADDEND)
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(p00), "=s"(c0) : "v"(a0), "v"(b0), "v"(c));
  else
    p00 = (u64)a0 * b0;
  const u64 p01 = (u64)a0 * b1;
  const u64 p11 = (u64)a1 * b1;
  u64 p10, cM;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(p10), "=s"(cM) : "v"(a1), "v"(b0), "v"(p01));
  u32 lo1, hl, hh;
  u64 cL, cH, cX;
  asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(lo1), "=s"(cL) : "v"((u32)(p00 >> 32)), "v"((u32)p10));
  asm("v_addc_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(hl), "=s"(cH) : "v"((u32)p11), "v"((u32)(p10 >> 32)), "s"(cL));
  asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(hh), "=s"(cX) : "v"((u32)(p11 >> 32)), "s"(cH));
  if (WITH_ADDEND) {  // the addend's carry out of p00 is worth 2^64: one more unit of hi64
    u64 cH2, cX2;
    asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(hl), "=s"(cH2) : "v"(hl), "s"(c0));
    asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(hh), "=s"(cX2) : "v"(hh), "s"(cH2));
  }
  const u64 lo = gl_pack((u32)p00, lo1);
  u64 T, cT, bb, bw;
  const u32 eps = 0xFFFFFFFFu;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(T), "=s"(cT) : "v"(hl), "v"(eps), "v"(lo));
  u32 u0, u1;
  asm("v_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(u0), "=s"(bb) : "v"((u32)T), "v"(hh), "s"(cM));
  asm("v_subbrev_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(u1), "=s"(bw) : "v"((u32)(T >> 32)), "s"(bb));
  return gl_fix_reduce<CANON>(u0, u1, cT, bw);
}
GLD u64 gl_mul_nc(u64 a, u64 b) { return gl_mulmad_t<false, false>(a, b, 0); }
GLD u64 gl_mul(u64 a, u64 b) { return gl_mulmad_t<false, true>(a, b, 0); }
GLD u64 gl_sqr(u64 a) { return gl_mul(a, a); }
// a*b + c (any u64 representatives) -> some representative
GLD u64 gl_mad_nc(u64 a, u64 b, u64 c) { return gl_mulmad_t<true, false>(a, b, c); }
// a*b + c, canonical result
GLD u64 gl_mad(u64 a, u64 b, u64 c) { return gl_mulmad_t<true, true>(a, b, c); }
// x * c for a SMALL constant c < 2^31 (the 7 of every F_p^2 product, the MDS coefficients, the base of a BaseSumGate: a third of the
// multiplications of a recursion circuit's gate programs) -> some representative.  x c = p1 2^32 + lo32(p0) with p1 < 2^63:
// lo64 + h 2^64 = lo64 + h EPS, one carry fix: 5-6 VALU instead of the 16 of a general multiply-reduce.
GLD u64 gl_mulc_tail(u64 lo, u32 h) {
  u64 T, cT;
  const u32 eps = 0xFFFFFFFFu;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(T), "=s"(cT) : "v"(h), "v"(eps), "v"(lo));
  u32 d0;
  asm("v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(d0) : "s"(cT));
  return gl_add32(T, d0);   // a carry is worth 2^64 = EPS; T_wrapped < h EPS <= 2^63, so adding EPS cannot carry again
}
GLD u64 gl_mulc_nc(u64 x, u32 c) {
  const u64 p0 = (u64)(u32)x * c;
  const u64 p1 = (u64)(u32)(x >> 32) * c + (p0 >> 32);
  return gl_mulc_tail(gl_pack((u32)p0, (u32)p1), (u32)(p1 >> 32));
}
// x * c + a (a: any representative) for c < 2^31 -> some representative
GLD u64 gl_madc_nc(u64 x, u32 c, u64 a) {
  u64 p0, c0, cz;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(p0), "=s"(c0) : "v"((u32)x), "v"(c), "v"(a));   // the carry is worth 2^64: one more unit of h
  const u64 p1 = (u64)(u32)(x >> 32) * c + (p0 >> 32);
  u32 h;
  asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(h), "=s"(cz) : "v"((u32)(p1 >> 32)), "s"(c0));   // h <= 2^31 - 1 + 1: no overflow
  return gl_mulc_tail(gl_pack((u32)p0, (u32)p1), h);
}
GLD u64 gl_pow(u64 b, u64 e) {
  u64 r = 1;
  while (e) {
    if (e & 1) r = gl_mul(r, b);
    b = gl_sqr(b);
    e >>= 1;
  }
  return r;
}
// a^(p-2) by an addition chain over the 2^32-structured exponent (p-2 = 2^64 - 2^32 - 1).
GLD u64 gl_inv(u64 a) {
  // t_k = a^(2^k - 1)
  u64 t2 = gl_mul(gl_sqr(a), a);            // 2 bits
  u64 t3 = gl_mul(gl_sqr(t2), a);           // 3
  u64 t6 = t3;
  for (int i = 0; i < 3; ++i) t6 = gl_sqr(t6);
  t6 = gl_mul(t6, t3);                      // 6
  u64 t12 = t6;
  for (int i = 0; i < 6; ++i) t12 = gl_sqr(t12);
  t12 = gl_mul(t12, t6);                    // 12
  u64 t24 = t12;
  for (int i = 0; i < 12; ++i) t24 = gl_sqr(t24);
  t24 = gl_mul(t24, t12);                   // 24
  u64 t30 = t24;
  for (int i = 0; i < 6; ++i) t30 = gl_sqr(t30);
  t30 = gl_mul(t30, t6);                    // 30
  u64 t31 = gl_mul(gl_sqr(t30), a);         // 31
  u64 t32 = gl_mul(gl_sqr(t31), a);         // 32
  // p-2 = (2^32-1)*2^32 - 1 ... = bits: 32 ones, then a zero, then 31 ones  => 0xFFFFFFFF_00000000 - 1
  // p - 2 = 0xFFFFFFFEFFFFFFFF = [31 ones][0][32 ones]
  u64 r = t31;
  r = gl_sqr(r);                            // append the 0 bit
  for (int i = 0; i < 32; ++i) r = gl_sqr(r);
  return gl_mul(r, t32);
}

struct ext2 {
  u64 a, b;
};
GLD ext2 ext_make(u64 a, u64 b) {
  ext2 r;
  r.a = a;
  r.b = b;
  return r;
}
GLD ext2 ext_add(ext2 x, ext2 y) { return ext_make(gl_add(x.a, y.a), gl_add(x.b, y.b)); }
GLD ext2 ext_sub(ext2 x, ext2 y) { return ext_make(gl_sub(x.a, y.a), gl_sub(x.b, y.b)); }
GLD ext2 ext_neg(ext2 x) { return ext_make(gl_neg(x.a), gl_neg(x.b)); }
GLD u64 gl_mul7(u64 x) {
  // 7x < 2^67
  u64 lo = x * 7, hi = __umul64hi(x, 7);
  return gl_reduce128(lo, hi);
}
GLD ext2 ext_mul(ext2 x, ext2 y) {
  u64 bb = gl_mul(x.b, y.b);
  return ext_make(gl_mad(x.a, y.a, gl_mul7(bb)), gl_mad(x.a, y.b, gl_mul(x.b, y.a)));
}
GLD ext2 ext_scale(ext2 x, u64 s) { return ext_make(gl_mul(x.a, s), gl_mul(x.b, s)); }
GLD ext2 ext_inv(ext2 x) {
  u64 d = gl_sub(gl_sqr(x.a), gl_mul7(gl_sqr(x.b)));
  u64 di = gl_inv(d);
  return ext_make(gl_mul(x.a, di), gl_mul(gl_neg(x.b), di));
}
