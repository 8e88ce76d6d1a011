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

GLD u64 gl_canon(u64 x) { return x >= GL_P ? x - GL_P : x; }
GLD u64 gl_add(u64 a, u64 b) {
  u64 s = a + b;
  if (s < a) s += GL_EPS;
  return gl_canon(s);
}
GLD u64 gl_sub(u64 a, u64 b) {
  u64 d = a - b;
  return a < b ? d + GL_P : d;  // wrapped d + p == a - b + p (mod 2^64)
}
GLD u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }
GLD u64 gl_dbl(u64 a) { return gl_add(a, a); }
// x = hi*2^64 + lo  ->  x mod p (canonical)
GLD u64 gl_reduce128(u64 lo, u64 hi) {
  u64 hh = hi >> 32, hl = hi & GL_EPS;
  u64 t = lo - hh;
  if (lo < hh) t -= GL_EPS;
  u64 m = (hl << 32) - hl;  // hl * (2^32-1)
  u64 r = t + m;
  if (r < m) r += GL_EPS;
  return gl_canon(r);
}
GLD u64 gl_mul(u64 a, u64 b) { return gl_reduce128(a * b, __umul64hi(a, b)); }
GLD u64 gl_sqr(u64 a) { return gl_mul(a, a); }
// a*b + c, c canonical
GLD u64 gl_mad(u64 a, u64 b, u64 c) {
  u64 lo = a * b, hi = __umul64hi(a, b);
  u64 l2 = lo + c;
  hi += (l2 < lo);
  return gl_reduce128(l2, hi);
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
