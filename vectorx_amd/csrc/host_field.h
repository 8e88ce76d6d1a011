// Host-side Goldilocks / Poseidon used by the PRODUCT library for the tiny sequential pieces that stay
// on the CPU: twiddle/shift table construction and the Fiat-Shamir challenger (plonky2::iop::challenger,
// SURVEY.md A.6).  This is deliberately independent of oracle/ (which is test infrastructure).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include "poseidon_constants.h"

namespace vxh {
typedef uint64_t u64;
typedef unsigned __int128 u128;
static const u64 P = 0xFFFFFFFF00000001ULL;
static const u64 EPS = 0xFFFFFFFFULL;

// The challenger is ~130 DEPENDENT permutations per plonky2 proof and ~1 000 per wide STARK table on one host core, inside every
// proof's wall time.  A carry that happens every other time on random data (the sum of two field elements, the low word plus hl * EPS)
// is a MASK — as a branch it mispredicted half the time —, one that happens once in 2^32 (the borrow of lo - hh, a result >= p) stays a
// predicted branch: 6.7 -> 3.3 us per permutation on the build machine (round 6).
static inline u64 canon(u64 x) { return __builtin_expect(x >= P, 0) ? x - P : x; }
static inline u64 add(u64 a, u64 b) {
  u64 s;
  const u64 c = __builtin_add_overflow(a, b, &s);
  s += EPS & (0 - c);
  return canon(s);
}
static inline u64 sub(u64 a, u64 b) {
  u64 d;
  const u64 bw = __builtin_sub_overflow(a, b, &d);
  return d + (P & (0 - bw));
}
static inline u64 neg(u64 a) { return a ? P - a : 0; }
static inline u64 reduce128(u128 x) {
  u64 lo = (u64)x, hi = (u64)(x >> 64);
  u64 hh = hi >> 32, hl = hi & EPS;
  u64 t = lo - hh;
  if (__builtin_expect(lo < hh, 0)) t -= EPS;
  u64 m = hl * EPS;
  u64 r;
  const u64 c = __builtin_add_overflow(t, m, &r);
  r += EPS & (0 - c);
  return canon(r);
}
static inline u64 mul(u64 a, u64 b) { return reduce128((u128)a * b); }
static inline u64 pow(u64 b, u64 e) {
  u64 r = 1;
  while (e) {
    if (e & 1) r = mul(r, b);
    b = mul(b, b);
    e >>= 1;
  }
  return r;
}
static inline u64 inv(u64 a) { return pow(a, P - 2); }
static const u64 POWER_OF_TWO_GENERATOR = 1753635133440165772ULL;
static inline u64 root_of_unity(int log_n) {
  u64 g = POWER_OF_TWO_GENERATOR;
  for (int i = log_n; i < 32; ++i) g = mul(g, g);
  return g;
}
static inline size_t reverse_bits(size_t x, int bits) {
  size_t r = 0;
  for (int i = 0; i < bits; ++i) {
    r = (r << 1) | (x & 1);
    x >>= 1;
  }
  return r;
}

struct Ext {
  u64 a, b;
};
static inline Ext emul(Ext x, Ext y) {
  return Ext{add(mul(x.a, y.a), mul(7, mul(x.b, y.b))), add(mul(x.a, y.b), mul(x.b, y.a))};
}
static inline Ext eadd(Ext x, Ext y) { return Ext{add(x.a, y.a), add(x.b, y.b)}; }
static inline Ext esub(Ext x, Ext y) { return Ext{sub(x.a, y.a), sub(x.b, y.b)}; }
static inline Ext einv(Ext x) {
  u64 d = sub(mul(x.a, x.a), mul(7, mul(x.b, x.b)));
  u64 di = inv(d);
  return Ext{mul(x.a, di), mul(neg(x.b), di)};
}
static inline Ext epow(Ext b, u64 e) {
  Ext r{1, 0};
  while (e) {
    if (e & 1) r = emul(r, b);
    b = emul(b, b);
    e >>= 1;
  }
  return r;
}

// Poseidon permutation (host copy, used only by the challenger and public-input hash).
static const u64 RC[VX_POSEIDON_N_ROUND_CONSTANTS] = VX_POSEIDON_ROUND_CONSTANTS_INIT;
static inline u64 sbox(u64 x) {
  u64 x2 = mul(x, x), x4 = mul(x2, x2), x3 = mul(x, x2);
  return mul(x3, x4);
}
// The linear layer on the 32-bit halves of the lanes (the circulant's entries are < 2^6, so twelve products of a half fit 64 bits
// without carries), four output lanes per AVX2 register; 2x the scalar loop below, which stays as the portable path.
static inline u64 mds_fold(u64 al, u64 ah) { return reduce128((u128)al + ((u128)ah << 32)); }   // al + 2^32 ah, both < 2^42
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
}  // namespace vxh
#include <immintrin.h>
namespace vxh {
__attribute__((target("avx2"))) static inline __m256i vx_ltu64(__m256i a, __m256i b) {   // a < b, unsigned, per 64-bit lane
  const __m256i sign = _mm256_set1_epi64x((long long)0x8000000000000000ULL);
  return _mm256_cmpgt_epi64(_mm256_xor_si256(b, sign), _mm256_xor_si256(a, sign));
}
__attribute__((target("avx2"))) static inline void mds_layer_avx2(u64* s) {
  static const u64 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  alignas(32) u64 lo[24], hi[24];
  const __m256i eps = _mm256_set1_epi64x((long long)EPS), pp = _mm256_set1_epi64x((long long)P);
  for (int i = 0; i < 12; i += 4) {
    const __m256i v = _mm256_loadu_si256((const __m256i*)(s + i));
    const __m256i l = _mm256_and_si256(v, eps), h = _mm256_srli_epi64(v, 32);
    _mm256_store_si256((__m256i*)(lo + i), l);
    _mm256_store_si256((__m256i*)(lo + i + 12), l);
    _mm256_store_si256((__m256i*)(hi + i), h);
    _mm256_store_si256((__m256i*)(hi + i + 12), h);
  }
  for (int k = 0; k < 12; k += 4) {
    __m256i al = _mm256_setzero_si256(), ah = _mm256_setzero_si256();
    for (int i = 0; i < 12; ++i) {
      const __m256i c = _mm256_set1_epi64x((long long)C[i]);
      al = _mm256_add_epi64(al, _mm256_mul_epu32(c, _mm256_loadu_si256((const __m256i*)(lo + i + k))));
      ah = _mm256_add_epi64(ah, _mm256_mul_epu32(c, _mm256_loadu_si256((const __m256i*)(hi + i + k))));
    }
    if (k == 0) {   // the diagonal's extra 8 on lane 0
      al = _mm256_add_epi64(al, _mm256_set_epi64x(0, 0, 0, (long long)(8 * lo[0])));
      ah = _mm256_add_epi64(ah, _mm256_set_epi64x(0, 0, 0, (long long)(8 * hi[0])));
    }
    // al + 2^32 ah  (both < 2^42)  =  t + 2^64 (ah >> 32 + carry),  2^64 = EPS (mod p)
    const __m256i yl = _mm256_slli_epi64(ah, 32);
    __m256i t = _mm256_add_epi64(al, yl);
    const __m256i carry = _mm256_srli_epi64(vx_ltu64(t, yl), 63);
    const __m256i m = _mm256_mul_epu32(_mm256_add_epi64(_mm256_srli_epi64(ah, 32), carry), eps);   // < 2^43
    __m256i r = _mm256_add_epi64(t, m);
    r = _mm256_add_epi64(r, _mm256_and_si256(vx_ltu64(r, m), eps));                                  // wrapped: + 2^64 = EPS
    r = _mm256_sub_epi64(r, _mm256_andnot_si256(vx_ltu64(r, pp), pp));                               // canonical
    _mm256_storeu_si256((__m256i*)(s + k), r);
  }
}
#define VXH_HAVE_AVX2_MDS 1
#endif
static inline void mds_layer(u64* s) {
  static const u64 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  u64 o[12];
  for (int k = 0; k < 12; ++k) {
    u128 acc = 0;
    for (int i = 0; i < 12; ++i) acc += (u128)C[i] * s[(i + k) % 12];
    if (k == 0) acc += (u128)8 * s[0];
    o[k] = reduce128(acc);
  }
  for (int i = 0; i < 12; ++i) s[i] = o[i];
}
static inline void poseidon(u64* s) {
#ifdef VXH_HAVE_AVX2_MDS
  static const bool avx2 = __builtin_cpu_supports("avx2");
#endif
  int rc = 0;
  for (int r = 0; r < 30; ++r) {
    for (int i = 0; i < 12; ++i) s[i] = add(s[i], RC[rc++]);
    if (r < 4 || r >= 26)
      for (int i = 0; i < 12; ++i) s[i] = sbox(s[i]);
    else
      s[0] = sbox(s[0]);
#ifdef VXH_HAVE_AVX2_MDS
    if (avx2) {
      mds_layer_avx2(s);
      continue;
    }
#endif
    mds_layer(s);
  }
}
}  // namespace vxh
