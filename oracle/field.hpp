// ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the plonky2 v0.2.0 prover arithmetic.
// Nothing under vectorx_amd/ may include, link or call this; only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg use it (as the checker / reported baseline, never the product).
//
// PARITY STATUS: "parity unpinned" for everything except the Poseidon permutation (pinned by the
// known-answer vectors of SURVEY.md Appendix B.2) and the field constants (Appendix B.3).  The
// upstream source (0xPolygonZero/plonky2 @ 7445ec91, tag v0.2.0) is NOT vendored in /root/reference
// (Cargo.lock:4848-4905 names it as a git dependency) and there is no Rust toolchain, so every
// function below follows the *published algorithm* as restated in SURVEY.md Appendix A and cites the
// upstream file it restates by path (no line numbers can be checked).
//
// field.hpp restates  field/src/goldilocks_field.rs, field/src/goldilocks_extensions.rs,
//                     field/src/extension/quadratic.rs            
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>
#include <cassert>

namespace vxo {

typedef uint64_t u64;
typedef unsigned __int128 u128;

static const u64 P = 0xFFFFFFFF00000001ULL;
static const u64 EPS = 0xFFFFFFFFULL;  // 2^64 mod p
static const u64 MULTIPLICATIVE_GENERATOR = 7;
static const u64 POWER_OF_TWO_GENERATOR = 1753635133440165772ULL;  // 7^((p-1)/2^32)
static const int TWO_ADICITY = 32;

// All values are kept CANONICAL (< p) in the oracle; plonky2 allows non-canonical u64 internally
// but canonicalises on serialisation, so canonical-everywhere is output-equivalent.
// (round 6: a carry / borrow that happens every other time on random data is a MASK — as a branch it mispredicted half the time and made
//  a field multiplication cost ~7 ns; one that happens with probability 2^-32 stays a branch, which the predictor gets right for free.
//  The textbook forms made the oracle a 1.5x slower CPU baseline than the same algorithm deserves.)
static inline u64 canon(u64 x) { return __builtin_expect(x >= P, 0) ? x - P : x; }
static inline u64 add(u64 a, u64 b) {
  u64 s;
  const u64 c = __builtin_add_overflow(a, b, &s);
  s += EPS & (0 - c);  // + 2^64 mod p: a + b - p < p, canonical; without a carry s < 2^64 can still reach p (rarely)
  return canon(s);
}
static inline u64 sub(u64 a, u64 b) {
  u64 d;
  const u64 bw = __builtin_sub_overflow(a, b, &d);
  return d + (P & (0 - bw));   // a < b: a - b + p (mod 2^64)
}
static inline u64 neg(u64 a) { return a ? P - a : 0; }
// goldilocks_field.rs::reduce128
static inline u64 reduce128(u128 x) {
  u64 lo = (u64)x, hi = (u64)(x >> 64);
  u64 hh = hi >> 32, hl = hi & EPS;
  u64 t = lo - hh;
  if (__builtin_expect(lo < hh, 0)) t -= EPS;  // borrow (lo < 2^32: once in 2^32): subtract 2^64 mod p
  u64 m = hl * EPS;                            // < 2^64
  u64 r;
  const u64 c = __builtin_add_overflow(t, m, &r);
  r += EPS & (0 - c);
  return canon(r);
}
static inline u64 mul(u64 a, u64 b) { return reduce128((u128)a * b); }
static inline u64 sqr(u64 a) { return mul(a, a); }
static inline u64 pow(u64 b, u64 e) {
  u64 r = 1;
  while (e) {
    if (e & 1) r = mul(r, b);
    b = sqr(b);
    e >>= 1;
  }
  return r;
}
static inline u64 inv(u64 a) { return pow(a, P - 2); }  // a != 0
static inline u64 from_i64(long long v) { return v >= 0 ? (u64)v % P : P - ((u64)(-v) % P); }
// primitive_root_of_unity(k): g^(2^(32-k))
static inline u64 root_of_unity(int log_n) {
  assert(log_n <= TWO_ADICITY);
  u64 g = POWER_OF_TWO_GENERATOR;
  for (int i = log_n; i < TWO_ADICITY; ++i) g = sqr(g);
  return g;
}

// ---------------------------------------------------------------------------------------------
// Quadratic extension F_p[X]/(X^2 - 7)   (goldilocks_extensions.rs: W = 7, DTH_ROOT = p-1)
// ---------------------------------------------------------------------------------------------
struct Ext {
  u64 a, b;  // a + b*X
  Ext() : a(0), b(0) {}
  Ext(u64 a_, u64 b_ = 0) : a(a_), b(b_) {}
  bool operator==(const Ext& o) const { return a == o.a && b == o.b; }
  bool operator!=(const Ext& o) const { return !(*this == o); }
};
static const u64 EXT_W = 7;
static inline Ext operator+(Ext x, Ext y) { return Ext(add(x.a, y.a), add(x.b, y.b)); }
static inline Ext operator-(Ext x, Ext y) { return Ext(sub(x.a, y.a), sub(x.b, y.b)); }
static inline Ext operator-(Ext x) { return Ext(neg(x.a), neg(x.b)); }
static inline Ext operator*(Ext x, Ext y) {
  return Ext(add(mul(x.a, y.a), mul(EXT_W, mul(x.b, y.b))), add(mul(x.a, y.b), mul(x.b, y.a)));
}
static inline Ext scale(Ext x, u64 s) { return Ext(mul(x.a, s), mul(x.b, s)); }
static inline Ext ext_inv(Ext x) {
  // 1/(a+bX) = (a - bX)/(a^2 - 7 b^2)
  u64 d = sub(sqr(x.a), mul(EXT_W, sqr(x.b)));
  u64 di = inv(d);
  return Ext(mul(x.a, di), mul(neg(x.b), di));
}
static inline Ext ext_pow(Ext b, u64 e) {
  Ext r(1);
  while (e) {
    if (e & 1) r = r * b;
    b = b * b;
    e >>= 1;
  }
  return r;
}

// Thin wrapper so gate evaluators can be written once over {Fp, Ext}.
struct Fp {
  u64 v;
  Fp() : v(0) {}
  Fp(u64 x) : v(x) {}
  bool operator==(const Fp& o) const { return v == o.v; }
  bool operator!=(const Fp& o) const { return v != o.v; }
};
static inline Fp operator+(Fp x, Fp y) { return Fp(add(x.v, y.v)); }
static inline Fp operator-(Fp x, Fp y) { return Fp(sub(x.v, y.v)); }
static inline Fp operator-(Fp x) { return Fp(neg(x.v)); }
static inline Fp operator*(Fp x, Fp y) { return Fp(mul(x.v, y.v)); }
static inline Fp scale(Fp x, u64 s) { return Fp(mul(x.v, s)); }

// ---------------------------------------------------------------------------------------------
// util helpers (plonky2_util: log2_strict, reverse_bits, reverse_index_bits_in_place)
// ---------------------------------------------------------------------------------------------
static inline int log2_strict(size_t n) {
  int l = 0;
  while (((size_t)1 << l) < n) ++l;
  assert(((size_t)1 << l) == n);
  return l;
}
static inline size_t reverse_bits(size_t x, int bits) {
  size_t r = 0;
  for (int i = 0; i < bits; ++i) {
    r = (r << 1) | (x & 1);
    x >>= 1;
  }
  return r;
}
template <class T>
static inline void reverse_index_bits_in_place(std::vector<T>& v) {
  int lb = log2_strict(v.size());
  for (size_t i = 0; i < v.size(); ++i) {
    size_t j = reverse_bits(i, lb);
    if (i < j) std::swap(v[i], v[j]);
  }
}
// Montgomery batch inversion (plonky2_field::batch_util / Field::batch_multiplicative_inverse)
static inline void batch_inverse(u64* x, size_t n) {
  if (!n) return;
  std::vector<u64> pre(n);
  u64 acc = 1;
  for (size_t i = 0; i < n; ++i) {
    pre[i] = acc;
    acc = mul(acc, x[i]);
  }
  u64 ia = inv(acc);
  for (size_t i = n; i-- > 0;) {
    u64 xi = x[i];
    x[i] = mul(ia, pre[i]);
    ia = mul(ia, xi);
  }
}

}  // namespace vxo
