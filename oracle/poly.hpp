// ORACLE — TEST INFRASTRUCTURE ONLY (see field.hpp header for the full notice and parity status).
// Restates  field/src/fft.rs, field/src/polynomial/mod.rs  and  plonky2/src/fri/oracle.rs
// (PolynomialBatch::from_values / from_coeffs / lde_values / get_lde_values) of plonky2 v0.2.0.
// SURVEY.md A.3, A.5.  Exact field arithmetic => any correct algorithm is bit-identical; only
// the INDEX ORDER conventions matter and they are stated at each function.
#pragma once
#include "poseidon.hpp"

namespace vxo {

// fft.rs::fft — natural order in, natural order out:  out[k] = sum_j c[j] * g^(j k)
static inline void fft_inplace(u64* a, int log_n) {
  size_t n = (size_t)1 << log_n;
  for (size_t i = 0; i < n; ++i) {
    size_t j = reverse_bits(i, log_n);
    if (i < j) std::swap(a[i], a[j]);
  }
  for (int s = 1; s <= log_n; ++s) {
    size_t m = (size_t)1 << s, h = m >> 1;
    u64 wm = root_of_unity(s);
    std::vector<u64> tw(h);
    tw[0] = 1;
    for (size_t i = 1; i < h; ++i) tw[i] = mul(tw[i - 1], wm);
    for (size_t k = 0; k < n; k += m)
      for (size_t j = 0; j < h; ++j) {
        u64 t = mul(tw[j], a[k + j + h]);
        u64 u = a[k + j];
        a[k + j] = add(u, t);
        a[k + j + h] = sub(u, t);
      }
  }
}
// fft.rs::ifft — c[j] = (1/n) sum_k v[k] g^(-j k)   (upstream: fft, scale by 1/n, reverse [1..])
static inline void ifft_inplace(u64* a, int log_n) {
  size_t n = (size_t)1 << log_n;
  fft_inplace(a, log_n);
  u64 ninv = inv((u64)n % P);
  for (size_t i = 0; i < n; ++i) a[i] = mul(a[i], ninv);
  for (size_t i = 1; i < n - i; ++i) std::swap(a[i], a[n - i]);
}
// polynomial/mod.rs::coset_fft(shift): c[j] *= shift^j, then fft
static inline void coset_fft_inplace(u64* a, int log_n, u64 shift) {
  size_t n = (size_t)1 << log_n;
  u64 p = 1;
  for (size_t i = 0; i < n; ++i) {
    a[i] = mul(a[i], p);
    p = mul(p, shift);
  }
  fft_inplace(a, log_n);
}
// polynomial/mod.rs::coset_ifft(shift): ifft, then c[j] *= shift^-j
static inline void coset_ifft_inplace(u64* a, int log_n, u64 shift) {
  size_t n = (size_t)1 << log_n;
  ifft_inplace(a, log_n);
  u64 si = inv(shift), p = 1;
  for (size_t i = 0; i < n; ++i) {
    a[i] = mul(a[i], p);
    p = mul(p, si);
  }
}
// Extension-field transforms: the roots and the shift are base-field, so an F_p^2 NTT is two
// independent F_p NTTs on the components (polynomial/mod.rs is generic over the field).
static inline void coset_fft_ext_inplace(std::vector<Ext>& v, int log_n, u64 shift) {
  size_t n = (size_t)1 << log_n;
  std::vector<u64> a(n), b(n);
  for (size_t i = 0; i < n; ++i) a[i] = v[i].a, b[i] = v[i].b;
  coset_fft_inplace(a.data(), log_n, shift);
  coset_fft_inplace(b.data(), log_n, shift);
  for (size_t i = 0; i < n; ++i) v[i] = Ext(a[i], b[i]);
}
// PolynomialCoeffs::eval at an extension point (Horner), base-field coefficients
static inline Ext eval_poly_ext(const u64* c, size_t n, Ext x) {
  Ext acc;
  for (size_t i = n; i-- > 0;) acc = acc * x + Ext(c[i]);
  return acc;
}
static inline Ext eval_extpoly_ext(const Ext* c, size_t n, Ext x) {
  Ext acc;
  for (size_t i = n; i-- > 0;) acc = acc * x + c[i];
  return acc;
}

// ---------------------------------------------------------------------------------------------
// fri/oracle.rs::PolynomialBatch  (zero_knowledge = false: no salt columns, no blinding)
// ---------------------------------------------------------------------------------------------
struct PolynomialBatch {
  int degree_log = 0, rate_bits = 0;
  size_t ncols = 0;
  std::vector<std::vector<u64>> coeffs;  // [col][n]   natural coefficient order
  MerkleTree tree;                       // leaves: 8n rows x ncols, row i = LDE index reverse_bits(i)

  size_t n() const { return (size_t)1 << degree_log; }
  size_t lde_size() const { return (size_t)1 << (degree_log + rate_bits); }

  // from_coeffs: lde_values[col] = coset_fft_7(lde(coeffs[col])) ; transpose ; reverse_index_bits ;
  //              MerkleTree::new(leaves, cap_height)
  void from_coeffs(std::vector<std::vector<u64>>&& polys, int rate_bits_, int cap_height) {
    coeffs = std::move(polys);
    ncols = coeffs.size();
    degree_log = log2_strict(coeffs[0].size());
    rate_bits = rate_bits_;
    int lg = degree_log + rate_bits;
    size_t N = (size_t)1 << lg, nn = n();
    // Same values as  leaves[reverse_bits(j)][c] = coset_fft_7(lde(coeffs[c]))[j]  (checked against direct
    // evaluation in tests/test_oracle_golden.py), computed the cache-friendly way so the oracle is a fair CPU
    // baseline: a decimation-in-frequency pass leaves X[rev(i)] at position i, i.e. already in leaf order, and
    // the column-major result is transposed into row-major leaves in blocks.
    std::vector<u64> leaves(N * ncols);
    std::vector<u64> colmajor(N * ncols);
    // shared tables: shift^i for i < n, and w_N^k for k < N/2
    std::vector<u64> shift_pows(nn), roots(N / 2 ? N / 2 : 1);
    {
      u64 p = 1;
      for (size_t i = 0; i < nn; ++i) shift_pows[i] = p, p = mul(p, MULTIPLICATIVE_GENERATOR);
      u64 w = root_of_unity(lg);
      p = 1;
      for (size_t k = 0; k < N / 2; ++k) roots[k] = p, p = mul(p, w);
    }
    long long nc = (long long)ncols;
#pragma omp parallel for schedule(dynamic)
    for (long long c = 0; c < nc; ++c) {
      u64* v = &colmajor[(size_t)c * N];
      for (size_t i = 0; i < nn; ++i) v[i] = mul(coeffs[c][i], shift_pows[i]);
      for (size_t i = nn; i < N; ++i) v[i] = 0;
      for (size_t half = N / 2, step = 1; half >= 1; half >>= 1, step <<= 1)  // Gentleman-Sande, no permutation
        for (size_t k = 0; k < N; k += 2 * half)
          for (size_t j = 0; j < half; ++j) {
            u64 a = v[k + j], b = v[k + j + half];
            v[k + j] = add(a, b);
            v[k + j + half] = mul(sub(a, b), roots[j * step]);
          }
    }
    {
      const size_t RB = 64;
      long long nblk = (long long)((N + RB - 1) / RB);
#pragma omp parallel for schedule(static)
      for (long long blk = 0; blk < nblk; ++blk) {
        size_t r0 = (size_t)blk * RB, r1 = std::min(N, r0 + RB);
        for (size_t c = 0; c < ncols; ++c) {
          const u64* src = &colmajor[c * N];
          for (size_t r = r0; r < r1; ++r) leaves[r * ncols + c] = src[r];
        }
      }
    }
    std::vector<u64>().swap(colmajor);
    tree.build(std::move(leaves), ncols, cap_height);
  }
  // from_values: ifft per column, then from_coeffs
  void from_values(std::vector<std::vector<u64>>&& values, int rate_bits_, int cap_height) {
    int lg = log2_strict(values[0].size());
    long long nc = (long long)values.size();
#pragma omp parallel for schedule(dynamic)
    for (long long c = 0; c < nc; ++c) ifft_inplace(values[c].data(), lg);
    from_coeffs(std::move(values), rate_bits_, cap_height);
  }
  // get_lde_values(index, step): leaves[reverse_bits(index*step, degree_log+rate_bits)]
  const u64* get_lde_values(size_t index, size_t step) const {
    return tree.leaf(reverse_bits(index * step, degree_log + rate_bits));
  }
};

}  // namespace vxo
