// ORACLE — TEST INFRASTRUCTURE ONLY (see field.hpp).  extern "C" surface of the CPU restatement so
// tests/, smoke() and bench.py's cpu_baseline leg can drive it through ctypes.  Built into
// oracle/liboracle.so by oracle/Makefile.  The product library (libvxprover.so) never links this.
#include "poly.hpp"
#include <omp.h>
#include <cstring>

using namespace vxo;

extern "C" {

int vxo_num_threads() { return omp_get_max_threads(); }
void vxo_set_num_threads(int t) { omp_set_num_threads(t); }

u64 vxo_mul(u64 a, u64 b) { return mul(canon(a), canon(b)); }
u64 vxo_add(u64 a, u64 b) { return add(canon(a), canon(b)); }
u64 vxo_sub(u64 a, u64 b) { return sub(canon(a), canon(b)); }
u64 vxo_inv(u64 a) { return inv(canon(a)); }
u64 vxo_pow(u64 a, u64 e) { return pow(canon(a), e); }
u64 vxo_root_of_unity(int log_n) { return root_of_unity(log_n); }
void vxo_ext_mul(const u64* x, const u64* y, u64* out) {
  Ext r = Ext(canon(x[0]), canon(x[1])) * Ext(canon(y[0]), canon(y[1]));
  out[0] = r.a, out[1] = r.b;
}
void vxo_ext_inv(const u64* x, u64* out) {
  Ext r = ext_inv(Ext(canon(x[0]), canon(x[1])));
  out[0] = r.a, out[1] = r.b;
}

void vxo_poseidon_permute(u64* state, size_t count) {
  long long c = (long long)count;
#pragma omp parallel for schedule(static)
  for (long long i = 0; i < c; ++i) {
    State s;
    for (int k = 0; k < 12; ++k) s[k] = canon(state[i * 12 + k]);
    permute(s);
    for (int k = 0; k < 12; ++k) state[i * 12 + k] = s[k];
  }
}
void vxo_hash_no_pad(const u64* in, size_t n, u64* out4) {
  std::vector<u64> t(in, in + n);
  for (auto& x : t) x = canon(x);
  Hash h = hash_no_pad(t.data(), n);
  memcpy(out4, h.e, 32);
}
void vxo_hash_or_noop(const u64* in, size_t n, u64* out4) {
  std::vector<u64> t(in, in + n);
  for (auto& x : t) x = canon(x);
  Hash h = hash_or_noop(t.data(), n);
  memcpy(out4, h.e, 32);
}
void vxo_two_to_one(const u64* l, const u64* r, u64* out4) {
  Hash a, b;
  for (int i = 0; i < 4; ++i) a.e[i] = canon(l[i]), b.e[i] = canon(r[i]);
  Hash h = two_to_one(a, b);
  memcpy(out4, h.e, 32);
}

// Column-major batch transforms, data[col*n + i], in place.  kind: 0 = fft, 1 = ifft,
// 2 = coset_fft(shift), 3 = coset_ifft(shift).  Natural order in and out (fft.rs conventions).
void vxo_ntt_batch(u64* data, int log_n, size_t ncols, int kind, u64 shift) {
  size_t n = (size_t)1 << log_n;
  long long nc = (long long)ncols;
#pragma omp parallel for schedule(dynamic)
  for (long long c = 0; c < nc; ++c) {
    u64* a = data + (size_t)c * n;
    for (size_t i = 0; i < n; ++i) a[i] = canon(a[i]);
    switch (kind) {
      case 0: fft_inplace(a, log_n); break;
      case 1: ifft_inplace(a, log_n); break;
      case 2: coset_fft_inplace(a, log_n, shift); break;
      default: coset_ifft_inplace(a, log_n, shift); break;
    }
  }
}

// Leaf digests + Merkle cap of a ROW-MAJOR leaf matrix (merkle_tree.rs::MerkleTree::new).
// digests_out (optional) = n_leaves*4, cap_out = 2^cap_height * 4.
void vxo_merkle(const u64* leaves, size_t n_leaves, size_t width, int cap_height, u64* digests_out,
                u64* cap_out) {
  std::vector<u64> lv(leaves, leaves + n_leaves * width);
  for (auto& x : lv) x = canon(x);
  MerkleTree t;
  t.build(std::move(lv), width, cap_height);
  if (digests_out) memcpy(digests_out, t.layers[0].data(), n_leaves * 32);
  memcpy(cap_out, t.cap().data(), t.cap().size() * 32);
}

// PolynomialBatch::from_values / from_coeffs (fri/oracle.rs).  `cols` is column-major [ncols][n]
// (values on H in natural order, or coefficients when is_coeffs != 0).  Outputs (each optional):
//   coeffs_out [ncols][n] natural coefficient order,
//   leaves_out [8n][ncols] row-major, row i = LDE point index reverse_bits(i),
//   digests_out [8n][4], cap_out [2^cap_height][4].
void vxo_commit(const u64* cols, int log_n, size_t ncols, int rate_bits, int cap_height, int is_coeffs,
                u64* coeffs_out, u64* leaves_out, u64* digests_out, u64* cap_out) {
  size_t n = (size_t)1 << log_n;
  std::vector<std::vector<u64>> v(ncols);
  for (size_t c = 0; c < ncols; ++c) {
    v[c].assign(cols + c * n, cols + (c + 1) * n);
    for (auto& x : v[c]) x = canon(x);
  }
  PolynomialBatch b;
  if (is_coeffs)
    b.from_coeffs(std::move(v), rate_bits, cap_height);
  else
    b.from_values(std::move(v), rate_bits, cap_height);
  if (coeffs_out)
    for (size_t c = 0; c < ncols; ++c) memcpy(coeffs_out + c * n, b.coeffs[c].data(), n * 8);
  if (leaves_out) memcpy(leaves_out, b.tree.leaves.data(), b.tree.leaves.size() * 8);
  if (digests_out) memcpy(digests_out, b.tree.layers[0].data(), b.tree.n_leaves * 32);
  if (cap_out) memcpy(cap_out, b.tree.cap().data(), b.tree.cap().size() * 32);
}

}  // extern "C"
