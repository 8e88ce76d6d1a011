// ORACLE — TEST INFRASTRUCTURE ONLY (see field.hpp header for the full notice and parity status).
// Restates  plonky2/src/hash/poseidon.rs, poseidon_goldilocks.rs, hashing.rs, hash_types.rs,
//           merkle_tree.rs, merkle_proofs.rs  of plonky2 v0.2.0          
// PINNED by the three permutation known-answer vectors of SURVEY.md Appendix B.2
// (tests/golden/poseidon_kat.json) and the round-constant checksum of B.1.
#pragma once
#include "field.hpp"
#include "poseidon_constants.h"
#include <array>
#include <cstring>

namespace vxo {

static const int SPONGE_WIDTH = 12, SPONGE_RATE = 8, HALF_N_FULL_ROUNDS = 4, N_PARTIAL_ROUNDS = 22;
static const int N_ROUNDS = 2 * HALF_N_FULL_ROUNDS + N_PARTIAL_ROUNDS;
static const u64 ROUND_CONSTANTS[VX_POSEIDON_N_ROUND_CONSTANTS] = VX_POSEIDON_ROUND_CONSTANTS_INIT;
static const u64 MDS_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static const u64 MDS_DIAG[12] = {8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

typedef std::array<u64, 12> State;

static inline u64 sbox(u64 x) {
  u64 x2 = sqr(x), x4 = sqr(x2), x3 = mul(x, x2);
  return mul(x3, x4);
}
// poseidon.rs::mds_layer (naive form)
static inline void mds_layer(State& s) {
  // All MDS entries are < 2^6: accumulate the low and high 32-bit halves of the state separately in
  // 64-bit lanes (sums < 2^42, auto-vectorisable) and fold once — same value as the u128 form upstream
  // uses (poseidon.rs::mds_row_shf), without 128-bit multiplies.
  uint64_t lo[24], hi[24];
  for (int i = 0; i < 12; ++i) {
    lo[i] = lo[i + 12] = s[i] & 0xFFFFFFFFULL;
    hi[i] = hi[i + 12] = s[i] >> 32;
  }
  State o;
  for (int r = 0; r < 12; ++r) {
    u64 al = 0, ah = 0;
    for (int i = 0; i < 12; ++i) {
      al += MDS_CIRC[i] * lo[i + r];
      ah += MDS_CIRC[i] * hi[i + r];
    }
    al += MDS_DIAG[r] * lo[r];
    ah += MDS_DIAG[r] * hi[r];
    o[r] = reduce128((u128)al + ((u128)ah << 32));
  }
  s = o;
}
// poseidon.rs::poseidon_naive — the reference form, pinned by the known-answer vectors.
static inline void permute_naive(State& s) {
  int rc = 0;
  for (int r = 0; r < N_ROUNDS; ++r) {
    for (int i = 0; i < 12; ++i) s[i] = add(s[i], ROUND_CONSTANTS[rc++]);
    bool full = r < HALF_N_FULL_ROUNDS || r >= HALF_N_FULL_ROUNDS + N_PARTIAL_ROUNDS;
    if (full)
      for (int i = 0; i < 12; ++i) s[i] = sbox(s[i]);
    else
      s[0] = sbox(s[0]);
    mds_layer(s);
  }
}

// poseidon.rs::poseidon — partial rounds in the sparse "fast" form upstream uses (its FAST_PARTIAL_* tables are
// unavailable; the equivalent constants are re-derived and proven equal to the naive form on the KATs and random
// states by tools/gen_poseidon_fast_constants.py, and tests/test_oracle_golden.py re-checks permute == permute_naive).
// Used for all hashing so that the oracle is a fair CPU baseline; gate constraints keep the naive formulas.
#include "poseidon_fast_constants.h"
static const u64 FAST_FIRST[12] = VX_FAST_PARTIAL_FIRST_ROUND_CONSTANT_INIT;
static const u64 FAST_K[22] = VX_FAST_PARTIAL_ROUND_CONSTANTS_INIT;
static const u64 FAST_INIT[11][11] = VX_FAST_PARTIAL_INITIAL_MATRIX_INIT;
static const u64 FAST_W_HATS[22][11] = VX_FAST_PARTIAL_W_HATS_INIT;
static const u64 FAST_VS[22][11] = VX_FAST_PARTIAL_VS_INIT;
// sum of up to 12 products < 12 * 2^128: accumulate in (u128 low, small high) and fold 2^128 = -2^32 (mod p)
static inline u64 dot_reduce(const u64* a, const u64* b, int n, u64 extra_a = 0, u64 extra_b = 0) {
  u128 acc = (u128)extra_a * extra_b;
  u64 over = 0;
  for (int i = 0; i < n; ++i) {
    u128 t = (u128)a[i] * b[i];
    acc += t;
    over += acc < t;
  }
  u64 r = reduce128(acc);
  return sub(r, mul(over, (u64)1 << 32));
}
static inline void permute(State& s) {
  int rc = 0;
  for (int r = 0; r < HALF_N_FULL_ROUNDS; ++r) {
    for (int i = 0; i < 12; ++i) s[i] = sbox(add(s[i], ROUND_CONSTANTS[rc++]));
    mds_layer(s);
  }
  for (int i = 0; i < 12; ++i) s[i] = add(s[i], FAST_FIRST[i]);
  {
    u64 t[11];
    for (int r = 0; r < 11; ++r) t[r] = dot_reduce(&s[1], FAST_INIT[r], 11);
    for (int r = 0; r < 11; ++r) s[1 + r] = t[r];
  }
  for (int r = 0; r < N_PARTIAL_ROUNDS; ++r) {
    u64 s0 = add(sbox(s[0]), FAST_K[r]);
    u64 d = dot_reduce(&s[1], FAST_W_HATS[r], 11, s0, MDS_CIRC[0] + MDS_DIAG[0]);
    for (int i = 0; i < 11; ++i) s[1 + i] = reduce128((u128)s0 * FAST_VS[r][i] + s[1 + i]);
    s[0] = d;
  }
  rc = 12 * (HALF_N_FULL_ROUNDS + N_PARTIAL_ROUNDS);
  for (int r = 0; r < HALF_N_FULL_ROUNDS; ++r) {
    for (int i = 0; i < 12; ++i) s[i] = sbox(add(s[i], ROUND_CONSTANTS[rc++]));
    mds_layer(s);
  }
}

// B independent permutations side by side (round 6): the partial rounds are one dependency chain per state — S-box of lane 0, dot
// product, next round — so a lone permutation leaves most of the core idle; interleaving the chains of B states lets the out-of-order
// core overlap them (1.4x at B = 4 on the build machine).  Same arithmetic per state as `permute`, step by step.
template <int B>
static inline void permute_batch(State* s) {
  int rc = 0;
  for (int r = 0; r < HALF_N_FULL_ROUNDS; ++r) {
    for (int b = 0; b < B; ++b)
      for (int i = 0; i < 12; ++i) s[b][i] = sbox(add(s[b][i], ROUND_CONSTANTS[rc + i]));
    rc += 12;
    for (int b = 0; b < B; ++b) mds_layer(s[b]);
  }
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < 12; ++i) s[b][i] = add(s[b][i], FAST_FIRST[i]);
  for (int b = 0; b < B; ++b) {
    u64 t[11];
    for (int r = 0; r < 11; ++r) t[r] = dot_reduce(&s[b][1], FAST_INIT[r], 11);
    for (int r = 0; r < 11; ++r) s[b][1 + r] = t[r];
  }
  for (int r = 0; r < N_PARTIAL_ROUNDS; ++r) {
    u64 s0[B];
    for (int b = 0; b < B; ++b) s0[b] = add(sbox(s[b][0]), FAST_K[r]);
    for (int b = 0; b < B; ++b) {
      u64 d = dot_reduce(&s[b][1], FAST_W_HATS[r], 11, s0[b], MDS_CIRC[0] + MDS_DIAG[0]);
      for (int i = 0; i < 11; ++i) s[b][1 + i] = reduce128((u128)s0[b] * FAST_VS[r][i] + s[b][1 + i]);
      s[b][0] = d;
    }
  }
  rc = 12 * (HALF_N_FULL_ROUNDS + N_PARTIAL_ROUNDS);
  for (int r = 0; r < HALF_N_FULL_ROUNDS; ++r) {
    for (int b = 0; b < B; ++b)
      for (int i = 0; i < 12; ++i) s[b][i] = sbox(add(s[b][i], ROUND_CONSTANTS[rc + i]));
    rc += 12;
    for (int b = 0; b < B; ++b) mds_layer(s[b]);
  }
}

struct Hash {
  u64 e[4];
  bool operator==(const Hash& o) const { return !memcmp(e, o.e, sizeof e); }
};

// hashing.rs::hash_n_to_m_no_pad with m = 4 (overwrite-mode sponge)
static inline Hash hash_no_pad(const u64* in, size_t n) {
  State s{};
  for (size_t off = 0; off < n; off += SPONGE_RATE) {
    size_t len = n - off < (size_t)SPONGE_RATE ? n - off : SPONGE_RATE;
    for (size_t i = 0; i < len; ++i) s[i] = in[off + i];
    permute(s);
  }
  if (n == 0) {
    // upstream: empty input performs zero permutations and squeezes the zero state
  }
  Hash h;
  for (int i = 0; i < 4; ++i) h.e[i] = s[i];
  return h;
}
// hashing.rs::hash_n_to_m_no_pad general squeeze
static inline std::vector<u64> hash_n_to_m_no_pad(const u64* in, size_t n, size_t m) {
  State s{};
  for (size_t off = 0; off < n; off += SPONGE_RATE) {
    size_t len = n - off < (size_t)SPONGE_RATE ? n - off : SPONGE_RATE;
    for (size_t i = 0; i < len; ++i) s[i] = in[off + i];
    permute(s);
  }
  std::vector<u64> out;
  for (;;) {
    for (int i = 0; i < SPONGE_RATE; ++i) {
      out.push_back(s[i]);
      if (out.size() == m) return out;
    }
    permute(s);
  }
}
// hash_types / Hasher::hash_or_noop
static inline Hash hash_or_noop(const u64* in, size_t n) {
  if (n <= 4) {
    Hash h{{0, 0, 0, 0}};
    for (size_t i = 0; i < n; ++i) h.e[i] = in[i];
    return h;
  }
  return hash_no_pad(in, n);
}
// PoseidonHash::two_to_one
static inline Hash two_to_one(const Hash& l, const Hash& r) {
  State s{};
  for (int i = 0; i < 4; ++i) s[i] = l.e[i], s[4 + i] = r.e[i];
  permute(s);
  Hash h;
  for (int i = 0; i < 4; ++i) h.e[i] = s[i];
  return h;
}

// hash_or_noop of PB leaves of the same width, and two_to_one of PB node pairs, PB permutations at a time (same outputs as the scalar forms)
static const int PB = 4;
static inline void hash_or_noop_batch(const u64* in, size_t w, Hash* out) {   // leaves in + k w, k < PB
  if (w <= 4) {
    for (int k = 0; k < PB; ++k) out[k] = hash_or_noop(in + (size_t)k * w, w);
    return;
  }
  State s[PB];
  for (int k = 0; k < PB; ++k) s[k] = State{};
  for (size_t off = 0; off < w; off += SPONGE_RATE) {
    size_t len = w - off < (size_t)SPONGE_RATE ? w - off : SPONGE_RATE;
    for (int k = 0; k < PB; ++k)
      for (size_t i = 0; i < len; ++i) s[k][i] = in[(size_t)k * w + off + i];
    permute_batch<PB>(s);
  }
  for (int k = 0; k < PB; ++k)
    for (int i = 0; i < 4; ++i) out[k].e[i] = s[k][i];
}
static inline void two_to_one_batch(const Hash* children, Hash* out) {       // out[k] = two_to_one(children[2k], children[2k + 1]), k < PB
  State s[PB];
  for (int k = 0; k < PB; ++k) {
    s[k] = State{};
    for (int i = 0; i < 4; ++i) s[k][i] = children[2 * k].e[i], s[k][4 + i] = children[2 * k + 1].e[i];
  }
  permute_batch<PB>(s);
  for (int k = 0; k < PB; ++k)
    for (int i = 0; i < 4; ++i) out[k].e[i] = s[k][i];
}

// ---------------------------------------------------------------------------------------------
// merkle_tree.rs::MerkleTree (outputs only: cap + proofs; upstream's interleaved digest buffer is
// an internal layout that does not affect either).  Leaves are row-major: leaf i = leaves[i*width..].
// ---------------------------------------------------------------------------------------------
struct MerkleTree {
  size_t n_leaves = 0, width = 0;
  int cap_height = 0;
  std::vector<u64> leaves;                 // n_leaves * width
  std::vector<std::vector<Hash>> layers;   // layers[0] = leaf digests, last = cap
  const std::vector<Hash>& cap() const { return layers.back(); }
  const u64* leaf(size_t i) const { return &leaves[i * width]; }

  void build(std::vector<u64>&& lv, size_t w, int cap_h) {
    leaves = std::move(lv);
    width = w;
    n_leaves = leaves.size() / w;
    int lg = log2_strict(n_leaves);
    assert(cap_h <= lg);
    cap_height = cap_h;
    layers.clear();
    layers.emplace_back(n_leaves);
    {
      std::vector<Hash>& d = layers[0];
      long long nl = (long long)n_leaves;
      if (nl % PB == 0) {
#pragma omp parallel for schedule(static)
        for (long long i = 0; i < nl; i += PB) hash_or_noop_batch(&leaves[(size_t)i * w], w, &d[i]);
      } else {
#pragma omp parallel for schedule(static)
        for (long long i = 0; i < nl; ++i) d[i] = hash_or_noop(&leaves[(size_t)i * w], w);
      }
    }
    for (int lvl = lg; lvl > cap_h; --lvl) {
      const std::vector<Hash>& prev = layers.back();
      std::vector<Hash> next(prev.size() / 2);
      long long nn = (long long)next.size();
      if (nn % PB == 0) {
#pragma omp parallel for schedule(static)
        for (long long i = 0; i < nn; i += PB) two_to_one_batch(&prev[2 * i], &next[i]);
      } else {
#pragma omp parallel for schedule(static)
        for (long long i = 0; i < nn; ++i) next[i] = two_to_one(prev[2 * i], prev[2 * i + 1]);
      }
      layers.push_back(std::move(next));
    }
  }
  // MerkleTree::prove — siblings bottom-up, length log2(n) - cap_height
  std::vector<Hash> prove(size_t idx) const {
    std::vector<Hash> sib;
    for (size_t l = 0; l + 1 < layers.size(); ++l) {
      sib.push_back(layers[l][idx ^ 1]);
      idx >>= 1;
    }
    return sib;
  }
};

// merkle_proofs.rs::verify_merkle_proof_to_cap
static inline bool verify_merkle_proof_to_cap(const u64* leaf, size_t width, size_t idx,
                                              const std::vector<Hash>& cap,
                                              const std::vector<Hash>& siblings) {
  Hash cur = hash_or_noop(leaf, width);
  for (const Hash& s : siblings) {
    cur = (idx & 1) ? two_to_one(s, cur) : two_to_one(cur, s);
    idx >>= 1;
  }
  return idx < cap.size() && cur == cap[idx];
}

}  // namespace vxo
