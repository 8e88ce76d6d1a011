// Synthetic circuit + witness generator — the CALLER side of the drop-in boundary.
//
// In the reference the prover is handed a compiled circuit and a finished witness by plonky2x
// (CircuitBuilder::build + witness generation: /root/reference/circuits/header_range.rs:144,167), both
// of which are out of scope for the GPU work.  The real header_range gate mix
// cannot be extracted without Rust, so benchmarks and parity tests use this DECLARED
// stand-in: a standard_recursion_config circuit (135 wires, 80 routed) over the gate set
//   NoopGate, ConstantGate{2}, PublicInputGate, ArithmeticGate{20 ops}, PoseidonGate
// laid out as
//   row 0  PublicInputGate (wires 0..3 = public_inputs_hash)
//   row 1  ConstantGate    (constants 0 and 1)
//   row 2  PoseidonGate    hashing the 4 public inputs in-circuit (outputs 0..3 copy-constrained to row 0)
//   then   P PoseidonGate rows forming a hash chain (output of one row copy-constrained to the next input),
//          A ArithmeticGate rows (20 ops each, op k's first multiplicand copy-constrained to op k-1's output),
//          and NoopGate padding to 2^degree_bits rows.
// Everything is derived deterministically from (degree_bits, seed, poseidon_percent).
// Built into vectorx_amd/libvxsynth.so (plain g++; no GPU code).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <numeric>
#include <vector>
#include "../../include/vxprover.h"
#include "../csrc/host_field.h"

using namespace vxh;

namespace {
struct SplitMix {
  u64 s;
  u64 next() {
    u64 z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
  }
  u64 field() {
    for (;;) {
      u64 v = next();
      if (v < P) return v;
    }
  }
};

struct Synth {
  int degree_bits;
  size_t n;
  std::vector<int32_t> gate_types, gate_params, selector_indices, group_starts, group_ends;
  std::vector<u64> constants_sigmas;  // [4 + 80][n]
  std::vector<u64> k_is;
  std::vector<uint32_t> pi_rows, pi_cols;
  std::vector<u64> witness;  // [135][n]
  std::vector<u64> public_inputs;
  size_t n_poseidon = 0, n_arith = 0, n_noop = 0;
  vx_circuit_desc desc;
};

// Fill one PoseidonGate row (gates/poseidon.rs wire layout) for the given 12 inputs, swap = 0.
void fill_poseidon_row(u64* w /* witness base */, size_t n, size_t row, const u64* in, u64* out) {
  const int START_FULL_0 = 29, START_PARTIAL = 65, START_FULL_1 = 87;
  auto W = [&](int col) -> u64& { return w[(size_t)col * n + row]; };
  static const u64 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  auto mds = [&](u64* s) {
    u64 o[12];
    for (int k = 0; k < 12; ++k) {
      u128 acc = 0;
      for (int i = 0; i < 12; ++i) acc += (u128)C[i] * s[(i + k) % 12];
      if (k == 0) acc += (u128)8 * s[0];
      o[k] = reduce128(acc);
    }
    memcpy(s, o, sizeof o);
  };
  u64 s[12];
  for (int i = 0; i < 12; ++i) W(i) = s[i] = in[i];
  W(24) = 0;                                 // swap
  for (int i = 0; i < 4; ++i) W(25 + i) = 0; // delta_i = swap * (rhs - lhs)
  int round = 0;
  for (int r = 0; r < 4; ++r) {
    for (int i = 0; i < 12; ++i) s[i] = add(s[i], RC[12 * round + i]);
    if (r != 0)
      for (int i = 0; i < 12; ++i) W(START_FULL_0 + 12 * (r - 1) + i) = s[i];
    for (int i = 0; i < 12; ++i) s[i] = sbox(s[i]);
    mds(s);
    ++round;
  }
  for (int r = 0; r < 22; ++r) {
    for (int i = 0; i < 12; ++i) s[i] = add(s[i], RC[12 * round + i]);
    W(START_PARTIAL + r) = s[0];
    s[0] = sbox(s[0]);
    mds(s);
    ++round;
  }
  for (int r = 0; r < 4; ++r) {
    for (int i = 0; i < 12; ++i) s[i] = add(s[i], RC[12 * round + i]);
    for (int i = 0; i < 12; ++i) W(START_FULL_1 + 12 * r + i) = s[i];
    for (int i = 0; i < 12; ++i) s[i] = sbox(s[i]);
    mds(s);
    ++round;
  }
  for (int i = 0; i < 12; ++i) W(12 + i) = out[i] = s[i];
}

struct DSU {
  std::vector<uint32_t> p;
  explicit DSU(size_t n) : p(n) { std::iota(p.begin(), p.end(), 0u); }
  uint32_t find(uint32_t x) {
    while (p[x] != x) {
      p[x] = p[p[x]];
      x = p[x];
    }
    return x;
  }
  void unite(uint32_t a, uint32_t b) {
    a = find(a), b = find(b);
    if (a != b) p[std::max(a, b)] = std::min(a, b);
  }
};
}  // namespace

extern "C" {

typedef struct vxs_circuit vxs_circuit;

vxs_circuit* vxs_build2(int degree_bits, uint64_t seed, int poseidon_percent, uint64_t witness_seed);
vxs_circuit* vxs_build(int degree_bits, uint64_t seed, int poseidon_percent) {
  return vxs_build2(degree_bits, seed, poseidon_percent, seed);
}
// `seed` fixes the CIRCUIT (gate layout, arithmetic constants, copy constraints); `witness_seed` only the witness
// values (public inputs, free wires) — many witnesses of one circuit, as the MapReduce jobs of one map/reduce circuit.
vxs_circuit* vxs_build2(int degree_bits, uint64_t seed, int poseidon_percent, uint64_t witness_seed) {
  if (degree_bits < 3 || degree_bits > 24 || poseidon_percent < 0 || poseidon_percent > 100) return nullptr;
  Synth* S = new Synth();
  S->degree_bits = degree_bits;
  const size_t n = S->n = (size_t)1 << degree_bits;
  const int NW = 135, NR = 80, NSEL = 2, NCONST = 4;
  SplitMix rng{seed};
  SplitMix wrng{witness_seed ^ 0xA5A5A5A55A5A5A5AULL};

  // gates sorted by (degree, id): Noop(0) < Constant(1) < PublicInput(1) < Arithmetic(3) < Poseidon(7);
  // selector_polynomials(max_degree = 9): 7 + 5 - 1 > 9 so two greedy groups [0,4) and [4,5).
  S->gate_types = {VX_GATE_NOOP, VX_GATE_CONSTANT, VX_GATE_PUBLIC_INPUT, VX_GATE_ARITHMETIC, VX_GATE_POSEIDON};
  S->gate_params = {0, 2, 0, 20, 0};
  S->selector_indices = {0, 0, 0, 0, 1};
  S->group_starts = {0, 0, 0, 0, 4};
  S->group_ends = {4, 4, 4, 4, 5};

  const size_t body = n - 3;
  size_t n_noop = std::max<size_t>(1, body / 64);
  if (n_noop > body) n_noop = body;
  size_t n_pos = (body - n_noop) * (size_t)poseidon_percent / 100;
  size_t n_arith = body - n_noop - n_pos;
  S->n_poseidon = n_pos + 1;
  S->n_arith = n_arith;
  S->n_noop = n_noop;

  S->witness.assign((size_t)NW * n, 0);
  S->constants_sigmas.assign((size_t)(NCONST + NR) * n, 0);
  u64* w = S->witness.data();
  u64* sel0 = &S->constants_sigmas[0];
  u64* sel1 = &S->constants_sigmas[n];
  u64* c0 = &S->constants_sigmas[2 * n];
  u64* c1 = &S->constants_sigmas[3 * n];
  const u64 UNUSED = 0xFFFFFFFFULL;
  std::vector<int> row_gate(n);
  auto set_gate = [&](size_t row, int g) {
    row_gate[row] = g;
    sel0[row] = g < 4 ? (u64)g : UNUSED;
    sel1[row] = g == 4 ? 4 : UNUSED;
  };
  DSU dsu((size_t)NR * n);
  auto cell = [&](int col, size_t row) { return (uint32_t)((size_t)col * n + row); };

  // public inputs
  S->public_inputs.resize(4);
  for (auto& v : S->public_inputs) v = wrng.field();
  // row 1: constants 0, 1
  set_gate(1, 1);
  c0[1] = 0;
  c1[1] = 1;
  w[0 * n + 1] = 0;
  w[1 * n + 1] = 1;
  // row 2: in-circuit hash of the public inputs
  set_gate(2, 4);
  u64 in[12] = {0}, out[12];
  for (int i = 0; i < 4; ++i) in[i] = S->public_inputs[i];
  fill_poseidon_row(w, n, 2, in, out);
  for (int i = 0; i < 4; ++i) {
    S->pi_rows.push_back(2);
    S->pi_cols.push_back((uint32_t)i);
  }
  for (int j = 4; j < 12; ++j) dsu.unite(cell(j, 2), cell(0, 1));  // zero padding of the sponge
  dsu.unite(cell(24, 2), cell(0, 1));                              // swap = 0
  // row 0: public input gate carries the hash
  set_gate(0, 2);
  for (int i = 0; i < 4; ++i) {
    w[(size_t)i * n + 0] = out[i];
    dsu.unite(cell(i, 0), cell(12 + i, 2));
  }
  // body: interleave Poseidon and arithmetic rows so both gate kinds are spread over the trace
  size_t row = 3, pos_left = n_pos, ar_left = n_arith;
  size_t prev_pos_row = 0;
  bool have_prev_pos = false;  // the hash chain starts from free inputs, so only rows 0 and 2 depend on the public inputs
  bool have_prev_arith = false;
  size_t prev_arith_row = 0;
  u64 prev_arith_out = 0;
  while (pos_left + ar_left > 0) {
    bool do_pos = pos_left * (n_arith + 1) >= ar_left * (n_pos + 1) ? pos_left > 0 : false;
    if (!do_pos && ar_left == 0) do_pos = true;
    if (do_pos) {
      set_gate(row, 4);
      if (have_prev_pos) memcpy(in, out, sizeof in);
      else for (int i = 0; i < 12; ++i) in[i] = wrng.field();
      fill_poseidon_row(w, n, row, in, out);
      if (have_prev_pos) for (int i = 0; i < 12; ++i) dsu.unite(cell(i, row), cell(12 + i, prev_pos_row));
      have_prev_pos = true;
      prev_pos_row = row;
      --pos_left;
    } else {
      set_gate(row, 3);
      u64 k0 = rng.field(), k1 = rng.field();
      c0[row] = k0;
      c1[row] = k1;
      for (int op = 0; op < 20; ++op) {
        u64 m0 = (op == 0 && !have_prev_arith) ? wrng.field() : prev_arith_out;
        u64 m1 = wrng.field(), ad = wrng.field();
        u64 o = add(mul(mul(m0, m1), k0), mul(ad, k1));
        w[(size_t)(4 * op) * n + row] = m0;
        w[(size_t)(4 * op + 1) * n + row] = m1;
        w[(size_t)(4 * op + 2) * n + row] = ad;
        w[(size_t)(4 * op + 3) * n + row] = o;
        if (op > 0) dsu.unite(cell(4 * op, row), cell(4 * op - 1, row));
        else if (have_prev_arith) dsu.unite(cell(0, row), cell(79, prev_arith_row));
        prev_arith_out = o;
      }
      have_prev_arith = true;
      prev_arith_row = row;
      --ar_left;
    }
    ++row;
  }
  for (; row < n; ++row) set_gate(row, 0);

  // k_is = 7^j (plonk_common / circuit_builder: get_unique_coset_shifts)
  S->k_is.resize(NR);
  {
    u64 acc = 1;
    for (int j = 0; j < NR; ++j) {
      S->k_is[j] = acc;
      acc = mul(acc, 7);
    }
  }
  // sigma: every copy class becomes one cycle over its cells in ascending (column-major) order
  {
    std::vector<u64> subgroup(n);
    u64 wg = root_of_unity(degree_bits);
    subgroup[0] = 1;
    for (size_t i = 1; i < n; ++i) subgroup[i] = mul(subgroup[i - 1], wg);
    const size_t ncells = (size_t)NR * n;
    std::vector<uint32_t> next(ncells);
    std::vector<uint32_t> last_of_root(ncells, 0xFFFFFFFFu), first_of_root(ncells, 0xFFFFFFFFu);
    for (size_t cidx = 0; cidx < ncells; ++cidx) {
      uint32_t r = dsu.find((uint32_t)cidx);
      if (first_of_root[r] == 0xFFFFFFFFu) first_of_root[r] = (uint32_t)cidx;
      else next[last_of_root[r]] = (uint32_t)cidx;
      last_of_root[r] = (uint32_t)cidx;
    }
    for (size_t cidx = 0; cidx < ncells; ++cidx) {
      uint32_t r = dsu.find((uint32_t)cidx);
      if (last_of_root[r] == cidx) next[cidx] = first_of_root[r];
    }
    u64* sig = &S->constants_sigmas[(size_t)NCONST * n];
    for (size_t cidx = 0; cidx < ncells; ++cidx) {
      size_t jc = next[cidx] / n, ir = next[cidx] % n;
      sig[cidx] = mul(S->k_is[jc], subgroup[ir]);
    }
  }
  vx_circuit_desc& d = S->desc;
  memset(&d, 0, sizeof d);
  d.degree_bits = degree_bits;
  d.num_wires = NW;
  d.num_routed_wires = NR;
  d.num_challenges = 2;
  d.rate_bits = 3;
  d.cap_height = std::min(4, degree_bits + 3);
  d.pow_bits = 16;
  d.num_query_rounds = 28;
  d.quotient_degree_factor = 8;
  d.num_gates = 5;
  d.gate_types = S->gate_types.data();
  d.gate_params = S->gate_params.data();
  d.selector_indices = S->selector_indices.data();
  d.group_starts = S->group_starts.data();
  d.group_ends = S->group_ends.data();
  d.num_selectors = NSEL;
  d.num_constants = NCONST;
  d.constants_sigmas = S->constants_sigmas.data();
  d.k_is = S->k_is.data();
  d.num_public_inputs = 4;
  d.pi_rows = S->pi_rows.data();
  d.pi_cols = S->pi_cols.data();
  return reinterpret_cast<vxs_circuit*>(S);
}

void vxs_free(vxs_circuit* c) { delete reinterpret_cast<Synth*>(c); }
/* New public inputs for the same circuit: only witness rows 0 (PublicInputGate) and 2 (in-circuit hash of the
 * public inputs) change.  Writes the two rows (num_wires values each) for the caller to patch into its copy of the
 * witness matrix, and updates the generator's own copy if it is still held. */
void vxs_patch_public_inputs(vxs_circuit* c, const uint64_t pi[4], uint64_t* row0_out, uint64_t* row2_out) {
  Synth* S = reinterpret_cast<Synth*>(c);
  const int NW = 135;
  std::vector<u64> tmp((size_t)NW * 4, 0);  // a 4-row scratch witness, rows 0 and 2 used
  u64 in[12] = {0}, out[12];
  for (int i = 0; i < 4; ++i) in[i] = canon(pi[i]);
  fill_poseidon_row(tmp.data(), 4, 2, in, out);
  for (int i = 0; i < 4; ++i) tmp[(size_t)i * 4 + 0] = out[i];
  for (int col = 0; col < NW; ++col) {
    row0_out[col] = tmp[(size_t)col * 4 + 0];
    row2_out[col] = tmp[(size_t)col * 4 + 2];
  }
  for (int i = 0; i < 4; ++i) S->public_inputs[i] = in[i];
  if (!S->witness.empty())
    for (int col = 0; col < NW; ++col) {
      S->witness[(size_t)col * S->n + 0] = row0_out[col];
      S->witness[(size_t)col * S->n + 2] = row2_out[col];
    }
}
const vx_circuit_desc* vxs_desc(vxs_circuit* c) { return &reinterpret_cast<Synth*>(c)->desc; }
const uint64_t* vxs_witness(vxs_circuit* c) { return reinterpret_cast<Synth*>(c)->witness.data(); }
const uint64_t* vxs_public_inputs(vxs_circuit* c) { return reinterpret_cast<Synth*>(c)->public_inputs.data(); }
void vxs_row_counts(vxs_circuit* c, uint64_t out[3]) {
  Synth* S = reinterpret_cast<Synth*>(c);
  out[0] = S->n_poseidon;
  out[1] = S->n_arith;
  out[2] = S->n_noop;
}
/* Drop the (large) sigma/witness host copies once they have been handed to a prover. */
void vxs_release_host_buffers(vxs_circuit* c, int witness, int preprocessed) {
  Synth* S = reinterpret_cast<Synth*>(c);
  if (witness) std::vector<u64>().swap(S->witness);
  if (preprocessed) {
    std::vector<u64>().swap(S->constants_sigmas);
    S->desc.constants_sigmas = nullptr;
  }
}

}  // extern "C"
