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
void vxo_poseidon_permute_naive(u64* state, size_t count) {
  for (size_t i = 0; i < count; ++i) {
    State s;
    for (int k = 0; k < 12; ++k) s[k] = canon(state[i * 12 + k]);
    permute_naive(s);
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

// ---------------------------------------------------------------------------------------------
// Whole-proof surface: circuit load, prove, verify (plonk.hpp)
// ---------------------------------------------------------------------------------------------
#include "plonk.hpp"

extern "C" {

// Same layout as vx_circuit_desc in include/vxprover.h (declared again here: the oracle shares no
// code with the product).
struct vxo_circuit_desc {
  int32_t degree_bits, num_wires, num_routed_wires, num_challenges, rate_bits, cap_height, pow_bits, num_query_rounds,
      quotient_degree_factor, num_gates;
  const int32_t *gate_types, *gate_params, *selector_indices, *group_starts, *group_ends;
  int32_t num_selectors, num_constants;
  const uint64_t* constants_sigmas;
  const uint64_t* k_is;
  int32_t num_public_inputs;
  const uint32_t *pi_rows, *pi_cols;
  int32_t programs_len;
  const uint64_t* programs;
  const int32_t* program_offsets;
  uint32_t override_flags;  // 1: circuit_digest, 2: fri_reduction_arity_bits, 4: num_partial_products
  int32_t hiding;
  uint64_t circuit_digest[4];
  int32_t num_fri_reduction_arity_bits;
  int32_t num_partial_products;
  const int32_t* fri_reduction_arity_bits;
  int32_t num_luts, num_lookup_selectors;
  const int32_t* lut_lens;
  const uint16_t *lut_inputs, *lut_outputs;
  const int32_t* lookup_rows;
};
static void load_overrides(const vxo_circuit_desc* d, Circuit* c) {
  if (d->hiding) throw std::runtime_error("zero-knowledge circuits are not restated");
  if (d->num_luts > 0) {  // lookup argument
    c->num_lookup_selectors = d->num_lookup_selectors;
    size_t off = 0;
    for (int t = 0; t < d->num_luts; ++t) {
      std::vector<std::pair<u64, u64>> lut;
      for (int k = 0; k < d->lut_lens[t]; ++k, ++off) lut.push_back({d->lut_inputs[off], d->lut_outputs[off]});
      c->luts.push_back(std::move(lut));
      c->lookup_rows.push_back({(size_t)d->lookup_rows[3 * t], (size_t)d->lookup_rows[3 * t + 1], (size_t)d->lookup_rows[3 * t + 2]});
    }
  }
  if (d->override_flags & 1) {
    c->has_digest_override = true;
    for (int i = 0; i < 4; ++i) c->digest_override.e[i] = canon(d->circuit_digest[i]);
  }
  if (d->override_flags & 2) {
    c->has_arity_override = true;
    c->arity_override.assign(d->fri_reduction_arity_bits, d->fri_reduction_arity_bits + d->num_fri_reduction_arity_bits);
  }
  if ((d->override_flags & 4) && d->num_partial_products != c->num_partial_products())
    throw std::runtime_error("num_partial_products disagrees with ceil(num_routed_wires / quotient_degree_factor) - 1");
}
static void load_gate_program(const vxo_circuit_desc* d, int i, Gate& g) {
  if (g.type != GATE_PROGRAM) return;
  if (!d->programs || !d->program_offsets || d->program_offsets[i] < 0) throw std::runtime_error("program gate without a program");
  for (int pc = d->program_offsets[i]; pc < d->programs_len; ++pc) {
    u64 ins = d->programs[pc];
    g.program.push_back(ins);
    int op = (int)(ins & 0xFF);
    if (op == OP_END) break;
    if (op == OP_LDI) g.program.push_back(d->programs[++pc]);
  }
}

void* vxo_circuit_create(const vxo_circuit_desc* d) {
  Circuit* c = new Circuit();
  c->degree_bits = d->degree_bits;
  c->num_wires = d->num_wires;
  c->num_routed_wires = d->num_routed_wires;
  c->num_challenges = d->num_challenges;
  c->rate_bits = d->rate_bits;
  c->cap_height = d->cap_height;
  c->pow_bits = d->pow_bits;
  c->num_query_rounds = d->num_query_rounds;
  c->quotient_degree_factor = d->quotient_degree_factor;
  for (int i = 0; i < d->num_gates; ++i) {
    Gate g;
    g.type = d->gate_types[i];
    g.param = d->gate_params[i];
    g.selector_index = d->selector_indices[i];
    g.group_start = d->group_starts[i];
    g.group_end = d->group_ends[i];
    load_gate_program(d, i, g);
    c->gates.push_back(g);
  }
  c->num_selectors = d->num_selectors;
  c->num_constants = d->num_constants;
  c->k_is.assign(d->k_is, d->k_is + d->num_routed_wires);
  for (int i = 0; i < d->num_public_inputs; ++i) c->public_inputs.push_back({d->pi_rows[i], d->pi_cols[i]});
  size_t n = (size_t)1 << d->degree_bits, m = (size_t)d->num_constants + d->num_routed_wires;
  std::vector<std::vector<u64>> cols(m);
  for (size_t k = 0; k < m; ++k) {
    cols[k].assign(d->constants_sigmas + k * n, d->constants_sigmas + (k + 1) * n);
    for (auto& x : cols[k]) x = canon(x);
  }
  load_overrides(d, c);
  c->finalize(std::move(cols));
  return c;
}
// VerifierOnlyCircuitData: the verifier needs the circuit parameters, the gate list, k_is and the
// constants_sigmas CAP (+ the digest derived from it) — not the preprocessed polynomials.  Used by the tests
// to check GPU proofs at sizes where committing 84 columns x 2^24 rows on the CPU would take minutes.
void* vxo_circuit_create_verifier(const vxo_circuit_desc* d, const u64* cap) {
  Circuit* c = new Circuit();
  c->degree_bits = d->degree_bits;
  c->num_wires = d->num_wires;
  c->num_routed_wires = d->num_routed_wires;
  c->num_challenges = d->num_challenges;
  c->rate_bits = d->rate_bits;
  c->cap_height = d->cap_height;
  c->pow_bits = d->pow_bits;
  c->num_query_rounds = d->num_query_rounds;
  c->quotient_degree_factor = d->quotient_degree_factor;
  for (int i = 0; i < d->num_gates; ++i) {
    Gate g;
    g.type = d->gate_types[i];
    g.param = d->gate_params[i];
    g.selector_index = d->selector_indices[i];
    g.group_start = d->group_starts[i];
    g.group_end = d->group_ends[i];
    load_gate_program(d, i, g);
    c->gates.push_back(g);
  }
  c->num_selectors = d->num_selectors;
  c->num_constants = d->num_constants;
  c->k_is.assign(d->k_is, d->k_is + d->num_routed_wires);
  for (int i = 0; i < d->num_public_inputs; ++i) c->public_inputs.push_back({d->pi_rows[i], d->pi_cols[i]});
  c->num_gate_constraints = 0;
  for (const Gate& g : c->gates) c->num_gate_constraints = std::max(c->num_gate_constraints, g.num_constraints());
  load_overrides(d, c);
  c->compute_fri_params();
  std::vector<Hash> capv((size_t)1 << d->cap_height);
  for (size_t i = 0; i < capv.size(); ++i)
    for (int k = 0; k < 4; ++k) capv[i].e[k] = canon(cap[4 * i + k]);
  c->constants_sigmas.tree.layers.clear();
  c->constants_sigmas.tree.layers.push_back(capv);
  c->constants_sigmas.tree.cap_height = d->cap_height;
  c->circuit_digest = c->has_digest_override ? c->digest_override : Circuit::digest_of_cap(capv, d->degree_bits);
  return c;
}
void vxo_circuit_free(void* c) { delete (Circuit*)c; }
void vxo_circuit_digest(void* c, u64* out4) { memcpy(out4, ((Circuit*)c)->circuit_digest.e, 32); }
void vxo_circuit_cap(void* c, u64* out) {
  const auto& cap = ((Circuit*)c)->constants_sigmas.tree.cap();
  memcpy(out, cap.data(), cap.size() * 32);
}

// Returns proof byte length (> 0), or -1 with the message in err.  timings_out (optional, 8 doubles):
// wires_commit, zs_pp, zs_pp_commit, quotient_eval, quotient_commit, openings, fri, total  (seconds).
long long vxo_prove(void* cv, const u64* wires, const u64* pow_hint, uint8_t* out, size_t cap, double* timings_out,
                    char* err, size_t err_cap) {
  Circuit* c = (Circuit*)cv;
  try {
    size_t n = c->n();
    std::vector<std::vector<u64>> w(c->num_wires);
    for (int k = 0; k < c->num_wires; ++k) {
      w[k].assign(wires + (size_t)k * n, wires + (size_t)(k + 1) * n);
      for (auto& x : w[k]) x = canon(x);
    }
    ProveOptions opt;
    if (pow_hint) opt.has_pow_hint = true, opt.pow_hint = *pow_hint;
    ProverTimings tm;
    Proof p = prove(*c, w, opt, &tm);
    std::vector<uint8_t> bytes = serialize_proof(p);
    if (timings_out) {
      double t[8] = {tm.wires_commit, tm.zs_pp, tm.zs_pp_commit, tm.quotient_eval, tm.quotient_commit, tm.openings, tm.fri, tm.total};
      memcpy(timings_out, t, sizeof t);
    }
    if (bytes.size() > cap) {
      snprintf(err, err_cap, "output buffer too small: need %zu bytes", bytes.size());
      return -1;
    }
    memcpy(out, bytes.data(), bytes.size());
    return (long long)bytes.size();
  } catch (const std::exception& e) {
    snprintf(err, err_cap, "%s", e.what());
    return -1;
  }
}

// 0 = proof accepted; -1 = rejected / malformed with the reason in err.
int vxo_verify(void* cv, const uint8_t* proof, size_t len, char* err, size_t err_cap) {
  Circuit* c = (Circuit*)cv;
  Proof p;
  if (!deserialize_proof(*c, proof, len, p)) {
    snprintf(err, err_cap, "malformed proof bytes");
    return -1;
  }
  std::string r = verify(*c, p);
  if (!r.empty()) {
    snprintf(err, err_cap, "%s", r.c_str());
    return -1;
  }
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// STARK spike (stark.hpp): the checker of vx_stark_prove
// ---------------------------------------------------------------------------------------------
#include "stark.hpp"

extern "C" {
// Same layout as vx_stark_desc in include/vxprover.h (declared again: the oracle shares no code with the product).
struct vxo_stark_desc {
  int32_t degree_bits, num_columns, num_public_inputs;
  int32_t rate_bits, cap_height, pow_bits, num_query_rounds, num_challenges;
  int32_t constraint_degree;
  int32_t program_len;
  const uint64_t* program;
  uint32_t override_flags;
  int32_t num_fri_reduction_arity_bits;
  const int32_t* fri_reduction_arity_bits;
  int32_t num_aux_columns, num_aux_challenges;
  int32_t num_aux_public_inputs;
};
// aux_fn(challenges[num_aux_challenges], aux_out[num_aux_columns][n] followed by [num_aux_public_inputs] closing sums, user):
// the caller's second-round column generator
typedef void (*vxo_stark_aux_fn)(const u64* challenges, u64* aux_out, void* user);
static StarkDesc vxo_to_desc(const vxo_stark_desc* d) {
  StarkDesc s;
  s.degree_bits = d->degree_bits, s.num_columns = d->num_columns, s.num_public_inputs = d->num_public_inputs;
  s.rate_bits = d->rate_bits, s.cap_height = d->cap_height, s.pow_bits = d->pow_bits, s.num_query_rounds = d->num_query_rounds;
  s.num_challenges = d->num_challenges, s.constraint_degree = d->constraint_degree;
  s.num_aux_columns = d->num_aux_columns, s.num_aux_challenges = d->num_aux_challenges, s.num_aux_public_inputs = d->num_aux_public_inputs;
  s.program.assign(d->program, d->program + d->program_len);
  if (d->override_flags & 2) s.arity_bits.assign(d->fri_reduction_arity_bits, d->fri_reduction_arity_bits + d->num_fri_reduction_arity_bits);
  else s.default_arities();
  s.openings_digest = (d->override_flags & 8u) != 0;      // VX_STARK_OPENINGS_DIGEST
  return s;
}
static std::vector<std::vector<u64>> vxo_to_trace(const vxo_stark_desc* d, const u64* trace) {
  const size_t n = (size_t)1 << d->degree_bits;
  std::vector<std::vector<u64>> tr(d->num_columns);
  for (int c = 0; c < d->num_columns; ++c) {
    tr[c].assign(trace + (size_t)c * n, trace + (size_t)(c + 1) * n);
    for (auto& x : tr[c]) x = canon(x);
  }
  return tr;
}
long long vxo_stark_prove3(const vxo_stark_desc* d, const u64* trace, const u64* pis, const u64* pow_hint, vxo_stark_aux_fn aux_fn, void* user,
                           const u64* shared_challenges, uint8_t* out, size_t cap, char* err, size_t err_cap);
// trace cap of one table (cap_out [2^cap_height][4]) and the joint challenges over several tables' caps
int vxo_stark_trace_cap(const vxo_stark_desc* d, const u64* trace, u64* cap_out) {
  try {
    const std::vector<Hash> cap = stark_trace_cap(vxo_to_desc(d), vxo_to_trace(d, trace));
    for (size_t i = 0; i < cap.size(); ++i) memcpy(cap_out + 4 * i, cap[i].e, 32);
    return 0;
  } catch (const std::exception&) {
    return -1;
  }
}
int vxo_stark_joint_challenges(const u64* const* caps, const int32_t* cap_heights, int ntables, int n, u64* out) {
  std::vector<std::vector<Hash>> cs(ntables);
  for (int t = 0; t < ntables; ++t) {
    cs[t].resize((size_t)1 << cap_heights[t]);
    for (size_t i = 0; i < cs[t].size(); ++i) memcpy(cs[t][i].e, caps[t] + 4 * i, 32);
  }
  const std::vector<u64> ch = stark_joint_challenges(cs, n);
  memcpy(out, ch.data(), ch.size() * 8);
  return 0;
}
long long vxo_stark_prove2(const vxo_stark_desc* d, const u64* trace, const u64* pis, const u64* pow_hint, vxo_stark_aux_fn aux_fn, void* user,
                           uint8_t* out, size_t cap, char* err, size_t err_cap);
long long vxo_stark_prove(const vxo_stark_desc* d, const u64* trace, const u64* pis, const u64* pow_hint, uint8_t* out, size_t cap,
                          char* err, size_t err_cap) {
  return vxo_stark_prove2(d, trace, pis, pow_hint, nullptr, nullptr, out, cap, err, err_cap);
}
long long vxo_stark_prove2(const vxo_stark_desc* d, const u64* trace, const u64* pis, const u64* pow_hint, vxo_stark_aux_fn aux_fn, void* user,
                           uint8_t* out, size_t cap, char* err, size_t err_cap) {
  return vxo_stark_prove3(d, trace, pis, pow_hint, aux_fn, user, nullptr, out, cap, err, err_cap);
}
long long vxo_stark_prove3(const vxo_stark_desc* d, const u64* trace, const u64* pis, const u64* pow_hint, vxo_stark_aux_fn aux_fn, void* user,
                           const u64* shared_challenges, uint8_t* out, size_t cap, char* err, size_t err_cap) {
  try {
    const StarkDesc s = vxo_to_desc(d);
    const size_t n = (size_t)1 << d->degree_bits;
    const std::vector<std::vector<u64>> tr = vxo_to_trace(d, trace);
    std::vector<u64> pi(pis, pis + d->num_public_inputs);
    ProveOptions opt;
    if (pow_hint) opt.has_pow_hint = true, opt.pow_hint = *pow_hint;
    StarkAuxFn fn;
    if (aux_fn)
      fn = [&](const std::vector<u64>& ch, std::vector<u64>& aux_pis) {
        std::vector<u64> flat((size_t)s.num_aux_columns * n + (size_t)s.num_aux_public_inputs);
        aux_fn(ch.data(), flat.data(), user);
        std::vector<std::vector<u64>> cols(s.num_aux_columns);
        for (int c = 0; c < s.num_aux_columns; ++c) cols[c].assign(flat.begin() + (size_t)c * n, flat.begin() + (size_t)(c + 1) * n);
        aux_pis.assign(flat.begin() + (size_t)s.num_aux_columns * n, flat.end());
        return cols;
      };
    std::vector<u64> shared;
    if (shared_challenges) shared.assign(shared_challenges, shared_challenges + s.num_aux_challenges);
    std::vector<uint8_t> bytes = serialize_stark_proof(stark_prove(s, tr, pi, opt, fn, shared_challenges ? &shared : nullptr));
    if (bytes.size() > cap) {
      snprintf(err, err_cap, "output buffer too small: need %zu bytes", bytes.size());
      return -1;
    }
    memcpy(out, bytes.data(), bytes.size());
    return (long long)bytes.size();
  } catch (const std::exception& e) {
    snprintf(err, err_cap, "%s", e.what());
    return -1;
  }
}
}  // extern "C"
