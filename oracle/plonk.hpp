// ORACLE — TEST INFRASTRUCTURE ONLY (see field.hpp header for the full notice and parity status).
// Restates, for plonky2 v0.2.0 with CircuitConfig::standard_recursion_config():
//   plonky2/src/iop/challenger.rs                         Challenger (duplex sponge, overwrite mode)
//   plonky2/src/plonk/prover.rs                           prove_with_partition_witness, compute_quotient_polys,
//                                                         wires_permutation_partial_products_round
//   plonky2/src/plonk/vanishing_poly.rs                   eval_vanishing_poly(_base_batch), evaluate_gate_constraints
//   plonky2/src/plonk/plonk_common.rs                     ZeroPolyOnCoset, eval_l_0, reduce_with_powers(_multi)
//   plonky2/src/plonk/permutation_argument.rs / circuit_builder.rs   sigma polynomials, k_is
//   plonky2/src/gates/{noop,constant,public_input,arithmetic_base,poseidon}.rs + gates/selectors.rs
//   plonky2/src/plonk/proof.rs                            OpeningSet, Proof
//   plonky2/src/fri/{oracle,prover,verifier,reduction_strategies,structure,proof}.rs
//   plonky2/src/plonk/verifier.rs                         verify_with_challenges
//   plonky2/src/util/serialization/mod.rs                 write_proof_with_public_inputs   
// "parity unpinned": upstream source is absent; conventions are recollection.
// Internal consistency is enforced the way the reference's own tests do it
// (/root/reference/circuits/header_range.rs:167-170: prove -> verify): the restated verifier below must
// accept restated proofs, and Fiat-Shamir makes nearly every ordering mistake fatal at verify time.
#pragma once
#include "poly.hpp"
#include <stdexcept>
#include <string>

namespace vxo {

// ---------------------------------------------------------------------------------------------
// iop/challenger.rs
// ---------------------------------------------------------------------------------------------
struct Challenger {
  State sponge{};
  std::vector<u64> input, output;
  void duplexing() {
    assert(input.size() <= (size_t)SPONGE_RATE);
    for (size_t i = 0; i < input.size(); ++i) sponge[i] = input[i];
    input.clear();
    permute(sponge);
    output.assign(sponge.begin(), sponge.begin() + SPONGE_RATE);
  }
  void observe_element(u64 e) {
    output.clear();
    input.push_back(e);
    if (input.size() == (size_t)SPONGE_RATE) duplexing();
  }
  void observe_elements(const u64* e, size_t n) {
    for (size_t i = 0; i < n; ++i) observe_element(e[i]);
  }
  void observe_hash(const Hash& h) { observe_elements(h.e, 4); }
  void observe_cap(const std::vector<Hash>& cap) {
    for (const Hash& h : cap) observe_hash(h);
  }
  void observe_ext(Ext x) {
    observe_element(x.a);
    observe_element(x.b);
  }
  u64 get_challenge() {
    if (!input.empty() || output.empty()) duplexing();
    u64 r = output.back();
    output.pop_back();
    return r;
  }
  Ext get_extension_challenge() {
    u64 a = get_challenge();
    u64 b = get_challenge();
    return Ext(a, b);
  }
};

// ---------------------------------------------------------------------------------------------
// Circuit description (CommonCircuitData + the prover-only parts the hot path needs).
// ---------------------------------------------------------------------------------------------
enum GateType { GATE_NOOP = 0, GATE_CONSTANT = 1, GATE_PUBLIC_INPUT = 2, GATE_ARITHMETIC = 3, GATE_POSEIDON = 4, GATE_PROGRAM = 5 };

// Constraint-program opcodes (same encoding as include/vxprover.h VX_OP_*; restated, not shared)
enum ProgOp { OP_END = 0, OP_LDW = 1, OP_LDC = 2, OP_LDI = 3, OP_ADD = 4, OP_SUB = 5, OP_MUL = 6, OP_PUSH = 7, OP_LDP = 8 };

struct Gate {
  int type = 0;
  int param = 0;           // ArithmeticGate: num_ops; ConstantGate: num_consts; program gate: degree
  int selector_index = 0;  // which selector polynomial (gates/selectors.rs SelectorsInfo)
  int group_start = 0, group_end = 0;
  std::vector<u64> program;  // GATE_PROGRAM: straight-line constraint program
  int num_constraints() const {
    switch (type) {
      case GATE_NOOP: return 0;
      case GATE_CONSTANT: return param;
      case GATE_PUBLIC_INPUT: return 4;
      case GATE_ARITHMETIC: return param;
      case GATE_PROGRAM: {
        int n = 0;
        for (size_t pc = 0; pc < program.size(); ++pc) {
          int op = (int)(program[pc] & 0xFF);
          if (op == OP_END) break;
          if (op == OP_PUSH) ++n;
          if (op == OP_LDI) ++pc;
        }
        return n;
      }
      default: return 123;  // PoseidonGate: 12*7 + 22 + 12 + 1 + 4
    }
  }
};

static const u64 UNUSED_SELECTOR = 0xFFFFFFFFULL;  // gates/selectors.rs: u32::MAX

struct Circuit {
  int degree_bits = 0;
  int num_wires = 135, num_routed_wires = 80, num_challenges = 2;
  int rate_bits = 3, cap_height = 4, pow_bits = 16, num_query_rounds = 28;
  int quotient_degree_factor = 8;
  std::vector<Gate> gates;  // sorted by (degree, id) as circuit_builder.rs does
  int num_selectors = 0;
  int num_constants = 0;  // selectors + gate constants (CommonCircuitData::num_constants)
  std::vector<u64> k_is;
  std::vector<std::pair<uint32_t, uint32_t>> public_inputs;  // (row, wire) targets
  // lookup argument (gates/lookup.rs, gates/lookup_table.rs, circuit_builder.rs::add_all_lookups); num_luts = 0: none
  int num_lookup_selectors = 0;                                 // LookupSelectors::StartEnd (= 4) + number of tables
  std::vector<std::vector<std::pair<u64, u64>>> luts;          // CommonCircuitData::luts: (input, output) pairs
  struct LookupWire {
    size_t last_lu_gate, last_lut_gate, first_lut_gate;
  };
  std::vector<LookupWire> lookup_rows;                          // ProverOnlyCircuitData::lookup_rows
  bool has_lookup() const { return !luts.empty(); }
  int num_lu_slots() const { return num_routed_wires / 2; }     // LookupGate::num_slots
  int num_lut_slots() const { return num_routed_wires / 3; }    // LookupTableGate::num_slots
  int lookup_degree() const { return quotient_degree_factor - 1; }
  int num_sldc_polys() const { return (num_lu_slots() + lookup_degree() - 1) / lookup_degree(); }
  int num_lookup_polys() const { return has_lookup() ? 1 + num_sldc_polys() : 0; }   // per challenge: RE + partial Sum/LDCs
  int gate_const_base() const { return num_selectors + num_lookup_selectors; }       // gates see constants after BOTH selector kinds
  // derived
  int num_gate_constraints = 0;
  std::vector<int> reduction_arity_bits;
  PolynomialBatch constants_sigmas;  // [selectors.., constants.., sigmas(80)]
  Hash circuit_digest;
  // values the caller holds in CommonCircuitData / VerifierOnlyCircuitData and may pass instead of having them derived
  bool has_digest_override = false, has_arity_override = false;
  Hash digest_override;
  std::vector<int> arity_override;

  size_t n() const { return (size_t)1 << degree_bits; }
  int num_partial_products() const { return (num_routed_wires + quotient_degree_factor - 1) / quotient_degree_factor - 1; }
  int num_preprocessed() const { return num_constants + num_routed_wires; }
  int num_zs_pp() const { return num_challenges * (1 + num_partial_products()); }                 // Z + partial products
  int num_zs_pp_lookup() const { return num_zs_pp() + num_challenges * num_lookup_polys(); }       // ... + lookup polynomials (one batch)
  int num_quotient() const { return num_challenges * quotient_degree_factor; }

  // fri/reduction_strategies.rs: ConstantArityBits(4, 5)
  void compute_fri_params() {
    reduction_arity_bits.clear();
    if (has_arity_override) {  // FriParams::reduction_arity_bits as given by the caller
      reduction_arity_bits = arity_override;
      return;
    }
    int db = degree_bits;
    const int arity_bits = 4, final_poly_bits = 5;
    while (db > final_poly_bits && db + rate_bits - arity_bits >= cap_height) {
      reduction_arity_bits.push_back(arity_bits);
      assert(db >= arity_bits);
      db -= arity_bits;
    }
  }
  // circuit_digest = hash_no_pad(cap.flatten() || hash_pad(domain_separator).elements || [degree_bits])   (circuit_builder.rs::build,
  // recalled; round 4: the separator enters as its PADDED hash).  plonky2x leaves the separator empty; hashing.rs::hash_pad pushes a 1,
  // zeros until one slot short of a multiple of the rate (8), and a closing 1.  THE one place of the rule in the oracle.
  static Hash digest_of_cap(const std::vector<Hash>& cap, int degree_bits) {
    std::vector<u64> sep;                    // the (empty) domain separator
    sep.push_back(1);
    while ((sep.size() + 1) % 8 != 0) sep.push_back(0);
    sep.push_back(1);
    const Hash sep_digest = hash_no_pad(sep.data(), sep.size());
    std::vector<u64> pre;
    for (const Hash& h : cap)
      for (int i = 0; i < 4; ++i) pre.push_back(h.e[i]);
    for (int i = 0; i < 4; ++i) pre.push_back(sep_digest.e[i]);
    pre.push_back((u64)degree_bits);
    return hash_no_pad(pre.data(), pre.size());
  }
  // circuit_builder.rs::build: constants_sigmas_commitment + circuit_digest
  void finalize(std::vector<std::vector<u64>>&& constants_sigmas_values) {
    num_gate_constraints = 0;
    for (const Gate& g : gates) num_gate_constraints = std::max(num_gate_constraints, g.num_constraints());
    compute_fri_params();
    constants_sigmas.from_values(std::move(constants_sigmas_values), rate_bits, cap_height);
    circuit_digest = has_digest_override ? digest_override : digest_of_cap(constants_sigmas.tree.cap(), degree_bits);
  }
};

// ---------------------------------------------------------------------------------------------
// Gate constraint formulas, generic over T in {Fp, Ext}  (gates/*.rs eval_unfiltered)
// ---------------------------------------------------------------------------------------------
template <class T>
static inline T t_const(u64 c) { return T(c); }
template <class T>
static inline T sbox_t(T x) {
  T x2 = x * x, x4 = x2 * x2, x3 = x * x2;
  return x3 * x4;
}
template <class T>
static inline void mds_t(T* s) {
  T o[12];
  for (int r = 0; r < 12; ++r) {
    T acc = T(0);
    for (int i = 0; i < 12; ++i) acc = acc + scale(s[(i + r) % 12], MDS_CIRC[i]);
    acc = acc + scale(s[r], MDS_DIAG[r]);
    o[r] = acc;
  }
  for (int i = 0; i < 12; ++i) s[i] = o[i];
}

// vars: local_constants (AFTER remove_prefix(num_selectors)), local_wires, public_inputs_hash
template <class T>
static void eval_gate_unfiltered(const Gate& g, const T* consts, const T* wires, const Hash& pih, std::vector<T>& out) {
  out.clear();
  switch (g.type) {
    case GATE_NOOP: break;
    case GATE_PROGRAM: {  // caller-supplied straight-line program: the gate's eval_unfiltered as data
      T r[64];
      for (size_t pc = 0; pc < g.program.size(); ++pc) {
        const u64 ins = g.program[pc];
        const int op = (int)(ins & 0xFF), dst = (int)((ins >> 8) & 0x3F), a = (int)((ins >> 16) & 0xFFFF), b = (int)((ins >> 32) & 0xFFFF);
        if (op == OP_END) break;
        switch (op) {
          case OP_LDW: r[dst] = wires[a]; break;
          case OP_LDC: r[dst] = consts[a]; break;
          case OP_LDI: r[dst] = T(canon(g.program[++pc])); break;
          case OP_ADD: r[dst] = r[a & 63] + r[b & 63]; break;
          case OP_SUB: r[dst] = r[a & 63] - r[b & 63]; break;
          case OP_MUL: r[dst] = r[a & 63] * r[b & 63]; break;
          case OP_PUSH: out.push_back(r[a & 63]); break;
          case OP_LDP: r[dst] = T(pih.e[a & 3]); break;
          default: throw std::runtime_error("bad opcode in a constraint program");
        }
      }
      break;
    }
    case GATE_CONSTANT:  // gates/constant.rs: constants[i] - wires[i]
      for (int i = 0; i < g.param; ++i) out.push_back(consts[i] - wires[i]);
      break;
    case GATE_PUBLIC_INPUT:  // gates/public_input.rs: wires[i] - public_inputs_hash[i]
      for (int i = 0; i < 4; ++i) out.push_back(wires[i] - T(pih.e[i]));
      break;
    case GATE_ARITHMETIC:  // gates/arithmetic_base.rs: output - (m0*m1*c0 + addend*c1)
      for (int i = 0; i < g.param; ++i) {
        T m0 = wires[4 * i], m1 = wires[4 * i + 1], ad = wires[4 * i + 2], o = wires[4 * i + 3];
        out.push_back(o - (m0 * m1 * consts[0] + ad * consts[1]));
      }
      break;
    case GATE_POSEIDON: {  // gates/poseidon.rs (partial rounds in the naive form: identical polynomials)
      const int WIRE_SWAP = 24, START_DELTA = 25, START_FULL_0 = 29, START_PARTIAL = 65, START_FULL_1 = 87;
      T swap = wires[WIRE_SWAP];
      out.push_back(swap * (swap - T(1)));
      for (int i = 0; i < 4; ++i) out.push_back(swap * (wires[i + 4] - wires[i]) - wires[START_DELTA + i]);
      T st[12];
      for (int i = 0; i < 4; ++i) {
        T d = wires[START_DELTA + i];
        st[i] = wires[i] + d;
        st[i + 4] = wires[i + 4] - d;
      }
      for (int i = 8; i < 12; ++i) st[i] = wires[i];
      int round = 0;
      for (int r = 0; r < 4; ++r) {
        for (int i = 0; i < 12; ++i) st[i] = st[i] + T(ROUND_CONSTANTS[12 * round + i]);
        if (r != 0)
          for (int i = 0; i < 12; ++i) {
            T in = wires[START_FULL_0 + 12 * (r - 1) + i];
            out.push_back(st[i] - in);
            st[i] = in;
          }
        for (int i = 0; i < 12; ++i) st[i] = sbox_t(st[i]);
        mds_t(st);
        ++round;
      }
      for (int r = 0; r < 22; ++r) {
        for (int i = 0; i < 12; ++i) st[i] = st[i] + T(ROUND_CONSTANTS[12 * round + i]);
        T in = wires[START_PARTIAL + r];
        out.push_back(st[0] - in);
        st[0] = sbox_t(in);
        mds_t(st);
        ++round;
      }
      for (int r = 0; r < 4; ++r) {
        for (int i = 0; i < 12; ++i) st[i] = st[i] + T(ROUND_CONSTANTS[12 * round + i]);
        for (int i = 0; i < 12; ++i) {
          T in = wires[START_FULL_1 + 12 * r + i];
          out.push_back(st[i] - in);
          st[i] = in;
        }
        for (int i = 0; i < 12; ++i) st[i] = sbox_t(st[i]);
        mds_t(st);
        ++round;
      }
      for (int i = 0; i < 12; ++i) out.push_back(st[i] - wires[12 + i]);
      break;
    }
  }
}

// gates/selectors.rs + gates/gate.rs::compute_filter
template <class T>
static inline T compute_filter(int row, const Gate& g, T s, bool many_selectors) {
  T f = T(1);
  for (int i = g.group_start; i < g.group_end; ++i)
    if (i != row) f = f * (T((u64)i) - s);
  if (many_selectors) f = f * (T(UNUSED_SELECTOR) - s);
  return f;
}

// vanishing_poly.rs::evaluate_gate_constraints: per-index SUM over gates of filter * constraint
template <class T>
static void evaluate_gate_constraints(const Circuit& c, const T* local_constants, const T* wires, const Hash& pih,
                                      std::vector<T>& constraints) {
  constraints.assign(c.num_gate_constraints, T(0));
  std::vector<T> tmp;
  for (size_t gi = 0; gi < c.gates.size(); ++gi) {
    const Gate& g = c.gates[gi];
    T filter = compute_filter<T>((int)gi, g, local_constants[g.selector_index], c.num_selectors > 1);
    eval_gate_unfiltered<T>(g, local_constants + c.gate_const_base(), wires, pih, tmp);
    for (size_t i = 0; i < tmp.size(); ++i) constraints[i] = constraints[i] + filter * tmp[i];
  }
}

// vanishing_poly.rs::get_lut_poly: the table as a polynomial in delta — sum_k (inp_k + b out_k) delta^(degree-1-k)
static u64 get_lut_poly(const Circuit& c, int lut_index, const u64* deltas, size_t degree) {
  const u64 b = deltas[1], delta = deltas[3];
  const auto& lut = c.luts[lut_index];
  u64 acc = 0;  // coefficients reversed: Horner over k = 0 .. degree-1 with zero padding after the table
  for (size_t k = 0; k < degree; ++k) {
    const u64 coeff = k < lut.size() ? add(lut[k].first % P, mul(b, lut[k].second % P)) : 0;
    acc = add(mul(acc, delta), coeff);
  }
  return acc;
}

// vanishing_poly.rs::check_lookup_constraints for ONE challenge (deltas = [a, b, alpha, delta] of that challenge).
// lookup_selectors: TransSre, TransLdc, InitSre, LastLdc, then one "ends" selector per table.
// local_zs / next_zs: [RE, SLDC_0 .. SLDC_{k-1}] at x and g x.  Appends 4 + num_luts + 2 k constraints.
template <class T>
static void check_lookup_constraints(const Circuit& c, const T* lookup_selectors, const T* wires, const T* local_zs, const T* next_zs,
                                     const u64* deltas, std::vector<T>& out) {
  const int num_lu_slots = c.num_lu_slots(), num_lut_slots = c.num_lut_slots(), lu_degree = c.lookup_degree();
  const int num_sldc = c.num_sldc_polys(), lut_degree = (num_lut_slots + num_sldc - 1) / num_sldc;
  const T z_re = local_zs[0], next_z_re = next_zs[0];
  const T* z_x = local_zs + 1;
  const T* z_gx = next_zs + 1;
  const u64 da = deltas[0], db = deltas[1], dalpha = deltas[2], ddelta = deltas[3];
  // combos: looking (LookupGate wires 2i, 2i+1), looked for Sum (a) and for RE (b) (LookupTableGate wires 3i, 3i+1)
  std::vector<T> looking(num_lu_slots), looked(num_lut_slots), looked_re(num_lut_slots);
  for (int i = 0; i < num_lu_slots; ++i) looking[i] = wires[2 * i] + scale(wires[2 * i + 1], da);
  for (int i = 0; i < num_lut_slots; ++i) {
    looked[i] = wires[3 * i] + scale(wires[3 * i + 1], da);
    looked_re[i] = wires[3 * i] + scale(wires[3 * i + 1], db);
  }
  // LDC ends at zero; Sum and RE start from zero
  out.push_back(lookup_selectors[3] * z_x[num_sldc - 1]);
  out.push_back(lookup_selectors[2] * z_x[0]);
  out.push_back(lookup_selectors[2] * z_re);
  // RE on a table's last row equals the table polynomial
  for (size_t r = 0; r < c.luts.size(); ++r) {
    const size_t rows = (c.luts[r].size() + num_lut_slots - 1) / num_lut_slots;
    out.push_back(lookup_selectors[4 + r] * (z_re - T(get_lut_poly(c, (int)r, deltas, (size_t)num_lut_slots * rows))));
  }
  // RE row transition
  {
    T cur = next_z_re;
    for (int i = 0; i < num_lut_slots; ++i) cur = scale(cur, ddelta) + looked_re[i];
    out.push_back(lookup_selectors[0] * (z_re - cur));
  }
  for (int poly = 0; poly < num_sldc; ++poly) {
    const int t0 = poly * lut_degree, t1 = std::min((poly + 1) * lut_degree, num_lut_slots);
    const int u0 = poly * lu_degree, u1 = std::min((poly + 1) * lu_degree, num_lu_slots);
    auto prod_except = [&](const std::vector<T>& v, int lo, int hi, int skip) {
      T acc = T(1);
      for (int j = lo; j < hi; ++j)
        if (j != skip) acc = acc * (T(dalpha) - v[j]);
      return acc;
    };
    const T lut_prod = prod_except(looked, t0, t1, -1), lu_prod = prod_except(looking, u0, u1, -1);
    T lu_sum = T(0), lut_sum_mul = T(0);
    for (int i = u0; i < u1; ++i) lu_sum = lu_sum + prod_except(looking, u0, u1, i);
    for (int i = t0; i < t1; ++i) lut_sum_mul = lut_sum_mul + wires[3 * i + 2] * prod_except(looked, t0, t1, i);
    const T prev = poly == 0 ? z_gx[num_sldc - 1] : z_x[poly - 1];
    out.push_back(lookup_selectors[0] * (lut_prod * (z_x[poly] - prev) - lut_sum_mul));   // Sum transition
    out.push_back(lookup_selectors[1] * (lu_prod * (z_x[poly] - prev) + lu_sum));         // LDC transition
  }
}

// vanishing_poly.rs::eval_vanishing_poly (one point).  x: the point; l0: L_0(x).
// local_constants: num_constants values; s_sigmas: 80; zs/next_zs: per challenge; pps: challenge-major;
// lookup_zs / next_lookup_zs: challenge-major [RE, SLDCs]; deltas: 4 per challenge.
template <class T>
static void eval_vanishing(const Circuit& c, T x, T l0, const T* local_constants, const T* s_sigmas, const T* wires,
                           const T* zs, const T* next_zs, const T* pps, const Hash& pih, const u64* betas,
                           const u64* gammas, const T* alphas, T* out /* num_challenges */, const T* lookup_zs = nullptr,
                           const T* next_lookup_zs = nullptr, const u64* deltas = nullptr) {
  const int npp = c.num_partial_products(), deg = c.quotient_degree_factor, nr = c.num_routed_wires;
  std::vector<T> z1_terms, pp_terms, lookup_terms, gate_terms;
  for (int ch = 0; ch < c.num_challenges; ++ch) {
    z1_terms.push_back(l0 * (zs[ch] - T(1)));
    std::vector<T> num(nr), den(nr);
    for (int j = 0; j < nr; ++j) {
      T s_id = scale(x, c.k_is[j]);
      num[j] = wires[j] + scale(s_id, betas[ch]) + T(gammas[ch]);
      den[j] = wires[j] + scale(s_sigmas[j], betas[ch]) + T(gammas[ch]);
    }
    // check_partial_products: accs = [z_x, pp_0.., z_gx]
    std::vector<T> accs;
    accs.push_back(zs[ch]);
    for (int k = 0; k < npp; ++k) accs.push_back(pps[ch * npp + k]);
    accs.push_back(next_zs[ch]);
    int chunk = 0;
    for (int j0 = 0; j0 < nr; j0 += deg, ++chunk) {
      T np = T(1), dp = T(1);
      for (int j = j0; j < std::min(nr, j0 + deg); ++j) {
        np = np * num[j];
        dp = dp * den[j];
      }
      pp_terms.push_back(accs[chunk] * np - accs[chunk + 1] * dp);
    }
    if (c.has_lookup()) {
      const int nlp = c.num_lookup_polys();
      check_lookup_constraints<T>(c, local_constants + c.num_selectors, wires, lookup_zs + ch * nlp, next_lookup_zs + ch * nlp, deltas + 4 * ch,
                                  lookup_terms);
    }
  }
  evaluate_gate_constraints<T>(c, local_constants, wires, pih, gate_terms);
  std::vector<T> terms;
  terms.insert(terms.end(), z1_terms.begin(), z1_terms.end());
  terms.insert(terms.end(), pp_terms.begin(), pp_terms.end());
  terms.insert(terms.end(), lookup_terms.begin(), lookup_terms.end());
  terms.insert(terms.end(), gate_terms.begin(), gate_terms.end());
  // plonk_common.rs::reduce_with_powers_multi: sum_i term_i alpha^i (Horner from the last term)
  for (int ch = 0; ch < c.num_challenges; ++ch) {
    T acc = T(0);
    for (size_t i = terms.size(); i-- > 0;) acc = acc * alphas[ch] + terms[i];
    out[ch] = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// Proof objects (plonk/proof.rs, fri/proof.rs)
// ---------------------------------------------------------------------------------------------
struct OpeningSet {
  std::vector<Ext> constants, plonk_sigmas, wires, plonk_zs, plonk_zs_next, partial_products, quotient_polys, lookup_zs, lookup_zs_next;
};
struct FriInitialTreeProof {
  std::vector<std::vector<u64>> evals;        // per oracle: leaf values
  std::vector<std::vector<Hash>> proofs;      // per oracle: Merkle siblings
};
struct FriQueryStep {
  std::vector<Ext> evals;  // 2^arity_bits values of the coset (uncompressed proof keeps all of them)
  std::vector<Hash> proof;
};
struct FriQueryRound {
  FriInitialTreeProof initial;
  std::vector<FriQueryStep> steps;
};
struct FriProof {
  std::vector<std::vector<Hash>> commit_phase_caps;
  std::vector<FriQueryRound> query_rounds;
  std::vector<Ext> final_poly;
  u64 pow_witness = 0;
};
struct Proof {
  std::vector<Hash> wires_cap, zs_pp_cap, quotient_cap;
  OpeningSet openings;
  FriProof fri;
  std::vector<u64> public_inputs;
};

// util/serialization: write_proof_with_public_inputs.  Field = u64 LE canonical.
struct ByteWriter {
  std::vector<uint8_t> b;
  void u8(uint8_t v) { b.push_back(v); }
  void f(u64 v) {
    for (int i = 0; i < 8; ++i) b.push_back((uint8_t)(v >> (8 * i)));
  }
  void ext(Ext e) {
    f(e.a);
    f(e.b);
  }
  void hash(const Hash& h) {
    for (int i = 0; i < 4; ++i) f(h.e[i]);
  }
  void cap(const std::vector<Hash>& c) {
    for (const Hash& h : c) hash(h);
  }
  void extvec(const std::vector<Ext>& v) {
    for (Ext e : v) ext(e);
  }
  void merkle_proof(const std::vector<Hash>& p) {
    u8((uint8_t)p.size());
    for (const Hash& h : p) hash(h);
  }
};
static std::vector<uint8_t> serialize_proof(const Proof& p) {
  ByteWriter w;
  w.cap(p.wires_cap);
  w.cap(p.zs_pp_cap);
  w.cap(p.quotient_cap);
  const OpeningSet& o = p.openings;
  w.extvec(o.constants);
  w.extvec(o.plonk_sigmas);
  w.extvec(o.wires);
  w.extvec(o.plonk_zs);
  w.extvec(o.plonk_zs_next);
  w.extvec(o.lookup_zs);  // util/serialization write_opening_set: lookup openings before the partial products
  w.extvec(o.lookup_zs_next);
  w.extvec(o.partial_products);
  w.extvec(o.quotient_polys);
  for (const auto& c : p.fri.commit_phase_caps) w.cap(c);
  for (const FriQueryRound& q : p.fri.query_rounds) {
    for (size_t t = 0; t < q.initial.evals.size(); ++t) {
      for (u64 v : q.initial.evals[t]) w.f(v);
      w.merkle_proof(q.initial.proofs[t]);
    }
    for (const FriQueryStep& s : q.steps) {
      w.extvec(s.evals);
      w.merkle_proof(s.proof);
    }
  }
  w.extvec(p.fri.final_poly);
  w.f(p.fri.pow_witness);
  for (u64 v : p.public_inputs) w.f(v);
  return w.b;
}

// ---------------------------------------------------------------------------------------------
// Prover
// ---------------------------------------------------------------------------------------------
struct ProverTimings {
  double wires_commit = 0, zs_pp = 0, zs_pp_commit = 0, quotient_eval = 0, quotient_commit = 0, openings = 0, fri = 0,
         total = 0;
};

// prover.rs::wires_permutation_partial_products_round for all challenges; returns columns in the batch order
// [Z_0, Z_1, pp_{0,0..}, pp_{1,0..}] (prover.rs: "Z is expected at the front of our batch").
static std::vector<std::vector<u64>> compute_zs_partial_products(const Circuit& c, const std::vector<std::vector<u64>>& wires,
                                                                 const std::vector<std::vector<u64>>& sigma_values,
                                                                 const u64* betas, const u64* gammas) {
  const size_t n = c.n();
  const int nr = c.num_routed_wires, deg = c.quotient_degree_factor, npp = c.num_partial_products(), nch = c.num_challenges;
  const int nchunks = npp + 1;
  std::vector<std::vector<u64>> out(nch * (1 + npp), std::vector<u64>(n));
  u64 w = root_of_unity(c.degree_bits);
  std::vector<u64> subgroup(n);
  subgroup[0] = 1;
  for (size_t i = 1; i < n; ++i) subgroup[i] = mul(subgroup[i - 1], w);
  for (int ch = 0; ch < nch; ++ch) {
    std::vector<u64> chunk_products(n * nchunks);
    long long nn = (long long)n;
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < nn; ++i) {
      std::vector<u64> num(nr), den(nr);
      for (int j = 0; j < nr; ++j) {
        u64 wv = wires[j][i];
        u64 s_id = mul(c.k_is[j], subgroup[i]);
        num[j] = add(add(wv, mul(betas[ch], s_id)), gammas[ch]);
        den[j] = add(add(wv, mul(betas[ch], sigma_values[j][i])), gammas[ch]);
      }
      batch_inverse(den.data(), nr);
      for (int k = 0; k < nchunks; ++k) {
        u64 p = 1;
        for (int j = k * deg; j < std::min(nr, (k + 1) * deg); ++j) p = mul(p, mul(num[j], den[j]));
        chunk_products[i * nchunks + k] = p;
      }
    }
    u64 z = 1;
    for (size_t i = 0; i < n; ++i) {
      out[ch][i] = z;  // Z(x_i)
      u64 acc = z;
      for (int k = 0; k < nchunks; ++k) {
        acc = mul(acc, chunk_products[i * nchunks + k]);
        if (k < npp) out[nch + ch * npp + k][i] = acc;
      }
      z = acc;  // Z(g x_i)
    }
  }
  return out;
}

// prover.rs::compute_lookup_polys for ONE challenge (deltas = [a, b, alpha, delta]): [RE, SLDC_0 .. SLDC_{k-1}] as values on H
static std::vector<std::vector<u64>> compute_lookup_polys(const Circuit& c, const std::vector<std::vector<u64>>& w, const u64* deltas) {
  const size_t n = c.n();
  const int num_lu_slots = c.num_lu_slots(), max_lookup_degree = c.lookup_degree(), num_partial = c.num_sldc_polys();
  const int num_lut_slots = c.num_lut_slots(), max_lut_degree = (num_lut_slots + num_partial - 1) / num_partial;
  const u64 da = deltas[0], db = deltas[1], dalpha = deltas[2], ddelta = deltas[3];
  std::vector<std::vector<u64>> polys(num_partial + 1, std::vector<u64>(n, 0));
  for (const auto& lr : c.lookup_rows) {
    // table rows, from first_lut_gate down to last_lut_gate: partial Sums and RE
    for (size_t row = lr.first_lut_gate + 1; row-- > lr.last_lut_gate;) {
      std::vector<u64> inv_a(num_lut_slots);
      u64 new_re = polys[0][row + 1];
      for (int s2 = 0; s2 < num_lut_slots; ++s2) {
        const u64 inp = w[3 * s2][row], out = w[3 * s2 + 1][row];
        inv_a[s2] = sub(dalpha, add(inp, mul(da, out)));
        new_re = add(mul(new_re, ddelta), add(inp, mul(db, out)));
      }
      batch_inverse(inv_a.data(), inv_a.size());
      polys[0][row] = new_re;
      for (int slot = 0; slot < num_partial; ++slot) {
        u64 acc = slot != 0 ? polys[slot][row] : polys[num_partial][row + 1];
        for (int s2 = slot * max_lut_degree; s2 < std::min((slot + 1) * max_lut_degree, num_lut_slots); ++s2)
          acc = add(acc, mul(w[3 * s2 + 2][row], inv_a[s2]));
        polys[slot + 1][row] = acc;
      }
    }
    // looking rows, from last_lut_gate - 1 down to last_lu_gate: partial LDCs
    for (size_t row = lr.last_lut_gate; row-- > lr.last_lu_gate;) {
      std::vector<u64> inv_a(num_lu_slots);
      for (int s2 = 0; s2 < num_lu_slots; ++s2) inv_a[s2] = sub(dalpha, add(w[2 * s2][row], mul(da, w[2 * s2 + 1][row])));
      batch_inverse(inv_a.data(), inv_a.size());
      for (int slot = 0; slot < num_partial; ++slot) {
        const u64 prev = slot == 0 ? polys[num_partial][row + 1] : polys[slot][row];
        u64 sum = 0;
        for (int s2 = slot * max_lookup_degree; s2 < std::min((slot + 1) * max_lookup_degree, num_lu_slots); ++s2) sum = add(sum, inv_a[s2]);
        polys[slot + 1][row] = sub(prev, sum);
      }
    }
  }
  return polys;
}

struct ProveOptions {
  bool has_pow_hint = false;  // use this pow_witness instead of grinding (upstream's choice is nondeterministic)
  u64 pow_hint = 0;
};

static double now_s();

static Proof prove(const Circuit& c, const std::vector<std::vector<u64>>& wire_values, const ProveOptions& opt = ProveOptions(),
                   ProverTimings* tm = nullptr) {
  const size_t n = c.n();
  const int lg = c.degree_bits, rb = c.rate_bits, LG = lg + rb;
  const size_t N = (size_t)1 << LG;
  const int nch = c.num_challenges, npp = c.num_partial_products(), qdf = c.quotient_degree_factor;
  if ((int)wire_values.size() != c.num_wires) throw std::runtime_error("wire matrix has the wrong number of columns");
  double t_start = now_s(), t0 = t_start;
  Proof proof;
  for (auto& pi : c.public_inputs) proof.public_inputs.push_back(wire_values[pi.second][pi.first]);
  Hash pih = hash_no_pad(proof.public_inputs.data(), proof.public_inputs.size());

  PolynomialBatch wires_b;
  {
    std::vector<std::vector<u64>> v = wire_values;
    wires_b.from_values(std::move(v), rb, c.cap_height);
  }
  if (tm) tm->wires_commit = now_s() - t0, t0 = now_s();
  Challenger ch;
  ch.observe_hash(c.circuit_digest);
  ch.observe_hash(pih);
  ch.observe_cap(wires_b.tree.cap());
  std::vector<u64> betas(nch), gammas(nch), alphas(nch);
  for (int i = 0; i < nch; ++i) betas[i] = ch.get_challenge();
  for (int i = 0; i < nch; ++i) gammas[i] = ch.get_challenge();

  // sigma values on H: recover from the preprocessed batch's coefficients (fft of the coefficient form)
  std::vector<std::vector<u64>> sigma_values(c.num_routed_wires);
  {
    long long nr = c.num_routed_wires;
#pragma omp parallel for schedule(dynamic)
    for (long long j = 0; j < nr; ++j) {
      sigma_values[j] = c.constants_sigmas.coeffs[c.num_constants + j];
      fft_inplace(sigma_values[j].data(), lg);
    }
  }
  // lookup challenges (prover.rs): 4 per challenge; the first 2 * num_challenges of them ARE the betas and gammas,
  // the rest is drawn now
  std::vector<u64> deltas;
  const int nlp = c.num_lookup_polys();
  if (c.has_lookup()) {
    deltas = betas;
    deltas.insert(deltas.end(), gammas.begin(), gammas.end());
    for (int i = 0; i < 2 * nch; ++i) deltas.push_back(ch.get_challenge());
  }
  PolynomialBatch zs_b;
  {
    auto cols = compute_zs_partial_products(c, wire_values, sigma_values, betas.data(), gammas.data());
    for (int k = 0; k < nch && c.has_lookup(); ++k)
      for (auto& col : compute_lookup_polys(c, wire_values, deltas.data() + 4 * k)) cols.push_back(std::move(col));
    if (tm) tm->zs_pp = now_s() - t0, t0 = now_s();
    zs_b.from_values(std::move(cols), rb, c.cap_height);
  }
  if (tm) tm->zs_pp_commit = now_s() - t0, t0 = now_s();
  ch.observe_cap(zs_b.tree.cap());
  for (int i = 0; i < nch; ++i) alphas[i] = ch.get_challenge();

  // ---- compute_quotient_polys ----
  std::vector<std::vector<u64>> qvals(nch, std::vector<u64>(N));
  {
    // ZeroPolyOnCoset
    const size_t rate = (size_t)1 << rb;
    std::vector<u64> zh(rate), zh_inv(rate);
    u64 shift_n = pow(MULTIPLICATIVE_GENERATOR, n), g_rate = root_of_unity(rb), p = 1;
    for (size_t i = 0; i < rate; ++i) {
      zh[i] = sub(mul(shift_n, p), 1);
      zh_inv[i] = inv(zh[i]);
      p = mul(p, g_rate);
    }
    u64 wN = root_of_unity(LG);
    const size_t next_step = (size_t)1 << rb;  // quotient_degree_bits == rate_bits
    long long NN = (long long)N;
#pragma omp parallel
    {
      std::vector<Fp> lc(c.num_constants), ss(c.num_routed_wires), lw(c.num_wires), zs(nch), nzs(nch), pps(nch * npp), al(nch), res(nch);
      std::vector<Fp> lzs(nch * nlp + 1), nlzs(nch * nlp + 1);
      for (int i = 0; i < nch; ++i) al[i] = Fp(alphas[i]);
#pragma omp for schedule(static)
      for (long long i = 0; i < NN; ++i) {
        u64 x = mul(MULTIPLICATIVE_GENERATOR, pow(wN, (u64)i));
        const u64* cs = c.constants_sigmas.get_lde_values((size_t)i, 1);
        const u64* wv = wires_b.get_lde_values((size_t)i, 1);
        const u64* zv = zs_b.get_lde_values((size_t)i, 1);
        const u64* zn = zs_b.get_lde_values(((size_t)i + next_step) % N, 1);
        for (int k = 0; k < c.num_constants; ++k) lc[k] = Fp(cs[k]);
        for (int k = 0; k < c.num_routed_wires; ++k) ss[k] = Fp(cs[c.num_constants + k]);
        for (int k = 0; k < c.num_wires; ++k) lw[k] = Fp(wv[k]);
        for (int k = 0; k < nch; ++k) zs[k] = Fp(zv[k]), nzs[k] = Fp(zn[k]);
        for (int k = 0; k < nch * npp; ++k) pps[k] = Fp(zv[nch + k]);
        for (int k = 0; k < nch * nlp; ++k) lzs[k] = Fp(zv[nch * (1 + npp) + k]), nlzs[k] = Fp(zn[nch * (1 + npp) + k]);
        // eval_l_0(i, x) = Z_H(x) / (n (x - 1))
        u64 l0 = mul(zh[i % rate], inv(mul((u64)n % P, sub(x, 1))));
        eval_vanishing<Fp>(c, Fp(x), Fp(l0), lc.data(), ss.data(), lw.data(), zs.data(), nzs.data(), pps.data(), pih,
                           betas.data(), gammas.data(), al.data(), res.data(), lzs.data(), nlzs.data(), deltas.data());
        for (int k = 0; k < nch; ++k) qvals[k][i] = mul(res[k].v, zh_inv[i % rate]);
      }
    }
  }
  if (tm) tm->quotient_eval = now_s() - t0, t0 = now_s();
  PolynomialBatch quot_b;
  {
    std::vector<std::vector<u64>> chunks;
    for (int k = 0; k < nch; ++k) {
      coset_ifft_inplace(qvals[k].data(), LG, MULTIPLICATIVE_GENERATOR);
      // trim_to_len(quotient_degree = qdf * n): with qdf * n == N nothing is cut; otherwise the tail must be zero
      for (size_t i = (size_t)qdf * n; i < N; ++i)
        if (qvals[k][i]) throw std::runtime_error("Quotient has failed, the vanishing polynomial is not divisible by Z_H");
      for (int q = 0; q < qdf; ++q) chunks.emplace_back(qvals[k].begin() + q * n, qvals[k].begin() + (q + 1) * n);
    }
    quot_b.from_coeffs(std::move(chunks), rb, c.cap_height);
  }
  if (tm) tm->quotient_commit = now_s() - t0, t0 = now_s();
  ch.observe_cap(quot_b.tree.cap());
  Ext zeta = ch.get_extension_challenge();
  u64 g = root_of_unity(lg);
  {
    Ext zp = zeta;
    for (int i = 0; i < lg; ++i) zp = zp * zp;
    if (zp == Ext(1)) throw std::runtime_error("Opening point is in the subgroup.");
  }
  // ---- OpeningSet::new ----
  const PolynomialBatch* oracles[4] = {&c.constants_sigmas, &wires_b, &zs_b, &quot_b};
  auto eval_batch = [&](const PolynomialBatch& b, Ext z) {
    std::vector<Ext> r(b.ncols);
    long long m = (long long)b.ncols;
#pragma omp parallel for schedule(dynamic)
    for (long long k = 0; k < m; ++k) r[k] = eval_poly_ext(b.coeffs[k].data(), b.coeffs[k].size(), z);
    return r;
  };
  Ext gzeta = scale(zeta, g);
  {
    std::vector<Ext> cs = eval_batch(c.constants_sigmas, zeta);
    std::vector<Ext> zp = eval_batch(zs_b, zeta), zpn = eval_batch(zs_b, gzeta);
    OpeningSet& o = proof.openings;
    o.constants.assign(cs.begin(), cs.begin() + c.num_constants);
    o.plonk_sigmas.assign(cs.begin() + c.num_constants, cs.end());
    o.wires = eval_batch(wires_b, zeta);
    o.plonk_zs.assign(zp.begin(), zp.begin() + nch);
    o.plonk_zs_next.assign(zpn.begin(), zpn.begin() + nch);
    o.partial_products.assign(zp.begin() + nch, zp.begin() + c.num_zs_pp());
    o.lookup_zs.assign(zp.begin() + c.num_zs_pp(), zp.end());
    o.lookup_zs_next.assign(zpn.begin() + c.num_zs_pp(), zpn.end());
    o.quotient_polys = eval_batch(quot_b, zeta);
    // challenger.observe_openings(to_fri_openings): batch 0 then batch 1
    for (auto* v : {&o.constants, &o.plonk_sigmas, &o.wires, &o.plonk_zs, &o.partial_products, &o.quotient_polys, &o.lookup_zs})
      for (Ext e : *v) ch.observe_ext(e);
    for (auto* v : {&o.plonk_zs_next, &o.lookup_zs_next})
      for (Ext e : *v) ch.observe_ext(e);
  }
  if (tm) tm->openings = now_s() - t0, t0 = now_s();
  proof.wires_cap = wires_b.tree.cap();
  proof.zs_pp_cap = zs_b.tree.cap();
  proof.quotient_cap = quot_b.tree.cap();

  // ---- PolynomialBatch::prove_openings ----
  Ext alpha = ch.get_extension_challenge();
  std::vector<Ext> final_poly(n, Ext());
  for (int batch = 0; batch < 2; ++batch) {
    // batch 0: all polys of oracles 0..3 at zeta; batch 1: the Z polys (oracle 2, first nch) at g*zeta
    std::vector<const std::vector<u64>*> polys;
    // circuit_data.rs::get_fri_instance: fri_all_polys = [preprocessed, wires, zs + partial products, quotient, lookup polys];
    // fri_next_batch_polys = [zs, lookup polys] — the lookup polynomials are the TAIL of the zs_partial_products oracle
    const size_t zs_pp = (size_t)c.num_zs_pp();
    if (batch == 0) {
      for (size_t k = 0; k < oracles[0]->ncols; ++k) polys.push_back(&oracles[0]->coeffs[k]);
      for (size_t k = 0; k < oracles[1]->ncols; ++k) polys.push_back(&oracles[1]->coeffs[k]);
      for (size_t k = 0; k < zs_pp; ++k) polys.push_back(&zs_b.coeffs[k]);
      for (size_t k = 0; k < oracles[3]->ncols; ++k) polys.push_back(&oracles[3]->coeffs[k]);
      for (size_t k = zs_pp; k < zs_b.ncols; ++k) polys.push_back(&zs_b.coeffs[k]);
    } else {
      for (int k = 0; k < nch; ++k) polys.push_back(&zs_b.coeffs[k]);
      for (size_t k = zs_pp; k < zs_b.ncols; ++k) polys.push_back(&zs_b.coeffs[k]);
    }
    Ext point = batch == 0 ? zeta : gzeta;
    // alpha.reduce_polys_base: sum_j alpha^j f_j
    std::vector<Ext> comp(n, Ext());
    {
      std::vector<Ext> apow(polys.size());
      Ext a(1);
      for (size_t j = 0; j < polys.size(); ++j) apow[j] = a, a = a * alpha;
      long long nn = (long long)n;
#pragma omp parallel for schedule(static)
      for (long long i = 0; i < nn; ++i) {
        Ext acc;
        for (size_t j = 0; j < polys.size(); ++j) acc = acc + scale(apow[j], (*polys[j])[i]);
        comp[i] = acc;
      }
    }
    // divide_by_linear(point): quotient of (comp - comp(point)) / (X - point), then pad with a zero
    std::vector<Ext> quo(n, Ext());
    {
      Ext acc;
      for (size_t i = n; i-- > 0;) {
        // bs[i] = comp[i] + point * bs[i+1]; quotient coefficient q[i-1] = bs[i]
        acc = acc * point + comp[i];
        if (i > 0) quo[i - 1] = acc;
      }
    }
    // alpha.shift_poly(final_poly): *= alpha^count, count = |this batch|
    Ext sh = ext_pow(alpha, polys.size());
    for (size_t i = 0; i < n; ++i) final_poly[i] = final_poly[i] * sh + quo[i];
  }
  // lde + coset_fft
  std::vector<Ext> coeffs(N, Ext());
  for (size_t i = 0; i < n; ++i) coeffs[i] = final_poly[i];
  std::vector<Ext> values = coeffs;
  coset_fft_ext_inplace(values, LG, MULTIPLICATIVE_GENERATOR);

  // ---- fri_proof: commit phase ----
  std::vector<MerkleTree> trees;
  {
    u64 shift = MULTIPLICATIVE_GENERATOR;
    for (int arity_bits : c.reduction_arity_bits) {
      size_t arity = (size_t)1 << arity_bits;
      reverse_index_bits_in_place(values);
      std::vector<u64> leaves(values.size() * 2);
      for (size_t i = 0; i < values.size(); ++i) leaves[2 * i] = values[i].a, leaves[2 * i + 1] = values[i].b;
      MerkleTree t;
      t.build(std::move(leaves), 2 * arity, c.cap_height);
      ch.observe_cap(t.cap());
      proof.fri.commit_phase_caps.push_back(t.cap());
      trees.push_back(std::move(t));
      Ext beta = ch.get_extension_challenge();
      std::vector<Ext> nc(coeffs.size() / arity);
      for (size_t k = 0; k < nc.size(); ++k) {
        Ext acc;
        for (size_t t2 = arity; t2-- > 0;) acc = acc * beta + coeffs[k * arity + t2];
        nc[k] = acc;
      }
      coeffs = std::move(nc);
      shift = pow(shift, arity);
      values = coeffs;
      coset_fft_ext_inplace(values, log2_strict(values.size()), shift);
    }
    coeffs.resize(coeffs.size() >> rb);
    for (Ext e : coeffs) ch.observe_ext(e);
    proof.fri.final_poly = coeffs;
  }
  // ---- proof of work (fri/prover.rs::fri_proof_of_work; smallest witness => deterministic) ----
  {
    // upstream's shortcut: pre-load the pending inputs into a copy of the sponge, put the candidate in the
    // next slot, permute, and look at the LAST squeezed element (what get_challenge() would pop).
    State base = ch.sponge;
    for (size_t i = 0; i < ch.input.size(); ++i) base[i] = ch.input[i];
    const size_t pos = ch.input.size();
    auto ok = [&](u64 cand) {
      State s = base;
      s[pos] = cand;
      permute(s);
      u64 resp = s[SPONGE_RATE - 1];
      return c.pow_bits == 0 || (resp >> (64 - c.pow_bits)) == 0;
    };
    u64 wts = 0;
    if (opt.has_pow_hint) {
      wts = opt.pow_hint;
      if (!ok(wts)) throw std::runtime_error("pow_witness hint does not satisfy the proof-of-work condition");
    } else {
      // smallest valid witness, searched in parallel blocks (upstream: rayon find_any => any valid witness)
      const u64 block = (u64)1 << 14;
      for (u64 base = 0;; base += block) {
        u64 best = ~(u64)0;
#pragma omp parallel for schedule(static) reduction(min : best)
        for (long long k = 0; k < (long long)block; ++k) {
          u64 cand = base + (u64)k;
          if (cand < best && ok(cand)) best = cand;
        }
        if (best != ~(u64)0) {
          wts = best;
          break;
        }
      }
    }
    proof.fri.pow_witness = wts;
    ch.observe_element(wts);
    (void)ch.get_challenge();  // pow_response
  }
  // ---- query rounds ----
  for (int q = 0; q < c.num_query_rounds; ++q) {
    size_t x_index = (size_t)(ch.get_challenge() % (u64)N);
    FriQueryRound qr;
    for (int o = 0; o < 4; ++o) {
      const MerkleTree& t = oracles[o]->tree;
      qr.initial.evals.emplace_back(t.leaf(x_index), t.leaf(x_index) + t.width);
      qr.initial.proofs.push_back(t.prove(x_index));
    }
    size_t xi = x_index;
    for (size_t r = 0; r < trees.size(); ++r) {
      int ab = c.reduction_arity_bits[r];
      size_t coset = xi >> ab;
      FriQueryStep st;
      const u64* lf = trees[r].leaf(coset);
      for (size_t k = 0; k < ((size_t)1 << ab); ++k) st.evals.push_back(Ext(lf[2 * k], lf[2 * k + 1]));
      st.proof = trees[r].prove(coset);
      qr.steps.push_back(std::move(st));
      xi = coset;
    }
    proof.fri.query_rounds.push_back(std::move(qr));
  }
  if (tm) tm->fri = now_s() - t0, tm->total = now_s() - t_start;
  return proof;
}

// ---------------------------------------------------------------------------------------------
// Verifier (plonk/verifier.rs::verify_with_challenges + fri/verifier.rs::verify_fri_proof)
// Needs only: circuit parameters, gates, k_is, constants_sigmas CAP and circuit_digest.
// ---------------------------------------------------------------------------------------------
static Ext interpolate_at(const std::vector<Ext>& xs, const std::vector<Ext>& ys, Ext x) {
  Ext r;
  for (size_t i = 0; i < xs.size(); ++i) {
    Ext num(1), den(1);
    for (size_t j = 0; j < xs.size(); ++j)
      if (j != i) {
        num = num * (x - xs[j]);
        den = den * (xs[i] - xs[j]);
      }
    r = r + ys[i] * num * ext_inv(den);
  }
  return r;
}

static std::string verify(const Circuit& c, const Proof& p) {
  const size_t n = c.n();
  const int lg = c.degree_bits, rb = c.rate_bits, LG = lg + rb;
  const size_t N = (size_t)1 << LG;
  const int nch = c.num_challenges, npp = c.num_partial_products(), qdf = c.quotient_degree_factor;
  const OpeningSet& o = p.openings;
  if ((int)o.constants.size() != c.num_constants || (int)o.plonk_sigmas.size() != c.num_routed_wires ||
      (int)o.wires.size() != c.num_wires || (int)o.plonk_zs.size() != nch || (int)o.plonk_zs_next.size() != nch ||
      (int)o.partial_products.size() != nch * npp || (int)o.quotient_polys.size() != nch * qdf ||
      (int)o.lookup_zs.size() != nch * c.num_lookup_polys() || o.lookup_zs_next.size() != o.lookup_zs.size())
    return "opening set has the wrong shape";
  if (p.public_inputs.size() != c.public_inputs.size()) return "wrong number of public inputs";
  Hash pih = hash_no_pad(p.public_inputs.data(), p.public_inputs.size());
  // ---- challenges (proof.rs::get_challenges) ----
  Challenger ch;
  ch.observe_hash(c.circuit_digest);
  ch.observe_hash(pih);
  ch.observe_cap(p.wires_cap);
  std::vector<u64> betas(nch), gammas(nch), alphas(nch);
  for (int i = 0; i < nch; ++i) betas[i] = ch.get_challenge();
  for (int i = 0; i < nch; ++i) gammas[i] = ch.get_challenge();
  std::vector<u64> deltas;
  if (c.has_lookup()) {
    deltas = betas;
    deltas.insert(deltas.end(), gammas.begin(), gammas.end());
    for (int i = 0; i < 2 * nch; ++i) deltas.push_back(ch.get_challenge());
  }
  ch.observe_cap(p.zs_pp_cap);
  for (int i = 0; i < nch; ++i) alphas[i] = ch.get_challenge();
  ch.observe_cap(p.quotient_cap);
  Ext zeta = ch.get_extension_challenge();
  for (auto* v : {&o.constants, &o.plonk_sigmas, &o.wires, &o.plonk_zs, &o.partial_products, &o.quotient_polys, &o.lookup_zs})
    for (Ext e : *v) ch.observe_ext(e);
  for (auto* v : {&o.plonk_zs_next, &o.lookup_zs_next})
    for (Ext e : *v) ch.observe_ext(e);
  Ext fri_alpha = ch.get_extension_challenge();
  if (p.fri.commit_phase_caps.size() != c.reduction_arity_bits.size()) return "wrong number of FRI commit-phase caps";
  std::vector<Ext> fri_betas;
  for (const auto& cap : p.fri.commit_phase_caps) {
    ch.observe_cap(cap);
    fri_betas.push_back(ch.get_extension_challenge());
  }
  for (Ext e : p.fri.final_poly) ch.observe_ext(e);
  ch.observe_element(p.fri.pow_witness);
  u64 pow_response = ch.get_challenge();
  std::vector<size_t> x_indices;
  for (int q = 0; q < c.num_query_rounds; ++q) x_indices.push_back((size_t)(ch.get_challenge() % (u64)N));

  // ---- vanishing polynomial identity at zeta ----
  {
    Ext zeta_pow_n = zeta;
    for (int i = 0; i < lg; ++i) zeta_pow_n = zeta_pow_n * zeta_pow_n;
    Ext z_h = zeta_pow_n - Ext(1);
    // eval_l_0(n, x) = (x^n - 1) / (n (x - 1))
    Ext l0 = z_h * ext_inv(scale(zeta - Ext(1), (u64)n % P));
    std::vector<Ext> al(nch), res(nch);
    for (int i = 0; i < nch; ++i) al[i] = Ext(alphas[i]);
    eval_vanishing<Ext>(c, zeta, l0, o.constants.data(), o.plonk_sigmas.data(), o.wires.data(), o.plonk_zs.data(),
                        o.plonk_zs_next.data(), o.partial_products.data(), pih, betas.data(), gammas.data(), al.data(),
                        res.data(), o.lookup_zs.data(), o.lookup_zs_next.data(), deltas.data());
    for (int i = 0; i < nch; ++i) {
      Ext acc;  // reduce_with_powers(chunk, zeta^n)
      for (int k = qdf; k-- > 0;) acc = acc * zeta_pow_n + o.quotient_polys[i * qdf + k];
      if (res[i] != z_h * acc) return "vanishing polynomial identity fails at zeta (challenge " + std::to_string(i) + ")";
    }
  }
  // ---- FRI ----
  if (c.pow_bits > 0 && (pow_response >> (64 - c.pow_bits)) != 0) return "proof of work check failed";
  if ((int)p.fri.query_rounds.size() != c.num_query_rounds) return "wrong number of FRI query rounds";
  size_t final_len = n;
  for (int ab : c.reduction_arity_bits) final_len >>= ab;
  if (p.fri.final_poly.size() != final_len) return "final polynomial has the wrong length";
  // PrecomputedReducedOpenings::from_os_and_alpha
  Ext g = Ext(root_of_unity(lg));
  Ext points[2] = {zeta, zeta * g};
  std::vector<Ext> batch_vals[2];
  for (auto* v : {&o.constants, &o.plonk_sigmas, &o.wires, &o.plonk_zs, &o.partial_products, &o.quotient_polys, &o.lookup_zs})
    batch_vals[0].insert(batch_vals[0].end(), v->begin(), v->end());
  batch_vals[1] = o.plonk_zs_next;
  batch_vals[1].insert(batch_vals[1].end(), o.lookup_zs_next.begin(), o.lookup_zs_next.end());
  Ext reduced_openings[2];
  for (int b = 0; b < 2; ++b) {
    Ext acc;
    for (size_t i = batch_vals[b].size(); i-- > 0;) acc = acc * fri_alpha + batch_vals[b][i];
    reduced_openings[b] = acc;
  }
  const std::vector<Hash>* caps[4] = {&c.constants_sigmas.tree.cap(), &p.wires_cap, &p.zs_pp_cap, &p.quotient_cap};
  const size_t widths[4] = {(size_t)c.num_preprocessed(), (size_t)c.num_wires, (size_t)c.num_zs_pp_lookup(), (size_t)c.num_quotient()};
  const size_t zs_pp = (size_t)c.num_zs_pp();
  u64 wN = root_of_unity(LG);
  for (int q = 0; q < c.num_query_rounds; ++q) {
    const FriQueryRound& qr = p.fri.query_rounds[q];
    size_t x_index = x_indices[q];
    if (qr.initial.evals.size() != 4 || qr.steps.size() != c.reduction_arity_bits.size()) return "malformed query round";
    for (int t = 0; t < 4; ++t) {
      if (qr.initial.evals[t].size() != widths[t]) return "initial leaf has the wrong width";
      if (!verify_merkle_proof_to_cap(qr.initial.evals[t].data(), widths[t], x_index, *caps[t], qr.initial.proofs[t]))
        return "initial Merkle proof fails (oracle " + std::to_string(t) + ")";
    }
    u64 subgroup_x = mul(MULTIPLICATIVE_GENERATOR, pow(wN, reverse_bits(x_index, LG)));
    // fri_combine_initial
    Ext sum;
    for (int b = 0; b < 2; ++b) {
      std::vector<u64> ev;
      const std::vector<u64>& zl = qr.initial.evals[2];  // [zs, partial products, lookup polys]
      if (b == 0) {
        ev.insert(ev.end(), qr.initial.evals[0].begin(), qr.initial.evals[0].end());
        ev.insert(ev.end(), qr.initial.evals[1].begin(), qr.initial.evals[1].end());
        ev.insert(ev.end(), zl.begin(), zl.begin() + zs_pp);
        ev.insert(ev.end(), qr.initial.evals[3].begin(), qr.initial.evals[3].end());
        ev.insert(ev.end(), zl.begin() + zs_pp, zl.end());
      } else {
        ev.assign(zl.begin(), zl.begin() + nch);
        ev.insert(ev.end(), zl.begin() + zs_pp, zl.end());
      }
      Ext red;
      for (size_t i = ev.size(); i-- > 0;) red = red * fri_alpha + Ext(ev[i]);
      Ext numer = red - reduced_openings[b];
      Ext denom = Ext(subgroup_x) - points[b];
      sum = sum * ext_pow(fri_alpha, ev.size());
      sum = sum + numer * ext_inv(denom);
    }
    Ext old_eval = sum;
    size_t xi = x_index;
    u64 sx = subgroup_x;
    for (size_t r = 0; r < c.reduction_arity_bits.size(); ++r) {
      int ab = c.reduction_arity_bits[r];
      size_t arity = (size_t)1 << ab;
      const FriQueryStep& st = qr.steps[r];
      if (st.evals.size() != arity) return "query step has the wrong number of evals";
      size_t coset = xi >> ab, within = xi & (arity - 1);
      if (st.evals[within] != old_eval) return "FRI consistency check fails at round " + std::to_string(r);
      // compute_evaluation
      {
        u64 gg = root_of_unity(ab);
        std::vector<Ext> ev = st.evals;
        reverse_index_bits_in_place(ev);
        size_t rev_within = reverse_bits(within, ab);
        u64 coset_start = mul(sx, pow(gg, arity - rev_within));
        std::vector<Ext> xs(arity);
        u64 pw = 1;
        for (size_t k = 0; k < arity; ++k) {
          xs[k] = Ext(mul(coset_start, pw));
          pw = mul(pw, gg);
        }
        old_eval = interpolate_at(xs, ev, fri_betas[r]);
      }
      std::vector<u64> flat(2 * arity);
      for (size_t k = 0; k < arity; ++k) flat[2 * k] = st.evals[k].a, flat[2 * k + 1] = st.evals[k].b;
      if (!verify_merkle_proof_to_cap(flat.data(), flat.size(), coset, p.fri.commit_phase_caps[r], st.proof))
        return "FRI commit-phase Merkle proof fails at round " + std::to_string(r);
      for (int k = 0; k < ab; ++k) sx = sqr(sx);
      xi = coset;
    }
    if (eval_extpoly_ext(p.fri.final_poly.data(), p.fri.final_poly.size(), Ext(sx)) != old_eval)
      return "final polynomial evaluation mismatch";
  }
  return "";
}

}  // namespace vxo

#include <chrono>
namespace vxo {
static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace vxo

// ---------------------------------------------------------------------------------------------
// util/serialization: read_proof_with_public_inputs (inverse of serialize_proof; lengths implied by
// the circuit's common data)
// ---------------------------------------------------------------------------------------------
namespace vxo {
struct ByteReader {
  const uint8_t* p;
  size_t len, pos = 0;
  bool ok = true;
  uint8_t u8() {
    if (pos + 1 > len) { ok = false; return 0; }
    return p[pos++];
  }
  u64 f() {
    if (pos + 8 > len) { ok = false; return 0; }
    u64 v = 0;
    for (int i = 0; i < 8; ++i) v |= (u64)p[pos + i] << (8 * i);
    pos += 8;
    if (v >= P) ok = false;  // serialised field elements are canonical
    return v;
  }
  Ext ext() { u64 a = f(); u64 b = f(); return Ext(a, b); }
  Hash hash() { Hash h; for (int i = 0; i < 4; ++i) h.e[i] = f(); return h; }
  std::vector<Hash> cap(int h) { std::vector<Hash> c((size_t)1 << h); for (auto& x : c) x = hash(); return c; }
  std::vector<Ext> extvec(size_t n) { std::vector<Ext> v(n); for (auto& x : v) x = ext(); return v; }
  std::vector<Hash> merkle_proof() { size_t n = u8(); std::vector<Hash> v(n); for (auto& x : v) x = hash(); return v; }
};

static bool deserialize_proof(const Circuit& c, const uint8_t* bytes, size_t len, Proof& p) {
  ByteReader r{bytes, len};
  const int nch = c.num_challenges;
  p.wires_cap = r.cap(c.cap_height);
  p.zs_pp_cap = r.cap(c.cap_height);
  p.quotient_cap = r.cap(c.cap_height);
  OpeningSet& o = p.openings;
  o.constants = r.extvec(c.num_constants);
  o.plonk_sigmas = r.extvec(c.num_routed_wires);
  o.wires = r.extvec(c.num_wires);
  o.plonk_zs = r.extvec(nch);
  o.plonk_zs_next = r.extvec(nch);
  o.lookup_zs = r.extvec((size_t)nch * c.num_lookup_polys());
  o.lookup_zs_next = r.extvec((size_t)nch * c.num_lookup_polys());
  o.partial_products = r.extvec((size_t)nch * c.num_partial_products());
  o.quotient_polys = r.extvec(c.num_quotient());
  p.fri.commit_phase_caps.clear();
  for (size_t i = 0; i < c.reduction_arity_bits.size(); ++i) p.fri.commit_phase_caps.push_back(r.cap(c.cap_height));
  const size_t widths[4] = {(size_t)c.num_preprocessed(), (size_t)c.num_wires, (size_t)c.num_zs_pp_lookup(), (size_t)c.num_quotient()};
  p.fri.query_rounds.clear();
  for (int q = 0; q < c.num_query_rounds && r.ok; ++q) {
    FriQueryRound qr;
    for (int t = 0; t < 4; ++t) {
      std::vector<u64> ev(widths[t]);
      for (auto& x : ev) x = r.f();
      qr.initial.evals.push_back(std::move(ev));
      qr.initial.proofs.push_back(r.merkle_proof());
    }
    for (int ab : c.reduction_arity_bits) {
      FriQueryStep st;
      st.evals = r.extvec((size_t)1 << ab);
      st.proof = r.merkle_proof();
      qr.steps.push_back(std::move(st));
    }
    p.fri.query_rounds.push_back(std::move(qr));
  }
  size_t final_len = c.n();
  for (int ab : c.reduction_arity_bits) final_len >>= ab;
  p.fri.final_poly = r.extvec(final_len);
  p.fri.pow_witness = r.f();
  p.public_inputs.resize(c.public_inputs.size());
  for (auto& x : p.public_inputs) x = r.f();
  return r.ok && r.pos == len;
}
}  // namespace vxo
