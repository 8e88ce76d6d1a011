// One validation of a vx_circuit_desc, shared by vx_circuit_create (prover key) and vx_verify_standalone (verifier-only
// hosts): every field a kernel or the host verifier later uses as an index, a shift or a size is range-checked HERE, so a
// malformed description is answered with VX_E_INVALID instead of an out-of-range read.  Also resolves the values that
// the caller may either pass (the Rust side holds them in plonky2's CommonCircuitData / VerifierOnlyCircuitData,
// plonk/circuit_data.rs — reached from /root/reference/circuits/header_range.rs:167 `circuit.prove(&input)`) or leave to
// the library's own derivation: the FRI reduction arities and the number of partial products.
#pragma once
#include <string>
#include <vector>
#include "../../include/vxprover.h"

#ifndef VX_MAX_GATES
#error "include plonk_kernels.hip.h (limits) before desc_check.h"
#endif

struct DescResolved {
  std::vector<int> arity_bits;  // FriParams::reduction_arity_bits
  int num_partial_products = 0;
  bool digest_given = false;
};

// fri/reduction_strategies.rs  FriReductionStrategy::ConstantArityBits(4, 5)  (standard_recursion_config): THE one
// place this recalled rule lives; a caller that passes VX_DESC_HAS_FRI_ARITIES bypasses it.
static inline std::vector<int> fri_constant_arity_bits_4_5(int degree_bits, int rate_bits, int cap_height) {
  std::vector<int> a;
  int db = degree_bits;
  while (db > 5 && db + rate_bits - 4 >= cap_height) {
    a.push_back(4);
    db -= 4;
  }
  return a;
}

// Returns an empty string when the description is usable, else the reason.
static inline std::string desc_check(const vx_circuit_desc* d, bool need_preprocessed, DescResolved* out) {
  auto bad = [](const char* what, long long v) { return std::string("circuit: ") + what + " (" + std::to_string(v) + ")"; };
  if (d->degree_bits < 1 || d->rate_bits < 1 || d->degree_bits + d->rate_bits > ROOT_TABLE_LOG) return bad("degree_bits / rate_bits unsupported", d->degree_bits);
  if ((1 << d->rate_bits) > VX_MAX_RATE) return bad("rate_bits unsupported", d->rate_bits);
  if (d->num_challenges < 1 || d->num_challenges > VX_MAX_CHALLENGES) return bad("num_challenges unsupported", d->num_challenges);
  if (d->num_gates < 1 || d->num_gates > VX_MAX_GATES) return bad("num_gates unsupported", d->num_gates);
  if (!d->gate_types || !d->gate_params || !d->selector_indices || !d->group_starts || !d->group_ends) return "circuit: NULL gate arrays";
  // CircuitConfig::max_quotient_degree_factor (8 in standard_recursion_config); prover.rs asserts log2_ceil(qdf) <= rate_bits
  if (d->quotient_degree_factor < 1 || d->quotient_degree_factor > (1 << d->rate_bits))
    return bad("quotient_degree_factor outside [1, 2^rate_bits]", d->quotient_degree_factor);
  if (d->num_wires < 1 || d->num_wires > 4096 || d->num_routed_wires < 1 || d->num_routed_wires > d->num_wires) return bad("bad wire counts", d->num_wires);
  const int chunks = (d->num_routed_wires + d->quotient_degree_factor - 1) / d->quotient_degree_factor;
  if (chunks > PERM_MAX_CHUNKS) return bad("too many partial-product chunks", chunks);
  if (d->cap_height < 0 || d->cap_height > d->degree_bits + d->rate_bits) return bad("cap_height outside [0, degree_bits + rate_bits]", d->cap_height);
  if (d->num_query_rounds < 1 || d->num_query_rounds > 4096) return bad("num_query_rounds outside [1, 4096]", d->num_query_rounds);
  if (d->pow_bits < 0 || d->pow_bits > 40) return bad("pow_bits unsupported", d->pow_bits);
  if (d->num_selectors < 1 || d->num_selectors > d->num_constants || d->num_constants > 4096) return bad("need 1 <= num_selectors <= num_constants", d->num_selectors);
  if (!d->k_is) return "circuit: NULL k_is";
  if (need_preprocessed && !d->constants_sigmas) return "circuit: NULL preprocessed data";
  if (d->num_public_inputs < 0 || d->num_public_inputs > (1 << 20) || (d->num_public_inputs && (!d->pi_rows || !d->pi_cols))) return bad("bad public input list", d->num_public_inputs);
  for (int i = 0; i < d->num_public_inputs; ++i)
    if ((uint64_t)d->pi_rows[i] >= ((uint64_t)1 << d->degree_bits) || (int64_t)d->pi_cols[i] >= d->num_wires) return bad("public input target out of range", i);
  if (d->programs_len < 0 || (d->programs_len && !d->programs)) return bad("bad programs_len", d->programs_len);
  // lookup argument: sizes first, the gate constants start after BOTH kinds of selectors
  if (d->num_luts < 0 || d->num_luts > VX_MAX_LUTS) return bad("num_luts unsupported (the prover's lookup path holds at most VX_MAX_LUTS tables)", d->num_luts);
  const int nls = d->num_luts > 0 ? d->num_lookup_selectors : 0;
  if (d->num_luts > 0) {
    if (d->num_lookup_selectors != 4 + d->num_luts) return bad("num_lookup_selectors must be 4 + num_luts (TransSre, TransLdc, InitSre, LastLdc + one per table)", d->num_lookup_selectors);
    if (!d->lut_lens || !d->lut_inputs || !d->lut_outputs || !d->lookup_rows) return "circuit: NULL lookup table data";
    if (d->num_selectors + nls > d->num_constants) return bad("lookup selectors exceed num_constants", nls);
    if (d->num_routed_wires < 6 || d->quotient_degree_factor < 2) return "circuit: lookups need >= 6 routed wires and quotient_degree_factor >= 2";
    {
      // slot groups of the partial Sum / LDC polynomials (lookup_degree = quotient_degree_factor - 1 looking slots, the table
      // slots spread evenly over the same number of polynomials) must fit the kernel's group buffers
      const int lu_deg = d->quotient_degree_factor - 1, nsl = (d->num_routed_wires / 2 + lu_deg - 1) / lu_deg;
      const int lut_deg = (d->num_routed_wires / 3 + nsl - 1) / nsl;
      if (lu_deg > VX_LOOKUP_GROUP_MAX || lut_deg > VX_LOOKUP_GROUP_MAX) return bad("lookup slot group too large", lu_deg > lut_deg ? lu_deg : lut_deg);
      if ((long long)d->num_challenges * (4 + d->num_luts + 2 * nsl) + 160 > VX_ALPHA_POWS) return bad("too many lookup constraint terms", nsl);
    }
    const long long nrows = (long long)1 << d->degree_bits;
    for (int t = 0; t < d->num_luts; ++t) {
      if (d->lut_lens[t] < 1 || d->lut_lens[t] > (1 << 20)) return bad("bad lookup table length", d->lut_lens[t]);
      const long long lu = d->lookup_rows[3 * t], lut = d->lookup_rows[3 * t + 1], first = d->lookup_rows[3 * t + 2];
      if (lu < 0 || lu > lut || lut > first || first + 1 >= nrows) return bad("lookup rows out of order / out of range (need last_lu <= last_lut <= first_lut < n - 1)", t);
      if ((first - lut + 1) * (long long)(d->num_routed_wires / 3) < d->lut_lens[t]) return bad("lookup table does not fit its LookupTableGate rows", t);
    }
  } else if (d->num_lookup_selectors != 0) return bad("lookup selectors without lookup tables", d->num_lookup_selectors);
  const int gate_consts = d->num_constants - d->num_selectors - nls;  // constants a gate may read: local_constants[num_selectors + num_lookup_selectors + q]
  int nprog = 0;
  for (int g = 0; g < d->num_gates; ++g) {
    const int t = d->gate_types[g], prm = d->gate_params[g];
    if (t < VX_GATE_NOOP || t > VX_GATE_LOOKUP_TABLE) return bad("gate type is not in the supported set", t);
    if ((t == VX_GATE_LOOKUP || t == VX_GATE_LOOKUP_TABLE) && d->num_luts < 1) return bad("lookup gate without a lookup table", g);
    if (t == VX_GATE_LOOKUP && prm != d->num_routed_wires / 2) return bad("LookupGate num_slots must be num_routed_wires / 2", prm);
    if (t == VX_GATE_LOOKUP_TABLE && prm != d->num_routed_wires / 3) return bad("LookupTableGate num_slots must be num_routed_wires / 3", prm);
    if (d->selector_indices[g] < 0 || d->selector_indices[g] >= d->num_selectors) return bad("bad selector index of gate", g);
    if (d->group_starts[g] < 0 || d->group_starts[g] > g || d->group_ends[g] <= g || d->group_ends[g] > d->num_gates) return bad("gate outside its selector group [start, end)", g);
    if (t == VX_GATE_CONSTANT && (prm < 0 || prm > gate_consts || prm > d->num_wires)) return bad("ConstantGate num_consts exceeds the constants / wires", prm);
    if (t == VX_GATE_ARITHMETIC && (prm < 1 || 4 * (long long)prm > d->num_wires || gate_consts < 2)) return bad("ArithmeticGate ops exceed the wires, or fewer than 2 gate constants", prm);
    if (t == VX_GATE_POSEIDON && d->num_wires < 135) return bad("PoseidonGate needs 135 wires", d->num_wires);
    {
      // filtered constraint degree = Gate::degree() + (gates sharing the selector - 1) + (1 for the UNUSED factor when there
      // are several selectors) must be <= quotient_degree_factor + 1 (gates/selectors.rs), or the quotient does not fit
      // its quotient_degree_factor chunks
      const int deg = t == VX_GATE_POSEIDON ? 7 : t == VX_GATE_ARITHMETIC ? 3 : (t == VX_GATE_CONSTANT || t == VX_GATE_PUBLIC_INPUT) ? 1 : t == VX_GATE_PROGRAM ? prm : 0;
      const int filtered = deg + (d->group_ends[g] - d->group_starts[g] - 1) + (d->num_selectors > 1 ? 1 : 0);
      if (filtered > d->quotient_degree_factor + 1) return bad("filtered constraint degree of a gate exceeds quotient_degree_factor + 1", g);
    }
    if (t == VX_GATE_PUBLIC_INPUT && d->num_wires < 4) return bad("PublicInputGate needs 4 wires", d->num_wires);
    if (t == VX_GATE_PROGRAM) {
      ++nprog;
      if (prm < 1 || prm > d->quotient_degree_factor + 1) return bad("program gate degree outside [1, quotient_degree_factor + 1]", prm);
      if (!d->programs || !d->program_offsets || d->program_offsets[g] < 0 || d->program_offsets[g] >= d->programs_len) return bad("program gate has no program", g);
      // terminated, known opcodes, operands in range, no register read before it is written
      bool ended = false;
      uint64_t defined = 0;
      for (int pc = d->program_offsets[g]; pc < d->programs_len && !ended; ++pc) {
        const uint64_t ins = d->programs[pc];
        const int op = (int)(ins & 0xFF), dst = (int)((ins >> 8) & 0xFF), a = (int)((ins >> 16) & 0xFFFF), b = (int)((ins >> 32) & 0xFFFF);
        auto is_def = [&](int r) { return r < VX_PROGRAM_REGS && ((defined >> r) & 1); };
        if (op == VX_OP_END) { ended = true; continue; }
        if (op < VX_OP_END || op > VX_OP_LDP) return bad("bad opcode in a constraint program", op);
        if (op != VX_OP_PUSH && dst >= VX_PROGRAM_REGS) return bad("constraint program writes a register out of range", dst);
        if (op == VX_OP_LDI) { if (++pc >= d->programs_len) return "circuit: truncated constraint program"; }
        else if (op == VX_OP_LDW) { if (a >= d->num_wires) return bad("constraint program reads a wire out of range", a); }
        else if (op == VX_OP_LDC) { if (a >= gate_consts) return bad("constraint program reads a constant out of range", a); }
        else if (op == VX_OP_LDP) { if (a >= 4) return bad("constraint program reads public_inputs_hash out of range", a); }
        else if (op == VX_OP_ADD || op == VX_OP_SUB || op == VX_OP_MUL) { if (!is_def(a) || !is_def(b)) return "circuit: constraint program reads a register before writing it"; }
        else if (op == VX_OP_PUSH) { if (!is_def(a)) return "circuit: constraint program pushes a register before writing it"; }
        if (op != VX_OP_PUSH) defined |= (uint64_t)1 << dst;
      }
      if (!ended) return "circuit: unterminated constraint program";
    }
  }
  if (nprog > VX_MAX_PROGRAM_GATES) return bad("too many program gates", nprog);

  // ---- values the caller may pass instead of having them re-derived (vxprover.h VX_DESC_HAS_*) ----
  const uint32_t known = VX_DESC_HAS_CIRCUIT_DIGEST | VX_DESC_HAS_FRI_ARITIES | VX_DESC_HAS_NUM_PARTIAL_PRODUCTS;
  if (d->override_flags & ~known) return bad("unknown override_flags bits", d->override_flags);
  if (d->hiding) return "circuit: zero-knowledge (FriParams::hiding / salted leaves, blinding rows) is not supported; VectorX uses standard_recursion_config with zero_knowledge = false";
  DescResolved r;
  r.digest_given = (d->override_flags & VX_DESC_HAS_CIRCUIT_DIGEST) != 0;
  if (d->override_flags & VX_DESC_HAS_FRI_ARITIES) {
    if (d->num_fri_reduction_arity_bits < 0 || d->num_fri_reduction_arity_bits > 32 || (d->num_fri_reduction_arity_bits && !d->fri_reduction_arity_bits))
      return bad("bad fri_reduction_arity_bits list", d->num_fri_reduction_arity_bits);
    int total = 0;
    for (int i = 0; i < d->num_fri_reduction_arity_bits; ++i) {
      const int a = d->fri_reduction_arity_bits[i];
      if (a < 1 || a > 4) return bad("FRI reduction arity bits outside [1, 4]", a);
      total += a;
    }
    if (total > d->degree_bits) return bad("FRI reduction arities fold below degree 1", total);
    r.arity_bits.assign(d->fri_reduction_arity_bits, d->fri_reduction_arity_bits + d->num_fri_reduction_arity_bits);
  } else {
    r.arity_bits = fri_constant_arity_bits_4_5(d->degree_bits, d->rate_bits, d->cap_height);
  }
  {
    // Merkle caps of the FRI layers: layer r has 2^(degree_bits + rate_bits - sum of the arities so far) values in
    // leaves of 2^arity, and its tree needs at least 2^cap_height leaves
    int lg = d->degree_bits + d->rate_bits;
    for (int a : r.arity_bits) {
      if (lg - a < d->cap_height) return bad("a FRI layer has fewer leaves than the Merkle cap", lg - a);
      lg -= a;
    }
  }
  r.num_partial_products = chunks - 1;  // plonk_common.rs num_partial_products: ceil(num_routed / quotient_degree_factor) - 1
  if ((d->override_flags & VX_DESC_HAS_NUM_PARTIAL_PRODUCTS) && d->num_partial_products != r.num_partial_products)
    return bad("num_partial_products disagrees with ceil(num_routed_wires / quotient_degree_factor) - 1: the partial-product layout of this "
               "library would not match the caller's CommonCircuitData", d->num_partial_products);
  if (out) *out = r;
  return std::string();
}
