// STARK prover / verifier on the SAME primitives as the plonk prover — the scoping spike for SURVEY.md §8 f-3.
//
// Every real VectorX map / outer proof embeds Curta (starkyx v1.0.0, /root/reference/Cargo.lock:7232-7234) STARKs:
// BLAKE2b (/root/reference/circuits/builder/header.rs:18), SHA-256 (circuits/builder/justification.rs:140-156) and
// Ed25519 (justification.rs:237).  A STARK proof is: commit the trace columns (PolynomialBatch::from_values), draw the
// constraint challenges, evaluate the AIR constraints over TWO ADJACENT ROWS on a coset, divide by Z_H, commit the
// quotient chunks, open everything at zeta (and the trace at g*zeta), FRI.  Everything but the constraint evaluator is
// what vx_prove already runs; this file adds
//   * the AIR evaluator interface: a constraint PROGRAM (vxprover.h VX_OP_*, plus VX_OP_LDN = next-row value and the
//     constraint kind in PUSH's b field) interpreted per row by `air_quotient_kernel`;
//   * the transcript of plonky2's `starky` prover (starky/src/prover.rs::prove_with_commitment, v0.2.0 — restated from
//     memory like everything upstream; Curta's own prover has the same shape with its own challenge schedule);
//   * the matching host verifier (starky/src/verifier.rs).
// Scope of the spike: no permutation / lookup (CTL) arguments, one process / one GPU; AIR programs are compiled with
// jit.hip.h exactly like gate programs (the interpreter below is the fallback and the cross-check); byte format = this library's own framing of
// StarkProofWithPublicInputs (starky v0.2.0 has no to_bytes): trace_cap | quotient_cap | local | next | quotient
// openings | FriProof (write_fri_proof) | public inputs.
#pragma once
#include <map>
#include <mutex>
#include "prover.hip.h"
#include "verifier.h"

struct AirParams {  // mirrored textually in jit.hip.h::jit_air_source
  const u64* trace;  // trace LDE, column stride = stride (rows of the whole LDE), rows in Merkle-leaf (bit-reversed) order
  const u64* aux;    // second-round columns' LDE (same stride) or null; program columns >= ncols address it
  const u64* program;
  size_t stride;
  size_t rows;  // n << qbits: the first 2^qbits cosets of the LDE = the size-(n * 2^qbits) coset 7 * H' (sharded: this rank's part of them)
  size_t row_base;  // global LDE row of local row 0 (a STARK sharded by coset: the rank's first quotient block * n); zh / zh_inv are indexed by LOCAL block
  int log_n, rate_bits, qbits, ncols, nch, npi;
  const u64 *root_lo, *root_hi;
  u64 alphas[VX_MAX_CHALLENGES];
  u64 pi[VX_AIR_MAX_PI];
  u64 chal[VX_AIR_MAX_CHALLENGES];  // the aux challenges (VX_OP_LDCH)
  u64 zh[VX_MAX_RATE], zh_inv[VX_MAX_RATE];  // Z_H on coset block z, and its inverse
  u64 last, n_inv;                            // g^-1 (the last element of H), 1/n
  u64* out;                                   // [nch][rows]
  u64* rowfac;                                // [3][rows]: z_last, L_first, L_last per row (compiled chunks read them)
  u64 tail_pow[VX_MAX_CHALLENGES];            // compiled chunks: alpha^(constraints after the chunk), see jit.hip.h
};
// z_last = x - g^-1, L_first, L_last on every row of the quotient domain: one field inversion per row, computed ONCE for all
// the chunk kernels of a compiled AIR program (the interpreter below computes them inline).
__global__ __launch_bounds__(256) void air_row_factors_kernel(AirParams p) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.rows) return;
  const int LG = p.log_n + p.rate_bits;
  const u32 z = (u32)(i >> p.log_n);
  const u32 j = bitrev32((u32)(i + p.row_base), LG);
  const u64 x = gl_mul7(root_pow24(p.root_lo, p.root_hi, j << (ROOT_TABLE_LOG - LG)));
  const u64 z_last = gl_sub(x, p.last), xm1 = gl_sub(x, 1);
  const u64 inv_both = gl_inv(gl_mul(z_last, xm1));
  const u64 zh_n = gl_mul(p.zh[z], p.n_inv);
  p.rowfac[i] = z_last;
  p.rowfac[p.rows + i] = gl_mul(zh_n, gl_mul(inv_both, z_last));
  p.rowfac[2 * p.rows + i] = gl_mul(gl_mul(zh_n, p.last), gl_mul(inv_both, xm1));
}
// One thread per row of the quotient domain.  Registers of the program live in per-thread private memory (interpreter).
__global__ __launch_bounds__(256) void air_quotient_kernel(AirParams p) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.rows) return;
  const int LG = p.log_n + p.rate_bits;
  const size_t n = (size_t)1 << p.log_n;
  const u32 z = (u32)(i >> p.log_n);                       // coset block (local)
  const u32 r = (u32)(i & (n - 1));                        // position inside the block (bit-reversed H index)
  const u32 j = bitrev32((u32)(i + p.row_base), LG);       // natural index of the LDE point
  const u64 x = gl_mul7(root_pow24(p.root_lo, p.root_hi, j << (ROOT_TABLE_LOG - LG)));
  const u32 rn = bitrev32((bitrev32(r, p.log_n) + 1) & (u32)(n - 1), p.log_n);   // row of g * x in the same block
  const size_t i_next = ((size_t)z << p.log_n) | rn;
  // z_last = x - g^-1;  L_first = Z_H / (n (x - 1));  L_last = Z_H * last / (n (x - last))
  const u64 z_last = gl_sub(x, p.last), xm1 = gl_sub(x, 1);
  const u64 inv_both = gl_inv(gl_mul(z_last, xm1));
  const u64 zh_n = gl_mul(p.zh[z], p.n_inv);
  const u64 l_first = gl_mul(zh_n, gl_mul(inv_both, z_last));
  const u64 l_last = gl_mul(gl_mul(zh_n, p.last), gl_mul(inv_both, xm1));
  u64 acc[VX_MAX_CHALLENGES] = {0, 0};
  u64 R[VX_PROGRAM_REGS];
  const u64* __restrict__ prog = p.program;
  for (int pc = 0;; ++pc) {
    const u64 ins = prog[pc];
    const int op = (int)(ins & 0xFF), dst = (int)((ins >> 8) & 63), a = (int)((ins >> 16) & 0xFFFF), b = (int)((ins >> 32) & 0xFFFF);
    if (op == VX_OP_END) break;
    switch (op) {
      case VX_OP_LDW: R[dst] = gl_canon(a < p.ncols ? p.trace[(size_t)a * p.stride + i] : p.aux[(size_t)(a - p.ncols) * p.stride + i]); break;
      case VX_OP_LDN: R[dst] = gl_canon(a < p.ncols ? p.trace[(size_t)a * p.stride + i_next] : p.aux[(size_t)(a - p.ncols) * p.stride + i_next]); break;
      case VX_OP_LDCH: R[dst] = p.chal[a]; break;
      case VX_OP_LDI: R[dst] = gl_canon(prog[++pc]); break;
      case VX_OP_LDP: R[dst] = p.pi[a]; break;
      case VX_OP_ADD: R[dst] = gl_add(R[a & 63], R[b & 63]); break;
      case VX_OP_SUB: R[dst] = gl_sub(R[a & 63], R[b & 63]); break;
      case VX_OP_MUL: R[dst] = gl_mul(R[a & 63], R[b & 63]); break;
      case VX_OP_PUSH: {
        u64 t = R[a & 63];
        if (b == VX_AIR_TRANSITION) t = gl_mul(t, z_last);
        else if (b == VX_AIR_FIRST_ROW) t = gl_mul(t, l_first);
        else if (b == VX_AIR_LAST_ROW) t = gl_mul(t, l_last);
#pragma unroll
        for (int c = 0; c < VX_MAX_CHALLENGES; ++c) acc[c] = gl_mad(acc[c], p.alphas[c], t);  // ConstraintConsumer: acc = acc * alpha + c
        break;
      }
      default: break;
    }
  }
  for (int c = 0; c < p.nch; ++c) p.out[(size_t)c * p.rows + i] = gl_mul(acc[c], p.zh_inv[z]);
}

struct StarkShape {
  int qdf = 1, qbits = 0;
  std::vector<int> arity_bits;
  int num_constraints = 0;
};
static inline int log2_ceil_int(int v) {
  int b = 0;
  while ((1 << b) < v) ++b;
  return b;
}
// Validation shared by prover and verifier.  Returns "" when usable.
static std::string stark_check(const vx_stark_desc* d, StarkShape* out) {
  auto bad = [](const char* what, long long v) { return std::string("stark: ") + what + " (" + std::to_string(v) + ")"; };
  if (d->degree_bits < 1 || d->rate_bits < 1 || d->degree_bits + d->rate_bits > ROOT_TABLE_LOG || (1 << d->rate_bits) > VX_MAX_RATE)
    return bad("degree_bits / rate_bits unsupported", d->degree_bits);
  if (d->num_columns < 1 || d->num_columns > 4096) return bad("bad column count", d->num_columns);
  if (d->num_aux_columns < 0 || d->num_aux_columns > 4096) return bad("bad aux column count", d->num_aux_columns);
  if (d->num_aux_challenges < 0 || d->num_aux_challenges > VX_AIR_MAX_CHALLENGES) return bad("too many aux challenges", d->num_aux_challenges);
  if (d->num_aux_columns == 0 && d->num_aux_challenges != 0) return bad("aux challenges without a second commitment round", d->num_aux_challenges);
  if (d->num_public_inputs < 0 || d->num_public_inputs > VX_AIR_MAX_PI) return bad("too many public inputs", d->num_public_inputs);
  if (d->num_aux_public_inputs < 0 || d->num_public_inputs + d->num_aux_public_inputs > VX_AIR_MAX_PI) return bad("too many aux public inputs", d->num_aux_public_inputs);
  if (d->num_aux_public_inputs > 0 && d->num_aux_columns == 0) return bad("aux public inputs without a second commitment round", d->num_aux_public_inputs);
  if (d->num_challenges < 1 || d->num_challenges > VX_MAX_CHALLENGES) return bad("num_challenges unsupported", d->num_challenges);
  if (d->cap_height < 0 || d->cap_height > d->degree_bits + d->rate_bits) return bad("cap_height out of range", d->cap_height);
  if (d->num_query_rounds < 1 || d->num_query_rounds > 4096) return bad("num_query_rounds out of range", d->num_query_rounds);
  if (d->pow_bits < 0 || d->pow_bits > 40) return bad("pow_bits unsupported", d->pow_bits);
  if (d->constraint_degree < 1 || d->constraint_degree > (1 << d->rate_bits) + 1) return bad("constraint degree needs a larger blow-up", d->constraint_degree);
  StarkShape s;
  s.qdf = d->constraint_degree > 1 ? d->constraint_degree - 1 : 1;  // Stark::quotient_degree_factor = max(1, degree - 1)
  s.qbits = log2_ceil_int(s.qdf);
  if (s.qbits > d->rate_bits) return bad("constraint degree higher than the rate is not supported (as in starky)", d->constraint_degree);
  if (!d->program || d->program_len < 1) return "stark: no constraint program";
  bool ended = false;
  uint64_t defined = 0;
  for (int pc = 0; pc < d->program_len && !ended; ++pc) {
    const uint64_t ins = d->program[pc];
    const int op = (int)(ins & 0xFF), dst = (int)((ins >> 8) & 0xFF), a = (int)((ins >> 16) & 0xFFFF), b = (int)((ins >> 32) & 0xFFFF);
    auto is_def = [&](int r) { return r < VX_PROGRAM_REGS && ((defined >> r) & 1); };
    if (op == VX_OP_END) { ended = true; continue; }
    if (op < VX_OP_END || op > VX_OP_LDCH || op == VX_OP_LDC) return bad("bad opcode in an AIR program", op);
    if (op != VX_OP_PUSH && dst >= VX_PROGRAM_REGS) return bad("AIR program writes a register out of range", dst);
    if (op == VX_OP_LDI) { if (++pc >= d->program_len) return "stark: truncated AIR program"; }
    else if (op == VX_OP_LDW || op == VX_OP_LDN) { if (a >= d->num_columns + d->num_aux_columns) return bad("AIR program reads a column out of range", a); }
    else if (op == VX_OP_LDCH) { if (a >= d->num_aux_challenges) return bad("AIR program reads an aux challenge out of range", a); }
    else if (op == VX_OP_LDP) { if (a >= d->num_public_inputs + d->num_aux_public_inputs) return bad("AIR program reads a public input out of range", a); }
    else if (op == VX_OP_ADD || op == VX_OP_SUB || op == VX_OP_MUL) { if (!is_def(a) || !is_def(b)) return "stark: AIR program reads a register before writing it"; }
    else if (op == VX_OP_PUSH) {
      if (!is_def(a)) return "stark: AIR program pushes a register before writing it";
      if (b > VX_AIR_LAST_ROW) return bad("unknown constraint kind", b);
      ++s.num_constraints;
    }
    if (op != VX_OP_PUSH) defined |= (uint64_t)1 << dst;
  }
  if (!ended) return "stark: unterminated AIR program";
  if (d->override_flags & ~(uint32_t)(VX_DESC_HAS_FRI_ARITIES | VX_STARK_OPENINGS_DIGEST)) return bad("unknown override_flags bits", d->override_flags);
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
    s.arity_bits.assign(d->fri_reduction_arity_bits, d->fri_reduction_arity_bits + d->num_fri_reduction_arity_bits);
  } else {
    s.arity_bits = fri_constant_arity_bits_4_5(d->degree_bits, d->rate_bits, d->cap_height);
  }
  int lgl = d->degree_bits + d->rate_bits;
  for (int a : s.arity_bits) {
    if (lgl - a < d->cap_height) return bad("a FRI layer has fewer leaves than the Merkle cap", lgl - a);
    lgl -= a;
  }
  if (out) *out = s;
  return std::string();
}

// The STATEMENT enters the transcript first (old starky observes nothing before the trace cap, so a prover who
// picks the statement could choose public inputs after seeing the alphas): shape and FRI configuration, a Poseidon digest
// of the AIR program (each 64-bit word as two 32-bit limbs — always canonical), then the canonical public inputs.  The
// byte format is this library's own, so nothing upstream constrains this prefix; the prover, vxsv::verify and the test checker
// build the same list.
static std::vector<vxh::u64> stark_statement(const vx_stark_desc* d, const StarkShape& sh, const vxh::u64* canonical_pis) {
  std::vector<vxh::u64> st = {(vxh::u64)d->degree_bits, (vxh::u64)d->rate_bits, (vxh::u64)d->cap_height, (vxh::u64)d->pow_bits, (vxh::u64)d->num_query_rounds,
                              (vxh::u64)d->num_challenges, (vxh::u64)d->constraint_degree, (vxh::u64)d->num_columns, (vxh::u64)d->num_aux_columns,
                              (vxh::u64)d->num_aux_challenges, (vxh::u64)d->num_public_inputs, (vxh::u64)d->num_aux_public_inputs,
                              (vxh::u64)sh.arity_bits.size()};
  for (int a : sh.arity_bits) st.push_back((vxh::u64)a);
  // Poseidon digest of the program: ~4 k host permutations for a chip-sized AIR, so it is computed once per distinct program
  // and process (the words up to END are the key)
  static std::mutex mu;
  static std::map<std::vector<uint64_t>, vxh::Hash4> digests;
  std::vector<uint64_t> key;
  key.reserve((size_t)d->program_len);
  for (int pc = 0; pc < d->program_len; ++pc) {
    key.push_back(d->program[pc]);
    if ((d->program[pc] & 0xFF) == VX_OP_END) break;   // words after END are not part of the AIR
    if ((d->program[pc] & 0xFF) == VX_OP_LDI && pc + 1 < d->program_len) key.push_back(d->program[++pc]);
  }
  vxh::Hash4 ph;
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = digests.find(key);
    if (it == digests.end()) {
      std::vector<vxh::u64> limbs;
      limbs.reserve(2 * key.size());
      for (uint64_t w : key) limbs.push_back(w & 0xFFFFFFFFu), limbs.push_back(w >> 32);
      it = digests.emplace(key, vxh::hash_no_pad(limbs.data(), limbs.size())).first;
    }
    ph = it->second;
  }
  st.insert(st.end(), ph.e, ph.e + 4);
  st.insert(st.end(), canonical_pis, canonical_pis + d->num_public_inputs);
  return st;
}

// One proof in two steps (include/vxprover.h vx_stark_begin / vx_stark_finish): the trace commitment and the transcript up
// to the aux challenges live here between the calls.  The description is deep-copied (the caller's may go away).
struct vx_stark_session {
  vx_ctx* c = nullptr;
  vx_stark_desc d;
  std::vector<uint64_t> program;
  std::vector<int32_t> arities;
  StarkShape sh;
  std::vector<vxh::u64> public_inputs, trace_cap, aux_challenges;
  bool shared_challenges = false;   // vx_stark_set_aux_challenges: the challenges of a cross-table argument, drawn by the caller over ALL tables' caps
  vx_batch* trace_b = nullptr;
  vxh::Challenger ch;
  bool finished = false;
  Shard shard;   // vx_stark_begin_sharded: this rank's share of every LDE (whole cosets), exchanges through shard.fn
  ~vx_stark_session() {
    if (c && trace_b) {
      hipSetDevice(c->device);
      batch_release(c, trace_b);
    }
  }
};

static int stark_begin_impl(vx_ctx* c, const vx_stark_desc* d_in, const u64* trace_in, bool on_device, const u64* pis, vx_stark_session& s,
                            const Shard& shard = Shard()) {
  using namespace vxh;
  s.shard = shard;
  {
    const std::string why = stark_check(d_in, &s.sh);
    if (!why.empty()) return vx_fail(VX_E_INVALID, "%s", why.c_str());
  }
  s.c = c;
  s.d = *d_in;
  s.program.assign(d_in->program, d_in->program + d_in->program_len);
  s.d.program = s.program.data();
  if (d_in->override_flags & VX_DESC_HAS_FRI_ARITIES) s.arities.assign(d_in->fri_reduction_arity_bits, d_in->fri_reduction_arity_bits + d_in->num_fri_reduction_arity_bits);
  s.d.fri_reduction_arity_bits = s.arities.empty() ? nullptr : s.arities.data();
  const vx_stark_desc* d = &s.d;
  const int lg = d->degree_bits, rb = d->rate_bits, ncols = d->num_columns;
  const size_t n = (size_t)1 << lg;
  const size_t cap_words = (size_t)4 << d->cap_height;
  Scratch S(c);
  s.public_inputs.assign(pis, pis + d->num_public_inputs);
  for (auto& v : s.public_inputs) v = canon(v);
  // ---- trace commitment: PolynomialBatch::from_values(trace_poly_values, rate_bits, false, cap_height) ----
  VXCHK(batch_alloc(c, lg, ncols, rb, d->cap_height, &s.trace_b, shard.rank, shard.lg));
  if (!on_device) {   // a host trace crosses PCIe behind its own transforms and leaf hashing, like a host witness in vx_prove
    u64* w = S.get((size_t)ncols * n);
    if (!w) return vx_fail(VX_E_NOMEM, "stark: out of device memory (trace)");
    if (!getenv("VX_NO_UPLOAD_OVERLAP") && shard.world == 1) {
      VXCHK(batch_commit_host(c, s.trace_b, trace_in, w, false));
    } else {
      HIPCHK(hipMemcpyAsync(w, trace_in, (size_t)ncols * n * 8, hipMemcpyHostToDevice, c->stream));
      VXCHK(batch_commit_device(c, s.trace_b, w, n, false));
    }
  } else {
    VXCHK(batch_commit_device(c, s.trace_b, trace_in, n, false));
  }
  VXCHK(gather_cap(c, shard, S, s.trace_b->tree + s.trace_b->cap_off * 4, s.trace_b->local_cap_words(), s.trace_cap));
  {
    const std::vector<u64> st = stark_statement(d, s.sh, s.public_inputs.data());
    s.ch.observe_elements(st.data(), st.size());
  }
  s.ch.observe_elements(s.trace_cap.data(), cap_words);
  // second commitment round: its challenges are drawn here, between the trace cap and the aux cap
  s.aux_challenges.resize(d->num_aux_columns > 0 ? d->num_aux_challenges : 0);
  for (auto& v : s.aux_challenges) v = s.ch.get_challenge();
  return VX_OK;
}

static int stark_finish_impl(vx_stark_session& s, const u64* aux_in, bool aux_on_device, const u64* aux_pis_in, const u64* pow_hint,
                             std::vector<uint8_t>& proof_out) {
  using namespace vxh;
  vx_ctx* c = s.c;
  const vx_stark_desc* d = &s.d;
  const StarkShape& sh = s.sh;
  const int lg = d->degree_bits, rb = d->rate_bits, LG = lg + rb, nch = d->num_challenges, ncols = d->num_columns, naux = d->num_aux_columns;
  const size_t n = (size_t)1 << lg, N = (size_t)1 << LG;
  const size_t cap_words = (size_t)4 << d->cap_height;
  if (naux > 0 && !aux_in) return vx_fail(VX_E_INVALID, "vx_stark_finish: this AIR has %d aux columns and none were given", naux);
  const int napi = d->num_aux_public_inputs;
  if (napi > 0 && !aux_pis_in) return vx_fail(VX_E_INVALID, "vx_stark_finish: this AIR has %d aux public inputs (closing sums): use vx_stark_finish2", napi);
  std::vector<u64> aux_pis(aux_pis_in, aux_pis_in + (napi > 0 ? napi : 0));
  for (auto& v : aux_pis) v = canon(v);
  Scratch S(c);
  vx_batch* trace_b = s.trace_b;
  vx_batch *aux_b = nullptr, *quot_b = nullptr;
  struct Cleanup {
    vx_ctx* c;
    vx_batch **a, **b;
    ~Cleanup() {
      batch_release(c, *a);
      batch_release(c, *b);
    }
  } cleanup{c, &aux_b, &quot_b};
  const std::vector<u64>& public_inputs = s.public_inputs;
  const std::vector<u64>& trace_cap = s.trace_cap;
  Challenger ch = s.ch;  // a copy: a failed call (e.g. an output buffer that is too small) leaves the session where vx_stark_begin left it
  const Shard& one = s.shard;   // world 1 unless the session was begun sharded
  std::vector<u64> aux_cap, quot_cap;
  if (naux > 0) {
    // ---- the caller's second-round columns: PolynomialBatch::from_values like the trace ----
    const u64* d_aux = aux_in;
    if (!aux_on_device) {
      u64* w = S.get((size_t)naux * n);
      if (!w) return vx_fail(VX_E_NOMEM, "stark: out of device memory (aux columns)");
      HIPCHK(hipMemcpyAsync(w, aux_in, (size_t)naux * n * 8, hipMemcpyHostToDevice, c->stream));
      d_aux = w;
    }
    VXCHK(batch_alloc(c, lg, naux, rb, d->cap_height, &aux_b, one.rank, one.lg));
    VXCHK(batch_commit_device(c, aux_b, d_aux, n, false));
    VXCHK(gather_cap(c, one, S, aux_b->tree + aux_b->cap_off * 4, aux_b->local_cap_words(), aux_cap));
    ch.observe_elements(aux_cap.data(), cap_words);
    if (napi > 0) ch.observe_elements(aux_pis.data(), aux_pis.size());   // the closing sums are bound before the alphas are drawn
  }
  u64 alphas[VX_MAX_CHALLENGES] = {0, 0};
  for (int i = 0; i < nch; ++i) alphas[i] = ch.get_challenge();

  // ---- quotient: constraints on the coset 7 * H' of size n * 2^qbits = the first 2^qbits blocks of the trace LDE ----
  // Sharded by coset: rank r holds LDE blocks [r zc, (r + 1) zc); the quotient domain is blocks [0, nz).  The first nz / zcq ranks
  // ("owners", zcq = min(zc, nz) blocks each — the FIRST zcq blocks of their local LDE) evaluate the constraints on their blocks and
  // run their per-coset inverse transforms; one all-gather of the coefficient blocks ([world][nch][zcq][n], the owners' slots filled)
  // gives every rank what the cross-coset transform needs.  world = 1: zcq = nz, one owner, no exchange.
  const int qb = sh.qbits, nz = 1 << qb;
  const int zc = (1 << rb) >> one.lg, zcq = zc < nz ? zc : nz, owners = nz / zcq;
  const bool own = one.rank < owners;
  const size_t rows = own ? (size_t)zcq * n : 0;          // local quotient rows
  const size_t slot = (size_t)nch * zcq * n;               // words per rank in the exchange
  const size_t Nl = N >> one.lg;                           // local LDE rows = column stride of the local batches
  {
    u64* qv = S.get(std::max<size_t>(slot, 1));
    u64* qu = S.get(slot * one.world);
    u64* d_prog = S.get((size_t)d->program_len);
    if (!qv || !qu || !d_prog) return vx_fail(VX_E_NOMEM, "stark: out of device memory (quotient)");
    HIPCHK(hipMemcpyAsync(d_prog, d->program, (size_t)d->program_len * 8, hipMemcpyHostToDevice, c->stream));
    AirParams ap;
    memset(&ap, 0, sizeof ap);
    ap.trace = trace_b->lde;
    ap.aux = aux_b ? aux_b->lde : nullptr;
    ap.program = d_prog;
    ap.stride = Nl;
    ap.rows = rows;
    ap.row_base = (size_t)one.rank * zcq * n;
    ap.log_n = lg, ap.rate_bits = rb, ap.qbits = qb, ap.ncols = ncols, ap.nch = nch, ap.npi = d->num_public_inputs;
    ap.root_lo = c->root_lo, ap.root_hi = c->root_hi;
    for (int i = 0; i < VX_MAX_CHALLENGES; ++i) ap.alphas[i] = alphas[i];
    for (int i = 0; i < d->num_public_inputs; ++i) ap.pi[i] = public_inputs[i];
    for (int i = 0; i < napi; ++i) ap.pi[d->num_public_inputs + i] = aux_pis[i];
    for (size_t i = 0; i < s.aux_challenges.size(); ++i) ap.chal[i] = s.aux_challenges[i];
    {
      // Z_H(x) on LDE block z: x^n = 7^n * w_{2^rb}^(rev_rb(z))
      const u64 shift_n = pow(7, n), g_rate = root_of_unity(rb);
      for (int zl = 0; zl < zcq; ++zl) {   // indexed by LOCAL block: global block z = rank * zcq + zl
        const size_t z = (size_t)one.rank * zcq + zl;
        ap.zh[zl] = sub(mul(shift_n, pow(g_rate, reverse_bits(z, rb))), 1);
        ap.zh_inv[zl] = inv(ap.zh[zl]);
      }
    }
    ap.last = inv(root_of_unity(lg));
    ap.n_inv = inv((u64)n % P);
    ap.out = qv;
    {
      // the AIR program compiled to native code (jit.hip.h: same lowering as gate programs — fused multiply-adds, lazy
      // canonicalisation, registers promoted to VGPRs); the interpreter is the fallback when hiprtc is not there
      std::string why;
      const std::vector<JitAirKernel> chunks = jit_air_get(d->program, nch, ncols, c->device, &why);
      ProfScope ps(c, !chunks.empty() ? "air_quotient_eval_jit" : "air_quotient_eval", 8.0 * (double)rows * 2.0 * ncols);
      if (rows == 0) {
        // a rank without a block of the quotient domain: nothing to evaluate
      } else if (!chunks.empty()) {
        u64* rowfac = S.get(3 * rows);
        if (!rowfac) return vx_fail(VX_E_NOMEM, "stark: out of device memory (row factors)");
        ap.rowfac = rowfac;
        hipLaunchKernelGGL(air_row_factors_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, c->stream, ap);
        HIPCHK(hipMemsetAsync(qv, 0, (size_t)nch * rows * 8, c->stream));
        for (const JitAirKernel& ck : chunks) {   // every chunk ADDS its part: acc_chunk * alpha^(constraints after the chunk) / Z_H
          for (int i = 0; i < VX_MAX_CHALLENGES; ++i) ap.tail_pow[i] = pow(alphas[i], (u64)(sh.num_constraints - ck.push_end));
          void* args[] = {&ap};
          HIPCHK(hipModuleLaunchKernel(ck.fn, (unsigned)((rows + 255) / 256), 1, 1, 256, 1, 1, 0, c->stream, args, nullptr));
        }
      } else {
        hipLaunchKernelGGL(air_quotient_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, c->stream, ap);
        HIPCHK(hipGetLastError());
      }
    }
    // coset_ifft(7) on the size-(n * 2^qb) domain = per-coset inverse NTTs + the cross-coset inverse DFT, which yields the
    // quotient_degree_factor chunks of n coefficients directly (same kernels as the plonk quotient, with 2^qb cosets)
    const u64 ninv = inv((u64)n % P);
    if (own) VXCHK(run_ntt(c, qv, qu + (size_t)one.rank * slot, rows, rows, n, n, lg, nch, zcq, true, true, nullptr, 0, ninv, "quotient_intt", 16.0 * rows * nch));
    VXCHK(shard_allgather(c, one, qu, slot * 8, "quotient coset coefficients"));
    VXCHK(batch_alloc(c, lg, (size_t)nch * sh.qdf, rb, d->cap_height, &quot_b, one.rank, one.lg));
    unsigned* tail_flag = (unsigned*)S.get(1);
    if (!tail_flag) return vx_fail(VX_E_NOMEM, "stark: out of device memory");
    HIPCHK(hipMemsetAsync(tail_flag, 0, 8, c->stream));
    {
      std::vector<u64> inv_shifts(nz);
      const u64 wq = root_of_unity(lg + qb);
      for (int z = 0; z < nz; ++z) inv_shifts[z] = inv(mul(7, pow(wq, reverse_bits((size_t)z, qb))));
      const int bits = lg / 2;
      u64* tab = nullptr;
      VXCHK(get_scale_tables(c, lg, bits, inv_shifts, 1, &tab));
      ChunkParams cp;
      memset(&cp, 0, sizeof cp);
      cp.u = qu;
      cp.t = quot_b->coeffs;
      cp.inv_tab = tab;
      cp.log_n = lg, cp.rb = qb, cp.bits = bits, cp.nch = nch, cp.zc = zcq;
      cp.keep = sh.qdf;  // trim_to_len(degree * quotient_degree_factor): chunks [0, qdf) of the 2^qb the transform yields
      cp.tail_nonzero = tail_flag;
      u64 wr_inv = qb ? inv(root_of_unity(qb)) : 1, pw = 1;
      for (int i = 0; i < nz; ++i) {
        cp.w_rate_inv_pows[i] = pw;
        pw = mul(pw, wr_inv);
      }
      const u64 s_inv = inv(pow(7, n)), nz_inv = inv((u64)nz);
      pw = nz_inv;
      for (int q = 0; q < nz; ++q) {
        cp.chunk_scale[q] = pw;
        pw = mul(pw, s_inv);
      }
      ProfScope ps(c, "quotient_chunks", 16.0 * (double)(n << qb) * nch);
      hipLaunchKernelGGL(quotient_chunks_kernel, dim3((unsigned)((n + 255) / 256), nch), dim3(256), 0, c->stream, cp);
      HIPCHK(hipGetLastError());
    }
    if (sh.qdf < nz) {
      // the dropped chunks must be zero (deg quotient < qdf * n), else the trace does not satisfy the AIR
      unsigned flag = 0;
      HIPCHK(hipMemcpyAsync(&flag, tail_flag, 4, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      if (flag) return vx_fail(VX_E_PROOF, "stark: quotient has degree >= quotient_degree_factor * n (trace does not satisfy the AIR?)");
    }
    VXCHK(batch_lde_and_tree(c, quot_b));
  }
  VXCHK(gather_cap(c, one, S, quot_b->tree + quot_b->cap_off * 4, quot_b->local_cap_words(), quot_cap));
  ch.observe_elements(quot_cap.data(), cap_words);
  const Ext zeta = ch.get_extension_challenge();
  {
    Ext zp = zeta;
    for (int i = 0; i < lg; ++i) zp = emul(zp, zp);
    if (zp.a == 1 && zp.b == 0) return vx_fail(VX_E_PROOF, "Opening point is in the subgroup.");
  }
  const u64 g = root_of_unity(lg);
  const Ext gzeta{mul(zeta.a, g), mul(zeta.b, g)};
  // ---- StarkOpeningSet: local_values, next_values [, aux local / next], quotient_polys ----
  std::vector<u64> ev_trace(2 * (size_t)ncols), ev_next(2 * (size_t)ncols), ev_aux(2 * (size_t)naux), ev_aux_next(2 * (size_t)naux), ev_quot(2 * quot_b->ncols);
  uint64_t openings_root[4] = {0, 0, 0, 0};            // VX_STARK_OPENINGS_DIGEST: the tree hash of the openings, from the device
  {
    const EvalJob jobs[5] = {{trace_b->coeffs, (size_t)ncols, 0, ev_trace.data()},
                             {naux ? aux_b->coeffs : nullptr, (size_t)naux, 0, ev_aux.data()},
                             {quot_b->coeffs, quot_b->ncols, 0, ev_quot.data()},
                             {trace_b->coeffs, (size_t)ncols, 1, ev_next.data()},
                             {naux ? aux_b->coeffs : nullptr, (size_t)naux, 1, ev_aux_next.data()}};
    VXCHK(batch_eval_ext_many(c, zeta, gzeta, lg, jobs, 5, (d->override_flags & VX_STARK_OPENINGS_DIGEST) ? openings_root : nullptr));
  }
  // to_fri_openings: zeta batch = [local_values, aux local, quotient_polys] (FRI-oracle order), zeta_next batch = [next_values, aux next]
  std::vector<Ext> batch0, batch1;
  for (int i = 0; i < ncols; ++i) batch0.push_back(Ext{ev_trace[2 * i], ev_trace[2 * i + 1]});
  for (int i = 0; i < naux; ++i) batch0.push_back(Ext{ev_aux[2 * i], ev_aux[2 * i + 1]});
  for (size_t i = 0; i < quot_b->ncols; ++i) batch0.push_back(Ext{ev_quot[2 * i], ev_quot[2 * i + 1]});
  for (int i = 0; i < ncols; ++i) batch1.push_back(Ext{ev_next[2 * i], ev_next[2 * i + 1]});
  for (int i = 0; i < naux; ++i) batch1.push_back(Ext{ev_aux_next[2 * i], ev_aux_next[2 * i + 1]});
  if (d->override_flags & VX_STARK_OPENINGS_DIGEST) {
    ch.observe_elements(openings_root, 4);
  } else {
    for (Ext e : batch0) ch.observe_ext(e);
    for (Ext e : batch1) ch.observe_ext(e);
  }
  // ---- opening proof: the same FRI prover as vx_prove ----
  FriProverParams fp;
  fp.degree_bits = lg, fp.rate_bits = rb, fp.cap_height = d->cap_height, fp.pow_bits = d->pow_bits, fp.num_queries = d->num_query_rounds;
  fp.arity_bits = sh.arity_bits;
  std::vector<vx_batch*> oracles = {trace_b};
  std::vector<FriRange> batch0_ranges = {FriRange{0, 0, (size_t)ncols}}, batch1_ranges = {FriRange{0, 0, (size_t)ncols}};
  if (naux) {
    oracles.push_back(aux_b);
    batch0_ranges.push_back(FriRange{1, 0, (size_t)naux});
    batch1_ranges.push_back(FriRange{1, 0, (size_t)naux});
  }
  oracles.push_back(quot_b);
  batch0_ranges.push_back(FriRange{(int)oracles.size() - 1, 0, quot_b->ncols});
  FriParts fri;
  VXCHK(fri_prove_openings(c, fp, oracles, batch0_ranges, batch1_ranges, zeta, gzeta, batch0, batch1, ch, pow_hint, one, S, fri));
  ByteSink w;
  w.words(trace_cap.data(), cap_words);
  if (naux) w.words(aux_cap.data(), cap_words);
  w.words(quot_cap.data(), cap_words);
  w.words(ev_trace.data(), ev_trace.size());
  w.words(ev_next.data(), ev_next.size());
  w.words(ev_aux.data(), ev_aux.size());
  w.words(ev_aux_next.data(), ev_aux_next.size());
  w.words(ev_quot.data(), ev_quot.size());
  write_fri_proof(w, fp, oracles, fri, one);
  w.words(public_inputs.data(), public_inputs.size());
  w.words(aux_pis.data(), aux_pis.size());
  proof_out.swap(w.b);
  return VX_OK;
}

// vx_stark_prove: both steps in one call (an AIR without a second commitment round)
static int stark_prove_impl(vx_ctx* c, const vx_stark_desc* d, const u64* trace_in, bool on_device, const u64* pis, const u64* pow_hint,
                            std::vector<uint8_t>& proof_out) {
  if (d->num_aux_columns != 0) return vx_fail(VX_E_INVALID, "vx_stark_prove: this AIR has a second commitment round: use vx_stark_begin / vx_stark_finish");
  vx_stark_session s;
  VXCHK(stark_begin_impl(c, d, trace_in, on_device, pis, s));
  return stark_finish_impl(s, nullptr, false, nullptr, pow_hint, proof_out);
}

// ---------------------------------------------------------------------------------------------------------------
// Host verifier (starky/src/verifier.rs::verify_stark_proof_with_challenges + fri/verifier.rs), straight on the bytes.
// ---------------------------------------------------------------------------------------------------------------
namespace vxsv {
using vxv::E;
using vxv::Reader;
using vxh::u64;

// The AIR program over extension-field openings (local = trace(zeta), next = trace(g zeta))
static inline void eval_air_ext(const vx_stark_desc* d, const E* local, const E* next, const u64* pis, const u64* aux_challenges, E z_last, E l_first,
                                E l_last, const u64* alphas, int nch, E* acc) {
  E R[VX_PROGRAM_REGS];
  for (int c = 0; c < nch; ++c) acc[c] = E();
  for (int pc = 0; pc < d->program_len; ++pc) {
    const uint64_t ins = d->program[pc];
    const int op = (int)(ins & 0xFF), dst = (int)((ins >> 8) & 63), a = (int)((ins >> 16) & 0xFFFF), b = (int)((ins >> 32) & 0xFFFF);
    if (op == VX_OP_END) break;
    switch (op) {
      case VX_OP_LDW: R[dst] = local[a]; break;
      case VX_OP_LDN: R[dst] = next[a]; break;
      case VX_OP_LDI: R[dst] = E(vxh::canon(d->program[++pc])); break;
      case VX_OP_LDP: R[dst] = E(vxh::canon(pis[a])); break;
      case VX_OP_LDCH: R[dst] = E(aux_challenges[a]); break;
      case VX_OP_ADD: R[dst] = R[a & 63] + R[b & 63]; break;
      case VX_OP_SUB: R[dst] = R[a & 63] - R[b & 63]; break;
      case VX_OP_MUL: R[dst] = R[a & 63] * R[b & 63]; break;
      case VX_OP_PUSH: {
        E t = R[a & 63];
        if (b == VX_AIR_TRANSITION) t = t * z_last;
        else if (b == VX_AIR_FIRST_ROW) t = t * l_first;
        else if (b == VX_AIR_LAST_ROW) t = t * l_last;
        for (int c = 0; c < nch; ++c) acc[c] = acc[c] * E(alphas[c]) + t;
        break;
      }
      default: break;
    }
  }
}

static std::string verify(const vx_stark_desc* d, const StarkShape& sh, const u64* pis_expected, const uint8_t* bytes, size_t len,
                          const u64* shared_challenges = nullptr, u64* aux_pis_out = nullptr) {
  const int lg = d->degree_bits, rb = d->rate_bits, LG = lg + rb, nch = d->num_challenges, ncols = d->num_columns;
  const size_t n = (size_t)1 << lg, N = (size_t)1 << LG, cap_len = (size_t)1 << d->cap_height, R = sh.arity_bits.size();
  const size_t nquot = (size_t)nch * sh.qdf;
  const int naux = d->num_aux_columns, noracles = naux ? 3 : 2;   // FRI oracles: trace [, aux], quotient
  Reader r{bytes, len};
  std::vector<u64> trace_cap, aux_cap, quot_cap;
  r.words(trace_cap, 4 * cap_len);
  if (naux) r.words(aux_cap, 4 * cap_len);
  r.words(quot_cap, 4 * cap_len);
  std::vector<E> o_local, o_next, o_aux, o_aux_next, o_quot;
  r.exts(o_local, ncols);
  r.exts(o_next, ncols);
  r.exts(o_aux, naux);
  r.exts(o_aux_next, naux);
  r.exts(o_quot, nquot);
  std::vector<std::vector<u64>> commit_caps(R);
  for (auto& cp : commit_caps) r.words(cp, 4 * cap_len);
  size_t widths[3] = {(size_t)ncols, nquot, 0};
  if (naux) widths[1] = (size_t)naux, widths[2] = nquot;
  struct Query {
    std::vector<u64> leaf[3], path[3];
    std::vector<std::vector<E>> step_evals;
    std::vector<std::vector<u64>> step_path;
  };
  std::vector<Query> queries(d->num_query_rounds);
  for (auto& q : queries) {
    for (int t = 0; t < noracles; ++t) {
      r.words(q.leaf[t], widths[t]);
      r.words(q.path[t], 4 * (size_t)r.u8());
    }
    q.step_evals.resize(R);
    q.step_path.resize(R);
    for (size_t k = 0; k < R; ++k) {
      r.exts(q.step_evals[k], (size_t)1 << sh.arity_bits[k]);
      r.words(q.step_path[k], 4 * (size_t)r.u8());
    }
    if (!r.ok) return "malformed proof";
  }
  size_t final_len = n;
  for (int ab : sh.arity_bits) final_len >>= ab;
  std::vector<E> final_poly;
  r.exts(final_poly, final_len);
  const u64 pow_witness = r.f();
  std::vector<u64> pis;
  r.words(pis, d->num_public_inputs);
  std::vector<u64> aux_pis;
  r.words(aux_pis, d->num_aux_public_inputs);
  if (!r.ok || r.pos != len) return "malformed proof (length or non-canonical field element)";
  for (int i = 0; i < d->num_public_inputs; ++i)
    if (pis[i] != vxh::canon(pis_expected[i])) return "public inputs differ from the expected ones";
  if (aux_pis_out)
    for (int i = 0; i < d->num_aux_public_inputs; ++i) aux_pis_out[i] = aux_pis[i];

  // ---- challenges: statement (shape, program digest, public inputs) -> trace cap -> [aux challenges -> aux cap] -> alphas
  //      -> quotient cap -> zeta -> openings -> FRI  (starky get_challenges, with the statement in front) ----
  vxh::Challenger ch;
  {
    const std::vector<u64> st = stark_statement(d, sh, pis.data());
    ch.observe_elements(st.data(), st.size());
  }
  ch.observe_elements(trace_cap.data(), trace_cap.size());
  std::vector<u64> aux_challenges(naux ? d->num_aux_challenges : 0);
  if (naux) {
    for (auto& v : aux_challenges) v = ch.get_challenge();
    if (shared_challenges) {   // a cross-table argument: the caller's challenges (drawn over every table's cap) replace the table's own
      for (size_t i = 0; i < aux_challenges.size(); ++i) aux_challenges[i] = vxh::canon(shared_challenges[i]);
      ch.observe_elements(aux_challenges.data(), aux_challenges.size());
    }
    ch.observe_elements(aux_cap.data(), aux_cap.size());
    if (!aux_pis.empty()) ch.observe_elements(aux_pis.data(), aux_pis.size());
  } else if (shared_challenges) {
    return "shared challenges given for an AIR without a second commitment round";
  }
  std::vector<u64> alphas(nch);
  for (auto& v : alphas) v = ch.get_challenge();
  ch.observe_elements(quot_cap.data(), quot_cap.size());
  const E zeta = ch.get_extension_challenge();
  if (d->override_flags & VX_STARK_OPENINGS_DIGEST) {     // the tree hash of the openings (include/vxprover.h), here on the host
    std::vector<u64> flat;
    for (auto* v : {&o_local, &o_aux, &o_quot, &o_next, &o_aux_next})
      for (E e : *v) flat.push_back(e.a), flat.push_back(e.b);
    size_t leaves = 2;
    while (leaves * 8 < flat.size()) leaves <<= 1;
    flat.resize(leaves * 8, 0);
    std::vector<vxh::Hash4> level(leaves);
    for (size_t i = 0; i < leaves; ++i) level[i] = vxh::hash_no_pad(flat.data() + 8 * i, 8);
    while (level.size() > 1) {
      std::vector<vxh::Hash4> next(level.size() / 2);
      for (size_t i = 0; i < next.size(); ++i) next[i] = vxv::two_to_one(level[2 * i], level[2 * i + 1]);
      level.swap(next);
    }
    ch.observe_elements(level[0].e, 4);
  } else {
    for (auto* v : {&o_local, &o_aux, &o_quot})
      for (E e : *v) ch.observe_ext(e.x());
    for (auto* v : {&o_next, &o_aux_next})
      for (E e : *v) ch.observe_ext(e.x());
  }
  const E fri_alpha = ch.get_extension_challenge();
  std::vector<E> fri_betas;
  for (auto& cp : commit_caps) {
    ch.observe_elements(cp.data(), cp.size());
    fri_betas.push_back(ch.get_extension_challenge());
  }
  for (E e : final_poly) ch.observe_ext(e.x());
  ch.observe_element(pow_witness);
  const u64 pow_response = ch.get_challenge();
  std::vector<size_t> x_indices(d->num_query_rounds);
  for (auto& x : x_indices) x = (size_t)(ch.get_challenge() % (u64)N);

  // ---- vanishing(zeta) = Z_H(zeta) * quotient(zeta) ----
  {
    E zeta_n = zeta;
    for (int i = 0; i < lg; ++i) zeta_n = zeta_n * zeta_n;
    const E z_h = zeta_n - E(1);
    const u64 last = vxh::inv(vxh::root_of_unity(lg)), n_inv = vxh::inv((u64)n % vxh::P);
    const E z_last = zeta - E(last);
    const E l_first = vxv::scale(z_h, n_inv) * vxv::inv(zeta - E(1));
    const E l_last = vxv::scale(vxv::scale(z_h, n_inv), last) * vxv::inv(z_last);
    E acc[VX_MAX_CHALLENGES];
    std::vector<E> all_local(o_local), all_next(o_next);   // program columns: trace, then aux
    all_local.insert(all_local.end(), o_aux.begin(), o_aux.end());
    all_next.insert(all_next.end(), o_aux_next.begin(), o_aux_next.end());
    std::vector<u64> all_pis(pis);     // VX_OP_LDP index space: public inputs, then the aux public inputs
    all_pis.insert(all_pis.end(), aux_pis.begin(), aux_pis.end());
    eval_air_ext(d, all_local.data(), all_next.data(), all_pis.data(), aux_challenges.data(), z_last, l_first, l_last, alphas.data(), nch, acc);
    for (int k = 0; k < nch; ++k) {
      E q;
      for (int j = sh.qdf; j-- > 0;) q = q * zeta_n + o_quot[(size_t)k * sh.qdf + j];
      if (acc[k] != z_h * q) return "constraint identity fails at zeta (challenge " + std::to_string(k) + ")";
    }
  }
  // ---- FRI ----
  if (d->pow_bits > 0 && (pow_response >> (64 - d->pow_bits)) != 0) return "proof of work check failed";
  const E points[2] = {zeta, vxv::scale(zeta, vxh::root_of_unity(lg))};
  E reduced[2];
  {
    std::vector<E> b0(o_local), b1(o_next);
    b0.insert(b0.end(), o_aux.begin(), o_aux.end());
    b0.insert(b0.end(), o_quot.begin(), o_quot.end());
    b1.insert(b1.end(), o_aux_next.begin(), o_aux_next.end());
    for (size_t i = b0.size(); i-- > 0;) reduced[0] = reduced[0] * fri_alpha + b0[i];
    for (size_t i = b1.size(); i-- > 0;) reduced[1] = reduced[1] * fri_alpha + b1[i];
  }
  const u64* caps[3] = {trace_cap.data(), naux ? aux_cap.data() : quot_cap.data(), quot_cap.data()};
  const u64 wN = vxh::root_of_unity(LG);
  for (int qi = 0; qi < d->num_query_rounds; ++qi) {
    const Query& q = queries[qi];
    size_t xi = x_indices[qi];
    for (int t = 0; t < noracles; ++t) {
      if (q.path[t].size() != 4 * (size_t)(LG - d->cap_height)) return "initial Merkle proof has the wrong length";
      if (!vxv::merkle_ok(q.leaf[t].data(), widths[t], xi, caps[t], cap_len, q.path[t])) return "initial Merkle proof fails (oracle " + std::to_string(t) + ")";
    }
    u64 sx = vxh::mul(7, vxh::pow(wN, vxh::reverse_bits(xi, LG)));
    E sum;
    for (int b = 0; b < 2; ++b) {
      std::vector<u64> ev;
      if (b == 0)
        for (int t = 0; t < noracles; ++t) ev.insert(ev.end(), q.leaf[t].begin(), q.leaf[t].end());
      else
        for (int t = 0; t < noracles - 1; ++t) ev.insert(ev.end(), q.leaf[t].begin(), q.leaf[t].end());   // trace [, aux] at g * zeta
      E red;
      for (size_t i = ev.size(); i-- > 0;) red = red * fri_alpha + E(ev[i]);
      sum = sum * E(vxh::epow(fri_alpha.x(), ev.size())) + (red - reduced[b]) * vxv::inv(E(sx) - points[b]);
    }
    E old_eval = sum;
    for (size_t k = 0; k < R; ++k) {
      const int ab = sh.arity_bits[k];
      const size_t arity = (size_t)1 << ab, coset = xi >> ab, within = xi & (arity - 1);
      const std::vector<E>& evals = q.step_evals[k];
      if (evals[within] != old_eval) return "FRI consistency check fails at round " + std::to_string(k);
      {
        const u64 g = vxh::root_of_unity(ab);
        const u64 start = vxh::mul(sx, vxh::pow(g, arity - vxh::reverse_bits(within, ab)));
        std::vector<E> xs(arity), ys(arity);
        u64 pw = 1;
        for (size_t j = 0; j < arity; ++j) {
          xs[j] = E(vxh::mul(start, pw));
          ys[j] = evals[vxh::reverse_bits(j, ab)];
          pw = vxh::mul(pw, g);
        }
        E acc;
        for (size_t i = 0; i < arity; ++i) {
          E num(1), den(1);
          for (size_t j = 0; j < arity; ++j)
            if (j != i) num = num * (fri_betas[k] - xs[j]), den = den * (xs[i] - xs[j]);
          acc = acc + ys[i] * num * vxv::inv(den);
        }
        old_eval = acc;
      }
      {
        int layer_bits = LG;
        for (size_t j = 0; j <= k; ++j) layer_bits -= sh.arity_bits[j];
        const int depth = layer_bits > d->cap_height ? layer_bits - d->cap_height : 0;
        if (q.step_path[k].size() != 4 * (size_t)depth) return "FRI commit-phase Merkle proof has the wrong length";
      }
      std::vector<u64> flat(2 * arity);
      for (size_t j = 0; j < arity; ++j) flat[2 * j] = evals[j].a, flat[2 * j + 1] = evals[j].b;
      if (!vxv::merkle_ok(flat.data(), flat.size(), coset, commit_caps[k].data(), cap_len, q.step_path[k]))
        return "FRI commit-phase Merkle proof fails at round " + std::to_string(k);
      for (int j = 0; j < ab; ++j) sx = vxh::mul(sx, sx);
      xi = coset;
    }
    E fin;
    for (size_t i = final_poly.size(); i-- > 0;) fin = fin * E(sx) + final_poly[i];
    if (fin != old_eval) return "final polynomial evaluation mismatch";
  }
  return "";
}
}  // namespace vxsv
