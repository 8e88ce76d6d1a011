// TEST INFRASTRUCTURE — CPU restatement of a STARK prover / verifier on plonky2's primitives (the checker for
// vectorx_amd/csrc/stark.hip.h, SURVEY.md §8 f-3 scoping spike).  parity unpinned.
//
// Follows plonky2's `starky` crate as recalled (starky/src/{prover,verifier,constraint_consumer,stark,proof}.rs, v0.2.0
// era; the crate is not vendored in /root/reference and Curta = starkyx v1.0.0, the prover the reference actually embeds
// — /root/reference/Cargo.lock:7232-7234, circuits/builder/header.rs:18 — has the same shape with its own challenge
// schedule).  Written in the straightforward textbook order (natural-order coset, Lagrange selectors by formula,
// coefficient-domain FRI fold), independent of the GPU code's layout tricks.
#pragma once
#include <functional>
#include "plonk.hpp"

namespace vxo {

static const int AIR_OP_END = 0, AIR_OP_LDW = 1, AIR_OP_LDI = 3, AIR_OP_ADD = 4, AIR_OP_SUB = 5, AIR_OP_MUL = 6, AIR_OP_PUSH = 7,
                 AIR_OP_LDP = 8, AIR_OP_LDN = 9, AIR_OP_LDCH = 10;
static const int AIR_ALL = 0, AIR_TRANSITION = 1, AIR_FIRST = 2, AIR_LAST = 3;

struct StarkDesc {
  int degree_bits = 0, num_columns = 0, num_public_inputs = 0;
  int rate_bits = 1, cap_height = 4, pow_bits = 16, num_query_rounds = 84, num_challenges = 2;  // StarkConfig::standard_fast_config
  int constraint_degree = 2;
  // second commitment round (the shape of Curta's lookup / bus accumulators and of starky's permutation Z columns): after
  // the trace cap `num_aux_challenges` challenges are drawn, the prover commits `num_aux_columns` more columns that may
  // depend on them, and the AIR program sees both (columns num_columns.. = aux columns, AIR_OP_LDCH = a challenge)
  int num_aux_columns = 0, num_aux_challenges = 0;
  // values announced after the second commitment (closing sums of bus / lookup accumulators; starky's ctl_zs_last): observed
  // after the aux cap, read by the program as public inputs num_public_inputs..
  int num_aux_public_inputs = 0;
  std::vector<u64> program;
  std::vector<int> arity_bits;
  // the transcript takes a tree hash of the openings (openings_digest below) instead of the openings themselves — the builder's own
  // option (include/vxprover.h VX_STARK_OPENINGS_DIGEST): a table of ~1000 columns otherwise costs ~1000 host permutations here
  bool openings_digest = false;
  int quotient_degree_factor() const { return constraint_degree > 1 ? constraint_degree - 1 : 1; }  // Stark::quotient_degree_factor
  void default_arities() {  // ConstantArityBits(4, 5)
    arity_bits.clear();
    int db = degree_bits;
    while (db > 5 && db + rate_bits - 4 >= cap_height) arity_bits.push_back(4), db -= 4;
  }
};
// Tree hash of a sequence of field elements: 8-element chunks (the last one zero-padded) hashed with hash_no_pad — one permutation
// each; the SEQUENCE is zero-padded to 8 * 2^k elements (k >= 1), every chunk is a leaf, and
// a binary two_to_one tree over the 2^k leaf digests gives the root.
static Hash openings_digest(const std::vector<u64>& flat) {
  size_t leaves = (flat.size() + 7) / 8, p2 = 2;
  while (p2 < leaves) p2 <<= 1;
  std::vector<u64> padded(flat);
  padded.resize(8 * p2, 0);
  std::vector<Hash> level(p2);
  for (size_t i = 0; i < p2; ++i) level[i] = hash_no_pad(padded.data() + 8 * i, 8);
  while (level.size() > 1) {
    std::vector<Hash> next(level.size() / 2);
    for (size_t i = 0; i < next.size(); ++i) next[i] = two_to_one(level[2 * i], level[2 * i + 1]);
    level.swap(next);
  }
  return level[0];
}
struct StarkProof {
  std::vector<Hash> trace_cap, aux_cap, quotient_cap;
  std::vector<Ext> local_values, next_values, aux_local_values, aux_next_values, quotient_polys;
  FriProof fri;
  std::vector<u64> public_inputs, aux_public_inputs;
};

// ConstraintConsumer + Stark::eval_packed_generic / eval_ext: the AIR as a straight-line program over (local, next, pis)
template <class T>
static void eval_air(const StarkDesc& d, const T* local, const T* next, const u64* pis, const u64* aux_challenges, T z_last, T l_first, T l_last,
                     const u64* alphas, T* acc) {
  T R[64];
  for (int c = 0; c < d.num_challenges; ++c) acc[c] = T();
  for (size_t pc = 0; pc < d.program.size(); ++pc) {
    const u64 ins = d.program[pc];
    const int op = (int)(ins & 0xFF), dst = (int)((ins >> 8) & 63), a = (int)((ins >> 16) & 0xFFFF), b = (int)((ins >> 32) & 0xFFFF);
    if (op == AIR_OP_END) break;
    if (op == AIR_OP_LDW) R[dst] = local[a];
    else if (op == AIR_OP_LDN) R[dst] = next[a];
    else if (op == AIR_OP_LDI) R[dst] = T(canon(d.program[++pc]));
    else if (op == AIR_OP_LDP) R[dst] = T(canon(pis[a]));
    else if (op == AIR_OP_LDCH) R[dst] = T(aux_challenges[a]);
    else if (op == AIR_OP_ADD) R[dst] = R[a & 63] + R[b & 63];
    else if (op == AIR_OP_SUB) R[dst] = R[a & 63] - R[b & 63];
    else if (op == AIR_OP_MUL) R[dst] = R[a & 63] * R[b & 63];
    else if (op == AIR_OP_PUSH) {
      T t = R[a & 63];
      if (b == AIR_TRANSITION) t = t * z_last;        // constraint_transition
      else if (b == AIR_FIRST) t = t * l_first;       // constraint_first_row
      else if (b == AIR_LAST) t = t * l_last;         // constraint_last_row
      for (int c = 0; c < d.num_challenges; ++c) acc[c] = acc[c] * T(alphas[c]) + t;  // acc *= alpha; acc += constraint
    } else
      throw std::runtime_error("bad AIR opcode");
  }
}

// fri::oracle::PolynomialBatch::prove_openings + fri_proof for two opening points: batch 0 = every polynomial of every
// oracle at z0, batch 1 = `batch1` at z1.
static void fri_prove_two_points(const std::vector<const PolynomialBatch*>& oracles, const std::vector<const std::vector<u64>*>& batch1, Ext z0, Ext z1,
                                 int lg, int rb, int cap_height, int pow_bits, int num_queries, const std::vector<int>& arity_bits, Challenger& ch,
                                 const ProveOptions& opt, FriProof& out) {
  const int LG = lg + rb;
  const size_t n = (size_t)1 << lg, N = (size_t)1 << LG;
  Ext alpha = ch.get_extension_challenge();
  std::vector<Ext> final_poly(n, Ext());
  for (int batch = 0; batch < 2; ++batch) {
    std::vector<const std::vector<u64>*> polys;
    if (batch == 0) {
      for (const PolynomialBatch* o : oracles)
        for (size_t k = 0; k < o->ncols; ++k) polys.push_back(&o->coeffs[k]);
    } else
      polys = batch1;
    const Ext point = batch == 0 ? z0 : z1;
    std::vector<Ext> comp(n, Ext());
    {
      std::vector<Ext> apow(polys.size());
      Ext a(1);
      for (size_t j = 0; j < polys.size(); ++j) apow[j] = a, a = a * alpha;
      for (size_t i = 0; i < n; ++i) {
        Ext acc;
        for (size_t j = 0; j < polys.size(); ++j) acc = acc + scale(apow[j], (*polys[j])[i]);
        comp[i] = acc;
      }
    }
    std::vector<Ext> quo(n, Ext());
    {
      Ext acc;
      for (size_t i = n; i-- > 0;) {
        acc = acc * point + comp[i];
        if (i > 0) quo[i - 1] = acc;
      }
    }
    const Ext sh = ext_pow(alpha, polys.size());
    for (size_t i = 0; i < n; ++i) final_poly[i] = final_poly[i] * sh + quo[i];
  }
  std::vector<Ext> coeffs(N, Ext());
  for (size_t i = 0; i < n; ++i) coeffs[i] = final_poly[i];
  std::vector<Ext> values = coeffs;
  coset_fft_ext_inplace(values, LG, MULTIPLICATIVE_GENERATOR);
  std::vector<MerkleTree> trees;
  {
    u64 shift = MULTIPLICATIVE_GENERATOR;
    for (int ab : arity_bits) {
      const size_t arity = (size_t)1 << ab;
      reverse_index_bits_in_place(values);
      std::vector<u64> leaves(values.size() * 2);
      for (size_t i = 0; i < values.size(); ++i) leaves[2 * i] = values[i].a, leaves[2 * i + 1] = values[i].b;
      MerkleTree t;
      t.build(std::move(leaves), 2 * arity, cap_height);
      ch.observe_cap(t.cap());
      out.commit_phase_caps.push_back(t.cap());
      trees.push_back(std::move(t));
      const Ext beta = ch.get_extension_challenge();
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
    out.final_poly = coeffs;
  }
  {
    State base = ch.sponge;
    for (size_t i = 0; i < ch.input.size(); ++i) base[i] = ch.input[i];
    const size_t pos = ch.input.size();
    auto ok = [&](u64 cand) {
      State s = base;
      s[pos] = cand;
      permute(s);
      return pow_bits == 0 || (s[SPONGE_RATE - 1] >> (64 - pow_bits)) == 0;
    };
    u64 wts = 0;
    if (opt.has_pow_hint) {
      wts = opt.pow_hint;
      if (!ok(wts)) throw std::runtime_error("pow_witness hint does not satisfy the proof-of-work condition");
    } else
      while (!ok(wts)) ++wts;  // smallest valid witness
    out.pow_witness = wts;
    ch.observe_element(wts);
    (void)ch.get_challenge();
  }
  for (int q = 0; q < num_queries; ++q) {
    const size_t x_index = (size_t)(ch.get_challenge() % (u64)N);
    FriQueryRound qr;
    for (const PolynomialBatch* o : oracles) {
      const MerkleTree& t = o->tree;
      qr.initial.evals.emplace_back(t.leaf(x_index), t.leaf(x_index) + t.width);
      qr.initial.proofs.push_back(t.prove(x_index));
    }
    size_t xi = x_index;
    for (size_t r = 0; r < trees.size(); ++r) {
      const int ab = arity_bits[r];
      const size_t coset = xi >> ab;
      FriQueryStep st;
      const u64* lf = trees[r].leaf(coset);
      for (size_t k = 0; k < ((size_t)1 << ab); ++k) st.evals.push_back(Ext(lf[2 * k], lf[2 * k + 1]));
      st.proof = trees[r].prove(coset);
      qr.steps.push_back(std::move(st));
      xi = coset;
    }
    out.query_rounds.push_back(std::move(qr));
  }
}

// starky/src/prover.rs::prove_with_commitment; with num_aux_columns > 0 a second commitment round sits between the trace
// cap and the alphas (where starky commits its permutation Z polynomials and Curta its accumulators): `aux_fn` maps the
// drawn challenges to the aux columns.
// aux_fn(challenges, aux_public_inputs_out) -> aux columns
typedef std::function<std::vector<std::vector<u64>>(const std::vector<u64>&, std::vector<u64>&)> StarkAuxFn;
// `shared_challenges` (cross-table arguments: several tables on one bus): challenges the caller drew over EVERY table's trace
// cap (stark_joint_challenges below); they replace the table's own and are observed into its transcript.
static StarkProof stark_prove(const StarkDesc& d, const std::vector<std::vector<u64>>& trace, const std::vector<u64>& pis, const ProveOptions& opt = ProveOptions(),
                              const StarkAuxFn& aux_fn = StarkAuxFn(), const std::vector<u64>* shared_challenges = nullptr) {
  const int lg = d.degree_bits, rb = d.rate_bits, nch = d.num_challenges;
  const size_t n = (size_t)1 << lg;
  StarkProof proof;
  proof.public_inputs = pis;
  for (auto& v : proof.public_inputs) v = canon(v);
  PolynomialBatch trace_b, aux_b, quot_b;
  {
    std::vector<std::vector<u64>> cols = trace;
    trace_b.from_values(std::move(cols), rb, d.cap_height);
  }
  Challenger ch;
  {
    // the statement first (shape + FRI configuration, Poseidon digest of the program as 32-bit limbs, public inputs): the
    // library's own transcript prefix — old starky observes nothing before the trace cap (weak Fiat-Shamir)
    std::vector<u64> st = {(u64)d.degree_bits, (u64)d.rate_bits, (u64)d.cap_height, (u64)d.pow_bits, (u64)d.num_query_rounds, (u64)d.num_challenges,
                           (u64)d.constraint_degree, (u64)d.num_columns, (u64)d.num_aux_columns, (u64)d.num_aux_challenges, (u64)d.num_public_inputs,
                           (u64)d.num_aux_public_inputs, (u64)d.arity_bits.size()};
    for (int a : d.arity_bits) st.push_back((u64)a);
    std::vector<u64> limbs;
    for (size_t pc = 0; pc < d.program.size(); ++pc) {
      const u64 w = d.program[pc];
      limbs.push_back(w & 0xFFFFFFFFu), limbs.push_back(w >> 32);
      if ((w & 0xFF) == (u64)AIR_OP_END) break;
      if ((w & 0xFF) == (u64)AIR_OP_LDI && pc + 1 < d.program.size()) {
        const u64 imm = d.program[++pc];
        limbs.push_back(imm & 0xFFFFFFFFu), limbs.push_back(imm >> 32);
      }
    }
    const Hash ph = hash_no_pad(limbs.data(), limbs.size());
    for (int i = 0; i < 4; ++i) st.push_back(ph.e[i]);
    if ((int)proof.public_inputs.size() != d.num_public_inputs) throw std::runtime_error("wrong number of public inputs");
    st.insert(st.end(), proof.public_inputs.begin(), proof.public_inputs.end());
    ch.observe_elements(st.data(), st.size());
  }
  ch.observe_cap(trace_b.tree.cap());
  const int naux = d.num_aux_columns, ntot = d.num_columns + naux;
  std::vector<u64> aux_challenges(d.num_aux_challenges);
  if (naux > 0) {
    for (auto& v : aux_challenges) v = ch.get_challenge();
    if (shared_challenges) {
      if (shared_challenges->size() != aux_challenges.size()) throw std::runtime_error("wrong number of shared challenges");
      for (size_t i = 0; i < aux_challenges.size(); ++i) aux_challenges[i] = canon((*shared_challenges)[i]);
      ch.observe_elements(aux_challenges.data(), aux_challenges.size());
    }
    if (!aux_fn) throw std::runtime_error("this AIR has a second commitment round: no aux column generator given");
    std::vector<std::vector<u64>> cols = aux_fn(aux_challenges, proof.aux_public_inputs);
    if ((int)proof.aux_public_inputs.size() != d.num_aux_public_inputs) throw std::runtime_error("aux column generator returned the wrong number of aux public inputs");
    for (auto& v : proof.aux_public_inputs) v = canon(v);
    if ((int)cols.size() != naux) throw std::runtime_error("aux column generator returned the wrong number of columns");
    for (auto& col : cols) {
      if (col.size() != n) throw std::runtime_error("aux column has the wrong length");
      for (auto& v : col) v = canon(v);
    }
    aux_b.from_values(std::move(cols), rb, d.cap_height);
    ch.observe_cap(aux_b.tree.cap());
    if (!proof.aux_public_inputs.empty()) ch.observe_elements(proof.aux_public_inputs.data(), proof.aux_public_inputs.size());
  }
  std::vector<u64> all_pis(proof.public_inputs);   // the program's LDP index space: public inputs, then aux public inputs
  all_pis.insert(all_pis.end(), proof.aux_public_inputs.begin(), proof.aux_public_inputs.end());
  std::vector<u64> alphas(nch);
  for (int i = 0; i < nch; ++i) alphas[i] = ch.get_challenge();
  // ---- compute_quotient_polys ----
  const int qdf = d.quotient_degree_factor();
  int qbits = 0;
  while ((1 << qbits) < qdf) ++qbits;  // log2_ceil
  if (qbits > rb) throw std::runtime_error("Having constraints of degree higher than the rate is not supported yet.");
  const size_t step = (size_t)1 << (rb - qbits), next_step = (size_t)1 << qbits, size = n << qbits;
  const u64 last = inv(root_of_unity(lg)), n_inv = inv((u64)n % P);
  std::vector<std::vector<u64>> qvals(nch, std::vector<u64>(size));
  {
    const u64 w = root_of_unity(lg + qbits);
    u64 x = MULTIPLICATIVE_GENERATOR;
    std::vector<Fp> local(ntot), next(ntot), acc(nch);
    for (size_t i = 0; i < size; ++i, x = mul(x, w)) {
      const size_t i_next = (i + next_step) % size;
      const u64* lv = trace_b.get_lde_values(i, step);
      const u64* nv = trace_b.get_lde_values(i_next, step);
      for (int c = 0; c < d.num_columns; ++c) local[c] = Fp(lv[c]), next[c] = Fp(nv[c]);
      if (naux > 0) {
        const u64* la = aux_b.get_lde_values(i, step);
        const u64* na = aux_b.get_lde_values(i_next, step);
        for (int c = 0; c < naux; ++c) local[d.num_columns + c] = Fp(la[c]), next[d.num_columns + c] = Fp(na[c]);
      }
      // Z_H(x), the Lagrange selectors of the first / last row of H evaluated on the coset, z_last = x - g^-1
      const u64 zh = sub(pow(x, (u64)n), 1);
      const u64 l_first = mul(mul(zh, n_inv), inv(sub(x, 1)));
      const u64 l_last = mul(mul(mul(zh, n_inv), last), inv(sub(x, last)));
      eval_air<Fp>(d, local.data(), next.data(), all_pis.data(), aux_challenges.data(), Fp(sub(x, last)), Fp(l_first), Fp(l_last),
                   alphas.data(), acc.data());
      const u64 zi = inv(zh);
      for (int c = 0; c < nch; ++c) qvals[c][i] = mul(acc[c].v, zi);
    }
  }
  {
    std::vector<std::vector<u64>> chunks;
    for (int c = 0; c < nch; ++c) {
      coset_ifft_inplace(qvals[c].data(), lg + qbits, MULTIPLICATIVE_GENERATOR);
      for (size_t i = (size_t)qdf * n; i < size; ++i)
        if (qvals[c][i]) throw std::runtime_error("quotient has degree >= quotient_degree_factor * n");  // trim_to_len
      for (int k = 0; k < qdf; ++k) chunks.emplace_back(qvals[c].begin() + (size_t)k * n, qvals[c].begin() + (size_t)(k + 1) * n);
    }
    quot_b.from_coeffs(std::move(chunks), rb, d.cap_height);
  }
  ch.observe_cap(quot_b.tree.cap());
  const Ext zeta = ch.get_extension_challenge();
  {
    Ext zp = zeta;
    for (int i = 0; i < lg; ++i) zp = zp * zp;
    if (zp == Ext(1)) throw std::runtime_error("Opening point is in the subgroup.");
  }
  const Ext gzeta = scale(zeta, root_of_unity(lg));
  auto eval_batch = [&](const PolynomialBatch& b, Ext z) {
    std::vector<Ext> r(b.ncols);
    for (size_t k = 0; k < b.ncols; ++k) r[k] = eval_poly_ext(b.coeffs[k].data(), b.coeffs[k].size(), z);
    return r;
  };
  proof.local_values = eval_batch(trace_b, zeta);
  proof.next_values = eval_batch(trace_b, gzeta);
  if (naux > 0) {
    proof.aux_local_values = eval_batch(aux_b, zeta);
    proof.aux_next_values = eval_batch(aux_b, gzeta);
  }
  proof.quotient_polys = eval_batch(quot_b, zeta);
  // zeta batch = [trace, aux, quotient] in FRI-oracle order, zeta_next batch = [trace, aux]
  if (d.openings_digest) {
    std::vector<u64> flat;
    for (auto* v : {&proof.local_values, &proof.aux_local_values, &proof.quotient_polys, &proof.next_values, &proof.aux_next_values})
      for (Ext e : *v) flat.push_back(e.a), flat.push_back(e.b);
    const Hash dg = openings_digest(flat);
    ch.observe_elements(dg.e, 4);
  } else {
    for (auto* v : {&proof.local_values, &proof.aux_local_values, &proof.quotient_polys})
      for (Ext e : *v) ch.observe_ext(e);
    for (auto* v : {&proof.next_values, &proof.aux_next_values})
      for (Ext e : *v) ch.observe_ext(e);
  }
  proof.trace_cap = trace_b.tree.cap();
  if (naux > 0) proof.aux_cap = aux_b.tree.cap();
  proof.quotient_cap = quot_b.tree.cap();
  std::vector<const std::vector<u64>*> batch1;
  for (size_t k = 0; k < trace_b.ncols; ++k) batch1.push_back(&trace_b.coeffs[k]);
  std::vector<const PolynomialBatch*> oracles = {&trace_b};
  if (naux > 0) {
    for (size_t k = 0; k < aux_b.ncols; ++k) batch1.push_back(&aux_b.coeffs[k]);
    oracles.push_back(&aux_b);
  }
  oracles.push_back(&quot_b);
  fri_prove_two_points(oracles, batch1, zeta, gzeta, lg, rb, d.cap_height, d.pow_bits, d.num_query_rounds, d.arity_bits, ch, opt, proof.fri);
  return proof;
}

static std::vector<uint8_t> serialize_stark_proof(const StarkProof& p) {
  ByteWriter w;
  w.cap(p.trace_cap);
  if (!p.aux_cap.empty()) w.cap(p.aux_cap);
  w.cap(p.quotient_cap);
  w.extvec(p.local_values);
  w.extvec(p.next_values);
  w.extvec(p.aux_local_values);
  w.extvec(p.aux_next_values);
  w.extvec(p.quotient_polys);
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
  for (u64 v : p.aux_public_inputs) w.f(v);
  return w.b;
}

// Challenges of a cross-table argument: one transcript over [number of tables, every table's trace cap in table order].
static std::vector<u64> stark_joint_challenges(const std::vector<std::vector<Hash>>& caps, int n) {
  Challenger ch;
  ch.observe_element((u64)caps.size());
  for (const auto& cap : caps) ch.observe_cap(cap);
  std::vector<u64> out(n);
  for (auto& v : out) v = ch.get_challenge();
  return out;
}
// The trace commitment alone (what a multi-table host needs before it can draw the joint challenges).
static std::vector<Hash> stark_trace_cap(const StarkDesc& d, const std::vector<std::vector<u64>>& trace) {
  PolynomialBatch b;
  std::vector<std::vector<u64>> cols = trace;
  b.from_values(std::move(cols), d.rate_bits, d.cap_height);
  return b.tree.cap();
}

}  // namespace vxo
