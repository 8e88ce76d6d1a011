// Host-side proof verification of the product library: `vx_verify`.
//
// Replaces plonky2::plonk::circuit_data::CircuitData::verify -> plonk/verifier.rs::verify_with_challenges +
// fri/verifier.rs::verify_fri_proof (plonky2 v0.2.0, un-vendored: /root/reference/Cargo.lock:4848-4905), the call the
// reference makes right after every prove (/root/reference/circuits/header_range.rs:167-170, circuits/rotate.rs:193).
// Verification touches 28 Merkle paths and a few hundred field elements — milliseconds of sequential work that the
// reference also does on the CPU — so it is plain host C++ on host_field.h / challenger.h; it needs only the
// verifier's view of the circuit (parameters, gate list incl. constraint programs, k_is, the constants_sigmas cap and
// the circuit digest).  It works straight on the serialised proof (util/serialization layout) and is
// independent of the test-infrastructure checker, which carries its own restated verifier (the two must agree).
#pragma once
#include <string>
#include <vector>
#include "challenger.h"

namespace vxv {
using vxh::u64;
using vxh::P;

struct E {  // F_p^2 = F_p[X]/(X^2 - 7)
  u64 a = 0, b = 0;
  E() {}
  E(u64 x) : a(x), b(0) {}
  E(u64 x, u64 y) : a(x), b(y) {}
  E(vxh::Ext e) : a(e.a), b(e.b) {}
  vxh::Ext x() const { return vxh::Ext{a, b}; }
  E operator+(E o) const { return E(vxh::add(a, o.a), vxh::add(b, o.b)); }
  E operator-(E o) const { return E(vxh::sub(a, o.a), vxh::sub(b, o.b)); }
  E operator*(E o) const { return E(vxh::emul(x(), o.x())); }
  bool operator==(E o) const { return a == o.a && b == o.b; }
  bool operator!=(E o) const { return !(*this == o); }
};
static inline E scale(E x, u64 s) { return E(vxh::mul(x.a, s), vxh::mul(x.b, s)); }
static inline E inv(E x) { return E(vxh::einv(x.x())); }

struct Reader {  // little-endian canonical field elements
  const uint8_t* p;
  size_t len, pos = 0;
  bool ok = true;
  uint8_t u8() {
    if (pos + 1 > len) return ok = false, 0;
    return p[pos++];
  }
  u64 f() {
    if (pos + 8 > len) return ok = false, 0;
    u64 v = 0;
    for (int i = 0; i < 8; ++i) v |= (u64)p[pos + i] << (8 * i);
    pos += 8;
    if (v >= P) ok = false;
    return v;
  }
  E ext() {
    u64 a = f(), b = f();
    return E(a, b);
  }
  void words(std::vector<u64>& v, size_t n) {
    v.resize(n);
    for (auto& x : v) x = f();
  }
  void exts(std::vector<E>& v, size_t n) {
    v.resize(n);
    for (auto& x : v) x = ext();
  }
};

// hashing.rs::hash_or_noop / two_to_one and merkle_proofs.rs::verify_merkle_proof_to_cap
static inline vxh::Hash4 hash_or_noop(const u64* v, size_t n) {
  if (n <= 4) {
    vxh::Hash4 h{{0, 0, 0, 0}};
    for (size_t i = 0; i < n; ++i) h.e[i] = v[i];
    return h;
  }
  return vxh::hash_no_pad(v, n);
}
static inline vxh::Hash4 two_to_one(const vxh::Hash4& l, const vxh::Hash4& r) {
  u64 s[12] = {l.e[0], l.e[1], l.e[2], l.e[3], r.e[0], r.e[1], r.e[2], r.e[3], 0, 0, 0, 0};
  vxh::poseidon(s);
  return vxh::Hash4{{s[0], s[1], s[2], s[3]}};
}
static inline bool merkle_ok(const u64* leaf, size_t width, size_t index, const u64* cap /* [2^h][4] */, size_t cap_len,
                             const std::vector<u64>& siblings /* [depth][4] */) {
  vxh::Hash4 cur = hash_or_noop(leaf, width);
  for (size_t d = 0; d < siblings.size() / 4; ++d) {
    vxh::Hash4 sib{{siblings[4 * d], siblings[4 * d + 1], siblings[4 * d + 2], siblings[4 * d + 3]}};
    cur = (index & 1) ? two_to_one(sib, cur) : two_to_one(cur, sib);
    index >>= 1;
  }
  if (index >= cap_len) return false;
  for (int i = 0; i < 4; ++i)
    if (cur.e[i] != cap[4 * index + i]) return false;
  return true;
}

struct GateV {
  int type, param, selector_index, group_start, group_end;
  const uint64_t* program;  // host copy, or nullptr
};
struct CircuitV {
  int degree_bits, num_wires, num_routed, num_challenges, rate_bits, cap_height, pow_bits, num_queries, qdf;
  int num_selectors, num_constants, num_public_inputs;
  std::vector<GateV> gates;
  std::vector<int> arity_bits;
  const u64* k_is;
  const u64* cs_cap;  // [2^cap_height][4]
  vxh::Hash4 digest;
  // lookup argument (vanishing_poly.rs check_lookup_constraints); num_luts = 0: none
  int num_luts = 0, num_lookup_selectors = 0;
  const int32_t* lut_lens = nullptr;
  const uint16_t *lut_inputs = nullptr, *lut_outputs = nullptr;
  int npp() const { return (num_routed + qdf - 1) / qdf - 1; }
  int num_sldc() const { return (num_routed / 2 + (qdf - 1) - 1) / (qdf - 1); }
  int num_lookup_polys() const { return num_luts > 0 ? 1 + num_sldc() : 0; }
  int gate_const_base() const { return num_selectors + num_lookup_selectors; }
};

static const u64 MDS_C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static inline E sbox(E x) {
  E x2 = x * x, x4 = x2 * x2, x3 = x * x2;
  return x3 * x4;
}
static inline void mds(E* s) {
  E o[12];
  for (int r = 0; r < 12; ++r) {
    E acc;
    for (int i = 0; i < 12; ++i) acc = acc + scale(s[(i + r) % 12], MDS_C[i]);
    if (r == 0) acc = acc + scale(s[0], 8);
    o[r] = acc;
  }
  for (int i = 0; i < 12; ++i) s[i] = o[i];
}

// gates/*.rs eval_unfiltered at an extension point; consts = local constants after the selectors.
static bool eval_gate(const GateV& g, const E* consts, const E* w, const vxh::Hash4& pih, std::vector<E>& out) {
  out.clear();
  switch (g.type) {
    case VX_GATE_NOOP: return true;
    case VX_GATE_LOOKUP: return true;        // gates/lookup.rs, lookup_table.rs: no gate constraints — the lookup
    case VX_GATE_LOOKUP_TABLE: return true;  // argument lives in the vanishing polynomial (lookup_terms below)
    case VX_GATE_CONSTANT:
      for (int i = 0; i < g.param; ++i) out.push_back(consts[i] - w[i]);
      return true;
    case VX_GATE_PUBLIC_INPUT:
      for (int i = 0; i < 4; ++i) out.push_back(w[i] - E(pih.e[i]));
      return true;
    case VX_GATE_ARITHMETIC:
      for (int i = 0; i < g.param; ++i) out.push_back(w[4 * i + 3] - (w[4 * i] * w[4 * i + 1] * consts[0] + w[4 * i + 2] * consts[1]));
      return true;
    case VX_GATE_POSEIDON: {
      const E swap = w[24];
      out.push_back(swap * (swap - E(1)));
      for (int i = 0; i < 4; ++i) out.push_back(swap * (w[i + 4] - w[i]) - w[25 + i]);
      E st[12];
      for (int i = 0; i < 4; ++i) st[i] = w[i] + w[25 + i], st[i + 4] = w[i + 4] - w[25 + i];
      for (int i = 8; i < 12; ++i) st[i] = w[i];
      int round = 0;
      auto constants = [&] {
        for (int i = 0; i < 12; ++i) st[i] = st[i] + E(vxh::RC[12 * round + i]);
      };
      for (int r = 0; r < 4; ++r, ++round) {
        constants();
        if (r)
          for (int i = 0; i < 12; ++i) out.push_back(st[i] - w[29 + 12 * (r - 1) + i]), st[i] = w[29 + 12 * (r - 1) + i];
        for (int i = 0; i < 12; ++i) st[i] = sbox(st[i]);
        mds(st);
      }
      for (int r = 0; r < 22; ++r, ++round) {
        constants();
        out.push_back(st[0] - w[65 + r]);
        st[0] = sbox(w[65 + r]);
        mds(st);
      }
      for (int r = 0; r < 4; ++r, ++round) {
        constants();
        for (int i = 0; i < 12; ++i) out.push_back(st[i] - w[87 + 12 * r + i]), st[i] = w[87 + 12 * r + i];
        for (int i = 0; i < 12; ++i) st[i] = sbox(st[i]);
        mds(st);
      }
      for (int i = 0; i < 12; ++i) out.push_back(st[i] - w[12 + i]);
      return true;
    }
    case VX_GATE_PROGRAM: {
      if (!g.program) return false;
      E r[64];
      for (size_t pc = 0;; ++pc) {
        const uint64_t ins = g.program[pc];
        const int op = (int)(ins & 0xFF), dst = (int)((ins >> 8) & 63), a = (int)((ins >> 16) & 0xFFFF), b = (int)((ins >> 32) & 0xFFFF);
        switch (op) {
          case VX_OP_END: return true;
          case VX_OP_LDW: r[dst] = w[a]; break;
          case VX_OP_LDC: r[dst] = consts[a]; break;
          case VX_OP_LDI: r[dst] = E(vxh::canon(g.program[++pc])); break;
          case VX_OP_ADD: r[dst] = r[a & 63] + r[b & 63]; break;
          case VX_OP_SUB: r[dst] = r[a & 63] - r[b & 63]; break;
          case VX_OP_MUL: r[dst] = r[a & 63] * r[b & 63]; break;
          case VX_OP_PUSH: out.push_back(r[a & 63]); break;
          case VX_OP_LDP: r[dst] = E(pih.e[a & 3]); break;
          default: return false;
        }
      }
    }
  }
  return false;
}

// vanishing_poly.rs::check_lookup_constraints at an extension point for ONE challenge: lookup selectors sel[0..4 + num_luts)
// = TransSre, TransLdc, InitSre, LastLdc, one "ends" selector per table; zs / zs_next = [RE, SLDC_0 ..]; d = [a, b, alpha, delta].
static void lookup_terms(const CircuitV& c, const E* sel, const E* w, const E* zs, const E* zs_next, const u64* d, std::vector<E>& out) {
  const int lu_slots = c.num_routed / 2, lut_slots = c.num_routed / 3, lu_deg = c.qdf - 1, nsl = c.num_sldc(), lut_deg = (lut_slots + nsl - 1) / nsl;
  const E* zx = zs + 1;
  const E* zgx = zs_next + 1;
  std::vector<E> looking(lu_slots), looked(lut_slots), looked_re(lut_slots);
  for (int i = 0; i < lu_slots; ++i) looking[i] = w[2 * i] + scale(w[2 * i + 1], d[0]);
  for (int i = 0; i < lut_slots; ++i) looked[i] = w[3 * i] + scale(w[3 * i + 1], d[0]), looked_re[i] = w[3 * i] + scale(w[3 * i + 1], d[1]);
  out.push_back(sel[3] * zx[nsl - 1]);
  out.push_back(sel[2] * zx[0]);
  out.push_back(sel[2] * zs[0]);
  size_t off = 0;
  for (int t = 0; t < c.num_luts; ++t) {  // get_lut_poly: sum_k (inp_k + b out_k) delta^(degree - 1 - k), zero-padded to whole rows
    const size_t len = (size_t)c.lut_lens[t], degree = (len + lut_slots - 1) / lut_slots * lut_slots;
    u64 acc = 0;
    for (size_t k = 0; k < degree; ++k) {
      const u64 coeff = k < len ? vxh::add((u64)c.lut_inputs[off + k], vxh::mul(d[1], (u64)c.lut_outputs[off + k])) : 0;
      acc = vxh::add(vxh::mul(acc, d[3]), coeff);
    }
    off += len;
    out.push_back(sel[4 + t] * (zs[0] - E(acc)));
  }
  {
    E cur = zs_next[0];
    for (int i = 0; i < lut_slots; ++i) cur = scale(cur, d[3]) + looked_re[i];
    out.push_back(sel[0] * (zs[0] - cur));
  }
  const E alpha(d[2]);
  for (int p = 0; p < nsl; ++p) {
    const int t0 = p * lut_deg, t1 = std::min((p + 1) * lut_deg, lut_slots), u0 = p * lu_deg, u1 = std::min((p + 1) * lu_deg, lu_slots);
    auto prod_except = [&](const std::vector<E>& v, int lo, int hi, int skip) {
      E acc(1);
      for (int j = lo; j < hi; ++j)
        if (j != skip) acc = acc * (alpha - v[j]);
      return acc;
    };
    E lu_sum, lut_sum;
    for (int i = u0; i < u1; ++i) lu_sum = lu_sum + prod_except(looking, u0, u1, i);
    for (int i = t0; i < t1; ++i) lut_sum = lut_sum + w[3 * i + 2] * prod_except(looked, t0, t1, i);
    const E prev = p == 0 ? zgx[nsl - 1] : zx[p - 1];
    out.push_back(sel[0] * (prod_except(looked, t0, t1, -1) * (zx[p] - prev) - lut_sum));
    out.push_back(sel[1] * (prod_except(looking, u0, u1, -1) * (zx[p] - prev) + lu_sum));
  }
}

// Returns "" when the proof is valid, else the reason.
static std::string verify(const CircuitV& c, const uint8_t* bytes, size_t len) {
  const int lg = c.degree_bits, rb = c.rate_bits, LG = lg + rb, nch = c.num_challenges, npp = c.npp(), qdf = c.qdf;
  const size_t n = (size_t)1 << lg, N = (size_t)1 << LG, cap_len = (size_t)1 << c.cap_height, R = c.arity_bits.size();
  const int nlp = c.num_lookup_polys();
  const size_t zs_pp = (size_t)nch * (1 + npp);
  const size_t widths[4] = {(size_t)c.num_constants + c.num_routed, (size_t)c.num_wires, zs_pp + (size_t)nch * nlp, (size_t)nch * qdf};
  Reader r{bytes, len};
  // ---- parse (read_proof_with_public_inputs; every length is implied by the circuit) ----
  std::vector<u64> wires_cap, zs_cap, quot_cap;
  r.words(wires_cap, 4 * cap_len);
  r.words(zs_cap, 4 * cap_len);
  r.words(quot_cap, 4 * cap_len);
  std::vector<E> o_const, o_sigma, o_wires, o_zs, o_zs_next, o_pp, o_quot, o_lzs, o_lzs_next;
  r.exts(o_const, c.num_constants);
  r.exts(o_sigma, c.num_routed);
  r.exts(o_wires, c.num_wires);
  r.exts(o_zs, nch);
  r.exts(o_zs_next, nch);
  r.exts(o_lzs, (size_t)nch * nlp);       // read_opening_set: the lookup openings sit between plonk_zs_next and the partial products
  r.exts(o_lzs_next, (size_t)nch * nlp);
  r.exts(o_pp, (size_t)nch * npp);
  r.exts(o_quot, (size_t)nch * qdf);
  std::vector<std::vector<u64>> commit_caps(R);
  for (auto& cp : commit_caps) r.words(cp, 4 * cap_len);
  struct Query {
    std::vector<u64> leaf[4], path[4];
    std::vector<std::vector<E>> step_evals;
    std::vector<std::vector<u64>> step_path;
  };
  std::vector<Query> queries(c.num_queries);
  for (auto& q : queries) {
    if (!r.ok) break;
    for (int t = 0; t < 4; ++t) {
      r.words(q.leaf[t], widths[t]);
      r.words(q.path[t], 4 * (size_t)r.u8());
    }
    q.step_evals.resize(R);
    q.step_path.resize(R);
    for (size_t k = 0; k < R; ++k) {
      r.exts(q.step_evals[k], (size_t)1 << c.arity_bits[k]);
      r.words(q.step_path[k], 4 * (size_t)r.u8());
    }
  }
  size_t final_len = n;
  for (int ab : c.arity_bits) final_len >>= ab;
  std::vector<E> final_poly;
  r.exts(final_poly, final_len);
  const u64 pow_witness = r.f();
  std::vector<u64> public_inputs;
  r.words(public_inputs, c.num_public_inputs);
  if (!r.ok || r.pos != len) return "malformed proof (length or non-canonical field element)";

  // ---- challenges (plonk/get_challenges.rs) ----
  const vxh::Hash4 pih = vxh::hash_no_pad(public_inputs.data(), public_inputs.size());
  vxh::Challenger ch;
  ch.observe_elements(c.digest.e, 4);
  ch.observe_elements(pih.e, 4);
  ch.observe_elements(wires_cap.data(), wires_cap.size());
  std::vector<u64> betas(nch), gammas(nch), alphas(nch);
  for (auto& v : betas) v = ch.get_challenge();
  for (auto& v : gammas) v = ch.get_challenge();
  std::vector<u64> deltas;  // lookup challenges: 4 per challenge; the first 2 nch of them are the betas and gammas
  if (c.num_luts > 0) {
    deltas = betas;
    deltas.insert(deltas.end(), gammas.begin(), gammas.end());
    for (int i = 0; i < 2 * nch; ++i) deltas.push_back(ch.get_challenge());
  }
  ch.observe_elements(zs_cap.data(), zs_cap.size());
  for (auto& v : alphas) v = ch.get_challenge();
  ch.observe_elements(quot_cap.data(), quot_cap.size());
  const E zeta = ch.get_extension_challenge();
  for (auto* v : {&o_const, &o_sigma, &o_wires, &o_zs, &o_pp, &o_quot, &o_lzs})
    for (E e : *v) ch.observe_ext(e.x());
  for (auto* v : {&o_zs_next, &o_lzs_next})
    for (E e : *v) ch.observe_ext(e.x());
  const E fri_alpha = ch.get_extension_challenge();
  std::vector<E> fri_betas;
  for (auto& cp : commit_caps) {
    ch.observe_elements(cp.data(), cp.size());
    fri_betas.push_back(ch.get_extension_challenge());
  }
  for (E e : final_poly) ch.observe_ext(e.x());
  ch.observe_element(pow_witness);
  const u64 pow_response = ch.get_challenge();
  std::vector<size_t> x_indices(c.num_queries);
  for (auto& x : x_indices) x = (size_t)(ch.get_challenge() % (u64)N);

  // ---- vanishing polynomial identity at zeta (vanishing_poly.rs::eval_vanishing_poly) ----
  {
    E zeta_n = zeta;
    for (int i = 0; i < lg; ++i) zeta_n = zeta_n * zeta_n;
    const E z_h = zeta_n - E(1);
    const E l0 = z_h * inv(scale(zeta - E(1), (u64)n % P));
    std::vector<E> terms;
    for (int k = 0; k < nch; ++k) terms.push_back(l0 * (o_zs[k] - E(1)));
    for (int k = 0; k < nch; ++k) {
      std::vector<E> accs;
      accs.push_back(o_zs[k]);
      for (int j = 0; j < npp; ++j) accs.push_back(o_pp[(size_t)k * npp + j]);
      accs.push_back(o_zs_next[k]);
      int chunk = 0;
      for (int j0 = 0; j0 < c.num_routed; j0 += qdf, ++chunk) {
        E np(1), dp(1);
        for (int j = j0; j < std::min(c.num_routed, j0 + qdf); ++j) {
          np = np * (o_wires[j] + scale(scale(zeta, c.k_is[j]), betas[k]) + E(gammas[k]));
          dp = dp * (o_wires[j] + scale(o_sigma[j], betas[k]) + E(gammas[k]));
        }
        terms.push_back(accs[chunk] * np - accs[chunk + 1] * dp);
      }
    }
    for (int k = 0; k < nch && c.num_luts > 0; ++k)
      lookup_terms(c, o_const.data() + c.num_selectors, o_wires.data(), o_lzs.data() + (size_t)k * nlp, o_lzs_next.data() + (size_t)k * nlp,
                   deltas.data() + 4 * k, terms);
    std::vector<E> gate_terms, tmp;
    for (size_t gi = 0; gi < c.gates.size(); ++gi) {
      const GateV& g = c.gates[gi];
      const E s = o_const[g.selector_index];
      E filter(1);
      for (int q = g.group_start; q < g.group_end; ++q)
        if (q != (int)gi) filter = filter * (E((u64)q) - s);
      if (c.num_selectors > 1) filter = filter * (E(0xFFFFFFFFULL) - s);
      if (!eval_gate(g, o_const.data() + c.gate_const_base(), o_wires.data(), pih, tmp)) return "a gate could not be evaluated";
      if (tmp.size() > gate_terms.size()) gate_terms.resize(tmp.size());
      for (size_t i = 0; i < tmp.size(); ++i) gate_terms[i] = gate_terms[i] + filter * tmp[i];
    }
    terms.insert(terms.end(), gate_terms.begin(), gate_terms.end());
    for (int k = 0; k < nch; ++k) {
      E lhs;
      for (size_t i = terms.size(); i-- > 0;) lhs = lhs * E(alphas[k]) + terms[i];
      E q;
      for (int j = qdf; j-- > 0;) q = q * zeta_n + o_quot[(size_t)k * qdf + j];
      if (lhs != z_h * q) return "vanishing polynomial identity fails at zeta (challenge " + std::to_string(k) + ")";
    }
  }
  // ---- FRI (fri/verifier.rs) ----
  if (c.pow_bits > 0 && (pow_response >> (64 - c.pow_bits)) != 0) return "proof of work check failed";
  const E points[2] = {zeta, scale(zeta, vxh::root_of_unity(lg))};
  E reduced[2];
  {
    std::vector<E> b0;
    for (auto* v : {&o_const, &o_sigma, &o_wires, &o_zs, &o_pp, &o_quot, &o_lzs}) b0.insert(b0.end(), v->begin(), v->end());
    for (size_t i = b0.size(); i-- > 0;) reduced[0] = reduced[0] * fri_alpha + b0[i];
    std::vector<E> b1(o_zs_next);
    b1.insert(b1.end(), o_lzs_next.begin(), o_lzs_next.end());
    for (size_t i = b1.size(); i-- > 0;) reduced[1] = reduced[1] * fri_alpha + b1[i];
  }
  const u64* caps[4] = {c.cs_cap, wires_cap.data(), zs_cap.data(), quot_cap.data()};
  const u64 wN = vxh::root_of_unity(LG);
  for (int qi = 0; qi < c.num_queries; ++qi) {
    const Query& q = queries[qi];
    size_t xi = x_indices[qi];
    for (int t = 0; t < 4; ++t) {
      if (q.path[t].size() != 4 * (size_t)(LG - c.cap_height)) return "initial Merkle proof has the wrong length";  // validate_fri_proof_shape
      if (!merkle_ok(q.leaf[t].data(), widths[t], xi, caps[t], cap_len, q.path[t])) return "initial Merkle proof fails (oracle " + std::to_string(t) + ")";
    }
    u64 sx = vxh::mul(7, vxh::pow(wN, vxh::reverse_bits(xi, LG)));
    // fri_combine_initial
    E sum;
    for (int b = 0; b < 2; ++b) {
      std::vector<u64> ev;
      const std::vector<u64>& zl = q.leaf[2];  // [zs, partial products | lookup polys]: the lookup tail is opened AFTER the quotient
      if (b == 0) {
        ev.insert(ev.end(), q.leaf[0].begin(), q.leaf[0].end());
        ev.insert(ev.end(), q.leaf[1].begin(), q.leaf[1].end());
        ev.insert(ev.end(), zl.begin(), zl.begin() + zs_pp);
        ev.insert(ev.end(), q.leaf[3].begin(), q.leaf[3].end());
        ev.insert(ev.end(), zl.begin() + zs_pp, zl.end());
      } else {
        ev.assign(zl.begin(), zl.begin() + nch);
        ev.insert(ev.end(), zl.begin() + zs_pp, zl.end());
      }
      E red;
      for (size_t i = ev.size(); i-- > 0;) red = red * fri_alpha + E(ev[i]);
      sum = sum * E(vxh::epow(fri_alpha.x(), ev.size())) + (red - reduced[b]) * inv(E(sx) - points[b]);
    }
    E old_eval = sum;
    for (size_t k = 0; k < R; ++k) {
      const int ab = c.arity_bits[k];
      const size_t arity = (size_t)1 << ab, coset = xi >> ab, within = xi & (arity - 1);
      const std::vector<E>& evals = q.step_evals[k];
      if (evals[within] != old_eval) return "FRI consistency check fails at round " + std::to_string(k);
      // compute_evaluation: interpolate the coset's values (bit-reversed order) at beta
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
          acc = acc + ys[i] * num * inv(den);
        }
        old_eval = acc;
      }
      {
        int depth = 0, layer_bits = LG;
        for (size_t j = 0; j <= k; ++j) layer_bits -= c.arity_bits[j];
        depth = layer_bits > c.cap_height ? layer_bits - c.cap_height : 0;
        if (q.step_path[k].size() != 4 * (size_t)depth) return "FRI commit-phase Merkle proof has the wrong length";
      }
      std::vector<u64> flat(2 * arity);
      for (size_t j = 0; j < arity; ++j) flat[2 * j] = evals[j].a, flat[2 * j + 1] = evals[j].b;
      if (!merkle_ok(flat.data(), flat.size(), coset, commit_caps[k].data(), cap_len, q.step_path[k]))
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
}  // namespace vxv
