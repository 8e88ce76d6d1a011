// `.vxcircuit` — the self-describing on-disk form of a vx_circuit_desc (+ optionally the preprocessed polynomial values and
// the constants_sigmas cap), host code only.
//
// Role in the reference: plonky2x's `build` writes the compiled circuits to ./build/*.circuit and `prove` loads them
// (/root/reference/succinct.json:7-8,17-18; `circuit.test_serializers(..)` round-trips that format at
// /root/reference/circuits/header_range.rs:117-126, circuits/rotate.rs:152-161).  plonky2's own CircuitData byte format
// (util/serialization + gate/generator registries) needs the Rust serializers and cannot be restated here; this is the
// library's OWN container for the same content, written by the Rust side from its CommonCircuitData / ProverOnlyCircuitData
// through vx_circuit_serialize.  Layout, all little-endian, every section 8-byte aligned:
//
//   magic "VXCIRCT1" | u32 version = 1 | u32 flags (bit 0: preprocessed values present, bit 1: constants_sigmas cap present)
//   i32 x 20: degree_bits, num_wires, num_routed_wires, num_challenges, rate_bits, cap_height, pow_bits, num_query_rounds,
//             quotient_degree_factor, num_gates, num_selectors, num_constants, num_public_inputs, programs_len,
//             override_flags, hiding, num_fri_reduction_arity_bits, num_partial_products, num_luts, num_lookup_selectors
//   u64 x 4 : circuit_digest (meaningful iff VX_DESC_HAS_CIRCUIT_DIGEST)
//   i32[num_gates] x 6: gate_types, gate_params, selector_indices, group_starts, group_ends, program_offsets
//   u64[num_routed_wires] k_is | u32[num_public_inputs] pi_rows | u32[num_public_inputs] pi_cols
//   u64[programs_len] programs | i32[num_fri_reduction_arity_bits] fri_reduction_arity_bits
//   i32[num_luts] lut_lens | i32[3 num_luts] lookup_rows | u16[sum lut_lens] lut_inputs | u16[sum lut_lens] lut_outputs
//   u64[4 << cap_height] constants_sigmas cap            (flag bit 1)
//   u64[(num_constants + num_routed_wires) << degree_bits] constants_sigmas values on H   (flag bit 0)
//   u64 FNV-1a-64 of every preceding byte
#pragma once
#include <cstring>
#include <string>
#include <vector>
#include "../../include/vxprover.h"

namespace vxio {

static const char MAGIC[8] = {'V', 'X', 'C', 'I', 'R', 'C', 'T', '1'};
static inline uint64_t fnv1a(const uint8_t* p, size_t n) {
  uint64_t h = 1469598103934665603ULL;
  for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 1099511628211ULL;
  return h;
}
static inline size_t pad8(size_t n) { return (n + 7) & ~(size_t)7; }

struct Sizes {
  size_t gates_i32, k_is, pi, programs, arities, lut_lens, lookup_rows, lut_u16, lut_total, cap, values, total;
};
static const int NHDR = 20;
static inline Sizes sizes_of(const vx_circuit_desc* d, bool with_cap, bool with_values) {
  Sizes s;
  s.gates_i32 = pad8((size_t)d->num_gates * 4);
  s.k_is = (size_t)d->num_routed_wires * 8;
  s.pi = pad8((size_t)d->num_public_inputs * 4);
  s.programs = (size_t)d->programs_len * 8;
  s.arities = pad8((size_t)((d->override_flags & VX_DESC_HAS_FRI_ARITIES) ? d->num_fri_reduction_arity_bits : 0) * 4);
  s.lut_total = 0;
  for (int t = 0; t < d->num_luts; ++t) s.lut_total += (size_t)d->lut_lens[t];
  s.lut_lens = pad8((size_t)d->num_luts * 4);
  s.lookup_rows = pad8((size_t)d->num_luts * 12);
  s.lut_u16 = pad8(s.lut_total * 2);
  s.cap = with_cap ? ((size_t)32 << d->cap_height) : 0;
  s.values = with_values ? (((size_t)d->num_constants + d->num_routed_wires) << d->degree_bits) * 8 : 0;
  s.total = 8 + 8 + NHDR * 4 + 32 + 6 * s.gates_i32 + s.k_is + 2 * s.pi + s.programs + s.arities + s.lut_lens + s.lookup_rows + 2 * s.lut_u16 +
            s.cap + s.values + 8;
  return s;
}

// The description must already have passed desc_check (the callers do that).
static inline void serialize(const vx_circuit_desc* d, const uint64_t* cap, bool with_values, uint8_t* out) {
  const Sizes S = sizes_of(d, cap != nullptr, with_values);
  uint8_t* p = out;
  auto put = [&](const void* src, size_t n, size_t padded) {
    if (n) memcpy(p, src, n);
    if (padded > n) memset(p + n, 0, padded - n);
    p += padded;
  };
  put(MAGIC, 8, 8);
  const uint32_t hdr[2] = {1u, (with_values ? 1u : 0u) | (cap ? 2u : 0u)};
  put(hdr, 8, 8);
  const int32_t f[NHDR] = {d->degree_bits, d->num_wires, d->num_routed_wires, d->num_challenges, d->rate_bits, d->cap_height, d->pow_bits,
                         d->num_query_rounds, d->quotient_degree_factor, d->num_gates, d->num_selectors, d->num_constants,
                         d->num_public_inputs, d->programs_len, (int32_t)d->override_flags, d->hiding,
                         (d->override_flags & VX_DESC_HAS_FRI_ARITIES) ? d->num_fri_reduction_arity_bits : 0, d->num_partial_products,
                         d->num_luts, d->num_luts > 0 ? d->num_lookup_selectors : 0};
  put(f, sizeof f, sizeof f);
  put(d->circuit_digest, 32, 32);
  const size_t g4 = (size_t)d->num_gates * 4;
  put(d->gate_types, g4, S.gates_i32);
  put(d->gate_params, g4, S.gates_i32);
  put(d->selector_indices, g4, S.gates_i32);
  put(d->group_starts, g4, S.gates_i32);
  put(d->group_ends, g4, S.gates_i32);
  if (d->program_offsets) put(d->program_offsets, g4, S.gates_i32);
  else {
    std::vector<int32_t> none((size_t)d->num_gates, -1);
    put(none.data(), g4, S.gates_i32);
  }
  put(d->k_is, S.k_is, S.k_is);
  put(d->pi_rows, (size_t)d->num_public_inputs * 4, S.pi);
  put(d->pi_cols, (size_t)d->num_public_inputs * 4, S.pi);
  put(d->programs, S.programs, S.programs);
  if (S.arities) put(d->fri_reduction_arity_bits, (size_t)d->num_fri_reduction_arity_bits * 4, S.arities);
  if (d->num_luts > 0) {
    put(d->lut_lens, (size_t)d->num_luts * 4, S.lut_lens);
    put(d->lookup_rows, (size_t)d->num_luts * 12, S.lookup_rows);
    put(d->lut_inputs, S.lut_total * 2, S.lut_u16);
    put(d->lut_outputs, S.lut_total * 2, S.lut_u16);
  }
  if (cap) put(cap, S.cap, S.cap);
  if (with_values) put(d->constants_sigmas, S.values, S.values);
  const uint64_t h = fnv1a(out, (size_t)(p - out));
  put(&h, 8, 8);
}

// A parsed file: the description's small arrays are owned copies; `constants_sigmas` BORROWS the caller's buffer when that
// is 8-byte aligned (a 2^21-row circuit carries 1.4 GB of values), else it is copied too.
struct Parsed {
  vx_circuit_desc desc;
  std::vector<int32_t> i32s;   // six gate arrays + arities
  std::vector<uint64_t> u64s;  // k_is + programs + cap (+ values when copied)
  std::vector<uint32_t> u32s;  // pi_rows + pi_cols
  std::vector<int32_t> lut_i32;   // lut_lens + lookup_rows
  std::vector<uint16_t> lut_u16;  // lut_inputs + lut_outputs
  const uint64_t* cap = nullptr;
};

// Returns "" on success.  Only structure is checked here (sizes, checksum); the field ranges are desc_check's job.
static inline std::string parse(const uint8_t* b, size_t len, Parsed* out) {
  if (len < 8 + 8 + (size_t)NHDR * 4 + 32 + 8) return "file too short";
  if (memcmp(b, MAGIC, 8) != 0) return "bad magic (not a .vxcircuit file)";
  uint32_t hdr[2];
  memcpy(hdr, b + 8, 8);
  if (hdr[0] != 1) return "unsupported .vxcircuit version " + std::to_string(hdr[0]);
  if (hdr[1] & ~3u) return "unknown flags";
  int32_t f[NHDR];
  memcpy(f, b + 16, sizeof f);
  vx_circuit_desc& d = out->desc;
  memset(&d, 0, sizeof d);
  d.degree_bits = f[0], d.num_wires = f[1], d.num_routed_wires = f[2], d.num_challenges = f[3], d.rate_bits = f[4], d.cap_height = f[5];
  d.pow_bits = f[6], d.num_query_rounds = f[7], d.quotient_degree_factor = f[8], d.num_gates = f[9], d.num_selectors = f[10];
  d.num_constants = f[11], d.num_public_inputs = f[12], d.programs_len = f[13], d.override_flags = (uint32_t)f[14], d.hiding = f[15];
  d.num_fri_reduction_arity_bits = f[16], d.num_partial_products = f[17], d.num_luts = f[18], d.num_lookup_selectors = f[19];
  // bound every count BEFORE it sizes anything
  if (d.degree_bits < 0 || d.degree_bits > 40 || d.num_gates < 0 || d.num_gates > 4096 || d.num_routed_wires < 0 || d.num_routed_wires > 4096 ||
      d.num_constants < 0 || d.num_constants > 4096 || d.num_public_inputs < 0 || d.num_public_inputs > (1 << 20) || d.programs_len < 0 ||
      d.programs_len > (1 << 24) || d.num_fri_reduction_arity_bits < 0 || d.num_fri_reduction_arity_bits > 64 || d.cap_height < 0 || d.cap_height > 40 ||
      d.num_luts < 0 || d.num_luts > 64)
    return "implausible counts in the header";
  if (!(d.override_flags & VX_DESC_HAS_FRI_ARITIES) && d.num_fri_reduction_arity_bits != 0) return "arity list without its flag";
  const bool with_values = hdr[1] & 1, with_cap = hdr[1] & 2;
  if (d.num_luts > 0) {
    // the table lengths size later sections: read them first (their own position depends only on the header)
    d.lut_lens = nullptr;
    const int keep = d.num_luts;
    d.num_luts = 0;
    const Sizes S0 = sizes_of(&d, false, false);
    d.num_luts = keep;
    const size_t off = S0.total - 8;  // everything before the lookup sections of a lookup-free file, minus the checksum
    if (off + pad8((size_t)keep * 4) > len) return "file too short for its lookup tables";
    out->lut_i32.resize((size_t)keep * 4);
    memcpy(out->lut_i32.data(), b + off, (size_t)keep * 4);
    for (int t = 0; t < keep; ++t)
      if (out->lut_i32[t] < 0 || out->lut_i32[t] > (1 << 20)) return "implausible lookup table length";
    d.lut_lens = out->lut_i32.data();
  }
  const Sizes S = sizes_of(&d, with_cap, with_values);
  if (S.total != len) return "length " + std::to_string(len) + " does not match the header (expected " + std::to_string(S.total) + ")";
  uint64_t want;
  memcpy(&want, b + len - 8, 8);
  if (fnv1a(b, len - 8) != want) return "checksum mismatch (corrupt file)";
  const uint8_t* p = b + 16 + sizeof f;
  memcpy(d.circuit_digest, p, 32);
  p += 32;
  const size_t ng = (size_t)d.num_gates, na = (size_t)d.num_fri_reduction_arity_bits;
  out->i32s.resize(6 * ng + na);
  for (int k = 0; k < 6; ++k) {
    if (ng) memcpy(out->i32s.data() + k * ng, p, ng * 4);
    p += S.gates_i32;
  }
  const size_t cap_words = with_cap ? ((size_t)4 << d.cap_height) : 0;
  const bool borrow = with_values && ((uintptr_t)b % 8 == 0);
  out->u64s.resize((size_t)d.num_routed_wires + (size_t)d.programs_len + cap_words + (with_values && !borrow ? S.values / 8 : 0));
  uint64_t* u = out->u64s.data();
  if (S.k_is) memcpy(u, p, S.k_is);
  d.k_is = u;
  u += d.num_routed_wires;
  p += S.k_is;
  out->u32s.resize(2 * (size_t)d.num_public_inputs);
  if (d.num_public_inputs) {
    memcpy(out->u32s.data(), p, (size_t)d.num_public_inputs * 4);
    memcpy(out->u32s.data() + d.num_public_inputs, p + S.pi, (size_t)d.num_public_inputs * 4);
  }
  p += 2 * S.pi;
  d.pi_rows = out->u32s.data();
  d.pi_cols = out->u32s.data() + d.num_public_inputs;
  if (S.programs) memcpy(u, p, S.programs);
  d.programs = d.programs_len ? u : nullptr;
  u += d.programs_len;
  p += S.programs;
  if (na) memcpy(out->i32s.data() + 6 * ng, p, na * 4);
  p += S.arities;
  if (d.num_luts > 0) {
    const size_t nl = (size_t)d.num_luts;
    memcpy(out->lut_i32.data(), p, nl * 4);
    p += S.lut_lens;
    memcpy(out->lut_i32.data() + nl, p, nl * 12);
    p += S.lookup_rows;
    out->lut_u16.resize(2 * S.lut_total);
    if (S.lut_total) memcpy(out->lut_u16.data(), p, S.lut_total * 2);
    p += S.lut_u16;
    if (S.lut_total) memcpy(out->lut_u16.data() + S.lut_total, p, S.lut_total * 2);
    p += S.lut_u16;
    d.lut_lens = out->lut_i32.data();
    d.lookup_rows = out->lut_i32.data() + nl;
    d.lut_inputs = out->lut_u16.data();
    d.lut_outputs = out->lut_u16.data() + S.lut_total;
  }
  if (with_cap) {
    memcpy(u, p, S.cap);
    out->cap = u;
    u += cap_words;
    p += S.cap;
  }
  if (with_values) {
    if (borrow) d.constants_sigmas = reinterpret_cast<const uint64_t*>(p);
    else {
      memcpy(u, p, S.values);
      d.constants_sigmas = u;
    }
    p += S.values;
  }
  int32_t* a = out->i32s.data();
  d.gate_types = a, d.gate_params = a + ng, d.selector_indices = a + 2 * ng, d.group_starts = a + 3 * ng, d.group_ends = a + 4 * ng;
  d.program_offsets = a + 5 * ng;
  d.fri_reduction_arity_bits = na ? a + 6 * ng : nullptr;
  return std::string();
}

}  // namespace vxio
