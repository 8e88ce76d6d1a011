// Host-side Fiat-Shamir transcript of the product library.
// Mirrors plonky2::iop::challenger::Challenger<GoldilocksField, PoseidonHash> (plonky2 v0.2.0
// plonky2/src/iop/challenger.rs): duplex sponge in overwrite mode, challenges popped from
// the END of the 8-element output buffer.  Handfuls of elements per proof — stays on the CPU.
#pragma once
#include <vector>
#include "host_field.h"

namespace vxh {

struct Hash4 {
  u64 e[4];
};

// hashing.rs::hash_n_to_hash_no_pad
static inline Hash4 hash_no_pad(const u64* in, size_t n) {
  u64 s[12] = {0};
  for (size_t off = 0; off < n; off += 8) {
    size_t len = n - off < 8 ? n - off : 8;
    for (size_t i = 0; i < len; ++i) s[i] = in[off + i];
    poseidon(s);
  }
  Hash4 h;
  for (int i = 0; i < 4; ++i) h.e[i] = s[i];
  return h;
}

struct Challenger {
  u64 sponge[12] = {0};
  std::vector<u64> input, output;
  void duplexing() {
    for (size_t i = 0; i < input.size(); ++i) sponge[i] = input[i];
    input.clear();
    poseidon(sponge);
    output.assign(sponge, sponge + 8);
  }
  void observe_element(u64 e) {
    output.clear();
    input.push_back(e);
    if (input.size() == 8) duplexing();
  }
  void observe_elements(const u64* e, size_t n) {
    for (size_t i = 0; i < n; ++i) observe_element(e[i]);
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
    return Ext{a, b};
  }
};

}  // namespace vxh
