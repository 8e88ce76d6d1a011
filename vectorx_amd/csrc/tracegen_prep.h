// Host side of the chip-table trace generators: message padding, block splitting and the CHAIN of chaining values (a message's
// blocks depend on each other; 600 SHA-256 or 2240 BLAKE2b compressions take well under a millisecond on one host core, whereas
// on the device they would be one lane walking a dependent chain).  Everything per-row — 0.5 GB of cells for a map job's BLAKE2b
// table — is expanded on the GPU from the block descriptions built here (tracegen.hip.h).  Pure C++: also compiled into the
// test-only host build (tests/tracegen_host.cpp).  Semantics: vectorx_amd/{sha256,sha512,blake2b_bytes}_air.py generate_trace.
#pragma once
#include <string.h>
#include <vector>
#include "tracegen_consts.h"
#include "tracegen_core.h"

namespace tg {

enum { PREP_OK = 0, PREP_TOO_MANY_BLOCKS = 1, PREP_BAD_ARGS = 2 };

template <class T>
struct Sha2Consts;
template <>
struct Sha2Consts<Sha256T> {
  static TG_HD const uint32_t* K() { return K256; }
  static TG_HD const uint32_t* IV() { return IV256; }
};
template <>
struct Sha2Consts<Sha512T> {
  static TG_HD const uint64_t* K() { return K512; }
  static TG_HD const uint64_t* IV() { return IV512; }
};
template <>
struct Sha2Consts<Sha512BusT> : Sha2Consts<Sha512T> {};

template <class T>
struct Sha2Prep {
  std::vector<Sha2Block<T>> blocks;          // ceil(n / PERIOD) blocks: the messages' blocks, then the endless zero-message
  std::vector<typename T::W> digests;        // 8 words per message
  uint64_t pis[8 * T::LIMBS];                // D in the last row
};

// `msgs` + `off[nmsg + 1]`: the messages back to back.  Every message must complete inside the 2^degree_bits rows.
template <class T>
int sha2_prepare(int degree_bits, const uint8_t* msgs, const uint64_t* off, int nmsg, Sha2Prep<T>& out) {
  typedef typename T::W W;
  if (degree_bits < 4 || degree_bits > 30 || nmsg < 0 || (nmsg && (!msgs || !off))) return PREP_BAD_ARGS;
  const size_t n = (size_t)1 << degree_bits, nb = (n + T::PERIOD - 1) / T::PERIOD;
  const int BB = 16 * sizeof(W);               // block bytes: 64 / 128
  const int LB = 2 * sizeof(W);                // length field bytes: 8 / 16
  const W* K = Sha2Consts<T>::K();
  const W* IV = Sha2Consts<T>::IV();
  out.blocks.clear();
  out.blocks.reserve(nb);
  out.digests.assign((size_t)nmsg * 8, 0);
  W h[8], d[8];
  for (int k = 0; k < 8; ++k) h[k] = IV[k], d[k] = 0;
  std::vector<uint8_t> buf;
  bool first_of_message = true;                // bus variant: the next block starts a message
  uint32_t latched[16] = {0};
  auto latch = [&](Sha2Block<T>& b) {          // the first 64 bytes of the message = big-endian words 0..7 of its first block, as LE 32-bit words
    if (first_of_message)
      for (int j = 0; j < 16; ++j) {
        const uint64_t w = (uint64_t)b.w[(j * 4) / sizeof(W)];
        const int byte0 = (j * 4) % (int)sizeof(W);                 // first byte (most significant = 0) of the LE word inside the BE word
        uint32_t v = 0;
        for (int q = 0; q < 4; ++q) v |= (uint32_t)((w >> (8 * (sizeof(W) - 1 - (byte0 + q)))) & 0xFF) << (8 * q);
        latched[j] = v;
      }
    b.first = first_of_message;
    for (int j = 0; j < 16; ++j) b.lw[j] = latched[j];
  };
  for (int mi = 0; mi < nmsg; ++mi)
    if (off[mi + 1] < off[mi]) return PREP_BAD_ARGS;       // offsets must not decrease (a wrapped length would ask for exabytes)
  for (int mi = 0; mi < nmsg; ++mi) {
    const size_t ml = (size_t)(off[mi + 1] - off[mi]);
    const size_t total = ((ml + 1 + LB + BB - 1) / BB) * BB;
    buf.assign(total, 0);
    if (ml) memcpy(buf.data(), msgs + off[mi], ml);
    buf[ml] = 0x80;
    const uint64_t bits = (uint64_t)ml * 8;
    for (int i = 0; i < 8; ++i) buf[total - 1 - i] = (uint8_t)(bits >> (8 * i));
    const size_t nblk = total / BB;
    for (size_t j = 0; j < nblk; ++j) {
      if (out.blocks.size() >= nb) return PREP_TOO_MANY_BLOCKS;
      Sha2Block<T> b;
      memset(&b, 0, sizeof b);
      for (int i = 0; i < 16; ++i) {
        W w = 0;
        for (size_t q = 0; q < sizeof(W); ++q) w = (W)((w << 8) | buf[j * BB + i * sizeof(W) + q]);   // big-endian words
        b.w[i] = w;
      }
      for (int k = 0; k < 8; ++k) b.h_in[k] = h[k], b.d[k] = d[k];
      b.nf = j + 1 == nblk;
      latch(b);
      first_of_message = b.nf;
      sha2_compress<T>(h, b.w, K);
      // the block's hand-over row must exist for the digest to be latched (and a later block to start from the IV)
      if ((out.blocks.size() + 1) * T::PERIOD > n) return PREP_TOO_MANY_BLOCKS;
      out.blocks.push_back(b);
      if (b.nf) {
        for (int k = 0; k < 8; ++k) d[k] = h[k], out.digests[(size_t)mi * 8 + k] = h[k], h[k] = IV[k];
      }
    }
  }
  while (out.blocks.size() < nb) {              // the endless zero-message: never final
    Sha2Block<T> b;
    memset(&b, 0, sizeof b);
    for (int k = 0; k < 8; ++k) b.h_in[k] = h[k], b.d[k] = d[k];
    latch(b);
    first_of_message = false;                    // the endless zero-message never ends
    sha2_compress<T>(h, b.w, K);
    out.blocks.push_back(b);
  }
  const Sha2Block<T>& last = out.blocks[(n - 1) / T::PERIOD];
  for (int k = 0; k < 8; ++k) {
    if (T::LIMBS == 1) out.pis[k] = last.d[k];
    else out.pis[2 * k] = (uint64_t)last.d[k] & 0xFFFFFFFFu, out.pis[2 * k + 1] = (uint64_t)last.d[k] >> 32;
  }
  return PREP_OK;
}

struct B2Prep {
  std::vector<b2::Block> blocks;
  std::vector<uint64_t> digests;   // 4 words per message
  uint64_t pis[8];
};
inline int b2_prepare(int degree_bits, const uint8_t* msgs, const uint64_t* off, int nmsg, B2Prep& out) {
  if (degree_bits < 16 || degree_bits > 30 || nmsg < 0 || (nmsg && (!msgs || !off))) return PREP_BAD_ARGS;
  const size_t n = (size_t)1 << degree_bits, nb = (n + b2::PERIOD - 1) / b2::PERIOD;
  out.blocks.clear();
  out.blocks.reserve(nb);
  out.digests.assign((size_t)nmsg * 4, 0);
  uint64_t h[8], dl[4] = {0, 0, 0, 0};
  for (int k = 0; k < 8; ++k) h[k] = B2_IVP[k];
  uint64_t tb_prev = 0;
  int dsrc = -1;
  auto push = [&](b2::Block& b) {
    for (int k = 0; k < 8; ++k) b.h_in[k] = h[k];
    b.tb_prev = tb_prev;
    b.dsrc = dsrc;
    uint64_t hn[8];
    for (int k = 0; k < 8; ++k) hn[k] = h[k];
    b2::compress(hn, b, B2_IV, B2_SIGMA);
    const size_t idx = out.blocks.size();
    out.blocks.push_back(b);
    if ((idx + 1) * b2::PERIOD <= n) {          // the block completes inside the trace
      if (b.fin) {
        dsrc = (int)idx;
        for (int k = 0; k < 4; ++k) dl[k] = hn[k];
        for (int k = 0; k < 8; ++k) h[k] = B2_IVP[k];
        tb_prev = 0;
      } else {
        for (int k = 0; k < 8; ++k) h[k] = hn[k];
        tb_prev = b.t;
      }
      return true;
    }
    return false;
  };
  for (int mi = 0; mi < nmsg; ++mi)
    if (off[mi + 1] < off[mi]) return PREP_BAD_ARGS;       // offsets must not decrease
  for (int mi = 0; mi < nmsg; ++mi) {
    const size_t ml = (size_t)(off[mi + 1] - off[mi]);
    const size_t nblk = ml ? (ml + 127) / 128 : 1;
    for (size_t j = 0; j < nblk; ++j) {
      if (out.blocks.size() >= nb) return PREP_TOO_MANY_BLOCKS;
      b2::Block b;
      memset(&b, 0, sizeof b);
      uint8_t raw[128];
      memset(raw, 0, sizeof raw);
      const size_t have = ml > 128 * j ? (ml - 128 * j < 128 ? ml - 128 * j : 128) : 0;
      if (have) memcpy(raw, msgs + off[mi] + 128 * j, have);
      for (int i = 0; i < 16; ++i) {
        uint64_t w = 0;
        for (int q = 7; q >= 0; --q) w = (w << 8) | raw[8 * i + q];   // little-endian words
        b.m[i] = w;
      }
      const bool last = j + 1 == nblk;
      b.t = last ? (uint64_t)ml : (uint64_t)128 * (j + 1);
      b.fin = last;
      if (!push(b)) return PREP_TOO_MANY_BLOCKS;   // a message block that does not complete inside the trace
      if (last)
        for (int k = 0; k < 4; ++k) out.digests[(size_t)mi * 4 + k] = dl[k];
    }
  }
  while (out.blocks.size() < nb) {               // the endless zero-message
    b2::Block b;
    memset(&b, 0, sizeof b);
    b.t = tb_prev + 128;
    push(b);
  }
  for (int k = 0; k < 4; ++k) out.pis[2 * k] = dl[k] & 0xFFFFFFFFu, out.pis[2 * k + 1] = dl[k] >> 32;
  return PREP_OK;
}

}  // namespace tg
