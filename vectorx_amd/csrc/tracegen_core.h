// Trace generation for the chip tables, the part that is the same on the host and on the device.
//
// A header_range map job hashes 8 headers with BLAKE2b and builds two SHA-256 trees over them, the outer job chains SHA-256 over
// the authority set and hashes the signed messages with SHA-512 (/root/reference/circuits/builder/subchain_verification.rs:148-231,
// builder/header.rs:14-19, builder/justification.rs:140-156): in the reference the Curta chips' witness generation fills those
// tables on the CPU.  Here a table is filled ON THE GPU, one thread per trace row (tracegen.hip.h); this header holds the row
// writers — plain integer code, `TG_HD` = __host__ __device__ under hipcc — so that the same functions can be compiled for the
// host by tests/tracegen_host.cpp and compared cell by cell with the numpy generators of vectorx_amd/{sha256,sha512,
// blake2b_bytes}_air.py, whose column maps and row semantics they follow (cited per function).  The product only ever calls the
// device build.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#define TG_HD __host__ __device__ __forceinline__
#else
#define TG_HD inline
#endif

namespace tg {

// ================================================================================================================================
// SHA-2 (sha256_air.py: Cols, generate_trace; sha512_air.py: the same table on 64-bit words, 80 rounds, 32-bit limbs)
// ================================================================================================================================
struct Sha256T {
  typedef uint32_t W;
  static constexpr bool BUS = false;
  static constexpr int BITS = 32, ROUNDS = 64, PERIOD = 66, LIMBS = 1, NCARRY = 3;
  static constexpr int S = 0, WB = 256, X0 = 768, X1 = 800, M = 832, Y0 = 864, Y1 = 896, SEL = 928, H = 994, D = 1002, FFC = 1010,
                       NF = 1018, CA = 1019, CE = 1020, CW = 1021, TBL = 1022, MULT = 1023, N = 1024;
  static constexpr int R_S0a = 2, R_S0b = 13, R_S0c = 22, R_S1a = 6, R_S1b = 11, R_S1c = 25, R_s0a = 7, R_s0b = 18, SH_s0 = 3,
                       R_s1a = 17, R_s1b = 19, SH_s1 = 10;
};
struct Sha512T {
  typedef uint64_t W;
  static constexpr bool BUS = false;
  static constexpr int BITS = 64, ROUNDS = 80, PERIOD = 82, LIMBS = 2, NCARRY = 6;
  static constexpr int S = 0, WB = 512, X0 = 1536, X1 = 1600, M = 1664, Y0 = 1728, Y1 = 1792, SEL = 1856, H = 1938, D = 1954,
                       FFC = 1970, NF = 1986, CA = 1987, CE = 1989, CW = 1991, TBL = 1993, MULT = 1994, N = 1995;
  static constexpr int R_S0a = 28, R_S0b = 34, R_S0c = 39, R_S1a = 14, R_S1b = 18, R_S1c = 41, R_s0a = 1, R_s0b = 8, SH_s0 = 7,
                       R_s1a = 19, R_s1b = 61, SH_s1 = 6;
};

// the bus variant of the SHA-512 table (sha512_air.py, bus=True; vectorx_amd/sig_link_air.py): 17 more columns — the first 64 bytes of the
// current message as 16 little-endian 32-bit words (an Ed25519 R, then A), latched over the whole message, and the first-block flag
struct Sha512BusT : Sha512T {
  static constexpr bool BUS = true;
  static constexpr int LW = 1995, FIRST = 2011, N = 2012;
};

template <class T>
TG_HD typename T::W rotr(typename T::W x, int r) {
  return (typename T::W)((x >> r) | (x << (T::BITS - r)));
}

// one block of a (padded) message as the table sees it
template <class T>
struct Sha2Block {
  typename T::W w[16];
  typename T::W h_in[8];   // chaining value the block starts from (the IV after a message's last block)
  typename T::W d[8];      // digest of the last message completed BEFORE this block's hand-over row (zeros: none yet)
  uint32_t nf;             // hand-over row: the next block starts a new message (last block of a real message)
  uint32_t first;          // bus variant: this block is the first of its message
  uint32_t lw[16];         // bus variant: the first 64 bytes of the block's message as little-endian 32-bit words
};
// what a thread per block precomputes for the row writers: the message schedule and the working state before every row
template <class T>
struct Sha2Expanded {
  typename T::W sched[T::ROUNDS];
  typename T::W state[T::PERIOD][8];
};

template <class T>
TG_HD void sha2_expand(const Sha2Block<T>& b, const typename T::W* K, Sha2Expanded<T>& e) {
  typedef typename T::W W;
  for (int i = 0; i < 16; ++i) e.sched[i] = b.w[i];
  for (int t = 16; t < T::ROUNDS; ++t) {
    const W w2 = e.sched[t - 2], w15 = e.sched[t - 15];
    const W s1 = rotr<T>(w2, T::R_s1a) ^ rotr<T>(w2, T::R_s1b) ^ (w2 >> T::SH_s1);
    const W s0 = rotr<T>(w15, T::R_s0a) ^ rotr<T>(w15, T::R_s0b) ^ (w15 >> T::SH_s0);
    e.sched[t] = (W)(s1 + e.sched[t - 7] + s0 + e.sched[t - 16]);
  }
  W st[8];
  for (int k = 0; k < 8; ++k) st[k] = b.h_in[k];
  for (int r = 0; r < T::ROUNDS; ++r) {
    for (int k = 0; k < 8; ++k) e.state[r][k] = st[k];
    const W a = st[0], bb = st[1], c = st[2], d = st[3], ee = st[4], f = st[5], g = st[6], h = st[7];
    const W S1 = rotr<T>(ee, T::R_S1a) ^ rotr<T>(ee, T::R_S1b) ^ rotr<T>(ee, T::R_S1c);
    const W ch = (ee & f) ^ (~ee & g);
    const W t1 = (W)(h + S1 + ch + K[r] + e.sched[r]);
    const W S0 = rotr<T>(a, T::R_S0a) ^ rotr<T>(a, T::R_S0b) ^ rotr<T>(a, T::R_S0c);
    const W mj = (a & bb) ^ (a & c) ^ (bb & c);
    st[7] = g, st[6] = f, st[5] = ee, st[4] = (W)(d + t1), st[3] = c, st[2] = bb, st[1] = a, st[0] = (W)(t1 + S0 + mj);
  }
  for (int k = 0; k < 8; ++k) e.state[T::ROUNDS][k] = st[k];                               // row ROUNDS: the state after the last round
  for (int k = 0; k < 8; ++k) e.state[T::ROUNDS + 1][k] = (W)(b.h_in[k] + st[k]);          // row ROUNDS + 1: the new chaining value
}
// the chaining value after the block (host side of the chain)
template <class T>
TG_HD void sha2_compress(typename T::W (&h)[8], const typename T::W (&w)[16], const typename T::W* K) {
  typedef typename T::W W;
  W s[16];
  for (int i = 0; i < 16; ++i) s[i] = w[i];
  W st[8];
  for (int k = 0; k < 8; ++k) st[k] = h[k];
  for (int r = 0; r < T::ROUNDS; ++r) {
    if (r >= 16) {
      const W w2 = s[(r - 2) & 15], w15 = s[(r - 15) & 15];
      const W s1 = rotr<T>(w2, T::R_s1a) ^ rotr<T>(w2, T::R_s1b) ^ (w2 >> T::SH_s1);
      const W s0 = rotr<T>(w15, T::R_s0a) ^ rotr<T>(w15, T::R_s0b) ^ (w15 >> T::SH_s0);
      s[r & 15] = (W)(s1 + s[(r - 7) & 15] + s0 + s[r & 15]);
    }
    const W a = st[0], bb = st[1], c = st[2], d = st[3], ee = st[4], f = st[5], g = st[6], hh = st[7];
    const W S1 = rotr<T>(ee, T::R_S1a) ^ rotr<T>(ee, T::R_S1b) ^ rotr<T>(ee, T::R_S1c);
    const W ch = (ee & f) ^ (~ee & g);
    const W t1 = (W)(hh + S1 + ch + K[r] + s[r & 15]);
    const W S0 = rotr<T>(a, T::R_S0a) ^ rotr<T>(a, T::R_S0b) ^ rotr<T>(a, T::R_S0c);
    const W mj = (a & bb) ^ (a & c) ^ (bb & c);
    st[7] = g, st[6] = f, st[5] = ee, st[4] = (W)(d + t1), st[3] = c, st[2] = bb, st[1] = a, st[0] = (W)(t1 + S0 + mj);
  }
  for (int k = 0; k < 8; ++k) h[k] = (W)(h[k] + st[k]);
}

// sum of `n` words limb-wise: the carry out of the low 32-bit limbs and (64-bit words) out of the high limbs — the values the table
// range-checks (sha256_air.py: `full >> 32`; sha512_air.py: `lo >> 32`, `hi >> 32`)
template <class T>
TG_HD void sha2_carries(const typename T::W* terms, int n, unsigned& c0, unsigned& c1) {
  uint64_t lo = 0, hi = 0;
  for (int i = 0; i < n; ++i) {
    lo += (uint64_t)(terms[i] & 0xFFFFFFFFu);
    if (T::LIMBS == 2) hi += (uint64_t)((uint64_t)terms[i] >> 32);
  }
  c0 = (unsigned)(lo >> 32);
  c1 = T::LIMBS == 2 ? (unsigned)((hi + (lo >> 32)) >> 32) : 0;
}

// Row `r` of block `b` (the cells of ONE trace row), written through put(column, value).  `prev` = the expansion of the block before
// (nullptr for the first block: the schedule window starts from zeros).  Returns the row's carry values in `carries[NCARRY]`
// (the lookups whose multiplicities the caller counts).  MULT is written as 0 — the caller patches the first 8 rows afterwards.
template <class T, class Put>
TG_HD void sha2_row(const Sha2Block<T>& blk, const Sha2Expanded<T>& e, const Sha2Expanded<T>* prev, int r, size_t row,
                    const typename T::W* K, Put put, unsigned* carries) {
  typedef typename T::W W;
  for (int k = 0; k < T::PERIOD; ++k) put(T::SEL + k, (uint64_t)(k == r));
  for (int k = 0; k < 8; ++k) {
    if (T::LIMBS == 1) {
      put(T::H + k, (uint64_t)blk.h_in[k]);
      put(T::D + k, (uint64_t)blk.d[k]);
    } else {
      put(T::H + 2 * k, (uint64_t)(blk.h_in[k] & 0xFFFFFFFFu));
      put(T::H + 2 * k + 1, (uint64_t)((uint64_t)blk.h_in[k] >> 32));
      put(T::D + 2 * k, (uint64_t)(blk.d[k] & 0xFFFFFFFFu));
      put(T::D + 2 * k + 1, (uint64_t)((uint64_t)blk.d[k] >> 32));
    }
  }
  // the schedule window W_{t-k}: this block's schedule, zeros on the two closing rows, the previous block's rows below that
  W win[16];
  for (int k = 0; k < 16; ++k) {
    const int idx = r - k;
    W v = 0;
    if (idx >= 0) {
      if (idx < T::ROUNDS) v = e.sched[idx];
    } else if (prev != nullptr) {
      const int p = T::PERIOD + idx;
      if (p < T::ROUNDS) v = prev->sched[p];
    }
    win[k] = v;
  }
  W st[8];
  for (int k = 0; k < 8; ++k) st[k] = e.state[r][k];
  for (int k = 0; k < 16; ++k)
    for (int i = 0; i < T::BITS; ++i) put(T::WB + T::BITS * k + i, (uint64_t)((win[k] >> i) & 1));
  for (int k = 0; k < 8; ++k)
    for (int i = 0; i < T::BITS; ++i) put(T::S + T::BITS * k + i, (uint64_t)((st[k] >> i) & 1));
  const W a = st[0], bb = st[1], c = st[2], d = st[3], ee = st[4], f = st[5], g = st[6], h = st[7];
  const W w1 = win[1], w14 = win[14];
  const W x0 = rotr<T>(a, T::R_S0a) ^ rotr<T>(a, T::R_S0b), x1 = rotr<T>(ee, T::R_S1a) ^ rotr<T>(ee, T::R_S1b), mm = a & bb;
  const W y0 = rotr<T>(w14, T::R_s0a) ^ rotr<T>(w14, T::R_s0b), y1 = rotr<T>(w1, T::R_s1a) ^ rotr<T>(w1, T::R_s1b);
  for (int i = 0; i < T::BITS; ++i) {
    put(T::X0 + i, (uint64_t)((x0 >> i) & 1));
    put(T::X1 + i, (uint64_t)((x1 >> i) & 1));
    put(T::M + i, (uint64_t)((mm >> i) & 1));
    put(T::Y0 + i, (uint64_t)((y0 >> i) & 1));
    put(T::Y1 + i, (uint64_t)((y1 >> i) & 1));
  }
  unsigned ca0 = 0, ca1 = 0, ce0 = 0, ce1 = 0, cw0 = 0, cw1 = 0;
  uint64_t ffc[16];
  for (int k = 0; k < 16; ++k) ffc[k] = 0;
  uint64_t nf = 0;
  if (r < T::ROUNDS) {
    const W S1 = x1 ^ rotr<T>(ee, T::R_S1c);
    const W ch = (ee & f) ^ (~ee & g);
    const W S0 = x0 ^ rotr<T>(a, T::R_S0c);
    const W mj = (a & bb) ^ (a & c) ^ (bb & c);
    const W ta[7] = {h, S1, ch, K[r], e.sched[r], S0, mj};
    const W te[6] = {h, S1, ch, K[r], e.sched[r], d};
    sha2_carries<T>(ta, 7, ca0, ca1);
    sha2_carries<T>(te, 6, ce0, ce1);
    if (r >= 15 && r <= T::ROUNDS - 2) {
      const W s1 = y1 ^ (w1 >> T::SH_s1), s0 = y0 ^ (w14 >> T::SH_s0);
      const W tw[4] = {s1, win[6], s0, win[15]};
      sha2_carries<T>(tw, 4, cw0, cw1);
    }
  } else if (r == T::ROUNDS) {
    for (int k = 0; k < 8; ++k) {
      const W two[2] = {blk.h_in[k], st[k]};
      unsigned c0, c1;
      sha2_carries<T>(two, 2, c0, c1);
      if (T::LIMBS == 1) ffc[k] = c0;
      else ffc[2 * k] = c0, ffc[2 * k + 1] = c1;
    }
  } else {
    nf = blk.nf;
  }
  for (int k = 0; k < 8 * T::LIMBS; ++k) put(T::FFC + k, ffc[k]);
  put(T::NF, nf);
  if (T::LIMBS == 1) {
    put(T::CA, (uint64_t)ca0), put(T::CE, (uint64_t)ce0), put(T::CW, (uint64_t)cw0);
    carries[0] = ca0, carries[1] = ce0, carries[2] = cw0;
  } else {
    put(T::CA, (uint64_t)ca0), put(T::CA + 1, (uint64_t)ca1), put(T::CE, (uint64_t)ce0), put(T::CE + 1, (uint64_t)ce1);
    put(T::CW, (uint64_t)cw0), put(T::CW + 1, (uint64_t)cw1);
    carries[0] = ca0, carries[1] = ca1, carries[2] = ce0, carries[3] = ce1, carries[4] = cw0, carries[5] = cw1;
  }
  put(T::TBL, (uint64_t)(row & 7));
  put(T::MULT, (uint64_t)0);
  if constexpr (T::BUS) {
    for (int j = 0; j < 16; ++j) put(T::LW + j, (uint64_t)blk.lw[j]);
    put(T::FIRST, (uint64_t)blk.first);
  }
}

// ================================================================================================================================
// BLAKE2b-256 on bytes with a XOR lookup table (blake2b_bytes_air.py: Cols, _g, generate_trace)
// ================================================================================================================================
namespace b2 {
constexpr int PERIOD = 28, ROW_G0 = 1, ROW_GLAST = 24, ROW_FIN0 = 25, ROW_HAND = 27, NSLOT = 4, NFING = 4, NTAB = 2, TAB_ROWS = 32768;
constexpr int NFIELD = 12;   // BIN DIN A1 E1 C1 F1 A2 E2 C2 F2 TOP B2: 8 byte columns each
enum Field { BIN = 0, DIN, A1, E1, C1, F1, A2, E2, C2, F2, TOP, B2 };
constexpr int SEL = 0, H = SEL + PERIOD, HN = H + 16, D = HN + 16, M = D + 8, T = M + 32, F = M + 33, TB = M + 34, V = M + 35,
              SLOT = V + 32, SLOT_W = 8 * NFIELD + 4 + 8, FIN = SLOT + NSLOT * SLOT_W, BY = FIN + 40 * NFING, TAB = BY + 8,
              N = TAB + 20 * NTAB, NTUP = NSLOT * 40 + 16 * NFING + 8;
static_assert(N == 775, "blake2b_bytes_air.Cols.N");
TG_HD int fcol(int g, int field, int j) { return SLOT + g * SLOT_W + 8 * field + j; }
TG_HD int fincol(int g, int which, int j) { return FIN + 40 * g + 8 * which + j; }   // which: FA FB FE FH FO
TG_HD int tabcol(int k, int off) { return TAB + 20 * k + off; }                       // TA 0, TB 1, TC 2, BITS 3..18, MULT 19

struct Block {
  uint64_t m[16];
  uint64_t h_in[8];
  uint64_t t, tb_prev;
  uint32_t fin;
  int32_t dsrc;      // index of the block whose HN[0..4] is the digest latched in D on this block's rows (-1: none yet)
};
struct Expanded {
  uint64_t v[25][16];   // the work vector before G row s = 1..24 (index s - 1), after the last (index 24)
  uint64_t hn[8];       // the chaining value the block hands over
};
TG_HD uint64_t rotr64(uint64_t x, int r) { return (x >> r) | (x << (64 - r)); }
TG_HD void init_v(const Block& b, const uint64_t* IV, uint64_t* v) {
  for (int i = 0; i < 8; ++i) v[i] = b.h_in[i];
  for (int i = 0; i < 4; ++i) v[8 + i] = IV[i];
  v[12] = IV[4] ^ b.t, v[13] = IV[5], v[14] = IV[6] ^ (b.fin ? ~(uint64_t)0 : 0), v[15] = IV[7];
}
TG_HD void slot_idx(int half, int i, int* idx) {
  idx[0] = i;
  if (half == 0) idx[1] = 4 + i, idx[2] = 8 + i, idx[3] = 12 + i;
  else idx[1] = 4 + (i + 1) % 4, idx[2] = 8 + (i + 2) % 4, idx[3] = 12 + (i + 3) % 4;
}
struct GOut {
  uint64_t f[NFIELD];   // the slot's 12 fields as 64-bit words
  uint64_t al, cl;
  unsigned k[8];
  uint64_t a2, b2, c2, d2;
};
TG_HD uint64_t add_k(uint64_t t0, uint64_t t1, uint64_t t2, unsigned& k0, unsigned& k1) {   // t0 + t1 + t2 by 32-bit halves, carries out
  const uint64_t lo = (t0 & 0xFFFFFFFFu) + (t1 & 0xFFFFFFFFu) + (t2 & 0xFFFFFFFFu);
  k0 = (unsigned)(lo >> 32);
  const uint64_t hi = (t0 >> 32) + (t1 >> 32) + (t2 >> 32) + k0;
  k1 = (unsigned)(hi >> 32);
  return (lo & 0xFFFFFFFFu) | ((hi & 0xFFFFFFFFu) << 32);
}
TG_HD void g_full(uint64_t a, uint64_t b, uint64_t c, uint64_t d, uint64_t x, uint64_t y, GOut& o) {
  const uint64_t a1 = add_k(a, b, x, o.k[0], o.k[1]);
  const uint64_t e1 = d ^ a1, d1 = rotr64(e1, 32);
  const uint64_t c1 = add_k(c, d1, 0, o.k[2], o.k[3]);
  const uint64_t f1 = b ^ c1, b1 = rotr64(f1, 24);
  const uint64_t a2 = add_k(a1, b1, y, o.k[4], o.k[5]);
  const uint64_t e2 = d1 ^ a2, d2 = rotr64(e2, 16);
  const uint64_t c2 = add_k(c1, d2, 0, o.k[6], o.k[7]);
  const uint64_t f2 = b1 ^ c2, b2v = rotr64(f2, 63);
  const uint64_t top = (f2 >> 7) & 0x0101010101010101ull;        // TOP byte j = bit 7 of f2's byte j
  o.f[BIN] = b, o.f[DIN] = d, o.f[A1] = a1, o.f[E1] = e1, o.f[C1] = c1, o.f[F1] = f1, o.f[A2] = a2, o.f[E2] = e2, o.f[C2] = c2, o.f[F2] = f2;
  o.f[TOP] = top, o.f[B2] = b2v;
  o.al = a, o.cl = c;
  o.a2 = a2, o.b2 = b2v, o.c2 = c2, o.d2 = d2;
}
TG_HD void g_row(uint64_t* v, const uint64_t* m, const uint8_t (*SIGMA)[16], int s) {   // G row s (1..24) applied to v
  const int r = (s - 1) >> 1, half = (s - 1) & 1;
  for (int i = 0; i < NSLOT; ++i) {
    int idx[4];
    slot_idx(half, i, idx);
    const uint64_t x = m[SIGMA[r % 10][2 * (4 * half + i)]], y = m[SIGMA[r % 10][2 * (4 * half + i) + 1]];
    uint64_t a = v[idx[0]], b = v[idx[1]], c = v[idx[2]], d = v[idx[3]];
    a = a + b + x, d = rotr64(d ^ a, 32), c = c + d, b = rotr64(b ^ c, 24);
    a = a + b + y, d = rotr64(d ^ a, 16), c = c + d, b = rotr64(b ^ c, 63);
    v[idx[0]] = a, v[idx[1]] = b, v[idx[2]] = c, v[idx[3]] = d;
  }
}
TG_HD void expand(const Block& b, const uint64_t* IV, const uint8_t (*SIGMA)[16], Expanded& e) {
  uint64_t v[16];
  init_v(b, IV, v);
  for (int s = ROW_G0; s <= ROW_GLAST; ++s) {
    for (int i = 0; i < 16; ++i) e.v[s - 1][i] = v[i];
    g_row(v, b.m, SIGMA, s);
  }
  for (int i = 0; i < 16; ++i) e.v[24][i] = v[i];
  for (int k = 0; k < 8; ++k) e.hn[k] = v[k] ^ v[k + 8] ^ b.h_in[k];
}
TG_HD void compress(uint64_t (&h)[8], const Block& b, const uint64_t* IV, const uint8_t (*SIGMA)[16]) {   // host side of the chain
  uint64_t v[16];
  init_v(b, IV, v);
  for (int s = ROW_G0; s <= ROW_GLAST; ++s) g_row(v, b.m, SIGMA, s);
  for (int k = 0; k < 8; ++k) h[k] = v[k] ^ v[k + 8] ^ b.h_in[k];
}

// Row `r` of block `blk`.  hn_prev = the HN the block before handed over (zeros for the first block), dlatch = the 4 words latched in
// D.  look(a, b) is called once per looked-up triple of the row with its (a, b) bytes — the triples of blake2b_bytes_air.tuples(),
// each as soon as its words exist (the callers count multiplicities, so the order is free, and nothing is held in arrays across the
// row: that was 592 B of scratch per lane; rows a caller must not count are its business).  The table columns' MULT is written as 0.
template <class Put, class Look>
TG_HD void row(const Block& blk, const Expanded& e, const uint64_t* hn_prev, const uint64_t* dlatch, int r, size_t rowi,
               const uint64_t* IV, const uint8_t (*SIGMA)[16], Put put, Look look) {
  for (int k = 0; k < PERIOD; ++k) put(SEL + k, (uint64_t)(k == r));
  for (int k = 0; k < 8; ++k) put(H + 2 * k, blk.h_in[k] & 0xFFFFFFFFu), put(H + 2 * k + 1, blk.h_in[k] >> 32);
  for (int k = 0; k < 16; ++k) put(M + 2 * k, blk.m[k] & 0xFFFFFFFFu), put(M + 2 * k + 1, blk.m[k] >> 32);
  for (int k = 0; k < 4; ++k) put(D + 2 * k, dlatch[k] & 0xFFFFFFFFu), put(D + 2 * k + 1, dlatch[k] >> 32);
  put(T, blk.t), put(F, (uint64_t)blk.fin), put(TB, blk.tb_prev);
  // HN: what the previous block handed over until the finalisation rows rebuild it, four words per row
  uint64_t hn[8];
  for (int k = 0; k < 8; ++k) hn[k] = hn_prev[k];
  if (r >= ROW_FIN0 + 1)
    for (int k = 0; k < 4; ++k) hn[k] = e.hn[k];
  if (r >= ROW_HAND)
    for (int k = 4; k < 8; ++k) hn[k] = e.hn[k];
  for (int k = 0; k < 8; ++k) put(HN + 2 * k, hn[k] & 0xFFFFFFFFu), put(HN + 2 * k + 1, hn[k] >> 32);
  const bool finrow = r == ROW_FIN0 || r == ROW_FIN0 + 1;
  for (int k = 0; k < 16; ++k) {
    const uint64_t w = finrow ? e.v[24][k] : 0;
    put(V + 2 * k, w & 0xFFFFFFFFu), put(V + 2 * k + 1, w >> 32);
  }
  // slots
#define TG_BYTE(w, j) ((unsigned)(((w) >> (8 * (j))) & 255))
  for (int g = 0; g < NSLOT; ++g) {
    GOut o;
    for (int q = 0; q < NFIELD; ++q) o.f[q] = 0;
    o.al = o.cl = 0;
    for (int q = 0; q < 8; ++q) o.k[q] = 0;
    if (r == 0) {
      // init_v's v[g], v[4 + (g + 1) % 4], v[8 + (g + 2) % 4], v[12 + (g + 3) % 4], read where they come from (no local vector)
      const int dk = (g + 3) % 4;
      const uint64_t a2 = blk.h_in[g], b2v = blk.h_in[4 + (g + 1) % 4], c2 = IV[(g + 2) % 4],
                     d2 = IV[4 + dk] ^ (dk == 0 ? blk.t : (dk == 2 && blk.fin ? ~(uint64_t)0 : 0));
      const uint64_t e2 = rotr64(d2, 48), e1 = rotr64(e2 ^ a2, 32);
      o.f[A2] = a2, o.f[B2] = b2v, o.f[C2] = c2, o.f[E2] = e2, o.f[E1] = e1, o.f[A1] = e1, o.f[F2] = c2;
    } else if (r >= ROW_G0 && r <= ROW_GLAST) {
      const int rr = (r - 1) >> 1, half = (r - 1) & 1;
      int idx[4];
      slot_idx(half, g, idx);
      const uint64_t* v = e.v[r - 1];
      const uint64_t x = blk.m[SIGMA[rr % 10][2 * (4 * half + g)]], y = blk.m[SIGMA[rr % 10][2 * (4 * half + g) + 1]];
      g_full(v[idx[0]], v[idx[1]], v[idx[2]], v[idx[3]], x, y, o);
    }
    for (int q = 0; q < NFIELD; ++q)
      for (int j = 0; j < 8; ++j) put(fcol(g, q, j), (o.f[q] >> (8 * j)) & 255);
    for (int j = 0; j < 8; ++j) {
      look(TG_BYTE(o.f[DIN], j), TG_BYTE(o.f[A1], j));
      look(TG_BYTE(o.f[BIN], j), TG_BYTE(o.f[C1], j));
      look(TG_BYTE(o.f[E1], (j + 4) % 8), TG_BYTE(o.f[A2], j));
      look(TG_BYTE(o.f[F1], (j + 3) % 8), TG_BYTE(o.f[C2], j));
      look(TG_BYTE(o.f[B2], j), 0u);
    }
    const int base = SLOT + g * SLOT_W + 8 * NFIELD;
    put(base + 0, o.al & 0xFFFFFFFFu), put(base + 1, o.al >> 32), put(base + 2, o.cl & 0xFFFFFFFFu), put(base + 3, o.cl >> 32);
    for (int q = 0; q < 8; ++q) put(base + 4 + q, (uint64_t)o.k[q]);
  }
  // finalisation groups: FA FB FE FH FO
  for (int g = 0; g < NFING; ++g) {
    uint64_t fa = 0, fb = 0, fe = 0, fh = 0, fo = 0;
    if (r == 0 && g == 0) {
      fa = blk.t, fb = IV[4], fe = blk.t ^ IV[4], fo = fe;
    } else if (finrow) {
      const int k = 4 * (r - ROW_FIN0) + g;
      fa = e.v[24][k], fb = e.v[24][k + 8], fe = fa ^ fb, fh = blk.h_in[k], fo = fe ^ fh;
    }
    for (int j = 0; j < 8; ++j) {
      put(fincol(g, 0, j), (uint64_t)TG_BYTE(fa, j)), put(fincol(g, 1, j), (uint64_t)TG_BYTE(fb, j)), put(fincol(g, 2, j), (uint64_t)TG_BYTE(fe, j));
      put(fincol(g, 3, j), (uint64_t)TG_BYTE(fh, j)), put(fincol(g, 4, j), (uint64_t)TG_BYTE(fo, j));
      look(TG_BYTE(fa, j), TG_BYTE(fb, j));
      look(TG_BYTE(fe, j), TG_BYTE(fh, j));
    }
  }
  const uint64_t by = (r >= 1 && r <= 16) ? blk.m[r - 1] : 0;
  for (int j = 0; j < 8; ++j) put(BY + j, (by >> (8 * j)) & 255);
  // the XOR table: half k lists a = 128 k + ((row >> 8) & 127), b = row & 255
  for (int k = 0; k < NTAB; ++k) {
    const uint64_t ta = 128 * k + ((rowi >> 8) & 127), tb = rowi & 255;
    put(tabcol(k, 0), ta), put(tabcol(k, 1), tb), put(tabcol(k, 2), ta ^ tb);
    for (int i = 0; i < 8; ++i) put(tabcol(k, 3 + i), (ta >> i) & 1), put(tabcol(k, 11 + i), (tb >> i) & 1);
    put(tabcol(k, 19), (uint64_t)0);
  }
  for (int j = 0; j < 8; ++j) look(TG_BYTE(by, j), 0u);
#undef TG_BYTE
}
}  // namespace b2
}  // namespace tg
