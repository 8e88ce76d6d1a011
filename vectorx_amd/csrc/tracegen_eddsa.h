// The batched-EdDSA table's trace (vectorx_amd/eddsa_air.py: Layout(16, NB), _simulate, _instance_blocks, generate_trace), the part that
// is the same on the host and on the device.  Two phases:
//   simulate_instance  one thread per INSTANCE walks the straight-line program of [S]B - [h]A (16 + 42 NB + 4 rows of 255-bit field
//                      arithmetic, each row z = x y + e mod p with its quotient) and leaves x, y, z, q of every row in a side buffer;
//   row                one thread per ROW expands that row's cells: 16-bit limbs, the 30 carries of the coefficient relation, the value
//                      every register holds on the row (= z of the row that last wrote it — a static function of the row's position),
//                      the scalar-bit bookkeeping columns.
// In the reference this witness generation is Curta's (`curta_eddsa_verify_sigs_conditional`,
// /root/reference/circuits/builder/justification.rs:237-243); the numpy generator it replaces here took 1.4 s per 2^20-row table.
// Limbs of 16 bits only (the production layout); compared cell by cell with the numpy generator in tests/test_tracegen.py.
#pragma once
#include "tracegen_core.h"
#include "tracegen_eddsa_ops.h"

namespace tg {
namespace ed {

constexpr int NL = 16, LB = 16, NC = 2 * NL - 2;
// column map of eddsa_air.Layout(limb_bits = 16, scalar_bits = NB)
// `full` = the FULL program (eddsa_air.Layout(full=True): decompression, digest mod L, S < L inside the instance): longer prologue and
// epilogue, 32 word columns for the statement (A's encoding, the digest, R's encoding)
struct Cols {
  int full, NP, NE, NT;      // the program: prologue rows, epilogue rows, row types
  int NB, NW, L, RT, REG, X, Y, Z, Q, W, NLOOK, BIT, KACC, BND, FIN, POS, J, SW, ACT, AENC, DW, RENC, TBL, MULT, N, XROW, YROW;
};
TG_HD Cols cols(int NB, int full = 0) {
  Cols c;
  c.full = full, c.NP = full ? NP_FULL : NP, c.NE = full ? NE_FULL : NE, c.NT = full ? NT_FULL : NT;
  c.NB = NB, c.NW = NB / 32, c.L = c.NP + NLOOP * NB + c.NE;
  c.RT = 0, c.REG = c.RT + c.NT, c.X = c.REG + NL * NREG, c.Y = c.X + NL, c.Z = c.Y + NL, c.Q = c.Z + NL, c.W = c.Q + NL;
  c.NLOOK = 2 * NL + 2 * NC;
  c.BIT = c.W + 2 * NC, c.KACC = c.BIT + 2, c.BND = c.KACC + 2, c.FIN = c.BND + 1, c.POS = c.FIN + 1, c.J = c.POS + 32, c.SW = c.J + c.NW;
  c.ACT = c.SW + 2 * c.NW, c.AENC = c.ACT + 1, c.DW = c.AENC + 8, c.RENC = c.DW + 16;
  c.TBL = full ? c.RENC + 8 : c.ACT + 1, c.MULT = c.TBL + 1, c.N = c.MULT + 1;
  c.XROW = c.L - c.NE + 2, c.YROW = c.L - c.NE + 3;
  return c;
}
TG_HD const Op& op_at(const Cols& c, int t) { return c.full ? OPS_FULL[t] : OPS[t]; }

struct Sig {
  uint64_t ax[4], ay[4], s[4], h[4];   // the public key (affine) and the two scalars, little-endian 64-bit words
  uint64_t d[8];                       // full program: the SHA-512 digest of R || A || M as a little-endian integer
};
struct RowVals {
  uint64_t x[4], y[4], z[4], q[4];
};

typedef unsigned __int128 u128;
// add / subtract with carry: clang's builtins hand the device compiler the carry chain (v_addc_co_u32) instead of compare-and-select
// sequences; g++ builds the TEST-ONLY host copy of this file
#if defined(__clang__)
#define TG_ADDC(a, b, cin, cout) __builtin_addcll((a), (b), (cin), (cout))
#define TG_SUBC(a, b, cin, cout) __builtin_subcll((a), (b), (cin), (cout))
#else
static inline unsigned long long tg_addc(unsigned long long a, unsigned long long b, unsigned long long cin, unsigned long long* cout) {
  const unsigned long long s1 = a + b, s2 = s1 + cin;
  *cout = (unsigned long long)(s1 < a) | (unsigned long long)(s2 < s1);
  return s2;
}
static inline unsigned long long tg_subc(unsigned long long a, unsigned long long b, unsigned long long cin, unsigned long long* cout) {
  const unsigned long long t1 = a - b, t2 = t1 - cin;
  *cout = (unsigned long long)(a < b) | (unsigned long long)(t1 < cin);
  return t2;
}
#define TG_ADDC(a, b, cin, cout) tg_addc((a), (b), (cin), (cout))
#define TG_SUBC(a, b, cin, cout) tg_subc((a), (b), (cin), (cout))
#endif
#if defined(__HIPCC__) || defined(__clang__)
#define TG_UNROLL _Pragma("unroll")
#else
#define TG_UNROLL
#endif
// z = (x y + e) mod p canonical, quo = (x y + e - z) / p, for p = 2^255 - 19 and x, y, e < 2^255.
// Round 6: every loop has a constant trip count and is unrolled, carries ride add-with-carry chains and the two "if it overflows"
// steps are masks — the round-5 form had data-dependent loop bounds (`k < 4 && carry`), which put its arrays behind dynamic register
// indexing (694 s_set_gpr_idx pairs and 1 800 s_nop in the kernel) on the one lane per signature that walks 10 772 dependent rows.
TG_HD void mul_add_divmod(const uint64_t* x, const uint64_t* y, const uint64_t* e, uint64_t* z, uint64_t* quo) {
  uint64_t v[8];
  {  // the 512-bit product, row by row: v += x[i] * y << 64 i
    unsigned long long carry = 0;
    TG_UNROLL for (int j = 0; j < 4; ++j) {
      const u128 t = (u128)x[0] * y[j] + carry;
      v[j] = (uint64_t)t;
      carry = (uint64_t)(t >> 64);
    }
    v[4] = carry;
    TG_UNROLL for (int i = 1; i < 4; ++i) {
      carry = 0;
      TG_UNROLL for (int j = 0; j < 4; ++j) {
        const u128 t = (u128)x[i] * y[j] + v[i + j] + carry;      // < 2^128: (2^64 - 1)^2 + 2 (2^64 - 1)
        v[i + j] = (uint64_t)t;
        carry = (uint64_t)(t >> 64);
      }
      v[i + 4] = carry;
    }
  }
  {
    unsigned long long c = 0;
    TG_UNROLL for (int k = 0; k < 4; ++k) v[k] = TG_ADDC(v[k], e[k], c, &c);
    TG_UNROLL for (int k = 4; k < 8; ++k) v[k] = TG_ADDC(v[k], 0ull, c, &c);
  }
  // V = H 2^255 + Lo = H p + (19 H + Lo)
  uint64_t H[4], lo[4];
  TG_UNROLL for (int k = 0; k < 4; ++k) H[k] = (v[k + 3] >> 63) | ((k + 4 < 8 ? v[k + 4] : 0) << 1);
  TG_UNROLL for (int k = 0; k < 4; ++k) lo[k] = v[k];
  lo[3] &= 0x7FFFFFFFFFFFFFFFull;
  uint64_t v1[5];
  {
    uint64_t carry = 0;
    TG_UNROLL for (int k = 0; k < 4; ++k) {
      const u128 t = (u128)H[k] * 19 + lo[k] + carry;
      v1[k] = (uint64_t)t;
      carry = (uint64_t)(t >> 64);
    }
    v1[4] = carry;
  }
  const uint64_t H1 = (v1[3] >> 63) | (v1[4] << 1);     // < 2^6
  v1[3] &= 0x7FFFFFFFFFFFFFFFull;
  {
    unsigned long long c = 0;                            // quo = H + H1
    quo[0] = TG_ADDC(H[0], H1, c, &c);
    TG_UNROLL for (int k = 1; k < 4; ++k) quo[k] = TG_ADDC(H[k], 0ull, c, &c);
  }
  {
    unsigned long long c = 0;                            // z = v1 + 19 H1
    z[0] = TG_ADDC(v1[0], 19 * H1, c, &c);
    TG_UNROLL for (int k = 1; k < 4; ++k) z[k] = TG_ADDC(v1[k], 0ull, c, &c);
  }
  // z < 2^255 + 2^11: at most one more p.  z >= p  <=>  z + 19 reaches 2^255; then z - p = (z + 19) mod 2^255 and quo += 1
  uint64_t w[4];
  {
    unsigned long long c = 0;
    w[0] = TG_ADDC(z[0], 19ull, c, &c);
    TG_UNROLL for (int k = 1; k < 4; ++k) w[k] = TG_ADDC(z[k], 0ull, c, &c);
  }
  const uint64_t ge = 0 - (w[3] >> 63);                  // all ones when z >= p
  w[3] &= 0x7FFFFFFFFFFFFFFFull;
  TG_UNROLL for (int k = 0; k < 4; ++k) z[k] = (w[k] & ge) | (z[k] & ~ge);
  {
    unsigned long long c = 0;
    quo[0] = TG_ADDC(quo[0], ge & 1, c, &c);
    TG_UNROLL for (int k = 1; k < 4; ++k) quo[k] = TG_ADDC(quo[k], 0ull, c, &c);
  }
}
// the same for the modulus L (the group order), for the FULL program's two reduction rows: plain shift-and-subtract long division of the
// 512-bit x y + e — two rows per instance, speed is irrelevant
TG_HD void mul_add_divmod_l(const uint64_t* x, const uint64_t* y, const uint64_t* e, uint64_t* z, uint64_t* quo) {
  uint64_t v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    uint64_t carry = 0;
    for (int j = 0; j < 4; ++j) {
      const u128 t = (u128)x[i] * y[j] + v[i + j] + carry;
      v[i + j] = (uint64_t)t;
      carry = (uint64_t)(t >> 64);
    }
    v[i + 4] = carry;
  }
  uint64_t carry = 0;
  for (int k = 0; k < 8; ++k) {
    const u128 t = (u128)v[k] + (k < 4 ? e[k] : 0) + carry;
    v[k] = (uint64_t)t;
    carry = (uint64_t)(t >> 64);
  }
  uint64_t r[5] = {0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int bit = 511; bit >= 0; --bit) {
    for (int k = 4; k > 0; --k) r[k] = (r[k] << 1) | (r[k - 1] >> 63);       // r = 2 r + bit  (r < L < 2^253 before: no overflow of 5 words)
    r[0] = (r[0] << 1) | ((v[bit >> 6] >> (bit & 63)) & 1);
    bool ge = r[4] != 0;
    if (!ge) {
      ge = true;
      for (int k = 3; k >= 0; --k)
        if (r[k] != ELL[k]) {
          ge = r[k] > ELL[k];
          break;
        }
    }
    if (ge) {
      uint64_t borrow = 0;
      for (int k = 0; k < 5; ++k) {
        const uint64_t sub = (k < 4 ? ELL[k] : 0);
        const uint64_t t1 = r[k] - sub, t2 = t1 - borrow;
        borrow = (r[k] < sub) | (t1 < borrow);
        r[k] = t2;
      }
      q[bit >> 6] |= (uint64_t)1 << (bit & 63);
    }
  }
  for (int k = 0; k < 4; ++k) z[k] = r[k], quo[k] = q[k];     // the quotient fits 256 bits for the operands the program feeds (x y < 2^384)
}
// ---- the LINEAR rows -----------------------------------------------------------------------------------------------------------------
// 20 of the 42 rows of a ladder step are not multiplications: the Y slot holds the CONSTANT 1, 2, p - 1 or p - 2 (additions, doublings,
// subtractions written as x (p - 1) + e).  For canonical x, e (< p) the same z = (x y + e) mod p and quo = (x y + e - z) / p follow from
// one or two additions / subtractions of p:
//   y = c in {0, 1, 2}:      V = c x + e < 3 p:           subtract p while V >= p, quo = how often;
//   y = p - c, c in {1, 2}:  V = x p - (c x - e):          z = e - c x + k p with the smallest k >= 0 (k <= c), quo = x - k.
// Taken only where the row's operation says so at COMPILE time (simulate_row is specialised per row of the ladder step): as a run-time
// test on y it cost more than it saved (profiles/r06_eddsa_simulate_analysis.md).  Branch-free: every lane walks its own signature.
TG_HD constexpr int linear_kind(const uint64_t* y) {   // 0: not linear; 1 + c for y = c (c = 0, 1, 2); 4 + c for y = p - c (c = 1, 2)
  return ((y[1] | y[2] | y[3]) == 0 && y[0] <= 2) ? 1 + (int)y[0]
       : (y[3] == 0x7FFFFFFFFFFFFFFFull && y[2] == ~0ull && y[1] == ~0ull && (y[0] == 0xFFFFFFFFFFFFFFECull || y[0] == 0xFFFFFFFFFFFFFFEBull))
             ? 4 + (int)(0xFFFFFFFFFFFFFFEDull - y[0]) : 0;
}
TG_HD bool below_p(const uint64_t* a) {
  return a[3] < 0x7FFFFFFFFFFFFFFFull || (a[3] == 0x7FFFFFFFFFFFFFFFull && !(a[2] == ~0ull && a[1] == ~0ull && a[0] >= 0xFFFFFFFFFFFFFFEDull));
}
TG_HD void linear_row(int kind, const uint64_t* x, const uint64_t* e, uint64_t* z, uint64_t* quo) {
  const uint64_t P5[5] = {0xFFFFFFFFFFFFFFEDull, ~0ull, ~0ull, 0x7FFFFFFFFFFFFFFFull, 0};
  const int c = kind <= 3 ? kind - 1 : kind - 4;             // the small factor
  uint64_t cx[5] = {0, 0, 0, 0, 0}, w[5] = {e[0], e[1], e[2], e[3], 0};
  if (c == 1) {
    TG_UNROLL for (int k = 0; k < 4; ++k) cx[k] = x[k];
  } else if (c == 2) {
    TG_UNROLL for (int k = 0; k < 4; ++k) cx[k] = (x[k] << 1) | (k ? x[k - 1] >> 63 : 0);
    cx[4] = x[3] >> 63;
  }
  if (kind <= 3) {                                           // V = c x + e < 3 p: take p away while it fits
    unsigned long long cy = 0;
    TG_UNROLL for (int k = 0; k < 5; ++k) w[k] = TG_ADDC(w[k], cx[k], cy, &cy);
    uint64_t cnt = 0;
    TG_UNROLL for (int t = 0; t < 2; ++t) {
      uint64_t d[5];
      unsigned long long bw = 0;
      TG_UNROLL for (int k = 0; k < 5; ++k) d[k] = TG_SUBC(w[k], P5[k], bw, &bw);
      const uint64_t keep = bw - 1;                          // all ones when w >= p (no borrow)
      TG_UNROLL for (int k = 0; k < 5; ++k) w[k] = (d[k] & keep) | (w[k] & ~keep);
      cnt += keep & 1;
    }
    TG_UNROLL for (int k = 0; k < 4; ++k) z[k] = w[k];
    quo[0] = cnt, quo[1] = quo[2] = quo[3] = 0;
  } else {                                                   // z = e - c x + k p with the smallest k >= 0, quo = x - k
    uint64_t d[5];
    unsigned long long bw = 0;
    TG_UNROLL for (int k = 0; k < 5; ++k) d[k] = TG_SUBC(w[k], cx[k], bw, &bw);     // two's complement, > -2^257
    uint64_t kk = 0;
    TG_UNROLL for (int t = 0; t < 2; ++t) {
      const uint64_t neg = 0 - (d[4] >> 63);                 // all ones while the value is negative
      unsigned long long cy = 0;
      TG_UNROLL for (int k = 0; k < 5; ++k) d[k] = TG_ADDC(d[k], P5[k] & neg, cy, &cy);
      kk += neg & 1;
    }
    TG_UNROLL for (int k = 0; k < 4; ++k) z[k] = d[k];
    unsigned long long b2 = 0;                               // quo = x - k
    quo[0] = TG_SUBC(x[0], kk, b2, &b2);
    TG_UNROLL for (int k = 1; k < 4; ++k) quo[k] = TG_SUBC(x[k], 0ull, b2, &b2);
  }
}
TG_HD void mulmod(const uint64_t* x, const uint64_t* y, uint64_t* z) {
  const uint64_t zero[4] = {0, 0, 0, 0};
  uint64_t q[4], t[4];
  mul_add_divmod(x, y, zero, t, q);
  for (int k = 0; k < 4; ++k) z[k] = t[k];
}
TG_HD void inverse(const uint64_t* a, uint64_t* out) {       // a^(p - 2), p - 2 = 2^255 - 21
  uint64_t r[4] = {1, 0, 0, 0}, b[4];
  for (int k = 0; k < 4; ++k) b[k] = a[k];
  const uint64_t ex[4] = {0xFFFFFFFFFFFFFFEBull, ~0ull, ~0ull, 0x7FFFFFFFFFFFFFFFull};
  for (int i = 0; i < 255; ++i) {
    if ((ex[i >> 6] >> (i & 63)) & 1) mulmod(r, b, r);
    mulmod(b, b, b);
  }
  for (int k = 0; k < 4; ++k) out[k] = r[k];
}
TG_HD int scalar_bit(const uint64_t* v, int NB, int step) {   // bit of step `step`: most significant of the NB bits first
  const int b = NB - 1 - step;
  return (int)((v[b >> 6] >> (b & 63)) & 1);
}
TG_HD int row_type(int rho, int NB, int np = NP) {
  if (rho < np) return rho;
  if (rho < np + NLOOP * NB) return np + (rho - np) % NLOOP;
  return np + NLOOP + (rho - np - NLOOP * NB);
}

// phase 1: one instance.  out[rho], rho < L.  Returns 0, or 1 when a row that must produce 1 does not (A not on the curve, Z = 0), 2 when
// a value that must be canonical is not (full program: a coordinate >= p, S or the reduced digest >= L), 3 when an integer identity of
// the full program does not hold (cannot happen for honest inputs).
// `regs` = the instance's register file, NREG x 4 words (the device keeps it in LDS: indexed by the row's op, it would otherwise live in
// scratch memory, whose latency every one of the 10 772 dependent rows would pay several times)
// one row of the program: `op` is the row's operation, `step` the ladder step it belongs to.  Force-inlined: called with a COMPILE-TIME op
// for the 42 rows of a ladder step (simulate_instance), the operation's fields fold into the code — no table fetch, no dispatch on the
// kind of row, register-file addresses as constants; the diagnostic of profiles/r06_eddsa_simulate_analysis.md put that bookkeeping at
// 13 of the kernel's 24 ms.
#if defined(__HIPCC__) || defined(__clang__) || defined(__GNUC__)
#define TG_ALWAYS_INLINE __attribute__((always_inline))
#else
#define TG_ALWAYS_INLINE
#endif
TG_HD TG_ALWAYS_INLINE void simulate_row(const Cols& c, const Sig& sg, RowVals* out, uint64_t (*regs)[4], const Op& op, int rho, int sbit, int hbit, int& bad) {
  {
    RowVals& rv = out[rho];
    uint64_t x[4] = {0, 0, 0, 0}, y[4] = {0, 0, 0, 0}, e[4] = {0, 0, 0, 0}, z[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (op.free_) {
      switch (op.wit) {
        case 0: for (int k = 0; k < 4; ++k) z[k] = sg.ax[k]; break;
        case 1: for (int k = 0; k < 4; ++k) z[k] = sg.ay[k]; break;
        case 2: inverse(regs[op.witreg], z); break;
        case 3: {                                       // bound - 1 - value
          uint64_t borrow = 1;                          // start from "- 1"
          for (int k = 0; k < 4; ++k) {
            const uint64_t a = op.bound[k], b = regs[op.witreg][k];
            const uint64_t t1 = a - b, t2 = t1 - borrow;
            borrow = (a < b) | (t1 < borrow);
            z[k] = t2;
          }
          if (borrow) bad = 2;
          break;
        }
        case 4: for (int k = 0; k < 4; ++k) z[k] = (regs[op.witreg][k] >> 1) | (k < 3 ? regs[op.witreg][k + 1] << 63 : 0); break;
        case 5: z[0] = regs[op.witreg][0] & 1; break;
        case 6: for (int k = 0; k < 4; ++k) z[k] = sg.d[4 + k]; break;
        case 7: for (int k = 0; k < 4; ++k) z[k] = sg.d[k]; break;
        default: for (int k = 0; k < 4; ++k) z[k] = sg.s[k]; break;
      }
    } else {
      for (int k = 0; k < 4; ++k) x[k] = regs[op.x][k];
      bool from_reg = op.ykind == 0, on = true;
      if (op.ykind == 2) on = sbit != 0;
      if (op.ykind == 3) on = hbit != 0, from_reg = on;
      for (int k = 0; k < 4; ++k) y[k] = from_reg ? regs[op.yreg][k] : (op.ykind == 1 || (op.ykind == 2 && on) ? op.con[k] : op.coff[k]);
      if (op.e >= 0) for (int k = 0; k < 4; ++k) e[k] = regs[op.e][k];
      if (op.qzero) {                                   // an integer identity: Z is what the row requires, x y + e must BE it
        const uint64_t* want = op.zkind == 1 ? op.zconst : regs[op.zreg];
        // y is 1 or 2 here: x y + e in 5 words
        uint64_t t[5] = {0, 0, 0, 0, 0}, carry = 0;
        for (int k = 0; k < 4; ++k) {
          const u128 w = (u128)x[k] * y[0] + e[k] + carry;
          t[k] = (uint64_t)w;
          carry = (uint64_t)(w >> 64);
        }
        t[4] = carry;
        if (y[1] | y[2] | y[3] | t[4]) bad = bad ? bad : 3;
        for (int k = 0; k < 4; ++k) {
          if (t[k] != want[k]) bad = bad ? bad : 3;
          z[k] = want[k];
        }
      } else if (op.modl) {
        mul_add_divmod_l(x, y, e, z, q);
      } else {
        // a row whose Y slot is a small / negated-small CONSTANT (known when `op` is a compile-time constant: the ladder rows)
        const int lk = op.ykind == 1 ? linear_kind(op.con) : 0;
        if (lk && below_p(x) && below_p(e)) linear_row(lk, x, e, z, q);
        else mul_add_divmod(x, y, e, z, q);
      }
    }
    if (op.one && !(z[0] == 1 && z[1] == 0 && z[2] == 0 && z[3] == 0)) bad = bad ? bad : 1;
    for (int k = 0; k < 4; ++k) regs[op.dst][k] = z[k], rv.x[k] = x[k], rv.y[k] = y[k], rv.z[k] = z[k], rv.q[k] = q[k];
  }
}
template <int I>
struct LoopRows {   // rows I .. NLOOP - 1 of a ladder step, each with its operation as a constant
  static TG_HD TG_ALWAYS_INLINE void run(const Cols& c, const Sig& sg, RowVals* out, uint64_t (*regs)[4], int rho0, int sbit, int hbit, int& bad) {
    simulate_row(c, sg, out, regs, OPS[NP + I], rho0 + I, sbit, hbit, bad);   // the base and the full program share the loop (the cell-by-cell tests cover both)
    LoopRows<I + 1>::run(c, sg, out, regs, rho0, sbit, hbit, bad);
  }
};
template <>
struct LoopRows<NLOOP> {
  static TG_HD TG_ALWAYS_INLINE void run(const Cols&, const Sig&, RowVals*, uint64_t (*)[4], int, int, int, int&) {}
};
TG_HD int simulate_instance(const Cols& c, const Sig& sg, RowVals* out, uint64_t (*regs)[4]) {
  for (int r = 0; r < NREG; ++r)
    for (int k = 0; k < 4; ++k) regs[r][k] = 0;
  const int NB = c.NB;
  int bad = 0;
  // the scalars' bits: a 64-bit word of S and of h is fetched once per 64 ladder steps (the rows used to read the signature from global
  // memory on every bit-dependent row: six dependent loads per step for the one wavefront there is)
  const int s0 = scalar_bit(sg.s, NB, 0), h0 = scalar_bit(sg.h, NB, 0);
  for (int rho = 0; rho < c.NP; ++rho) simulate_row(c, sg, out, regs, op_at(c, rho), rho, s0, h0, bad);
  uint64_t sw = 0, hw = 0;
  for (int step = 0; step < NB; ++step) {
    const int b = NB - 1 - step;
    if (step == 0 || (b & 63) == 63) sw = sg.s[b >> 6], hw = sg.h[b >> 6];
    LoopRows<0>::run(c, sg, out, regs, c.NP + NLOOP * step, (int)((sw >> (b & 63)) & 1), (int)((hw >> (b & 63)) & 1), bad);
  }
  for (int t = 0; t < c.NE; ++t) simulate_row(c, sg, out, regs, op_at(c, c.NP + NLOOP + t), c.NP + NLOOP * NB + t, s0, h0, bad);
  return bad;
}

// the row (relative to its instance's first row; may be NEGATIVE = in the previous instance) whose z register `r` shows on row `rho`
TG_HD int last_writer_in(const Op* ops, int count, int before, int r) {   // highest t < before with ops[t].dst == r, or -1
  for (int t = (before < count ? before : count) - 1; t >= 0; --t)
    if (ops[t].dst == r) return t;
  return -1;
}
TG_HD int reg_source(const Cols& c, int rho, int r) {
  const int NB = c.NB, NP = c.NP, NE = c.NE, L = c.L;      // (shadow the base program's constants)
  const Op* PRO = c.full ? OPS_FULL : OPS;
  const Op* LOOP = PRO + NP;
  const Op* EPI = PRO + NP + NLOOP;
  if (rho >= NP + NLOOP * NB) {                         // epilogue
    const int t = last_writer_in(EPI, NE, rho - NP - NLOOP * NB, r);
    if (t >= 0) return NP + NLOOP * NB + t;
    const int l = last_writer_in(LOOP, NLOOP, NLOOP, r);
    if (l >= 0) return NP + NLOOP * (NB - 1) + l;
    const int p = last_writer_in(PRO, NP, NP, r);
    if (p >= 0) return p;
  } else if (rho >= NP) {                               // loop
    const int step = (rho - NP) / NLOOP, i = (rho - NP) % NLOOP;
    const int t = last_writer_in(LOOP, NLOOP, i, r);
    if (t >= 0) return NP + NLOOP * step + t;
    if (step > 0) {
      const int l = last_writer_in(LOOP, NLOOP, NLOOP, r);
      if (l >= 0) return NP + NLOOP * (step - 1) + l;
    }
    const int p = last_writer_in(PRO, NP, NP, r);
    if (p >= 0) return p;
  } else {
    const int p = last_writer_in(PRO, NP, rho, r);
    if (p >= 0) return p;
  }
  // nothing in this instance yet: what the PREVIOUS instance left (every register is written in every instance)
  int f = last_writer_in(EPI, NE, NE, r);
  if (f >= 0) f += NP + NLOOP * NB;
  else {
    f = last_writer_in(LOOP, NLOOP, NLOOP, r);
    if (f >= 0) f += NP + NLOOP * (NB - 1);
    else f = last_writer_in(PRO, NP, NP, r);
  }
  return f - L;
}
// The same as a table: class of the row (prologue row | loop row of step 0 | loop row of a later step | epilogue row) x register ->
// {kind, value}: kind 0: the row `value` rows above; kind 1: row `value` of this instance; kind 2: row `value` of the PREVIOUS instance.
constexpr int NCLASS = NP_FULL + 2 * NLOOP + NE_FULL;      // room for either program
struct RegSrc {
  short kind[NCLASS][NREG];
  int value[NCLASS][NREG];
};
TG_HD int row_class(const Cols& c, int rho) {
  const int np = c.NP;
  if (rho < np) return rho;
  if (rho < np + NLOOP * c.NB) return np + ((rho - np) / NLOOP ? NLOOP : 0) + (rho - np) % NLOOP;
  return np + 2 * NLOOP + (rho - np - NLOOP * c.NB);
}
inline void make_reg_src(const Cols& c, RegSrc& t) {
  const int np = c.NP, nclass = np + 2 * NLOOP + c.NE;
  for (int cl = 0; cl < nclass; ++cl) {
    // a representative row of the class (NB >= 32: step 1 exists)
    const int rho = cl < np + 2 * NLOOP ? cl : np + NLOOP * c.NB + (cl - np - 2 * NLOOP);
    for (int r = 0; r < NREG; ++r) {
      const int src = reg_source(c, rho, r);
      if (src < 0) t.kind[cl][r] = 2, t.value[cl][r] = src + c.L;
      else if (src < np || cl < np) t.kind[cl][r] = 1, t.value[cl][r] = src;
      else t.kind[cl][r] = 0, t.value[cl][r] = rho - src;
    }
  }
}
TG_HD long long reg_source_of(const Cols& c, const RegSrc& t, int rho, int r) {
  const int cl = row_class(c, rho);
  const int v = t.value[cl][r];
  return t.kind[cl][r] == 0 ? rho - v : (t.kind[cl][r] == 1 ? v : (long long)v - c.L);
}
TG_HD unsigned limb16(const uint64_t* v, int i) { return (unsigned)((v[i >> 2] >> (16 * (i & 3))) & 0xFFFF); }

// phase 2: the cells of trace row `row` (instance `inst`, position `rho`).  vals = the side buffer of ALL instances ([inst * L + rho]);
// sigs[inst] for inst < nsig, the filler signature (A = B, S = h = 0) above.  look(limb) once per looked-up limb (Z, Q, W columns).
template <class Put, class Look>
TG_HD void row(const Cols& c, const RegSrc& rsrc, const RowVals* vals, const Sig* sigs, int nsig, const Sig& filler, size_t rowi, Put put, Look look) {
  const int L = c.L, NB = c.NB, np = c.NP;
  const size_t inst = rowi / L;
  const int rho = (int)(rowi % L);
  const int rt = row_type(rho, NB, np);
  const Op& op = op_at(c, rt);
  const RowVals& rv = vals[rowi];
  for (int t = 0; t < c.NT; ++t) put(c.RT + t, (uint64_t)(t == rt));
  // registers
  uint64_t ev[4] = {0, 0, 0, 0};
  for (int r = 0; r < NREG; ++r) {
    const long long src = reg_source_of(c, rsrc, rho, r);
    uint64_t v[4] = {0, 0, 0, 0};
    if (src >= 0 || inst > 0) {
      const RowVals& s = vals[(size_t)((long long)(inst * L) + src)];
      for (int k = 0; k < 4; ++k) v[k] = s.z[k];
    }
    for (int i = 0; i < NL; ++i) put(c.REG + NL * r + i, (uint64_t)limb16(v, i));
    if (!op.free_ && op.e == r)
      for (int k = 0; k < 4; ++k) ev[k] = v[k];
  }
  // X Y Z Q limbs and the carries of  X(t) Y(t) + E(t) - Z(t) - Q(t) P(t) = (t - 2^16) W(t);  P = p, or L on the full program's reduction rows
  // (round 6: the coefficient d_k is formed COLUMN BY COLUMN on the way down the carry chain — the 31 accumulators of the row-by-row form
  //  were 62 registers that the 1024-thread kernel does not have: 400 B of scratch per lane)
  unsigned xl[NL], yl[NL], zl[NL], ql[NL];
  for (int i = 0; i < NL; ++i) {
    xl[i] = limb16(rv.x, i), yl[i] = limb16(rv.y, i), zl[i] = limb16(rv.z, i), ql[i] = limb16(rv.q, i);
    put(c.X + i, (uint64_t)xl[i]), put(c.Y + i, (uint64_t)yl[i]), put(c.Z + i, (uint64_t)zl[i]), put(c.Q + i, (uint64_t)ql[i]);
    look(zl[i]);
  }
  for (int i = 0; i < NL; ++i) look(ql[i]);
  long long prev = 0;
  TG_UNROLL for (int k = 0; k < 2 * NL - 1; ++k) {
    long long dk = 0;
    if (!op.free_) {
      const int i0 = k < NL ? 0 : k - NL + 1, i1 = k < NL ? k : NL - 1;
      TG_UNROLL for (int i = i0; i <= i1; ++i) {
        const int j = k - i;
        const long long pj = op.modl ? (long long)limb16(ELL, j) : (j == 0 ? 0xFFED : (j == NL - 1 ? 0x7FFF : 0xFFFF));
        dk += (long long)xl[i] * yl[j] - (long long)ql[i] * pj;
      }
      if (k < NL) dk += (long long)limb16(ev, k) - (long long)zl[k];
    }
    const long long t = prev - dk;
    prev = t >> LB;                                      // exact: the low 16 bits of t are zero when the relation holds
    if (k <= 2 * NL - 3) {
      const long long off = prev + (1ll << (2 * LB - 1));
      const unsigned w0 = (unsigned)(off & 0xFFFF), w1 = (unsigned)((off >> LB) & 0xFFFF);
      put(c.W + 2 * k, (uint64_t)w0), put(c.W + 2 * k + 1, (uint64_t)w1);
      look(w0), look(w1);
    }
  }
  // scalar bookkeeping
  int step = rho >= np ? (rho - np) / NLOOP : 0;
  if (step > NB - 1) step = NB - 1;
  const bool after = rho >= np + NLOOP * NB;
  const Sig& me = (int)inst < nsig ? sigs[inst] : filler;
  const Sig& before = inst > 0 ? ((int)(inst - 1) < nsig ? sigs[inst - 1] : filler) : me;
  for (int s = 0; s < 2; ++s) {
    const uint64_t* mine = s ? me.h : me.s;
    const uint64_t* theirs = s ? before.h : before.s;
    const uint64_t cur = (uint64_t)scalar_bit(mine, NB, step);
    const uint64_t prev_last = inst > 0 ? (uint64_t)scalar_bit(theirs, NB, NB - 1) : (uint64_t)scalar_bit(mine, NB, 0);
    put(c.BIT + s, rho < np ? prev_last : cur);
    // the scalar's 32-bit words, most significant first; kacc = the bits of the current word so far
    const int wj = step / 32, hi_bit = NB - 32 * wj;      // word j covers bits [hi_bit - 32, hi_bit)
    const int lo_bit = hi_bit - 32;
    const uint64_t word = (mine[lo_bit >> 6] >> (lo_bit & 63)) & 0xFFFFFFFFull;
    const uint64_t kacc = word >> (31 - step % 32);
    put(c.KACC + s, rho < np ? prev_last : (after ? cur : kacc));
    for (int j = 0; j < c.NW; ++j) {
      const int lb = NB - 32 * (j + 1);
      put(c.SW + s * c.NW + j, (mine[lb >> 6] >> (lb & 63)) & 0xFFFFFFFFull);
    }
  }
  const int pos = after ? 0 : step % 32, word = after ? 0 : step / 32;
  for (int i = 0; i < 32; ++i) put(c.POS + i, (uint64_t)(i == pos));
  for (int j = 0; j < c.NW; ++j) put(c.J + j, (uint64_t)(j == word));
  const bool adv = rt == np + NLOOP - 1;
  const bool bnd = adv && step % 32 == 31;
  put(c.BND, (uint64_t)bnd);
  put(c.FIN, (uint64_t)(bnd && step / 32 == c.NW - 1));
  put(c.ACT, (uint64_t)((int)inst < nsig));
  if (c.full) {
    // the statement's 32-bit words, least significant first, constant over the instance: A's encoding (y, the parity of x as bit 255),
    // the digest, the encoding of the point the instance arrives at
    const uint64_t* rx = vals[inst * L + c.XROW].z;
    const uint64_t* ry = vals[inst * L + c.YROW].z;
    for (int j = 0; j < 8; ++j) {
      uint64_t a = (me.ay[j >> 1] >> (32 * (j & 1))) & 0xFFFFFFFFull, r = (ry[j >> 1] >> (32 * (j & 1))) & 0xFFFFFFFFull;
      if (j == 7) a |= (me.ax[0] & 1) << 31, r |= (rx[0] & 1) << 31;
      put(c.AENC + j, a), put(c.RENC + j, r);
    }
    for (int j = 0; j < 16; ++j) put(c.DW + j, (me.d[j >> 1] >> (32 * (j & 1))) & 0xFFFFFFFFull);
  }
  put(c.TBL, (uint64_t)(rowi & 0xFFFF));
  put(c.MULT, (uint64_t)0);
}

}  // namespace ed
}  // namespace tg
