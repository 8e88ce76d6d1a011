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
struct Cols {
  int NB, NW, L, RT, REG, X, Y, Z, Q, W, NLOOK, BIT, KACC, BND, FIN, POS, J, SW, ACT, TBL, MULT, N;
};
TG_HD Cols cols(int NB) {
  Cols c;
  c.NB = NB, c.NW = NB / 32, c.L = NP + NLOOP * NB + NE;
  c.RT = 0, c.REG = c.RT + NT, c.X = c.REG + NL * NREG, c.Y = c.X + NL, c.Z = c.Y + NL, c.Q = c.Z + NL, c.W = c.Q + NL;
  c.NLOOK = 2 * NL + 2 * NC;
  c.BIT = c.W + 2 * NC, c.KACC = c.BIT + 2, c.BND = c.KACC + 2, c.FIN = c.BND + 1, c.POS = c.FIN + 1, c.J = c.POS + 32, c.SW = c.J + c.NW;
  c.ACT = c.SW + 2 * c.NW, c.TBL = c.ACT + 1, c.MULT = c.TBL + 1, c.N = c.MULT + 1;
  return c;
}

struct Sig {
  uint64_t ax[4], ay[4], s[4], h[4];   // the public key (affine) and the two scalars, little-endian 64-bit words
};
struct RowVals {
  uint64_t x[4], y[4], z[4], q[4];
};

typedef unsigned __int128 u128;
// z = (x y + e) mod p canonical, quo = (x y + e - z) / p, for p = 2^255 - 19 and x, y, e < 2^255
TG_HD void mul_add_divmod(const uint64_t* x, const uint64_t* y, const uint64_t* e, uint64_t* z, uint64_t* quo) {
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
  {
    uint64_t carry = 0;
    for (int k = 0; k < 8; ++k) {
      const u128 t = (u128)v[k] + (k < 4 ? e[k] : 0) + carry;
      v[k] = (uint64_t)t;
      carry = (uint64_t)(t >> 64);
    }
  }
  // V = H 2^255 + Lo = H p + (19 H + Lo)
  uint64_t H[4], lo[4];
  for (int k = 0; k < 4; ++k) H[k] = (v[k + 3] >> 63) | ((k + 4 < 8 ? v[k + 4] : 0) << 1);
  for (int k = 0; k < 4; ++k) lo[k] = v[k];
  lo[3] &= 0x7FFFFFFFFFFFFFFFull;
  for (int k = 0; k < 4; ++k) quo[k] = H[k];
  uint64_t v1[5];
  {
    uint64_t carry = 0;
    for (int k = 0; k < 4; ++k) {
      const u128 t = (u128)H[k] * 19 + lo[k] + carry;
      v1[k] = (uint64_t)t;
      carry = (uint64_t)(t >> 64);
    }
    v1[4] = carry;
  }
  const uint64_t H1 = (v1[3] >> 63) | (v1[4] << 1);     // < 2^6
  v1[3] &= 0x7FFFFFFFFFFFFFFFull;
  {
    uint64_t carry = H1;
    for (int k = 0; k < 4 && carry; ++k) {
      const uint64_t s = quo[k] + carry;
      carry = s < quo[k];
      quo[k] = s;
    }
  }
  {
    uint64_t carry = 19 * H1;
    for (int k = 0; k < 4; ++k) {
      const uint64_t s = v1[k] + carry;
      carry = s < v1[k];
      z[k] = s;
    }
  }
  // z < 2^255 + 2^11: at most one more p
  const bool ge = z[3] > 0x7FFFFFFFFFFFFFFFull ||
                  (z[3] == 0x7FFFFFFFFFFFFFFFull && z[2] == ~0ull && z[1] == ~0ull && z[0] >= 0xFFFFFFFFFFFFFFEDull);
  if (ge) {
    // z -= p  <=>  z += 19, then drop bit 255
    uint64_t carry = 19;
    for (int k = 0; k < 4; ++k) {
      const uint64_t s = z[k] + carry;
      carry = s < z[k];
      z[k] = s;
    }
    z[3] &= 0x7FFFFFFFFFFFFFFFull;
    uint64_t c1 = 1;
    for (int k = 0; k < 4 && c1; ++k) {
      quo[k] += 1;
      c1 = quo[k] == 0;
    }
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
TG_HD int row_type(int rho, int NB) {
  if (rho < NP) return rho;
  if (rho < NP + NLOOP * NB) return NP + (rho - NP) % NLOOP;
  return NP + NLOOP + (rho - NP - NLOOP * NB);
}

// phase 1: one instance.  out[rho], rho < L.  Returns 0, or 1 when a row that must produce 1 does not (A not on the curve, Z = 0).
// `regs` = the instance's register file, NREG x 4 words (the device keeps it in LDS: indexed by the row's op, it would otherwise live in
// scratch memory, whose latency every one of the 10 772 dependent rows would pay several times)
TG_HD int simulate_instance(const Sig& sg, int NB, RowVals* out, uint64_t (*regs)[4]) {
  for (int r = 0; r < NREG; ++r)
    for (int k = 0; k < 4; ++k) regs[r][k] = 0;
  const int L = NP + NLOOP * NB + NE;
  int bad = 0;
  for (int rho = 0; rho < L; ++rho) {
    const Op& op = OPS[row_type(rho, NB)];
    const int step = rho < NP ? 0 : (rho - NP) / NLOOP;
    RowVals& rv = out[rho];
    uint64_t x[4] = {0, 0, 0, 0}, y[4] = {0, 0, 0, 0}, e[4] = {0, 0, 0, 0}, z[4], q[4] = {0, 0, 0, 0};
    if (op.free_) {
      if (op.dst == AX) for (int k = 0; k < 4; ++k) z[k] = sg.ax[k];
      else if (op.dst == AY) for (int k = 0; k < 4; ++k) z[k] = sg.ay[k];
      else inverse(regs[Z1], z);
    } else {
      for (int k = 0; k < 4; ++k) x[k] = regs[op.x][k];
      bool from_reg = op.ykind == 0, on = true;
      if (op.ykind == 2) on = scalar_bit(sg.s, NB, step) != 0;
      if (op.ykind == 3) on = scalar_bit(sg.h, NB, step) != 0, from_reg = on;
      for (int k = 0; k < 4; ++k) y[k] = from_reg ? regs[op.yreg][k] : (op.ykind == 1 || (op.ykind == 2 && on) ? op.con[k] : op.coff[k]);
      if (op.e >= 0) for (int k = 0; k < 4; ++k) e[k] = regs[op.e][k];
      mul_add_divmod(x, y, e, z, q);
    }
    if (op.one && !(z[0] == 1 && z[1] == 0 && z[2] == 0 && z[3] == 0)) bad = 1;
    for (int k = 0; k < 4; ++k) regs[op.dst][k] = z[k], rv.x[k] = x[k], rv.y[k] = y[k], rv.z[k] = z[k], rv.q[k] = q[k];
  }
  return bad;
}

// the row (relative to its instance's first row; may be NEGATIVE = in the previous instance) whose z register `r` shows on row `rho`
TG_HD int last_writer_in(const Op* ops, int count, int before, int r) {   // highest t < before with ops[t].dst == r, or -1
  for (int t = (before < count ? before : count) - 1; t >= 0; --t)
    if (ops[t].dst == r) return t;
  return -1;
}
TG_HD int reg_source(int rho, int r, int NB) {
  const int L = NP + NLOOP * NB + NE;
  const Op* PRO = OPS;
  const Op* LOOP = OPS + NP;
  const Op* EPI = OPS + NP + NLOOP;
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
constexpr int NCLASS = NP + 2 * NLOOP + NE;
struct RegSrc {
  short kind[NCLASS][NREG];
  int value[NCLASS][NREG];
};
TG_HD int row_class(int rho, int NB) {
  if (rho < NP) return rho;
  if (rho < NP + NLOOP * NB) return NP + ((rho - NP) / NLOOP ? NLOOP : 0) + (rho - NP) % NLOOP;
  return NP + 2 * NLOOP + (rho - NP - NLOOP * NB);
}
inline void make_reg_src(int NB, RegSrc& t) {
  const int L = NP + NLOOP * NB + NE;
  for (int c = 0; c < NCLASS; ++c) {
    // a representative row of the class (NB >= 32: step 1 exists)
    const int rho = c < NP ? c : (c < NP + NLOOP ? c : (c < NP + 2 * NLOOP ? c : NP + NLOOP * NB + (c - NP - 2 * NLOOP)));
    for (int r = 0; r < NREG; ++r) {
      const int src = reg_source(rho, r, NB);
      if (src < 0) t.kind[c][r] = 2, t.value[c][r] = src + L;
      else if (src < NP || c < NP) t.kind[c][r] = 1, t.value[c][r] = src;
      else t.kind[c][r] = 0, t.value[c][r] = rho - src;
    }
  }
}
TG_HD long long reg_source_of(const RegSrc& t, int rho, int r, int NB) {
  const int c = row_class(rho, NB);
  const int v = t.value[c][r];
  const int L = NP + NLOOP * NB + NE;
  return t.kind[c][r] == 0 ? rho - v : (t.kind[c][r] == 1 ? v : (long long)v - L);
}
TG_HD unsigned limb16(const uint64_t* v, int i) { return (unsigned)((v[i >> 2] >> (16 * (i & 3))) & 0xFFFF); }

// phase 2: the cells of trace row `row` (instance `inst`, position `rho`).  vals = the side buffer of ALL instances ([inst * L + rho]);
// sigs[inst] for inst < nsig, the filler signature (A = B, S = h = 0) above.  look(limb) once per looked-up limb (Z, Q, W columns).
template <class Put, class Look>
TG_HD void row(const Cols& c, const RegSrc& rsrc, const RowVals* vals, const Sig* sigs, int nsig, const Sig& filler, size_t rowi, Put put, Look look) {
  const int L = c.L, NB = c.NB;
  const size_t inst = rowi / L;
  const int rho = (int)(rowi % L);
  const int rt = row_type(rho, NB);
  const Op& op = OPS[rt];
  const RowVals& rv = vals[rowi];
  for (int t = 0; t < NT; ++t) put(c.RT + t, (uint64_t)(t == rt));
  // registers
  uint64_t ev[4] = {0, 0, 0, 0};
  for (int r = 0; r < NREG; ++r) {
    const long long src = reg_source_of(rsrc, rho, r, NB);
    uint64_t v[4] = {0, 0, 0, 0};
    if (src >= 0 || inst > 0) {
      const RowVals& s = vals[(size_t)((long long)(inst * L) + src)];
      for (int k = 0; k < 4; ++k) v[k] = s.z[k];
    }
    for (int i = 0; i < NL; ++i) put(c.REG + NL * r + i, (uint64_t)limb16(v, i));
    if (!op.free_ && op.e == r)
      for (int k = 0; k < 4; ++k) ev[k] = v[k];
  }
  // X Y Z Q limbs and the carries of  X(t) Y(t) + E(t) - Z(t) - Q(t) P(t) = (t - 2^16) W(t)
  long long d[2 * NL - 1];
  for (int k = 0; k < 2 * NL - 1; ++k) d[k] = 0;
  unsigned xl[NL], yl[NL], zl[NL], ql[NL];
  for (int i = 0; i < NL; ++i) {
    xl[i] = limb16(rv.x, i), yl[i] = limb16(rv.y, i), zl[i] = limb16(rv.z, i), ql[i] = limb16(rv.q, i);
    put(c.X + i, (uint64_t)xl[i]), put(c.Y + i, (uint64_t)yl[i]), put(c.Z + i, (uint64_t)zl[i]), put(c.Q + i, (uint64_t)ql[i]);
    look(zl[i]);
  }
  for (int i = 0; i < NL; ++i) look(ql[i]);
  if (!op.free_) {
    for (int i = 0; i < NL; ++i)
      for (int j = 0; j < NL; ++j) {
        const long long pj = j == 0 ? 0xFFED : (j == NL - 1 ? 0x7FFF : 0xFFFF);
        d[i + j] += (long long)xl[i] * yl[j] - (long long)ql[i] * pj;
      }
    for (int i = 0; i < NL; ++i) d[i] += (long long)limb16(ev, i) - (long long)zl[i];
  }
  long long prev = 0;
  for (int k = 0; k < 2 * NL - 1; ++k) {
    const long long t = prev - d[k];
    prev = t >> LB;                                      // exact: the low 16 bits of t are zero when the relation holds
    if (k <= 2 * NL - 3) {
      const long long off = prev + (1ll << (2 * LB - 1));
      const unsigned w0 = (unsigned)(off & 0xFFFF), w1 = (unsigned)(off >> LB);
      put(c.W + 2 * k, (uint64_t)w0), put(c.W + 2 * k + 1, (uint64_t)w1);
      look(w0), look(w1);
    }
  }
  // scalar bookkeeping
  const int in_loop_or_after = rho >= NP;
  int step = in_loop_or_after ? (rho - NP) / NLOOP : 0;
  if (step > NB - 1) step = NB - 1;
  const bool after = rho >= NP + NLOOP * NB;
  const Sig& me = (int)inst < nsig ? sigs[inst] : filler;
  const Sig& before = inst > 0 ? ((int)(inst - 1) < nsig ? sigs[inst - 1] : filler) : me;
  for (int s = 0; s < 2; ++s) {
    const uint64_t* mine = s ? me.h : me.s;
    const uint64_t* theirs = s ? before.h : before.s;
    const uint64_t cur = (uint64_t)scalar_bit(mine, NB, step);
    const uint64_t prev_last = inst > 0 ? (uint64_t)scalar_bit(theirs, NB, NB - 1) : (uint64_t)scalar_bit(mine, NB, 0);
    put(c.BIT + s, rho < NP ? prev_last : cur);
    // the scalar's 32-bit words, most significant first; kacc = the bits of the current word so far
    const int wj = step / 32, hi_bit = NB - 32 * wj;      // word j covers bits [hi_bit - 32, hi_bit)
    const int lo_bit = hi_bit - 32;
    const uint64_t word = (mine[lo_bit >> 6] >> (lo_bit & 63)) & 0xFFFFFFFFull;
    const uint64_t kacc = word >> (31 - step % 32);
    put(c.KACC + s, rho < NP ? prev_last : (after ? cur : kacc));
    for (int j = 0; j < c.NW; ++j) {
      const int lb = NB - 32 * (j + 1);
      put(c.SW + s * c.NW + j, (mine[lb >> 6] >> (lb & 63)) & 0xFFFFFFFFull);
    }
  }
  const int pos = after ? 0 : step % 32, word = after ? 0 : step / 32;
  for (int i = 0; i < 32; ++i) put(c.POS + i, (uint64_t)(i == pos));
  for (int j = 0; j < c.NW; ++j) put(c.J + j, (uint64_t)(j == word));
  const bool adv = rt == NP + NLOOP - 1;
  const bool bnd = adv && step % 32 == 31;
  put(c.BND, (uint64_t)bnd);
  put(c.FIN, (uint64_t)(bnd && step / 32 == c.NW - 1));
  put(c.ACT, (uint64_t)((int)inst < nsig));
  put(c.TBL, (uint64_t)(rowi & 0xFFFF));
  put(c.MULT, (uint64_t)0);
}

}  // namespace ed
}  // namespace tg
