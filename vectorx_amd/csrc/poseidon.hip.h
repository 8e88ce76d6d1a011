// Poseidon-Goldilocks permutation (width 12, x^7, 8 full + 22 partial rounds) for gfx950.
// Replaces plonky2::hash::poseidon / poseidon_goldilocks and the AVX2/NEON hand-scheduled forms
// (plonky2 v0.2.0, un-vendored: /root/reference/Cargo.lock:4848-4905; algorithm per SURVEY.md A.2).
//
// This path is bound by the integer ALU, not HBM (measured: tools/ubench_int.hip — on gfx950 every
// multiply-class / VOP3 instruction, v_mad_u64_u32 included, issues at ~4.4-5.2 cycles per wavefront, a
// plain v_add_u32 at ~2.8), so the design minimises the INSTRUCTION COUNT per permutation:
//  * lanes are kept as arbitrary u64 representatives (not canonical) between operations; one
//    conditional subtraction at the very end canonicalises;
//  * the 22 partial rounds use the "fast" sparse form (one dense 11x11 matrix up front, then per round a
//    12-term dot product and a rank-one update of the 11 passive lanes instead of a dense 12x12 MDS) — constants
//    re-derived and proven equivalent in tools/gen_poseidon_fast_constants.py (the role of upstream's
//    FAST_PARTIAL_*) — and run in BLOCKS: the passive lanes are only brought up to date once per block;
//  * every dot product with full-size constants is carry-free (constants pre-split into 22-bit limbs, one
//    v_mad_u64_u32 per partial product into plain 64-bit accumulators, one recombination per dot product);
//  * the dense MDS of the 8 full rounds multiplies the low/high 32-bit halves by the <2^6 circulant
//    entries with v_mad_u64_u32 (a 32x32+64 multiply-accumulate in ONE instruction) and folds once.
// The 12-lane state lives in VGPRs; round constants sit in constant memory and, because the loops
// are unrolled, are fetched with scalar loads shared by the whole wavefront.
#pragma once
#include "goldilocks.hip.h"
#include "poseidon_constants.h"
#include "poseidon_fast_constants.h"

__constant__ u64 POSEIDON_RC[VX_POSEIDON_N_ROUND_CONSTANTS] = VX_POSEIDON_ROUND_CONSTANTS_INIT;
__constant__ u64 POSEIDON_FAST_FIRST[12] = VX_FAST_PARTIAL_FIRST_ROUND_CONSTANT_INIT;
__constant__ u64 POSEIDON_FAST_K[22] = VX_FAST_PARTIAL_ROUND_CONSTANTS_INIT;

#define POSEIDON_WIDTH 12
#define POSEIDON_RATE 8

// ---- non-canonical ("nc") arithmetic: values are any u64 congruent to the field element ----------
// a: any u64, b: CANONICAL (< p)  ->  a + b (nc).  One carry fix suffices because b <= 2^64 - 2^32.
GLD u64 gl_add_nc_c(u64 a, u64 b) {
  u64 s = a + b;
  if (s < a) s += GL_EPS;
  return s;
}
// a: any u64, b: CANONICAL  ->  a - b (nc).  On a borrow the wrapped difference is >= 2^64 - (p-1) = EPS, so one fix suffices.
GLD u64 gl_sub_nc_c(u64 a, u64 b) {
  u64 d = a - b;
  if (a < b) d -= GL_EPS;
  return d;
}
GLD u64 poseidon_sbox_nc(u64 x) {
  const u64 x2 = gl_mul_nc(x, x), x4 = gl_mul_nc(x2, x2), x3 = gl_mul_nc(x, x2);
  return gl_mul_nc(x3, x4);
}
GLD u64 poseidon_sbox(u64 x) { return gl_canon(poseidon_sbox_nc(x)); }

// al + ah * 2^32 (mod p) for al, ah < 2^42 — the fold at the end of an MDS row — as SOME u64 representative.
//   ah * 2^32 = ah_lo * 2^32 + ah_hi * 2^64 = ah_lo * 2^32 + ah_hi * EPS  (mod p);  X = ah_hi * EPS + al < 2^43 cannot
//   overflow (one v_mad_u64_u32);  adding ah_lo * 2^32 only touches the high dword; if that carries, the wrapped
//   value is < 2^43 and the 2^64 it lost is worth EPS, which cannot overflow again.  5 VALU instructions, against
//   ~15 for the generic 128-bit route (shift, add, carry detect, gl_reduce128_nc).
GLD u64 mds_fold_nc(u64 al, u64 ah) {
  const u32 eps = 0xFFFFFFFFu;
  u64 X, cX;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(X), "=s"(cX) : "v"((u32)(ah >> 32)), "v"(eps), "v"(al));
  u32 h, d, l2, h2;
  u64 c, c2, c3;
  asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(h), "=s"(c) : "v"((u32)(X >> 32)), "v"((u32)ah));
  asm("v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(d) : "s"(c));
  asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(l2), "=s"(c2) : "v"((u32)X), "v"(d));
  asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(h2), "=s"(c3) : "v"(h), "s"(c2));
  return gl_pack(l2, h2);
}

// Dense MDS layer: out[r] = sum_i CIRC[i] * v[(i+r) % 12] + DIAG[r]*v[r]; entries < 2^6, so the low and
// high 32-bit halves are accumulated separately (< 2^42 each) and folded once:  lo + hi*2^32 (mod p).
GLD void poseidon_mds_nc(u64 (&s)[12]) {
  const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  u32 lo[12], hi[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    lo[i] = (u32)s[i];
    hi[i] = (u32)(s[i] >> 32);
  }
#pragma unroll
  for (int r = 0; r < 12; ++r) {
    u64 al = 0, ah = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      al += (u64)C[i] * lo[(i + r) % 12];
      ah += (u64)C[i] * hi[(i + r) % 12];
    }
    if (r == 0) {
      al += (u64)8 * lo[0];
      ah += (u64)8 * hi[0];
    }
    s[r] = mds_fold_nc(al, ah);  // al, ah < 2^42
  }
}
// canonical-in / canonical-out wrapper used by the quotient kernel's PoseidonGate evaluation
GLD void poseidon_mds(u64 (&s)[12]) {
  poseidon_mds_nc(s);
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = gl_canon(s[i]);
}

// ---- dot products with full-size constants, carry-free -----------------------------------------------------------
// sum_i a_i * b_i with a_i any u64 and b_i a 64-bit constant.  a = a0 + 2^32 a1, so a*b = a0*b + a1*b' with the
// second constant b' = 2^32 b mod p; both constants are pre-split into limbs of 22 + 22 + 20 bits.  Limb k of b
// meets a0 and limb k of b' meets a1 in the SAME accumulator S_k: <= 26 products below 2^54 each, so a plain 64-bit
// register and ONE v_mad_u64_u32 per product suffice — no carries anywhere — and the result is
// S_0 + 2^22 S_1 + 2^44 S_2 < 2^104, recombined and reduced once per dot product.  6 multiply-adds per term instead
// of a 64x64->128 product (4 multiplies + 4 pair-forming moves + 1 add) followed by a 5-instruction carry chain.
struct Limbs3x2 {
  u32 lo[3];  // limbs of b      (multiply the low  half of the state element)
  u32 hi[3];  // limbs of 2^32 b (multiply the high half)
};
template <int R, int C>
struct Limbs3Table {
  Limbs3x2 v[R][C];
};
constexpr u64 gl_mul_2_32_const(u64 b) { return (u64)((((unsigned __int128)b) << 32) % (unsigned __int128)GL_P); }
template <int R, int C>
constexpr Limbs3Table<R, C> make_limbs3(const u64 (&raw)[R][C]) {
  Limbs3Table<R, C> t{};
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c) {
      const u64 b = raw[r][c] % GL_P, bh = gl_mul_2_32_const(b);
      t.v[r][c].lo[0] = (u32)(b & 0x3FFFFFu);
      t.v[r][c].lo[1] = (u32)((b >> 22) & 0x3FFFFFu);
      t.v[r][c].lo[2] = (u32)(b >> 44);
      t.v[r][c].hi[0] = (u32)(bh & 0x3FFFFFu);
      t.v[r][c].hi[1] = (u32)((bh >> 22) & 0x3FFFFFu);
      t.v[r][c].hi[2] = (u32)(bh >> 44);
    }
  return t;
}
constexpr u64 POSEIDON_FAST_INIT_RAW[11][11] = VX_FAST_PARTIAL_INITIAL_MATRIX_INIT;
constexpr u64 POSEIDON_FAST_W_HATS_RAW[22][11] = VX_FAST_PARTIAL_W_HATS_INIT;
__constant__ Limbs3Table<11, 11> POSEIDON_FAST_INIT3 = make_limbs3<11, 11>(POSEIDON_FAST_INIT_RAW);
__constant__ Limbs3Table<22, 11> POSEIDON_FAST_W_HATS3 = make_limbs3<22, 11>(POSEIDON_FAST_W_HATS_RAW);

// Blocked partial rounds: inside a block of PB rounds the 11 passive lanes are NOT updated; round r's dot product is
// taken on the block's starting state plus cross terms y_q * KK[r][q] (y_q = the S-box outputs of the block's earlier
// rounds, KK[r][q] = sum_i w_hat_r[i] v_q[i]), and the lanes are brought up to date once per block with an 11-term
// dot product each.  That trades the 11 multiply-add-reduce per round for carry-free multiply-adds.
#ifndef POSEIDON_PB
#define POSEIDON_PB 5
#endif
constexpr u64 gl_mulmod_const(u64 a, u64 b) { return (u64)(((unsigned __int128)a * b) % (unsigned __int128)GL_P); }
constexpr u64 POSEIDON_FAST_VS_RAW[22][11] = VX_FAST_PARTIAL_VS_INIT;
struct PoseidonBlockTables {
  Limbs3x2 kk[22][POSEIDON_PB];  // kk[r][q]: cross term of round r with the q-th round of its block (q < r - block start)
  Limbs3x2 vs[22][11];           // v_r[i]
};
constexpr Limbs3x2 make_limbs3x2(u64 b) {
  Limbs3x2 t{};
  const u64 bh = gl_mul_2_32_const(b);
  t.lo[0] = (u32)(b & 0x3FFFFFu), t.lo[1] = (u32)((b >> 22) & 0x3FFFFFu), t.lo[2] = (u32)(b >> 44);
  t.hi[0] = (u32)(bh & 0x3FFFFFu), t.hi[1] = (u32)((bh >> 22) & 0x3FFFFFu), t.hi[2] = (u32)(bh >> 44);
  return t;
}
constexpr PoseidonBlockTables make_block_tables() {
  PoseidonBlockTables t{};
  for (int r = 0; r < 22; ++r) {
    const int r0 = (r / POSEIDON_PB) * POSEIDON_PB;
    for (int q = 0; q < POSEIDON_PB; ++q) {
      u64 acc = 0;
      if (r0 + q < r)
        for (int i = 0; i < 11; ++i) {
          const u64 term = gl_mulmod_const(POSEIDON_FAST_W_HATS_RAW[r][i] % GL_P, POSEIDON_FAST_VS_RAW[r0 + q][i] % GL_P);
          acc = (u64)(((unsigned __int128)acc + term) % (unsigned __int128)GL_P);
        }
      t.kk[r][q] = make_limbs3x2(acc);
    }
    for (int i = 0; i < 11; ++i) t.vs[r][i] = make_limbs3x2(POSEIDON_FAST_VS_RAW[r][i] % GL_P);
  }
  return t;
}
__constant__ PoseidonBlockTables POSEIDON_BLOCK = make_block_tables();

struct dot3 {
  u64 s0, s1, s2;
};
GLD void dot3_mac(dot3& D, u64 a, const Limbs3x2& b) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32);
  D.s0 += (u64)a0 * b.lo[0];
  D.s1 += (u64)a0 * b.lo[1];
  D.s2 += (u64)a0 * b.lo[2];
  D.s0 += (u64)a1 * b.hi[0];
  D.s1 += (u64)a1 * b.hi[1];
  D.s2 += (u64)a1 * b.hi[2];
}
GLD u64 dot3_reduce_nc(const dot3& D) {
  typedef unsigned __int128 u128;
  const u128 V = (u128)D.s0 + ((u128)D.s1 << 22) + ((u128)D.s2 << 44);  // < 2^104
  return gl_reduce128_nc((u64)V, (u64)(V >> 64));
}
GLD u64 dot3_reduce_add_nc(const dot3& D, u64 addend) {
  typedef unsigned __int128 u128;
  const u128 V = (u128)D.s0 + ((u128)D.s1 << 22) + ((u128)D.s2 << 44) + (u128)addend;
  return gl_reduce128_nc((u64)V, (u64)(V >> 64));
}

// The 22 partial rounds (fast form, blocks of POSEIDON_PB).  `lane0(r, x)` is handed the S-box INPUT x of partial round r
// and returns the value that actually goes through the S-box: the identity for the permutation.  (A PoseidonGate
// evaluation could pass the wire the gate constrains to equal x — the recurrences are linear in everything but the
// S-box outputs — and that is byte-identical, but in the quotient kernel the extra live registers cost more than the
// dense MDS it saves: 23.7 -> 27.1 ms, so the gate keeps the naive rounds.)
template <class F>
GLD void poseidon_partial_rounds_nc(u64 (&s)[12], F&& lane0) {
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = gl_add_nc_c(s[i], POSEIDON_FAST_FIRST[i]);
  {
    u64 t[11];
#pragma unroll 1
    for (int r = 0; r < 11; ++r) {
      dot3 D = {0, 0, 0};
#pragma unroll
      for (int c = 0; c < 11; ++c) dot3_mac(D, s[1 + c], POSEIDON_FAST_INIT3.v[r][c]);
      t[r] = dot3_reduce_nc(D);
    }
#pragma unroll
    for (int r = 0; r < 11; ++r) s[1 + r] = t[r];
  }
#pragma unroll 1
  for (int r0 = 0; r0 < 22; r0 += POSEIDON_PB) {
    const int nb = 22 - r0 < POSEIDON_PB ? 22 - r0 : POSEIDON_PB;  // the last block may be shorter
    u64 y[POSEIDON_PB];
#pragma unroll
    for (int j = 0; j < POSEIDON_PB; ++j) {
      y[j] = 0;
      if (j < nb) {
        const u64 yj = gl_add_nc_c(poseidon_sbox_nc(lane0(r0 + j, s[0])), POSEIDON_FAST_K[r0 + j]);
        y[j] = yj;
        // M[0][0] = CIRC[0] + DIAG[0] = 25:  25 y = 25 lo(y) + 2^22 * (25 * 2^10) hi(y)
        dot3 D = {(u64)(u32)yj * 25u, (u64)(u32)(yj >> 32) * 25600u, 0};
#pragma unroll
        for (int i = 0; i < 11; ++i) dot3_mac(D, s[1 + i], POSEIDON_FAST_W_HATS3.v[r0 + j][i]);
#pragma unroll
        for (int q = 0; q < j; ++q) dot3_mac(D, y[q], POSEIDON_BLOCK.kk[r0 + j][q]);
        s[0] = dot3_reduce_nc(D);
      }
    }
#pragma unroll
    for (int i = 0; i < 11; ++i) {
      dot3 D = {0, 0, 0};
#pragma unroll
      for (int j = 0; j < POSEIDON_PB; ++j)
        if (j < nb) dot3_mac(D, y[j], POSEIDON_BLOCK.vs[r0 + j][i]);
      s[1 + i] = dot3_reduce_add_nc(D, s[1 + i]);
    }
  }
}

// Permutation on arbitrary-u64 lanes; outputs are arbitrary u64 representatives (NOT canonical).
GLD void poseidon_permute_nc(u64 (&s)[12]) {
#pragma unroll 1
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = poseidon_sbox_nc(gl_add_nc_c(s[i], POSEIDON_RC[r * 12 + i]));
    poseidon_mds_nc(s);
  }
  poseidon_partial_rounds_nc(s, [](int, u64 x) { return x; });
#pragma unroll 1
  for (int r = 26; r < 30; ++r) {
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = poseidon_sbox_nc(gl_add_nc_c(s[i], POSEIDON_RC[r * 12 + i]));
    poseidon_mds_nc(s);
  }
}

GLD void poseidon_permute(u64 (&s)[12]) {
  poseidon_permute_nc(s);
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = gl_canon(s[i]);
}

// PoseidonHash::two_to_one (hashing.rs): state = [l0..l3, r0..r3, 0,0,0,0] -> permute -> [0..4]
GLD void poseidon_two_to_one(const u64* l, const u64* r, u64* out) {
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s[i] = l[i];
    s[4 + i] = r[i];
    s[8 + i] = 0;
  }
  poseidon_permute_nc(s);
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = gl_canon(s[i]);
}

// ------------------------------------------------------------------------------------------------
// Lane-cooperative permutation for LATENCY-bound launches (the small levels of a Merkle tree, small FRI layers):
// 16 lanes per state, lane g < 12 holds s[g] (lanes 12..15 of a group shadow lanes 0..3 and are ignored), so one
// wavefront carries 4 states and a permutation is ~30 x (S-box + 12 cross-lane multiply-adds) = ~4k instructions
// deep instead of ~22k.  Below ~16k states a launch of the one-thread-per-state form costs one full permutation
// latency (~60 us) whatever its size; this form costs ~1/5 of that.  Plain (non-"fast") partial rounds: every lane
// adds its round constant, lane 0 takes the S-box, then the same cross-lane MDS as in the full rounds.
// ------------------------------------------------------------------------------------------------
GLD u64 poseidon_coop_mds_nc(u64 v, int g, int group_base) {
  const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  const u32 lo = (u32)v, hi = (u32)(v >> 32);
  u64 al = 0, ah = 0;
  int idx = g < 12 ? g : g - 12;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const int src = group_base + idx;
    al += (u64)C[i] * (u32)__shfl((int)lo, src, 64);
    ah += (u64)C[i] * (u32)__shfl((int)hi, src, 64);
    idx = idx == 11 ? 0 : idx + 1;
  }
  if (g == 0) {
    al += (u64)8 * lo;
    ah += (u64)8 * hi;
  }
  return mds_fold_nc(al, ah);
}
// v: this lane's state element (any u64 representative); returns the permuted element (NOT canonical).
GLD u64 poseidon_permute_coop_nc(u64 v, int g, int group_base) {
  const int gi = g < 12 ? g : g - 12;
#pragma unroll 1
  for (int r = 0; r < 30; ++r) {
    v = gl_add_nc_c(v, POSEIDON_RC[r * 12 + gi]);
    const bool full = r < 4 || r >= 26;
    if (full || g == 0) v = poseidon_sbox_nc(v);
    v = poseidon_coop_mds_nc(v, g, group_base);
  }
  return v;
}
