// Poseidon-Goldilocks permutation (width 12, x^7, 8 full + 22 partial rounds) for gfx950.
// Replaces plonky2::hash::poseidon / poseidon_goldilocks and the AVX2/NEON hand-scheduled forms
// (plonky2 v0.2.0, un-vendored: /root/reference/Cargo.lock:4848-4905; algorithm per SURVEY.md A.2).
// The 12-lane state lives in VGPRs; round constants sit in constant memory and, because every
// loop is fully unrolled, are fetched with scalar loads shared by the whole wavefront.
#pragma once
#include "goldilocks.hip.h"
#include "poseidon_constants.h"

__constant__ u64 POSEIDON_RC[VX_POSEIDON_N_ROUND_CONSTANTS] = VX_POSEIDON_ROUND_CONSTANTS_INIT;

#define POSEIDON_WIDTH 12
#define POSEIDON_RATE 8

GLD u64 poseidon_sbox(u64 x) {
  u64 x2 = gl_sqr(x), x4 = gl_sqr(x2), x3 = gl_mul(x, x2);
  return gl_mul(x3, x4);
}

// MDS layer: out[r] = sum_i CIRC[i] * v[(i+r) % 12] + DIAG[r]*v[r].  All constants are < 2^6, so
// each output is accumulated separately over the low and high 32-bit halves of the state (sums
// stay < 2^42) with 32x32->64 multiply-adds and folded once:  lo + hi*2^32  (mod p).
GLD void poseidon_mds(u64 (&s)[12]) {
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
    // value = al + ah*2^32, ah < 2^42:  ah*2^32 = (ah>>32)*2^64 + (ah&M)*2^32
    s[r] = gl_reduce128(al + (ah << 32), (ah >> 32) + ((al + (ah << 32)) < al ? 1 : 0));
  }
}

GLD void poseidon_permute(u64 (&s)[12]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = poseidon_sbox(gl_add(s[i], POSEIDON_RC[r * 12 + i]));
    poseidon_mds(s);
  }
#pragma unroll 1
  for (int r = 4; r < 26; ++r) {
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = gl_add(s[i], POSEIDON_RC[r * 12 + i]);
    s[0] = poseidon_sbox(s[0]);
    poseidon_mds(s);
  }
#pragma unroll
  for (int r = 26; r < 30; ++r) {
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = poseidon_sbox(gl_add(s[i], POSEIDON_RC[r * 12 + i]));
    poseidon_mds(s);
  }
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
  poseidon_permute(s);
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = s[i];
}
