// Second-generation NTT pass for the hot shapes (tiles of 8192 elements, 512 threads): same math, same
// data layout and same plan as ntt.hip.h's generic pass — which stays as the path for small transforms — but
// built to issue fewer instructions per element (this kernel is integer-ALU bound):
//  * the first round of a tile is fed straight from global memory and, for strided passes, the last round
//    stores straight to global memory: two LDS round trips and two barriers fewer per pass;
//  * rounds are TWIDDLE-FACTORED radix-16/8/4 DFTs held in registers: the butterflies inside a radix-2^e
//    block only need 2^e-th roots of unity, and in the Goldilocks field those are powers of two
//    (2^96 = -1; plonky2's w_64 = 2^39, so w_16 = 2^156 = -2^60, w_8 = -2^24, w_4 = 2^48): a shift and a
//    fold instead of a general multiply, no twiddle loads; one general multiply per element per round
//    (w_R^(k*base_low), from an LDS table) replaces the e/2 general multiplies of e radix-2 stages;
//  * strided / bit-reversed-input / prescale / inverse are template parameters, so the index arithmetic of
//    each variant is straight-line code.
// Replaces plonky2_field::fft (field/src/fft.rs) exactly like ntt.hip.h; results are bit-identical to it
// (tests/test_gpu_parity.py compares every length and kind against the oracle).
#pragma once
#include "ntt.hip.h"

#define NTT2_THREADS 512
#define NTT2_TILE_LOG 13

// 2^S mod p for 0 <= S < 192 (2^96 = -1)
__host__ __device__ constexpr u64 gl_pow2_const(int S) {
  S %= 192;
  bool neg = S >= 96;
  if (neg) S -= 96;
  // 2^S for S < 96:  S < 64 -> plain;  64 <= S < 96 -> 2^64 * 2^(S-64) = (2^32-1) * 2^(S-64)  (< 2^64, already reduced)
  u64 v = S < 64 ? ((u64)1 << S) : (((u64)0xFFFFFFFFULL) << (S - 64));
  // S < 64: 2^S < p except it is always < 2^64 - 2^32 + 1 for S <= 63;   S >= 64: (2^32-1)*2^k with k < 32 < p
  return neg ? (GL_P - v) : v;
}

// x canonical -> x * 2^S canonical, 0 <= S < 96
template <int S>
GLD u64 gl_mul_2exp(u64 x) {
  if constexpr (S == 0) {
    return x;
  } else if constexpr (S < 32) {
    // (x << S) = hi*2^64 + lo with hi < 2^32:  lo + hi*EPS, one carry fold
    const u64 lo = x << S;
    const u32 hi = (u32)(x >> (64 - S));
    u64 T, cT;
    const u32 eps = 0xFFFFFFFFu;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(T), "=s"(cT) : "v"(hi), "v"(eps), "v"(lo));
    // carry: wrapped T < hi*EPS < 2^63, so T + EPS is the canonical value;  no carry: T < 2^64 may exceed p, and
    // T - p = T + EPS (mod 2^64): one lane mask, one addend
    const u64 m = cT | __builtin_amdgcn_uicmpl(T, (u64)GL_P, 35 /* ICMP_UGE */);
    u32 d0;
    asm("v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(d0) : "s"(m));
    return gl_add32(T, d0);
  } else if constexpr (S < 64) {
    return gl_reduce128(x << S, x >> (64 - S));
  } else {
#ifdef VX_NTT2_NO_SHIFT64
    return gl_mul(x, gl_pow2_const(S));
#else
    // S = 64 + k:  x 2^k = Yh 2^32 + y0  (y0 = its low 32 bits, Yh = x >> (32 - k) < 2^63)  =>  x 2^S = y0 2^64 + Yh 2^96 = y0 EPS - Yh.
    // y0 EPS <= (2^32 - 1)^2 < p and Yh < p, so one canonical subtraction finishes it: 8 instructions against the 17 of a general multiply
    // by the constant (round 4; two of the eight first-stage twiddles of a radix-16 DFT and one of the four of its second stage land here)
    constexpr int K = S - 64;
    const u32 y0 = (u32)x << K;
    const u64 Yh = x >> (32 - K);
    u64 T, cT;
    const u32 eps = 0xFFFFFFFFu;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(T), "=s"(cT) : "v"(y0), "v"(eps));
    return gl_sub(T, Yh);
#endif
  }
}

// DIF butterfly with twiddle 2^E (E mod 192):  a' = a + b,  b' = (a - b) * 2^E
template <int E>
GLD void bfly2(u64& a, u64& b) {
  constexpr int EE = ((E % 192) + 192) % 192;
  const u64 s = gl_add(a, b);
  u64 d;
  if constexpr (EE == 0)
    d = gl_sub(a, b);
  else if constexpr (EE < 96)
    d = gl_mul_2exp<EE>(gl_sub(a, b));
  else
    d = gl_mul_2exp<EE - 96>(gl_sub(b, a));
  a = s;
  b = d;
}
#define NTT2_BF(E, i, j) bfly2<(INV ? (192 - (E)) : (E))>(x[i], x[j]);

// In-register DIF DFTs with plonky2's roots; output k sits at index rev(k).
template <bool INV>
GLD void dft16(u64 (&x)[16]) {
  NTT2_BF(0, 0, 8) NTT2_BF(156, 1, 9) NTT2_BF(120, 2, 10) NTT2_BF(84, 3, 11)
  NTT2_BF(48, 4, 12) NTT2_BF(12, 5, 13) NTT2_BF(168, 6, 14) NTT2_BF(132, 7, 15)
#pragma unroll
  for (int o = 0; o < 16; o += 8) {
    NTT2_BF(0, o + 0, o + 4) NTT2_BF(120, o + 1, o + 5) NTT2_BF(48, o + 2, o + 6) NTT2_BF(168, o + 3, o + 7)
  }
#pragma unroll
  for (int o = 0; o < 16; o += 4) {
    NTT2_BF(0, o + 0, o + 2) NTT2_BF(48, o + 1, o + 3)
  }
#pragma unroll
  for (int o = 0; o < 16; o += 2) {
    NTT2_BF(0, o, o + 1)
  }
}
template <bool INV>
GLD void dft8(u64 (&x)[8]) {
  NTT2_BF(0, 0, 4) NTT2_BF(120, 1, 5) NTT2_BF(48, 2, 6) NTT2_BF(168, 3, 7)
#pragma unroll
  for (int o = 0; o < 8; o += 4) {
    NTT2_BF(0, o + 0, o + 2) NTT2_BF(48, o + 1, o + 3)
  }
#pragma unroll
  for (int o = 0; o < 8; o += 2) {
    NTT2_BF(0, o, o + 1)
  }
}
template <bool INV>
GLD void dft4(u64 (&x)[4]) {
  NTT2_BF(0, 0, 2) NTT2_BF(48, 1, 3)
  NTT2_BF(0, 0, 1) NTT2_BF(0, 2, 3)
}
template <bool INV>
GLD void dft2(u64 (&x)[2]) {
  NTT2_BF(0, 0, 1)
}
template <int E, bool INV>
GLD void dft_regs(u64 (&x)[1 << E]) {
  if constexpr (E == 4) dft16<INV>(x);
  if constexpr (E == 3) dft8<INV>(x);
  if constexpr (E == 2) dft4<INV>(x);
  if constexpr (E == 1) dft2<INV>(x);
}
template <int E>
GLD constexpr u32 rev_c(u32 q) {
  u32 r = 0;
  for (int i = 0; i < E; ++i) r |= ((q >> i) & 1u) << (E - 1 - i);
  return r;
}

// The LDS twiddle table holds w_R^(+-j) for j < min(R, 1024); for R = 2048 the upper half is obtained by
// negation (w^(j+R/2) = -w^j) so that two 8192-element tiles still fit in a CU's 160 KiB of LDS.
template <int R_LOG>
GLD constexpr u32 w_table_len() { return R_LOG > 10 ? (1u << (R_LOG - 1)) : (1u << R_LOG); }

// y[q'] *= w_R^(rev(q') * base_low * 2^(R_LOG-LO-E))   (q' != 0; skipped entirely when LO == 0)
template <int R_LOG, int LO, int E>
GLD void round_twiddles(u64 (&x)[1 << E], const u64* __restrict__ W, u32 base_low) {
  if constexpr (LO > 0) {
#pragma unroll
    for (int q = 1; q < (1 << E); ++q) {
      const u32 k = rev_c<E>((u32)q);
      const u32 idx = ((k * base_low) << (R_LOG - LO - E)) & ((1u << R_LOG) - 1);
      if constexpr (R_LOG > 10) {
        // w^(j + R/2) = -w^j: the sign goes onto the TABLE value (never zero, so p - w is canonical without a zero test)
        u64 w = W[idx & ((1u << (R_LOG - 1)) - 1)];
        if (idx >> (R_LOG - 1)) w = GL_P - w;
        x[q] = gl_mul(x[q], w);
      } else {
        x[q] = gl_mul(x[q], W[idx]);
      }
    }
  }
}

struct Ntt2Params {
  const u64* in;
  u64* out;
  size_t in_col_stride, out_col_stride, in_z_stride, out_z_stride;
  int log_n, b_lo;
  const u64 *root_lo, *root_hi;
  const u64* pre;  // [z][2^(log_n-pre_bits) + 2^pre_bits] or null
  const u64* pre_beta;  // [z][16]: shift_z^(q * n/16), behind the slices of `pre`
  int pre_bits;
  u64 post_scale;
  // nz_fold > 0: the coset (z) dimension is folded into blockIdx.x so that the nz blocks that read the SAME
  // coefficient tile (one per coset of an LDE) are dispatched back to back on the SAME XCD — workgroups go to XCDs
  // round-robin, so block id = ((tile / 8) * nz + z) * 8 + tile % 8 — and 7 of the 8 reads hit that XCD's L2.
  int nz_fold;
  // Full-size inter-pass twiddle table (round 4), or null: entry (m << b_lo) | l = w_span^(+-l * rev(m)) * post_scale for a strided pass of
  // 2^(b_lo + R_LOG) points.  Composing that factor from the two-level root tables costs one general multiply per element (16 VALU
  // instructions) on top of the multiply that applies it; read from the table it costs a load — which pays where the load hits L2: in a
  // coset LDE the nz blocks of a tile run back to back on one XCD and share the tile's 64 KB slice (vx_runtime.hip.h decides).
  const u64* tw_full;
};
// table builder (one launch per table and context: vx_runtime.hip.h caches them)
__global__ void ntt2_tw_table_kernel(u64* __restrict__ tab, int span_log, int r_log, int b_lo, int inv, u64 post_scale,
                                     const u64* __restrict__ root_lo, const u64* __restrict__ root_hi) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >> span_log) return;
  const u64 m = idx >> b_lo, l = idx & (((u64)1 << b_lo) - 1);
  const u64 ex = (l * (u64)bitrev32((u32)m, r_log)) & (((u64)1 << span_log) - 1);
  u32 e = (u32)(ex << (ROOT_TABLE_LOG - span_log));
  if (inv) e = ((1u << ROOT_TABLE_LOG) - e) & ((1u << ROOT_TABLE_LOG) - 1);
  u64 v = root_pow24(root_lo, root_hi, e);
  if (post_scale != 1) v = gl_mul(v, post_scale);
  tab[idx] = v;
}

// LDS index of tile element (m, t): strided tiles are [m][t], contiguous tiles are [t][m]
template <int R_LOG, bool STRIDED>
GLD u32 tile_idx(u32 m, u32 t) {
  constexpr int T_LOG = NTT2_TILE_LOG - R_LOG;
  return lds_pad(STRIDED ? ((m << T_LOG) | t) : ((t << R_LOG) | m));
}

// One LDS->LDS round over stage bits [LO, LO+E)
template <int R_LOG, int LO, int E, bool STRIDED, bool INV>
GLD void ntt2_round_lds(u64* __restrict__ tile, const u64* __restrict__ W) {
  constexpr int T_LOG = NTT2_TILE_LOG - R_LOG;
  constexpr u32 n_groups = 1u << (NTT2_TILE_LOG - E);
#pragma unroll
  for (u32 g0 = 0; g0 < n_groups; g0 += NTT2_THREADS) {
    const u32 g = g0 + threadIdx.x;
    u32 t, base_low, base_high;
    if constexpr (STRIDED) {
      t = g & ((1u << T_LOG) - 1);
      const u32 rest = g >> T_LOG;
      base_low = rest & ((1u << LO) - 1);
      base_high = rest >> LO;
    } else {
      base_low = g & ((1u << LO) - 1);
      const u32 rest = g >> LO;
      base_high = rest & ((1u << (R_LOG - E - LO)) - 1);
      t = rest >> (R_LOG - E - LO);
    }
    const u32 m0 = (base_high << (LO + E)) | base_low;
    u64 x[1 << E];
#pragma unroll
    for (int q = 0; q < (1 << E); ++q) x[q] = tile[tile_idx<R_LOG, STRIDED>(m0 | ((u32)q << LO), t)];
    dft_regs<E, INV>(x);
    round_twiddles<R_LOG, LO, E>(x, W, base_low);
#pragma unroll
    for (int q = 0; q < (1 << E); ++q) tile[tile_idx<R_LOG, STRIDED>(m0 | ((u32)q << LO), t)] = x[q];
  }
}

// Kernel: one tile of 2^13 elements per block.
//   STRIDED: stages [b_lo, b_lo+R_LOG) over T = 2^(13-R_LOG) adjacent low indices; else 2^(13-R_LOG) contiguous
//            chunks of 2^R_LOG (final pass, b_lo == 0).
//   IN_BITREV (first pass only): input stored in bit-reversed order.   PRE: prescale tables present.
template <int R_LOG, int E1, int E2, int E3, bool STRIDED, bool IN_BITREV, bool PRE, bool INV>
__global__ __launch_bounds__(NTT2_THREADS) void ntt2_pass_kernel(Ntt2Params p) {
  static_assert(E1 == 4 && E1 + E2 + E3 == R_LOG, "round plan");
  extern __shared__ __attribute__((aligned(16))) u64 smem[];
  constexpr int T_LOG = NTT2_TILE_LOG - R_LOG;
  constexpr u32 TILE = 1u << NTT2_TILE_LOG;
  u64* tile = smem;
  u64* W = smem + lds_pad(TILE) + 1;  // w_R^(+-j), j < R
  const int span_log = p.b_lo + R_LOG;

  for (u32 j = threadIdx.x; j < w_table_len<R_LOG>(); j += NTT2_THREADS) {
    u32 e = j << (ROOT_TABLE_LOG - R_LOG);
    if (INV) e = ((1u << ROOT_TABLE_LOG) - e) & ((1u << ROOT_TABLE_LOG) - 1);
    W[j] = root_pow24(p.root_lo, p.root_hi, e);
  }

  u32 bz = blockIdx.z;
  size_t tile_id = blockIdx.x;
  if (p.nz_fold > 0) {
    const u32 rest = blockIdx.x >> 3;
    bz = rest % (u32)p.nz_fold;
    tile_id = (size_t)(rest / (u32)p.nz_fold) * 8 + (blockIdx.x & 7);
  }
  size_t base;
  if constexpr (STRIDED) {
    const size_t tiles_per_span = (size_t)1 << (p.b_lo - T_LOG);
    const size_t H = tile_id / tiles_per_span;
    const size_t l0 = (tile_id % tiles_per_span) << T_LOG;
    base = (H << span_log) | l0;
  } else {
    base = tile_id << NTT2_TILE_LOG;
  }
  const u64* __restrict__ in = p.in + (size_t)blockIdx.y * p.in_col_stride + (size_t)bz * p.in_z_stride;
  u64* __restrict__ out = p.out + (size_t)blockIdx.y * p.out_col_stride + (size_t)bz * p.out_z_stride;

  // ---------------- round 1: stage bits [R_LOG-4, R_LOG), fed from global memory ----------------
  {
    constexpr int LO = R_LOG - E1;
    const u32 g = threadIdx.x;  // 2^(13-4) = 512 groups = one per thread
    u32 t, base_low;
    if constexpr (STRIDED) {
      t = g & ((1u << T_LOG) - 1);
      base_low = g >> T_LOG;
    } else {
      base_low = g & ((1u << LO) - 1);
      t = g >> LO;
    }
    u64 x[16];
    if constexpr (STRIDED && IN_BITREV) {
      // bit-reversed input: the 16 operands of this thread's radix-16 DFT (q = the top 4 bits of m) are the 16
      // CONSECUTIVE words at rev(base_low) * 16 of chunk t, in the order rev4(q): eight 16-byte loads instead of
      // sixteen 8-byte ones (x[q] and x[q + 8] are neighbours)
      const size_t l = (base & (((size_t)1 << p.b_lo) - 1)) + t;
      const size_t g0 = ((size_t)bitrev32((u32)l, p.b_lo) << R_LOG) | bitrev32(base_low, R_LOG);
      const ulonglong2* __restrict__ src = reinterpret_cast<const ulonglong2*>(in + g0);
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        const ulonglong2 v = src[h];                 // positions 2h, 2h+1 = rev4(q), rev4(q + 8) with q = rev3(h)
        const int q = (int)rev_c<3>((u32)h);
        // with a prescale the multiply that follows takes any u64 representative
        x[q] = PRE ? v.x : gl_canon(v.x);
        x[q + 8] = PRE ? v.y : gl_canon(v.y);
      }
      if constexpr (PRE) {
        // shift^j for j = ((base_low | q << LO) << b_lo) | l  =  shift^j0 * (shift^(n/16))^q: one composed table
        // product per thread, then the 16 per-coset factors, which are wave-uniform (scalar loads)
        const u64* pre = p.pre + (size_t)bz * (((size_t)1 << (p.log_n - p.pre_bits)) + ((size_t)1 << p.pre_bits));
        const u64* __restrict__ beta = p.pre_beta + (size_t)bz * 16;
        const size_t j0 = ((size_t)base_low << p.b_lo) | l;
        const u64 c0 = gl_mul_nc(pre[j0 >> p.pre_bits], pre[((size_t)1 << (p.log_n - p.pre_bits)) + (j0 & (((size_t)1 << p.pre_bits) - 1))]);
        x[0] = gl_mul(x[0], c0);
#pragma unroll
        for (int q = 1; q < 16; ++q) x[q] = gl_mul(x[q], gl_mul_nc(c0, beta[q]));
      }
    } else {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const u32 m = base_low | ((u32)q << LO);
      size_t gi, j;  // gi: where to read;  j: natural index inside the transform (for the prescale)
      if constexpr (STRIDED) {
        const size_t l = (base & (((size_t)1 << p.b_lo) - 1)) + t;
        j = ((size_t)m << p.b_lo) | l;
        if constexpr (IN_BITREV)
          gi = ((size_t)bitrev32((u32)l, p.b_lo) << R_LOG) | bitrev32(m, R_LOG);
        else
          gi = base + ((size_t)m << p.b_lo) + t;
      } else {
        j = base + (((size_t)t << R_LOG) | m);
        gi = IN_BITREV ? (size_t)bitrev32((u32)j, p.log_n) : j;
      }
      u64 v = in[gi];
      if constexpr (PRE) {
        const u64* pre = p.pre + (size_t)bz * (((size_t)1 << (p.log_n - p.pre_bits)) + ((size_t)1 << p.pre_bits));
        const u64 s = gl_mul_nc(pre[j >> p.pre_bits], pre[((size_t)1 << (p.log_n - p.pre_bits)) + (j & (((size_t)1 << p.pre_bits) - 1))]);
        v = gl_mul(v, s);
      } else {
        // a first pass may be handed any u64 representative; the contiguous pass is never first (run_ntt: plans of >= 2 passes end in it)
        // and reads a strided pass's canonical products
        if constexpr (STRIDED) v = gl_canon(v);
      }
      x[q] = v;
    }
    }
    dft_regs<4, INV>(x);
    __syncthreads();  // W table complete
    round_twiddles<R_LOG, LO, 4>(x, W, base_low);
#pragma unroll
    for (int q = 0; q < 16; ++q) tile[tile_idx<R_LOG, STRIDED>(base_low | ((u32)q << LO), t)] = x[q];
  }
  __syncthreads();
  // ---------------- middle round ----------------
  if constexpr (E3 > 0) {
    ntt2_round_lds<R_LOG, E3, E2, STRIDED, INV>(tile, W);
    __syncthreads();
  }
  // ---------------- last round: stage bits [0, EL) ----------------
  constexpr int EL = E3 > 0 ? E3 : E2;
  if constexpr (STRIDED) {
    // straight to global, with the inter-pass ("four-step") twiddle w_{2^span}^(l * rev(m'))
    constexpr u32 n_groups = 1u << (NTT2_TILE_LOG - EL);
    // w^(l * rev(m0 | q)) = w^(l * rev(m0)) * (w^(l * 2^(R-EL)))^rev_EL(q): l belongs to the thread (t does not change
    // over the loop), so the 2^EL - 1 step factors are composed once per thread and every group costs ONE table
    // composition (two gathers) instead of 2^EL of them
    const u32 t = threadIdx.x & ((1u << T_LOG) - 1);
    const u64 l = (base & (((size_t)1 << p.b_lo) - 1)) + t;
    auto root_at = [&](u64 ex) {
      u32 e = (u32)((ex & (((u64)1 << span_log) - 1)) << (ROOT_TABLE_LOG - span_log));
      if (INV) e = ((1u << ROOT_TABLE_LOG) - e) & ((1u << ROOT_TABLE_LOG) - 1);
      return root_pow24_nc(p.root_lo, p.root_hi, e);  // only ever a factor of the products below
    };
    if (p.tw_full) {
      // the twiddle of output (m, l) sits at the output's own offset inside the transform: (m << b_lo) | l
      const u64* __restrict__ twp = p.tw_full + l;
#pragma unroll
      for (u32 g0 = 0; g0 < n_groups; g0 += NTT2_THREADS) {
        const u32 g = g0 + threadIdx.x;
        const u32 m0 = (g >> T_LOG) << EL;
        u64 x[1 << EL], tw[1 << EL];
#pragma unroll
        for (int q = 0; q < (1 << EL); ++q) tw[q] = twp[(size_t)(m0 | (u32)q) << p.b_lo];
#pragma unroll
        for (int q = 0; q < (1 << EL); ++q) x[q] = tile[tile_idx<R_LOG, true>(m0 | (u32)q, t)];
        dft_regs<EL, INV>(x);
#pragma unroll
        for (int q = 0; q < (1 << EL); ++q) out[base + ((size_t)(m0 | (u32)q) << p.b_lo) + t] = gl_mul(x[q], tw[q]);
      }
      return;
    }
    u64 step[1 << EL];
#pragma unroll
    for (int k = 1; k < (1 << EL); ++k) step[k] = root_at(l * ((u64)k << (R_LOG - EL)));
#pragma unroll
    for (u32 g0 = 0; g0 < n_groups; g0 += NTT2_THREADS) {
      const u32 g = g0 + threadIdx.x;
      const u32 base_high = g >> T_LOG;
      const u32 m0 = base_high << EL;
      u64 x[1 << EL];
#pragma unroll
      for (int q = 0; q < (1 << EL); ++q) x[q] = tile[tile_idx<R_LOG, true>(m0 | (u32)q, t)];
      dft_regs<EL, INV>(x);
      u64 tw0 = root_at(l * (u64)bitrev32(m0, R_LOG));
      if (p.post_scale != 1) tw0 = gl_mul_nc(tw0, p.post_scale);
#pragma unroll
      for (int q = 0; q < (1 << EL); ++q) {
        const u32 m = m0 | (u32)q;
        const u32 k = rev_c<EL>((u32)q);
        const u64 tw = k ? gl_mul_nc(tw0, step[k]) : tw0;
        out[base + ((size_t)m << p.b_lo) + t] = gl_mul(x[q], tw);
      }
    }
  } else {
#ifdef VX_NTT2_DIRECT_STORE
    // EXPERIMENT (round 4, VERDICT r3 #7's bounded attempt; MEASURED 2 % SLOWER, not adopted: profiles/r04_ntt_experiment.md) — the last
    // contiguous round stores straight from registers: the 2^EL outputs
    // of a thread's DFT are 2^EL CONSECUTIVE words of the output (LO = 0), so the LDS write-back, the barrier and the copy loop go away;
    // the price is 64-byte-per-lane stores (a wave still covers 4 KB contiguous, in 2^EL / 2 instructions of 16 B per lane at a 64 B stride)
    constexpr u32 n_groups = 1u << (NTT2_TILE_LOG - EL);
#pragma unroll
    for (u32 g0 = 0; g0 < n_groups; g0 += NTT2_THREADS) {
      const u32 g = g0 + threadIdx.x;
      const u32 base_high = g & ((1u << (R_LOG - EL)) - 1);
      const u32 t = g >> (R_LOG - EL);
      const u32 m0 = base_high << EL;
      u64 x[1 << EL];
#pragma unroll
      for (int q = 0; q < (1 << EL); ++q) x[q] = tile[tile_idx<R_LOG, false>(m0 | (u32)q, t)];
      dft_regs<EL, INV>(x);
      u64* dst = out + base + (((size_t)t << R_LOG) | m0);
#pragma unroll
      for (int q = 0; q < (1 << EL); q += 2) {
        u64 v0 = x[q], v1 = x[q + 1];
        if (p.post_scale != 1) v0 = gl_mul(v0, p.post_scale), v1 = gl_mul(v1, p.post_scale);
        *reinterpret_cast<ulonglong2*>(dst + q) = make_ulonglong2(v0, v1);
      }
    }
#else
    ntt2_round_lds<R_LOG, 0, EL, false, INV>(tile, W);
    __syncthreads();
    for (u32 idx = threadIdx.x * 2; idx < TILE; idx += NTT2_THREADS * 2) {  // 16 bytes per lane
      u64 v0 = tile[lds_pad(idx)], v1 = tile[lds_pad(idx + 1)];
      if (p.post_scale != 1) v0 = gl_mul(v0, p.post_scale), v1 = gl_mul(v1, p.post_scale);
      *reinterpret_cast<ulonglong2*>(out + base + idx) = make_ulonglong2(v0, v1);
    }
#endif
  }
}
