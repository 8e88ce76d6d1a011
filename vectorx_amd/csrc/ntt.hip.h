// Batched Goldilocks NTT / iNTT / coset-LDE passes for gfx950 (MI355X).
// Replaces plonky2_field::fft::{fft, ifft}, PolynomialCoeffs::{lde, coset_fft}, PolynomialValues::
// {ifft, coset_ifft} (plonky2 v0.2.0 field/src/fft.rs, field/src/polynomial/mod.rs — un-vendored git
// dependency, /root/reference/Cargo.lock:4848-4905; conventions per SURVEY.md A.3).
//
// Design (MI355X-first, HBM-bound):
//  * A length-2^L transform is 1-3 decimation-in-frequency PASSES.  Each pass streams a tile of
//    2^R x T elements through LDS exactly once (one coalesced HBM read + one coalesced HBM write per
//    element per pass): a strided pass takes all 2^R points of T adjacent "low index" values (T*8 B
//    contiguous segments), the last pass takes 2^R contiguous points.
//  * Inside the tile the 2^R-point DIF is done in ROUNDS of up to 4 stages held in VGPRs
//    (radix-16 register butterflies), exchanging through LDS between rounds; the tile's own twiddles
//    (omega_R^k) live in LDS, the inter-pass ("four-step") twiddle omega_N'^(l*k) is rebuilt from two
//    4096-entry root tables that stay L2-resident.
//  * DIF leaves results in bit-reversed positions, which is precisely the row order plonky2's
//    PolynomialBatch wants for Merkle leaves (reverse_index_bits_in_place in fri/oracle.rs), so no
//    transpose / bit-reversal pass is ever materialised.  Coefficients are kept on the device in
//    bit-reversed order too; the first pass of the next transform can read that order directly
//    (the T "columns" of its tile become T contiguous chunks).
//  * The coset LDE (blow-up 8) is run as 8 independent size-n coset NTTs (shift 7*w_8n^r) instead of
//    one zero-padded size-8n NTT: 1/3 fewer passes, and coset r lands in the contiguous block
//    [rev3(r)*n, (rev3(r)+1)*n) of the bit-reversed LDE.
#pragma once
#include "goldilocks.hip.h"

#define NTT_THREADS 512
#define NTT_MAX_TILE_LOG 13  // 8192 elements = 64 KiB of LDS
#define ROOT_TABLE_LOG 24    // root tables cover sizes up to 2^24
#define ROOT_SPLIT 12

struct NttPassParams {
  const u64* in;
  u64* out;
  size_t in_col_stride, out_col_stride;  // elements between columns
  size_t in_z_stride, out_z_stride;      // elements between grid.z slices (cosets)
  int log_n;                             // transform length 2^log_n
  int b_lo;                              // this pass does stages b_lo .. b_lo+R_LOG-1
  int t_log;                             // tile has 2^t_log "columns"
  int in_bitrev;                         // first pass only: input stored in bit-reversed order
  int inverse;                           // use omega^-1
  const u64* root_lo;                    // w_{2^24}^k,        k < 4096
  const u64* root_hi;                    // w_{2^24}^(4096 k), k < 4096
  const u64* pre;                        // optional prescale tables [z][ 2^(log_n-pre_bits) + 2^pre_bits ]
  int pre_bits;
  u64 post_scale;                        // every output *= post_scale (1 = skip)
};

GLD u32 bitrev32(u32 x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

GLD u64 root_pow24(const u64* __restrict__ lo, const u64* __restrict__ hi, u32 e) {
  // w_{2^24}^e
  return gl_mul(lo[e & 4095u], hi[(e >> 12) & 4095u]);
}
// the same root as SOME u64 representative: for values that only feed multiplications
GLD u64 root_pow24_nc(const u64* __restrict__ lo, const u64* __restrict__ hi, u32 e) {
  return gl_mul_nc(lo[e & 4095u], hi[(e >> 12) & 4095u]);
}

// LDS padding: one extra 8-byte word every 32 words keeps power-of-two strides conflict-free.
GLD u32 lds_pad(u32 i) { return i + (i >> 5) + (i >> 9); }  // second level spreads the bit-reversed first-pass load (stride 2^9+)

template <int E>
GLD void reg_butterflies(u64 (&x)[1 << E], const u64* __restrict__ tw, u32 base_low, int lo_bits,
                         int r_log) {
  // DIF stages lo_bits+E-1 ... lo_bits on 2^E register-resident points whose local index is
  // m(q) = (q << lo_bits) | base_low (+ high bits that do not enter the twiddle).
#pragma unroll
  for (int u = E - 1; u >= 0; --u) {
    const int half = 1 << u;
    const int tshift = r_log - (lo_bits + u) - 1;
#pragma unroll
    for (int q = 0; q < (1 << E); ++q) {
      if (q & half) continue;
      u64 a = x[q], b = x[q + half];
      u32 k = ((((u32)(q & (half - 1))) << lo_bits) | base_low) << tshift;
      x[q] = gl_add(a, b);
      x[q + half] = gl_mul(gl_sub(a, b), tw[k]);
    }
  }
}

// One round = stages [lo_bits, lo_bits+E) of the local 2^R_LOG-point DIF for all tile columns.
template <int R_LOG, int E>
GLD void lds_round(u64* __restrict__ tile, const u64* __restrict__ tw, int lo_bits, int t_log,
                   bool strided) {
  const u32 n_groups = 1u << (R_LOG - E + t_log);
  for (u32 g = threadIdx.x; g < n_groups; g += NTT_THREADS) {
    u32 t, base_low, base_high;
    if (strided) {
      t = g & ((1u << t_log) - 1);
      u32 rest = g >> t_log;
      base_low = rest & ((1u << lo_bits) - 1);
      base_high = rest >> lo_bits;
    } else {
      base_low = g & ((1u << lo_bits) - 1);
      u32 rest = g >> lo_bits;
      base_high = rest & ((1u << (R_LOG - E - lo_bits)) - 1);
      t = rest >> (R_LOG - E - lo_bits);
    }
    const u32 m0 = (base_high << (lo_bits + E)) | base_low;
    u64 x[1 << E];
#pragma unroll
    for (int q = 0; q < (1 << E); ++q) {
      u32 m = m0 | ((u32)q << lo_bits);
      u32 idx = strided ? ((m << t_log) | t) : ((t << R_LOG) | m);
      x[q] = tile[lds_pad(idx)];
    }
    reg_butterflies<E>(x, tw, base_low, lo_bits, R_LOG);
#pragma unroll
    for (int q = 0; q < (1 << E); ++q) {
      u32 m = m0 | ((u32)q << lo_bits);
      u32 idx = strided ? ((m << t_log) | t) : ((t << R_LOG) | m);
      tile[lds_pad(idx)] = x[q];
    }
  }
}

template <int R_LOG>
__global__ __launch_bounds__(NTT_THREADS) void ntt_pass_kernel(NttPassParams p) {
  extern __shared__ __attribute__((aligned(16))) u64 smem[];
  const int t_log = p.t_log;
  const u32 tile_elems = 1u << (R_LOG + t_log);
  u64* tile = smem;                              // lds_pad(tile_elems) words
  u64* tw = smem + lds_pad(tile_elems) + 1;      // 2^(R_LOG-1) local twiddles
  const bool strided = p.b_lo > 0;
  const int span_log = p.b_lo + R_LOG;           // this pass works inside blocks of 2^span_log

  // ---- local twiddle table  w_R^k (or its inverse) ----
  for (u32 k = threadIdx.x; k < (1u << (R_LOG > 0 ? R_LOG - 1 : 0)); k += NTT_THREADS) {
    u32 e = k << (ROOT_TABLE_LOG - R_LOG);
    if (p.inverse) e = ((1u << ROOT_TABLE_LOG) - e) & ((1u << ROOT_TABLE_LOG) - 1);
    tw[k] = root_pow24(p.root_lo, p.root_hi, e);
  }

  // ---- tile coordinates ----
  // strided:  tile = { (H, m, l0+t) }  i = H<<span | m<<b_lo | (l0+t)
  // contig :  tile = 2^(R_LOG+t_log) consecutive elements (t = consecutive H)
  const size_t tile_id = blockIdx.x;
  size_t base;  // global element offset of tile element (m=0,t=0)
  if (strided) {
    const size_t tiles_per_span = (size_t)1 << (p.b_lo - t_log);
    const size_t H = tile_id / tiles_per_span;
    const size_t l0 = (tile_id % tiles_per_span) << t_log;
    base = (H << span_log) | l0;
  } else {
    base = tile_id << (R_LOG + t_log);
  }
  const u64* __restrict__ in = p.in + (size_t)blockIdx.y * p.in_col_stride + (size_t)blockIdx.z * p.in_z_stride;
  u64* __restrict__ out = p.out + (size_t)blockIdx.y * p.out_col_stride + (size_t)blockIdx.z * p.out_z_stride;
  const u64* __restrict__ pre = p.pre ? p.pre + (size_t)blockIdx.z * (((size_t)1 << (p.log_n - p.pre_bits)) + ((size_t)1 << p.pre_bits)) : nullptr;

  // ---- load tile (coalesced) ----
  if (!p.in_bitrev) {
    for (u32 idx = threadIdx.x; idx < tile_elems; idx += NTT_THREADS) {
      size_t gi;
      if (strided) {
        u32 m = idx >> t_log, t = idx & ((1u << t_log) - 1);
        gi = base + ((size_t)m << p.b_lo) + t;
      } else {
        gi = base + idx;
      }
      u64 v = gl_canon(in[gi]);
      if (pre) {
        u64 s = gl_mul(pre[gi >> p.pre_bits], pre[((size_t)1 << (p.log_n - p.pre_bits)) + (gi & (((size_t)1 << p.pre_bits) - 1))]);
        v = gl_mul(v, s);
      }
      tile[lds_pad(idx)] = v;
    }
  } else {
    // Input holds x[j] at position rev_L(j).  Natural index j = base + (m<<b_lo) + t  (first pass: H == 0
    // for strided; for a single contiguous pass j = base + idx with base == 0 and t_log == 0).
    for (u32 idx = threadIdx.x; idx < tile_elems; idx += NTT_THREADS) {
      // enumerate in INPUT-contiguous order: c fastest within chunk t
      u32 c = idx & ((1u << R_LOG) - 1), t = idx >> R_LOG;
      u32 m = bitrev32(c, R_LOG);
      size_t j, pos;
      u32 lidx;
      if (strided) {
        size_t l = (base & (((size_t)1 << p.b_lo) - 1)) + t;
        j = ((size_t)m << p.b_lo) | l;
        pos = ((size_t)bitrev32((u32)l, p.b_lo) << R_LOG) | c;
        lidx = (m << t_log) | t;
      } else {
        j = base + (((size_t)t << R_LOG) | m);
        pos = bitrev32((u32)j, p.log_n);
        lidx = (t << R_LOG) | m;
      }
      u64 v = gl_canon(in[pos]);
      if (pre) {
        u64 s = gl_mul(pre[j >> p.pre_bits], pre[((size_t)1 << (p.log_n - p.pre_bits)) + (j & (((size_t)1 << p.pre_bits) - 1))]);
        v = gl_mul(v, s);
      }
      tile[lds_pad(lidx)] = v;
    }
  }
  __syncthreads();

  // ---- local 2^R_LOG-point DIF in rounds of <= 4 register-resident stages ----
  {
    constexpr int E = 4;
    int top = R_LOG;  // stages [0, top) remain
    // full radix-16 rounds
#pragma unroll
    for (int rd = 0; rd < R_LOG / E; ++rd) {
      top -= E;
      lds_round<R_LOG, E>(tile, tw, top, t_log, strided);
      __syncthreads();
    }
    constexpr int REM = R_LOG % E;
    if (REM == 3) { lds_round<R_LOG, (REM == 3 ? 3 : 1)>(tile, tw, 0, t_log, strided); __syncthreads(); }
    if (REM == 2) { lds_round<R_LOG, (REM == 2 ? 2 : 1)>(tile, tw, 0, t_log, strided); __syncthreads(); }
    if (REM == 1) { lds_round<R_LOG, 1>(tile, tw, 0, t_log, strided); __syncthreads(); }
  }

  // ---- inter-pass twiddle, optional scale, store (coalesced, in natural tile positions) ----
  for (u32 idx = threadIdx.x; idx < tile_elems; idx += NTT_THREADS) {
    u64 v = tile[lds_pad(idx)];
    size_t gi;
    if (strided) {
      u32 m = idx >> t_log, t = idx & ((1u << t_log) - 1);
      gi = base + ((size_t)m << p.b_lo) + t;
      // element (m', l) *= w_{2^span}^(l * rev_R(m'))
      u64 l = (base & (((size_t)1 << p.b_lo) - 1)) + t;
      u64 ex = (l * (u64)bitrev32(m, R_LOG)) & (((u64)1 << span_log) - 1);
      u32 e = (u32)(ex << (ROOT_TABLE_LOG - span_log));
      if (p.inverse) e = ((1u << ROOT_TABLE_LOG) - e) & ((1u << ROOT_TABLE_LOG) - 1);
      v = gl_mul(v, root_pow24(p.root_lo, p.root_hi, e));
    } else {
      gi = base + idx;
    }
    if (p.post_scale != 1) v = gl_mul(v, p.post_scale);
    out[gi] = v;
  }
}

// out[i] = in[rev_L(i)]  (column batched).  Used only at the C-ABI surface to hand natural-order
// results to callers of the L1 primitives; the prover itself never needs it.
__global__ void bitrev_permute_kernel(const u64* __restrict__ in, u64* __restrict__ out, int log_n,
                                      size_t in_col_stride, size_t out_col_stride) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >> log_n) return;
  const u64* src = in + (size_t)blockIdx.y * in_col_stride;
  u64* dst = out + (size_t)blockIdx.y * out_col_stride;
  dst[i] = gl_canon(src[bitrev32((u32)i, log_n)]);
}
