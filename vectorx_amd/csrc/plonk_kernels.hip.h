// PLONK-side kernels of the gfx950 prover: permutation argument (Z + partial products), quotient
// (vanishing-polynomial) evaluation over the LDE, quotient chunking, batch reduction for the opening
// proof, FRI folding, proof-of-work grinding, query gathers.
// Replaces, from plonky2 v0.2.0 (un-vendored, /root/reference/Cargo.lock:4848-4905):
//   plonk/prover.rs   wires_permutation_partial_products_round, compute_quotient_polys
//   plonk/vanishing_poly.rs eval_vanishing_poly_base_batch, evaluate_gate_constraints_base_batch
//   gates/{noop,constant,public_input,arithmetic_base,poseidon}.rs eval_unfiltered_base_batch
//   fri/oracle.rs     prove_openings (reduce_polys_base, divide_by_linear)
//   fri/prover.rs     fri_committed_trees, fri_proof_of_work, fri_prover_query_rounds
// Every kernel is one-thread-per-row over COLUMN-MAJOR data, so the 64 lanes of a wavefront read 64
// consecutive rows of one column (512 B coalesced) for every column they touch.
#pragma once
#include "ntt.hip.h"
#include "poseidon.hip.h"

#define VX_MAX_GATES 64          /* gates per circuit (kernel-argument table: 20 B each); plonky2x registers ~40 gate types in total */
#define VX_MAX_CHALLENGES 2
#define VX_MAX_RATE 16
#define VX_MAX_LUTS 8            /* lookup tables per circuit: sizes LookupParams::lut_poly and the lookup kernel's selector array; desc_check refuses more */

struct GateDev {
  int type, param, selector_index, group_start, group_end;
};

// ------------------------------------------------------------------------------------------------
// Permutation argument, step 1: per-row chunk quotients  prod(num)/prod(den) over chunks of `deg` wires.
// cp[(ch*nchunks + k)*n + i].  (prover.rs: quotient_chunk_products)
// ------------------------------------------------------------------------------------------------
struct PermParams {
  const u64* wires;   // [>=nr][n] natural row order
  const u64* sigmas;  // [nr][n] sigma values on H
  const u64* k_is;    // [nr]
  const u64 *root_lo, *root_hi;
  size_t n;
  int log_n, nr, deg, nchunks, nch;
  u64 betas[VX_MAX_CHALLENGES], gammas[VX_MAX_CHALLENGES];
  u64* cp;
};

#define PERM_MAX_CHUNKS 40  /* ceil(80 routed wires / quotient_degree_factor 2) */
// MAXC bounds nchunks at compile time: every loop over chunks is fully unrolled under a `k < nchunks` guard, so the three per-chunk
// arrays are indexed statically and live in REGISTERS (round 5; the dynamically indexed form kept 976 B per lane in scratch and ran
// at 7.5 cycles per instruction: profiles/r04_kernel_resources.md).  MAXC = 10 is standard_recursion_config (80 routed wires / 8).
template <int MAXC>
__global__ __launch_bounds__(256) void perm_chunk_products_kernel(PermParams p) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.n) return;
  const int ch = blockIdx.y;
  const u64 beta = p.betas[ch], gamma = p.gammas[ch];
  const u64 x = root_pow24(p.root_lo, p.root_hi, (u32)(i << (ROOT_TABLE_LOG - p.log_n)));
  const u64 bx = gl_mul(beta, x);
  u64 np[MAXC], dp[MAXC], pre[MAXC];
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    u64 a = 1, b = 1;
    if (k < p.nchunks) {
      const int j1 = min(p.nr, (k + 1) * p.deg);
      for (int j = k * p.deg; j < j1; ++j) {
        u64 w = gl_canon(p.wires[(size_t)j * p.n + i]);
        u64 wg = gl_add(w, gamma);
        u64 num = gl_mad(p.k_is[j], bx, wg);
        u64 den = gl_mad(beta, p.sigmas[(size_t)j * p.n + i], wg);
        a = gl_mul(a, num);
        b = gl_mul(b, den);
      }
    }
    np[k] = a;
    dp[k] = b;
  }
  // Montgomery batch inversion of dp[0..nchunks) (the chunks beyond nchunks hold 1 and drop out)
  u64 acc = 1;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    pre[k] = acc;
    acc = gl_mul(acc, dp[k]);
  }
  u64 ia = gl_inv(acc);
#pragma unroll
  for (int k = MAXC - 1; k >= 0; --k) {
    if (k < p.nchunks) {
      u64 inv_k = gl_mul(ia, pre[k]);
      ia = gl_mul(ia, dp[k]);
      p.cp[((size_t)ch * p.nchunks + k) * p.n + i] = gl_mul(np[k], inv_k);
    }
  }
}
static void launch_perm_chunk_products(const PermParams& pp, dim3 grid, hipStream_t s) {
  if (pp.nchunks <= 10) hipLaunchKernelGGL(perm_chunk_products_kernel<10>, grid, dim3(256), 0, s, pp);
  else if (pp.nchunks <= 16) hipLaunchKernelGGL(perm_chunk_products_kernel<16>, grid, dim3(256), 0, s, pp);
  else if (pp.nchunks <= 27) hipLaunchKernelGGL(perm_chunk_products_kernel<27>, grid, dim3(256), 0, s, pp);
  else hipLaunchKernelGGL(perm_chunk_products_kernel<PERM_MAX_CHUNKS>, grid, dim3(256), 0, s, pp);
}

// step 2: product of each 256-row block  (the prefix product over rows is the one serial loop of the
// CPU prover; here it is a 3-phase multiplicative scan)
__device__ __forceinline__ u64 block_prod_256(u64 v, u64* sh) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = gl_mul(v, __shfl_down(v, off, 64));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return gl_mul(gl_mul(sh[0], sh[1]), gl_mul(sh[2], sh[3]));
}
__global__ __launch_bounds__(256) void perm_block_products_kernel(const u64* __restrict__ cp, size_t n, int nchunks,
                                                                  u64* __restrict__ block_prod) {
  __shared__ u64 sh[4];
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int ch = blockIdx.y;
  u64 v = 1;
  if (i < n)
    for (int k = 0; k < nchunks; ++k) v = gl_mul(v, cp[((size_t)ch * nchunks + k) * n + i]);
  u64 t = block_prod_256(v, sh);
  if (threadIdx.x == 0) block_prod[(size_t)ch * gridDim.x + blockIdx.x] = t;
}
// step 3: exclusive scan of the block products (one wavefront per challenge; sequential over chunks of 64)
__global__ void perm_scan_blocks_kernel(u64* __restrict__ block_prod, size_t nblocks) {
  const int ch = blockIdx.x;
  u64* bp = block_prod + (size_t)ch * nblocks;
  const int lane = threadIdx.x;  // 64 threads
  u64 carry = 1;
  for (size_t base = 0; base < nblocks; base += 64) {
    u64 v = base + lane < nblocks ? bp[base + lane] : 1;
    u64 incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      u64 o = __shfl_up(incl, off, 64);
      if (lane >= off) incl = gl_mul(incl, o);
    }
    u64 excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 1;
    if (base + lane < nblocks) bp[base + lane] = gl_mul(carry, excl);
    carry = gl_mul(carry, __shfl(incl, 63, 64));
  }
}
// step 4: Z and the partial products.  out columns: [Z_0.., Z_{nch-1}, pp_{0,0..npp-1}, pp_{1,..}]
__global__ __launch_bounds__(256) void perm_write_kernel(const u64* __restrict__ cp, size_t n, int nchunks, int nch,
                                                         const u64* __restrict__ block_carry, u64* __restrict__ out) {
  __shared__ u64 sc[256];
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int ch = blockIdx.y;
  const int npp = nchunks - 1;
  u64 v = 1;
  if (i < n)
    for (int k = 0; k < nchunks; ++k) v = gl_mul(v, cp[((size_t)ch * nchunks + k) * n + i]);
  // inclusive scan of v over the block (Hillis-Steele in LDS)
  sc[threadIdx.x] = v;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    u64 t = threadIdx.x >= off ? sc[threadIdx.x - off] : 1;
    __syncthreads();
    sc[threadIdx.x] = gl_mul(sc[threadIdx.x], t);
    __syncthreads();
  }
  u64 excl = threadIdx.x ? sc[threadIdx.x - 1] : 1;
  if (i >= n) return;
  u64 z = gl_mul(block_carry[(size_t)ch * gridDim.x + blockIdx.x], excl);
  out[(size_t)ch * n + i] = z;
  u64 acc = z;
  for (int k = 0; k < npp; ++k) {
    acc = gl_mul(acc, cp[((size_t)ch * nchunks + k) * n + i]);
    out[((size_t)nch + (size_t)ch * npp + k) * n + i] = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// Quotient evaluation: one thread per LDE row (bit-reversed row order, like every LDE buffer).
// ------------------------------------------------------------------------------------------------
struct QuotientParams {
  const u64 *cs, *wires, *zs;  // LDE buffers: cs has column stride N and is indexed by the GLOBAL row; wires/zs/out
                               // have column stride `stride_w` and hold rows [row_base, row_base + rows) (a coset shard)
  const u64* k_is;
  const u64 *root_lo, *root_hi;
  size_t N, rows, row_base, stride_w;
  int log_n, rate_bits;
  int num_selectors, num_constants, nr, num_wires, nch, npp, deg;
  int num_gates;
  int const_base;    // first gate constant among the preprocessed columns: num_selectors + num_lookup_selectors
  int extra_terms;   // constraints that sit between the partial-product checks and the gate constraints (lookup argument)
  GateDev gates[VX_MAX_GATES];
  u64 betas[VX_MAX_CHALLENGES], gammas[VX_MAX_CHALLENGES], alphas[VX_MAX_CHALLENGES];
  u64 pih[4];
  u64 zh[VX_MAX_RATE], zh_inv[VX_MAX_RATE];  // ZeroPolyOnCoset evals / inverses, indexed by coset r
  u64 n_field;                               // n mod p
  const u64* __restrict__ l0;                // L_0 on the LDE rows (global row index), built once per circuit: l0_table_kernel
  const u64* __restrict__ alpha_pows;        // [VX_MAX_CHALLENGES][VX_ALPHA_POWS]: alpha_c^i
  const Limbs3x2* __restrict__ alpha_limbs;  // the same powers pre-split for carry-free accumulation (poseidon.hip.h dot3)
  u64* out;                                  // [nch][stride_w]
};

// L_0(x) = Z_H(x) / (n (x - 1)) on every row of the LDE domain — it depends on the circuit's size only, and its field inversion
// (~70 multiplications) was 7 % of quotient_kernel<0>'s instructions on every row of every proof: one table per circuit instead
// (8 bytes per LDE row, 0.6 % of the preprocessed commitment it sits next to).
struct L0Params {
  const u64 *root_lo, *root_hi;
  size_t N;
  int log_n, rate_bits;
  u64 zh[VX_MAX_RATE];
  u64 n_field;
  u64* out;
};
__global__ __launch_bounds__(256) void l0_table_kernel(L0Params p) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.N) return;
  const u32 nmask = (1u << p.log_n) - 1;
  const u32 r = bitrev32((u32)(i >> p.log_n), p.rate_bits);
  const u32 k = bitrev32((u32)i & nmask, p.log_n);
  const u32 j = (k << p.rate_bits) | r;
  const u64 x = gl_mul7(root_pow24(p.root_lo, p.root_hi, j << (ROOT_TABLE_LOG - p.log_n - p.rate_bits)));
  p.out[i] = gl_mul(p.zh[r], gl_inv(gl_mul(p.n_field, gl_sub(x, 1))));
}

// reduce_with_powers: acc += term * alpha^idx per challenge.  The powers come from a table built on the host
// (alpha_pows[c * VX_ALPHA_POWS + idx]); idx is wave-uniform, so the loads are scalar and the running-power
// multiply of the naive form disappears.
#define VX_ALPHA_POWS 512
struct AlphaAcc {
  dot3 acc[VX_MAX_CHALLENGES];  // sum_i term_i alpha^i as three limb-class sums (<= 1024 terms: no carries, no reductions)
  int idx;
};
GLD void acc_push(AlphaAcc& a, const QuotientParams& p, u64 term) {
#pragma unroll
  for (int c = 0; c < VX_MAX_CHALLENGES; ++c) dot3_mac(a.acc[c], term, p.alpha_limbs[c * VX_ALPHA_POWS + a.idx]);
  ++a.idx;
}

#define UNUSED_SELECTOR_U64 0xFFFFFFFFULL

#ifndef VX_QUOTIENT_BLOCKS
#define VX_QUOTIENT_BLOCKS 4   /* 128 VGPRs: room for the fixed registers of the single-block multiply (poseidon_sbox_fx) in the PoseidonGate: -0.25 ms against 5 blocks with the generic S-box */
#endif
#ifndef VX_QUOTIENT_PERM_BLOCKS
#define VX_QUOTIENT_PERM_BLOCKS 8
#endif
#ifndef VX_QUOTIENT_SMALL_GATE_BLOCKS
#define VX_QUOTIENT_SMALL_GATE_BLOCKS 8
#endif
// Two launches per quotient (round 3): the vanishing polynomial is a sum, and its two halves want different machines.
//   PART 0  L_0 (Z - 1) and the partial-product checks: 640 multiplications per row on 80 wires + 80 sigmas + 20 Z columns —
//           a long stream of loads with a small live state (two running products), so it runs at 8 blocks per CU and hides
//           its memory latency behind other waves;  writes  out = A / Z_H.
//   PART 1  the SMALL native gates (Constant, PublicInput, Arithmetic): a few loads and multiply-adds per row;  adds  G / Z_H  to out.
//   PART 2  the PoseidonGate alone (round 4: ~19 k instructions on a 12-lane state — register-bound; inside the generic gate loop of
//           rounds 1-3 it carried 50 spilled VGPRs and 204 B of scratch per lane at 128 VGPRs);  adds  G / Z_H  to out.
// Fused (rounds 1-2) the kernel sat at 96 VGPRs with 81 spilled values and the VALU busy 82 % of the time
// (profiles/r03_pmc_sq_prove.md: 4.88 cycles per instruction against 3.98 for the hash kernel).
template <int PART>
__global__ __launch_bounds__(256, PART == 0 ? VX_QUOTIENT_PERM_BLOCKS : PART == 1 ? VX_QUOTIENT_SMALL_GATE_BLOCKS : VX_QUOTIENT_BLOCKS) void quotient_kernel(QuotientParams p) {
  const size_t il = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (il >= p.rows) return;
  const size_t N = p.N, SW = p.stride_w, i = il + p.row_base;
  const int log_n = p.log_n, rb = p.rate_bits;
  const u32 nmask = (1u << log_n) - 1;
  const u32 z = (u32)(i >> log_n);
  const u32 r = bitrev32(z, rb);
#define CS(c) (p.cs[(size_t)(c) * N + i])
#define WIRE(c) (p.wires[(size_t)(c) * SW + il])
#define ZS(c) (p.zs[(size_t)(c) * SW + il])
  const u64 zi = p.zh_inv[r];
  if constexpr (PART == 0) {
  const u32 k = bitrev32((u32)i & nmask, log_n);
  const u32 j = (k << rb) | r;  // natural LDE index
  const size_t il_next = (((size_t)z << log_n) | bitrev32((k + 1) & nmask, log_n)) - p.row_base;  // same coset: local
  const u64 x = gl_mul7(root_pow24(p.root_lo, p.root_hi, j << (ROOT_TABLE_LOG - log_n - rb)));
  AlphaAcc A;
#pragma unroll
  for (int c = 0; c < VX_MAX_CHALLENGES; ++c) A.acc[c] = dot3{0, 0, 0};
  A.idx = 0;

  // (1) L_0(x) (Z(x) - 1) per challenge;  eval_l_0 = Z_H(x) / (n (x - 1)) from the circuit's table
  const u64 l0 = p.l0[i];
  for (int ch = 0; ch < p.nch; ++ch) acc_push(A, p, gl_mul(l0, gl_sub(ZS(ch), 1)));
  // (2) partial-product checks.  Both challenges walk the wires TOGETHER: every wire and sigma value is loaded once and
  // feeds the (beta, gamma) pairs of both — loading them per challenge (rounds 1-2) moved 43 GB through this kernel at
  // n = 2^21, 5 TB/s of its 8.6 ms, which is what held its clock at 1.65 - 1.9 GHz.  The constraints keep their order in
  // the alpha-power table (challenge-major), hence the explicit index.
  {
    const int nchunks = p.npp + 1;
    auto push_at = [&](int idx, u64 term) {
#pragma unroll
      for (int c = 0; c < VX_MAX_CHALLENGES; ++c) dot3_mac(A.acc[c], term, p.alpha_limbs[c * VX_ALPHA_POWS + idx]);
    };
    if (p.nch == 2) {
      const u64 beta0 = p.betas[0], gamma0 = p.gammas[0], beta1 = p.betas[1], gamma1 = p.gammas[1];
      const u64 bx0 = gl_mul(beta0, x), bx1 = gl_mul(beta1, x);
      u64 prev0 = ZS(0), prev1 = ZS(1);
      for (int kk = 0; kk < nchunks; ++kk) {
        u64 np0 = 1, dp0 = 1, np1 = 1, dp1 = 1;
        const int j1 = min(p.nr, (kk + 1) * p.deg);
        for (int jj = kk * p.deg; jj < j1; ++jj) {   // products kept as arbitrary u64 representatives until the end
          const u64 w = WIRE(jj), sg = CS(p.num_constants + jj), kj = p.k_is[jj];
          const u64 wg0 = gl_add(w, gamma0), wg1 = gl_add(w, gamma1);
          np0 = gl_mul_nc(np0, gl_mad_nc(kj, bx0, wg0));
          dp0 = gl_mul_nc(dp0, gl_mad_nc(beta0, sg, wg0));
          np1 = gl_mul_nc(np1, gl_mad_nc(kj, bx1, wg1));
          dp1 = gl_mul_nc(dp1, gl_mad_nc(beta1, sg, wg1));
        }
        const u64 next0 = kk < p.npp ? ZS(2 + kk) : p.zs[il_next];
        const u64 next1 = kk < p.npp ? ZS(2 + p.npp + kk) : p.zs[SW + il_next];
        push_at(2 + kk, gl_sub(gl_mul(prev0, np0), gl_mul(next0, dp0)));
        push_at(2 + nchunks + kk, gl_sub(gl_mul(prev1, np1), gl_mul(next1, dp1)));
        prev0 = next0, prev1 = next1;
      }
    } else {
      for (int ch = 0; ch < p.nch; ++ch) {
        const u64 beta = p.betas[ch], gamma = p.gammas[ch];
        const u64 bx = gl_mul(beta, x);
        u64 prev = ZS(ch);
        for (int kk = 0; kk < nchunks; ++kk) {
          u64 np = 1, dp = 1;
          const int j1 = min(p.nr, (kk + 1) * p.deg);
          for (int jj = kk * p.deg; jj < j1; ++jj) {
            u64 wg = gl_add(WIRE(jj), gamma);
            np = gl_mul_nc(np, gl_mad_nc(p.k_is[jj], bx, wg));
            dp = gl_mul_nc(dp, gl_mad_nc(beta, CS(p.num_constants + jj), wg));
          }
          u64 next = kk < p.npp ? ZS(p.nch + ch * p.npp + kk) : p.zs[(size_t)ch * SW + il_next];
          push_at(p.nch + ch * nchunks + kk, gl_sub(gl_mul(prev, np), gl_mul(next, dp)));
          prev = next;
        }
      }
    }
  }
  for (int ch = 0; ch < p.nch; ++ch) p.out[(size_t)ch * SW + il] = gl_mul(gl_canon(dot3_reduce_nc(A.acc[ch])), zi);
  } else {
  u64 gates_sum[VX_MAX_CHALLENGES] = {0, 0};  // sum_g filter_g * (the gate's alpha-weighted constraints)
  // (3) gate constraints: sum_g filter_g * sum_i c_{g,i} alpha^(i + offset)
  const int base_idx = p.nch * (2 + p.npp) + p.extra_terms;   // after the L_0 terms, the partial-product checks and the lookup terms
  for (int g = 0; g < p.num_gates; ++g) {
    const GateDev gd = p.gates[g];
    if (gd.type == 0 || gd.type >= 5) continue;  // NoopGate / lookup gates: no constraints; program gates: program_gates_kernel
    if ((PART == 2) != (gd.type == 4)) continue;  // the PoseidonGate has its own launch
    const u64 s = CS(gd.selector_index);
    u64 filter = 1;
    for (int q = gd.group_start; q < gd.group_end; ++q)
      if (q != g) filter = gl_mul(filter, gl_sub((u64)q, s));
    if (p.num_selectors > 1) filter = gl_mul(filter, gl_sub(UNUSED_SELECTOR_U64, s));
    AlphaAcc G;
#pragma unroll
    for (int c = 0; c < VX_MAX_CHALLENGES; ++c) G.acc[c] = dot3{0, 0, 0};
    G.idx = base_idx;
    const int c0 = p.const_base;  // gate constants start after the selectors (and the lookup selectors)
    if constexpr (PART == 1) {
    if (gd.type == 1) {              // ConstantGate
      for (int q = 0; q < gd.param; ++q) acc_push(G, p, gl_sub(CS(c0 + q), WIRE(q)));
    } else if (gd.type == 2) {  // PublicInputGate
      for (int q = 0; q < 4; ++q) acc_push(G, p, gl_sub(WIRE(q), p.pih[q]));
    } else if (gd.type == 3) {  // ArithmeticGate
      const u64 k0 = CS(c0), k1 = CS(c0 + 1);
      for (int q = 0; q < gd.param; ++q) {
        u64 m0 = WIRE(4 * q), m1 = WIRE(4 * q + 1), ad = WIRE(4 * q + 2), o = WIRE(4 * q + 3);
        u64 rhs = gl_mad(gl_mul_nc(m0, m1), k0, gl_mul_nc(ad, k1));
        acc_push(G, p, gl_sub(o, rhs));
      }
    }
    } else {  // PART 2: PoseidonGate (gates/poseidon.rs wire layout)
      // The running state is kept as arbitrary u64 representatives ("nc", poseidon.hip.h): wires are canonical, so
      // state - wire and state + constant need one carry fix each, and acc_push's multiply-add takes any representative.
      const u64 swap = WIRE(24);
      acc_push(G, p, gl_mul(swap, gl_sub(swap, 1)));
      u64 st[12];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        u64 lhs = WIRE(q), rhs = WIRE(q + 4), d = WIRE(25 + q);
        acc_push(G, p, gl_sub(gl_mul(swap, gl_sub(rhs, lhs)), d));
        st[q] = gl_add(lhs, d);
        st[q + 4] = gl_sub(rhs, d);
      }
#pragma unroll
      for (int q = 8; q < 12; ++q) st[q] = WIRE(q);
      int round = 0;
#pragma unroll 1
      for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
        for (int q = 0; q < 12; ++q) st[q] = gl_add_nc_c(st[q], POSEIDON_RC[12 * round + q]);
        if (rr != 0) {
#pragma unroll
          for (int q = 0; q < 12; ++q) {
            u64 in = WIRE(29 + 12 * (rr - 1) + q);
            acc_push(G, p, gl_sub_nc_c(st[q], in));
            st[q] = in;
          }
        }
#pragma unroll
        for (int q = 0; q < 12; ++q) st[q] = poseidon_sbox_fx(st[q]);
        poseidon_mds_nc(st);
        ++round;
      }
      // the 22 partial rounds in integer-power blocks (poseidon.hip.h): the S-box input of every round is a short dot
      // product on the block's starting state; the gate constrains it to equal the wire sbox_in and continues from the
      // wire (gates/poseidon.rs), which is what `lane0` does.  The blocks' K constants already contain the NEXT round's
      // round constants, so only the first partial round adds its own.
#pragma unroll
      for (int q = 0; q < 12; ++q) st[q] = gl_add_nc_c(st[q], POSEIDON_RC[12 * round + q]);
      {
        int r0 = 0;
        auto sbox = [](u64 x) { return poseidon_sbox_fx(x); };
        auto lane0 = [&](int j, u64 x) {
          const u64 in = WIRE(65 + r0 + j);
          acc_push(G, p, gl_sub_nc_c(x, in));
          return in;
        };
#pragma unroll 1
        for (int blk = 0; blk < POSEIDON_NBLOCKS - 1; ++blk, r0 += POSEIDON_BLOCK_B)
          poseidon_partial_block_g<POSEIDON_BLOCK_B>(st, POSEIDON_BLK.kappa[blk], POSEIDON_BLK.K[blk], sbox, lane0);
        poseidon_partial_block_g<2>(st, POSEIDON_BLK.kappa[POSEIDON_NBLOCKS - 1], POSEIDON_BLK.K[POSEIDON_NBLOCKS - 1], sbox, lane0);
      }
      round += 22;
#pragma unroll 1
      for (int rr = 0; rr < 4; ++rr) {
        if (rr != 0) {  // round 26's constants came in through the last block's K
#pragma unroll
          for (int q = 0; q < 12; ++q) st[q] = gl_add_nc_c(st[q], POSEIDON_RC[12 * round + q]);
        }
#pragma unroll
        for (int q = 0; q < 12; ++q) {
          u64 in = WIRE(87 + 12 * rr + q);
          acc_push(G, p, gl_sub_nc_c(st[q], in));
          st[q] = in;
        }
#pragma unroll
        for (int q = 0; q < 12; ++q) st[q] = poseidon_sbox_fx(st[q]);
        poseidon_mds_nc(st);
        ++round;
      }
#pragma unroll
      for (int q = 0; q < 12; ++q) acc_push(G, p, gl_sub_nc_c(st[q], WIRE(12 + q)));
    }
#pragma unroll
    for (int c = 0; c < VX_MAX_CHALLENGES; ++c) gates_sum[c] = gl_mad(filter, dot3_reduce_nc(G.acc[c]), gates_sum[c]);
  }
  for (int ch = 0; ch < p.nch; ++ch) {
    u64* o = p.out + (size_t)ch * SW + il;
    *o = gl_add(*o, gl_mul(gates_sum[ch], zi));
  }
  }
#undef CS
#undef WIRE
#undef ZS
}

// ------------------------------------------------------------------------------------------------
// Quotient chunking.  After a per-coset inverse NTT (scaled by 1/n), u[z][pos] holds, for coset
// r = rev(z) and coefficient j = rev_n(pos), U_r[j] * s_r^j with s_r = 7 w_N^r.  The chunk polynomials
// t_c (t(X) = sum_c X^(cn) t_c(X)) follow from a size-2^rb inverse DFT across cosets:
//   t_c[j] = 7^(-nc) / 2^rb * sum_r w_rate^(-rc) * s_r^(-j) * u_r[j]
// in : u   [nch][2^rb][n]   (block z of challenge ch at (ch*2^rb + z)*n)
// out: t   [nch*2^rb][n]    bit-reversed coefficient order (chunk c of challenge ch = column ch*2^rb+c)
// ------------------------------------------------------------------------------------------------
struct ChunkParams {
  const u64* u;
  u64* t;
  const u64* inv_tab;  // [2^rb slices z][hi(2^(log_n-bits)) + lo(2^bits)] tables of s_{rev(z)}^(-j)
  int log_n, rb, bits;
  int nch, zc;  // u is laid out [rate/zc rank blocks][nch][zc][n] (zc = cosets per rank; rate when not sharded)
  u64 w_rate_inv_pows[VX_MAX_RATE];  // w_rate^(-k)
  u64 chunk_scale[VX_MAX_RATE];      // 7^(-nc) / 2^rb
  // trim_to_len(quotient_degree_factor * n): only chunks [0, keep) of a challenge are written, t is [nch][keep][n];
  // a nonzero coefficient in a dropped chunk (the vanishing polynomial is not divisible by Z_H) sets *tail_nonzero
  int keep;
  unsigned* tail_nonzero;
};
__global__ __launch_bounds__(256) void quotient_chunks_kernel(ChunkParams p) {
  const size_t n = (size_t)1 << p.log_n;
  const size_t pos = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (pos >= n) return;
  const int ch = blockIdx.y, rate = 1 << p.rb;
  const u32 j = bitrev32((u32)pos, p.log_n);
  const size_t slice = ((size_t)1 << (p.log_n - p.bits)) + ((size_t)1 << p.bits);
  u64 U[VX_MAX_RATE];
  for (int z = 0; z < rate; ++z) {
    const u64* tab = p.inv_tab + (size_t)z * slice;
    u64 sc = gl_mul(tab[j >> p.bits], tab[((size_t)1 << (p.log_n - p.bits)) + (j & ((1u << p.bits) - 1))]);
    int r = (int)bitrev32((u32)z, p.rb);
    U[r] = gl_mul(p.u[(((size_t)(z / p.zc) * p.nch + ch) * p.zc + (z % p.zc)) * n + pos], sc);
  }
  bool tail = false;
  for (int c = 0; c < rate; ++c) {
    u64 acc = 0;
    for (int r = 0; r < rate; ++r) acc = gl_mad(U[r], p.w_rate_inv_pows[(r * c) & (rate - 1)], acc);
    if (c < p.keep)
      p.t[((size_t)ch * p.keep + c) * n + pos] = gl_mul(acc, p.chunk_scale[c]);
    else
      tail |= acc != 0;
  }
  if (tail) atomicOr(p.tail_nonzero, 1u);
}

// ------------------------------------------------------------------------------------------------
// Opening proof: F[pos] = sum_j alpha^j f_j[pos] over a list of column groups (reduce_polys_base).
// out: 2 columns (a then b), each n.  Column groups are (ptr, ncols) with column stride n.
// ------------------------------------------------------------------------------------------------
#define REDUCE_MAX_GROUPS 8
struct ReduceParams {
  const u64* cols[REDUCE_MAX_GROUPS];
  int ncols[REDUCE_MAX_GROUPS];
  int ngroups;
  const u64* alpha_pows;  // [total][2]
  size_t n;
  u64* out;  // [2][n]
};
__global__ __launch_bounds__(256) void reduce_polys_kernel(ReduceParams p) {
  size_t pos = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (pos >= p.n) return;
  u64 a = 0, b = 0;
  int jj = 0;
  for (int g = 0; g < p.ngroups; ++g) {
    const u64* base = p.cols[g];
    const int nc = p.ncols[g];
    int c = 0;
    // eight loads in flight per lane: a table of ~1000 columns and few rows (the chip STARKs: 2^13 - 2^17 rows) walked one dependent
    // load at a time was one memory latency per column — 1.0 ms for 2^16 x 1013, 0.5 TB/s short of nothing
    for (; c + 8 <= nc; c += 8, jj += 8) {
      u64 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = base[(size_t)(c + k) * p.n + pos];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        a = gl_mad(v[k], p.alpha_pows[2 * (jj + k)], a);
        b = gl_mad(v[k], p.alpha_pows[2 * (jj + k) + 1], b);
      }
    }
    for (; c < nc; ++c, ++jj) {
      u64 v = base[(size_t)c * p.n + pos];
      a = gl_mad(v, p.alpha_pows[2 * jj], a);
      b = gl_mad(v, p.alpha_pows[2 * jj + 1], b);
    }
  }
  p.out[pos] = a;
  p.out[p.n + pos] = b;
}

// final[i] = shift0 * (F0(x_i) - y0)/(x_i - z0) + (F1(x_i) - y1)/(x_i - z1)   in F_p^2, x_i on the LDE coset.
// fl: [4][N] LDE of (F0.a, F0.b, F1.a, F1.b); out interleaved ext [N][2] (bit-reversed row order).
struct CombineParams {
  const u64* fl;  // [4][rows] (column stride = rows): LDE rows [row_base, row_base + rows)
  const u64 *root_lo, *root_hi;
  size_t rows, row_base;
  int log_N;
  u64 y0[2], y1[2], z0[2], z1[2], shift0[2];
  u64* out;
};
__global__ __launch_bounds__(256) void fri_combine_kernel(CombineParams p) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.rows) return;
  u32 j = bitrev32((u32)(i + p.row_base), p.log_N);
  u64 x = gl_mul7(root_pow24(p.root_lo, p.root_hi, j << (ROOT_TABLE_LOG - p.log_N)));
  ext2 d0 = ext_make(gl_sub(x, p.z0[0]), gl_neg(p.z0[1]));
  ext2 d1 = ext_make(gl_sub(x, p.z1[0]), gl_neg(p.z1[1]));
  ext2 inv01 = ext_inv(ext_mul(d0, d1));
  ext2 i0 = ext_mul(inv01, d1), i1 = ext_mul(inv01, d0);
  ext2 n0 = ext_make(gl_sub(p.fl[i], p.y0[0]), gl_sub(p.fl[p.rows + i], p.y0[1]));
  ext2 n1 = ext_make(gl_sub(p.fl[2 * p.rows + i], p.y1[0]), gl_sub(p.fl[3 * p.rows + i], p.y1[1]));
  ext2 q0 = ext_mul(n0, i0), q1 = ext_mul(n1, i1);
  ext2 res = ext_add(ext_mul(q0, ext_make(p.shift0[0], p.shift0[1])), q1);
  reinterpret_cast<ulonglong2*>(p.out)[i] = make_ulonglong2(res.a, res.b);
}

// ------------------------------------------------------------------------------------------------
// FRI folding in the VALUE domain (bit-identical to fri_committed_trees' coefficient fold + coset_fft):
// leaf k holds v[16k + t] = f(y0 * w_16^rev4(t)),  y0 = shift * w_M^rev(k).  Writing
// f(X) = sum_t X^t f_t(X^16):  d_t = y0^t f_t(y0^16) = (1/16) sum_s w_16^(-st) e_s  and the folded value is
// sum_t beta^t f_t(y0^16) = sum_t d_t (beta / y0)^t.
// ------------------------------------------------------------------------------------------------
struct FoldParams {
  const u64* in;  // [M][2]
  u64* out;       // [M >> arity_bits][2]
  const u64 *root_lo, *root_hi;
  size_t M;       // number of input values handled here (a shard of the layer when k_base != 0 or M < layer size)
  size_t k_base;  // global index of the first output
  int log_M, arity_bits;  // log_M = log2 of the WHOLE layer
  u64 beta[2];
  u64 shift_inv;        // (7^(16^round))^-1
  u64 w_inv_pows[16];   // w_arity^(-k)
  u64 arity_inv;        // 1/arity
};
// AB = arity_bits at compile time (round 5): the 2^AB values of a leaf stay in registers (the dynamically sized form kept 272 B per
// lane in scratch) and the size-2^AB inverse DFT is a radix-2 decimation-in-time network — the leaf stores e_s at position rev(s),
// which IS the bit-reversed input order that network wants — instead of a 2^AB x 2^AB matrix product: 17 twiddle multiplies per
// component at arity 16 instead of 256.  Exact field arithmetic: the folded values are the same field elements.
template <int AB>
__global__ __launch_bounds__(256) void fri_fold_kernel(FoldParams p) {
  constexpr int AR = 1 << AB;
  const size_t Mo = p.M >> AB;
  size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= Mo) return;
  const int log_Mo = p.log_M - AB;
  u64 ea[AR], eb[AR];
  const ulonglong2* src = reinterpret_cast<const ulonglong2*>(p.in) + k * AR;
#pragma unroll
  for (int t = 0; t < AR; ++t) {
    ulonglong2 v = src[t];
    ea[t] = gl_canon(v.x);
    eb[t] = gl_canon(v.y);
  }
  // d_t = sum_s w^(-st) e_s for t = 0..AR-1 in natural order, in place
#pragma unroll
  for (int len = 2; len <= AR; len <<= 1) {
#pragma unroll
    for (int i = 0; i < AR; i += len) {
#pragma unroll
      for (int j = 0; j < len / 2; ++j) {
        const int lo = i + j, hi = i + j + len / 2;
        u64 va = ea[hi], vb = eb[hi];
        if (j) {
          const u64 w = p.w_inv_pows[(AR / len) * j];
          va = gl_mul(va, w);
          vb = gl_mul(vb, w);
        }
        const u64 ua = ea[lo], ub = eb[lo];
        ea[lo] = gl_add(ua, va);
        eb[lo] = gl_add(ub, vb);
        ea[hi] = gl_sub(ua, va);
        eb[hi] = gl_sub(ub, vb);
      }
    }
  }
  // y0^-1 = shift^-1 * w_M^(-rev(k))
  u32 kr = bitrev32((u32)(k + p.k_base), log_Mo);
  u32 ex = kr << (ROOT_TABLE_LOG - p.log_M);
  ex = ((1u << ROOT_TABLE_LOG) - ex) & ((1u << ROOT_TABLE_LOG) - 1);
  u64 y0inv = gl_mul(p.shift_inv, root_pow24(p.root_lo, p.root_hi, ex));
  ext2 bq = ext_scale(ext_make(p.beta[0], p.beta[1]), y0inv);  // beta / y0
  // Horner over t from the top: acc = acc * bq + d_t
  ext2 acc = ext_make(0, 0);
#pragma unroll
  for (int t = AR - 1; t >= 0; --t) acc = ext_add(ext_mul(acc, bq), ext_make(ea[t], eb[t]));
  acc = ext_scale(acc, p.arity_inv);
  reinterpret_cast<ulonglong2*>(p.out)[k] = make_ulonglong2(acc.a, acc.b);
}
static hipError_t launch_fri_fold(const FoldParams& fp, size_t leaves, hipStream_t s) {
  const dim3 grid((unsigned)((leaves + 255) / 256)), block(256);
  switch (fp.arity_bits) {
    case 1: hipLaunchKernelGGL(fri_fold_kernel<1>, grid, block, 0, s, fp); break;
    case 2: hipLaunchKernelGGL(fri_fold_kernel<2>, grid, block, 0, s, fp); break;
    case 3: hipLaunchKernelGGL(fri_fold_kernel<3>, grid, block, 0, s, fp); break;
    case 4: hipLaunchKernelGGL(fri_fold_kernel<4>, grid, block, 0, s, fp); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Proof of work (fri/prover.rs::fri_proof_of_work): candidate c goes into sponge slot `pos` of the
// pre-loaded duplex state; accept when the LAST squeezed element (state[7]) has >= pow_bits leading
// zero bits.  atomicMin keeps the SMALLEST valid witness of the batch, so the result is deterministic
// (upstream's rayon find_any is not).
// ------------------------------------------------------------------------------------------------
struct PowParams {
  u64 state[12];
  int pos, pow_bits;
  u64 base;
  unsigned long long* result;  // initialised to ~0
};
__global__ __launch_bounds__(256) void pow_grind_kernel(PowParams p) {
  u64 cand = p.base + (u64)blockIdx.x * 256 + threadIdx.x;
  if (cand >= GL_P) return;
  u64 s[12];
#pragma unroll
  for (int q = 0; q < 12; ++q) s[q] = p.state[q];
#pragma unroll
  for (int q = 0; q < 8; ++q)
    if (q == p.pos) s[q] = cand;
  poseidon_permute(s);
  if ((s[7] >> (64 - p.pow_bits)) == 0) atomicMin(p.result, (unsigned long long)cand);
}

// ------------------------------------------------------------------------------------------------
// Query gathers (fri_prover_query_rounds: tree.get(i) + tree.prove(i)).
// One block per query; out row = [width leaf values][depth x 4 sibling digests].
// colmajor: leaf value c = data[c*stride + idx];  else data[idx*width + c].
// ------------------------------------------------------------------------------------------------
__global__ void gather_open_kernel(const u64* __restrict__ data, size_t stride, int width, int colmajor,
                                   const u64* __restrict__ tree, size_t n_leaves, int depth,
                                   const u64* __restrict__ indices, u64* __restrict__ out) {
  const size_t idx = (size_t)indices[blockIdx.x];
  u64* o = out + (size_t)blockIdx.x * ((size_t)width + 4 * (size_t)depth);
  for (int c = threadIdx.x; c < width; c += blockDim.x)
    o[c] = gl_canon(colmajor ? data[(size_t)c * stride + idx] : data[idx * (size_t)width + c]);
  for (int t = threadIdx.x; t < 4 * depth; t += blockDim.x) {
    int lvl = t >> 2, e = t & 3;
    // level offset = sum_{l<lvl} n_leaves >> l
    size_t off = 0, nl = n_leaves;
    for (int l = 0; l < lvl; ++l) {
      off += nl;
      nl >>= 1;
    }
    size_t node = (idx >> lvl) ^ 1;
    o[width + t] = tree[(off + node) * 4 + e];
  }
}

// ------------------------------------------------------------------------------------------------
// Lookup argument (plonk/vanishing_poly.rs::check_lookup_constraints): one thread per LDE row ADDS
//   zh_inv * sum_t lookup_term_t * alpha^(base_idx + t)
// to the quotient values, for the terms of every challenge in order: LastLdc * SLDC_last, InitSre * SLDC_0, InitSre * RE,
// one "ends" check per table, the RE row transition, then per partial polynomial the Sum and the LDC transition.
// Wire layout: LookupGate slots (2i, 2i+1) = (input, output); LookupTableGate slots (3i, 3i+1, 3i+2) = (input, output,
// multiplicity).  Products of (alpha - combo) over a slot group and its leave-one-out sums are formed with prefix /
// suffix products (groups have at most quotient_degree_factor - 1 = 7 members).
// ------------------------------------------------------------------------------------------------
#define VX_LOOKUP_GROUP_MAX 16
struct LookupParams {
  const u64 *cs, *wires, *zs;  // as in QuotientParams (cs: global rows, stride N; wires / zs / out: local rows, stride_w)
  const u64* alpha_pows;       // [VX_MAX_CHALLENGES][VX_ALPHA_POWS]
  u64* out;
  size_t N, rows, row_base, stride_w;
  int log_n, rate_bits, nch;
  int sel_base;   // column of the first lookup selector = num_selectors
  int zs_base;    // column of challenge 0's RE polynomial inside the zs batch = nch * (1 + npp)
  int nlp;        // lookup polynomials per challenge = 1 + num_sldc
  int lu_slots, lut_slots, lu_deg, lut_deg, num_luts, base_idx;
  u64 deltas[VX_MAX_CHALLENGES][4];
  u64 lut_poly[VX_MAX_CHALLENGES][VX_MAX_LUTS];  // get_lut_poly per challenge and table
  u64 zh_inv[VX_MAX_RATE];
};
// prod_j (alpha - v_j) and sum_i w_i prod_{j != i} (alpha - v_j) over k <= VX_LOOKUP_GROUP_MAX members (w_i = 1 when w == nullptr)
GLD void lookup_group(const u64* f, const u64* w, int k, u64& prod, u64& sum) {
  u64 pre[VX_LOOKUP_GROUP_MAX + 1];
  pre[0] = 1;
  for (int j = 0; j < k; ++j) pre[j + 1] = gl_mul(pre[j], f[j]);
  prod = pre[k];
  u64 suf = 1, acc = 0;
  for (int j = k - 1; j >= 0; --j) {
    const u64 lo = gl_mul(pre[j], suf);
    acc = gl_add(acc, w ? gl_mul(w[j], lo) : lo);
    suf = gl_mul(suf, f[j]);
  }
  sum = acc;
}
// The GENERIC form: any shape desc_check admits, dynamic loops over scratch arrays (~10 cycles per instruction).  The shape of
// standard_recursion_config has its own kernel below (lookup_terms_static_kernel); VX_LOOKUP_TERMS_GENERIC=1 sends that shape here
// too, which is how the two are cross-checked (same proof bytes: tests/test_gpu_prover.py).
__global__ __launch_bounds__(256, 4) void lookup_terms_kernel(LookupParams p) {
  const size_t il = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (il >= p.rows) return;
  const size_t N = p.N, SW = p.stride_w, i = il + p.row_base;
  const int log_n = p.log_n;
  const u32 nmask = (1u << log_n) - 1;
  const u32 z = (u32)(i >> log_n);
  const u32 r = bitrev32(z, p.rate_bits);
  const u32 k = bitrev32((u32)i & nmask, log_n);
  const size_t il_next = (((size_t)z << log_n) | bitrev32((k + 1) & nmask, log_n)) - p.row_base;
#define LW(c) gl_canon(p.wires[(size_t)(c) * SW + il])
  const int nsl = p.nlp - 1;
  const int lut_slots = p.lut_slots, lu_slots = p.lu_slots, lut_deg = p.lut_deg, lu_deg = p.lu_deg;
  u64 sel[4 + VX_MAX_LUTS];
  for (int q = 0; q < 4 + p.num_luts; ++q) sel[q] = p.cs[(size_t)(p.sel_base + q) * N + i];
  u64 total[VX_MAX_CHALLENGES] = {0, 0};
  int idx = p.base_idx;
  auto push = [&](u64 term) {
#pragma unroll
    for (int c = 0; c < VX_MAX_CHALLENGES; ++c) total[c] = gl_mad(term, p.alpha_pows[c * VX_ALPHA_POWS + idx], total[c]);
    ++idx;
  };
  for (int ch = 0; ch < p.nch; ++ch) {
    const u64 da = p.deltas[ch][0], db = p.deltas[ch][1], dalpha = p.deltas[ch][2], ddelta = p.deltas[ch][3];
    const u64* zl = p.zs + (size_t)(p.zs_base + ch * p.nlp) * SW;
    auto Z = [&](int q) { return zl[(size_t)q * SW + il]; };
    auto ZN = [&](int q) { return zl[(size_t)q * SW + il_next]; };
    const u64 z_re = Z(0);
    push(gl_mul(sel[3], Z(nsl)));      // LastLdc * SLDC_{last}
    push(gl_mul(sel[2], Z(1)));        // InitSre * SLDC_0
    push(gl_mul(sel[2], z_re));        // InitSre * RE
    for (int t = 0; t < p.num_luts; ++t) push(gl_mul(sel[4 + t], gl_sub(z_re, p.lut_poly[ch][t])));
    {
      u64 cur = ZN(0);                 // RE row transition: Horner over the looked combos with challenge b
      for (int s2 = 0; s2 < lut_slots; ++s2) cur = gl_add(gl_mul(cur, ddelta), gl_mad(db, LW(3 * s2 + 1), LW(3 * s2)));
      push(gl_mul(sel[0], gl_sub(z_re, cur)));
    }
#pragma unroll 1
    for (int poly = 0; poly < nsl; ++poly) {   // one group of slots at a time: its <= 5 + 7 combos are loaded together, the next group's are not
      const int t0 = poly * lut_deg, t1 = min((poly + 1) * lut_deg, lut_slots);
      const int u0 = poly * lu_deg, u1 = min((poly + 1) * lu_deg, lu_slots);
      (void)t0, (void)t1, (void)u0, (void)u1;
      u64 lut_prod, lut_sum, lu_prod, lu_sum;
      u64 f[VX_LOOKUP_GROUP_MAX], w[VX_LOOKUP_GROUP_MAX];
      for (int s2 = t0; s2 < t1; ++s2) {
        f[s2 - t0] = gl_sub(dalpha, gl_mad(da, LW(3 * s2 + 1), LW(3 * s2)));
        w[s2 - t0] = LW(3 * s2 + 2);
      }
      lookup_group(f, w, max(t1 - t0, 0), lut_prod, lut_sum);   // a polynomial past the last table slot has an empty group: product 1, sum 0
      for (int s2 = u0; s2 < u1; ++s2) f[s2 - u0] = gl_sub(dalpha, gl_mad(da, LW(2 * s2 + 1), LW(2 * s2)));
      lookup_group(f, nullptr, max(u1 - u0, 0), lu_prod, lu_sum);
      const u64 prev = poly == 0 ? ZN(nsl) : Z(poly);
      const u64 d = gl_sub(Z(poly + 1), prev);
      push(gl_mul(sel[0], gl_sub(gl_mul(lut_prod, d), lut_sum)));   // Sum transition
      push(gl_mul(sel[1], gl_add(gl_mul(lu_prod, d), lu_sum)));     // LDC transition
    }
  }
#undef LW
  const u64 zi = p.zh_inv[r];
  for (int ch = 0; ch < p.nch; ++ch) {
    u64* o = p.out + (size_t)ch * SW + il;
    *o = gl_add(*o, gl_mul(total[ch], zi));
  }
}

// ------------------------------------------------------------------------------------------------
// The same terms for the shape of standard_recursion_config (80 routed wires, quotient_degree_factor 8: 26 table slots in groups of
// 5, 40 looking slots in groups of 7, 6 partial sums), re-formulated for instruction count (round 6: 18.8 k -> ~11 k VALU instructions
// per row; the kernel is issue-bound like every Goldilocks kernel here):
//   * a group's wires are loaded ONCE and serve both challenges (the walk is group-major, the terms' alpha powers are addressed
//     explicitly: term t of challenge c sits at base_idx + c T + t);
//   * prod (alpha - v_j) and sum_i w_i prod_{j != i} (alpha - v_j) come from a PRODUCT TREE over (P, S) pairs — merge
//     (P_a P_b, S_a P_b + P_a S_b), leaves (f_i, w_i) — 11 multiplications for seven unweighted members and 12 for five weighted ones
//     instead of 28 + 20 with prefix / suffix products;
//   * the RE row transition  RE(next) delta^26 + sum_s (in_s + b out_s) delta^(25 - s)  is a dot product with constants the host
//     knows (delta^k, b delta^k as pre-split limbs in the parameter block): carry-free accumulation, 6 multiply-adds per wire, one
//     reduction, instead of a 26-step Horner chain of full multiply-reduces;
//   * terms that share a selector are accumulated against their alpha powers FIRST and multiplied by the selector once;
//   * every alpha-power multiply is the carry-free dot3 form against the pre-split powers (as in the gate kernels).
// ------------------------------------------------------------------------------------------------
struct LookupStaticExtra {
  const Limbs3x2* alpha_limbs;                  // [VX_MAX_CHALLENGES][VX_ALPHA_POWS]
  Limbs3x2 re_in[VX_MAX_CHALLENGES][26];        // delta^(25 - s)
  Limbs3x2 re_out[VX_MAX_CHALLENGES][26];       // b delta^(25 - s)
  u64 delta26[VX_MAX_CHALLENGES];
};
struct PSnode {
  u64 P, S;
};
GLD PSnode ps_merge(PSnode a, PSnode b) { return PSnode{gl_mul_nc(a.P, b.P), gl_mad_nc(a.S, b.P, gl_mul_nc(a.P, b.S))}; }
template <int WPS>
__global__ __launch_bounds__(256, WPS) void lookup_terms_static_kernel(LookupParams p, LookupStaticExtra x) {
  const size_t il = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (il >= p.rows) return;
  const size_t N = p.N, SW = p.stride_w, i = il + p.row_base;
  const int log_n = p.log_n;
  const u32 nmask = (1u << log_n) - 1;
  const u32 z = (u32)(i >> log_n);
  const u32 r = bitrev32(z, p.rate_bits);
  const u32 k = bitrev32((u32)i & nmask, log_n);
  const size_t il_next = (((size_t)z << log_n) | bitrev32((k + 1) & nmask, log_n)) - p.row_base;
  constexpr int NSL = 6;
  const int T = 4 + p.num_luts + 2 * NSL;          // terms per challenge
  const u64* __restrict__ wires = p.wires;
  const Limbs3x2* __restrict__ AL = x.alpha_limbs;
  // accumulators: [selector class][alpha challenge]; class 0 = TransSre terms, 1 = TransLdc terms, 2 = everything else (already multiplied)
  dot3 acc[3][VX_MAX_CHALLENGES];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int c = 0; c < VX_MAX_CHALLENGES; ++c) acc[q][c] = dot3{0, 0, 0};
  auto push = [&](int cls, u64 term, int idx) {
#pragma unroll
    for (int c = 0; c < VX_MAX_CHALLENGES; ++c) dot3_mac(acc[cls][c], term, AL[c * VX_ALPHA_POWS + idx]);
  };
  dot3 re_acc[VX_MAX_CHALLENGES];
#pragma unroll
  for (int c = 0; c < VX_MAX_CHALLENGES; ++c) re_acc[c] = dot3{0, 0, 0};
  const u64* zl0 = p.zs + (size_t)p.zs_base * SW;
  const size_t zstride = (size_t)p.nlp * SW;      // challenge stride inside the zs batch
#pragma unroll 1
  for (int poly = 0; poly < NSL; ++poly) {
    u64 in5[5], out5[5], mu5[5], in7[7], out7[7];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const int s2 = poly * 5 + q;
      const int col = s2 < 26 ? 3 * s2 : 0;
      in5[q] = gl_canon(wires[(size_t)col * SW + il]);
      out5[q] = gl_canon(wires[(size_t)(col + 1) * SW + il]);
      mu5[q] = gl_canon(wires[(size_t)(col + 2) * SW + il]);
    }
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const int s2 = poly * 7 + q;
      const int col = s2 < 40 ? 2 * s2 : 0;
      in7[q] = gl_canon(wires[(size_t)col * SW + il]);
      out7[q] = gl_canon(wires[(size_t)(col + 1) * SW + il]);
    }
#pragma unroll
    for (int ch = 0; ch < VX_MAX_CHALLENGES; ++ch) {
      if (ch >= p.nch) break;
      const u64 da = p.deltas[ch][0], dalpha = p.deltas[ch][2];
      const u64* zl = zl0 + (size_t)ch * zstride;
      PSnode L5[5], L7[7];
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const bool on = poly * 5 + q < 26;
        L5[q].P = on ? gl_sub(dalpha, gl_canon(gl_mad_nc(da, out5[q], in5[q]))) : 1;
        L5[q].S = on ? mu5[q] : 0;
        if (on) {   // the RE row transition's dot product rides on the same loads
          dot3_mac(re_acc[ch], in5[q], x.re_in[ch][poly * 5 + q]);
          dot3_mac(re_acc[ch], out5[q], x.re_out[ch][poly * 5 + q]);
        }
      }
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        const bool on = poly * 7 + q < 40;
        L7[q].P = on ? gl_sub(dalpha, gl_canon(gl_mad_nc(da, out7[q], in7[q]))) : 1;
        L7[q].S = on ? 1 : 0;
      }
      const PSnode t5 = ps_merge(ps_merge(ps_merge(L5[0], L5[1]), ps_merge(L5[2], L5[3])), L5[4]);
      // unweighted leaves: (f1, w1) + (f2, w2) = (f1 f2, w1 f2 + w2 f1) with w in {0, 1} — the looking slots' "off" members only occur
      // in the last group, so the general merge is used there as well (neutral members cost what real ones do)
      const PSnode a = ps_merge(L7[0], L7[1]), b = ps_merge(L7[2], L7[3]), cc = ps_merge(L7[4], L7[5]);
      const PSnode t7 = ps_merge(ps_merge(a, b), ps_merge(cc, L7[6]));
      const u64 prev = poly == 0 ? zl[(size_t)NSL * SW + il_next] : zl[(size_t)poly * SW + il];
      const u64 d = gl_sub(gl_canon(zl[(size_t)(poly + 1) * SW + il]), gl_canon(prev));
      const int idx = p.base_idx + ch * T + 4 + p.num_luts + 2 * poly;
      push(0, gl_sub(gl_canon(gl_mul_nc(t5.P, d)), gl_canon(t5.S)), idx);          // Sum transition  (x TransSre)
      push(1, gl_add(gl_canon(gl_mul_nc(t7.P, d)), gl_canon(t7.S)), idx + 1);      // LDC transition  (x TransLdc)
    }
  }
  u64 sel[4 + VX_MAX_LUTS];
  for (int q = 0; q < 4 + p.num_luts; ++q) sel[q] = p.cs[(size_t)(p.sel_base + q) * N + i];
#pragma unroll
  for (int ch = 0; ch < VX_MAX_CHALLENGES; ++ch) {
    if (ch >= p.nch) break;
    const u64* zl = zl0 + (size_t)ch * zstride;
    const u64 z_re = gl_canon(zl[il]);
    const int idx = p.base_idx + ch * T;
    push(2, gl_mul_nc(sel[3], zl[(size_t)NSL * SW + il]), idx);          // LastLdc * SLDC_{last}
    push(2, gl_mul_nc(sel[2], zl[(size_t)1 * SW + il]), idx + 1);        // InitSre * SLDC_0
    push(2, gl_mul_nc(sel[2], z_re), idx + 2);                           // InitSre * RE
    for (int t = 0; t < p.num_luts; ++t) push(2, gl_mul_nc(sel[4 + t], gl_sub(z_re, p.lut_poly[ch][t])), idx + 3 + t);
    const u64 cur = gl_canon(gl_mad_nc(zl[il_next], x.delta26[ch], dot3_reduce_nc(re_acc[ch])));
    push(0, gl_sub(z_re, cur), idx + 3 + p.num_luts);                    // RE row transition (x TransSre)
  }
  const u64 zi = p.zh_inv[r];
#pragma unroll
  for (int c = 0; c < VX_MAX_CHALLENGES; ++c) {
    if (c >= p.nch) break;
    u64 tot = gl_mad_nc(sel[0], dot3_reduce_nc(acc[0][c]), dot3_reduce_nc(acc[2][c]));
    tot = gl_mad(sel[1], dot3_reduce_nc(acc[1][c]), tot);
    u64* o = p.out + (size_t)c * SW + il;
    *o = gl_add(*o, gl_mul(tot, zi));
  }
}

// ------------------------------------------------------------------------------------------------
// The lookup polynomials themselves (prover.rs::compute_lookup_polys), on the device since round 6.  For every challenge the
// columns [RE, SLDC_0 .. SLDC_{k-1}] are zero outside the rows [last_lu_row, first_lut_row] of the table and, inside them,
// recurrences that walk the rows DOWNWARDS:
//   table rows    RE(row) = RE(row + 1) delta^slots + Horner_delta(input_s + b output_s),
//                 SLDC_j(row) = S(row + 1) + sum_{groups <= j} sum_{s in group} multiplicity_s / (alpha - input_s - a output_s)
//   looking rows  SLDC_j(row) = S(row + 1) - sum_{groups <= j} sum_{s in group} 1 / (alpha - input_s - a output_s)
// with S(row) = SLDC_{k-1}(row), S(first_lut_row + 1) = 0.  The per-row parts (one batch inversion of the row's denominators,
// the group sums, the row's Horner value) are independent: one thread per (row, challenge); the two recurrences across rows are a
// suffix sum of the rows' totals (one workgroup: chunk sums, a scan of 1024 partials in LDS, the walk back) and an affine chain
// over the table rows (one lane: a table has a handful of rows).  Rounds 3-5 gathered the rows to the host and ran all of this
// there, with two stream synchronisations in the middle of every proof (2.6 ms at 2^19 rows).  A zero denominator
// (probability 2^-64 per slot) contributes zero, as the host version's batch inversion did.
// ------------------------------------------------------------------------------------------------
struct LookupPolyParams {
  const u64* wires;    // [nr][n] witness, natural rows
  u64* dst;            // [nch * nlp][n]: the lookup columns of the zs batch (zeroed by the caller)
  u64* gc;             // [nch][nsl][R] scratch: prefix sums of the group terms inside a row
  u64* hrow;           // [nch][R] scratch: the row's Horner value (table rows)
  size_t n;
  int last_lu, last_lut, first_lut, R;
  int nch, nsl, lu_slots, lut_slots, lu_deg, lut_deg;
  u64 deltas[VX_MAX_CHALLENGES][4];
  u64 delta_pow_slots[VX_MAX_CHALLENGES];   // delta^lut_slots
};
#define VX_LOOKUP_SLOTS_MAX 64
__global__ __launch_bounds__(64) void lookup_poly_rows_kernel(LookupPolyParams p) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= p.R * p.nch) return;
  const int ch = t / p.R, rr = t % p.R;
  const size_t row = (size_t)p.last_lu + rr;
  const bool table = (int)row >= p.last_lut;
  const u64 da = p.deltas[ch][0], db = p.deltas[ch][1], dalpha = p.deltas[ch][2], ddelta = p.deltas[ch][3];
  const int slots = table ? p.lut_slots : p.lu_slots, step = table ? 3 : 2, deg = table ? p.lut_deg : p.lu_deg;
  auto Wv = [&](int col) { return gl_canon(p.wires[(size_t)col * p.n + row]); };
  u64 den[VX_LOOKUP_SLOTS_MAX], pre[VX_LOOKUP_SLOTS_MAX];
  u64 h = 0, acc = 1;
  for (int s2 = 0; s2 < slots; ++s2) {
    const u64 inp = Wv(step * s2), outp = Wv(step * s2 + 1);
    den[s2] = gl_sub(dalpha, gl_mad(da, outp, inp));
    if (table) h = gl_add(gl_mul(h, ddelta), gl_mad(db, outp, inp));
    pre[s2] = acc;
    if (den[s2]) acc = gl_mul(acc, den[s2]);
  }
  acc = gl_inv(acc);
  for (int s2 = slots - 1; s2 >= 0; --s2) {   // Montgomery's trick backwards: den[s2] <- 1 / den[s2]
    if (!den[s2]) continue;
    const u64 inv = gl_mul(acc, pre[s2]);
    acc = gl_mul(acc, den[s2]);
    den[s2] = inv;
  }
  u64 run = 0;
  for (int slot = 0; slot < p.nsl; ++slot) {
    u64 g = 0;
    for (int s2 = slot * deg; s2 < min((slot + 1) * deg, slots); ++s2) g = gl_add(g, table ? gl_mul(Wv(3 * s2 + 2), den[s2]) : den[s2]);
    run = table ? gl_add(run, g) : gl_sub(run, g);
    p.gc[((size_t)ch * p.nsl + slot) * p.R + rr] = run;
  }
  p.hrow[(size_t)ch * p.R + rr] = h;
}
__global__ __launch_bounds__(1024) void lookup_poly_scan_kernel(LookupPolyParams p) {
  __shared__ u64 part[1024];
  const int tid = threadIdx.x, ch = blockIdx.x;
  const int nlp = p.nsl + 1, R = p.R;
  const int chunk = (R + 1023) / 1024;
  // rows are walked from the TOP (first_lut_row) down: position q = R - 1 - rr, thread tid owns q in [tid chunk, (tid + 1) chunk)
  const u64* total = p.gc + ((size_t)ch * p.nsl + (p.nsl - 1)) * R;
  u64 sum = 0;
  for (int q = tid * chunk; q < min((tid + 1) * chunk, R); ++q) sum = gl_add(sum, total[R - 1 - q]);
  part[tid] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {   // inclusive scan of the chunk sums
    const u64 v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] = gl_add(part[tid], v);
    __syncthreads();
  }
  u64 carry = tid ? part[tid - 1] : 0;         // S(row + 1) of this thread's first row
  for (int q = tid * chunk; q < min((tid + 1) * chunk, R); ++q) {
    const int rr = R - 1 - q;
    const size_t row = (size_t)p.last_lu + rr;
    for (int slot = 0; slot < p.nsl; ++slot)
      p.dst[((size_t)ch * nlp + slot + 1) * p.n + row] = gl_add(carry, p.gc[((size_t)ch * p.nsl + slot) * R + rr]);
    carry = gl_add(carry, total[rr]);
  }
  if (tid == 0) {                              // RE down the table rows
    u64 re = 0;
    for (int row = p.first_lut; row >= p.last_lut; --row) {
      re = gl_add(gl_mul(re, p.delta_pow_slots[ch]), p.hrow[(size_t)ch * R + (row - p.last_lu)]);
      p.dst[((size_t)ch * nlp) * p.n + row] = re;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Constraint programs (include/vxprover.h VX_OP_*): gates outside the native set arrive as straight-line
// programs over the row's wires / constants.  One thread per LDE row interprets the program — every lane
// executes the same instruction, so fetch and decode are wave-uniform — and ADDS
//   zh_inv * sum_g filter_g * sum_i c_{g,i} alpha^(i + offset)
// to the quotient values the native kernel already wrote.  The virtual register file is per-thread private
// memory; this path is for the long tail of cold gates, the hot gates stay compiled.
// ------------------------------------------------------------------------------------------------
#define VX_MAX_PROGRAM_GATES 64
struct ProgramGateDev {
  int gate_index, selector_index, group_start, group_end, prog_off;
};
struct ProgramParams {
  const u64 *cs, *wires;  // as in QuotientParams: cs global rows / stride N; wires, out local rows / stride_w
  const u64* programs;
  size_t N, rows, row_base, stride_w;
  int log_n, rate_bits, num_selectors, nch, num_gates;
  int const_base;
  ProgramGateDev gates[VX_MAX_PROGRAM_GATES];
  u64 alphas[VX_MAX_CHALLENGES], base_pw[VX_MAX_CHALLENGES];  // base_pw = alpha^(number of terms before the gate constraints)
  u64 pih[4];
  u64 zh_inv[VX_MAX_RATE];
  u64* out;  // [nch][N], accumulated into
};
__global__ __launch_bounds__(256) void program_gates_kernel(ProgramParams p) {
  const size_t il = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (il >= p.rows) return;
  const size_t N = p.N, SW = p.stride_w, i = il + p.row_base;
  const u32 r = bitrev32((u32)(i >> p.log_n), p.rate_bits);
  u64 total[VX_MAX_CHALLENGES] = {0, 0};
  u64 R[64];
  for (int g = 0; g < p.num_gates; ++g) {
    const ProgramGateDev gd = p.gates[g];
    const u64 s = p.cs[(size_t)gd.selector_index * N + i];
    u64 filter = 1;
    for (int q = gd.group_start; q < gd.group_end; ++q)
      if (q != gd.gate_index) filter = gl_mul(filter, gl_sub((u64)q, s));
    if (p.num_selectors > 1) filter = gl_mul(filter, gl_sub(UNUSED_SELECTOR_U64, s));
    u64 acc[VX_MAX_CHALLENGES] = {0, 0}, pw[VX_MAX_CHALLENGES] = {p.base_pw[0], p.base_pw[1]};
    const u64* __restrict__ prog = p.programs + gd.prog_off;
    for (int pc = 0;; ++pc) {
      const u64 ins = prog[pc];
      const int op = (int)(ins & 0xFF), dst = (int)((ins >> 8) & 63), a = (int)((ins >> 16) & 0xFFFF), b = (int)((ins >> 32) & 0xFFFF);
      if (op == 0) break;
      switch (op) {
        case 1: R[dst] = gl_canon(p.wires[(size_t)a * SW + il]); break;
        case 2: R[dst] = p.cs[(size_t)(p.const_base + a) * N + i]; break;
        case 3: R[dst] = gl_canon(prog[++pc]); break;
        case 4: R[dst] = gl_add(R[a & 63], R[b & 63]); break;
        case 5: R[dst] = gl_sub(R[a & 63], R[b & 63]); break;
        case 6: R[dst] = gl_mul(R[a & 63], R[b & 63]); break;
        case 7: {
          const u64 term = R[a & 63];
#pragma unroll
          for (int c = 0; c < VX_MAX_CHALLENGES; ++c) {
            acc[c] = gl_mad(term, pw[c], acc[c]);
            pw[c] = gl_mul(pw[c], p.alphas[c]);
          }
          break;
        }
        case 8: R[dst] = p.pih[a & 3]; break;
        default: break;
      }
    }
#pragma unroll
    for (int c = 0; c < VX_MAX_CHALLENGES; ++c) total[c] = gl_mad(filter, acc[c], total[c]);
  }
  const u64 zi = p.zh_inv[r];
  for (int ch = 0; ch < p.nch; ++ch) {
    u64* o = p.out + (size_t)ch * SW + il;
    *o = gl_add(*o, gl_mul(total[ch], zi));
  }
}
