// Second-round columns of a STARK table ON THE GPU (round 4).  Between vx_stark_begin and vx_stark_finish the caller owes the prover its
// lookup / bus columns — witness data, but of one shape everywhere: FRACTIONS  num(row) / den(row)  whose numerator and denominator are
// small polynomial expressions of the row's trace values and the challenges (a log-derivative helper 1/(g - x) + 1/(g - y), a table term
// mult / (g - t), a bus term flag / (g - tuple)), and RUNNING SUMS over the rows of signed combinations of those fractions.  Computed on
// the host in numpy they cost 10 - 20 x the proof they belong to (1.65 s for the 2^20-row EdDSA table against a 94 ms proof); here the
// expressions arrive as one more VX_OP program and the columns are produced where the trace already lives:
//   aux_fraction_kernel   one thread per row interprets the program (VX_OP_LDW local trace value, VX_OP_LDCH challenge, LDI / ADD / SUB /
//                         MUL; every pair of PUSHes = numerator, denominator of the next fraction) and writes num * den^-1 — the
//                         denominators of a row inverted 8 at a time (Montgomery's trick: one Fermat inversion per batch); on short
//                         traces the fractions are dealt to several threads per row (blockIdx.y) so that the chip has waves to run;
//   aux_rowsum_kernel     the signed combination of a row's fractions that each running sum accumulates;
//   aux_scan_*            the EXCLUSIVE prefix sum over rows (mod p) in three phases: thread-sequential runs of 16, an LDS scan of the
//                         block's 256 run totals, a scan of the block totals, and the offsets added back.
// The closing sums (the value of each running sum on the last row = the sum over rows 0 .. n-2, what a bus announces) come back to the host.
#pragma once
#include "goldilocks.hip.h"

#define VX_AUX_MAX_CHALLENGES 16
#define VX_AUX_SCAN_THREADS 256
#define VX_AUX_SCAN_RUN 16

struct AuxFracParams {
  const u64* trace;        // [ncols][n] natural row order
  const u64* program;
  const int* frac_out;     // output column of fraction k
  u64* out;                // [num_out][n]
  size_t n;
  int ncols, nfrac;
  int parts;               // blockIdx.y = part: every thread interprets the whole program (cheap) but inverts and stores only the fractions
                           // k with k % parts == part — short traces get `parts` times the waves and a `parts` times shorter chain of inversions
  u64 chal[VX_AUX_MAX_CHALLENGES];
};
#define VX_AUX_BATCH 8   /* fractions inverted together (Montgomery's trick): one Fermat inversion + 3 multiplications each instead of ~75 */
__global__ __launch_bounds__(256) void aux_fraction_kernel(AuxFracParams p) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.n) return;
  u64 R[VX_PROGRAM_REGS];
  const u64* __restrict__ prog = p.program;
  u64 num = 0;
  int pushes = 0;
  const int part = (int)blockIdx.y;
  u64 bn[VX_AUX_BATCH], bd[VX_AUX_BATCH], pre[VX_AUX_BATCH];
  int bk[VX_AUX_BATCH], cnt = 0;
  auto flush = [&]() {                      // a zero denominator takes part as 1 and yields the fraction 0 (what gl_inv(0) = 0 gave)
    u64 acc = 1;
    for (int j = 0; j < cnt; ++j) {
      pre[j] = acc;
      acc = gl_mul(acc, bd[j] ? bd[j] : 1);
    }
    u64 inv = gl_inv(acc);
    for (int j = cnt - 1; j >= 0; --j) {
      const u64 d = bd[j] ? bd[j] : 1;
      const u64 dinv = gl_mul(inv, pre[j]);
      inv = gl_mul(inv, d);
      p.out[(size_t)p.frac_out[bk[j]] * p.n + i] = bd[j] ? gl_mul(bn[j], dinv) : 0;
    }
    cnt = 0;
  };
  for (int pc = 0;; ++pc) {
    const u64 ins = prog[pc];
    const int op = (int)(ins & 0xFF), dst = (int)((ins >> 8) & 63), a = (int)((ins >> 16) & 0xFFFF), b = (int)((ins >> 32) & 0xFFFF);
    if (op == VX_OP_END) break;
    switch (op) {
      case VX_OP_LDW: R[dst] = gl_canon(p.trace[(size_t)a * p.n + i]); break;
      case VX_OP_LDCH: R[dst] = p.chal[a]; break;
      case VX_OP_LDI: R[dst] = gl_canon(prog[++pc]); break;
      case VX_OP_ADD: R[dst] = gl_add(R[a & 63], R[b & 63]); break;
      case VX_OP_SUB: R[dst] = gl_sub(R[a & 63], R[b & 63]); break;
      case VX_OP_MUL: R[dst] = gl_mul(R[a & 63], R[b & 63]); break;
      case VX_OP_PUSH: {
        const u64 v = R[a & 63];
        if ((pushes & 1) == 0) {
          num = v;
        } else {
          const int k = pushes >> 1;
          if (k < p.nfrac && k % p.parts == part) {
            bn[cnt] = num, bd[cnt] = v, bk[cnt] = k;
            if (++cnt == VX_AUX_BATCH) flush();
          }
        }
        ++pushes;
        break;
      }
      default: break;
    }
  }
  if (cnt) flush();
}

struct AuxSumParams {
  u64* out;               // [num_out][n]
  const int* frac_out;    // [nfrac]
  const int* sum_out;     // [nsum]
  const signed char* coeff;  // [nsum][nfrac] in {-1, 0, +1}
  size_t n;
  int nfrac, nsum;
};
// out[sum_out[j]][i] = sum_k coeff[j][k] * fraction_k(i)   (the per-row term; the scan turns it into the exclusive running sum)
__global__ __launch_bounds__(256) void aux_rowsum_kernel(AuxSumParams p) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.n) return;
  for (int j = 0; j < p.nsum; ++j) {
    u64 s = 0;
    const signed char* c = p.coeff + (size_t)j * p.nfrac;
    for (int k = 0; k < p.nfrac; ++k) {
      if (c[k] == 0) continue;                       // wave-uniform
      const u64 v = p.out[(size_t)p.frac_out[k] * p.n + i];
      s = c[k] > 0 ? gl_add(s, v) : gl_sub(s, v);
    }
    p.out[(size_t)p.sum_out[j] * p.n + i] = s;
  }
}

// Phase A: every block turns its 4096 values into exclusive prefix sums WITHIN the block and writes the block's total.
__global__ __launch_bounds__(VX_AUX_SCAN_THREADS) void aux_scan_blocks_kernel(u64* __restrict__ col, size_t n, u64* __restrict__ block_totals) {
  __shared__ u64 sh[VX_AUX_SCAN_THREADS];
  const size_t base = ((size_t)blockIdx.x * VX_AUX_SCAN_THREADS + threadIdx.x) * VX_AUX_SCAN_RUN;
  u64 v[VX_AUX_SCAN_RUN];
  u64 run = 0;
#pragma unroll
  for (int q = 0; q < VX_AUX_SCAN_RUN; ++q) {
    v[q] = base + q < n ? col[base + q] : 0;
    run = gl_add(run, v[q]);
  }
  sh[threadIdx.x] = run;
  __syncthreads();
  for (int off = 1; off < VX_AUX_SCAN_THREADS; off <<= 1) {       // inclusive Hillis-Steele scan of the 256 run totals
    const u64 add = threadIdx.x >= (unsigned)off ? sh[threadIdx.x - off] : 0;
    __syncthreads();
    sh[threadIdx.x] = gl_add(sh[threadIdx.x], add);
    __syncthreads();
  }
  u64 acc = threadIdx.x ? sh[threadIdx.x - 1] : 0;                 // exclusive offset of this thread's run inside the block
  if (threadIdx.x == VX_AUX_SCAN_THREADS - 1) block_totals[blockIdx.x] = sh[threadIdx.x];
#pragma unroll
  for (int q = 0; q < VX_AUX_SCAN_RUN; ++q) {
    if (base + q < n) col[base + q] = acc;
    acc = gl_add(acc, v[q]);
  }
}
// Phase B: exclusive scan of the block totals by ONE block (up to 2^24 rows / 4096 = 4096 totals: 16 per thread), the grand total last.
__global__ __launch_bounds__(VX_AUX_SCAN_THREADS) void aux_scan_totals_kernel(u64* __restrict__ totals, size_t nblocks, u64* __restrict__ grand) {
  __shared__ u64 sh[VX_AUX_SCAN_THREADS];
  const size_t per = (nblocks + VX_AUX_SCAN_THREADS - 1) / VX_AUX_SCAN_THREADS;
  const size_t b0 = (size_t)threadIdx.x * per;
  u64 run = 0;
  for (size_t q = 0; q < per; ++q)
    if (b0 + q < nblocks) run = gl_add(run, totals[b0 + q]);
  sh[threadIdx.x] = run;
  __syncthreads();
  for (int off = 1; off < VX_AUX_SCAN_THREADS; off <<= 1) {
    const u64 add = threadIdx.x >= (unsigned)off ? sh[threadIdx.x - off] : 0;
    __syncthreads();
    sh[threadIdx.x] = gl_add(sh[threadIdx.x], add);
    __syncthreads();
  }
  u64 acc = threadIdx.x ? sh[threadIdx.x - 1] : 0;
  if (threadIdx.x == VX_AUX_SCAN_THREADS - 1) *grand = sh[threadIdx.x];
  for (size_t q = 0; q < per; ++q)
    if (b0 + q < nblocks) {
      const u64 t = totals[b0 + q];
      totals[b0 + q] = acc;
      acc = gl_add(acc, t);
    }
}
// Phase C: add every block's offset.
__global__ __launch_bounds__(256) void aux_scan_offsets_kernel(u64* __restrict__ col, size_t n, const u64* __restrict__ offsets) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  col[i] = gl_add(col[i], offsets[i / ((size_t)VX_AUX_SCAN_THREADS * VX_AUX_SCAN_RUN)]);
}
