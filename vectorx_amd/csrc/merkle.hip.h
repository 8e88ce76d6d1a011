// Poseidon Merkle-tree kernels for gfx950.
// Replaces plonky2::hash::merkle_tree::MerkleTree::new (leaf hash_or_noop + two_to_one levels, cap at
// depth cap_height) and hashing::hash_n_to_m_no_pad — plonky2 v0.2.0 plonky2/src/hash/{merkle_tree,
// hashing}.rs, un-vendored (/root/reference/Cargo.lock:4848-4905); semantics per SURVEY.md A.2/A.4.
//
// Leaf hashing reads the LDE in its native COLUMN-MAJOR layout: one thread per LDE row, so the 64
// lanes of a wavefront read 64 consecutive rows of one column (512 B coalesced) — the transpose that
// plonky2 materialises on the CPU (fri/oracle.rs: transpose + reverse_index_bits_in_place) never exists.
#pragma once
#include "poseidon.hip.h"

#define HASH_THREADS 256

// digests[row] = hash_or_noop(row of `ncols` values), column-major source.
// clk (optional, profiling only): every 1024th block's first wave records {shader-clock ticks, 100 MHz ticks} it spent in
// the kernel into clk[2 * slot] — the clock this kernel actually sustains (it follows the power budget), read by
// vx_clock_probe for the integer-ALU roofline.  Two scalar timestamp reads per sampled wave: no measurable cost.
#define HASH_CLK_SLOTS 64
__global__ __launch_bounds__(HASH_THREADS, 4) void hash_leaves_colmajor_kernel(
    const u64* __restrict__ cols, size_t col_stride, size_t nrows, int ncols, u64* __restrict__ digests, u64* __restrict__ clk) {
  size_t row = (size_t)blockIdx.x * HASH_THREADS + threadIdx.x;
  const bool sample = clk != nullptr && (blockIdx.x & 1023) == 512 && threadIdx.x < 64;   // wave-uniform
  uint64_t t0 = 0, r0 = 0;
  if (sample) asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));
  if (row >= nrows) return;
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = 0;
  if (ncols <= 4) {
    for (int c = 0; c < ncols; ++c) s[c] = gl_canon(cols[(size_t)c * col_stride + row]);
  } else {
    // one call site of the (inlined, ~28 KB) permutation; the tail chunk is handled by the uniform bound check
#pragma unroll 1
    for (int c = 0; c < ncols; c += 8) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (c + i < ncols) s[i] = cols[(size_t)(c + i) * col_stride + row];   // any u64 representative: the permutation works on those
      poseidon_sponge_step_nc(s, c + 16 <= ncols, c + 8 >= ncols);
    }
  }
  u64* d = digests + row * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) d[i] = gl_canon(s[i]);
  if (sample) {
    uint64_t t1, r1;
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
    if (threadIdx.x == 0) {
      const unsigned slot = (blockIdx.x >> 10) % HASH_CLK_SLOTS;
      clk[2 * slot] = t1 - t0;
      clk[2 * slot + 1] = r1 - r0;
    }
  }
}

// The same sponge over the column range [c0, c1) with its 12-word state carried between launches (state[l * nrows + row], any u64
// representatives): lets the leaf hashing of a batch whose columns arrive over PCIe start before the last column is there
// (batch_commit_host).  c1 - c0 is a multiple of 8 except in the last launch, which writes the digests instead of the state.
// Only for rows of more than 4 columns (hash_or_noop's copy case never gets here).
__global__ __launch_bounds__(HASH_THREADS, 4) void hash_leaves_colmajor_part_kernel(
    const u64* __restrict__ cols, size_t col_stride, size_t nrows, int c0, int c1, u64* __restrict__ state, int first, int last,
    u64* __restrict__ digests) {
  size_t row = (size_t)blockIdx.x * HASH_THREADS + threadIdx.x;
  if (row >= nrows) return;
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = first ? 0 : state[(size_t)i * nrows + row];
#pragma unroll 1
  for (int c = c0; c < c1; c += 8) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (c + i < c1) s[i] = cols[(size_t)(c + i) * col_stride + row];
    // inside a launch the next chunk is known; across a launch boundary it is not (the full layer), unless this is the last launch
    poseidon_sponge_step_nc(s, c + 16 <= c1, last && c + 8 >= c1);
  }
  if (last) {
    u64* d = digests + row * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = gl_canon(s[i]);
  } else {
#pragma unroll
    for (int i = 0; i < 12; ++i) state[(size_t)i * nrows + row] = s[i];
  }
}

// Row-major leaves [nrows][width] (C-ABI vx_merkle_cap and the FRI commit-phase trees, whose leaves
// are 16 consecutive F_p^2 values = 32 contiguous u64).
__global__ __launch_bounds__(HASH_THREADS, 4) void hash_leaves_rowmajor_kernel(
    const u64* __restrict__ leaves, size_t nrows, int width, u64* __restrict__ digests) {
  size_t row = (size_t)blockIdx.x * HASH_THREADS + threadIdx.x;
  if (row >= nrows) return;
  const u64* src = leaves + row * (size_t)width;
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = 0;
  if (width <= 4) {
    for (int c = 0; c < width; ++c) s[c] = gl_canon(src[c]);
  } else {
#pragma unroll 1
    for (int c = 0; c < width; c += 8) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (c + i < width) s[i] = src[c + i];
      poseidon_sponge_step_nc(s, c + 16 <= width, c + 8 >= width);
    }
  }
  u64* d = digests + row * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) d[i] = gl_canon(s[i]);
}

// parents[i] = two_to_one(children[2i], children[2i+1])
__global__ __launch_bounds__(HASH_THREADS, 4) void merkle_level_kernel(const u64* __restrict__ children,
                                                                    u64* __restrict__ parents, size_t n_parents) {
  size_t i = (size_t)blockIdx.x * HASH_THREADS + threadIdx.x;
  if (i >= n_parents) return;
  const ulonglong2* c = reinterpret_cast<const ulonglong2*>(children + i * 8);
  ulonglong2 v0 = c[0], v1 = c[1], v2 = c[2], v3 = c[3];
  u64 s[12] = {v0.x, v0.y, v1.x, v1.y, v2.x, v2.y, v3.x, v3.y, 0, 0, 0, 0};
  poseidon_two_to_one_permute_nc(s);   // capacity lanes zero in, four lanes out: 488 instructions fewer than the generic permutation
  ulonglong2* o = reinterpret_cast<ulonglong2*>(parents + i * 4);
  o[0] = make_ulonglong2(gl_canon(s[0]), gl_canon(s[1]));
  o[1] = make_ulonglong2(gl_canon(s[2]), gl_canon(s[3]));
}

// Lane-cooperative forms (poseidon.hip.h): 16 lanes per node / leaf, for launches too small to fill the chip.
#define COOP_MAX_NODES 16384   /* below this a level is latency-bound in the one-thread-per-node form */
#define COOP_MAX_LEAVES 8192
__global__ __launch_bounds__(HASH_THREADS) void merkle_level_coop_kernel(const u64* __restrict__ children,
                                                                         u64* __restrict__ parents, size_t n_parents) {
  const size_t t = (size_t)blockIdx.x * HASH_THREADS + threadIdx.x;
  const size_t node = t >> 4;
  const int g = (int)(t & 15), group_base = (int)(threadIdx.x & 63) & ~15;
  const bool live = node < n_parents;
  u64 v = (live && g < 8) ? children[node * 8 + g] : 0;
  v = poseidon_permute_coop_nc(v, g, group_base);
  if (live && g < 4) parents[node * 4 + g] = gl_canon(v);
}
__global__ __launch_bounds__(HASH_THREADS) void hash_leaves_rowmajor_coop_kernel(
    const u64* __restrict__ leaves, size_t nrows, int width, u64* __restrict__ digests) {
  const size_t t = (size_t)blockIdx.x * HASH_THREADS + threadIdx.x;
  const size_t row = t >> 4;
  const int g = (int)(t & 15), group_base = (int)(threadIdx.x & 63) & ~15;
  const bool live = row < nrows;
  const u64* src = leaves + (live ? row : 0) * (size_t)width;
  u64 v = 0;
  if (width <= 4) {
    if (g < width) v = src[g];
  } else {
    for (int c = 0; c < width; c += 8) {   // overwrite-mode sponge: rate lanes take the next 8 inputs
      if (g < 8 && c + g < width) v = gl_canon(src[c + g]);
      v = poseidon_permute_coop_nc(v, g, group_base);
    }
  }
  if (live && g < 4) digests[row * 4 + g] = gl_canon(v);
}

// Column-major leaves (the LDE of a PolynomialBatch) with 16 lanes per row, for SMALL traces (round 4): a 2^11-row x 1030-column STARK
// table has 4096 LDE rows = 64 wavefronts in the thread-per-row kernel — one wave on 6 % of the SIMDs walking 129 dependent
// permutations (~4.5 ms whatever the row count); here 4096 rows are 1024 wavefronts and a permutation is ~4 k dependent instructions
// instead of 13 k.  Lane g < 8 of a row's group loads column c + g (a wave reads 4 consecutive rows of 8 columns: 32 B segments —
// irrelevant at these sizes).
#define COOP_COLMAJOR_MAX_ROWS 8192   /* measured (profiles/r04_small_trace_latency.jsonl): 5.0 -> 2.4 - 2.9 ms up to 8192 rows, no gain at 16384, a loss at 32768 */
__global__ __launch_bounds__(HASH_THREADS) void hash_leaves_colmajor_coop_kernel(
    const u64* __restrict__ cols, size_t col_stride, size_t nrows, int ncols, u64* __restrict__ digests) {
  const size_t t = (size_t)blockIdx.x * HASH_THREADS + threadIdx.x;
  const size_t row = t >> 4;
  const int g = (int)(t & 15), group_base = (int)(threadIdx.x & 63) & ~15;
  const bool live = row < nrows;
  const size_t r = live ? row : 0;
  u64 v = 0;
  if (ncols <= 4) {            // hash_or_noop: short leaves are padded, not hashed
    if (g < ncols) v = cols[(size_t)g * col_stride + r];
  } else {
    for (int c = 0; c < ncols; c += 8) {   // overwrite-mode sponge: the rate lanes take the next 8 columns
      if (g < 8 && c + g < ncols) v = cols[(size_t)(c + g) * col_stride + r];
      v = poseidon_permute_coop_nc(v, g, group_base);
    }
  }
  if (live && g < 4) digests[row * 4 + g] = gl_canon(v);
}

__global__ void poseidon_permute_kernel(u64* __restrict__ states, size_t count) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  u64 s[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) s[k] = gl_canon(states[i * 12 + k]);
  poseidon_permute(s);
#pragma unroll
  for (int k = 0; k < 12; ++k) states[i * 12 + k] = s[k];
}
