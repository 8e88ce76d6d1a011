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
// The TOP of a tree in ONE launch (round 5): every level from `n` children (n <= MTOP_MAX_CHILDREN) down to the cap.  Until round 4
// these were up to 11 launches of merkle_level_coop_kernel, each a single dependent permutation 24 us long whatever its size
// (profiles/r04_prove_kernel_stats.md: 145 launches = 3.5 ms of a 2^21 proof, a third of the kernel time of a 2^14-row one).
//   phase 1: a workgroup takes 2^cw_log consecutive children (<= 128) into LDS and reduces them to ONE node, 16 lanes per node,
//            levels ping-ponging between two LDS buffers (every level is also written to the level-concatenated tree);
//   phase 2: the workgroups of one cap subtree count themselves on a counter; the LAST one to arrive (agent-scope release /
//            acquire around the atomic) gathers the subtree's phase-1 nodes and reduces them to the cap entry.  Nobody waits for
//            anybody: no spinning, so no forward-progress assumption; the last arriver re-zeroes the counter for the next launch.
#define MTOP_THREADS 1024
#define MTOP_GROUPS (MTOP_THREADS / 16)
#define MTOP_MAX_CHILDREN (2 * COOP_MAX_NODES)
#define MTOP_MAX_COUNTERS 256
// `publish`: the LAST level's nodes are what another workgroup will read — they are stored with agent-scope atomic stores (sc1: to
// memory, not into this XCD's L2, whose 128-byte lines are shared with nodes of workgroups on other XCDs).
GLD void merkle_top_reduce(u64 (*buf)[256 * 4], int& cur, unsigned cnt, int nlev, u64* __restrict__ tree, size_t& lvl_off,
                           size_t& lvl_n, size_t& idx0, const u64* rc, bool publish) {
  const int tid = (int)threadIdx.x, grp = tid >> 4, g = tid & 15, group_base = (tid & 63) & ~15;
  for (int lev = 0; lev < nlev; ++lev) {
    const unsigned parents = cnt >> 1;
    lvl_off += lvl_n;
    lvl_n >>= 1;
    idx0 >>= 1;
    for (unsigned base = 0; base < parents; base += MTOP_GROUPS) {
      if (base + (unsigned)(grp & ~3) < parents) {   // wave-uniform: a wavefront none of whose four groups has a node sits the level out
        const unsigned p = base + grp;
        const bool live = p < parents;
        u64 v = (live && g < 8) ? buf[cur][p * 8 + g] : 0;
        v = poseidon_permute_coop_lds_nc(v, g, group_base, rc);
        if (live && g < 4) {
          v = gl_canon(v);
          buf[cur ^ 1][p * 4 + g] = v;
          if (publish && lev == nlev - 1) __hip_atomic_store(&tree[(lvl_off + idx0 + p) * 4 + g], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else tree[(lvl_off + idx0 + p) * 4 + g] = v;
        }
      }
    }
    __syncthreads();
    cur ^= 1;
    cnt = parents;
  }
}
__global__ __launch_bounds__(MTOP_THREADS) void merkle_top_kernel(u64* __restrict__ tree, size_t off, unsigned n, unsigned ncap,
                                                                  int cw_log, unsigned* __restrict__ counters) {
  __shared__ u64 rc[VX_POSEIDON_N_ROUND_CONSTANTS];
  __shared__ u64 buf[2][256 * 4];
  __shared__ unsigned s_last;
  const int tid = (int)threadIdx.x;
  for (int i = tid; i < VX_POSEIDON_N_ROUND_CONSTANTS; i += MTOP_THREADS) rc[i] = POSEIDON_RC[i];
  const unsigned cw = 1u << cw_log;
  const u64* src = tree + (off + (size_t)blockIdx.x * cw) * 4;
  for (unsigned i = tid; i < cw * 4; i += MTOP_THREADS) buf[0][i] = src[i];
  __syncthreads();
  int cur = 0;
  size_t lvl_off = off, lvl_n = n, idx0 = (size_t)blockIdx.x * cw;
  const unsigned nwg = n >> cw_log;
  merkle_top_reduce(buf, cur, cw, cw_log, tree, lvl_off, lvl_n, idx0, rc, nwg > ncap);
  if (nwg <= ncap) return;                 // one workgroup per cap subtree: done
  const unsigned m = nwg / ncap, sub = blockIdx.x / m;   // m workgroups (= m phase-1 nodes) per cap subtree
  // hand-off (cdna_hip_programming.md Guideline 16): the node went out with agent-scope stores; every wave drains its stores, the
  // workgroup meets, ONE lane releases at agent scope and THEN draws its ticket (this order; the asm wait after the fence is the one the
  // compiler may not drop)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned prev = __hip_atomic_fetch_add(&counters[sub], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = prev == m - 1;
    if (prev == m - 1) {
      __hip_atomic_store(&counters[sub], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  }
  __syncthreads();
  if (!s_last) return;
  const u64* nodes = tree + (lvl_off + (size_t)sub * m) * 4;   // agent-scope loads: served past this CU's L1, like the stores that wrote them
  for (unsigned i = tid; i < m * 4; i += MTOP_THREADS) buf[cur][i] = __hip_atomic_load(nodes + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  idx0 = (size_t)sub * m;
  int mlog = 0;
  while ((1u << mlog) < m) ++mlog;
  merkle_top_reduce(buf, cur, m, mlog, tree, lvl_off, lvl_n, idx0, rc, false);
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
