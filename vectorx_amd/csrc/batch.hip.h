// Device-resident PolynomialBatch (plonky2::fri::oracle::PolynomialBatch, plonky2 v0.2.0
// plonky2/src/fri/oracle.rs — un-vendored, /root/reference/Cargo.lock:4848-4905).
//
// HBM layout of one batch (n = 2^log_n rows, N = n * 2^rate_bits LDE rows, m columns):
//   coeffs [m][n]  u64, column-major, BIT-REVERSED coefficient order (position rev(j) holds c_j)
//   lde    [m][N]  u64, column-major, row index = MerkleTree leaf index = rev_N(LDE point index);
//                  coset r (points 7*w_N^(r + 2^rate_bits * k)) is the contiguous row block
//                  [rev(r)*n, (rev(r)+1)*n)
//   tree   level 0 = N leaf digests (4 u64 each), then N/2, ... down to the 2^cap_height cap.
//
// Coset shard (vx_prove_sharded, world = 2^shard_lg ranks): coeffs are complete on every rank, but `lde` holds only
// the rows [rank * N/world, (rank+1) * N/world) — whole cosets, 2^rate_bits / world of them — with column stride
// N/world, and `tree` is the Merkle subtree over those leaves down to this rank's 2^(cap_height - shard_lg) cap
// entries (the top shard_lg bits of a leaf index are the rank, so the global cap is the ranks' caps concatenated).
#pragma once
#include "vx_runtime.hip.h"

struct vx_batch {
  vx_ctx* ctx = nullptr;
  int log_n = 0, rate_bits = 0, cap_height = 0;
  size_t ncols = 0;
  u64* coeffs = nullptr;
  u64* lde = nullptr;
  u64* tree = nullptr;
  size_t cap_off = 0;  // digest index of the cap level inside `tree`
  int shard_rank = 0, shard_lg = 0;
  size_t n() const { return (size_t)1 << log_n; }
  size_t rows() const { return (n() << rate_bits) >> shard_lg; }       // local LDE rows (= column stride of `lde`)
  size_t row_base() const { return rows() * (size_t)shard_rank; }       // global index of the first local row
  int local_cap_height() const { return cap_height - shard_lg; }
  size_t local_cap_words() const { return (size_t)4 << local_cap_height(); }
};

__global__ void canon_kernel(u64* __restrict__ x, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = gl_canon(x[i]);
}

static int get_scale_tables(vx_ctx* c, int log_n, int bits, const std::vector<u64>& shifts, u64 pre_mul, u64** out) {
  std::string key = std::to_string(log_n) + ":" + std::to_string(bits) + ":" + std::to_string(pre_mul);
  for (u64 s : shifts) key += ":" + std::to_string(s);
  auto it = c->scale_cache.find(key);
  if (it != c->scale_cache.end()) {
    *out = it->second;
    return VX_OK;
  }
  u64* d = nullptr;
  VXCHK(build_scale_tables(c, log_n, bits, shifts, pre_mul, &d));
  c->scale_cache[key] = d;
  *out = d;
  return VX_OK;
}

// fft / ifft / coset_fft / coset_ifft on natural-order input; result in bit-reversed order.
static int ntt_natural_to_bitrev(vx_ctx* c, const u64* src, u64* dst, int log_n, size_t ncols, int kind, u64 shift) {
  using namespace vxh;
  if (log_n < 1 || log_n > ROOT_TABLE_LOG) return vx_fail(VX_E_INVALID, "ntt: log_n=%d out of range [1,24]", log_n);
  size_t n = (size_t)1 << log_n;
  double bytes = 16.0 * (double)n * (double)ncols;
  u64 ninv = inv((u64)n % P);
  shift = canon(shift);
  switch (kind) {
    case VX_NTT_FFT:
      return run_ntt(c, src, dst, n, n, 0, 0, log_n, ncols, 1, false, false, nullptr, 0, 1, "ntt_fwd", bytes);
    case VX_NTT_IFFT:
      return run_ntt(c, src, dst, n, n, 0, 0, log_n, ncols, 1, true, false, nullptr, 0, ninv, "ntt_inv", bytes);
    case VX_NTT_COSET_FFT: {
      if (shift == 0) return vx_fail(VX_E_INVALID, "coset_fft: shift must be non-zero");
      int bits = log_n / 2;
      u64* tab = nullptr;
      VXCHK(get_scale_tables(c, log_n, bits, {shift}, 1, &tab));
      return run_ntt(c, src, dst, n, n, 0, 0, log_n, ncols, 1, false, false, tab, bits, 1, "ntt_coset_fwd", bytes);
    }
    default: {
      if (shift == 0) return vx_fail(VX_E_INVALID, "coset_ifft: shift must be non-zero");
      int bits = log_n / 2;
      u64* tab = nullptr;
      VXCHK(get_scale_tables(c, log_n, bits, {inv(shift)}, ninv, &tab));
      VXCHK(run_ntt(c, src, dst, n, n, 0, 0, log_n, ncols, 1, true, false, nullptr, 0, 1, "ntt_coset_inv", bytes));
      hipLaunchKernelGGL(scale_bitrev_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)ncols), dim3(256), 0,
                         c->stream, dst, n, log_n, tab, bits);
      HIPCHK(hipGetLastError());
      return VX_OK;
    }
  }
}

static int batch_alloc(vx_ctx* c, int log_n, size_t ncols, int rate_bits, int cap_height, vx_batch** out,
                       int shard_rank = 0, int shard_lg = 0) {
  if (shard_lg > rate_bits || shard_lg > cap_height) return vx_fail(VX_E_INVALID, "batch_alloc: too many shards");
  vx_batch* b = new vx_batch();
  b->ctx = c;
  b->log_n = log_n;
  b->rate_bits = rate_bits;
  b->cap_height = cap_height;
  b->ncols = ncols;
  b->shard_rank = shard_rank;
  b->shard_lg = shard_lg;
  size_t n = (size_t)1 << log_n, N = b->rows();
  size_t nd = merkle_tree_digest_count(N, b->local_cap_height());
  if (c->pool_alloc((void**)&b->coeffs, n * ncols * 8) != hipSuccess ||
      c->pool_alloc((void**)&b->lde, N * ncols * 8) != hipSuccess ||
      c->pool_alloc((void**)&b->tree, nd * 32) != hipSuccess) {
    c->pool_free(b->coeffs);
    c->pool_free(b->lde);
    c->pool_free(b->tree);
    delete b;
    return vx_fail(VX_E_NOMEM, "batch_alloc: out of device memory (n=2^%d, %zu cols, blow-up 2^%d)", log_n, ncols, rate_bits);
  }
  *out = b;
  return VX_OK;
}

// coefficients (bit-reversed, device) -> LDE of the columns [col0, col0 + ncols)
static int batch_lde_cols(vx_ctx* c, vx_batch* b, size_t col0, size_t ncols) {
  using namespace vxh;
  const int log_n = b->log_n, rb = b->rate_bits;
  const size_t n = (size_t)1 << log_n, N = b->rows();
  const int nz = (1 << rb) >> b->shard_lg, z0 = nz * b->shard_rank;  // local cosets [z0, z0 + nz)
  // block z of the bit-reversed LDE holds coset r = rev_rb(z): shift 7 * w_N^r
  std::vector<u64> shifts(nz);
  u64 wN = root_of_unity(log_n + rb);
  for (int z = 0; z < nz; ++z) shifts[z] = mul(7, pow(wN, reverse_bits((size_t)(z0 + z), rb)));
  int bits = log_n / 2;
  u64* tab = nullptr;
  VXCHK(get_scale_tables(c, log_n, bits, shifts, 1, &tab));
  return run_ntt(c, b->coeffs + col0 * n, b->lde + col0 * N, n, N, 0, n, log_n, ncols, nz, false, true, tab, bits, 1, "lde",
                 (double)ncols * 8.0 * ((double)n + (double)N));
}
// LDE -> leaf digests -> Merkle levels
static int batch_hash_tree(vx_ctx* c, vx_batch* b) {
  const size_t N = b->rows(), m = b->ncols;
  {
    ProfScope ps(c, "hash_leaves", (double)m * 8.0 * (double)N);
    static const size_t coop_max = [] {   // VX_COOP_LEAF_MAX_ROWS: A/B knob (0 = never); default from the measurement in profiles/r04_small_trace_latency.md
      const char* e = getenv("VX_COOP_LEAF_MAX_ROWS");
      return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)COOP_COLMAJOR_MAX_ROWS;
    }();
    if (N <= coop_max && m > 8) {   // small trace, long sponge: 16 lanes per row
      hipLaunchKernelGGL(hash_leaves_colmajor_coop_kernel, dim3((unsigned)((N * 16 + HASH_THREADS - 1) / HASH_THREADS)), dim3(HASH_THREADS), 0,
                         c->stream, b->lde, N, N, (int)m, b->tree);
    } else {
      hipLaunchKernelGGL(hash_leaves_colmajor_kernel, dim3((unsigned)((N + HASH_THREADS - 1) / HASH_THREADS)),
                         dim3(HASH_THREADS), 0, c->stream, b->lde, N, N, (int)m, b->tree, c->prof_on ? c->hash_clk : nullptr);
    }
    HIPCHK(hipGetLastError());
  }
  VXCHK(build_merkle_levels(c, b->tree, N, b->local_cap_height(), &b->cap_off));
  return VX_OK;
}
static int batch_lde_and_tree(vx_ctx* c, vx_batch* b) {
  VXCHK(batch_lde_cols(c, b, 0, b->ncols));
  return batch_hash_tree(c, b);
}

static int batch_commit_device(vx_ctx* c, vx_batch* b, const u64* src, size_t n, bool is_coeffs) {
  using namespace vxh;
  const size_t m = b->ncols;
  if (is_coeffs) {
    ProfScope ps(c, "bitrev_permute", 16.0 * (double)n * (double)m);
    hipLaunchKernelGGL(bitrev_permute_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)m), dim3(256), 0, c->stream, src,
                       b->coeffs, b->log_n, n, n);
    HIPCHK(hipGetLastError());
  } else {
    u64 ninv = inv((u64)n % P);
    VXCHK(run_ntt(c, src, b->coeffs, n, n, 0, 0, b->log_n, m, 1, true, false, nullptr, 0, ninv, "intt",
                  16.0 * (double)n * (double)m));
  }
  return batch_lde_and_tree(c, b);
}

// The same from a HOST matrix, with the upload hidden: the matrix crosses PCIe in column blocks on the context's copy stream
// while the interpolation (or bit-reversal) and coset extension of the previous block run on the main stream — columns are
// independent polynomials.  The leaf hashing needs them all, but it is a sponge over the columns in order: for large batches it
// runs in several launches that carry the sponge state (after the first 8 columns, after 56, at the end; wide traces: a dozen
// pieces), so the GPU has hashing to do while the later blocks are still on the bus and only the first 8-column block's transfer
// stays exposed
// (n = 2^21 x 135 columns: 2.27 GB = ~41 ms of PCIe against ~27 ms of transforms; host-witness proof 222.2 -> 211.9 ms on a box whose HBM-resident proof takes 208).
// The host loop is "copy k, launch k", so the overlap also happens with pageable memory, whose asynchronous copies block
// the host.  `dev` ([m][n], caller-owned) receives the uploaded matrix.
static size_t hash_pipeline_min_bytes() {
  const char* e = getenv("VX_HASH_PIPELINE_MIN_BYTES");   // tests lower it to exercise the carried-state kernel on small batches
  return e ? (size_t)strtoull(e, nullptr, 10) : ((size_t)64 << 20);   // below ~64 MB the transfer is too short to matter
}
static int batch_commit_host(vx_ctx* c, vx_batch* b, const u64* host, u64* dev, bool is_coeffs) {
  using namespace vxh;
  const size_t n = b->n(), m = b->ncols, N = b->rows();
  if (!c->copy_stream) HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  HIPCHK(hipStreamSynchronize(c->stream));  // `dev` and the batch buffers may be recycled blocks the main stream still owns
  const bool pipelined = m >= 64 && n * m * 8 >= hash_pipeline_min_bytes();
  std::vector<size_t> starts;                // column blocks: 16 each; pipelined: a first block of 8 so that hashing starts early
  for (size_t c0 = 0; c0 < m; c0 += (pipelined && c0 == 0) ? 8 : 16) starts.push_back(c0);
  const size_t nblocks = starts.size();
  // the sponge is cut after these many columns (multiples of the rate, on block boundaries): 8, 56, m for the prover's batches
  // (<= 160 columns); a wide STARK trace in a dozen pieces so that only the last one is left when its upload ends
  std::vector<size_t> group_end;
  if (pipelined) {
    const size_t step = m <= 160 ? 48 : 16 * ((m - 8 + 16 * 12 - 1) / (16 * 12));
    for (size_t e = 8; e < m && (m > 160 || group_end.size() < 2); e += step) group_end.push_back(e);
    group_end.push_back(m);
  }
  u64* state = nullptr;
  if (pipelined && c->pool_alloc((void**)&state, 12 * N * 8) != hipSuccess) return vx_fail(VX_E_NOMEM, "commit: out of device memory (sponge state)");
  std::vector<hipEvent_t> ev(nblocks, nullptr);
  const u64 ninv = inv((u64)n % P);
  int rc = VX_OK, group = 0;
  for (size_t bk = 0; bk < nblocks && rc == VX_OK; ++bk) {
    const size_t c0 = starts[bk], nc = (bk + 1 < nblocks ? starts[bk + 1] : m) - c0;
    if (hipEventCreateWithFlags(&ev[bk], hipEventDisableTiming) != hipSuccess ||
        hipMemcpyAsync(dev + c0 * n, host + c0 * n, nc * n * 8, hipMemcpyHostToDevice, c->copy_stream) != hipSuccess ||
        hipEventRecord(ev[bk], c->copy_stream) != hipSuccess || hipStreamWaitEvent(c->stream, ev[bk], 0) != hipSuccess)
      rc = vx_fail(VX_E_HIP, "upload of a column block failed");
    if (rc == VX_OK && is_coeffs) {
      ProfScope ps(c, "bitrev_permute", 16.0 * (double)n * (double)nc);
      hipLaunchKernelGGL(bitrev_permute_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)nc), dim3(256), 0, c->stream, dev + c0 * n,
                         b->coeffs + c0 * n, b->log_n, n, n);
    } else if (rc == VX_OK) {
      rc = run_ntt(c, dev + c0 * n, b->coeffs + c0 * n, n, n, 0, 0, b->log_n, nc, 1, true, false, nullptr, 0, ninv, "intt",
                   16.0 * (double)n * (double)nc);
    }
    if (rc == VX_OK) rc = batch_lde_cols(c, b, c0, nc);
    if (rc == VX_OK && pipelined && c0 + nc == group_end[group]) {   // (group < group_end.size(): the last entry is m)
      const size_t g0 = group ? group_end[group - 1] : 0, g1 = group_end[group];
      const bool first = group == 0, last = group + 1 == (int)group_end.size();
      ProfScope ps(c, "hash_leaves", (double)(g1 - g0) * 8.0 * (double)N + (first ? 0.0 : 96.0 * N) + (last ? 0.0 : 96.0 * N));
      hipLaunchKernelGGL(hash_leaves_colmajor_part_kernel, dim3((unsigned)((N + HASH_THREADS - 1) / HASH_THREADS)), dim3(HASH_THREADS), 0, c->stream,
                         b->lde, N, N, (int)g0, (int)g1, state, first ? 1 : 0, last ? 1 : 0, b->tree);
      if (hipGetLastError() != hipSuccess) rc = vx_fail(VX_E_HIP, "leaf hashing launch failed");
      ++group;
    }
  }
  hipStreamSynchronize(c->copy_stream);
  for (hipEvent_t e : ev)
    if (e) hipEventDestroy(e);
  c->pool_free(state);   // stream-ordered reuse, like every pool block
  VXCHK(rc);
  if (pipelined) return build_merkle_levels(c, b->tree, N, b->local_cap_height(), &b->cap_off);
  return batch_hash_tree(c, b);
}

// ------------------------------------------------------------------------------------------------
// Evaluation of every column at an extension point (plonk/proof.rs OpeningSet::new ->
// PolynomialCoeffs::to_extension().eval(zeta)).  Coefficients are in bit-reversed order, so the
// kernel pairs position pos with zeta^rev(pos) from a table built once per zeta.
// ------------------------------------------------------------------------------------------------
__global__ void zeta_table_kernel(const u64* __restrict__ pows /* [log_n][2]: zeta^(2^b) */, int log_n, u64* __restrict__ ztab) {
  size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >> log_n) return;
  u32 j = bitrev32((u32)pos, log_n);
  ext2 acc = ext_make(1, 0);
  for (int b = 0; b < log_n; ++b)
    if ((j >> b) & 1) acc = ext_mul(acc, ext_make(pows[2 * b], pows[2 * b + 1]));
  ztab[2 * pos] = acc.a;
  ztab[2 * pos + 1] = acc.b;
}

static int build_zeta_table(vx_ctx* c, vxh::Ext zeta, int log_n, u64* ztab) {
  std::vector<u64> pows(2 * (log_n > 0 ? log_n : 1));
  vxh::Ext p = zeta;
  for (int b = 0; b < log_n; ++b) {
    pows[2 * b] = p.a;
    pows[2 * b + 1] = p.b;
    p = vxh::emul(p, p);
  }
  // (from the context's pool: hipMalloc / hipFree per proof cost tens of microseconds each, and hipFree waits for the WHOLE device —
  // the other lanes of the process included)
  void* dv = nullptr;
  if (c->pool_alloc(&dv, pows.size() * 8) != hipSuccess) return vx_fail(VX_E_NOMEM, "build_zeta_table: out of device memory");
  u64* d = (u64*)dv;
  hipError_t e = hipMemcpyAsync(d, pows.data(), pows.size() * 8, hipMemcpyHostToDevice, c->stream);
  size_t n = (size_t)1 << log_n;
  if (e == hipSuccess) {
    ProfScope ps(c, "zeta_table");
    hipLaunchKernelGGL(zeta_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d, log_n, ztab);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // `pows` is a stack-scoped source of the copy
  c->pool_free(dv);
  if (e != hipSuccess) return vx_fail(VX_E_HIP, "build_zeta_table: %s", hipGetErrorString(e));
  return VX_OK;
}

#define EVAL_BLOCKS 128
#define EVAL_COLS 8
__global__ __launch_bounds__(256) void eval_ext_kernel(const u64* __restrict__ coeffs, size_t n, size_t ncols,
                                                       const u64* __restrict__ ztab, u64* __restrict__ partial) {
  const size_t c0 = (size_t)blockIdx.y * EVAL_COLS;
  u64 aa[EVAL_COLS], ab[EVAL_COLS];
#pragma unroll
  for (int g = 0; g < EVAL_COLS; ++g) aa[g] = ab[g] = 0;
  for (size_t pos = (size_t)blockIdx.x * 256 + threadIdx.x; pos < n; pos += (size_t)EVAL_BLOCKS * 256) {
    const ulonglong2 z = reinterpret_cast<const ulonglong2*>(ztab)[pos];
#pragma unroll
    for (int g = 0; g < EVAL_COLS; ++g) {
      if (c0 + g < ncols) {
        u64 v = coeffs[(c0 + g) * n + pos];
        aa[g] = gl_mad(v, z.x, aa[g]);
        ab[g] = gl_mad(v, z.y, ab[g]);
      }
    }
  }
  __shared__ u64 red[4][EVAL_COLS][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int g = 0; g < EVAL_COLS; ++g) {
    u64 a = aa[g], b = ab[g];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      a = gl_add(a, __shfl_down(a, off, 64));
      b = gl_add(b, __shfl_down(b, off, 64));
    }
    if (lane == 0) {
      red[wave][g][0] = a;
      red[wave][g][1] = b;
    }
  }
  __syncthreads();
  if (threadIdx.x < EVAL_COLS * 2) {
    int g = threadIdx.x >> 1, k = threadIdx.x & 1;
    u64 s = gl_add(gl_add(red[0][g][k], red[1][g][k]), gl_add(red[2][g][k], red[3][g][k]));
    if (c0 + g < ncols) partial[((c0 + g) * EVAL_BLOCKS + blockIdx.x) * 2 + k] = s;
  }
}

// partial[(col * EVAL_BLOCKS + block) * 2 + k] -> out[col * 2 + k]: the sum over the blocks, on the device
__global__ __launch_bounds__(256) void eval_ext_reduce_kernel(const u64* __restrict__ partial, size_t ncols, u64* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= 2 * ncols) return;
  const size_t col = t >> 1, k = t & 1;
  u64 s = 0;
  for (int b = 0; b < EVAL_BLOCKS; ++b) s = gl_add(s, partial[(col * EVAL_BLOCKS + b) * 2 + k]);
  out[t] = s;
}

// Every opening of one proof in ONE submission (round 5): the power tables of both points, one evaluation launch per (oracle,
// point), the per-block partial sums added up on the device, one copy, one synchronisation.  The round-1 form — a table, then per
// oracle: launch, copy 128 partial sums per column, synchronise, add them on the host — cost a lone 2^16-row proof ~0.3 ms of host
// time for 0.12 ms of kernels (host timeline of `vx_prove`, 7 synchronisations in a row).
struct EvalJob {
  const u64* coeffs;   // [ncols][n]
  size_t ncols;
  int point;           // 0: zeta, 1: g * zeta
  uint64_t* out_host;  // [ncols][2]
};
// digest_out (optional): the tree hash of all the values in job order (include/vxprover.h VX_STARK_OPENINGS_DIGEST), computed on the
// device from the buffer the evaluations land in: one leaf-hash launch + the fused tree top, inside the same submission.
static int build_merkle_levels(vx_ctx* c, u64* tree, size_t n_leaves, int cap_height, size_t* cap_offset_digests);
static int batch_eval_ext_many(vx_ctx* c, vxh::Ext z0, vxh::Ext z1, int log_n, const EvalJob* jobs, int njobs, uint64_t* digest_out = nullptr) {
  const size_t n = (size_t)1 << log_n;
  const int lp = log_n > 0 ? log_n : 1;
  std::vector<u64> pows(4 * (size_t)lp);
  vxh::Ext pz[2] = {z0, z1};
  for (int q = 0; q < 2; ++q) {
    vxh::Ext p = pz[q];
    for (int b = 0; b < log_n; ++b) {
      pows[(size_t)q * 2 * lp + 2 * b] = p.a;
      pows[(size_t)q * 2 * lp + 2 * b + 1] = p.b;
      p = vxh::emul(p, p);
    }
  }
  size_t total = 0;
  bool need[2] = {false, false};
  for (int j = 0; j < njobs; ++j) {
    total += jobs[j].ncols;
    if (jobs[j].ncols) need[jobs[j].point & 1] = true;
  }
  if (!total) return VX_OK;
  void *d_pows = nullptr, *ztab[2] = {nullptr, nullptr}, *partial = nullptr, *d_out = nullptr, *d_tree = nullptr;
  auto release = [&] { c->pool_free(d_pows), c->pool_free(ztab[0]), c->pool_free(ztab[1]), c->pool_free(partial), c->pool_free(d_out), c->pool_free(d_tree); };
  size_t leaves = 2;                                   // digest: the values, zero-padded to 8 * leaves elements
  while (leaves * 8 < total * 2) leaves <<= 1;
  const size_t out_words = digest_out ? leaves * 8 : total * 2;
  bool ok = c->pool_alloc(&d_pows, pows.size() * 8) == hipSuccess && c->pool_alloc(&partial, total * EVAL_BLOCKS * 16) == hipSuccess &&
            c->pool_alloc(&d_out, out_words * 8) == hipSuccess;
  if (ok && digest_out) ok = c->pool_alloc(&d_tree, (2 * leaves) * 32) == hipSuccess;
  for (int q = 0; q < 2 && ok; ++q)
    if (need[q]) ok = c->pool_alloc(&ztab[q], n * 16) == hipSuccess;
  if (!ok) {
    release();
    return vx_fail(VX_E_NOMEM, "openings: out of device memory");
  }
  std::vector<u64> h(total * 2);
  hipError_t e = hipMemcpyAsync(d_pows, pows.data(), pows.size() * 8, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    ProfScope ps(c, "zeta_table");
    for (int q = 0; q < 2; ++q)
      if (need[q])
        hipLaunchKernelGGL(zeta_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const u64*)d_pows + (size_t)q * 2 * lp, log_n,
                           (u64*)ztab[q]);
  }
  if (e == hipSuccess) {
    size_t off = 0;
    double bytes = 0;
    for (int j = 0; j < njobs; ++j) bytes += 8.0 * (double)n * (double)jobs[j].ncols;
    ProfScope ps(c, "eval_ext", bytes);
    for (int j = 0; j < njobs; ++j) {
      if (!jobs[j].ncols) continue;
      hipLaunchKernelGGL(eval_ext_kernel, dim3(EVAL_BLOCKS, (unsigned)((jobs[j].ncols + EVAL_COLS - 1) / EVAL_COLS)), dim3(256), 0, c->stream, jobs[j].coeffs, n,
                         jobs[j].ncols, (const u64*)ztab[jobs[j].point & 1], (u64*)partial + off * EVAL_BLOCKS * 2);
      off += jobs[j].ncols;
    }
    if (digest_out) (void)hipMemsetAsync((u64*)d_out + total * 2, 0, (out_words - total * 2) * 8, c->stream);
    hipLaunchKernelGGL(eval_ext_reduce_kernel, dim3((unsigned)((2 * total + 255) / 256)), dim3(256), 0, c->stream, (const u64*)partial, total, (u64*)d_out);
    e = hipGetLastError();
  }
  size_t root_off = 0;
  if (e == hipSuccess && digest_out) {
    ProfScope ps(c, "openings_digest", 8.0 * (double)out_words);
    hipLaunchKernelGGL(hash_leaves_rowmajor_kernel, dim3((unsigned)((leaves + HASH_THREADS - 1) / HASH_THREADS)), dim3(HASH_THREADS), 0, c->stream,
                       (const u64*)d_out, leaves, 8, (u64*)d_tree);
    ps.end();
    if (build_merkle_levels(c, (u64*)d_tree, leaves, 0, &root_off) != VX_OK) e = hipErrorUnknown;
  }
  if (e == hipSuccess && digest_out) e = hipMemcpyAsync(digest_out, (u64*)d_tree + root_off * 4, 32, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // `pows` and `h` are stack-scoped ends of the copies
  release();
  if (e != hipSuccess) return vx_fail(VX_E_HIP, "openings: %s", hipGetErrorString(e));
  size_t off = 0;
  for (int j = 0; j < njobs; ++j) {
    if (jobs[j].ncols) memcpy(jobs[j].out_host, h.data() + 2 * off, jobs[j].ncols * 16);
    off += jobs[j].ncols;
  }
  return VX_OK;
}

static int batch_eval_ext(vx_ctx* c, const u64* coeffs, size_t n, int log_n, size_t ncols, const u64* ztab, uint64_t* out_host) {
  (void)log_n;
  void* pv = nullptr;
  if (c->pool_alloc(&pv, ncols * EVAL_BLOCKS * 16) != hipSuccess) return vx_fail(VX_E_NOMEM, "batch_eval_ext: out of device memory");
  u64* partial = (u64*)pv;
  {
    ProfScope ps(c, "eval_ext", 8.0 * (double)n * (double)ncols);
    hipLaunchKernelGGL(eval_ext_kernel, dim3(EVAL_BLOCKS, (unsigned)((ncols + EVAL_COLS - 1) / EVAL_COLS)), dim3(256), 0,
                       c->stream, coeffs, n, ncols, ztab, partial);
  }
  std::vector<u64> h(ncols * EVAL_BLOCKS * 2);
  hipError_t e = hipMemcpyAsync(h.data(), partial, h.size() * 8, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  c->pool_free(pv);
  if (e != hipSuccess) return vx_fail(VX_E_HIP, "batch_eval_ext: %s", hipGetErrorString(e));
  for (size_t col = 0; col < ncols; ++col) {
    u64 a = 0, b = 0;
    for (int k = 0; k < EVAL_BLOCKS; ++k) {
      a = vxh::add(a, h[(col * EVAL_BLOCKS + k) * 2]);
      b = vxh::add(b, h[(col * EVAL_BLOCKS + k) * 2 + 1]);
    }
    out_host[2 * col] = a;
    out_host[2 * col + 1] = b;
  }
  return VX_OK;
}
