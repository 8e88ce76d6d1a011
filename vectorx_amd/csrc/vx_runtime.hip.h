// Runtime plumbing of libvxprover.so: error reporting, context (device + stream + root tables),
// HIP-event profiler, NTT planner/launcher, Merkle builder.  Host code around hand-written kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <string>
#include <vector>
#include "../../include/vxprover.h"
#include "host_field.h"
#include "merkle.hip.h"
#include "ntt.hip.h"
#include "ntt2.hip.h"

static thread_local char g_err[512] = "";
static int vx_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}
#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess)                                                                         \
      return vx_fail(_e == hipErrorOutOfMemory ? VX_E_NOMEM : VX_E_HIP, "%s failed: %s (%s:%d)",  \
                     #expr, hipGetErrorString(_e), __FILE__, __LINE__);                           \
  } while (0)
#define VXCHK(expr)        \
  do {                     \
    int _r = (expr);       \
    if (_r != VX_OK) return _r; \
  } while (0)

struct ProfEntry {
  double ms = 0, alg_bytes = 0;
  uint64_t calls = 0;
};
struct ProfPending {
  std::string name;
  hipEvent_t a, b;
  double alg_bytes;
};

struct vx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;  // host->device witness upload, overlapped with the first transforms (prover.hip.h)
  u64* root_lo = nullptr;  // w_{2^24}^k
  u64* root_hi = nullptr;  // w_{2^24}^(4096k)
  unsigned* merkle_counters = nullptr;  // merkle_top_kernel: arrival counters per cap subtree, zero between launches
  u64* hash_clk = nullptr; // {shader ticks, 100 MHz ticks} samples written by hash_leaves_colmajor_kernel while profiling
  bool prof_on = false;
  bool rehearsal = false;  // vx_circuit_warm: a proof of an all-zero witness that only exists to prime the pool and the kernel caches
  std::vector<ProfPending> pending;
  std::vector<hipEvent_t> event_pool;
  std::map<std::string, ProfEntry> prof;
  std::vector<std::string> prof_order;
  std::map<std::string, u64*> scale_cache;  // device scale tables keyed by (log_n, bits, shifts)
  hipDeviceProp_t props;
  // Size-bucketed caching allocator: a prover re-uses the same multi-GB shapes proof after proof, and
  // hipMalloc/hipFree of 20 GB costs hundreds of ms.  Blocks are recycled by exact size; everything
  // runs on the context's single stream, so reuse is stream-ordered and safe without extra syncs.
  std::multimap<size_t, void*> free_blocks;
  std::map<void*, size_t> live_blocks;
  size_t pooled_bytes = 0;
  hipError_t pool_alloc(void** out, size_t bytes) {
    if (bytes == 0) bytes = 8;
    bytes = (bytes + 255) & ~(size_t)255;
    auto it = free_blocks.find(bytes);
    if (it != free_blocks.end()) {
      *out = it->second;
      free_blocks.erase(it);
      pooled_bytes -= bytes;
      live_blocks[*out] = bytes;
      return hipSuccess;
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess && !free_blocks.empty()) {
      (void)hipGetLastError();
      pool_trim();
      e = hipMalloc(out, bytes);
    }
    if (e == hipSuccess) live_blocks[*out] = bytes;
    return e;
  }
  void pool_free(void* p) {
    if (!p) return;
    auto it = live_blocks.find(p);
    if (it == live_blocks.end()) {
      hipFree(p);
      return;
    }
    free_blocks.insert({it->second, p});
    pooled_bytes += it->second;
    live_blocks.erase(it);
  }
  void pool_trim() {
    hipStreamSynchronize(stream);
    for (auto& kv : free_blocks) hipFree(kv.second);
    free_blocks.clear();
    pooled_bytes = 0;
  }

  hipEvent_t get_event() {
    if (!event_pool.empty()) {
      hipEvent_t e = event_pool.back();
      event_pool.pop_back();
      return e;
    }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
  }
  // a stage measured on the HOST clock (an exchange the host performs between two stream segments): kept apart from the
  // HIP-event stages so that waiting for other ranks is never booked as compute
  void prof_add_host(const char* name, double ms, double bytes) {
    if (!prof_on) return;
    if (!prof.count(name)) prof_order.push_back(name);
    ProfEntry& e = prof[name];
    e.ms += ms;
    e.calls += 1;
    e.alg_bytes += bytes;
  }
  void fold() {
    for (auto& p : pending) {
      hipEventSynchronize(p.b);
      float ms = 0;
      hipEventElapsedTime(&ms, p.a, p.b);
      if (!prof.count(p.name)) prof_order.push_back(p.name);
      ProfEntry& e = prof[p.name];
      e.ms += ms;
      e.calls += 1;
      e.alg_bytes += p.alg_bytes;
      event_pool.push_back(p.a);
      event_pool.push_back(p.b);
    }
    pending.clear();
  }
};

// RAII bracket: events on the context's own stream around one kernel family.
struct ProfScope {
  vx_ctx* c;
  hipEvent_t a{}, b{};
  const char* name;
  double bytes;
  ProfScope(vx_ctx* ctx, const char* n, double alg_bytes = 0) : c(ctx), name(n), bytes(alg_bytes) {
    if (c->prof_on) {
      a = c->get_event();
      b = c->get_event();
      hipEventRecord(a, c->stream);
    }
  }
  bool ended = false;
  void end() {   // close the bracket early (something that is not this stage follows inside the C++ scope)
    if (c->prof_on && !ended) {
      hipEventRecord(b, c->stream);
      c->pending.push_back({name, a, b, bytes});
    }
    ended = true;
  }
  ~ProfScope() { end(); }
};

// ------------------------------------------------------------------------------------------------
// NTT planning / launching
// ------------------------------------------------------------------------------------------------
struct NttPass {
  int r_log, t_log, b_lo;
};
static std::vector<NttPass> plan_ntt(int log_n) {
  std::vector<NttPass> v;
  if (log_n <= 12) {
    v.push_back({log_n, 0, 0});
    return v;
  }
  // 13 <= log_n <= 18 (the map / reduce proofs of a header_range DAG, 2^16 and 2^18 rows; chip-sized STARK traces): 8 strided stages + a contiguous pass of
  // log_n - 8, so that BOTH passes run the second-generation kernel on full 8192-element tiles; with an 11-stage final pass the
  // strided one would have 5..7 stages and fall back to the generic kernel, which made the NTTs 39 % of the GPU time of a DAG
  const int r_final = (log_n >= 13 && log_n <= 18 && !getenv("VX_NTT_V1")) ? log_n - 8 : 11;
  int rem = log_n - r_final;
  int k = (rem + 9) / 10;  // strided passes of <= 10 stages
  int b = log_n;
  for (int i = 0; i < k; ++i) {
    int r = (rem + (k - i) - 1) / (k - i);
    rem -= r;
    b -= r;
    int t = NTT_MAX_TILE_LOG - r;
    if (t > 4 && !(k == 1 && r >= 8)) t = 4;  // single strided pass of 8..10 stages: full 8192-element tile (ntt2 kernel)
    v.push_back({r, t, b});
  }
  int t = NTT_MAX_TILE_LOG - r_final;
  if (t > log_n - r_final) t = log_n - r_final;
  v.push_back({r_final, t, 0});
  return v;
}

template <int R>
static hipError_t launch_ntt_pass_r(const NttPassParams& p, dim3 grid, size_t lds, hipStream_t s) {
  static bool attr_set[16] = {};
  int dev = 0;
  hipGetDevice(&dev);
  if (!attr_set[dev & 15]) {
    hipError_t e = hipFuncSetAttribute((const void*)ntt_pass_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       100 * 1024);
    if (e != hipSuccess) return e;
    attr_set[dev & 15] = true;
  }
  hipLaunchKernelGGL(ntt_pass_kernel<R>, grid, dim3(NTT_THREADS), lds, s, p);
  return hipGetLastError();
}
static hipError_t launch_ntt_pass(int r_log, const NttPassParams& p, dim3 grid, size_t lds, hipStream_t s) {
  switch (r_log) {
    case 1: return launch_ntt_pass_r<1>(p, grid, lds, s);
    case 2: return launch_ntt_pass_r<2>(p, grid, lds, s);
    case 3: return launch_ntt_pass_r<3>(p, grid, lds, s);
    case 4: return launch_ntt_pass_r<4>(p, grid, lds, s);
    case 5: return launch_ntt_pass_r<5>(p, grid, lds, s);
    case 6: return launch_ntt_pass_r<6>(p, grid, lds, s);
    case 7: return launch_ntt_pass_r<7>(p, grid, lds, s);
    case 8: return launch_ntt_pass_r<8>(p, grid, lds, s);
    case 9: return launch_ntt_pass_r<9>(p, grid, lds, s);
    case 10: return launch_ntt_pass_r<10>(p, grid, lds, s);
    case 11: return launch_ntt_pass_r<11>(p, grid, lds, s);
    case 12: return launch_ntt_pass_r<12>(p, grid, lds, s);
    default: return hipErrorInvalidValue;
  }
}

// ---- second-generation pass (ntt2.hip.h) for full 8192-element tiles --------------------------------------
static bool ntt2_eligible(const NttPass& ps, bool strided) {
  static const bool disabled = getenv("VX_NTT_V1") != nullptr;  // A/B switch: force the generic kernel
  if (disabled || ps.r_log + ps.t_log != NTT2_TILE_LOG) return false;
  return strided ? (ps.r_log >= 8 && ps.r_log <= 10 && ps.b_lo >= ps.t_log) : (ps.r_log >= 5 && ps.r_log <= 11);
}
template <int R, int E2, int E3, bool STRIDED, bool IN_BITREV, bool PRE, bool INV>
static hipError_t launch_ntt2_k(const Ntt2Params& q, dim3 grid, hipStream_t s) {
  auto kern = ntt2_pass_kernel<R, 4, E2, E3, STRIDED, IN_BITREV, PRE, INV>;
  static bool attr_set[16] = {};
  int dev = 0;
  hipGetDevice(&dev);
  const size_t tile = (size_t)1 << NTT2_TILE_LOG;
  const size_t lds = (tile + (tile >> 5) + (tile >> 9) + 1 + (R > 10 ? ((size_t)1 << (R - 1)) : ((size_t)1 << R))) * 8 + 16;
  if (!attr_set[dev & 15]) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    if (e != hipSuccess) return e;
    attr_set[dev & 15] = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(NTT2_THREADS), lds, s, q);
  return hipGetLastError();
}
template <int R, int E2, int E3, bool STRIDED>
static hipError_t launch_ntt2_r(const NttPassParams& p, const Ntt2Params& q, dim3 grid, hipStream_t s) {
  const bool inv = p.inverse != 0, br = p.in_bitrev != 0, pre = p.pre != nullptr;
  if (!STRIDED) {  // final contiguous pass: never first, never prescaled
    return inv ? launch_ntt2_k<R, E2, E3, STRIDED, false, false, true>(q, grid, s)
               : launch_ntt2_k<R, E2, E3, STRIDED, false, false, false>(q, grid, s);
  }
  if (!br && !pre) return inv ? launch_ntt2_k<R, E2, E3, STRIDED, false, false, true>(q, grid, s)
                              : launch_ntt2_k<R, E2, E3, STRIDED, false, false, false>(q, grid, s);
  if (!br && pre && !inv) return launch_ntt2_k<R, E2, E3, STRIDED, false, true, false>(q, grid, s);
  if (br && pre && !inv) return launch_ntt2_k<R, E2, E3, STRIDED, true, true, false>(q, grid, s);
  if (br && !pre && inv) return launch_ntt2_k<R, E2, E3, STRIDED, true, false, true>(q, grid, s);
  return hipErrorInvalidValue;  // combination not instantiated: caller falls back
}
// Full-size inter-pass twiddle table of the second-generation strided pass (ntt2.hip.h: tw_full), built on first use and kept for the life
// of the context.
static const u64* ntt2_tw_table(vx_ctx* c, const NttPassParams& p, int r_log) {
  const int span_log = p.b_lo + r_log;
  const std::string key = "tw:" + std::to_string(span_log) + ":" + std::to_string(r_log) + ":" + std::to_string(p.inverse) + ":" + std::to_string(p.post_scale);
  auto it = c->scale_cache.find(key);
  if (it != c->scale_cache.end()) return it->second;
  u64* d = nullptr;
  const size_t n = (size_t)1 << span_log;
  if (hipMalloc(&d, n * 8) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;                       // no memory for the table: the kernel composes the factors as before
  }
  hipLaunchKernelGGL(ntt2_tw_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d, span_log, r_log, p.b_lo, p.inverse ? 1 : 0,
                     p.post_scale, p.root_lo, p.root_hi);
  c->scale_cache[key] = d;
  return d;
}
static hipError_t launch_ntt2_pass(vx_ctx* c, const NttPass& ps, const NttPassParams& p, dim3 grid, hipStream_t s) {
  static const bool no_tw_table = getenv("VX_NTT2_NO_TW_TABLE") != nullptr;   // A/B switch
  Ntt2Params q;
  q.in = p.in;
  q.out = p.out;
  q.in_col_stride = p.in_col_stride;
  q.out_col_stride = p.out_col_stride;
  q.in_z_stride = p.in_z_stride;
  q.out_z_stride = p.out_z_stride;
  q.log_n = p.log_n;
  q.b_lo = p.b_lo;
  q.root_lo = p.root_lo;
  q.root_hi = p.root_hi;
  q.pre = p.pre;
  q.pre_beta = p.pre ? p.pre + (size_t)grid.z * (((size_t)1 << (p.log_n - p.pre_bits)) + ((size_t)1 << p.pre_bits)) : nullptr;
  q.pre_bits = p.pre_bits;
  q.post_scale = p.post_scale;
  q.nz_fold = 0;
  q.tw_full = nullptr;
  const bool shared_tiles = p.pre && grid.z > 1 && p.in_z_stride == 0 && grid.x % 8 == 0;   // coset LDE from shared coefficients
  // The twiddle table pays where its slices are re-read from L2: the nz blocks of a tile of a coset LDE run back to back on one XCD and share
  // the tile's 64 KB slice (LDE -3 %, profiles/r04_ntt_experiment.md); a plain transform would read every slice once per column from HBM
  // (iNTT +8 %), so it keeps composing its factors.
  if (ps.b_lo > 0 && shared_tiles && !no_tw_table) q.tw_full = ntt2_tw_table(c, p, ps.r_log);
  if (shared_tiles && !getenv("VX_NTT_NO_ZFOLD")) {
    q.nz_fold = (int)grid.z;
    grid = dim3(grid.x * grid.z, grid.y, 1);
  }
  if (ps.b_lo > 0) {
    switch (ps.r_log) {
      case 8: return launch_ntt2_r<8, 4, 0, true>(p, q, grid, s);
      case 9: return launch_ntt2_r<9, 3, 2, true>(p, q, grid, s);
      case 10: return launch_ntt2_r<10, 4, 2, true>(p, q, grid, s);
      default: return hipErrorInvalidValue;
    }
  }
  switch (ps.r_log) {   // final contiguous pass
    case 5: return launch_ntt2_r<5, 1, 0, false>(p, q, grid, s);
    case 6: return launch_ntt2_r<6, 2, 0, false>(p, q, grid, s);
    case 7: return launch_ntt2_r<7, 3, 0, false>(p, q, grid, s);
    case 8: return launch_ntt2_r<8, 4, 0, false>(p, q, grid, s);
    case 9: return launch_ntt2_r<9, 3, 2, false>(p, q, grid, s);
    case 10: return launch_ntt2_r<10, 4, 2, false>(p, q, grid, s);
    default: return launch_ntt2_r<11, 4, 3, false>(p, q, grid, s);
  }
}

// Scale tables for x[j] *= base_mul * shift^j, split as hi[j >> bits] * lo[j & mask]; `nz` slices with
// per-slice shift.  Layout per slice: [2^(log_n-bits) hi entries][2^bits lo entries]; hi carries `pre_mul`.
static int build_scale_tables(vx_ctx* c, int log_n, int bits, const std::vector<u64>& shifts, u64 pre_mul,
                              u64** dptr_out) {
  using namespace vxh;
  size_t nh = (size_t)1 << (log_n - bits), nl = (size_t)1 << bits;
  // after the per-slice [hi][lo] tables: 16 entries per slice, shift^(q * n/16) — the factor between the 16 operands
  // of a first-round radix-16 DFT (ntt2.hip.h), so that a thread composes ONE table product and walks these
  std::vector<u64> host(shifts.size() * (nh + nl + 16));
  for (size_t z = 0; z < shifts.size(); ++z) {
    u64* beta = &host[shifts.size() * (nh + nl) + z * 16];
    const u64 b1 = log_n >= 4 ? pow(shifts[z], (u64)1 << (log_n - 4)) : 1;
    beta[0] = 1;
    for (int q = 1; q < 16; ++q) beta[q] = mul(beta[q - 1], b1);
  }
  for (size_t z = 0; z < shifts.size(); ++z) {
    u64* hi = &host[z * (nh + nl)];
    u64* lo = hi + nh;
    u64 s = shifts[z];
    u64 acc = 1;
    for (size_t i = 0; i < nl; ++i) {
      lo[i] = acc;
      acc = mul(acc, s);
    }
    u64 step = acc;  // s^(2^bits)
    acc = pre_mul;
    for (size_t i = 0; i < nh; ++i) {
      hi[i] = acc;
      acc = mul(acc, step);
    }
  }
  u64* d = nullptr;
  HIPCHK(hipMalloc(&d, host.size() * 8));
  HIPCHK(hipMemcpyAsync(d, host.data(), host.size() * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));  // host vector goes out of scope
  *dptr_out = d;
  return VX_OK;
}

// Post-scale kernel for coset_ifft: element at bit-reversed position pos holds c_j, j = rev(pos);
// multiply by hi[j >> bits] * lo[j & mask].
__global__ void scale_bitrev_kernel(u64* __restrict__ data, size_t col_stride, int log_n, const u64* __restrict__ tab,
                                    int bits) {
  size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >> log_n) return;
  u32 j = bitrev32((u32)pos, log_n);
  u64* col = data + (size_t)blockIdx.y * col_stride;
  u64 s = gl_mul(tab[j >> bits], tab[((size_t)1 << (log_n - bits)) + (j & ((1u << bits) - 1))]);
  col[pos] = gl_mul(col[pos], s);
}

// Run a full batched transform of length 2^log_n on `ncols` columns x `nz` slices.
//   in_bitrev : input stored in bit-reversed order (true for device-resident coefficients)
//   pre       : optional prescale tables (device), one slice per z
// Output is in bit-reversed order at `out` (in place after the first pass).
static int run_ntt(vx_ctx* c, const u64* in, u64* out, size_t in_col_stride, size_t out_col_stride, size_t in_z_stride,
                   size_t out_z_stride, int log_n, size_t ncols, int nz, bool inverse, bool in_bitrev, const u64* pre,
                   int pre_bits, u64 post_scale, const char* label, double alg_bytes) {
  if (log_n < 1 || log_n > ROOT_TABLE_LOG) return vx_fail(VX_E_INVALID, "run_ntt: log_n=%d out of range [1,24]", log_n);
  if (ncols == 0) return VX_OK;
  std::vector<NttPass> plan = plan_ntt(log_n);
  ProfScope ps(c, label, alg_bytes);
  for (size_t i = 0; i < plan.size(); ++i) {
    const NttPass& ps_ = plan[i];
    NttPassParams p;
    bool first = i == 0, last = i + 1 == plan.size();
    p.in = first ? in : out;
    p.out = out;
    p.in_col_stride = first ? in_col_stride : out_col_stride;
    p.out_col_stride = out_col_stride;
    p.in_z_stride = first ? in_z_stride : out_z_stride;
    p.out_z_stride = out_z_stride;
    p.log_n = log_n;
    p.b_lo = ps_.b_lo;
    p.t_log = ps_.t_log;
    p.in_bitrev = first && in_bitrev;
    p.inverse = inverse;
    p.root_lo = c->root_lo;
    p.root_hi = c->root_hi;
    p.pre = first ? pre : nullptr;
    p.pre_bits = pre_bits;
    // the scale commutes with the passes: in a multi-pass transform it rides on the FIRST (strided) pass, where it is
    // folded into the inter-pass twiddle for free, instead of costing a multiply per element in the last pass
    p.post_scale = (plan.size() > 1 ? first : last) ? post_scale : 1;
    size_t tile = (size_t)1 << (ps_.r_log + ps_.t_log);
    size_t lds = (tile + (tile >> 5) + (tile >> 9) + 1 + ((size_t)1 << (ps_.r_log - 1 > 0 ? ps_.r_log - 1 : 0))) * 8 + 16;
    dim3 grid((unsigned)((size_t)1 << (log_n - ps_.r_log - ps_.t_log)), (unsigned)ncols, (unsigned)nz);
    // gridDim.y is limited to 65535; column counts here are < 1000.
    hipError_t e = hipErrorInvalidValue;
    if (ntt2_eligible(ps_, ps_.b_lo > 0)) e = launch_ntt2_pass(c, ps_, p, grid, c->stream);
    if (e == hipErrorInvalidValue) {  // not a hot shape (or a variant that is not instantiated): generic kernel
      (void)hipGetLastError();
      e = launch_ntt_pass(ps_.r_log, p, grid, lds, c->stream);
    }
    if (e != hipSuccess) return vx_fail(VX_E_HIP, "ntt pass launch failed: %s", hipGetErrorString(e));
  }
  return VX_OK;
}

// ------------------------------------------------------------------------------------------------
// Merkle tree over precomputed leaf digests: `tree` holds level 0 (n_leaves digests) followed by each
// parent level down to the cap (2^cap_height digests).  Returns the digest offset of the cap level.
// ------------------------------------------------------------------------------------------------
static size_t merkle_tree_digest_count(size_t n_leaves, int cap_height) {
  size_t total = 0;
  for (size_t n = n_leaves; n >= ((size_t)1 << cap_height); n >>= 1) {
    total += n;
    if (n == 1) break;
  }
  return total;
}
static int build_merkle_levels(vx_ctx* c, u64* tree, size_t n_leaves, int cap_height, size_t* cap_offset_digests) {
  ProfScope ps(c, "merkle_levels");
  size_t off = 0, n = n_leaves;
  const size_t ncap = (size_t)1 << cap_height;
  // VX_MTOP_MAX_CHILDREN: A/B knob — from how many children on the fused top kernel takes over (default: its maximum, 32 768).  The
  // cooperative permutation it runs costs ~5 x the issue slots of the one-thread-per-node form: a lower threshold trades a lone
  // proof's latency (one more launch per level) for issue slots when many proofs share the chip.
  static const size_t mtop_max = [] {
    const char* e = getenv("VX_MTOP_MAX_CHILDREN");
    const size_t v = e ? (size_t)strtoull(e, nullptr, 10) : (size_t)MTOP_MAX_CHILDREN;
    return v > (size_t)MTOP_MAX_CHILDREN ? (size_t)MTOP_MAX_CHILDREN : v;
  }();
  while (n > ncap) {
    if (n <= mtop_max && ncap <= MTOP_MAX_COUNTERS) {
      // every remaining level in one launch (merkle.hip.h: merkle_top_kernel)
      int sub_log = 0;
      while (((size_t)ncap << (sub_log + 1)) <= n) ++sub_log;   // children per cap subtree = 2^sub_log
      const int cw_log = sub_log < 7 ? sub_log : 7;
      hipLaunchKernelGGL(merkle_top_kernel, dim3((unsigned)(n >> cw_log)), dim3(MTOP_THREADS), 0, c->stream, tree, off, (unsigned)n,
                         (unsigned)ncap, cw_log, c->merkle_counters);
      while (n > ncap) {
        off += n;
        n >>= 1;
      }
      break;
    }
    size_t np = n >> 1;
    if (np <= COOP_MAX_NODES && ncap > MTOP_MAX_COUNTERS)  // latency-bound level: 16 lanes per node (only with a cap wider than MTOP_MAX_COUNTERS)
      hipLaunchKernelGGL(merkle_level_coop_kernel, dim3((unsigned)((np * 16 + HASH_THREADS - 1) / HASH_THREADS)),
                         dim3(HASH_THREADS), 0, c->stream, tree + off * 4, tree + (off + n) * 4, np);
    else
      hipLaunchKernelGGL(merkle_level_kernel, dim3((unsigned)((np + HASH_THREADS - 1) / HASH_THREADS)),
                         dim3(HASH_THREADS), 0, c->stream, tree + off * 4, tree + (off + n) * 4, np);
    off += n;
    n = np;
  }
  HIPCHK(hipGetLastError());
  *cap_offset_digests = off;
  return VX_OK;
}
