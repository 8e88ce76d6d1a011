// libvxprover.so — C ABI (include/vxprover.h) over the hand-written gfx950 kernels.
// There is NO CPU fallback anywhere in this file: without a usable HIP device every compute entry
// point returns VX_E_NO_DEVICE.
#include "vx_runtime.hip.h"
#include "batch.hip.h"
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <errno.h>
#include <thread>
#include <unistd.h>
#include "plonk_kernels.hip.h"
#include "aux.hip.h"
#include "jit.hip.h"
#include "prover.hip.h"
#include "circuit_io.h"
#include "stark.hip.h"
#include "tracegen.hip.h"
#include <memory>
#include <cstddef>
#include "verifier.h"


// ---- shader-clock probe (vx_clock_probe): every wave runs Poseidon permutations — the instruction mix of the kernels that
// take > 65 % of a proof, at the same 4 waves per SIMD — between two readings of s_memtime (shader-clock ticks) and
// s_memrealtime (constant 100 MHz); effective clock = ticks / realtime.  The clock follows the power budget and therefore
// the instruction mix (a pure multiply-add loop settles ~0.4 GHz lower than the permutation), so bench.py prices the
// integer-ALU roofline with the clock measured in its own run under the permutation's own load.
__global__ __launch_bounds__(HASH_THREADS, 4) void clock_probe_kernel(uint64_t* __restrict__ out, int iters) {
  uint64_t t0, r0, t1, r1;
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = (u64)threadIdx.x * 0x9E3779B97F4A7C15ull + i;
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));
  for (int it = 0; it < iters; ++it) poseidon_permute_nc(s);
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
  u64 x = 0;
#pragma unroll
  for (int i = 0; i < 12; ++i) x ^= s[i];
  if ((threadIdx.x & 63) == 0 || x == 0x123456789ull) {
    const size_t w = ((size_t)blockIdx.x * HASH_THREADS + threadIdx.x) >> 6;
    out[2 * w] = t1 - t0;
    out[2 * w + 1] = r1 - r0;
  }
}

extern "C" {

const char* vx_last_error(void) { return g_err; }
const char* vx_version(void) { return "vxprover 0.1 (gfx950)"; }

int vx_device_max_clock_khz(int device) {
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, device) != hipSuccess) {
    (void)hipGetLastError();
    return vx_fail(VX_E_HIP, "vx_device_max_clock_khz: hipDeviceGetAttribute failed");
  }
  return khz;
}
int vx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int vx_ctx_create(int device, vx_ctx** out) {
  if (!out) return vx_fail(VX_E_INVALID, "vx_ctx_create: out is NULL");
  *out = nullptr;
  int n = vx_device_count();
  if (n <= 0) return vx_fail(VX_E_NO_DEVICE, "no HIP device visible (libvxprover has no CPU fallback)");
  if (device < 0 || device >= n) return vx_fail(VX_E_INVALID, "device %d out of range (have %d)", device, n);
  HIPCHK(hipSetDevice(device));
  vx_ctx* c = new vx_ctx();
  c->device = device;
  HIPCHK(hipGetDeviceProperties(&c->props, device));
  HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  // root tables for w = primitive 2^24-th root of unity
  {
    using namespace vxh;
    std::vector<u64> lo(4096), hi(4096);
    u64 w = root_of_unity(ROOT_TABLE_LOG);
    u64 acc = 1;
    for (int i = 0; i < 4096; ++i) {
      lo[i] = acc;
      acc = mul(acc, w);
    }
    u64 w12 = acc;  // w^4096
    acc = 1;
    for (int i = 0; i < 4096; ++i) {
      hi[i] = acc;
      acc = mul(acc, w12);
    }
    HIPCHK(hipMalloc(&c->root_lo, 4096 * 8));
    HIPCHK(hipMalloc(&c->root_hi, 4096 * 8));
    HIPCHK(hipMemcpy(c->root_lo, lo.data(), 4096 * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->root_hi, hi.data(), 4096 * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&c->hash_clk, 2 * HASH_CLK_SLOTS * 8));
    HIPCHK(hipMemset(c->hash_clk, 0, 2 * HASH_CLK_SLOTS * 8));
    HIPCHK(hipMalloc(&c->merkle_counters, MTOP_MAX_COUNTERS * sizeof(unsigned)));
    HIPCHK(hipMemset(c->merkle_counters, 0, MTOP_MAX_COUNTERS * sizeof(unsigned)));
  }
  *out = c;
  return VX_OK;
}

void vx_ctx_destroy(vx_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  c->fold();
  for (auto e : c->event_pool) hipEventDestroy(e);
  for (auto& kv : c->scale_cache) hipFree(kv.second);
  c->pool_trim();
  for (auto& kv : c->live_blocks) hipFree(kv.first);
  hipFree(c->root_lo);
  hipFree(c->root_hi);
  hipFree(c->hash_clk);
  hipFree(c->merkle_counters);
  if (c->copy_stream) hipStreamDestroy(c->copy_stream);
  hipStreamDestroy(c->stream);
  delete c;
}

int vx_ctx_sync(vx_ctx* c) {
  if (!c) return vx_fail(VX_E_INVALID, "ctx is NULL");
  HIPCHK(hipStreamSynchronize(c->stream));
  return VX_OK;
}
int vx_ctx_trim(vx_ctx* c) {
  if (!c) return vx_fail(VX_E_INVALID, "ctx is NULL");
  HIPCHK(hipSetDevice(c->device));
  c->pool_trim();
  return VX_OK;
}
void* vx_ctx_stream(vx_ctx* c) { return c ? (void*)c->stream : nullptr; }

int vx_clock_probe(vx_ctx* c, double* ghz_out) {
  if (!c || !ghz_out) return vx_fail(VX_E_INVALID, "vx_clock_probe: NULL argument");
  HIPCHK(hipSetDevice(c->device));
  // 1. samples taken INSIDE the leaf-hashing launches since the last vx_prof_reset (profiling on): the clock that kernel ran at
  if (c->hash_clk) {
    std::vector<uint64_t> h(2 * HASH_CLK_SLOTS);
    HIPCHK(hipMemcpyAsync(h.data(), c->hash_clk, h.size() * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    double ticks = 0, rt = 0;
    for (int i = 0; i < HASH_CLK_SLOTS; ++i)
      if (h[2 * i + 1]) ticks += (double)h[2 * i], rt += (double)h[2 * i + 1];
    if (rt > 0) {
      *ghz_out = ticks / (rt / 100e6) / 1e9;
      return VX_OK;
    }
  }
  // 2. no samples: a stand-alone probe under the same instruction mix
  const int blocks = c->props.multiProcessorCount * 4, waves = blocks * (HASH_THREADS / 64);
  void* d = nullptr;
  if (c->pool_alloc(&d, (size_t)waves * 16) != hipSuccess) return vx_fail(VX_E_NOMEM, "vx_clock_probe: out of device memory");
  hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(HASH_THREADS), 0, c->stream, (uint64_t*)d, 8);   // warm the clock up
  hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(HASH_THREADS), 0, c->stream, (uint64_t*)d, 48);  // ~4 ms, measured
  std::vector<uint64_t> h((size_t)waves * 2);
  hipError_t e = hipMemcpyAsync(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost, c->stream);
  hipError_t e2 = hipStreamSynchronize(c->stream);
  c->pool_free(d);
  if (e != hipSuccess || e2 != hipSuccess) return vx_fail(VX_E_HIP, "vx_clock_probe: %s", hipGetErrorString(e != hipSuccess ? e : e2));
  double ticks = 0, rt = 0;
  for (int w = 0; w < waves; ++w) ticks += (double)h[2 * w], rt += (double)h[2 * w + 1];
  if (rt <= 0) return vx_fail(VX_E_HIP, "vx_clock_probe: s_memrealtime did not advance");
  *ghz_out = ticks / (rt / 100e6) / 1e9;
  return VX_OK;
}

int vx_prof_enable(vx_ctx* c, int enable) {
  if (!c) return vx_fail(VX_E_INVALID, "ctx is NULL");
  c->prof_on = enable != 0;
  return VX_OK;
}
int vx_prof_reset(vx_ctx* c) {
  if (!c) return vx_fail(VX_E_INVALID, "ctx is NULL");
  HIPCHK(hipStreamSynchronize(c->stream));
  c->fold();
  c->prof.clear();
  c->prof_order.clear();
  if (c->hash_clk) HIPCHK(hipMemsetAsync(c->hash_clk, 0, 2 * HASH_CLK_SLOTS * 8, c->stream));  // clock samples of the hash launches
  return VX_OK;
}
int vx_prof_count(vx_ctx* c) {
  if (!c) return vx_fail(VX_E_INVALID, "ctx is NULL");
  if (hipStreamSynchronize(c->stream) != hipSuccess) return vx_fail(VX_E_HIP, "stream sync failed");
  c->fold();
  return (int)c->prof_order.size();
}
int vx_prof_get(vx_ctx* c, int index, char* name_out, size_t name_cap, double* ms_out, uint64_t* calls_out,
                double* alg_bytes_out) {
  if (!c) return vx_fail(VX_E_INVALID, "ctx is NULL");
  if (index < 0 || index >= (int)c->prof_order.size()) return vx_fail(VX_E_INVALID, "prof index out of range");
  const std::string& nm = c->prof_order[index];
  const ProfEntry& e = c->prof[nm];
  if (name_out && name_cap) {
    strncpy(name_out, nm.c_str(), name_cap - 1);
    name_out[name_cap - 1] = 0;
  }
  if (ms_out) *ms_out = e.ms;
  if (calls_out) *calls_out = e.calls;
  if (alg_bytes_out) *alg_bytes_out = e.alg_bytes;
  return VX_OK;
}

int vx_dev_alloc(vx_ctx* c, size_t bytes, void** dptr) {
  if (!c || !dptr) return vx_fail(VX_E_INVALID, "vx_dev_alloc: NULL argument");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMalloc(dptr, bytes ? bytes : 8));
  return VX_OK;
}
int vx_dev_free(vx_ctx* c, void* dptr) {
  if (!c) return vx_fail(VX_E_INVALID, "ctx is NULL");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipFree(dptr));
  return VX_OK;
}
int vx_host_alloc(vx_ctx* c, size_t bytes, void** hptr) {
  if (!c || !hptr) return vx_fail(VX_E_INVALID, "vx_host_alloc: NULL argument");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipHostMalloc(hptr, bytes ? bytes : 8, hipHostMallocDefault));
  return VX_OK;
}
int vx_host_free(vx_ctx* c, void* hptr) {
  if (!c) return vx_fail(VX_E_INVALID, "ctx is NULL");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipHostFree(hptr));
  return VX_OK;
}
int vx_dev_upload(vx_ctx* c, void* dptr, const void* host, size_t bytes) {
  if (!c || !dptr || !host) return vx_fail(VX_E_INVALID, "vx_dev_upload: NULL argument");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemcpyAsync(dptr, host, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return VX_OK;
}
int vx_dev_upload_strided(vx_ctx* c, void* dptr, size_t dst_stride_bytes, const uint64_t* host, size_t count) {
  if (!c || !dptr || !host) return vx_fail(VX_E_INVALID, "vx_dev_upload_strided: NULL argument");
  if (dst_stride_bytes < 8) return vx_fail(VX_E_INVALID, "vx_dev_upload_strided: stride must be >= 8");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemcpy2DAsync(dptr, dst_stride_bytes, host, 8, 8, count, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return VX_OK;
}
int vx_dev_download(vx_ctx* c, void* host, const void* dptr, size_t bytes) {
  if (!c || !dptr || !host) return vx_fail(VX_E_INVALID, "vx_dev_download: NULL argument");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemcpyAsync(host, dptr, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return VX_OK;
}

// ---------------------------------------------------------------------------------------------
// L1
// ---------------------------------------------------------------------------------------------
int vx_ntt_batch_dev(vx_ctx* c, const uint64_t* src, uint64_t* dst, int log_n, size_t ncols, int kind, uint64_t shift) {
  if (!c || !src || !dst) return vx_fail(VX_E_INVALID, "vx_ntt_batch_dev: NULL argument");
  if (kind < 0 || kind > 3) return vx_fail(VX_E_INVALID, "vx_ntt_batch_dev: kind %d", kind);
  HIPCHK(hipSetDevice(c->device));
  return ntt_natural_to_bitrev(c, src, dst, log_n, ncols, kind, shift);
}

int vx_ntt_batch(vx_ctx* c, uint64_t* data, int log_n, size_t ncols, int kind, uint64_t shift) {
  if (!c) return vx_fail(VX_E_INVALID, "vx_ntt_batch: ctx is NULL");
  if (kind < 0 || kind > 3) return vx_fail(VX_E_INVALID, "vx_ntt_batch: kind %d", kind);
  if (log_n < 0 || log_n > ROOT_TABLE_LOG) return vx_fail(VX_E_INVALID, "vx_ntt_batch: log_n %d", log_n);
  if (ncols == 0) return VX_OK;
  if (!data) return vx_fail(VX_E_INVALID, "vx_ntt_batch: data is NULL");
  if ((kind == VX_NTT_COSET_FFT || kind == VX_NTT_COSET_IFFT) && vxh::canon(shift) == 0)
    return vx_fail(VX_E_INVALID, "vx_ntt_batch: coset shift must be non-zero");
  HIPCHK(hipSetDevice(c->device));
  size_t n = (size_t)1 << log_n, bytes = n * ncols * 8;
  u64 *a = nullptr, *b = nullptr;
  HIPCHK(hipMalloc(&a, bytes));
  HIPCHK(hipMalloc(&b, bytes));
  int rc = VX_OK;
  do {
    if (hipMemcpyAsync(a, data, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = vx_fail(VX_E_HIP, "upload failed"); break; }
    if (log_n == 0) {
      // length-1 transform is the identity up to canonicalisation
      hipLaunchKernelGGL(canon_kernel, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, c->stream, a, ncols);
      if (hipMemcpyAsync(data, a, bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { rc = vx_fail(VX_E_HIP, "download failed"); break; }
      break;
    }
    rc = ntt_natural_to_bitrev(c, a, a, log_n, ncols, kind, shift);
    if (rc) break;
    hipLaunchKernelGGL(bitrev_permute_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)ncols), dim3(256), 0, c->stream,
                       a, b, log_n, n, n);
    if (hipMemcpyAsync(data, b, bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { rc = vx_fail(VX_E_HIP, "download failed"); break; }
  } while (0);
  hipError_t e = hipStreamSynchronize(c->stream);
  hipFree(a);
  hipFree(b);
  if (rc == VX_OK && e != hipSuccess) rc = vx_fail(VX_E_HIP, "vx_ntt_batch: %s", hipGetErrorString(e));
  return rc;
}

int vx_poseidon_permute(vx_ctx* c, uint64_t* states, size_t count) {
  if (!c || !states) return vx_fail(VX_E_INVALID, "vx_poseidon_permute: NULL argument");
  if (!count) return VX_OK;
  HIPCHK(hipSetDevice(c->device));
  u64* d = nullptr;
  HIPCHK(hipMalloc(&d, count * 96));
  hipError_t e = hipMemcpyAsync(d, states, count * 96, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    ProfScope ps(c, "poseidon_permute");
    hipLaunchKernelGGL(poseidon_permute_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, d, count);
  }
  if (e == hipSuccess) e = hipMemcpyAsync(states, d, count * 96, hipMemcpyDeviceToHost, c->stream);
  hipError_t e2 = hipStreamSynchronize(c->stream);
  hipFree(d);
  if (e != hipSuccess || e2 != hipSuccess)
    return vx_fail(VX_E_HIP, "vx_poseidon_permute: %s", hipGetErrorString(e != hipSuccess ? e : e2));
  return VX_OK;
}

int vx_merkle_cap(vx_ctx* c, const uint64_t* leaves, size_t n_leaves, size_t width, int cap_height,
                  uint64_t* digests_out, uint64_t* cap_out) {
  if (!c || !leaves || !cap_out) return vx_fail(VX_E_INVALID, "vx_merkle_cap: NULL argument");
  if (n_leaves == 0 || (n_leaves & (n_leaves - 1))) return vx_fail(VX_E_INVALID, "vx_merkle_cap: n_leaves must be a power of two");
  if (cap_height < 0 || ((size_t)1 << cap_height) > n_leaves) return vx_fail(VX_E_INVALID, "vx_merkle_cap: cap_height %d too large", cap_height);
  if (width == 0) return vx_fail(VX_E_INVALID, "vx_merkle_cap: width 0");
  HIPCHK(hipSetDevice(c->device));
  u64 *dl = nullptr, *tree = nullptr;
  size_t nd = merkle_tree_digest_count(n_leaves, cap_height);
  HIPCHK(hipMalloc(&dl, n_leaves * width * 8));
  if (hipMalloc(&tree, nd * 32) != hipSuccess) { hipFree(dl); return vx_fail(VX_E_NOMEM, "vx_merkle_cap: out of device memory"); }
  int rc = VX_OK;
  size_t cap_off = 0;
  do {
    if (hipMemcpyAsync(dl, leaves, n_leaves * width * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = vx_fail(VX_E_HIP, "upload failed"); break; }
    {
      ProfScope ps(c, "hash_leaves_rowmajor");
      hipLaunchKernelGGL(hash_leaves_rowmajor_kernel, dim3((unsigned)((n_leaves + HASH_THREADS - 1) / HASH_THREADS)),
                         dim3(HASH_THREADS), 0, c->stream, dl, n_leaves, (int)width, tree);
    }
    rc = build_merkle_levels(c, tree, n_leaves, cap_height, &cap_off);
    if (rc) break;
    if (digests_out && hipMemcpyAsync(digests_out, tree, n_leaves * 32, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { rc = vx_fail(VX_E_HIP, "download failed"); break; }
    if (hipMemcpyAsync(cap_out, tree + cap_off * 4, ((size_t)32) << cap_height, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { rc = vx_fail(VX_E_HIP, "download failed"); break; }
  } while (0);
  hipError_t e = hipStreamSynchronize(c->stream);
  hipFree(dl);
  hipFree(tree);
  if (rc == VX_OK && e != hipSuccess) rc = vx_fail(VX_E_HIP, "vx_merkle_cap: %s", hipGetErrorString(e));
  return rc;
}

__global__ void field_op_kernel(int op, const u64* __restrict__ a, const u64* __restrict__ b, u64* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 xr = a[i], yr = b[i];
  const u64 x = gl_canon(xr), y = gl_canon(yr);
  u64 r = 0;
  switch (op) {
    case 0: r = gl_mul(xr, yr); break;          // multiply accepts any representatives
    case 1: r = gl_add(x, y); break;            // add / sub are defined on canonical inputs
    case 2: r = gl_sub(x, y); break;
    case 3: r = gl_mad(xr, yr, xr); break;
    case 4: r = x ? gl_inv(x) : 0; break;
    default: r = gl_canon(gl_mul_nc(xr, yr)); break;
  }
  out[i] = r;
}

int vx_field_op(vx_ctx* c, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  if (!c || !a || !b || !out) return vx_fail(VX_E_INVALID, "vx_field_op: NULL argument");
  if (op < 0 || op > 5) return vx_fail(VX_E_INVALID, "vx_field_op: op %d", op);
  if (!n) return VX_OK;
  HIPCHK(hipSetDevice(c->device));
  u64 *da = nullptr, *db = nullptr, *dout = nullptr;
  HIPCHK(hipMalloc(&da, n * 8));
  HIPCHK(hipMalloc(&db, n * 8));
  HIPCHK(hipMalloc(&dout, n * 8));
  hipError_t e = hipMemcpyAsync(da, a, n * 8, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(db, b, n * 8, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(field_op_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, op, da, db, dout, n);
    e = hipMemcpyAsync(out, dout, n * 8, hipMemcpyDeviceToHost, c->stream);
  }
  hipError_t e2 = hipStreamSynchronize(c->stream);
  hipFree(da);
  hipFree(db);
  hipFree(dout);
  if (e != hipSuccess || e2 != hipSuccess) return vx_fail(VX_E_HIP, "vx_field_op: %s", hipGetErrorString(e != hipSuccess ? e : e2));
  return VX_OK;
}

// ---------------------------------------------------------------------------------------------
// L2
// ---------------------------------------------------------------------------------------------
int vx_batch_commit(vx_ctx* c, const uint64_t* cols, int src_on_device, int log_n, size_t ncols, int rate_bits,
                    int cap_height, int is_coeffs, vx_batch** out) {
  if (!c || !cols || !out) return vx_fail(VX_E_INVALID, "vx_batch_commit: NULL argument");
  *out = nullptr;
  if (log_n < 1 || log_n + rate_bits > ROOT_TABLE_LOG) return vx_fail(VX_E_INVALID, "vx_batch_commit: log_n=%d rate_bits=%d unsupported (need 1 <= log_n, log_n+rate_bits <= 24)", log_n, rate_bits);
  if (rate_bits < 0 || rate_bits > 4) return vx_fail(VX_E_INVALID, "vx_batch_commit: rate_bits %d", rate_bits);
  if (ncols == 0 || ncols > 4096) return vx_fail(VX_E_INVALID, "vx_batch_commit: ncols %zu", ncols);
  if (cap_height < 0 || cap_height > log_n + rate_bits) return vx_fail(VX_E_INVALID, "vx_batch_commit: cap_height %d", cap_height);
  HIPCHK(hipSetDevice(c->device));
  vx_batch* b = nullptr;
  int rc = batch_alloc(c, log_n, ncols, rate_bits, cap_height, &b);
  if (rc) return rc;
  size_t n = (size_t)1 << log_n;
  u64* staging = nullptr;
  if (src_on_device) {
    rc = batch_commit_device(c, b, cols, n, is_coeffs != 0);
  } else {
    if (hipMalloc(&staging, n * ncols * 8) != hipSuccess) { vx_batch_free(b); return vx_fail(VX_E_NOMEM, "vx_batch_commit: staging alloc failed"); }
    if (!getenv("VX_NO_UPLOAD_OVERLAP")) {
      rc = batch_commit_host(c, b, cols, staging, is_coeffs != 0);  // the upload is hidden behind the transforms
    } else if (hipMemcpyAsync(staging, cols, n * ncols * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
      rc = vx_fail(VX_E_HIP, "vx_batch_commit: upload failed");
    } else {
      rc = batch_commit_device(c, b, staging, n, is_coeffs != 0);
    }
  }
  hipError_t e = hipStreamSynchronize(c->stream);
  if (staging) hipFree(staging);
  if (rc == VX_OK && e != hipSuccess) rc = vx_fail(VX_E_HIP, "vx_batch_commit: %s", hipGetErrorString(e));
  if (rc) { vx_batch_free(b); return rc; }
  *out = b;
  return VX_OK;
}

void vx_batch_free(vx_batch* b) {
  if (!b) return;
  hipSetDevice(b->ctx->device);
  b->ctx->pool_free(b->coeffs);
  b->ctx->pool_free(b->lde);
  b->ctx->pool_free(b->tree);
  delete b;
}

int vx_batch_cap(vx_batch* b, uint64_t* cap_out) {
  if (!b || !cap_out) return vx_fail(VX_E_INVALID, "vx_batch_cap: NULL argument");
  return vx_dev_download(b->ctx, cap_out, b->tree + b->cap_off * 4, ((size_t)32) << b->cap_height);
}

int vx_batch_coeffs(vx_batch* b, size_t col, uint64_t* out) {
  if (!b || !out) return vx_fail(VX_E_INVALID, "vx_batch_coeffs: NULL argument");
  if (col >= b->ncols) return vx_fail(VX_E_INVALID, "vx_batch_coeffs: column %zu out of range", col);
  vx_ctx* c = b->ctx;
  HIPCHK(hipSetDevice(c->device));
  size_t n = (size_t)1 << b->log_n;
  u64* tmp = nullptr;
  HIPCHK(hipMalloc(&tmp, n * 8));
  hipLaunchKernelGGL(bitrev_permute_kernel, dim3((unsigned)((n + 255) / 256), 1), dim3(256), 0, c->stream,
                     b->coeffs + col * n, tmp, b->log_n, n, n);
  int rc = vx_dev_download(c, out, tmp, n * 8);
  hipFree(tmp);
  return rc;
}

int vx_batch_open_row(vx_batch* b, size_t row, uint64_t* values_out, uint64_t* path_out) {
  if (!b || !values_out || !path_out) return vx_fail(VX_E_INVALID, "vx_batch_open_row: NULL argument");
  size_t N = (size_t)1 << (b->log_n + b->rate_bits);
  if (row >= N) return vx_fail(VX_E_INVALID, "vx_batch_open_row: row %zu out of range", row);
  vx_ctx* c = b->ctx;
  HIPCHK(hipSetDevice(c->device));
  // strided gather of one row: a 2D copy (pitch = column stride)
  HIPCHK(hipMemcpy2DAsync(values_out, 8, b->lde + row, N * 8, 8, b->ncols, hipMemcpyDeviceToHost, c->stream));
  size_t off = 0, n = N, idx = row;
  int k = 0;
  while (n > ((size_t)1 << b->cap_height)) {
    HIPCHK(hipMemcpyAsync(path_out + 4 * k, b->tree + (off + (idx ^ 1)) * 4, 32, hipMemcpyDeviceToHost, c->stream));
    off += n;
    n >>= 1;
    idx >>= 1;
    ++k;
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return VX_OK;
}

int vx_batch_digests(vx_batch* b, uint64_t* out) {
  if (!b || !out) return vx_fail(VX_E_INVALID, "vx_batch_digests: NULL argument");
  size_t N = (size_t)1 << (b->log_n + b->rate_bits);
  return vx_dev_download(b->ctx, out, b->tree, N * 32);
}

int vx_batch_lde_rows(vx_batch* b, size_t row0, size_t nrows, uint64_t* out) {
  if (!b || !out) return vx_fail(VX_E_INVALID, "vx_batch_lde_rows: NULL argument");
  size_t N = (size_t)1 << (b->log_n + b->rate_bits);
  if (row0 + nrows > N) return vx_fail(VX_E_INVALID, "vx_batch_lde_rows: range out of bounds");
  vx_ctx* c = b->ctx;
  HIPCHK(hipSetDevice(c->device));
  // out[r*ncols + col] = lde[col*N + row0 + r]: one 2D copy per column (dst pitch = ncols*8)
  for (size_t col = 0; col < b->ncols; ++col)
    HIPCHK(hipMemcpy2DAsync(out + col, b->ncols * 8, b->lde + col * N + row0, 8, 8, nrows, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return VX_OK;
}

int vx_batch_eval_ext(vx_batch* b, const uint64_t zeta[2], uint64_t* out) {
  if (!b || !zeta || !out) return vx_fail(VX_E_INVALID, "vx_batch_eval_ext: NULL argument");
  vx_ctx* c = b->ctx;
  HIPCHK(hipSetDevice(c->device));
  u64* ztab = nullptr;
  size_t n = (size_t)1 << b->log_n;
  HIPCHK(hipMalloc(&ztab, n * 16));
  int rc = build_zeta_table(c, vxh::Ext{vxh::canon(zeta[0]), vxh::canon(zeta[1])}, b->log_n, ztab);
  if (rc == VX_OK) rc = batch_eval_ext(c, b->coeffs, n, b->log_n, b->ncols, ztab, out);
  hipStreamSynchronize(c->stream);
  hipFree(ztab);
  return rc;
}

int vx_lde_columns_dev(vx_ctx* c, const uint64_t* values, int log_n, size_t ncols, int rate_bits, uint64_t* lde_out,
                       uint64_t* coeffs_out) {
  if (!c || !values || !lde_out) return vx_fail(VX_E_INVALID, "vx_lde_columns_dev: NULL argument");
  if (log_n < 1 || rate_bits < 0 || rate_bits > 4 || log_n + rate_bits > ROOT_TABLE_LOG)
    return vx_fail(VX_E_INVALID, "vx_lde_columns_dev: log_n=%d rate_bits=%d unsupported", log_n, rate_bits);
  if (ncols == 0) return VX_OK;
  HIPCHK(hipSetDevice(c->device));
  using namespace vxh;
  const size_t n = (size_t)1 << log_n, N = n << rate_bits;
  u64* coeffs = coeffs_out;
  if (!coeffs && c->pool_alloc((void**)&coeffs, n * ncols * 8) != hipSuccess) return vx_fail(VX_E_NOMEM, "vx_lde_columns_dev: out of device memory");
  int rc = run_ntt(c, values, coeffs, n, n, 0, 0, log_n, ncols, 1, true, false, nullptr, 0, inv((u64)n % P), "intt", 16.0 * n * ncols);
  if (rc == VX_OK) {
    const int nz = 1 << rate_bits;
    std::vector<u64> shifts(nz);
    u64 wN = root_of_unity(log_n + rate_bits);
    for (int z = 0; z < nz; ++z) shifts[z] = mul(7, pow(wN, reverse_bits((size_t)z, rate_bits)));
    u64* tab = nullptr;
    rc = get_scale_tables(c, log_n, log_n / 2, shifts, 1, &tab);
    if (rc == VX_OK)
      rc = run_ntt(c, coeffs, lde_out, n, N, 0, n, log_n, ncols, nz, false, true, tab, log_n / 2, 1, "lde", (double)ncols * 8.0 * ((double)n + N));
  }
  hipError_t e = hipStreamSynchronize(c->stream);
  if (!coeffs_out) c->pool_free(coeffs);
  if (rc == VX_OK && e != hipSuccess) rc = vx_fail(VX_E_HIP, "vx_lde_columns_dev: %s", hipGetErrorString(e));
  return rc;
}

size_t vx_merkle_digest_count(size_t n_leaves, int cap_height) { return merkle_tree_digest_count(n_leaves, cap_height); }

int vx_hash_rows_dev(vx_ctx* c, const uint64_t* cols, size_t col_stride, size_t nrows, size_t ncols, int cap_height,
                     uint64_t* tree_out, uint64_t* cap_out) {
  if (!c || !cols || !cap_out) return vx_fail(VX_E_INVALID, "vx_hash_rows_dev: NULL argument");
  if (nrows == 0 || (nrows & (nrows - 1)) || ((size_t)1 << cap_height) > nrows || cap_height < 0)
    return vx_fail(VX_E_INVALID, "vx_hash_rows_dev: nrows must be a power of two >= 2^cap_height");
  if (ncols == 0 || ncols > 4096) return vx_fail(VX_E_INVALID, "vx_hash_rows_dev: ncols %zu", ncols);
  HIPCHK(hipSetDevice(c->device));
  u64* tree = tree_out;
  const size_t nd = merkle_tree_digest_count(nrows, cap_height);
  if (!tree && c->pool_alloc((void**)&tree, nd * 32) != hipSuccess) return vx_fail(VX_E_NOMEM, "vx_hash_rows_dev: out of device memory");
  {
    ProfScope ps(c, "hash_leaves", (double)ncols * 8.0 * (double)nrows);
    hipLaunchKernelGGL(hash_leaves_colmajor_kernel, dim3((unsigned)((nrows + HASH_THREADS - 1) / HASH_THREADS)), dim3(HASH_THREADS), 0,
                       c->stream, cols, col_stride, nrows, (int)ncols, tree, (u64*)nullptr);
  }
  size_t cap_off = 0;
  int rc = build_merkle_levels(c, tree, nrows, cap_height, &cap_off);
  hipError_t e = hipSuccess;
  if (rc == VX_OK) e = hipMemcpyAsync(cap_out, tree + cap_off * 4, (size_t)32 << cap_height, hipMemcpyDeviceToHost, c->stream);
  hipError_t e2 = hipStreamSynchronize(c->stream);
  if (!tree_out) c->pool_free(tree);
  if (rc == VX_OK && (e != hipSuccess || e2 != hipSuccess)) rc = vx_fail(VX_E_HIP, "vx_hash_rows_dev: %s", hipGetErrorString(e != hipSuccess ? e : e2));
  return rc;
}

// ---------------------------------------------------------------------------------------------
// L3
// ---------------------------------------------------------------------------------------------
// One rehearsal proof of an all-zero witness (discarded; the two "witness does not satisfy the circuit" checks are off): every device
// buffer shape a proof of this circuit asks the pool for is allocated once and cached, the circuit's program-gate kernels are loaded,
// the twiddle / scale tables of its sizes exist — so that the FIRST real proof costs what every later one does (round 3 measured 250 ms
// instead of 57 for the first 2^19-row proof on some boxes: first-use hipMallocs).  Circuits with lookup tables are rehearsed too since
// round 6 (the lookup polynomials are computed on the device and a zero denominator contributes zero; rounds 3-5 skipped them, which
// left the first pass of a DAG of recursion-shaped circuits 1.2 s slower than the later ones).
int vx_circuit_warm(vx_ctx* c, vx_circuit* k) {
  if (!c || !k) return vx_fail(VX_E_INVALID, "vx_circuit_warm: NULL argument");
  if (k->ctx != c) return vx_fail(VX_E_INVALID, "vx_circuit_warm: circuit belongs to a different context");
  HIPCHK(hipSetDevice(c->device));
  // Several processes (the DAG's worker pool, ranks emulated on one device) and several contexts of one process load circuits on the same
  // device at the same time; each would see the same free memory below and then all allocate.  One rehearsal at a time per device: an
  // advisory file lock for the other processes, a mutex for this one's threads (both best effort: a rehearsal is an optimisation).
  // Both waits are BOUNDED (VX_WARM_LOCK_TIMEOUT_S, default 120 s): a worker that hangs inside its rehearsal must not block every other
  // process's vx_circuit_create on that device for ever — after the wait the rehearsal is skipped, which costs the first proof some
  // hipMallocs and nothing else.  The mutex is per device; the lock file lives in a directory of this user (mode 0700, no symlink followed).
  const char* te = getenv("VX_WARM_LOCK_TIMEOUT_S");
  const double wait_s = te && atof(te) >= 0 ? atof(te) : 120.0;
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(wait_s));
  static std::timed_mutex warm_mutex[16];
  std::unique_lock<std::timed_mutex> in_process(warm_mutex[(unsigned)c->device % 16], std::defer_lock);
  if (!in_process.try_lock_until(deadline)) return VX_OK;
  struct FileLock {
    int fd = -1;
    bool timed_out = false;
    FileLock(int device, std::chrono::steady_clock::time_point until) {
      char dir[256], path[320];
      const char* rt = getenv("XDG_RUNTIME_DIR");
      if (rt && *rt) snprintf(dir, sizeof dir, "%s/vxprover", rt);
      else snprintf(dir, sizeof dir, "/tmp/vxprover-%u", (unsigned)geteuid());
      struct stat st;
      if (mkdir(dir, 0700) != 0 && errno != EEXIST) return;                      // no lock directory: rehearse unlocked (best effort)
      if (lstat(dir, &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != geteuid() || (st.st_mode & 0077)) return;
      snprintf(path, sizeof path, "%s/warm_dev%d.lock", dir, device);
      fd = open(path, O_CREAT | O_RDWR | O_NOFOLLOW | O_CLOEXEC, 0600);
      if (fd < 0) return;
      while (flock(fd, LOCK_EX | LOCK_NB) != 0) {
        if (errno != EWOULDBLOCK && errno != EINTR) { close(fd); fd = -1; return; }
        if (std::chrono::steady_clock::now() >= until) { close(fd); fd = -1; timed_out = true; return; }
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
      }
    }
    ~FileLock() {
      if (fd >= 0) {
        flock(fd, LOCK_UN);
        close(fd);
      }
    }
  } across_processes(c->device, deadline);
  if (across_processes.timed_out) return VX_OK;
  {
    // a proof's working set is about 8 N (wires + Z / partial products + quotient chunks) for the LDEs plus trees, coefficients and
    // scratch: rehearse only when twice a generous estimate is free — a host that packs many contexts onto one device (the
    // single-device emulation of a sharded proof) must not have every one of them cache a full-size working set
    size_t free_b = 0, total_b = 0;
    const size_t N = k->n() << k->rate_bits;
    const size_t need = (size_t)16 * N * ((size_t)k->num_wires + 40);
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < 2 * need) {
      (void)hipGetLastError();
      return VX_OK;
    }
  }
  const size_t bytes = (size_t)k->num_wires * k->n() * 8;
  void* zero = nullptr;
  if (c->pool_alloc(&zero, bytes) != hipSuccess) return vx_fail(VX_E_NOMEM, "vx_circuit_warm: out of device memory");
  int rc = VX_OK;
  const bool prof_was = c->prof_on;
  c->prof_on = false;
  c->rehearsal = true;
  try {
    if (hipMemsetAsync(zero, 0, bytes, c->stream) != hipSuccess) rc = vx_fail(VX_E_HIP, "vx_circuit_warm: hipMemsetAsync failed");
    std::vector<uint8_t> discard;
    if (rc == VX_OK) rc = prove_impl(c, k, (const u64*)zero, /*wires_on_device=*/true, nullptr, discard);
  } catch (const std::exception& e) {
    rc = vx_fail(VX_E_INVALID, "vx_circuit_warm: %s", e.what());
  }
  c->rehearsal = false;
  c->prof_on = prof_was;
  hipStreamSynchronize(c->stream);
  c->pool_free(zero);
  if (rc != VX_OK) c->pool_trim();   // a rehearsal that ran out of memory must not sit on what it did get
  return rc;
}
// circuit load: the rehearsal is best effort — a failed one leaves a perfectly usable circuit AND no stale error string behind
static void warm_on_load(vx_ctx* c, vx_circuit* k) {
  if (getenv("VX_NO_WARM_ON_LOAD")) return;
  if (vx_circuit_warm(c, k) != VX_OK) {
    (void)hipGetLastError();
    g_err[0] = 0;
  }
}
int vx_circuit_create(vx_ctx* c, const vx_circuit_desc* desc, vx_circuit** out) {
  if (!c || !desc || !out) return vx_fail(VX_E_INVALID, "vx_circuit_create: NULL argument");
  *out = nullptr;
  HIPCHK(hipSetDevice(c->device));
  try {  // nothing unwinds across the ABI
    const int rc = circuit_create(c, desc, out);
    if (rc == VX_OK) warm_on_load(c, *out);   // on by default (VX_NO_WARM_ON_LOAD=1 turns it off)
    return rc;
  } catch (const std::bad_alloc&) {
    return vx_fail(VX_E_NOMEM, "vx_circuit_create: out of host memory");
  } catch (const std::exception& e) {
    return vx_fail(VX_E_INVALID, "vx_circuit_create: %s", e.what());
  }
}
// ---- .vxcircuit: the compiled-circuit container (circuit_io.h); host code, no device needed except vx_circuit_load ----
size_t vx_circuit_serialized_size(const vx_circuit_desc* d, int with_cap, int with_preprocessed) {
  if (!d) return 0;
  const std::string why = desc_check(d, with_preprocessed != 0, nullptr);
  if (!why.empty()) {
    vx_fail(VX_E_INVALID, "vx_circuit_serialized_size: %s", why.c_str());
    return 0;
  }
  return vxio::sizes_of(d, with_cap != 0, with_preprocessed != 0).total;
}
int vx_circuit_serialize(const vx_circuit_desc* d, const uint64_t* constants_sigmas_cap, int with_preprocessed, uint8_t* out, size_t* len) {
  if (!d || !out || !len) return vx_fail(VX_E_INVALID, "vx_circuit_serialize: NULL argument");
  try {
    const std::string why = desc_check(d, with_preprocessed != 0, nullptr);
    if (!why.empty()) return vx_fail(VX_E_INVALID, "vx_circuit_serialize: %s", why.c_str());
    const size_t need = vxio::sizes_of(d, constants_sigmas_cap != nullptr, with_preprocessed != 0).total;
    if (*len < need) {
      *len = need;
      return vx_fail(VX_E_INVALID, "vx_circuit_serialize: output buffer too small, need %zu bytes", need);
    }
    vxio::serialize(d, constants_sigmas_cap, with_preprocessed != 0, out);
    *len = need;
    return VX_OK;
  } catch (const std::exception& e) {
    return vx_fail(VX_E_NOMEM, "vx_circuit_serialize: %s", e.what());
  }
}
int vx_circuit_parse(const uint8_t* bytes, size_t len, const vx_circuit_desc** desc_out, const uint64_t** cap_out) {
  if (!bytes || !desc_out) return vx_fail(VX_E_INVALID, "vx_circuit_parse: NULL argument");
  *desc_out = nullptr;
  if (cap_out) *cap_out = nullptr;
  try {
    std::unique_ptr<vxio::Parsed> P(new vxio::Parsed());
    std::string why = vxio::parse(bytes, len, P.get());
    if (why.empty()) why = desc_check(&P->desc, P->desc.constants_sigmas != nullptr, nullptr);
    if (!why.empty()) return vx_fail(VX_E_INVALID, "vx_circuit_parse: %s", why.c_str());
    if (cap_out) *cap_out = P->cap;
    *desc_out = &P.release()->desc;  // desc is the first member: vx_circuit_desc_free casts back
    return VX_OK;
  } catch (const std::exception& e) {
    return vx_fail(VX_E_NOMEM, "vx_circuit_parse: %s", e.what());
  }
}
void vx_circuit_desc_free(const vx_circuit_desc* d) {
  static_assert(offsetof(vxio::Parsed, desc) == 0, "desc must be the first member");
  delete reinterpret_cast<vxio::Parsed*>(const_cast<vx_circuit_desc*>(d));
}
int vx_circuit_load(vx_ctx* c, const uint8_t* bytes, size_t len, vx_circuit** out) {
  if (!c || !bytes || !out) return vx_fail(VX_E_INVALID, "vx_circuit_load: NULL argument");
  *out = nullptr;
  const vx_circuit_desc* d = nullptr;
  int rc = vx_circuit_parse(bytes, len, &d, nullptr);
  if (rc) return rc;
  if (!d->constants_sigmas) {
    vx_circuit_desc_free(d);
    return vx_fail(VX_E_INVALID, "vx_circuit_load: the file holds verifier data only (no preprocessed polynomial values)");
  }
  rc = vx_circuit_create(c, d, out);
  vx_circuit_desc_free(d);
  return rc;
}
void vx_circuit_free(vx_circuit* k) { circuit_free(k); }
int vx_circuit_digest(vx_circuit* k, uint64_t digest_out[4]) {
  if (!k || !digest_out) return vx_fail(VX_E_INVALID, "vx_circuit_digest: NULL argument");
  memcpy(digest_out, k->digest.e, 32);
  return VX_OK;
}
int vx_circuit_constants_sigmas_cap(vx_circuit* k, uint64_t* cap_out) {
  if (!k || !cap_out) return vx_fail(VX_E_INVALID, "vx_circuit_constants_sigmas_cap: NULL argument");
  return vx_batch_cap(k->cs, cap_out);
}
size_t vx_proof_size_bound(vx_circuit* k) { return k ? proof_size_bound(k) : 0; }
int vx_circuit_program_gates(vx_circuit* k, int* total_out, int* compiled_out, char* note_out, size_t note_cap) {
  if (!k) return vx_fail(VX_E_INVALID, "vx_circuit_program_gates: NULL argument");
  int total = 0, compiled = 0;
  for (size_t g = 0; g < k->prog_off.size(); ++g)
    if (k->prog_off[g] >= 0) {
      ++total;
      compiled += !k->jit_fns.empty() || k->jit_fused_fn != nullptr;
    }
  if (total_out) *total_out = total;
  if (compiled_out) *compiled_out = compiled;
  if (note_out && note_cap) {
    strncpy(note_out, k->jit_note.c_str(), note_cap - 1);
    note_out[note_cap - 1] = 0;
  }
  return VX_OK;
}

int vx_prove(vx_ctx* c, vx_circuit* k, const uint64_t* wires, int wires_on_device, const uint64_t* pow_witness_hint,
             uint8_t* out_buf, size_t* out_len) {
  if (!c || !k || !wires || !out_buf || !out_len) return vx_fail(VX_E_INVALID, "vx_prove: NULL argument");
  if (k->ctx != c) return vx_fail(VX_E_INVALID, "vx_prove: circuit belongs to a different context");
  HIPCHK(hipSetDevice(c->device));
  std::vector<uint8_t> proof;
  int rc;
  try {  // nothing unwinds across the ABI
    rc = prove_impl(c, k, wires, wires_on_device != 0, pow_witness_hint, proof);
  } catch (const std::bad_alloc&) {
    rc = vx_fail(VX_E_NOMEM, "vx_prove: out of host memory");
  } catch (const std::exception& e) {
    rc = vx_fail(VX_E_INVALID, "vx_prove: %s", e.what());
  }
  if (rc != VX_OK) {
    hipStreamSynchronize(c->stream);
    return rc;
  }
  if (proof.size() > *out_len) {
    *out_len = proof.size();
    return vx_fail(VX_E_INVALID, "vx_prove: output buffer too small, need %zu bytes", proof.size());
  }
  memcpy(out_buf, proof.data(), proof.size());
  *out_len = proof.size();
  return VX_OK;
}

int vx_verify(vx_circuit* k, const uint8_t* proof, size_t proof_len) {
  if (!k || !proof) return vx_fail(VX_E_INVALID, "vx_verify: NULL argument");
  vxv::CircuitV v;
  v.degree_bits = k->degree_bits;
  v.num_wires = k->num_wires;
  v.num_routed = k->nr;
  v.num_challenges = k->nch;
  v.rate_bits = k->rate_bits;
  v.cap_height = k->cap_height;
  v.pow_bits = k->pow_bits;
  v.num_queries = k->num_queries;
  v.qdf = k->qdf;
  v.num_selectors = k->num_selectors;
  v.num_constants = k->num_constants;
  v.num_public_inputs = (int)k->pi_rows.size();
  for (size_t g = 0; g < k->gates.size(); ++g)
    v.gates.push_back(vxv::GateV{k->gates[g].type, k->gates[g].param, k->gates[g].selector_index, k->gates[g].group_start,
                                 k->gates[g].group_end, k->prog_off[g] >= 0 ? k->programs_host.data() + k->prog_off[g] : nullptr});
  v.arity_bits = k->arity_bits;
  v.num_luts = k->num_luts, v.num_lookup_selectors = k->num_lookup_selectors;
  v.lut_lens = k->lut_lens.data(), v.lut_inputs = k->lut_inputs.data(), v.lut_outputs = k->lut_outputs.data();
  v.k_is = k->k_is_host.data();
  v.cs_cap = k->cs_cap_host.data();
  v.digest = k->digest;
  std::string why;
  try {
    why = vxv::verify(v, proof, proof_len);
  } catch (const std::exception& e) {  // never unwind across the ABI
    why = std::string("exception: ") + e.what();
  }
  if (!why.empty()) return vx_fail(VX_E_PROOF, "vx_verify: %s", why.c_str());
  return VX_OK;
}

int vx_verify_standalone(const vx_circuit_desc* d, const uint64_t* cs_cap, const uint8_t* proof, size_t proof_len) {
  if (!d || !cs_cap || !proof) return vx_fail(VX_E_INVALID, "vx_verify_standalone: NULL argument");
  try {
    DescResolved res;
    {  // the same validation vx_circuit_create runs (desc_check.h); a verifier holds no preprocessed polynomials
      const std::string why = desc_check(d, /*need_preprocessed=*/false, &res);
      if (!why.empty()) return vx_fail(VX_E_INVALID, "vx_verify_standalone: %s", why.c_str());
    }
    vxv::CircuitV v;
    v.degree_bits = d->degree_bits;
    v.num_wires = d->num_wires;
    v.num_routed = d->num_routed_wires;
    v.num_challenges = d->num_challenges;
    v.rate_bits = d->rate_bits;
    v.cap_height = d->cap_height;
    v.pow_bits = d->pow_bits;
    v.num_queries = d->num_query_rounds;
    v.qdf = d->quotient_degree_factor;
    v.num_selectors = d->num_selectors;
    v.num_constants = d->num_constants;
    v.num_public_inputs = d->num_public_inputs;
    for (int g = 0; g < d->num_gates; ++g)
      v.gates.push_back(vxv::GateV{d->gate_types[g], d->gate_params[g], d->selector_indices[g], d->group_starts[g], d->group_ends[g],
                                   d->gate_types[g] == VX_GATE_PROGRAM ? d->programs + d->program_offsets[g] : nullptr});
    v.arity_bits = res.arity_bits;
    if (d->num_luts > 0) {
      v.num_luts = d->num_luts, v.num_lookup_selectors = d->num_lookup_selectors;
      v.lut_lens = d->lut_lens, v.lut_inputs = d->lut_inputs, v.lut_outputs = d->lut_outputs;
    }
    std::vector<u64> k_is(d->k_is, d->k_is + d->num_routed_wires);
    for (auto& x : k_is) x = vxh::canon(x);
    v.k_is = k_is.data();
    const size_t cap_words = (size_t)4 << d->cap_height;
    std::vector<u64> pre(cap_words + 1);
    for (size_t i = 0; i < cap_words; ++i) pre[i] = vxh::canon(cs_cap[i]);
    pre.back() = (u64)d->degree_bits;
    v.cs_cap = pre.data();
    v.digest = circuit_digest_of(d, pre.data(), pre.size());
    const std::string why = vxv::verify(v, proof, proof_len);
    if (!why.empty()) return vx_fail(VX_E_PROOF, "vx_verify: %s", why.c_str());
    return VX_OK;
  } catch (const std::bad_alloc&) {
    return vx_fail(VX_E_NOMEM, "vx_verify_standalone: out of host memory");
  } catch (const std::exception& e) {  // never unwind across the ABI
    return vx_fail(VX_E_PROOF, "vx_verify: exception: %s", e.what());
  }
}

// ---- STARK spike (stark.hip.h) ----
int vx_stark_prove(vx_ctx* c, const vx_stark_desc* d, const uint64_t* trace, int trace_on_device, const uint64_t* public_inputs,
                   const uint64_t* pow_witness_hint, uint8_t* out_buf, size_t* out_len) {
  if (!c || !d || !trace || !out_buf || !out_len || (d->num_public_inputs > 0 && !public_inputs)) return vx_fail(VX_E_INVALID, "vx_stark_prove: NULL argument");
  HIPCHK(hipSetDevice(c->device));
  std::vector<uint8_t> proof;
  int rc;
  try {
    rc = stark_prove_impl(c, d, trace, trace_on_device != 0, public_inputs, pow_witness_hint, proof);
  } catch (const std::bad_alloc&) {
    rc = vx_fail(VX_E_NOMEM, "vx_stark_prove: out of host memory");
  } catch (const std::exception& e) {
    rc = vx_fail(VX_E_INVALID, "vx_stark_prove: %s", e.what());
  }
  if (rc != VX_OK) {
    hipStreamSynchronize(c->stream);
    return rc;
  }
  if (proof.size() > *out_len) {
    *out_len = proof.size();
    return vx_fail(VX_E_INVALID, "vx_stark_prove: output buffer too small, need %zu bytes", proof.size());
  }
  memcpy(out_buf, proof.data(), proof.size());
  *out_len = proof.size();
  return VX_OK;
}
int vx_stark_begin(vx_ctx* c, const vx_stark_desc* d, const uint64_t* trace, int trace_on_device, const uint64_t* public_inputs,
                   uint64_t* aux_challenges_out, vx_stark_session** out) {
  return vx_stark_begin_sharded(c, d, trace, trace_on_device, public_inputs, 0, 1, nullptr, nullptr, aux_challenges_out, out);
}
int vx_stark_begin_sharded(vx_ctx* c, const vx_stark_desc* d, const uint64_t* trace, int trace_on_device, const uint64_t* public_inputs,
                           int rank, int world, vx_allgather_fn allgather, void* user, uint64_t* aux_challenges_out, vx_stark_session** out) {
  if (!c || !d || !trace || !out || (d->num_public_inputs > 0 && !public_inputs)) return vx_fail(VX_E_INVALID, "vx_stark_begin: NULL argument");
  *out = nullptr;
  Shard shard;
  shard.rank = rank;
  shard.world = world;
  while ((1 << shard.lg) < world) ++shard.lg;
  if (world < 1 || (1 << shard.lg) != world || shard.lg > d->rate_bits || shard.lg > d->cap_height)
    return vx_fail(VX_E_INVALID, "vx_stark_begin_sharded: world=%d must be a power of two <= 2^rate_bits and <= 2^cap_height", world);
  if (rank < 0 || rank >= world) return vx_fail(VX_E_INVALID, "vx_stark_begin_sharded: rank %d outside [0, %d)", rank, world);
  if (world > 1 && !allgather) return vx_fail(VX_E_INVALID, "vx_stark_begin_sharded: world > 1 needs an all-gather callback");
  shard.fn = allgather;
  shard.user = user;
  HIPCHK(hipSetDevice(c->device));
  int rc;
  vx_stark_session* s = nullptr;
  try {
    s = new vx_stark_session();
    rc = stark_begin_impl(c, d, trace, trace_on_device != 0, public_inputs, *s, shard);
    if (rc == VX_OK && !s->aux_challenges.empty() && !aux_challenges_out) rc = vx_fail(VX_E_INVALID, "vx_stark_begin: NULL aux_challenges_out");
  } catch (const std::bad_alloc&) {
    rc = vx_fail(VX_E_NOMEM, "vx_stark_begin: out of host memory");
  } catch (const std::exception& e) {
    rc = vx_fail(VX_E_INVALID, "vx_stark_begin: %s", e.what());
  }
  if (rc != VX_OK) {
    hipStreamSynchronize(c->stream);
    delete s;
    return rc;
  }
  for (size_t i = 0; i < s->aux_challenges.size(); ++i) aux_challenges_out[i] = s->aux_challenges[i];
  *out = s;
  return VX_OK;
}
int vx_stark_finish(vx_stark_session* s, const uint64_t* aux_columns, int aux_on_device, const uint64_t* pow_witness_hint, uint8_t* out_buf,
                    size_t* out_len) {
  return vx_stark_finish2(s, aux_columns, aux_on_device, nullptr, pow_witness_hint, out_buf, out_len);
}
int vx_circuit_precompile(const vx_circuit_desc* d, int* num_program_gates_out) {
  if (!d) return vx_fail(VX_E_INVALID, "vx_circuit_precompile: NULL argument");
  try {
    const std::string bad = desc_check(d, false, nullptr);
    if (!bad.empty()) return vx_fail(VX_E_INVALID, "vx_circuit_precompile: %s", bad.c_str());
    std::vector<const uint64_t*> progs;
    for (int g = 0; g < d->num_gates; ++g)
      if (d->gate_types[g] == VX_GATE_PROGRAM) progs.push_back(d->programs + d->program_offsets[g]);
    if (num_program_gates_out) *num_program_gates_out = (int)progs.size();
    if (progs.empty()) return 0;
    std::string why;
    const int rc = jit_gates_precompile(progs, d->num_challenges, &why);
    if (rc < 0) return vx_fail(VX_E_INVALID, "vx_circuit_precompile: %s", why.c_str());
    return rc;
  } catch (const std::exception& e) {
    return vx_fail(VX_E_INVALID, "vx_circuit_precompile: %s", e.what());
  }
}
int vx_stark_precompile(const vx_stark_desc* d, int* num_chunks_out) {
  if (!d) return vx_fail(VX_E_INVALID, "vx_stark_precompile: NULL argument");
  try {
    StarkShape sh;
    const std::string bad = stark_check(d, &sh);
    if (!bad.empty()) return vx_fail(VX_E_INVALID, "vx_stark_precompile: %s", bad.c_str());
    std::string why;
    const int rc = jit_air_precompile(d->program, d->num_challenges, d->num_columns, num_chunks_out, &why);
    if (rc < 0) return vx_fail(VX_E_INVALID, "vx_stark_precompile: %s", why.c_str());
    return rc;
  } catch (const std::exception& e) {
    return vx_fail(VX_E_INVALID, "vx_stark_precompile: %s", e.what());
  }
}
int vx_stark_session_trace_cap(vx_stark_session* s, uint64_t* cap_out) {
  if (!s || !cap_out) return vx_fail(VX_E_INVALID, "vx_stark_session_trace_cap: NULL argument");
  memcpy(cap_out, s->trace_cap.data(), s->trace_cap.size() * 8);
  return VX_OK;
}
int vx_stark_set_aux_challenges(vx_stark_session* s, const uint64_t* shared) {
  if (!s || !shared) return vx_fail(VX_E_INVALID, "vx_stark_set_aux_challenges: NULL argument");
  if (s->finished || s->shared_challenges) return vx_fail(VX_E_INVALID, "vx_stark_set_aux_challenges: the session's challenges are already fixed");
  if (s->aux_challenges.empty()) return vx_fail(VX_E_INVALID, "vx_stark_set_aux_challenges: this AIR has no second-round challenges");
  for (size_t i = 0; i < s->aux_challenges.size(); ++i) s->aux_challenges[i] = vxh::canon(shared[i]);
  s->ch.observe_elements(s->aux_challenges.data(), s->aux_challenges.size());   // the shared challenges are part of THIS proof's transcript
  s->shared_challenges = true;
  return VX_OK;
}
int vx_stark_joint_challenges(const uint64_t* const* trace_caps, const int32_t* cap_heights, int num_tables, int num_challenges, uint64_t* out) {
  if (!trace_caps || !cap_heights || !out || num_tables < 1 || num_tables > 64 || num_challenges < 1 || num_challenges > VX_AIR_MAX_CHALLENGES)
    return vx_fail(VX_E_INVALID, "vx_stark_joint_challenges: bad argument");
  vxh::Challenger ch;
  ch.observe_element((uint64_t)num_tables);
  for (int t = 0; t < num_tables; ++t) {
    if (!trace_caps[t] || cap_heights[t] < 0 || cap_heights[t] > 24) return vx_fail(VX_E_INVALID, "vx_stark_joint_challenges: bad cap %d", t);
    std::vector<vxh::u64> cap(trace_caps[t], trace_caps[t] + ((size_t)4 << cap_heights[t]));
    for (auto& v : cap) v = vxh::canon(v);
    ch.observe_elements(cap.data(), cap.size());
  }
  for (int i = 0; i < num_challenges; ++i) out[i] = ch.get_challenge();
  return VX_OK;
}
int vx_stark_proof_trace_cap(const vx_stark_desc* d, const uint8_t* proof, size_t proof_len, uint64_t* cap_out) {
  if (!d || !proof || !cap_out || d->cap_height < 0 || d->cap_height > 24) return vx_fail(VX_E_INVALID, "vx_stark_proof_trace_cap: bad argument");
  const size_t bytes = (size_t)32 << d->cap_height;
  if (proof_len < bytes) return vx_fail(VX_E_PROOF, "vx_stark_proof_trace_cap: proof shorter than its trace cap");
  memcpy(cap_out, proof, bytes);
  return VX_OK;
}
int vx_stark_finish2(vx_stark_session* s, const uint64_t* aux_columns, int aux_on_device, const uint64_t* aux_public_inputs,
                     const uint64_t* pow_witness_hint, uint8_t* out_buf, size_t* out_len) {
  if (!s || !out_buf || !out_len) return vx_fail(VX_E_INVALID, "vx_stark_finish: NULL argument");
  if (s->finished) return vx_fail(VX_E_INVALID, "vx_stark_finish: the session has already produced its proof");
  HIPCHK(hipSetDevice(s->c->device));
  std::vector<uint8_t> proof;
  int rc;
  try {
    rc = stark_finish_impl(*s, aux_columns, aux_on_device != 0, aux_public_inputs, pow_witness_hint, proof);
  } catch (const std::bad_alloc&) {
    rc = vx_fail(VX_E_NOMEM, "vx_stark_finish: out of host memory");
  } catch (const std::exception& e) {
    rc = vx_fail(VX_E_INVALID, "vx_stark_finish: %s", e.what());
  }
  if (rc != VX_OK) {
    hipStreamSynchronize(s->c->stream);
    return rc;
  }
  if (proof.size() > *out_len) {
    *out_len = proof.size();
    return vx_fail(VX_E_INVALID, "vx_stark_finish: output buffer too small, need %zu bytes", proof.size());
  }
  memcpy(out_buf, proof.data(), proof.size());
  *out_len = proof.size();
  s->finished = true;
  return VX_OK;
}
void vx_stark_session_free(vx_stark_session* s) { delete s; }
// Compile an aux program ahead of time (no GPU): -> 1 if compiled now, 0 if it was cached, negative VX_E_*.
int vx_stark_aux_precompile(const vx_aux_desc* d) {
  if (!d || !d->program || d->program_len < 1 || d->num_fractions < 1) return vx_fail(VX_E_INVALID, "vx_stark_aux_precompile: bad argument");
  try {
    bool ended = false;
    for (int pc = 0; pc < d->program_len; ++pc) {
      const int op = (int)(d->program[pc] & 0xFF);
      if (op == VX_OP_END) {
        ended = true;
        break;
      }
      if (op == VX_OP_LDI) ++pc;
    }
    if (!ended) return vx_fail(VX_E_INVALID, "vx_stark_aux_precompile: the program does not END");
    std::string why;
    const int rc = jit_aux_precompile(d->program, d->num_fractions, &why);
    if (rc < 0) return vx_fail(VX_E_INVALID, "vx_stark_aux_precompile: %s", why.c_str());
    return rc;
  } catch (const std::exception& e) {
    return vx_fail(VX_E_INVALID, "vx_stark_aux_precompile: %s", e.what());
  }
}
// The caller's second-round columns computed on the device (aux.hip.h): fractions num / den of per-row expressions + running sums.
int vx_stark_aux_columns(vx_ctx* c, const vx_aux_desc* d, const uint64_t* trace_dev, int degree_bits, const uint64_t* challenges,
                         uint64_t* out_dev, uint64_t* closing_sums_out) {
  if (!c || !d || !trace_dev || !out_dev || !d->program) return vx_fail(VX_E_INVALID, "vx_stark_aux_columns: NULL argument");
  if (degree_bits < 1 || degree_bits > ROOT_TABLE_LOG) return vx_fail(VX_E_INVALID, "vx_stark_aux_columns: degree_bits out of range");
  if (d->num_columns < 1 || d->num_columns > 8192 || d->num_challenges < 0 || d->num_challenges > VX_AUX_MAX_CHALLENGES || d->num_fractions < 1 ||
      d->num_fractions > 4096 || d->num_sums < 0 || d->num_sums > 64 || d->program_len < 1 || d->program_len > (1 << 22))
    return vx_fail(VX_E_INVALID, "vx_stark_aux_columns: description out of range");
  if (d->num_challenges > 0 && !challenges) return vx_fail(VX_E_INVALID, "vx_stark_aux_columns: NULL challenges");
  if (d->num_sums > 0 && (!d->sum_coeffs || !closing_sums_out)) return vx_fail(VX_E_INVALID, "vx_stark_aux_columns: running sums need coefficients and a closing-sum buffer");
  try {
    const int nf = d->num_fractions, ns = d->num_sums, nout = nf + ns;
    // the program: straight-line, reads only what the description declares, pushes numerator / denominator pairs
    int pushes = 0;
    bool ended = false;
    for (int pc = 0; pc < d->program_len; ++pc) {
      const uint64_t ins = d->program[pc];
      const int op = (int)(ins & 0xFF), a = (int)((ins >> 16) & 0xFFFF);
      if (op == VX_OP_END) {
        ended = true;
        break;
      }
      if (op == VX_OP_LDI) {
        if (++pc >= d->program_len) return vx_fail(VX_E_INVALID, "vx_stark_aux_columns: LDI without its immediate");
      } else if (op == VX_OP_LDW) {
        if (a >= d->num_columns) return vx_fail(VX_E_INVALID, "vx_stark_aux_columns: the program reads column %d of %d", a, d->num_columns);
      } else if (op == VX_OP_LDCH) {
        if (a >= d->num_challenges) return vx_fail(VX_E_INVALID, "vx_stark_aux_columns: the program reads challenge %d of %d", a, d->num_challenges);
      } else if (op == VX_OP_PUSH) {
        ++pushes;
      } else if (op != VX_OP_ADD && op != VX_OP_SUB && op != VX_OP_MUL) {
        return vx_fail(VX_E_INVALID, "vx_stark_aux_columns: instruction %d is not allowed in an aux program", op);
      }
    }
    if (!ended || pushes != 2 * nf) return vx_fail(VX_E_INVALID, "vx_stark_aux_columns: the program must END and push %d (numerator, denominator) pairs, it pushes %d values", nf, pushes);
    std::vector<int> fo(nf), so(ns);
    std::vector<char> used((size_t)nout, 0);
    for (int k = 0; k < nf; ++k) fo[k] = d->fraction_out ? d->fraction_out[k] : k;
    for (int j = 0; j < ns; ++j) so[j] = d->sum_out ? d->sum_out[j] : nf + j;
    for (int v : fo)
      if (v < 0 || v >= nout || used[v]++) return vx_fail(VX_E_INVALID, "vx_stark_aux_columns: output columns must be a permutation of 0 .. %d", nout - 1);
    for (int v : so)
      if (v < 0 || v >= nout || used[v]++) return vx_fail(VX_E_INVALID, "vx_stark_aux_columns: output columns must be a permutation of 0 .. %d", nout - 1);
    for (int j = 0; j < ns; ++j)
      for (int k = 0; k < nf; ++k)
        if (d->sum_coeffs[(size_t)j * nf + k] < -1 || d->sum_coeffs[(size_t)j * nf + k] > 1) return vx_fail(VX_E_INVALID, "vx_stark_aux_columns: sum coefficients are -1, 0 or +1");
    HIPCHK(hipSetDevice(c->device));
    const size_t n = (size_t)1 << degree_bits;
    Scratch S(c);
    const size_t nblocks = (n + (size_t)VX_AUX_SCAN_THREADS * VX_AUX_SCAN_RUN - 1) / ((size_t)VX_AUX_SCAN_THREADS * VX_AUX_SCAN_RUN);
    u64* d_prog = S.get((size_t)d->program_len);
    int* d_fo = (int*)S.get((size_t)(nf + 1) / 2 + 1);
    int* d_so = (int*)S.get((size_t)(ns + 1) / 2 + 1);
    signed char* d_co = (signed char*)S.get(((size_t)ns * nf + 7) / 8 + 1);
    u64* totals = S.get(nblocks + 1);
    if (!d_prog || !d_fo || !d_so || !d_co || !totals) return vx_fail(VX_E_NOMEM, "vx_stark_aux_columns: out of device memory");
    HIPCHK(hipMemcpyAsync(d_prog, d->program, (size_t)d->program_len * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_fo, fo.data(), (size_t)nf * sizeof(int), hipMemcpyHostToDevice, c->stream));
    if (ns) {
      HIPCHK(hipMemcpyAsync(d_so, so.data(), (size_t)ns * sizeof(int), hipMemcpyHostToDevice, c->stream));
      HIPCHK(hipMemcpyAsync(d_co, d->sum_coeffs, (size_t)ns * nf, hipMemcpyHostToDevice, c->stream));
    }
    {
      ProfScope ps(c, "aux_fractions", 8.0 * (double)n * (double)nf);
      AuxFracParams fp;
      memset(&fp, 0, sizeof fp);
      fp.trace = trace_dev, fp.program = d_prog, fp.frac_out = d_fo, fp.out = out_dev, fp.n = n, fp.ncols = d->num_columns, fp.nfrac = nf;
      for (int i = 0; i < d->num_challenges; ++i) fp.chal[i] = vxh::canon(challenges[i]);
      // short traces: fewer than ~4 waves per SIMD (1024 SIMDs x 64 lanes) -> split the fractions over up to 8 parts
      int parts = (int)(((size_t)4 * 1024 * 64 + n - 1) / n);
      parts = std::max(1, std::min(std::min(parts, 8), nf));
      fp.parts = parts;
      // the program compiled to native code (jit.hip.h::jit_aux_source: registers in VGPRs, fused multiply-adds, the denominators
      // inverted 8 at a time) — the interpreter walks ~400 cycles per instruction per wavefront; it stays as the fallback (VX_NO_JIT=1)
      std::string why;
      hipFunction_t fn = jit_aux_get(d->program, nf, c->device, &why);
      if (fn) {
        fp.parts = 1;
        void* args[] = {&fp};
        HIPCHK(hipModuleLaunchKernel(fn, (unsigned)((n + 255) / 256), 1, 1, 256, 1, 1, 0, c->stream, args, nullptr));
      } else {
        hipLaunchKernelGGL(aux_fraction_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)parts), dim3(256), 0, c->stream, fp);
      }
      HIPCHK(hipGetLastError());
    }
    if (ns) {
      ProfScope ps(c, "aux_running_sums", 16.0 * (double)n * (double)ns);
      AuxSumParams sp;
      memset(&sp, 0, sizeof sp);
      sp.out = out_dev, sp.frac_out = d_fo, sp.sum_out = d_so, sp.coeff = d_co, sp.n = n, sp.nfrac = nf, sp.nsum = ns;
      hipLaunchKernelGGL(aux_rowsum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, sp);
      for (int j = 0; j < ns; ++j) {
        u64* col = out_dev + (size_t)so[j] * n;
        hipLaunchKernelGGL(aux_scan_blocks_kernel, dim3((unsigned)nblocks), dim3(VX_AUX_SCAN_THREADS), 0, c->stream, col, n, totals);
        hipLaunchKernelGGL(aux_scan_totals_kernel, dim3(1), dim3(VX_AUX_SCAN_THREADS), 0, c->stream, totals, nblocks, totals + nblocks);
        hipLaunchKernelGGL(aux_scan_offsets_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, col, n, totals);
        HIPCHK(hipMemcpyAsync(&closing_sums_out[j], col + (n - 1), 8, hipMemcpyDeviceToHost, c->stream));   // exclusive prefix on the last row = the sum over rows 0 .. n-2
      }
      HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return VX_OK;
  } catch (const std::bad_alloc&) {
    return vx_fail(VX_E_NOMEM, "vx_stark_aux_columns: out of host memory");
  } catch (const std::exception& e) {
    return vx_fail(VX_E_INVALID, "vx_stark_aux_columns: %s", e.what());
  }
}
int vx_stark_verify(const vx_stark_desc* d, const uint64_t* public_inputs, const uint8_t* proof, size_t proof_len) {
  // A table that announces closing sums is only proven once somebody has looked at them.  This entry point has no argument for
  // them, so it takes the one meaning a lone table can have: every sum is zero (what it sent it received itself).  Tables whose
  // sums cancel against OTHER tables go through vx_stark_verify_bus.
  if (!d) return vx_fail(VX_E_INVALID, "vx_stark_verify: NULL argument");
  const int ns = d->num_aux_public_inputs;
  if (ns < 0 || ns > (1 << 16)) return vx_fail(VX_E_INVALID, "vx_stark_verify: bad num_aux_public_inputs");
  std::vector<uint64_t> sums((size_t)std::max(ns, 1), 0);
  const int rc = vx_stark_verify_shared(d, public_inputs, proof, proof_len, nullptr, sums.data());
  if (rc != VX_OK) return rc;
  for (int i = 0; i < ns; ++i)
    if (vxh::canon(sums[i]) != 0)
      return vx_fail(VX_E_PROOF, "vx_stark_verify: closing sum %d of a table verified on its own is not zero (tables on a shared bus: vx_stark_verify_bus)", i);
  return VX_OK;
}
int vx_stark_verify_bus(const vx_stark_desc* const* descs, const uint64_t* const* public_inputs, const uint8_t* const* proofs,
                        const size_t* proof_lens, int num_tables, uint64_t* closing_sums_out) {
  if (!descs || !public_inputs || !proofs || !proof_lens || num_tables < 1 || num_tables > 4096)
    return vx_fail(VX_E_INVALID, "vx_stark_verify_bus: bad argument");
  try {
    for (int t = 0; t < num_tables; ++t)
      if (!descs[t] || !proofs[t]) return vx_fail(VX_E_INVALID, "vx_stark_verify_bus: table %d: NULL description or proof", t);
    const int nch = descs[0]->num_aux_challenges, ns = descs[0]->num_aux_public_inputs;
    if (nch < 1 || ns < 1) return vx_fail(VX_E_INVALID, "vx_stark_verify_bus: the tables of a bus draw shared challenges and announce closing sums");
    for (int t = 1; t < num_tables; ++t)
      if (descs[t]->num_aux_challenges != nch || descs[t]->num_aux_public_inputs != ns)
        return vx_fail(VX_E_INVALID, "vx_stark_verify_bus: table %d declares %d shared challenges / %d closing sums, table 0 %d / %d", t,
                       descs[t]->num_aux_challenges, descs[t]->num_aux_public_inputs, nch, ns);
    // the joint challenges come from the trace caps INSIDE the proofs (table order is part of the statement)
    std::vector<std::vector<uint64_t>> caps((size_t)num_tables);
    std::vector<const uint64_t*> cap_ptrs((size_t)num_tables);
    std::vector<int32_t> heights((size_t)num_tables);
    for (int t = 0; t < num_tables; ++t) {
      if (descs[t]->cap_height < 0 || descs[t]->cap_height > 24) return vx_fail(VX_E_INVALID, "vx_stark_verify_bus: table %d: bad cap_height", t);
      caps[t].resize((size_t)4 << descs[t]->cap_height);
      const int rc = vx_stark_proof_trace_cap(descs[t], proofs[t], proof_lens[t], caps[t].data());
      if (rc != VX_OK) return rc;
      cap_ptrs[t] = caps[t].data();
      heights[t] = descs[t]->cap_height;
    }
    std::vector<uint64_t> shared((size_t)nch);
    int rc = vx_stark_joint_challenges(cap_ptrs.data(), heights.data(), num_tables, nch, shared.data());
    if (rc != VX_OK) return rc;
    std::vector<uint64_t> sums((size_t)num_tables * ns);
    for (int t = 0; t < num_tables; ++t) {
      rc = vx_stark_verify_shared(descs[t], public_inputs[t], proofs[t], proof_lens[t], shared.data(), sums.data() + (size_t)t * ns);
      if (rc != VX_OK) return rc;
    }
    if (closing_sums_out) memcpy(closing_sums_out, sums.data(), sums.size() * sizeof(uint64_t));
    for (int i = 0; i < ns; ++i) {
      uint64_t acc = 0;
      for (int t = 0; t < num_tables; ++t) acc = vxh::add(acc, vxh::canon(sums[(size_t)t * ns + i]));
      if (acc != 0) return vx_fail(VX_E_PROOF, "vx_stark_verify_bus: closing sum %d does not cancel over the %d tables (every proof is valid; the bus is unbalanced)", i, num_tables);
    }
    return VX_OK;
  } catch (const std::bad_alloc&) {
    return vx_fail(VX_E_NOMEM, "vx_stark_verify_bus: out of host memory");
  } catch (const std::exception& e) {
    return vx_fail(VX_E_PROOF, "vx_stark_verify_bus: exception: %s", e.what());
  }
}
int vx_stark_verify_shared(const vx_stark_desc* d, const uint64_t* public_inputs, const uint8_t* proof, size_t proof_len,
                           const uint64_t* shared_challenges, uint64_t* aux_public_inputs_out) {
  if (!d || !proof || (d->num_public_inputs > 0 && !public_inputs)) return vx_fail(VX_E_INVALID, "vx_stark_verify: NULL argument");
  try {
    StarkShape sh;
    std::string why = stark_check(d, &sh);
    if (!why.empty()) return vx_fail(VX_E_INVALID, "vx_stark_verify: %s", why.c_str());
    why = vxsv::verify(d, sh, public_inputs, proof, proof_len, shared_challenges, aux_public_inputs_out);
    if (!why.empty()) return vx_fail(VX_E_PROOF, "vx_stark_verify: %s", why.c_str());
    return VX_OK;
  } catch (const std::bad_alloc&) {
    return vx_fail(VX_E_NOMEM, "vx_stark_verify: out of host memory");
  } catch (const std::exception& e) {
    return vx_fail(VX_E_PROOF, "vx_stark_verify: exception: %s", e.what());
  }
}

int vx_prove_sharded(vx_ctx* c, vx_circuit* k, const uint64_t* wires, int wires_on_device, int rank, int world,
                     vx_allgather_fn allgather, void* user, const uint64_t* pow_witness_hint, uint8_t* out_buf, size_t* out_len) {
  if (!c || !k || !wires || !out_buf || !out_len) return vx_fail(VX_E_INVALID, "vx_prove_sharded: NULL argument");
  if (k->ctx != c) return vx_fail(VX_E_INVALID, "vx_prove_sharded: circuit belongs to a different context");
  Shard sh;
  sh.rank = rank;
  sh.world = world;
  while ((1 << sh.lg) < world) ++sh.lg;
  if (world < 1 || (1 << sh.lg) != world || sh.lg > k->rate_bits || sh.lg > k->cap_height)
    return vx_fail(VX_E_INVALID, "vx_prove_sharded: world=%d must be a power of two <= 2^rate_bits and <= 2^cap_height", world);
  if (rank < 0 || rank >= world) return vx_fail(VX_E_INVALID, "vx_prove_sharded: rank %d outside [0, %d)", rank, world);
  if (world > 1 && !allgather) return vx_fail(VX_E_INVALID, "vx_prove_sharded: world > 1 needs an all-gather callback");
  sh.fn = allgather;
  sh.user = user;
  HIPCHK(hipSetDevice(c->device));
  std::vector<uint8_t> proof;
  int rc;
  try {  // nothing unwinds across the ABI
    rc = prove_impl(c, k, wires, wires_on_device != 0, pow_witness_hint, proof, sh);
  } catch (const std::bad_alloc&) {
    rc = vx_fail(VX_E_NOMEM, "vx_prove_sharded: out of host memory");
  } catch (const std::exception& e) {
    rc = vx_fail(VX_E_INVALID, "vx_prove_sharded: %s", e.what());
  }
  if (rc != VX_OK) {
    hipStreamSynchronize(c->stream);
    return rc;
  }
  if (proof.size() > *out_len) {
    *out_len = proof.size();
    return vx_fail(VX_E_INVALID, "vx_prove_sharded: output buffer too small, need %zu bytes", proof.size());
  }
  memcpy(out_buf, proof.data(), proof.size());
  *out_len = proof.size();
  return VX_OK;
}

// ---------------------------------------------------------------------------------------------
// In-process rank group: host threads + peer copies (include/vxprover.h)
// ---------------------------------------------------------------------------------------------
struct vx_group_member {
  vx_group* g;
  int rank;
  vx_ctx* ctx;
};
struct vx_group {
  int world = 0;
  std::mutex mu;
  std::condition_variable cv;
  int waiting = 0;
  uint64_t generation = 0;
  bool aborted = false;
  bool peer_staged = false;  // some pair of member devices has no direct peer access: its copies are host-staged
  std::vector<vx_group_member> members;
  std::vector<void*> bufs;
  std::vector<size_t> sizes;
  // returns false when the group was aborted — or when a member did not arrive within `timeout_ms` (a rank that died
  // cannot call vx_group_abort: its peers must not wait for it forever).  A timeout aborts the group for everybody.
  long long timeout_ms = 120000;
  bool timed_out = false;
  bool barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (aborted) return false;
    const uint64_t gen = generation;
    if (++waiting == world) {
      waiting = 0;
      ++generation;
      cv.notify_all();
      return true;
    }
    if (!cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return generation != gen || aborted; })) {
      aborted = true;
      timed_out = true;
      cv.notify_all();
      return false;
    }
    return !aborted;
  }
};

int vx_group_create(int world, vx_group** out) {
  if (!out || world < 1 || world > 64) return vx_fail(VX_E_INVALID, "vx_group_create: bad argument");
  vx_group* g = new vx_group();
  g->world = world;
  g->members.resize(world);
  g->bufs.assign(world, nullptr);
  g->sizes.assign(world, 0);
  for (int r = 0; r < world; ++r) g->members[r] = vx_group_member{g, r, nullptr};
  *out = g;
  return VX_OK;
}
void vx_group_destroy(vx_group* g) { delete g; }
int vx_group_set_timeout_ms(vx_group* g, long long ms) {
  if (!g || ms < 1) return vx_fail(VX_E_INVALID, "vx_group_set_timeout_ms: bad argument");
  std::lock_guard<std::mutex> lk(g->mu);
  g->timeout_ms = ms;
  return VX_OK;
}
void vx_group_abort(vx_group* g) {
  if (!g) return;
  std::lock_guard<std::mutex> lk(g->mu);
  g->aborted = true;
  g->cv.notify_all();
}
int vx_group_join(vx_group* g, int rank, vx_ctx* ctx, void** member_out) {
  if (!g || !ctx || !member_out || rank < 0 || rank >= g->world) return vx_fail(VX_E_INVALID, "vx_group_join: bad argument");
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->members[rank].ctx = ctx;
    // Direct xGMI peer copies need peer access enabled in BOTH directions between every pair of distinct devices of
    // the group (without it hipMemcpyPeerAsync is staged through host memory).  Enable it against every member that
    // has joined so far; "already enabled" is fine.  A pair without peer capability keeps working through the staged path.
    for (int r = 0; r < g->world; ++r) {
      vx_ctx* o = g->members[r].ctx;
      if (!o || r == rank || o->device == ctx->device) continue;
      const int pairs[2][2] = {{ctx->device, o->device}, {o->device, ctx->device}};
      for (auto& pr : pairs) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, pr[0], pr[1]) != hipSuccess || !can) {
          g->peer_staged = true;
          continue;
        }
        if (hipSetDevice(pr[0]) != hipSuccess) return vx_fail(VX_E_HIP, "vx_group_join: hipSetDevice(%d) failed", pr[0]);
        hipError_t e = hipDeviceEnablePeerAccess(pr[1], 0);
        if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();  // clear the sticky error
        else if (e != hipSuccess) return vx_fail(VX_E_HIP, "vx_group_join: hipDeviceEnablePeerAccess(%d -> %d): %s", pr[0], pr[1], hipGetErrorString(e));
      }
    }
    (void)hipSetDevice(ctx->device);
  }
  *member_out = &g->members[rank];
  return VX_OK;
}
int vx_group_peer_staged(vx_group* g) { return g && g->peer_staged ? 1 : 0; }
int vx_group_allgather(void* member, void* dev_buf, size_t bytes_per_rank) {
  vx_group_member* m = (vx_group_member*)member;
  if (!m || !m->g || !m->ctx || !dev_buf) return vx_fail(VX_E_INVALID, "vx_group_allgather: bad argument");
  vx_group* g = m->g;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->bufs[m->rank] = dev_buf;
    g->sizes[m->rank] = bytes_per_rank;
  }
  if (!g->barrier()) return vx_fail(VX_E_COMM, g->timed_out ? "vx_group_allgather: a rank did not arrive (timeout): group aborted" : "vx_group_allgather: group aborted");
  // every rank's buffer is published and its producer stream is idle: pull the other ranks' slots.  EVERY early return
  // below aborts the group first — the peers are blocked in a barrier that this rank will never reach.
  {
    hipError_t e = hipSetDevice(m->ctx->device);
    if (e != hipSuccess) {
      vx_group_abort(g);
      return vx_fail(VX_E_HIP, "vx_group_allgather: hipSetDevice(%d): %s", m->ctx->device, hipGetErrorString(e));
    }
  }
  for (int s = 0; s < g->world; ++s) {
    if (s == m->rank) continue;
    if (g->sizes[s] != bytes_per_rank) {
      vx_group_abort(g);
      return vx_fail(VX_E_COMM, "vx_group_allgather: rank %d offers %zu bytes, rank %d %zu", s, g->sizes[s], m->rank, bytes_per_rank);
    }
    const char* src = (const char*)g->bufs[s] + (size_t)s * bytes_per_rank;
    char* dst = (char*)dev_buf + (size_t)s * bytes_per_rank;
    hipError_t e = hipMemcpyPeerAsync(dst, m->ctx->device, src, g->members[s].ctx->device, bytes_per_rank, m->ctx->stream);
    if (e != hipSuccess) {
      vx_group_abort(g);
      return vx_fail(VX_E_HIP, "vx_group_allgather: hipMemcpyPeerAsync: %s", hipGetErrorString(e));
    }
  }
  hipError_t e = hipStreamSynchronize(m->ctx->stream);
  if (e != hipSuccess) {
    vx_group_abort(g);
    return vx_fail(VX_E_HIP, "vx_group_allgather: %s", hipGetErrorString(e));
  }
  // nobody may overwrite its slot before all ranks have read it
  if (!g->barrier()) return vx_fail(VX_E_COMM, g->timed_out ? "vx_group_allgather: a rank did not arrive (timeout): group aborted" : "vx_group_allgather: group aborted");
  return VX_OK;
}

}  // extern "C"
