// Trace generation for the chip tables ON THE GPU (round 5): vx_trace_sha256 / vx_trace_sha512 / vx_trace_blake2b.
//
// In the reference the Curta chips' witness generation fills the hash tables of every map / reduce / outer job on the CPU
// (/root/reference/circuits/builder/header.rs:14-19, subchain_verification.rs:148-231, justification.rs:140-156); until round 4
// this repository did the same in numpy, 100x slower than the proof the trace feeds.  Here the host only pads the messages and walks
// the chain of chaining values (tracegen_prep.h); the device expands:
//   * one thread per BLOCK computes the message schedule / work vectors the rows of that block need (tg_*_expand_kernel);
//   * one thread per ROW writes that row's cells, column-major — the 64 lanes of a wavefront write 64 consecutive rows of one
//     column (512 B coalesced) for each of the ~1000 columns (tg_*_rows_kernel), and counts the row's lookups;
//   * a last small kernel writes the lookup multiplicities into the table rows.
// HBM-bound by construction: 8 B per cell written once, nothing re-read (a 2^16 x 775 BLAKE2b table = 406 MB).
// Row semantics: tracegen_core.h (compared cell by cell with the numpy generators in tests/test_tracegen.py on the CPU and in
// tests/test_gpu_tracegen.py through this file).
#pragma once
#include "tracegen_prep.h"
#include "tracegen_eddsa.h"
#include "vx_runtime.hip.h"

#define TG_THREADS 256

// The lookup histograms of the BLAKE2b and EdDSA tables: 232 increments per BLAKE2b row (15 M per map job), ~92 per EdDSA row (96 M per
// 2^20-row table, on keys as good as random), into 65 536 bins.  What was measured on the way (profiles/README.md, round 5):
//   * device-scope atomics are performed at the fabric, not in an XCD's L2 (the eight L2s are not coherent with each other):
//     1.2 G/s, 12.9 ms per BLAKE2b table;
//   * one copy of the histogram PER XCD, atomics that stay in its L2 (workgroup scope: all the CUs that touch copy x sit behind L2 x;
//     HW_REG_XCC_ID names the XCD a workgroup runs on), equal keys merged across the wavefront first: 7 - 16 G/s — 1.9 ms per BLAKE2b
//     table, and 12.7 of the EdDSA rows kernel's 14 ms (0.3 ms were its 4.4 GB of cells; measured with either switched off);
//   * now: a workgroup counts in LDS.  1024 threads walk `rows_per_wg` rows and increment 32-bit counters for ONE HALF of the key
//     space (32 768 bins = 128 KB of the CU's 160 KB; blockIdx.y names the half and only half 0 stores the cells — the row arithmetic
//     is repeated, it is the cheap part), then add their non-zero bins to their XCD's copy: 32 768 global atomics per workgroup
//     instead of 232 / 92 per row.  0.38 ms per BLAKE2b table (1.1 TB/s of cells), 3 ms for the EdDSA rows.
// The kernel that writes the multiplicity columns adds the eight copies up.
#define TG_HIST_COPIES 8
__device__ __forceinline__ unsigned tg_xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & (TG_HIST_COPIES - 1);
}

template <class T>
// The flush of a workgroup's LDS histogram into ITS XCD's copy in global memory.  Different workgroups add to the same copy,
// so under the HSA memory model the read-modify-write needs AGENT scope; on gfx942 / gfx950 an agent-scope atomic is carried out at the
// memory side of the fabric (measured: 12.9 ms per BLAKE2b table against 0.38 ms), while a workgroup-scope one is carried out in the
// XCD's own L2 — which IS the point of coherence for every workgroup that can touch this copy, because HW_REG_XCC_ID picks the copy of
// the XCD the workgroup runs on.  That hardware fact is what the narrower scope relies on, so it is taken only when compiling for those
// two targets (anything else gets agent scope), and tests/test_gpu_tracegen.py compares the multiplicity columns with the host
// generators at the production shapes (a lost count would also leave the table's lookup bus unbalanced: the proof would not verify).
#if defined(__gfx942__) || defined(__gfx950__)
#define TG_XCD_COPY_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
#else
#define TG_XCD_COPY_SCOPE __HIP_MEMORY_SCOPE_AGENT
#endif
__global__ __launch_bounds__(TG_THREADS) void tg_sha2_expand_kernel(const tg::Sha2Block<T>* __restrict__ blocks, int nb,
                                                                    tg::Sha2Expanded<T>* __restrict__ exp) {
  const int b = blockIdx.x * TG_THREADS + threadIdx.x;
  if (b >= nb) return;
  tg::sha2_expand<T>(blocks[b], tg::Sha2Consts<T>::K(), exp[b]);
}
template <class T>
__global__ __launch_bounds__(TG_THREADS) void tg_sha2_rows_kernel(const tg::Sha2Block<T>* __restrict__ blocks,
                                                                  const tg::Sha2Expanded<T>* __restrict__ exp, u64* __restrict__ trace,
                                                                  size_t n, unsigned* __restrict__ hist) {
  __shared__ unsigned lh[8];
  if (threadIdx.x < 8) lh[threadIdx.x] = 0;
  __syncthreads();
  const size_t row = (size_t)blockIdx.x * TG_THREADS + threadIdx.x;
  if (row < n) {
    const size_t b = row / T::PERIOD;
    const int r = (int)(row % T::PERIOD);
    unsigned carries[T::NCARRY];
    tg::sha2_row<T>(blocks[b], exp[b], b ? &exp[b - 1] : nullptr, r, row, tg::Sha2Consts<T>::K(),
                    [&](int col, uint64_t v) { trace[(size_t)col * n + row] = v; }, carries);
    if (row + 1 < n)       // the last row is inert: its lookups are not counted
      for (int q = 0; q < T::NCARRY; ++q) atomicAdd(&lh[carries[q] & 7], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 8 && lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
}
// trace[col][i] = sum over the `copies` histograms (each `stride` entries apart) of hist[i], for i < count
__global__ void tg_patch_mult_kernel(u64* __restrict__ col, const unsigned* __restrict__ hist, unsigned count, int copies, size_t stride) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  u64 s = 0;
  for (int x = 0; x < copies; ++x) s += hist[(size_t)x * stride + i];
  col[i] = s;
}

__global__ __launch_bounds__(TG_THREADS) void tg_b2_expand_kernel(const tg::b2::Block* __restrict__ blocks, int nb,
                                                                  tg::b2::Expanded* __restrict__ exp) {
  const int b = blockIdx.x * TG_THREADS + threadIdx.x;
  if (b >= nb) return;
  tg::b2::expand(blocks[b], tg::B2_IV, tg::B2_SIGMA, exp[b]);
}
#define TG_LDS_THREADS 1024
#define TG_LDS_BINS 32768
static __device__ const uint64_t tg_zero8[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // the hand-over of "no block before" (in memory once, not in every lane's scratch)
__global__ __launch_bounds__(TG_LDS_THREADS) void tg_b2_rows_kernel(const tg::b2::Block* __restrict__ blocks,
                                                                    const tg::b2::Expanded* __restrict__ exp, u64* __restrict__ trace, size_t n,
                                                                    size_t rows_per_wg, unsigned* __restrict__ hist) {
  extern __shared__ __attribute__((aligned(16))) unsigned tg_lh[];
  for (int b = threadIdx.x; b < TG_LDS_BINS; b += TG_LDS_THREADS) tg_lh[b] = 0;
  __syncthreads();
  const unsigned half = blockIdx.y;
  const size_t base = (size_t)blockIdx.x * rows_per_wg;
  const size_t end = base + rows_per_wg < n ? base + rows_per_wg : n;
  for (size_t row = base + threadIdx.x; row < end; row += TG_LDS_THREADS) {
    const size_t b = row / tg::b2::PERIOD;
    const int r = (int)(row % tg::b2::PERIOD);
    const tg::b2::Block& blk = blocks[b];
    const uint64_t* hn_prev = b ? exp[b - 1].hn : tg_zero8;
    const uint64_t* dl = blk.dsrc >= 0 ? exp[blk.dsrc].hn : tg_zero8;
    const bool count = row + 1 < n;
    size_t nn = n;
    asm volatile("" : "+s"(nn));   // 775 column bases hoisted out of this loop were 957 spilled SGPRs; col * nn is two scalar multiplies
    tg::b2::row(blk, exp[b], hn_prev, dl, r, row, tg::B2_IV, tg::B2_SIGMA,
                [&](int col, uint64_t v) { if (half == 0) trace[(size_t)col * nn + row] = v; },
                [&](unsigned a, unsigned bb) {
                  const unsigned key = a * 256u + bb;
                  if (count && (key >> 15) == half) __hip_atomic_fetch_add(&tg_lh[key & (TG_LDS_BINS - 1)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                });
  }
  __syncthreads();
  unsigned* my_hist = hist + (size_t)tg_xcc_id() * 65536 + (size_t)half * TG_LDS_BINS;
  for (int b = threadIdx.x; b < TG_LDS_BINS; b += TG_LDS_THREADS) {
    const unsigned v = tg_lh[b];
    if (v) __hip_atomic_fetch_add(&my_hist[b], v, __ATOMIC_RELAXED, TG_XCD_COPY_SCOPE);
  }
}
// one workgroup per CU and half: up to 128 row chunks x 2 halves fill the 256 CUs of the chip in one round
static void tg_lds_grid(size_t n, size_t* chunks_out, size_t* rows_per_wg_out) {
  size_t chunks = (n + TG_LDS_THREADS - 1) / TG_LDS_THREADS;
  if (chunks > 128) chunks = 128;
  const size_t rows_per_wg = ((n + chunks - 1) / chunks + TG_LDS_THREADS - 1) / TG_LDS_THREADS * TG_LDS_THREADS;
  *chunks_out = (n + rows_per_wg - 1) / rows_per_wg, *rows_per_wg_out = rows_per_wg;
}
static hipError_t tg_launch_b2_rows(hipStream_t s, const tg::b2::Block* blocks, const tg::b2::Expanded* exp, u64* trace, size_t n, unsigned* hist) {
  static bool attr_set[16] = {};
  int dev = 0;
  hipGetDevice(&dev);
  const size_t lds = (size_t)TG_LDS_BINS * sizeof(unsigned);
  if (!attr_set[dev & 15]) {
    hipError_t e = hipFuncSetAttribute((const void*)tg_b2_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set[dev & 15] = true;
  }
  size_t chunks, rows_per_wg;
  tg_lds_grid(n, &chunks, &rows_per_wg);
  hipLaunchKernelGGL(tg_b2_rows_kernel, dim3((unsigned)chunks, 2), dim3(TG_LDS_THREADS), lds, s, blocks, exp, trace, n, rows_per_wg, hist);
  return hipGetLastError();
}

// ---- C ABI ----------------------------------------------------------------------------------------------------------------------
template <class T>
static int tg_trace_sha2(vx_ctx* c, int degree_bits, const uint8_t* msgs, const uint64_t* off, int nmsg, void* d_trace,
                         uint64_t* pis_out, uint8_t* digests_out, const char* name) {
  if (!c || !d_trace) return vx_fail(VX_E_INVALID, "%s: NULL argument", name);
  HIPCHK(hipSetDevice(c->device));
  tg::Sha2Prep<T> prep;
  const int pr = tg::sha2_prepare<T>(degree_bits, msgs, off, nmsg, prep);
  if (pr == tg::PREP_TOO_MANY_BLOCKS) return vx_fail(VX_E_INVALID, "%s: the messages do not complete inside 2^%d rows", name, degree_bits);
  if (pr != tg::PREP_OK) return vx_fail(VX_E_INVALID, "%s: bad arguments", name);
  const size_t n = (size_t)1 << degree_bits;
  const int nb = (int)prep.blocks.size();
  void *d_blocks = nullptr, *d_exp = nullptr, *d_hist = nullptr;
  const size_t bb = (size_t)nb * sizeof(tg::Sha2Block<T>), eb = (size_t)nb * sizeof(tg::Sha2Expanded<T>);
  if (c->pool_alloc(&d_blocks, bb) != hipSuccess || c->pool_alloc(&d_exp, eb) != hipSuccess || c->pool_alloc(&d_hist, 8 * sizeof(unsigned)) != hipSuccess) {
    c->pool_free(d_blocks), c->pool_free(d_exp), c->pool_free(d_hist);
    return vx_fail(VX_E_NOMEM, "%s: out of device memory", name);
  }
  int rc = VX_OK;
  {
    ProfScope ps(c, "trace_generation", 8.0 * T::N * n);
    hipError_t e = hipMemcpyAsync(d_blocks, prep.blocks.data(), bb, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_hist, 0, 8 * sizeof(unsigned), c->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(tg_sha2_expand_kernel<T>, dim3((nb + TG_THREADS - 1) / TG_THREADS), dim3(TG_THREADS), 0, c->stream,
                         (const tg::Sha2Block<T>*)d_blocks, nb, (tg::Sha2Expanded<T>*)d_exp);
      hipLaunchKernelGGL(tg_sha2_rows_kernel<T>, dim3((unsigned)((n + TG_THREADS - 1) / TG_THREADS)), dim3(TG_THREADS), 0, c->stream,
                         (const tg::Sha2Block<T>*)d_blocks, (const tg::Sha2Expanded<T>*)d_exp, (u64*)d_trace, n, (unsigned*)d_hist);
      hipLaunchKernelGGL(tg_patch_mult_kernel, dim3(1), dim3(64), 0, c->stream, (u64*)d_trace + (size_t)T::MULT * n, (const unsigned*)d_hist, 8u, 1, (size_t)0);
      e = hipGetLastError();
    }
    { const hipError_t es = hipStreamSynchronize(c->stream); if (e == hipSuccess) e = es; }   // ALWAYS: the host block list goes out of scope, the pool buffers go back
    if (e != hipSuccess) rc = vx_fail(VX_E_HIP, "%s: %s", name, hipGetErrorString(e));
  }
  c->pool_free(d_blocks), c->pool_free(d_exp), c->pool_free(d_hist);
  if (rc != VX_OK) return rc;
  if (pis_out) memcpy(pis_out, prep.pis, sizeof prep.pis);
  if (digests_out)
    for (int mi = 0; mi < nmsg; ++mi)
      for (int k = 0; k < 8; ++k) {
        const typename T::W w = prep.digests[(size_t)mi * 8 + k];
        for (size_t q = 0; q < sizeof w; ++q) digests_out[((size_t)mi * 8 + k) * sizeof w + q] = (uint8_t)(w >> (8 * (sizeof w - 1 - q)));
      }
  return VX_OK;
}

// no C++ exception crosses the ABI: the preparation steps allocate host vectors
#define VX_TRACE_GUARD(name, call)                                                      \
  try {                                                                                 \
    return call;                                                                        \
  } catch (const std::bad_alloc&) {                                                     \
    return vx_fail(VX_E_NOMEM, name ": out of host memory");                            \
  } catch (const std::exception& e) {                                                   \
    return vx_fail(VX_E_INVALID, name ": %s", e.what());                                \
  }
int vx_trace_sha256(vx_ctx* c, int degree_bits, const uint8_t* msgs, const uint64_t* offsets, int num_msgs, void* trace_dev,
                    uint64_t* public_inputs_out, uint8_t* digests_out) {
  VX_TRACE_GUARD("vx_trace_sha256", tg_trace_sha2<tg::Sha256T>(c, degree_bits, msgs, offsets, num_msgs, trace_dev, public_inputs_out, digests_out, "vx_trace_sha256"))
}
int vx_trace_sha512(vx_ctx* c, int degree_bits, const uint8_t* msgs, const uint64_t* offsets, int num_msgs, void* trace_dev,
                    uint64_t* public_inputs_out, uint8_t* digests_out) {
  VX_TRACE_GUARD("vx_trace_sha512", tg_trace_sha2<tg::Sha512T>(c, degree_bits, msgs, offsets, num_msgs, trace_dev, public_inputs_out, digests_out, "vx_trace_sha512"))
}
int vx_trace_sha512_bus(vx_ctx* c, int degree_bits, const uint8_t* msgs, const uint64_t* offsets, int num_msgs, void* trace_dev,
                        uint64_t* public_inputs_out, uint8_t* digests_out) {
  VX_TRACE_GUARD("vx_trace_sha512_bus", tg_trace_sha2<tg::Sha512BusT>(c, degree_bits, msgs, offsets, num_msgs, trace_dev, public_inputs_out, digests_out, "vx_trace_sha512_bus"))
}
static int tg_trace_blake2b(vx_ctx* c, int degree_bits, const uint8_t* msgs, const uint64_t* offsets, int num_msgs, void* trace_dev,
                            uint64_t* public_inputs_out, uint8_t* digests_out);
int vx_trace_blake2b(vx_ctx* c, int degree_bits, const uint8_t* msgs, const uint64_t* offsets, int num_msgs, void* trace_dev,
                     uint64_t* public_inputs_out, uint8_t* digests_out) {
  VX_TRACE_GUARD("vx_trace_blake2b", tg_trace_blake2b(c, degree_bits, msgs, offsets, num_msgs, trace_dev, public_inputs_out, digests_out))
}
static int tg_trace_blake2b(vx_ctx* c, int degree_bits, const uint8_t* msgs, const uint64_t* offsets, int num_msgs, void* trace_dev,
                            uint64_t* public_inputs_out, uint8_t* digests_out) {
  if (!c || !trace_dev) return vx_fail(VX_E_INVALID, "vx_trace_blake2b: NULL argument");
  HIPCHK(hipSetDevice(c->device));
  tg::B2Prep prep;
  const int pr = tg::b2_prepare(degree_bits, msgs, offsets, num_msgs, prep);
  if (pr == tg::PREP_TOO_MANY_BLOCKS) return vx_fail(VX_E_INVALID, "vx_trace_blake2b: the messages do not complete inside 2^%d rows", degree_bits);
  if (pr != tg::PREP_OK) return vx_fail(VX_E_INVALID, "vx_trace_blake2b: bad arguments (the XOR table needs degree_bits >= 16)");
  const size_t n = (size_t)1 << degree_bits;
  const int nb = (int)prep.blocks.size();
  void *d_blocks = nullptr, *d_exp = nullptr, *d_hist = nullptr;
  const size_t bb = (size_t)nb * sizeof(tg::b2::Block), eb = (size_t)nb * sizeof(tg::b2::Expanded), hb = (size_t)TG_HIST_COPIES * 65536 * sizeof(unsigned);
  if (c->pool_alloc(&d_blocks, bb) != hipSuccess || c->pool_alloc(&d_exp, eb) != hipSuccess || c->pool_alloc(&d_hist, hb) != hipSuccess) {
    c->pool_free(d_blocks), c->pool_free(d_exp), c->pool_free(d_hist);
    return vx_fail(VX_E_NOMEM, "vx_trace_blake2b: out of device memory");
  }
  int rc = VX_OK;
  {
    ProfScope ps(c, "trace_generation", 8.0 * tg::b2::N * n);
    hipError_t e = hipMemcpyAsync(d_blocks, prep.blocks.data(), bb, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_hist, 0, hb, c->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(tg_b2_expand_kernel, dim3((nb + TG_THREADS - 1) / TG_THREADS), dim3(TG_THREADS), 0, c->stream,
                         (const tg::b2::Block*)d_blocks, nb, (tg::b2::Expanded*)d_exp);
      e = tg_launch_b2_rows(c->stream, (const tg::b2::Block*)d_blocks, (const tg::b2::Expanded*)d_exp, (u64*)trace_dev, n, (unsigned*)d_hist);
      for (int k = 0; k < tg::b2::NTAB; ++k)
        hipLaunchKernelGGL(tg_patch_mult_kernel, dim3(tg::b2::TAB_ROWS / 256), dim3(256), 0, c->stream,
                           (u64*)trace_dev + (size_t)tg::b2::tabcol(k, 19) * n, (const unsigned*)d_hist + (size_t)tg::b2::TAB_ROWS * k,
                           (unsigned)tg::b2::TAB_ROWS, TG_HIST_COPIES, (size_t)65536);
      if (e == hipSuccess) e = hipGetLastError();
    }
    { const hipError_t es = hipStreamSynchronize(c->stream); if (e == hipSuccess) e = es; }   // ALWAYS, also after an error: copies into stack slots / host blocks may be queued and the pool buffers go back below
    if (e != hipSuccess) rc = vx_fail(VX_E_HIP, "vx_trace_blake2b: %s", hipGetErrorString(e));
  }
  c->pool_free(d_blocks), c->pool_free(d_exp), c->pool_free(d_hist);
  if (rc != VX_OK) return rc;
  if (public_inputs_out) memcpy(public_inputs_out, prep.pis, sizeof prep.pis);
  if (digests_out)
    for (int mi = 0; mi < num_msgs; ++mi)
      for (int k = 0; k < 4; ++k)
        for (int q = 0; q < 8; ++q) digests_out[((size_t)mi * 4 + k) * 8 + q] = (uint8_t)(prep.digests[(size_t)mi * 4 + k] >> (8 * q));
  return VX_OK;
}


// ---- the batched EdDSA table -------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void tg_ed_simulate_kernel(tg::ed::Cols c, const tg::ed::Sig* __restrict__ sigs, int nsig, tg::ed::Sig filler, int ninst,
                                                            tg::ed::RowVals* __restrict__ vals, int* __restrict__ bad) {
  __shared__ uint64_t regs[64][tg::ed::NREG][4];
  const int u = blockIdx.x * 64 + threadIdx.x;
  if (u >= ninst) return;
  const int why = tg::ed::simulate_instance(c, u < nsig ? sigs[u] : filler, vals + (size_t)u * c.L, regs[threadIdx.x]);
  if (why && u < nsig) atomicExch(bad, (why << 24) | (u + 1));
}
// ---- the same simulation with FOUR LANES PER SIGNATURE (round 6) --------------------------------------------------------------------
// One lane per signature walks 10 772 dependent rows and leaves all but two wavefronts of the chip idle; nothing hides its instruction
// latency.  But the 42 rows of a ladder step are 13 LEVELS of up to four mutually independent rows of one kind — three or four
// multiplications of a doubling / a mixed addition, three or four additions and subtractions (tools/gen_tracegen_consts.py levelise ->
// LEVELS in tracegen_eddsa_ops.h) — so four adjacent lanes take a level at a time: every lane reads its row's operands from the
// signature's register file in LDS (all reads of a level precede its writes: lockstep lanes of one wavefront), runs the level's kind
// of row (mul_add_divmod or linear_row: the same functions the one-lane walk and the host test build use), writes its destination
// register and its row's x / y / z / q.  13 row-times per step instead of 42.  The prologue and the epilogue (20 - 43 rows of 10 772, the
// inversions among them) stay with lane 0 of the group.  Same cells as the one-lane kernel (tests/test_gpu_tracegen.py compares the
// trace with the numpy generator cell by cell); VX_TRACE_EDDSA_ONE_LANE=1 keeps the one-lane kernel.
template <int LV>
__device__ __forceinline__ void tg_ed_level(tg::ed::RowVals* __restrict__ out, uint64_t (*regs)[4], const uint64_t (*ycon)[4][4], const uint64_t (*ycoff)[4][4],
                                            int j, bool active, int rho0, int sbit, int hbit) {
  using namespace tg::ed;
  constexpr LvRow r0 = LEVELS[LV][0], r1 = LEVELS[LV][1], r2 = LEVELS[LV][2], r3 = LEVELS[LV][3];
  auto sel = [&](int a0, int a1, int a2, int a3) { return j == 0 ? a0 : j == 1 ? a1 : j == 2 ? a2 : a3; };
  const int row = sel(r0.row, r1.row, r2.row, r3.row), xi = sel(r0.x, r1.x, r2.x, r3.x), ei = sel(r0.e, r1.e, r2.e, r3.e), dst = sel(r0.dst, r1.dst, r2.dst, r3.dst);
  uint64_t x[4], y[4], e[4], z[4], q[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) x[k] = regs[xi][k];
#pragma unroll
  for (int k = 0; k < 4; ++k) e[k] = regs[ei < 0 ? 0 : ei][k] & (ei < 0 ? 0ull : ~0ull);
  if constexpr (LEVEL_LIN[LV] == 0) {
    const int yk = sel(r0.ykind, r1.ykind, r2.ykind, r3.ykind), yreg = sel(r0.yreg, r1.yreg, r2.yreg, r3.yreg);
    const bool on = yk == 2 ? sbit != 0 : hbit != 0;
    const bool from_reg = yk == 0 || (yk == 3 && on), from_con = yk == 1 || (yk == 2 && on);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint64_t yr = regs[yreg][k], yc = ycon[LV][j][k], yo = ycoff[LV][j][k];
      y[k] = from_reg ? yr : (from_con ? yc : yo);
    }
    mul_add_divmod(x, y, e, z, q);
  } else {
    const int lin = sel(r0.lin, r1.lin, r2.lin, r3.lin);          // linear_kind's code: 1 + c for Y = c, 4 + c for Y = p - c
    const uint64_t cc = (uint64_t)(lin <= 3 ? lin - 1 : lin - 4);
    const uint64_t neg = lin <= 3 ? 0ull : ~0ull;                  // Y = p - c: the words of p with c taken off the lowest
    y[0] = neg ? 0xFFFFFFFFFFFFFFEDull - cc : cc, y[1] = neg, y[2] = neg, y[3] = neg & 0x7FFFFFFFFFFFFFFFull;
    if (below_p(x) && below_p(e)) linear_row(lin, x, e, z, q);
    else mul_add_divmod(x, y, e, z, q);                            // (a non-canonical register cannot occur in the ladder; exactness first)
  }
  if (row >= 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) regs[dst][k] = z[k];
    if (active) {
      RowVals& rv = out[rho0 + row];
#pragma unroll
      for (int k = 0; k < 4; ++k) rv.x[k] = x[k], rv.y[k] = y[k], rv.z[k] = z[k], rv.q[k] = q[k];
    }
  }
  __syncthreads();   // one wavefront per block: orders the level's LDS writes before the next level's reads, for the compiler too
}
template <int LV>
struct TgEdLevels {
  static __device__ __forceinline__ void run(tg::ed::RowVals* __restrict__ out, uint64_t (*regs)[4], const uint64_t (*ycon)[4][4], const uint64_t (*ycoff)[4][4], int j,
                                             bool active, int rho0, int sbit, int hbit) {
    tg_ed_level<LV>(out, regs, ycon, ycoff, j, active, rho0, sbit, hbit);
    TgEdLevels<LV + 1>::run(out, regs, ycon, ycoff, j, active, rho0, sbit, hbit);
  }
};
template <>
struct TgEdLevels<tg::ed::NLEV> {
  static __device__ __forceinline__ void run(tg::ed::RowVals*, uint64_t (*)[4], const uint64_t (*)[4][4], const uint64_t (*)[4][4], int, bool, int, int, int) {}
};
__global__ __launch_bounds__(64) void tg_ed_simulate4_kernel(tg::ed::Cols c, const tg::ed::Sig* __restrict__ sigs, int nsig, tg::ed::Sig filler, int ninst,
                                                             tg::ed::RowVals* __restrict__ vals, int* __restrict__ bad) {
  using namespace tg::ed;
  __shared__ uint64_t regs[16][NREG][4];
  __shared__ uint64_t ycon[NLEV][4][4], ycoff[NLEV][4][4];
  for (int t = threadIdx.x; t < NLEV * 16; t += 64) {
    (&ycon[0][0][0])[t] = (&LEVEL_CON[0][0][0])[t];
    (&ycoff[0][0][0])[t] = (&LEVEL_COFF[0][0][0])[t];
  }
  const int grp = threadIdx.x >> 2, j = threadIdx.x & 3;
  const int u = blockIdx.x * 16 + grp;
  const bool active = u < ninst;
  const Sig& sg = active && u < nsig ? sigs[u] : filler;
  RowVals* out = vals + (size_t)(active ? u : 0) * c.L;
  uint64_t (*rg)[4] = regs[grp];
  const int NB = c.NB;
  int why = 0;
  const int s0 = scalar_bit(sg.s, NB, 0), h0 = scalar_bit(sg.h, NB, 0);
  if (j == 0) {
    for (int r = 0; r < NREG; ++r)
      for (int k = 0; k < 4; ++k) rg[r][k] = 0;
    if (active)
      for (int rho = 0; rho < c.NP; ++rho) simulate_row(c, sg, out, rg, op_at(c, rho), rho, s0, h0, why);
  }
  __syncthreads();
  uint64_t sw = 0, hw = 0;
  for (int step = 0; step < NB; ++step) {
    const int b = NB - 1 - step;
    if (step == 0 || (b & 63) == 63) sw = sg.s[b >> 6], hw = sg.h[b >> 6];
    TgEdLevels<0>::run(out, rg, ycon, ycoff, j, active, c.NP + NLOOP * step, (int)((sw >> (b & 63)) & 1), (int)((hw >> (b & 63)) & 1));
  }
  if (j == 0 && active) {
    for (int t = 0; t < c.NE; ++t) simulate_row(c, sg, out, rg, op_at(c, c.NP + NLOOP + t), c.NP + NLOOP * NB + t, s0, h0, why);
    if (why && u < nsig) atomicExch(bad, (why << 24) | (u + 1));
  }
}
// results[u][w] = z of rows XROW / YROW of instance u (the affine x, y the instance arrives at)
__global__ void tg_ed_results_kernel(const tg::ed::RowVals* __restrict__ vals, int nsig, int L, int xrow, u64* __restrict__ results) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nsig * 8) return;
  const int u = i >> 3, w = (i >> 2) & 1, k = i & 3;
  results[i] = vals[(size_t)u * L + xrow + w].z[k];
}
// (the LDS histogram described at the top of this file; 512 lanes per workgroup — one workgroup per CU either way, the histogram is 128 KB —
// so that a lane may hold the row's four 16-limb operands in registers: at 1024 lanes, 128 VGPRs, 53 of them spilled)
#define TG_ED_ROWS_THREADS 512
__global__ __launch_bounds__(TG_ED_ROWS_THREADS) void tg_ed_rows_kernel(tg::ed::Cols c, const tg::ed::RegSrc* __restrict__ rsrc, const tg::ed::RowVals* __restrict__ vals,
                                                                    const tg::ed::Sig* __restrict__ sigs, int nsig, tg::ed::Sig filler,
                                                                    u64* __restrict__ trace, size_t n, size_t rows_per_wg, unsigned* __restrict__ hist) {
  extern __shared__ __attribute__((aligned(16))) unsigned tg_lh[];
  for (int b = threadIdx.x; b < TG_LDS_BINS; b += TG_ED_ROWS_THREADS) tg_lh[b] = 0;
  __syncthreads();
  const unsigned half = blockIdx.y;
  const size_t base = (size_t)blockIdx.x * rows_per_wg;
  const size_t end = base + rows_per_wg < n ? base + rows_per_wg : n;
  for (size_t row = base + threadIdx.x; row < end; row += TG_ED_ROWS_THREADS) {
    const bool count = row + 1 < n;
    tg::ed::row(c, *rsrc, vals, sigs, nsig, filler, row, [&](int col, uint64_t v) { if (half == 0) trace[(size_t)col * n + row] = v; },
                [&](unsigned limb) {
                  if (count && (limb >> 15) == half) __hip_atomic_fetch_add(&tg_lh[limb & (TG_LDS_BINS - 1)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                });
  }
  __syncthreads();
  unsigned* my_hist = hist + (size_t)tg_xcc_id() * 65536 + (size_t)half * TG_LDS_BINS;
  for (int b = threadIdx.x; b < TG_LDS_BINS; b += TG_ED_ROWS_THREADS) {
    const unsigned v = tg_lh[b];
    if (v) __hip_atomic_fetch_add(&my_hist[b], v, __ATOMIC_RELAXED, TG_XCD_COPY_SCOPE);
  }
}
static hipError_t tg_launch_ed_rows(hipStream_t s, const tg::ed::Cols& cl, const tg::ed::RegSrc* rsrc, const tg::ed::RowVals* vals, const tg::ed::Sig* sigs, int nsig,
                                    const tg::ed::Sig& filler, u64* trace, size_t n, unsigned* hist) {
  static bool attr_set[16] = {};
  int dev = 0;
  hipGetDevice(&dev);
  const size_t lds = (size_t)TG_LDS_BINS * sizeof(unsigned);
  if (!attr_set[dev & 15]) {
    hipError_t e = hipFuncSetAttribute((const void*)tg_ed_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set[dev & 15] = true;
  }
  size_t chunks, rows_per_wg;
  tg_lds_grid(n, &chunks, &rows_per_wg);
  hipLaunchKernelGGL(tg_ed_rows_kernel, dim3((unsigned)chunks, 2), dim3(TG_ED_ROWS_THREADS), lds, s, cl, rsrc, vals, sigs, nsig, filler, trace, n, rows_per_wg, hist);
  return hipGetLastError();
}

static int tg_trace_eddsa(vx_ctx* c, int degree_bits, int scalar_bits, int full, const uint64_t* sigs, int num_sigs, void* trace_dev, uint64_t* results_out);
int vx_trace_eddsa(vx_ctx* c, int degree_bits, int scalar_bits, int full, const uint64_t* sigs, int num_sigs, void* trace_dev, uint64_t* results_out) {
  VX_TRACE_GUARD("vx_trace_eddsa", tg_trace_eddsa(c, degree_bits, scalar_bits, full, sigs, num_sigs, trace_dev, results_out))
}
static int tg_trace_eddsa(vx_ctx* c, int degree_bits, int scalar_bits, int full, const uint64_t* sigs, int num_sigs, void* trace_dev, uint64_t* results_out) {
  if (!c || !trace_dev || (num_sigs && !sigs)) return vx_fail(VX_E_INVALID, "vx_trace_eddsa: NULL argument");
  if (scalar_bits < 32 || scalar_bits > 256 || scalar_bits % 32) return vx_fail(VX_E_INVALID, "vx_trace_eddsa: scalar_bits must be a multiple of 32 in [32, 256]");
  if (full && scalar_bits != 256) return vx_fail(VX_E_INVALID, "vx_trace_eddsa: the full program reduces a SHA-512 digest mod L: scalar_bits must be 256");
  const tg::ed::Cols cl = tg::ed::cols(scalar_bits, full ? 1 : 0);
  if (degree_bits <= tg::ed::LB || degree_bits > 26) return vx_fail(VX_E_INVALID, "vx_trace_eddsa: the 16-bit limb table needs more than 2^16 rows");
  const size_t n = (size_t)1 << degree_bits;
  const size_t cap = (n - 1) / cl.L;
  if (num_sigs < 0 || (size_t)num_sigs > cap) return vx_fail(VX_E_INVALID, "vx_trace_eddsa: 2^%d rows hold %zu instances, %d given", degree_bits, cap, num_sigs);
  const int stride = full ? 24 : 16;                       // words per signature: A.x, A.y, S, h [, digest low half, digest high half]
  std::vector<tg::ed::Sig> host((size_t)(num_sigs ? num_sigs : 1));
  memset(host.data(), 0, host.size() * sizeof(tg::ed::Sig));
  for (int i = 0; i < num_sigs; ++i) {
    const uint64_t* w = sigs + (size_t)i * stride;
    memcpy(host[i].ax, w, 32), memcpy(host[i].ay, w + 4, 32), memcpy(host[i].s, w + 8, 32), memcpy(host[i].h, w + 12, 32);
    if (full) memcpy(host[i].d, w + 16, 64);
    for (int k = 0; k < 4; ++k)
      if (64 * k >= scalar_bits && (host[i].h[k] || (!full && host[i].s[k])))
        return vx_fail(VX_E_INVALID, "vx_trace_eddsa: a scalar of signature %d exceeds %d bits", i, scalar_bits);
  }
  HIPCHK(hipSetDevice(c->device));
  const int ninst = (int)((n + cl.L - 1) / cl.L);
  tg::ed::Sig filler;
  memset(&filler, 0, sizeof filler);
  for (int k = 0; k < 4; ++k) filler.ax[k] = tg::ed::BX[k], filler.ay[k] = tg::ed::BY[k];
  void *d_sigs = nullptr, *d_vals = nullptr, *d_hist = nullptr, *d_bad = nullptr, *d_rsrc = nullptr, *d_res = nullptr;
  const size_t sb = host.size() * sizeof(tg::ed::Sig), vb = (size_t)ninst * cl.L * sizeof(tg::ed::RowVals),
               hb = (size_t)TG_HIST_COPIES * 65536 * sizeof(unsigned), rb = host.size() * 64;
  if (c->pool_alloc(&d_sigs, sb) != hipSuccess || c->pool_alloc(&d_vals, vb) != hipSuccess || c->pool_alloc(&d_hist, hb) != hipSuccess ||
      c->pool_alloc(&d_bad, 256) != hipSuccess || c->pool_alloc(&d_rsrc, sizeof(tg::ed::RegSrc)) != hipSuccess || c->pool_alloc(&d_res, rb) != hipSuccess) {
    c->pool_free(d_sigs), c->pool_free(d_vals), c->pool_free(d_hist), c->pool_free(d_bad), c->pool_free(d_rsrc), c->pool_free(d_res);
    return vx_fail(VX_E_NOMEM, "vx_trace_eddsa: out of device memory");
  }
  tg::ed::RegSrc rsrc;
  memset(&rsrc, 0, sizeof rsrc);
  tg::ed::make_reg_src(cl, rsrc);
  int rc = VX_OK, bad = 0;
  {
    ProfScope ps(c, "trace_generation", 8.0 * cl.N * n);
    hipError_t e = hipMemcpyAsync(d_sigs, host.data(), sb, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_hist, 0, hb, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, 256, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_rsrc, &rsrc, sizeof rsrc, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
      static const bool one_lane = getenv("VX_TRACE_EDDSA_ONE_LANE") != nullptr;   // the one-lane-per-signature kernel of rounds 5-6 (A/B, cross-check)
      if (one_lane)
        hipLaunchKernelGGL(tg_ed_simulate_kernel, dim3((ninst + 63) / 64), dim3(64), 0, c->stream, cl, (const tg::ed::Sig*)d_sigs, num_sigs, filler, ninst,
                           (tg::ed::RowVals*)d_vals, (int*)d_bad);
      else
        hipLaunchKernelGGL(tg_ed_simulate4_kernel, dim3((ninst + 15) / 16), dim3(64), 0, c->stream, cl, (const tg::ed::Sig*)d_sigs, num_sigs, filler, ninst,
                           (tg::ed::RowVals*)d_vals, (int*)d_bad);
      e = tg_launch_ed_rows(c->stream, cl, (const tg::ed::RegSrc*)d_rsrc, (const tg::ed::RowVals*)d_vals, (const tg::ed::Sig*)d_sigs, num_sigs, filler,
                            (u64*)trace_dev, n, (unsigned*)d_hist);
      hipLaunchKernelGGL(tg_patch_mult_kernel, dim3(65536 / 256), dim3(256), 0, c->stream, (u64*)trace_dev + (size_t)cl.MULT * n, (const unsigned*)d_hist,
                         65536u, TG_HIST_COPIES, (size_t)65536);
      if (e == hipSuccess) e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, c->stream);
    // the results: affine (x, y) every given instance arrives at, gathered on the device, one copy
    if (e == hipSuccess && results_out && num_sigs) {
      hipLaunchKernelGGL(tg_ed_results_kernel, dim3((num_sigs * 8 + 255) / 256), dim3(256), 0, c->stream, (const tg::ed::RowVals*)d_vals, num_sigs, cl.L, cl.XROW,
                         (u64*)d_res);
      e = hipMemcpyAsync(results_out, d_res, (size_t)num_sigs * 64, hipMemcpyDeviceToHost, c->stream);
    }
    { const hipError_t es = hipStreamSynchronize(c->stream); if (e == hipSuccess) e = es; }   // ALWAYS, also after an error: copies into stack slots / host blocks may be queued and the pool buffers go back below
    if (e != hipSuccess) rc = vx_fail(VX_E_HIP, "vx_trace_eddsa: %s", hipGetErrorString(e));
  }
  c->pool_free(d_sigs), c->pool_free(d_vals), c->pool_free(d_hist), c->pool_free(d_bad), c->pool_free(d_rsrc), c->pool_free(d_res);
  if (rc == VX_OK && bad) {
    const int why = bad >> 24, who = (bad & 0xFFFFFF) - 1;
    rc = why == 1 ? vx_fail(VX_E_INVALID, "vx_trace_eddsa: instance %d: a row that must produce 1 does not (A not on the curve, or Z = 0)", who)
       : why == 2 ? vx_fail(VX_E_INVALID, "vx_trace_eddsa: instance %d: a value that must be canonical is not (a coordinate >= p, or S / the reduced digest >= L)", who)
                  : vx_fail(VX_E_INVALID, "vx_trace_eddsa: instance %d: an integer identity of the full program does not hold", who);
  }
  return rc;
}
