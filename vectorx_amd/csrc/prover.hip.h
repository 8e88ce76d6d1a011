// Host orchestration of a whole proof on one MI355X: the MI355X-native counterpart of
// plonky2::plonk::prover::prove_with_partition_witness + fri::oracle::PolynomialBatch::prove_openings +
// fri::prover::fri_proof (plonky2 v0.2.0, un-vendored: /root/reference/Cargo.lock:4848-4905; reached from
// /root/reference/circuits/header_range.rs:167 `circuit.prove(&input)`; transcript order per SURVEY.md A.6-A.8).
// Everything heavy is a kernel launch on the context's stream; the host keeps only the Fiat-Shamir
// transcript (a few hundred field elements per proof) and the proof assembly.
#pragma once
#include <chrono>
#include "batch.hip.h"
#include "challenger.h"
#include "plonk_kernels.hip.h"
#include "desc_check.h"

struct vx_circuit {
  vx_ctx* ctx = nullptr;
  int degree_bits = 0, num_wires = 0, nr = 0, nch = 0, rate_bits = 0, cap_height = 0, pow_bits = 0, num_queries = 0, qdf = 0;
  int num_selectors = 0, num_constants = 0;
  std::vector<GateDev> gates;
  std::vector<u64> k_is_host;
  std::vector<uint32_t> pi_rows, pi_cols;
  std::vector<int> arity_bits;
  std::vector<int> prog_off;   // per gate: word offset into `programs`, -1 for native gates
  u64* programs = nullptr;     // device copy of the constraint programs
  std::vector<hipFunction_t> jit_fns;  // native kernels (jit.hip.h), one per GROUP of program gates; empty -> interpreter
  std::vector<std::vector<size_t>> jit_groups;  // jit_fns[i] evaluates the gates jit_gates[jit_groups[i][..]]
  std::vector<int> jit_gates;         // gate index of each block of that kernel, in order
  int jit_fused_waves = 0;               // wavefronts per workgroup of that kernel (a function of the gate set: jit_fused_plan)
  hipFunction_t jit_fused_fn = nullptr;  // ALL program gates in one kernel that stages the wires through LDS once (jit.hip.h, round 6); then jit_fns is empty
  std::vector<std::string> jit_stage;  // profile stage of each launch: "qgate_<gate index>[+<gate index>...]" (vx_prof_get; the caller knows its gate order)
  std::vector<uint64_t> programs_host;  // host copy of the programs (vx_verify evaluates gates at zeta on the host)
  std::vector<u64> cs_cap_host;         // constants_sigmas cap (verifier data)
  std::string jit_note;               // why the program gates stayed on the interpreter (diagnostics)
  vx_batch* cs = nullptr;  // constants_sigmas commitment (resident across proofs)
  u64* sigmas = nullptr;   // [nr][n] sigma VALUES on H (natural order) for the permutation argument
  u64* k_is = nullptr;     // device copy
  u64* l0_lde = nullptr;   // [N] L_0 on the LDE rows (plonk_kernels.hip.h l0_table_kernel)
  vxh::Hash4 digest;
  // lookup argument (host copies of CommonCircuitData::luts / ProverOnlyCircuitData::lookup_rows); num_luts = 0: none
  int num_luts = 0, num_lookup_selectors = 0;
  std::vector<int32_t> lut_lens, lookup_rows;
  std::vector<uint16_t> lut_inputs, lut_outputs;
  int num_sldc() const { return (nr / 2 + (qdf - 1) - 1) / (qdf - 1); }
  int nlp() const { return num_luts > 0 ? 1 + num_sldc() : 0; }   // lookup polynomials per challenge: RE + partial Sum/LDCs
  int const_base() const { return num_selectors + num_lookup_selectors; }
  int npp() const { return (nr + qdf - 1) / qdf - 1; }
  size_t n() const { return (size_t)1 << degree_bits; }
};

// VerifierOnlyCircuitData::circuit_digest.  The caller's value when it passes one (VX_DESC_HAS_CIRCUIT_DIGEST) — the Rust
// side holds it, and then no recalled convention of this library is involved; otherwise THE one place the rule is
// restated (plonk/circuit_builder.rs::build, recalled — two independent reviews recalled the same form, round 4):
//   hash_no_pad(constants_sigmas_cap.flatten() || hash_pad(domain_separator).elements || [degree_bits])
// with plonky2x's default EMPTY domain separator; hash_pad = pad10*1 to a multiple of the rate, so
// hash_pad([]) = hash_no_pad([1,0,0,0,0,0,0,1]).  `cap_and_degree` = the cap's elements followed by degree_bits.
static vxh::Hash4 circuit_digest_of(const vx_circuit_desc* d, const u64* cap_and_degree, size_t len) {
  if (d->override_flags & VX_DESC_HAS_CIRCUIT_DIGEST) {
    vxh::Hash4 h;
    for (int i = 0; i < 4; ++i) h.e[i] = vxh::canon(d->circuit_digest[i]);
    return h;
  }
  const u64 padded_empty_separator[8] = {1, 0, 0, 0, 0, 0, 0, 1};
  const vxh::Hash4 sep = vxh::hash_no_pad(padded_empty_separator, 8);
  std::vector<u64> parts(cap_and_degree, cap_and_degree + (len - 1));
  for (int i = 0; i < 4; ++i) parts.push_back(sep.e[i]);
  parts.push_back(cap_and_degree[len - 1]);
  return vxh::hash_no_pad(parts.data(), parts.size());
}

static void circuit_free(vx_circuit* k) {
  if (!k) return;
  hipSetDevice(k->ctx->device);
  hipStreamSynchronize(k->ctx->stream);
  if (k->cs) {
    k->ctx->pool_free(k->cs->coeffs);
    k->ctx->pool_free(k->cs->lde);
    k->ctx->pool_free(k->cs->tree);
    delete k->cs;
  }
  hipFree(k->sigmas);
  hipFree(k->k_is);
  hipFree(k->l0_lde);
  hipFree(k->programs);
  delete k;
}

static int circuit_create(vx_ctx* c, const vx_circuit_desc* d, vx_circuit** out) {
  DescResolved res;
  {
    const std::string why = desc_check(d, /*need_preprocessed=*/true, &res);
    if (!why.empty()) return vx_fail(VX_E_INVALID, "%s", why.c_str());
  }
  // every failure below goes through circuit_free(k) (and frees the staging buffer): nothing leaks on an early return
  vx_circuit* k = new vx_circuit();
  u64* staging = nullptr;
  struct Guard {
    vx_circuit*& k;
    u64*& staging;
    ~Guard() {
      if (staging) hipFree(staging);
      if (k) circuit_free(k);
    }
  } guard{k, staging};
  k->ctx = c;
  k->degree_bits = d->degree_bits;
  k->num_wires = d->num_wires;
  k->nr = d->num_routed_wires;
  k->nch = d->num_challenges;
  k->rate_bits = d->rate_bits;
  k->cap_height = d->cap_height;
  k->pow_bits = d->pow_bits;
  k->num_queries = d->num_query_rounds;
  k->qdf = d->quotient_degree_factor;
  k->num_selectors = d->num_selectors;
  k->num_constants = d->num_constants;
  k->arity_bits = res.arity_bits;  // the caller's FriParams::reduction_arity_bits, or ConstantArityBits(4, 5)
  if (d->num_luts > 0) {
    k->num_luts = d->num_luts;
    k->num_lookup_selectors = d->num_lookup_selectors;
    size_t total = 0;
    for (int t = 0; t < d->num_luts; ++t) total += (size_t)d->lut_lens[t];
    k->lut_lens.assign(d->lut_lens, d->lut_lens + d->num_luts);
    k->lookup_rows.assign(d->lookup_rows, d->lookup_rows + 3 * d->num_luts);
    k->lut_inputs.assign(d->lut_inputs, d->lut_inputs + total);
    k->lut_outputs.assign(d->lut_outputs, d->lut_outputs + total);
  }
  for (int g = 0; g < d->num_gates; ++g)
    k->gates.push_back(GateDev{d->gate_types[g], d->gate_params[g], d->selector_indices[g], d->group_starts[g], d->group_ends[g]});
  {
    int nprog = 0;
    for (int g = 0; g < d->num_gates; ++g) {
      k->prog_off.push_back(d->gate_types[g] == VX_GATE_PROGRAM ? d->program_offsets[g] : -1);
      nprog += d->gate_types[g] == VX_GATE_PROGRAM;
    }
    if (nprog) {
      k->programs_host.assign(d->programs, d->programs + d->programs_len);
      if (hipMalloc(&k->programs, (size_t)d->programs_len * 8) != hipSuccess) return vx_fail(VX_E_NOMEM, "circuit: out of device memory");
      HIPCHK(hipMemcpy(k->programs, d->programs, (size_t)d->programs_len * 8, hipMemcpyHostToDevice));
    }
    // compile the programs to native code (jit.hip.h): one kernel for the whole gate set; on any failure the gates
    // stay on the interpreter
    if (nprog) {
      const int nterms = d->num_challenges * (1 + (d->num_routed_wires + d->quotient_degree_factor - 1) / d->quotient_degree_factor + 1);
      std::vector<const uint64_t*> progs;
      std::string why;
      for (int g = 0; g < d->num_gates && why.empty(); ++g) {
        if (k->prog_off[g] < 0) continue;
        const uint64_t* prog = d->programs + k->prog_off[g];
        int pushes = 0;
        for (int pc = 0; (prog[pc] & 0xFF) != VX_OP_END; ++pc) {
          if ((prog[pc] & 0xFF) == VX_OP_PUSH) ++pushes;
          if ((prog[pc] & 0xFF) == VX_OP_LDI) ++pc;
        }
        if (nterms + pushes > VX_ALPHA_POWS) why = "gate " + std::to_string(g) + " has more constraints than the alpha-power table holds";
        progs.push_back(prog);
        k->jit_gates.push_back(g);
      }
      if (why.empty() && jit_fused_applicable(progs, d->num_challenges)) {
        std::string fwhy;   // a failure here is not fatal: the gates fall back to one kernel each
        k->jit_fused_fn = jit_get_fused_gates(progs, d->num_challenges, c->device, &fwhy);
        if (k->jit_fused_fn) k->jit_fused_waves = jit_fused_plan(progs, d->num_challenges).waves;
      }
      if (why.empty() && !k->jit_fused_fn) k->jit_groups = jit_gate_groups(progs);
      for (size_t gi = 0; gi < k->jit_groups.size() && why.empty(); ++gi) {
        std::vector<const uint64_t*> sub;
        for (size_t q : k->jit_groups[gi]) sub.push_back(progs[q]);
        hipFunction_t fn = jit_get_gates(sub, d->num_challenges, c->device, &why);
        if (fn) k->jit_fns.push_back(fn);
        std::string nm = "qgate";
        for (size_t q : k->jit_groups[gi]) nm += (nm.size() == 5 ? "_" : "+") + std::to_string(k->jit_gates[q]);
        k->jit_stage.push_back(nm);
      }
      if (k->jit_fused_fn) {
        // nothing else to compile
      } else if (!why.empty() || k->jit_fns.size() != k->jit_groups.size() || k->jit_fns.empty()) {   // all or nothing: the interpreter takes every program gate
        k->jit_note = why;
        k->jit_gates.clear();
        k->jit_fns.clear();
        k->jit_groups.clear();
        k->jit_stage.clear();
      }
    }
  }
  k->k_is_host.assign(d->k_is, d->k_is + d->num_routed_wires);
  for (auto& v : k->k_is_host) v = vxh::canon(v);
  k->pi_rows.assign(d->pi_rows, d->pi_rows + d->num_public_inputs);
  k->pi_cols.assign(d->pi_cols, d->pi_cols + d->num_public_inputs);
  const size_t n = k->n();
  const size_t m = (size_t)k->num_constants + k->nr;
  int rc = batch_alloc(c, k->degree_bits, m, k->rate_bits, k->cap_height, &k->cs);
  if (rc) return rc;
  if (hipMalloc(&staging, m * n * 8) != hipSuccess || hipMalloc(&k->sigmas, (size_t)k->nr * n * 8) != hipSuccess ||
      hipMalloc(&k->k_is, k->nr * 8) != hipSuccess)
    return vx_fail(VX_E_NOMEM, "circuit: out of device memory");
  HIPCHK(hipMemcpyAsync(staging, d->constants_sigmas, m * n * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(k->k_is, k->k_is_host.data(), k->nr * 8, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(canon_kernel, dim3((unsigned)((m * n + 255) / 256)), dim3(256), 0, c->stream, staging, m * n);
  HIPCHK(hipMemcpyAsync(k->sigmas, staging + (size_t)k->num_constants * n, (size_t)k->nr * n * 8, hipMemcpyDeviceToDevice, c->stream));
  rc = batch_commit_device(c, k->cs, staging, n, false);
  {
    // L_0 on the LDE domain (per circuit size; read by quotient_kernel<0> of every proof)
    const int rb = k->rate_bits, rate = 1 << rb;
    const size_t N = n << rb;
    if (hipMalloc(&k->l0_lde, N * 8) != hipSuccess) return vx_fail(VX_E_NOMEM, "circuit: out of device memory (L_0 table)");
    L0Params lp;
    memset(&lp, 0, sizeof lp);
    lp.root_lo = c->root_lo, lp.root_hi = c->root_hi;
    lp.N = N, lp.log_n = k->degree_bits, lp.rate_bits = rb;
    const u64 shift_n = vxh::pow(7, n), g_rate = vxh::root_of_unity(rb);
    u64 pw = 1;
    for (int r = 0; r < rate; ++r) {
      lp.zh[r] = vxh::sub(vxh::mul(shift_n, pw), 1);
      pw = vxh::mul(pw, g_rate);
    }
    lp.n_field = (u64)n % vxh::P;
    lp.out = k->l0_lde;
    hipLaunchKernelGGL(l0_table_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, lp);
  }
  hipError_t e = hipStreamSynchronize(c->stream);
  if (rc == VX_OK && e != hipSuccess) rc = vx_fail(VX_E_HIP, "circuit: %s", hipGetErrorString(e));
  if (rc) return rc;
  std::vector<u64> pre(((size_t)4 << k->cap_height) + 1);
  HIPCHK(hipMemcpy(pre.data(), k->cs->tree + k->cs->cap_off * 4, (size_t)32 << k->cap_height, hipMemcpyDeviceToHost));
  pre.back() = (u64)k->degree_bits;
  k->cs_cap_host.assign(pre.begin(), pre.end() - 1);
  k->digest = circuit_digest_of(d, pre.data(), pre.size());
  *out = k;
  k = nullptr;  // ownership passes to the caller
  return VX_OK;
}

static void batch_release(vx_ctx* c, vx_batch* b) {
  if (!b) return;
  c->pool_free(b->coeffs);
  c->pool_free(b->lde);
  c->pool_free(b->tree);
  delete b;
}

struct ByteSink {
  std::vector<uint8_t> b;
  void f(u64 v) {
    for (int i = 0; i < 8; ++i) b.push_back((uint8_t)(v >> (8 * i)));
  }
  void words(const u64* p, size_t n) {
    for (size_t i = 0; i < n; ++i) f(p[i]);
  }
  void u8(uint8_t v) { b.push_back(v); }
};

static size_t proof_size_bound(const vx_circuit* k) {
  size_t cap = (size_t)32 << k->cap_height;
  size_t depth0 = k->degree_bits + k->rate_bits - k->cap_height;
  size_t widths = (size_t)k->num_constants + k->nr + k->num_wires + (size_t)k->nch * (1 + k->npp() + k->nlp()) + (size_t)k->nch * k->qdf;
  size_t open = 16 * (widths + k->nch * (1 + k->nlp()));
  size_t per_query = 8 * widths + 4 * (1 + 32 * depth0) + k->arity_bits.size() * (16 * 16 + 1 + 32 * depth0);
  size_t final_len = k->n();  // final polynomial: n >> (sum of the reduction arities) coefficients in F_p^2 (caller-supplied arities may leave it long)
  for (int ab : k->arity_bits) final_len >>= ab;
  return (3 + k->arity_bits.size()) * cap + open + k->num_queries * per_query + 16 * final_len + 8 + 8 * k->pi_rows.size() + 4096;
}

// One device scratch allocation that is returned to the pool when the proof ends.
struct Scratch {
  vx_ctx* c;
  std::vector<void*> ptrs;
  explicit Scratch(vx_ctx* ctx) : c(ctx) {}
  ~Scratch() {
    for (void* p : ptrs) c->pool_free(p);
  }
  u64* get(size_t words) {
    void* p = nullptr;
    if (c->pool_alloc(&p, words * 8) != hipSuccess) return nullptr;
    ptrs.push_back(p);
    return (u64*)p;
  }
};

// One proof split across `world` ranks by LDE coset (include/vxprover.h vx_prove_sharded).  world == 1: plain vx_prove.
struct Shard {
  int rank = 0, world = 1, lg = 0;
  vx_allgather_fn fn = nullptr;
  void* user = nullptr;
};
// In-place all-gather of a device buffer of world * bytes_per_rank bytes whose slot `rank` is filled.
static int shard_allgather(vx_ctx* c, const Shard& sh, void* dev, size_t bytes_per_rank, const char* what) {
  if (sh.world == 1) return VX_OK;
  HIPCHK(hipStreamSynchronize(c->stream));
  static const bool trace = getenv("VX_TRACE_EXCHANGES") != nullptr;  // one line per exchange: which rank waits for what
  if (trace) fprintf(stderr, "[vx rank %d/%d] all-gather: %s, %zu bytes per rank\n", sh.rank, sh.world, what, bytes_per_rank);
  const auto t0 = std::chrono::steady_clock::now();
  int rc = sh.fn(sh.user, dev, bytes_per_rank);
  // host wall time of the exchange INCLUDING the wait for the slowest rank — its own stage ("exchange_host_wait"), so that
  // the HIP-event stages around it are compute only (query_gather looked 60x slower sharded)
  c->prof_add_host("exchange_host_wait", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
                   (double)bytes_per_rank * (sh.world - 1));
  if (trace) fprintf(stderr, "[vx rank %d/%d] all-gather done: %s (rc %d)\n", sh.rank, sh.world, what, rc);
  if (rc) return vx_fail(VX_E_COMM, "prove: all-gather of %s failed on rank %d (callback returned %d)", what, sh.rank, rc);
  return VX_OK;
}
// The 2^cap_height cap of a (possibly sharded) tree, on the host: the ranks' local caps concatenated in rank order.
static int gather_cap(vx_ctx* c, const Shard& sh, Scratch& S, const u64* local_cap_dev, size_t local_words, std::vector<u64>& cap) {
  cap.resize(local_words * sh.world);
  const u64* src = local_cap_dev;
  if (sh.world > 1) {
    u64* x = S.get(cap.size());
    if (!x) return vx_fail(VX_E_NOMEM, "prove: out of device memory (cap exchange)");
    HIPCHK(hipMemcpyAsync(x + local_words * sh.rank, local_cap_dev, local_words * 8, hipMemcpyDeviceToDevice, c->stream));
    VXCHK(shard_allgather(c, sh, x, local_words * 8, "Merkle cap"));
    src = x;
  }
  HIPCHK(hipMemcpyAsync(cap.data(), src, cap.size() * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return VX_OK;
}

// fri::oracle::PolynomialBatch::prove_openings + fri::prover::fri_proof for TWO opening points (plonky2's and starky's
// FriInstanceInfo shape): batch 0 = EVERY column of every oracle, opened at z0; batch 1 = the listed column ranges, opened
// at z1 (plonk: Z polynomials [+ lookup polynomials] at g*zeta; a STARK: the whole trace at g*zeta).  The caller has
// observed the openings; this draws alpha, combines, commits the FRI layers, grinds, answers the queries.  Sharded like
// every LDE (Shard).  Result: the pieces of FriProof, serialised by write_fri_proof.
struct FriProverParams {
  int degree_bits = 0, rate_bits = 0, cap_height = 0, pow_bits = 0, num_queries = 0;
  std::vector<int> arity_bits;
};
struct FriRange {
  int oracle;
  size_t col0, ncols;
};
struct FriParts {
  std::vector<std::vector<u64>> commit_caps;
  std::vector<vxh::Ext> final_poly;
  u64 pow_witness = 0;
  std::vector<std::vector<u64>> init_out, step_out;  // query openings: [oracle] / [round], row-major per (slot, query)
  std::vector<int> owner;                              // rank that owns query q's leaf in the sharded trees
  std::vector<size_t> flen;
  int depth0 = 0;
};
static int fri_prove_openings(vx_ctx* c, const FriProverParams& fpp, const std::vector<vx_batch*>& oracles, const std::vector<FriRange>& batch0_ranges,
                              const std::vector<FriRange>& batch1_ranges,
                              vxh::Ext zeta, vxh::Ext gzeta, const std::vector<vxh::Ext>& batch0, const std::vector<vxh::Ext>& batch1,
                              vxh::Challenger& ch, const u64* pow_hint, const Shard& sh, Scratch& S, FriParts& out) {
  using namespace vxh;
  const int lg = fpp.degree_bits, rb = fpp.rate_bits, LG = lg + rb, rate = 1 << rb;
  const size_t n = (size_t)1 << lg, N = (size_t)1 << LG;
  const size_t cap_words = (size_t)4 << fpp.cap_height;
  const size_t Nl = N >> sh.lg, row_base = Nl * (size_t)sh.rank;
  const int zc = rate >> sh.lg, z0 = zc * sh.rank;
  Ext alpha = ch.get_extension_challenge();
  const size_t nb0 = batch0.size(), nb1 = batch1.size();
  std::vector<u64> apows(2 * nb0);
  {
    Ext a{1, 0};
    for (size_t j = 0; j < nb0; ++j) {
      apows[2 * j] = a.a;
      apows[2 * j + 1] = a.b;
      a = emul(a, alpha);
    }
  }
  // F_b(z_b) = sum_j alpha^j opening_j   (= the verifier's PrecomputedReducedOpenings)
  Ext y0{0, 0}, y1{0, 0};
  for (size_t j = nb0; j-- > 0;) y0 = eadd(emul(y0, alpha), batch0[j]);
  for (size_t j = nb1; j-- > 0;) y1 = eadd(emul(y1, alpha), batch1[j]);
  u64* fcoef = S.get(4 * n);
  u64* flde = S.get(4 * Nl);
  u64* d_apows = S.get(2 * nb0);
  if (!fcoef || !flde || !d_apows) return vx_fail(VX_E_NOMEM, "prove: out of device memory (opening proof)");
  HIPCHK(hipMemcpyAsync(d_apows, apows.data(), apows.size() * 8, hipMemcpyHostToDevice, c->stream));
  {
    ReduceParams rp;
    memset(&rp, 0, sizeof rp);
    if (batch0_ranges.size() > REDUCE_MAX_GROUPS) return vx_fail(VX_E_INVALID, "fri: too many column ranges in the first opening batch");
    rp.ngroups = (int)batch0_ranges.size();
    size_t total = 0;
    for (size_t g = 0; g < batch0_ranges.size(); ++g) {
      rp.cols[g] = oracles[batch0_ranges[g].oracle]->coeffs + batch0_ranges[g].col0 * n;
      rp.ncols[g] = (int)batch0_ranges[g].ncols;
      total += batch0_ranges[g].ncols;
    }
    if (total != batch0.size()) return vx_fail(VX_E_INVALID, "fri: opening batch 0 has %zu values for %zu polynomials", batch0.size(), total);
    rp.alpha_pows = d_apows;
    rp.n = n;
    rp.out = fcoef;
    ProfScope ps(c, "reduce_polys", 8.0 * n * total);
    hipLaunchKernelGGL(reduce_polys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, rp);
    if (batch1_ranges.size() > REDUCE_MAX_GROUPS) return vx_fail(VX_E_INVALID, "fri: too many column ranges in the second opening batch");
    rp.ngroups = (int)batch1_ranges.size();
    for (size_t g = 0; g < batch1_ranges.size(); ++g) {
      rp.cols[g] = oracles[batch1_ranges[g].oracle]->coeffs + batch1_ranges[g].col0 * n;
      rp.ncols[g] = (int)batch1_ranges[g].ncols;
    }
    rp.out = fcoef + 2 * n;
    hipLaunchKernelGGL(reduce_polys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, rp);
    HIPCHK(hipGetLastError());
  }
  // LDE of (F0.a, F0.b, F1.a, F1.b): an F_p^2 NTT with base-field roots is two F_p NTTs
  {
    std::vector<u64> shifts(zc);
    u64 wN = root_of_unity(LG);
    for (int z = 0; z < zc; ++z) shifts[z] = mul(7, pow(wN, reverse_bits((size_t)(z0 + z), rb)));
    int bits = lg / 2;
    u64* tab = nullptr;
    VXCHK(get_scale_tables(c, lg, bits, shifts, 1, &tab));
    VXCHK(run_ntt(c, fcoef, flde, n, Nl, 0, n, lg, 4, zc, false, true, tab, bits, 1, "fri_lde", 4.0 * 8.0 * ((double)n + Nl)));
  }
  // FRI value arrays per round (interleaved ext, bit-reversed order) and their trees
  const size_t R = fpp.arity_bits.size();
  std::vector<u64*> fvals(R + 1, nullptr), ftrees(R, nullptr);
  std::vector<size_t> flen(R + 1), fcapoff(R, 0);
  flen[0] = N;
  for (size_t r = 0; r < R; ++r) flen[r + 1] = flen[r] >> fpp.arity_bits[r];
  for (size_t r = 0; r <= R; ++r) {
    fvals[r] = S.get(2 * flen[r]);
    if (!fvals[r]) return vx_fail(VX_E_NOMEM, "prove: out of device memory (FRI)");
  }
  {
    CombineParams cp;
    memset(&cp, 0, sizeof cp);
    cp.fl = flde;
    cp.root_lo = c->root_lo;
    cp.root_hi = c->root_hi;
    cp.rows = Nl;
    cp.row_base = row_base;
    cp.log_N = LG;
    cp.y0[0] = y0.a, cp.y0[1] = y0.b, cp.y1[0] = y1.a, cp.y1[1] = y1.b;
    cp.z0[0] = zeta.a, cp.z0[1] = zeta.b, cp.z1[0] = gzeta.a, cp.z1[1] = gzeta.b;
    Ext sh = epow(alpha, nb1);  // alpha.shift_poly: *= alpha^|batch 1|
    cp.shift0[0] = sh.a, cp.shift0[1] = sh.b;
    cp.out = fvals[0] + 2 * row_base;
    ProfScope ps(c, "fri_combine", 48.0 * Nl);
    hipLaunchKernelGGL(fri_combine_kernel, dim3((unsigned)((Nl + 255) / 256)), dim3(256), 0, c->stream, cp);
    HIPCHK(hipGetLastError());
  }
  if (R == 0) VXCHK(shard_allgather(c, sh, fvals[0], 16 * Nl, "FRI values"));
  // ---- FRI commit phase ----
  std::vector<std::vector<u64>>& commit_caps = out.commit_caps;
  commit_caps.clear();
  {
    u64 shift = 7;
    for (size_t r = 0; r < R; ++r) {
      const int ab = fpp.arity_bits[r];
      // the first layer is sharded like every LDE (rows [row_base, row_base + Nl)); later layers are replicated
      const Shard shr = r == 0 ? sh : Shard();
      const size_t M = flen[r] >> shr.lg, leaves = M >> ab, in_base = r == 0 ? row_base : 0;
      const int width = 2 << ab, ch_l = fpp.cap_height - shr.lg;
      size_t nd = merkle_tree_digest_count(leaves, ch_l);
      ftrees[r] = S.get(nd * 4);
      if (!ftrees[r]) return vx_fail(VX_E_NOMEM, "prove: out of device memory (FRI trees)");
      {
        ProfScope ps(c, "fri_hash_leaves", 16.0 * M);
        if (leaves <= COOP_MAX_LEAVES)
          hipLaunchKernelGGL(hash_leaves_rowmajor_coop_kernel, dim3((unsigned)((leaves * 16 + HASH_THREADS - 1) / HASH_THREADS)),
                             dim3(HASH_THREADS), 0, c->stream, fvals[r] + 2 * in_base, leaves, width, ftrees[r]);
        else
          hipLaunchKernelGGL(hash_leaves_rowmajor_kernel, dim3((unsigned)((leaves + HASH_THREADS - 1) / HASH_THREADS)),
                             dim3(HASH_THREADS), 0, c->stream, fvals[r] + 2 * in_base, leaves, width, ftrees[r]);
        HIPCHK(hipGetLastError());
      }
      VXCHK(build_merkle_levels(c, ftrees[r], leaves, ch_l, &fcapoff[r]));
      std::vector<u64> cap;
      VXCHK(gather_cap(c, shr, S, ftrees[r] + fcapoff[r] * 4, (size_t)4 << ch_l, cap));
      ch.observe_elements(cap.data(), cap_words);
      commit_caps.push_back(cap);
      Ext beta = ch.get_extension_challenge();
      FoldParams fp;
      memset(&fp, 0, sizeof fp);
      fp.in = fvals[r] + 2 * in_base;
      fp.out = fvals[r + 1] + 2 * (in_base >> ab);
      fp.k_base = in_base >> ab;
      fp.root_lo = c->root_lo;
      fp.root_hi = c->root_hi;
      fp.M = M;
      fp.log_M = LG;
      for (size_t q = 0; q < r; ++q) fp.log_M -= fpp.arity_bits[q];
      fp.arity_bits = ab;
      fp.beta[0] = beta.a, fp.beta[1] = beta.b;
      fp.shift_inv = inv(shift);
      u64 wa_inv = inv(root_of_unity(ab)), pw = 1;
      for (int q = 0; q < (1 << ab); ++q) {
        fp.w_inv_pows[q] = pw;
        pw = mul(pw, wa_inv);
      }
      fp.arity_inv = inv((u64)1 << ab);
      {
        ProfScope ps(c, "fri_fold", 16.0 * M);
        HIPCHK(launch_fri_fold(fp, leaves, c->stream));
      }
      VXCHK(shard_allgather(c, shr, fvals[r + 1], 16 * leaves, "folded FRI layer"));
      shift = pow(shift, (u64)1 << ab);
    }
    // final polynomial: the last value array (bit-reversed, on the coset shift*H) -> coefficients, on the host
    const size_t Mf = flen[R];
    const int lMf = [&] { int l = 0; while (((size_t)1 << l) < Mf) ++l; return l; }();
    std::vector<u64> hv(2 * Mf);
    HIPCHK(hipMemcpyAsync(hv.data(), fvals[R], hv.size() * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (u64& v : hv) v = canon(v);
    std::vector<Ext>& coeffs = out.final_poly;
    coeffs.assign(Mf, Ext{0, 0});
    {
      // coefficient j = shift^-j / Mf * sum_k v_k w^(-jk), v_k at natural index k = rev(position): the array IS the
      // bit-reversed input of a decimation-in-time inverse transform, which leaves the coefficients in natural order.
      // O(Mf log Mf) on the host: Mf is 2^8 with the standard arities but anything up to the whole LDE with a
      // caller-supplied list (a quadratic loop here once took minutes for a single arity-2 reduction at n = 2^17).
      const u64 w_inv = inv(root_of_unity(lMf)), s_inv = inv(shift), m_inv = inv((u64)Mf % P);
      std::vector<u64> wp(std::max<size_t>(Mf / 2, 1));
      wp[0] = 1;
      for (size_t i = 1; i < wp.size(); ++i) wp[i] = mul(wp[i - 1], w_inv);
      for (size_t len = 2; len <= Mf; len <<= 1) {
        const size_t half = len >> 1, stride = Mf / len;
        for (size_t i = 0; i < Mf; i += len)
          for (size_t j = 0; j < half; ++j) {
            const u64 w = wp[j * stride];
            for (int e = 0; e < 2; ++e) {
              const u64 u = hv[2 * (i + j) + e], v = mul(hv[2 * (i + j + half) + e], w);
              hv[2 * (i + j) + e] = add(u, v);
              hv[2 * (i + j + half) + e] = sub(u, v);
            }
          }
      }
      u64 sj = m_inv;
      for (size_t j = 0; j < Mf; ++j) {
        coeffs[j] = Ext{mul(hv[2 * j], sj), mul(hv[2 * j + 1], sj)};
        sj = mul(sj, s_inv);
      }
    }
    const size_t keep = Mf >> rb;
    for (size_t j = keep; j < Mf; ++j)
      if ((coeffs[j].a || coeffs[j].b) && !c->rehearsal) return vx_fail(VX_E_PROOF, "FRI final polynomial has non-zero high coefficients (witness does not satisfy the circuit?)");
    coeffs.resize(keep);
    for (Ext e : coeffs) ch.observe_ext(e);
    // ---- proof of work ----
    u64& pow_witness = out.pow_witness;
    pow_witness = 0;
    {
      auto check = [&](u64 cand) {
        Challenger c2 = ch;
        c2.observe_element(cand);
        return (c2.get_challenge() >> (64 - fpp.pow_bits)) == 0;
      };
      if (pow_hint) {
        pow_witness = canon(*pow_hint);
        if (fpp.pow_bits > 0 && !check(pow_witness)) return vx_fail(VX_E_INVALID, "pow_witness hint does not satisfy the proof-of-work condition");
      } else if (fpp.pow_bits > 0) {
        PowParams pp;
        memset(&pp, 0, sizeof pp);
        for (int q = 0; q < 12; ++q) pp.state[q] = ch.sponge[q];
        for (size_t q = 0; q < ch.input.size(); ++q) pp.state[q] = ch.input[q];
        pp.pos = (int)ch.input.size();
        pp.pow_bits = fpp.pow_bits;
        unsigned long long* d_res = (unsigned long long*)S.get(1);
        if (!d_res) return vx_fail(VX_E_NOMEM, "prove: out of device memory (pow)");
        pp.result = d_res;
        // Candidates are searched in increasing ranges and the smallest hit of a range wins, so the witness does not
        // depend on the range sizes.  The first range is sized to succeed with probability ~1 - e^-2 at the cost of a
        // couple of waves per SIMD; later ranges grow to keep the launch count logarithmic.
        u64 batch = (u64)2 << std::min(fpp.pow_bits, 21);
        unsigned long long res = ~0ull;
        ProfScope ps(c, "pow_grind");
        for (u64 base = 0; res == ~0ull; base += batch, batch = std::min(batch * 4, (u64)1 << 24)) {
          HIPCHK(hipMemsetAsync(d_res, 0xFF, 8, c->stream));
          pp.base = base;
          hipLaunchKernelGGL(pow_grind_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, c->stream, pp);
          HIPCHK(hipMemcpyAsync(&res, d_res, 8, hipMemcpyDeviceToHost, c->stream));
          HIPCHK(hipStreamSynchronize(c->stream));
          if (base > ((u64)1 << 44)) return vx_fail(VX_E_PROOF, "Proof of work failed. This is highly unlikely!");
        }
        pow_witness = res;
        if (!check(pow_witness)) return vx_fail(VX_E_PROOF, "internal error: GPU proof-of-work witness rejected by the host transcript");
      }
      ch.observe_element(pow_witness);
      (void)ch.get_challenge();  // pow_response
    }
    // ---- query rounds ----
    const int nq = fpp.num_queries;
    std::vector<u64> x_indices(nq);
    for (int q = 0; q < nq; ++q) x_indices[q] = ch.get_challenge() % (u64)N;
    const int depth0 = LG - fpp.cap_height;
    // A sharded tree (wires / Z / quotient oracles, first FRI layer) is opened by the rank that owns the leaf — the
    // top shard_lg bits of the index — and the rows are all-gathered: slot `owner[q]` holds query q's real opening.
    out.init_out.assign(oracles.size(), std::vector<u64>());
    std::vector<std::vector<u64>>& init_out = out.init_out;
    out.step_out.assign(R, std::vector<u64>());
    std::vector<std::vector<u64>>& step_out = out.step_out;
    out.owner.assign(nq, 0);
    std::vector<int>& owner = out.owner;
    {
      ProfScope ps(c, "query_gather");
      u64* d_idx = S.get((size_t)nq * (R + 3));
      if (!d_idx) return vx_fail(VX_E_NOMEM, "prove: out of device memory (queries)");
      std::vector<u64> idx_host((size_t)nq * (R + 3));
      u64* loc_x = &idx_host[(R + 1) * nq];   // local leaf index in a sharded oracle (0 when another rank owns it)
      u64* loc_f0 = &idx_host[(R + 2) * nq];  // local leaf index in the first FRI layer's tree
      for (int q = 0; q < nq; ++q) {
        u64 xi = x_indices[q];
        idx_host[q] = xi;
        owner[q] = (int)(xi >> (LG - sh.lg));
        const bool mine = owner[q] == sh.rank;
        loc_x[q] = mine ? xi - row_base : 0;
        loc_f0[q] = mine && R > 0 ? (xi >> fpp.arity_bits[0]) - (row_base >> fpp.arity_bits[0]) : 0;
        for (size_t r = 0; r < R; ++r) {
          xi >>= fpp.arity_bits[r];
          idx_host[(r + 1) * nq + q] = xi;
        }
      }
      HIPCHK(hipMemcpyAsync(d_idx, idx_host.data(), idx_host.size() * 8, hipMemcpyHostToDevice, c->stream));
      // Every sharded tree (the oracles' and the first FRI layer's) is opened by the rank that owns the leaf; ALL those rows
      // travel in ONE all-gather: slot s of `d_sh` = [oracle 0 rows | oracle 1 rows | .. | FRI layer 0 rows] of rank s
      // (round 2 did one exchange per tree: 5 barriers where 1 does).  Unsharded trees are gathered into `d_un`.
      struct Piece { size_t rowlen, off; bool sharded; std::vector<u64>* host; };
      std::vector<Piece> pieces;
      size_t per_rank = 0, unsharded = 0;
      for (size_t o = 0; o < oracles.size(); ++o) {
        const bool sharded = oracles[o]->shard_lg > 0;
        const size_t rowlen = oracles[o]->ncols + 4 * (size_t)depth0;
        pieces.push_back(Piece{rowlen, sharded ? per_rank : unsharded, sharded, &init_out[o]});
        (sharded ? per_rank : unsharded) += rowlen * nq;
      }
      std::vector<int> depth_r(R);
      for (size_t r = 0; r < R; ++r) {
        const int ab = fpp.arity_bits[r];
        const bool sharded = r == 0 && sh.world > 1;
        const size_t leaves_all = flen[r] >> ab;
        int depth = 0;
        while (((size_t)1 << (depth + fpp.cap_height)) < leaves_all) ++depth;
        depth_r[r] = depth;
        const size_t rowlen = ((size_t)2 << ab) + 4 * (size_t)depth;
        pieces.push_back(Piece{rowlen, sharded ? per_rank : unsharded, sharded, &step_out[r]});
        (sharded ? per_rank : unsharded) += rowlen * nq;
      }
      u64* d_sh = per_rank ? S.get(per_rank * sh.world) : nullptr;
      u64* d_un = unsharded ? S.get(unsharded) : nullptr;
      if ((per_rank && !d_sh) || (unsharded && !d_un)) return vx_fail(VX_E_NOMEM, "prove: out of device memory (queries)");
      u64* my_slot = d_sh ? d_sh + per_rank * sh.rank : nullptr;
      for (size_t o = 0; o < oracles.size(); ++o) {
        const Piece& pc = pieces[o];
        const size_t rows_o = oracles[o]->rows();
        hipLaunchKernelGGL(gather_open_kernel, dim3(nq), dim3(128), 0, c->stream, oracles[o]->lde, rows_o, (int)oracles[o]->ncols, 1,
                           oracles[o]->tree, rows_o, depth0, pc.sharded ? d_idx + (R + 1) * nq : d_idx, (pc.sharded ? my_slot : d_un) + pc.off);
      }
      for (size_t r = 0; r < R; ++r) {
        const Piece& pc = pieces[oracles.size() + r];
        const int ab = fpp.arity_bits[r];
        const size_t leaves_all = flen[r] >> ab, leaves = pc.sharded ? leaves_all >> sh.lg : leaves_all;
        hipLaunchKernelGGL(gather_open_kernel, dim3(nq), dim3(128), 0, c->stream, fvals[r] + (pc.sharded ? 2 * row_base : 0), 0, 2 << ab, 0,
                           ftrees[r], leaves, depth_r[r], pc.sharded ? d_idx + (R + 2) * nq : d_idx + (r + 1) * nq,
                           (pc.sharded ? my_slot : d_un) + pc.off);
      }
      HIPCHK(hipGetLastError());
      ps.end();   // the kernels; the exchange below is its own (host-clock) stage
      if (per_rank && sh.world > 1) VXCHK(shard_allgather(c, sh, d_sh, per_rank * 8, "query openings (all sharded trees)"));
      ProfScope ps2(c, "query_gather");
      std::vector<u64> h_sh(per_rank * (per_rank ? sh.world : 0)), h_un(unsharded);
      if (per_rank) HIPCHK(hipMemcpyAsync(h_sh.data(), d_sh, h_sh.size() * 8, hipMemcpyDeviceToHost, c->stream));
      if (unsharded) HIPCHK(hipMemcpyAsync(h_un.data(), d_un, h_un.size() * 8, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      for (const Piece& pc : pieces) {   // the layout write_fri_proof reads: [slot][query][row] per tree
        const size_t len = pc.rowlen * nq, slots = pc.sharded ? (size_t)sh.world : 1;
        pc.host->resize(len * slots);
        for (size_t sl = 0; sl < slots; ++sl)
          memcpy(pc.host->data() + len * sl, pc.sharded ? h_sh.data() + per_rank * sl + pc.off : h_un.data() + pc.off, len * 8);
      }
    }
    out.flen = flen;
    out.depth0 = depth0;
  }
  return VX_OK;
}

// FriProof part of util/serialization::write_proof: commit-phase caps, query rounds, final polynomial, pow witness.
static void write_fri_proof(ByteSink& w, const FriProverParams& fp, const std::vector<vx_batch*>& oracles, const FriParts& f, const Shard& sh) {
  const size_t cap_words = (size_t)4 << fp.cap_height, R = fp.arity_bits.size();
  const int nq = fp.num_queries;
  auto slot_of = [&](bool sharded, int q) { return sharded ? (size_t)f.owner[q] : (size_t)0; };
  for (auto& cp : f.commit_caps) w.words(cp.data(), cap_words);
  for (int q = 0; q < nq; ++q) {
    for (size_t o = 0; o < oracles.size(); ++o) {
      size_t width = oracles[o]->ncols, rowlen = width + 4 * (size_t)f.depth0;
      const u64* row = &f.init_out[o][rowlen * (slot_of(oracles[o]->shard_lg > 0, q) * nq + q)];
      w.words(row, width);
      w.u8((uint8_t)f.depth0);
      w.words(row + width, 4 * (size_t)f.depth0);
    }
    for (size_t r = 0; r < R; ++r) {
      const int ab = fp.arity_bits[r];
      const size_t leaves = f.flen[r] >> ab;
      int depth = 0;
      while (((size_t)1 << (depth + fp.cap_height)) < leaves) ++depth;
      size_t width = (size_t)2 << ab, rowlen = width + 4 * (size_t)depth;
      const u64* row = &f.step_out[r][rowlen * (slot_of(r == 0 && sh.world > 1, q) * nq + q)];
      w.words(row, width);
      w.u8((uint8_t)depth);
      w.words(row + width, 4 * (size_t)depth);
    }
  }
  for (vxh::Ext e : f.final_poly) {
    w.f(e.a);
    w.f(e.b);
  }
  w.f(f.pow_witness);
}

// prover.rs::compute_lookup_polys for every challenge: the polynomials [RE, SLDC_0 .. SLDC_{k-1}] are zero outside the
// lookup rows [last_lu_row, first_lut_row] of each table, and inside them they are short recurrences ACROSS rows (RE is a
// Horner chain down the table rows, the partial Sums / LDCs running sums) over at most a few thousand rows — so the rows
// are gathered from the device-resident witness (80 routed columns x the row range), the recurrences run on the host,
// and only the non-zero row range of the nch * nlp columns is written back.  dst: [nch * nlp][n] on the device.
// Montgomery's trick: v[i] <- 1 / v[i] for `count` non-zero values with ONE field inversion (3 multiplications per element
// instead of a ~70-multiplication exponentiation each: the looking rows of a 2^20-row circuit hold 2.6 M denominators, which took
// 0.57 s per proof inverted one by one — round 3).  A zero denominator (probability 2^-64 per element) maps to zero.
static inline void batch_inverse(vxh::u64* v, size_t count, std::vector<vxh::u64>& scratch) {
  using namespace vxh;
  scratch.resize(count);
  u64 acc = 1;
  for (size_t i = 0; i < count; ++i) {
    scratch[i] = acc;
    if (v[i]) acc = mul(acc, v[i]);
  }
  acc = inv(acc);
  for (size_t i = count; i-- > 0;) {
    if (!v[i]) continue;
    const u64 t = mul(acc, scratch[i]);
    acc = mul(acc, v[i]);
    v[i] = t;
  }
}
// the device version (plonk_kernels.hip.h lookup_poly_*_kernel): no copy to the host, no synchronisation; VX_LOOKUP_POLYS_HOST=1 keeps the
// host recurrences of rounds 3-5 (cross-check, A/B)
static int lookup_polys_host(vx_ctx* c, const vx_circuit* k, const u64* d_wires, const std::vector<u64>& deltas, u64* dst);
static int lookup_polys_to_device(vx_ctx* c, const vx_circuit* k, const u64* d_wires, const std::vector<u64>& deltas, u64* dst, Scratch& S) {
  using namespace vxh;
  const size_t n = k->n();
  const int nch = k->nch, nlp = k->nlp(), nsl = nlp - 1;
  const int lu_slots = k->nr / 2, lut_slots = k->nr / 3, lu_deg = k->qdf - 1, lut_deg = (lut_slots + nsl - 1) / nsl;
  static const bool on_host = getenv("VX_LOOKUP_POLYS_HOST") != nullptr;
  if (on_host || lu_slots > VX_LOOKUP_SLOTS_MAX || lut_slots > VX_LOOKUP_SLOTS_MAX) return lookup_polys_host(c, k, d_wires, deltas, dst);
  HIPCHK(hipMemsetAsync(dst, 0, (size_t)nch * nlp * n * 8, c->stream));
  ProfScope ps(c, "lookup_polys");
  for (int t = 0; t < k->num_luts; ++t) {
    LookupPolyParams lp;
    memset(&lp, 0, sizeof lp);
    lp.wires = d_wires, lp.dst = dst, lp.n = n;
    lp.last_lu = k->lookup_rows[3 * t], lp.last_lut = k->lookup_rows[3 * t + 1], lp.first_lut = k->lookup_rows[3 * t + 2];
    lp.R = lp.first_lut - lp.last_lu + 1;
    lp.nch = nch, lp.nsl = nsl, lp.lu_slots = lu_slots, lp.lut_slots = lut_slots, lp.lu_deg = lu_deg, lp.lut_deg = lut_deg;
    for (int cI = 0; cI < nch; ++cI) {
      for (int q = 0; q < 4; ++q) lp.deltas[cI][q] = deltas[4 * cI + q];
      lp.delta_pow_slots[cI] = pow(deltas[4 * cI + 3], (u64)lut_slots);
    }
    lp.gc = S.get((size_t)nch * nsl * lp.R);
    lp.hrow = S.get((size_t)nch * lp.R);
    if (!lp.gc || !lp.hrow) return vx_fail(VX_E_NOMEM, "prove: out of device memory (lookup polynomials)");
    hipLaunchKernelGGL(lookup_poly_rows_kernel, dim3((unsigned)(((size_t)lp.R * nch + 63) / 64)), dim3(64), 0, c->stream, lp);
    hipLaunchKernelGGL(lookup_poly_scan_kernel, dim3((unsigned)nch), dim3(1024), 0, c->stream, lp);
    HIPCHK(hipGetLastError());
  }
  return VX_OK;
}
static int lookup_polys_host(vx_ctx* c, const vx_circuit* k, const u64* d_wires, const std::vector<u64>& deltas, u64* dst) {
  using namespace vxh;
  std::vector<u64> inv_scratch;
  const size_t n = k->n();
  const int nch = k->nch, nlp = k->nlp(), nsl = nlp - 1;
  const int lu_slots = k->nr / 2, lut_slots = k->nr / 3, lu_deg = k->qdf - 1, lut_deg = (lut_slots + nsl - 1) / nsl;
  HIPCHK(hipMemsetAsync(dst, 0, (size_t)nch * nlp * n * 8, c->stream));
  for (int t = 0; t < k->num_luts; ++t) {
    const size_t last_lu = (size_t)k->lookup_rows[3 * t], last_lut = (size_t)k->lookup_rows[3 * t + 1], first_lut = (size_t)k->lookup_rows[3 * t + 2];
    const size_t R = first_lut - last_lu + 1;
    std::vector<u64> w((size_t)k->nr * R);  // [col][row - last_lu]
    HIPCHK(hipMemcpy2DAsync(w.data(), R * 8, d_wires + last_lu, n * 8, R * 8, (size_t)k->nr, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (auto& v : w) v = canon(v);
    auto W = [&](int col, size_t row) { return w[(size_t)col * R + (row - last_lu)]; };
    std::vector<u64> polys((size_t)nch * nlp * (R + 1), 0);  // [(ch, poly)][row - last_lu], one extra zero row above first_lut
    for (int cI = 0; cI < nch; ++cI) {
      const u64 da = deltas[4 * cI], db = deltas[4 * cI + 1], dalpha = deltas[4 * cI + 2], ddelta = deltas[4 * cI + 3];
      auto PV = [&](int poly, size_t row) -> u64& { return polys[((size_t)cI * nlp + poly) * (R + 1) + (row - last_lu)]; };
      std::vector<u64> den(std::max(lu_slots, lut_slots));
      for (size_t row = first_lut + 1; row-- > last_lut;) {   // table rows: RE and the partial Sums
        u64 re = PV(0, row + 1);
        for (int s2 = 0; s2 < lut_slots; ++s2) {
          const u64 inp = W(3 * s2, row), out = W(3 * s2 + 1, row);
          den[s2] = sub(dalpha, add(inp, mul(da, out)));
          re = add(mul(re, ddelta), add(inp, mul(db, out)));
        }
        batch_inverse(den.data(), (size_t)lut_slots, inv_scratch);
        PV(0, row) = re;
        for (int slot = 0; slot < nsl; ++slot) {
          u64 acc = slot ? PV(slot, row) : PV(nsl, row + 1);
          for (int s2 = slot * lut_deg; s2 < std::min((slot + 1) * lut_deg, lut_slots); ++s2) acc = add(acc, mul(W(3 * s2 + 2, row), den[s2]));
          PV(slot + 1, row) = acc;
        }
      }
      for (size_t row = last_lut; row-- > last_lu;) {        // looking rows: the partial LDCs
        for (int s2 = 0; s2 < lu_slots; ++s2) den[s2] = sub(dalpha, add(W(2 * s2, row), mul(da, W(2 * s2 + 1, row))));
        batch_inverse(den.data(), (size_t)lu_slots, inv_scratch);
        for (int slot = 0; slot < nsl; ++slot) {
          const u64 prev = slot ? PV(slot, row) : PV(nsl, row + 1);
          u64 sum = 0;
          for (int s2 = slot * lu_deg; s2 < std::min((slot + 1) * lu_deg, lu_slots); ++s2) sum = add(sum, den[s2]);
          PV(slot + 1, row) = sub(prev, sum);
        }
      }
    }
    HIPCHK(hipMemcpy2DAsync(dst + last_lu, n * 8, polys.data(), (R + 1) * 8, R * 8, (size_t)nch * nlp, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));  // `polys` and `w` are stack-scoped sources / sinks of the copies
  }
  return VX_OK;
}

static int prove_impl(vx_ctx* c, vx_circuit* k, const u64* wires_in, bool wires_on_device, const u64* pow_hint,
                      std::vector<uint8_t>& proof_out, const Shard& sh = Shard()) {
  using namespace vxh;
  const size_t n = k->n();
  const int lg = k->degree_bits, rb = k->rate_bits, LG = lg + rb;
  const size_t N = (size_t)1 << LG;
  const int nch = k->nch, npp = k->npp(), nchunks = npp + 1, qdf = k->qdf, rate = 1 << rb;
  const size_t cap_words = (size_t)4 << k->cap_height;
  // this rank's share of every LDE: rows [row_base, row_base + Nl) = cosets [z0, z0 + zc) in bit-reversed order
  const size_t Nl = N >> sh.lg, row_base = Nl * (size_t)sh.rank;
  const int zc = rate >> sh.lg;
  Scratch S(c);
  // host sources of asynchronous uploads on the context's stream: they must outlive the copies, so they live as long as the proof
  std::vector<u64> ap;
  std::vector<Limbs3x2> al;
  vx_batch *wires_b = nullptr, *zs_b = nullptr, *quot_b = nullptr;
  struct Cleanup {
    vx_ctx* c;
    vx_batch **a, **b, **d;
    ~Cleanup() {
      batch_release(c, *a);
      batch_release(c, *b);
      batch_release(c, *d);
    }
  } cleanup{c, &wires_b, &zs_b, &quot_b};

  // ---- witness on device ----
  const u64* d_wires = wires_in;
  bool wires_committed = false;
  std::vector<u64> public_inputs(k->pi_rows.size());
  if (!wires_on_device && sh.world == 1 && !getenv("VX_NO_UPLOAD_OVERLAP")) {
    // Host witness, one GPU: the upload (2.27 GB at n = 2^21, ~40 ms over PCIe) is hidden behind the interpolation and
    // coset extension of the wires (batch_commit_host)
    u64* w = S.get((size_t)k->num_wires * n);
    if (!w) return vx_fail(VX_E_NOMEM, "prove: out of device memory (witness)");
    for (size_t i = 0; i < public_inputs.size(); ++i) public_inputs[i] = wires_in[(size_t)k->pi_cols[i] * n + k->pi_rows[i]];
    VXCHK(batch_alloc(c, lg, k->num_wires, rb, k->cap_height, &wires_b, 0, 0));
    VXCHK(batch_commit_host(c, wires_b, wires_in, w, false));
    d_wires = w;
    wires_committed = true;
  } else if (!wires_on_device) {
    // each rank uploads 1/world of the columns over its own PCIe link; the ranks then all-gather the matrix
    const size_t per = ((size_t)k->num_wires + sh.world - 1) / sh.world;
    u64* w = S.get(per * sh.world * n);
    if (!w) return vx_fail(VX_E_NOMEM, "prove: out of device memory (witness)");
    const size_t c0 = std::min((size_t)k->num_wires, per * sh.rank), c1 = std::min((size_t)k->num_wires, c0 + per);
    {
      ProfScope ps(c, "h2d_witness");
      if (c1 > c0) HIPCHK(hipMemcpyAsync(w + c0 * n, wires_in + c0 * n, (c1 - c0) * n * 8, hipMemcpyHostToDevice, c->stream));
    }
    VXCHK(shard_allgather(c, sh, w, per * n * 8, "witness columns"));
    d_wires = w;
  }
  // public inputs = witness.get_targets(public_inputs); public_inputs_hash = hash_no_pad(..)
  if (!wires_committed) {
    for (size_t i = 0; i < public_inputs.size(); ++i) {
      const u64* src = d_wires + (size_t)k->pi_cols[i] * n + k->pi_rows[i];
      HIPCHK(hipMemcpyAsync(&public_inputs[i], src, 8, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  for (auto& v : public_inputs) v = canon(v);
  Hash4 pih = hash_no_pad(public_inputs.data(), public_inputs.size());

  // ---- wires commitment ----
  if (!wires_committed) {
    VXCHK(batch_alloc(c, lg, k->num_wires, rb, k->cap_height, &wires_b, sh.rank, sh.lg));
    VXCHK(batch_commit_device(c, wires_b, d_wires, n, false));
  }
  std::vector<u64> wires_cap, zs_cap, quot_cap;
  VXCHK(gather_cap(c, sh, S, wires_b->tree + wires_b->cap_off * 4, wires_b->local_cap_words(), wires_cap));

  Challenger ch;
  ch.observe_elements(k->digest.e, 4);
  ch.observe_elements(pih.e, 4);
  ch.observe_elements(wires_cap.data(), cap_words);
  u64 betas[VX_MAX_CHALLENGES] = {0}, gammas[VX_MAX_CHALLENGES] = {0}, alphas[VX_MAX_CHALLENGES] = {0};
  for (int i = 0; i < nch; ++i) betas[i] = ch.get_challenge();
  for (int i = 0; i < nch; ++i) gammas[i] = ch.get_challenge();
  // lookup challenges (prover.rs): NUM_COINS_LOOKUP = 4 per challenge; the first 2 nch of them ARE the betas and gammas
  const int nlp = k->nlp(), nlook = nch * nlp;
  std::vector<u64> deltas;
  if (k->num_luts > 0) {
    deltas.assign(betas, betas + nch);
    deltas.insert(deltas.end(), gammas, gammas + nch);
    for (int i = 0; i < 2 * nch; ++i) deltas.push_back(ch.get_challenge());
  }

  // ---- Z and partial products ----
  {
    const size_t nblocks = (n + 255) / 256;
    u64* cp = S.get((size_t)nch * nchunks * n);
    u64* bp = S.get((size_t)nch * nblocks);
    u64* zs_vals = S.get(((size_t)nch * (1 + npp) + nlook) * n);
    if (!cp || !bp || !zs_vals) return vx_fail(VX_E_NOMEM, "prove: out of device memory (permutation argument)");
    {
      ProfScope ps(c, "perm_z_partial_products", (double)nch * 8.0 * n * (2.0 * k->nr + 1 + npp));
      PermParams pp;
      pp.wires = d_wires;
      pp.sigmas = k->sigmas;
      pp.k_is = k->k_is;
      pp.root_lo = c->root_lo;
      pp.root_hi = c->root_hi;
      pp.n = n;
      pp.log_n = lg;
      pp.nr = k->nr;
      pp.deg = qdf;
      pp.nchunks = nchunks;
      pp.nch = nch;
      for (int i = 0; i < VX_MAX_CHALLENGES; ++i) pp.betas[i] = betas[i], pp.gammas[i] = gammas[i];
      pp.cp = cp;
      launch_perm_chunk_products(pp, dim3((unsigned)nblocks, nch), c->stream);
      hipLaunchKernelGGL(perm_block_products_kernel, dim3((unsigned)nblocks, nch), dim3(256), 0, c->stream, cp, n, nchunks, bp);
      hipLaunchKernelGGL(perm_scan_blocks_kernel, dim3(nch), dim3(64), 0, c->stream, bp, nblocks);
      hipLaunchKernelGGL(perm_write_kernel, dim3((unsigned)nblocks, nch), dim3(256), 0, c->stream, cp, n, nchunks, nch, bp, zs_vals);
      HIPCHK(hipGetLastError());
    }
    if (k->num_luts > 0) VXCHK(lookup_polys_to_device(c, k, d_wires, deltas, zs_vals + (size_t)nch * (1 + npp) * n, S));
    VXCHK(batch_alloc(c, lg, (size_t)nch * (1 + npp) + nlook, rb, k->cap_height, &zs_b, sh.rank, sh.lg));
    VXCHK(batch_commit_device(c, zs_b, zs_vals, n, false));
  }
  VXCHK(gather_cap(c, sh, S, zs_b->tree + zs_b->cap_off * 4, zs_b->local_cap_words(), zs_cap));
  ch.observe_elements(zs_cap.data(), cap_words);
  for (int i = 0; i < nch; ++i) alphas[i] = ch.get_challenge();

  // ---- quotient polynomials ----
  {
    u64* qv = S.get((size_t)nch * Nl);  // quotient values on the local cosets
    u64* qu = S.get((size_t)nch * N);   // per-coset coefficients of ALL cosets: [world][nch][zc][n]
    if (!qv || !qu) return vx_fail(VX_E_NOMEM, "prove: out of device memory (quotient)");
    {
      QuotientParams qp;
      memset(&qp, 0, sizeof qp);
      qp.cs = k->cs->lde;
      qp.wires = wires_b->lde;
      qp.zs = zs_b->lde;
      qp.k_is = k->k_is;
      qp.root_lo = c->root_lo;
      qp.root_hi = c->root_hi;
      qp.N = N;
      qp.rows = Nl;
      qp.row_base = row_base;
      qp.stride_w = Nl;
      qp.log_n = lg;
      qp.rate_bits = rb;
      qp.num_selectors = k->num_selectors;
      qp.const_base = k->const_base();
      const int lookup_terms = k->num_luts > 0 ? nch * (4 + k->num_luts + 2 * k->num_sldc()) : 0;
      qp.extra_terms = lookup_terms;
      qp.num_constants = k->num_constants;
      qp.nr = k->nr;
      qp.num_wires = k->num_wires;
      qp.nch = nch;
      qp.npp = npp;
      qp.deg = qdf;
      qp.num_gates = (int)k->gates.size();
      for (size_t g = 0; g < k->gates.size(); ++g) qp.gates[g] = k->gates[g];
      for (int i = 0; i < VX_MAX_CHALLENGES; ++i) qp.betas[i] = betas[i], qp.gammas[i] = gammas[i], qp.alphas[i] = alphas[i];
      for (int i = 0; i < 4; ++i) qp.pih[i] = pih.e[i];
      // ZeroPolyOnCoset: Z_H(x) on coset r = 7^n * w_rate^r - 1
      u64 shift_n = pow(7, n), g_rate = root_of_unity(rb), pw = 1;
      for (int r = 0; r < rate; ++r) {
        qp.zh[r] = sub(mul(shift_n, pw), 1);
        qp.zh_inv[r] = inv(qp.zh[r]);
        pw = mul(pw, g_rate);
      }
      qp.n_field = (u64)n % P;
      qp.l0 = k->l0_lde;
      {
        // alpha powers for reduce_with_powers: L_0 terms, partial-product checks, then the widest gate (123 constraints)
        if ((size_t)nch * (1 + nchunks) + (size_t)lookup_terms + 160 > VX_ALPHA_POWS) return vx_fail(VX_E_INVALID, "prove: too many constraint terms");
        ap.assign((size_t)VX_MAX_CHALLENGES * VX_ALPHA_POWS, 0);
        for (int cI = 0; cI < nch; ++cI) {
          u64 pw = 1;
          for (int i = 0; i < VX_ALPHA_POWS; ++i) {
            ap[(size_t)cI * VX_ALPHA_POWS + i] = pw;
            pw = mul(pw, alphas[cI]);
          }
        }
        u64* d_ap = S.get(ap.size());
        if (!d_ap) return vx_fail(VX_E_NOMEM, "prove: out of device memory (alpha powers)");
        HIPCHK(hipMemcpyAsync(d_ap, ap.data(), ap.size() * 8, hipMemcpyHostToDevice, c->stream));
        qp.alpha_pows = d_ap;
        // the same powers split for the quotient kernel's carry-free accumulation: limbs of a and of 2^32 a
        al.assign(ap.size(), Limbs3x2{});
        for (size_t i = 0; i < ap.size(); ++i) {
          const u64 bb = ap[i], bh = mul(bb, (u64)1 << 32);
          al[i].lo[0] = (u32)(bb & 0x3FFFFFu), al[i].lo[1] = (u32)((bb >> 22) & 0x3FFFFFu), al[i].lo[2] = (u32)(bb >> 44);
          al[i].hi[0] = (u32)(bh & 0x3FFFFFu), al[i].hi[1] = (u32)((bh >> 22) & 0x3FFFFFu), al[i].hi[2] = (u32)(bh >> 44);
        }
        Limbs3x2* d_al = (Limbs3x2*)S.get((al.size() * sizeof(Limbs3x2) + 7) / 8);
        if (!d_al) return vx_fail(VX_E_NOMEM, "prove: out of device memory (alpha powers)");
        HIPCHK(hipMemcpyAsync(d_al, al.data(), al.size() * sizeof(Limbs3x2), hipMemcpyHostToDevice, c->stream));
        qp.alpha_limbs = d_al;
      }
      qp.out = qv;
      size_t bytes_read = 8ull * Nl * ((size_t)k->num_constants + k->nr + k->num_wires + (size_t)nch * (2 + npp));
      ProfScope ps(c, "quotient_eval", (double)bytes_read);
      // every kernel of the quotient is its own profile stage INSIDE quotient_eval (round 6: "quotient_eval by kernel")
      {
        ProfScope pk(c, "quotient_l0_permutation");
        hipLaunchKernelGGL(quotient_kernel<0>, dim3((unsigned)((Nl + 255) / 256)), dim3(256), 0, c->stream, qp);   // out = (L_0 + permutation terms) / Z_H
      }
      bool small_gates = false, poseidon_gate = false;
      for (const GateDev& gd : k->gates) {
        small_gates |= gd.type >= 1 && gd.type <= 3;
        poseidon_gate |= gd.type == 4;
      }
      if (small_gates) {
        ProfScope pk(c, "quotient_small_native_gates");
        hipLaunchKernelGGL(quotient_kernel<1>, dim3((unsigned)((Nl + 255) / 256)), dim3(256), 0, c->stream, qp);   // out += Constant / PublicInput / Arithmetic gate terms / Z_H
      }
      if (poseidon_gate) {
        ProfScope pk(c, "quotient_poseidon_gate");
        hipLaunchKernelGGL(quotient_kernel<2>, dim3((unsigned)((Nl + 255) / 256)), dim3(256), 0, c->stream, qp);  // out += PoseidonGate terms / Z_H
      }
      HIPCHK(hipGetLastError());
      if (k->num_luts > 0) {  // the lookup argument's terms sit between the partial-product checks and the gate constraints
        LookupParams lp;
        memset(&lp, 0, sizeof lp);
        lp.cs = k->cs->lde, lp.wires = wires_b->lde, lp.zs = zs_b->lde;
        lp.alpha_pows = qp.alpha_pows;
        lp.out = qv;
        lp.N = N, lp.rows = Nl, lp.row_base = row_base, lp.stride_w = Nl;
        lp.log_n = lg, lp.rate_bits = rb, lp.nch = nch;
        lp.sel_base = k->num_selectors;
        lp.zs_base = nch * (1 + npp);
        lp.nlp = nlp;
        lp.lu_slots = k->nr / 2, lp.lut_slots = k->nr / 3, lp.lu_deg = qdf - 1;
        lp.lut_deg = (lp.lut_slots + k->num_sldc() - 1) / k->num_sldc();
        lp.num_luts = k->num_luts;
        lp.base_idx = nch * (1 + nchunks);
        for (int cI = 0; cI < nch; ++cI) {
          for (int q = 0; q < 4; ++q) lp.deltas[cI][q] = deltas[4 * cI + q];
          size_t off = 0;
          for (int t = 0; t < k->num_luts; ++t) {  // vanishing_poly.rs::get_lut_poly, zero-padded to whole table rows
            const size_t len = (size_t)k->lut_lens[t], degree = (len + lp.lut_slots - 1) / lp.lut_slots * lp.lut_slots;
            u64 acc = 0;
            for (size_t e = 0; e < degree; ++e) {
              const u64 coeff = e < len ? add((u64)k->lut_inputs[off + e], mul(deltas[4 * cI + 1], (u64)k->lut_outputs[off + e])) : 0;
              acc = add(mul(acc, deltas[4 * cI + 3]), coeff);
            }
            off += len;
            lp.lut_poly[cI][t] = acc;
          }
        }
        for (int r = 0; r < rate; ++r) lp.zh_inv[r] = qp.zh_inv[r];
        ProfScope psl(c, "quotient_lookup_terms");
        static const bool terms_generic = getenv("VX_LOOKUP_TERMS_GENERIC") != nullptr;   // cross-check: the generic kernel for every shape
        if (lp.nlp == 7 && lp.lut_slots == 26 && lp.lu_slots == 40 && lp.lut_deg == 5 && lp.lu_deg == 7 && !terms_generic) {   // standard_recursion_config
          LookupStaticExtra lx;
          memset(&lx, 0, sizeof lx);
          lx.alpha_limbs = qp.alpha_limbs;
          auto limbs_of = [](u64 bb) {
            Limbs3x2 t;
            const u64 bh = mul(bb, (u64)1 << 32);
            t.lo[0] = (u32)(bb & 0x3FFFFFu), t.lo[1] = (u32)((bb >> 22) & 0x3FFFFFu), t.lo[2] = (u32)(bb >> 44);
            t.hi[0] = (u32)(bh & 0x3FFFFFu), t.hi[1] = (u32)((bh >> 22) & 0x3FFFFFu), t.hi[2] = (u32)(bh >> 44);
            return t;
          };
          for (int cI = 0; cI < nch; ++cI) {
            const u64 db = deltas[4 * cI + 1], dd = deltas[4 * cI + 3];
            u64 pw = 1;
            for (int sl = 25; sl >= 0; --sl) {   // slot sl meets delta^(25 - sl)
              lx.re_in[cI][sl] = limbs_of(pw);
              lx.re_out[cI][sl] = limbs_of(mul(db, pw));
              pw = mul(pw, dd);
            }
            lx.delta26[cI] = pw;
          }
          static const int wps = getenv("VX_LOOKUP_WPS") ? atoi(getenv("VX_LOOKUP_WPS")) : 3;   // waves per SIMD the kernel is compiled for: 3 = 168 VGPRs, no spills (0.87 ms at 2^18 rows); 4 = 128 VGPRs, 44 spills (1.08 ms)
          if (wps != 4) hipLaunchKernelGGL(lookup_terms_static_kernel<3>, dim3((unsigned)((Nl + 255) / 256)), dim3(256), 0, c->stream, lp, lx);
          else hipLaunchKernelGGL(lookup_terms_static_kernel<4>, dim3((unsigned)((Nl + 255) / 256)), dim3(256), 0, c->stream, lp, lx);
        } else {
          hipLaunchKernelGGL(lookup_terms_kernel, dim3((unsigned)((Nl + 255) / 256)), dim3(256), 0, c->stream, lp);
        }
        HIPCHK(hipGetLastError());
      }
      if (k->programs) {  // gates supplied as constraint programs add their share to the same quotient values
        const u64 nterms_before = (u64)nch * (1 + nchunks) + (u64)lookup_terms;  // L_0 terms, partial-product checks [, lookup terms] come first
        {
          JitGateParams jp;
          memset(&jp, 0, sizeof jp);
          jp.cs = k->cs->lde;
          jp.wires = wires_b->lde;
          jp.alpha_pows = qp.alpha_pows;
          jp.alpha_limbs = qp.alpha_limbs;
          jp.out = qv;
          jp.N = N;
          jp.rows = Nl;
          jp.row_base = row_base;
          jp.stride_w = Nl;
          jp.log_n = lg;
          jp.rate_bits = rb;
          jp.num_selectors = k->num_selectors;
          jp.const_base = k->const_base();
          jp.nch = nch;
          jp.base_idx = (int)nterms_before;
          for (int i = 0; i < 4; ++i) jp.pih[i] = pih.e[i];
          for (int r = 0; r < rate; ++r) jp.zh_inv[r] = qp.zh_inv[r];
          if (k->jit_fused_fn) {
            JitFusedParams fp;
            memset(&fp, 0, sizeof fp);
            fp.cs = jp.cs, fp.wires = jp.wires, fp.alpha_limbs = jp.alpha_limbs, fp.out = jp.out;
            fp.N = jp.N, fp.rows = jp.rows, fp.row_base = jp.row_base, fp.stride_w = jp.stride_w;
            fp.log_n = jp.log_n, fp.rate_bits = jp.rate_bits, fp.num_selectors = jp.num_selectors, fp.nch = jp.nch;
            fp.base_idx = jp.base_idx, fp.const_base = jp.const_base;
            for (int i = 0; i < 4; ++i) fp.pih[i] = jp.pih[i];
            for (int r = 0; r < rate; ++r) fp.zh_inv[r] = jp.zh_inv[r];
            fp.ngates = (int)k->jit_gates.size();
            for (int q = 0; q < fp.ngates; ++q) {
              const int g = k->jit_gates[q];
              fp.g[q] = JitGateRt{g, k->gates[g].selector_index, k->gates[g].group_start, k->gates[g].group_end};
            }
            const int waves = k->jit_fused_waves;
            void* args[] = {&fp};
            ProfScope psj(c, "quotient_program_gates_jit");
            HIPCHK(hipModuleLaunchKernel(k->jit_fused_fn, (unsigned)((Nl + 63) / 64), 1, 1, 64 * waves, 1, 1, 0, c->stream, args, nullptr));
          } else if (!k->jit_fns.empty()) {
            ProfScope psj(c, "quotient_program_gates_jit");
            for (size_t gi = 0; gi < k->jit_fns.size(); ++gi) {   // one launch per group of program gates, each adds its share
              jp.ngates = (int)k->jit_groups[gi].size();
              for (int q = 0; q < jp.ngates; ++q) {
                const int g = k->jit_gates[k->jit_groups[gi][q]];
                jp.g[q] = JitGateRt{g, k->gates[g].selector_index, k->gates[g].group_start, k->gates[g].group_end};
              }
              void* args[] = {&jp};
              ProfScope psg(c, k->jit_stage[gi].c_str());
              HIPCHK(hipModuleLaunchKernel(k->jit_fns[gi], (unsigned)((Nl + 255) / 256), 1, 1, 256, 1, 1, 0, c->stream, args, nullptr));
            }
          }
        }
        ProgramParams pg;
        memset(&pg, 0, sizeof pg);
        pg.cs = k->cs->lde;
        pg.wires = wires_b->lde;
        pg.programs = k->programs;
        pg.N = N;
        pg.rows = Nl;
        pg.row_base = row_base;
        pg.stride_w = Nl;
        pg.log_n = lg;
        pg.rate_bits = rb;
        pg.num_selectors = k->num_selectors;
        pg.const_base = k->const_base();
        pg.nch = nch;
        for (size_t g = 0; g < k->gates.size(); ++g)
          if (k->prog_off[g] >= 0 && k->jit_fns.empty() && !k->jit_fused_fn)  // not compiled: interpreter
            pg.gates[pg.num_gates++] = ProgramGateDev{(int)g, k->gates[g].selector_index, k->gates[g].group_start, k->gates[g].group_end, k->prog_off[g]};
        for (int i = 0; i < VX_MAX_CHALLENGES; ++i) pg.alphas[i] = alphas[i], pg.base_pw[i] = pow(alphas[i], nterms_before);
        for (int i = 0; i < 4; ++i) pg.pih[i] = pih.e[i];
        for (int r = 0; r < rate; ++r) pg.zh_inv[r] = qp.zh_inv[r];
        pg.out = qv;
        if (pg.num_gates) {
          ProfScope ps2(c, "quotient_program_gates");
          hipLaunchKernelGGL(program_gates_kernel, dim3((unsigned)((Nl + 255) / 256)), dim3(256), 0, c->stream, pg);
          HIPCHK(hipGetLastError());
        }
      }
    }
    // per-coset inverse NTT (input rows of each block are in bit-reversed order), then the cross-coset
    // inverse DFT that separates the degree-n chunks
    u64 ninv = inv((u64)n % P);
    VXCHK(run_ntt(c, qv, qu + (size_t)sh.rank * nch * Nl, Nl, Nl, n, n, lg, nch, zc, true, true, nullptr, 0, ninv, "quotient_intt",
                  16.0 * Nl * nch));
    VXCHK(shard_allgather(c, sh, qu, (size_t)nch * Nl * 8, "quotient coset coefficients"));
    VXCHK(batch_alloc(c, lg, (size_t)nch * qdf, rb, k->cap_height, &quot_b, sh.rank, sh.lg));
    unsigned* tail_flag = (unsigned*)S.get(1);
    if (!tail_flag) return vx_fail(VX_E_NOMEM, "vx_prove: out of device memory");
    HIPCHK(hipMemsetAsync(tail_flag, 0, 8, c->stream));
    {
      std::vector<u64> inv_shifts(rate);
      u64 wN = root_of_unity(LG);
      for (int z = 0; z < rate; ++z) inv_shifts[z] = inv(mul(7, pow(wN, reverse_bits((size_t)z, rb))));
      int bits = lg / 2;
      u64* tab = nullptr;
      VXCHK(get_scale_tables(c, lg, bits, inv_shifts, 1, &tab));
      ChunkParams cp;
      memset(&cp, 0, sizeof cp);
      cp.u = qu;
      cp.t = quot_b->coeffs;
      cp.inv_tab = tab;
      cp.log_n = lg;
      cp.rb = rb;
      cp.bits = bits;
      cp.nch = nch;
      cp.zc = zc;
      cp.keep = qdf;  // quotient_poly.trim_to_len(quotient_degree_factor * n), then chunks(n)
      cp.tail_nonzero = tail_flag;
      u64 wr_inv = inv(root_of_unity(rb)), pw = 1;
      for (int i = 0; i < rate; ++i) {
        cp.w_rate_inv_pows[i] = pw;
        pw = mul(pw, wr_inv);
      }
      u64 s_inv = inv(pow(7, n)), rate_inv = inv((u64)rate);
      pw = rate_inv;
      for (int q = 0; q < rate; ++q) {
        cp.chunk_scale[q] = pw;
        pw = mul(pw, s_inv);
      }
      ProfScope ps(c, "quotient_chunks", 16.0 * N * nch);
      hipLaunchKernelGGL(quotient_chunks_kernel, dim3((unsigned)((n + 255) / 256), nch), dim3(256), 0, c->stream, cp);
      HIPCHK(hipGetLastError());
    }
    if (qdf < rate) {
      // plonky2: trim_to_len(..).expect("Quotient has failed, the vanishing polynomial is not divisible by Z_H")
      unsigned flag = 0;
      HIPCHK(hipMemcpyAsync(&flag, tail_flag, 4, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      if (flag && !c->rehearsal) return vx_fail(VX_E_PROOF, "vx_prove: the quotient has degree >= quotient_degree_factor * n (witness does not satisfy the circuit?)");
    }
    VXCHK(batch_lde_and_tree(c, quot_b));
  }
  VXCHK(gather_cap(c, sh, S, quot_b->tree + quot_b->cap_off * 4, quot_b->local_cap_words(), quot_cap));
  ch.observe_elements(quot_cap.data(), cap_words);
  Ext zeta = ch.get_extension_challenge();
  {
    Ext zp = zeta;
    for (int i = 0; i < lg; ++i) zp = emul(zp, zp);
    if (zp.a == 1 && zp.b == 0) return vx_fail(VX_E_PROOF, "Opening point is in the subgroup.");
  }
  const u64 g = root_of_unity(lg);
  Ext gzeta{mul(zeta.a, g), mul(zeta.b, g)};

  // ---- openings ----
  vx_batch* oracles[4] = {k->cs, wires_b, zs_b, quot_b};
  std::vector<u64> ev[4], zs_next(2 * (size_t)nch), lzs_next(2 * (size_t)nlook);
  const size_t zs_pp = (size_t)nch * (1 + npp);
  {
    EvalJob jobs[6];
    for (int o = 0; o < 4; ++o) {
      ev[o].resize(2 * oracles[o]->ncols);
      jobs[o] = EvalJob{oracles[o]->coeffs, oracles[o]->ncols, 0, ev[o].data()};
    }
    jobs[4] = EvalJob{zs_b->coeffs, (size_t)nch, 1, zs_next.data()};
    jobs[5] = EvalJob{zs_b->coeffs + zs_pp * n, (size_t)nlook, 1, lzs_next.data()};
    static const bool replicate = getenv("VX_SHARD_REPLICATE_OPENINGS") != nullptr;   // rounds 2-5: every rank evaluated every polynomial (A/B)
    if (sh.world == 1 || replicate) {
      VXCHK(batch_eval_ext_many(c, zeta, gzeta, lg, jobs, 6));
    } else {
      // One proof over G ranks (round 6): the coefficients are replicated, so the ~260 evaluations at zeta / g zeta shard by COLUMN RANGE
      // for free — rank r takes ceil(ncols / G) columns of every job — and one more small in-place all-gather (2 words per polynomial:
      // ~4 KB per proof) puts every value on every rank.  1.7 ms of 38.5 per rank were this stage at G = 8 (single-device emulation).
      const size_t G = (size_t)sh.world;
      EvalJob sub[6];
      std::vector<u64> mine_vals[6];
      size_t per[6], slot = 0;
      for (int j = 0; j < 6; ++j) {
        per[j] = (jobs[j].ncols + G - 1) / G;
        const size_t c0 = std::min(jobs[j].ncols, per[j] * (size_t)sh.rank), c1 = std::min(jobs[j].ncols, c0 + per[j]);
        mine_vals[j].assign(2 * per[j] + 2, 0);
        sub[j] = EvalJob{jobs[j].coeffs + c0 * n, c1 - c0, jobs[j].point, mine_vals[j].data()};
        slot += 2 * per[j];
      }
      VXCHK(batch_eval_ext_many(c, zeta, gzeta, lg, sub, 6));
      std::vector<u64> all(slot * G, 0);
      {
        size_t off = 0;
        for (int j = 0; j < 6; ++j) {
          memcpy(all.data() + slot * sh.rank + off, mine_vals[j].data(), 2 * per[j] * 8);
          off += 2 * per[j];
        }
      }
      u64* d_x = slot ? S.get(slot * G) : nullptr;
      if (slot && !d_x) return vx_fail(VX_E_NOMEM, "prove: out of device memory (openings exchange)");
      if (slot) {
        HIPCHK(hipMemcpyAsync(d_x + slot * sh.rank, all.data() + slot * sh.rank, slot * 8, hipMemcpyHostToDevice, c->stream));
        VXCHK(shard_allgather(c, sh, d_x, slot * 8, "openings at zeta and g zeta"));
        HIPCHK(hipMemcpyAsync(all.data(), d_x, slot * G * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
      }
      for (size_t r2 = 0; r2 < G; ++r2) {
        size_t off = 0;
        for (int j = 0; j < 6; ++j) {
          const size_t c0 = std::min(jobs[j].ncols, per[j] * r2), c1 = std::min(jobs[j].ncols, c0 + per[j]);
          if (c1 > c0) memcpy(jobs[j].out_host + 2 * c0, all.data() + slot * r2 + off, 2 * (c1 - c0) * 8);
          off += 2 * per[j];
        }
      }
    }
  }
  // to_fri_openings: batch 0 = [constants, sigmas, wires, zs, partial products, quotient, lookup_zs], batch 1 = [zs_next,
  // lookup_zs_next] — the lookup polynomials are the TAIL of the zs_partial_products oracle but are opened after the quotient
  std::vector<Ext> batch0, batch1;
  for (int o = 0; o < 4; ++o) {
    const size_t cnt = o == 2 ? zs_pp : oracles[o]->ncols;
    for (size_t i = 0; i < cnt; ++i) batch0.push_back(Ext{ev[o][2 * i], ev[o][2 * i + 1]});
  }
  for (size_t i = zs_pp; i < zs_pp + nlook; ++i) batch0.push_back(Ext{ev[2][2 * i], ev[2][2 * i + 1]});
  for (int i = 0; i < nch; ++i) batch1.push_back(Ext{zs_next[2 * i], zs_next[2 * i + 1]});
  for (int i = 0; i < nlook; ++i) batch1.push_back(Ext{lzs_next[2 * i], lzs_next[2 * i + 1]});
  for (Ext e : batch0) ch.observe_ext(e);
  for (Ext e : batch1) ch.observe_ext(e);

  // ---- prove_openings + fri_proof ----
  FriProverParams fp;
  fp.degree_bits = lg, fp.rate_bits = rb, fp.cap_height = k->cap_height, fp.pow_bits = k->pow_bits, fp.num_queries = k->num_queries;
  fp.arity_bits = k->arity_bits;
  std::vector<vx_batch*> fri_oracles(oracles, oracles + 4);
  // circuit_data.rs::get_fri_instance: fri_all_polys = [preprocessed, wires, zs + partial products, quotient, lookup polys],
  // fri_next_batch_polys = [zs, lookup polys]
  std::vector<FriRange> batch0_ranges = {FriRange{0, 0, oracles[0]->ncols}, FriRange{1, 0, oracles[1]->ncols}, FriRange{2, 0, zs_pp},
                                         FriRange{3, 0, oracles[3]->ncols}};
  std::vector<FriRange> batch1_ranges = {FriRange{2, 0, (size_t)nch}};  // plonk_zs_next: the Z polynomials lead the zs_partial_products batch
  if (nlook) {
    batch0_ranges.push_back(FriRange{2, zs_pp, (size_t)nlook});
    batch1_ranges.push_back(FriRange{2, zs_pp, (size_t)nlook});
  }
  FriParts fri;
  VXCHK(fri_prove_openings(c, fp, fri_oracles, batch0_ranges, batch1_ranges, zeta, gzeta, batch0, batch1, ch, pow_hint, sh, S, fri));
  {
    // ---- serialise (util/serialization::write_proof_with_public_inputs) ----
    ByteSink w;
    w.b.reserve(proof_size_bound(k));
    w.words(wires_cap.data(), cap_words);
    w.words(zs_cap.data(), cap_words);
    w.words(quot_cap.data(), cap_words);
    // write_opening_set (util/serialization): constants, plonk_sigmas, wires, plonk_zs, plonk_zs_next, lookup_zs,
    // lookup_zs_next, partial_products, quotient_polys — the serializer's order, NOT the struct's field order (the two
    // lookup vectors are empty for a circuit without tables, so lookup-free proofs are the same bytes either way)
    w.words(ev[0].data(), 2 * (size_t)k->num_constants);
    w.words(ev[0].data() + 2 * (size_t)k->num_constants, 2 * (size_t)k->nr);
    w.words(ev[1].data(), ev[1].size());
    w.words(ev[2].data(), 2 * (size_t)nch);
    w.words(zs_next.data(), 2 * (size_t)nch);
    w.words(ev[2].data() + 2 * zs_pp, 2 * (size_t)nlook);   // lookup_zs
    w.words(lzs_next.data(), lzs_next.size());               // lookup_zs_next
    w.words(ev[2].data() + 2 * (size_t)nch, 2 * (zs_pp - (size_t)nch));
    w.words(ev[3].data(), ev[3].size());
    write_fri_proof(w, fp, fri_oracles, fri, sh);
    w.words(public_inputs.data(), public_inputs.size());
    proof_out.swap(w.b);
  }
  return VX_OK;
}
