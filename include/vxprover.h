/* vxprover.h — C ABI of libvxprover.so, the MI355X (gfx950) prover backend for VectorX's plonky2x
 * circuits (header_range_256 / header_range_512 / rotate).
 *
 * This is the drop-in boundary of SURVEY.md §8(b).  The reference reaches the hot path through
 *   /root/reference/circuits/header_range.rs:167-170   circuit.prove(&input) / circuit.verify(..)
 *   /root/reference/circuits/rotate.rs:193             (same, rotate)
 *   /root/reference/bin/header_range_512.rs:16         HeaderRangeCircuit::<..>::entrypoint()
 * which land in the un-vendored plonky2 v0.2.0 crate (/root/reference/Cargo.lock:4848-4905):
 *   plonky2::fri::oracle::PolynomialBatch::{from_values, from_coeffs}      -> vx_batch_commit
 *   plonky2::hash::merkle_tree::MerkleTree::{new, prove}                   -> vx_merkle_*, vx_batch_open_row
 *   plonky2_field::fft::{fft, ifft}, polynomial::{coset_fft, coset_ifft}   -> vx_ntt_batch
 *   plonky2::hash::poseidon::Poseidon::poseidon                            -> vx_poseidon_permute
 *   plonky2::plonk::prover::prove_with_partition_witness                   -> vx_prove
 *   plonky2::plonk::circuit_data::CircuitData::verify                      -> vx_verify (host code)
 * INTEGRATION.md shows the Rust `extern "C"` block a maintainer adds on the reference side.
 *
 * Conventions (all entry points):
 *   - field element  = little-endian uint64_t; inputs may be non-canonical (< 2^64), outputs are
 *     canonical (< p = 2^64 - 2^32 + 1);  F_p^2 element = two consecutive u64 [c0, c1].
 *   - matrices are COLUMN-MAJOR [ncols][n]  (plonky2 `wire_values[wire][row]`).
 *   - a hash / Merkle digest = 4 u64.
 *   - return 0 on success, negative VX_E_* on failure; vx_last_error() gives a thread-local message.
 *     Nothing unwinds or aborts across this boundary.  There is NO CPU fallback: every compute entry
 *     point fails with VX_E_NO_DEVICE when no gfx950 device is usable.
 *   - the caller owns every host buffer; the library owns the opaque handles (freed by vx_*_free /
 *     vx_ctx_destroy).  A vx_ctx is bound to one device and one HIP stream; use one ctx per host
 *     thread.  Contexts are independent, so 8 threads can drive 8 GPUs.
 */
#ifndef VXPROVER_H
#define VXPROVER_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VX_OK 0
#define VX_E_INVALID (-1)    /* bad argument */
#define VX_E_NO_DEVICE (-2)  /* no usable HIP device (the library never falls back to the CPU) */
#define VX_E_HIP (-3)        /* a HIP runtime call failed */
#define VX_E_NOMEM (-4)
#define VX_E_PROOF (-5)      /* prover-level failure (e.g. zeta in the subgroup, unsatisfied witness check) */
#define VX_E_COMM (-6)       /* the caller's all-gather (vx_prove_sharded) reported a failure */

typedef struct vx_ctx vx_ctx;
typedef struct vx_batch vx_batch;     /* device-resident PolynomialBatch: coeffs + LDE + Merkle tree */
typedef struct vx_circuit vx_circuit; /* device-resident prover key: CommonCircuitData + ProverOnlyCircuitData */

/* ---- library / context ------------------------------------------------------------------- */
const char* vx_last_error(void);
const char* vx_version(void);
int vx_device_count(void);
int vx_device_max_clock_khz(int device); /* hipDeviceAttributeClockRate: the engine clock ceiling (bench.py prices the ALU bound at it); < 0 on error */
int vx_ctx_create(int device, vx_ctx** out);
void vx_ctx_destroy(vx_ctx* ctx);
int vx_ctx_sync(vx_ctx* ctx);
/* Give the context's cached device buffers (the size-bucketed pool a prover recycles proof after proof) back to the driver: for a
 * host that is about to hand the GPU to other processes (vectorx_amd/dag_pool.py).  Live objects are untouched. */
int vx_ctx_trim(vx_ctx* ctx);
/* Opaque hipStream_t of the context (so a harness can record its own events on the right stream). */
void* vx_ctx_stream(vx_ctx* ctx);

/* Effective shader clock (GHz) of the leaf-hashing kernel, measured on the device itself from s_memtime (shader-clock ticks)
 * over s_memrealtime (constant 100 MHz): with profiling on (vx_prof_enable) every 1024th block of every
 * hash_leaves_colmajor_kernel launch records its first wave's ticks, and this call averages the samples taken since the last
 * vx_prof_reset — the clock the timed launches really ran at.  Without samples it runs a ~4 ms stand-alone probe of Poseidon
 * permutations.  The clock follows the power budget (1.9 - 2.4 GHz on MI355X), so an instruction-issue roofline must use a
 * measured value. */
int vx_clock_probe(vx_ctx* ctx, double* ghz_out);

/* Per-kernel-family HIP-event timing.  enable=1 brackets every launch family with events on the
 * context's stream; vx_prof_get returns accumulated milliseconds and launch counts since the last
 * vx_prof_reset.  names/ms/calls may be NULL to query the entry count. */
int vx_prof_enable(vx_ctx* ctx, int enable);
int vx_prof_reset(vx_ctx* ctx);
int vx_prof_count(vx_ctx* ctx);
int vx_prof_get(vx_ctx* ctx, int index, char* name_out, size_t name_cap, double* ms_out, uint64_t* calls_out,
                double* alg_bytes_out);

/* ---- raw device buffers (keep inputs resident in HBM between calls) ----------------------- */
int vx_dev_alloc(vx_ctx* ctx, size_t bytes, void** dptr);
int vx_dev_free(vx_ctx* ctx, void* dptr);
int vx_dev_upload(vx_ctx* ctx, void* dptr, const void* host, size_t bytes);
int vx_dev_download(vx_ctx* ctx, void* host, const void* dptr, size_t bytes);
/* Scatter `count` 8-byte values from a contiguous host array into device memory at dptr + i*dst_stride_bytes
 * (e.g. one ROW of a column-major witness matrix: stride = 8 * 2^degree_bits). */
int vx_dev_upload_strided(vx_ctx* ctx, void* dptr, size_t dst_stride_bytes, const uint64_t* host, size_t count);

/* Page-locked host buffers: a witness generated into one is uploaded by vx_prove at full PCIe Gen5 rate
 * (pageable memory goes through a staging copy at a fraction of it). */
int vx_host_alloc(vx_ctx* ctx, size_t bytes, void** hptr);
int vx_host_free(vx_ctx* ctx, void* hptr);

/* ---- L1 primitives on host buffers --------------------------------------------------------- */
/* plonky2_field::fft conventions, natural order in and out, in place, column-major [ncols][2^log_n].
 * kind: 0 fft, 1 ifft, 2 coset_fft(shift), 3 coset_ifft(shift). */
#define VX_NTT_FFT 0
#define VX_NTT_IFFT 1
#define VX_NTT_COSET_FFT 2
#define VX_NTT_COSET_IFFT 3
int vx_ntt_batch(vx_ctx* ctx, uint64_t* data, int log_n, size_t ncols, int kind, uint64_t shift);
/* Same transform on a DEVICE buffer, results left in the library's native bit-reversed order
 * (no permutation pass); this is the form the benchmark times. dst may equal src. */
int vx_ntt_batch_dev(vx_ctx* ctx, const uint64_t* src, uint64_t* dst, int log_n, size_t ncols, int kind,
                     uint64_t shift);
/* Poseidon permutation on `count` independent 12-element states (Poseidon::poseidon). */
int vx_poseidon_permute(vx_ctx* ctx, uint64_t* states, size_t count);
/* MerkleTree::new over ROW-MAJOR leaves [n_leaves][width] with hash_or_noop leaves and two_to_one
 * nodes.  digests_out (optional) [n_leaves][4]; cap_out [2^cap_height][4]. */
int vx_merkle_cap(vx_ctx* ctx, const uint64_t* leaves, size_t n_leaves, size_t width, int cap_height,
                  uint64_t* digests_out, uint64_t* cap_out);

/* Element-wise field arithmetic on the device (testing aid for the hand-written Goldilocks primitives):
 * op 0: a*b, 1: a+b, 2: a-b, 3: a*b+c (c = a), 4: a^-1 (b ignored), 5: a*b via the non-canonical path then
 * canonicalised.  Inputs may be non-canonical; outputs canonical. */
int vx_field_op(vx_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);

/* ---- L2: PolynomialBatch handles ------------------------------------------------------------ */
/* PolynomialBatch::from_values (is_coeffs = 0: values on H, natural order) or from_coeffs
 * (is_coeffs = 1: coefficients, natural order).  cols: column-major [ncols][2^log_n]; src_on_device
 * selects whether `cols` is a host pointer or a device pointer obtained from vx_dev_alloc.
 * Afterwards the batch owns device copies of the coefficients, the 2^rate_bits blow-up LDE on the
 * coset 7*H' in bit-reversed row order, all leaf digests and the Merkle tree down to the cap.
 * A host matrix is uploaded in column blocks on a second stream while the transforms of the previous block run
 * (vx_prove does the same with a host witness), so the PCIe transfer is almost entirely hidden. */
int vx_batch_commit(vx_ctx* ctx, const uint64_t* cols, int src_on_device, int log_n, size_t ncols, int rate_bits,
                    int cap_height, int is_coeffs, vx_batch** out);
void vx_batch_free(vx_batch* b);
int vx_batch_cap(vx_batch* b, uint64_t* cap_out /* [2^cap_height][4] */);
/* coefficients of column `col`, natural order, n values */
int vx_batch_coeffs(vx_batch* b, size_t col, uint64_t* out);
/* LDE leaf `row` (bit-reversed row order, i.e. MerkleTree leaf index): ncols values + the Merkle
 * path (log2(8n) - cap_height digests, bottom-up) — MerkleTree::prove + tree.get. */
int vx_batch_open_row(vx_batch* b, size_t row, uint64_t* values_out, uint64_t* path_out);
/* all leaf digests [8n][4] (testing aid) */
int vx_batch_digests(vx_batch* b, uint64_t* out);
/* Download rows [row0, row0+nrows) of the LDE, row-major [nrows][ncols] (testing aid). */
int vx_batch_lde_rows(vx_batch* b, size_t row0, size_t nrows, uint64_t* out);
/* PolynomialCoeffs::to_extension().eval(zeta) for every column: out [ncols][2]. */
int vx_batch_eval_ext(vx_batch* b, const uint64_t zeta[2], uint64_t* out);

/* ---- L2 pieces on caller-owned DEVICE buffers (building blocks of the multi-GPU sharded commitment,
 * vectorx_amd/sharded.py: column-shard iNTT+LDE -> one all-to-all -> row-shard leaf hashing + subtrees).
 * vx_lde_columns_dev: values on H (natural order, column-major [ncols][2^log_n]) -> coset LDE
 *   [ncols][2^(log_n+rate_bits)] in bit-reversed row order (the layout of vx_batch); coeffs_out (optional,
 *   [ncols][2^log_n], bit-reversed coefficient order) receives the interpolated coefficients.
 * vx_hash_rows_dev: MerkleTree::new over `nrows` leaves whose values are column-major with column stride
 *   `col_stride` (u64 elements); writes the 2^cap_height cap digests to cap_out (HOST) and, if tree_out (device,
 *   merkle digest count x 4 u64, see vx_merkle_digest_count) is non-NULL, every level. */
int vx_lde_columns_dev(vx_ctx* ctx, const uint64_t* values, int log_n, size_t ncols, int rate_bits, uint64_t* lde_out,
                       uint64_t* coeffs_out);
int vx_hash_rows_dev(vx_ctx* ctx, const uint64_t* cols, size_t col_stride, size_t nrows, size_t ncols, int cap_height,
                     uint64_t* tree_out, uint64_t* cap_out);
size_t vx_merkle_digest_count(size_t n_leaves, int cap_height);

/* ---- L3: circuits and whole proofs ------------------------------------------------------------
 * vx_circuit_desc carries what plonky2's CommonCircuitData + ProverOnlyCircuitData hold for the hot
 * path (plonk/circuit_data.rs): the configuration, the gate list with its selector grouping
 * (gates/selectors.rs SelectorsInfo), the coset shifts k_is, the public-input targets, and the VALUES
 * of the preprocessed polynomials [selectors.., constants.., sigmas..] on H (column-major, natural
 * row order) from which constants_sigmas_commitment and circuit_digest are derived at load time.
 * Gates with hand-written kernels: NoopGate, ConstantGate, PublicInputGate, ArithmeticGate (base), PoseidonGate.
 * Every other gate is handed over as a constraint program (VX_GATE_PROGRAM below), which vx_circuit_create compiles
 * to native code; the recursive verifier's whole gate set has been exercised that way.  The lookup
 * argument (LookupGate / LookupTableGate, tables in the description's tail) is supported.  Limits: 64 gates per circuit,
 * degree_bits + rate_bits <= 24, 1 <= quotient_degree_factor <= 2^rate_bits (CircuitConfig::max_quotient_degree_factor: 8 in
 * standard_recursion_config, which is what plonky2x builds with; every gate's degree must be <= quotient_degree_factor + 1). */
#define VX_GATE_NOOP 0
#define VX_GATE_CONSTANT 1
#define VX_GATE_PUBLIC_INPUT 2
#define VX_GATE_ARITHMETIC 3
#define VX_GATE_POSEIDON 4
/* 5 = VX_GATE_PROGRAM (below) */
#define VX_GATE_LOOKUP 6       /* LookupGate { num_slots }: wires (2i, 2i+1) = (looking input, output); no gate constraints */
#define VX_GATE_LOOKUP_TABLE 7 /* LookupTableGate { num_slots }: wires (3i, 3i+1, 3i+2) = (looked input, output, multiplicity) */

typedef struct vx_circuit_desc {
  int32_t degree_bits;
  int32_t num_wires, num_routed_wires, num_challenges;  /* 135, 80, 2 */
  int32_t rate_bits, cap_height, pow_bits, num_query_rounds; /* 3, 4, 16, 28 */
  int32_t quotient_degree_factor;                       /* 8 (any value in [1, 2^rate_bits]) */
  int32_t num_gates;
  const int32_t* gate_types;       /* [num_gates], sorted by (degree, id) as CircuitBuilder::build does */
  const int32_t* gate_params;      /* ArithmeticGate: num_ops; ConstantGate: num_consts; else 0 */
  const int32_t* selector_indices; /* selector polynomial of each gate */
  const int32_t* group_starts;     /* SelectorsInfo.groups[selector_indices[g]] = [start, end) */
  const int32_t* group_ends;
  int32_t num_selectors;
  int32_t num_constants;           /* selectors + gate constants (CommonCircuitData::num_constants) */
  const uint64_t* constants_sigmas; /* [num_constants + num_routed_wires][2^degree_bits] values on H */
  const uint64_t* k_is;            /* [num_routed_wires] */
  int32_t num_public_inputs;
  const uint32_t* pi_rows;         /* public input target = wire (pi_rows[i], pi_cols[i]) */
  const uint32_t* pi_cols;
  /* Constraint programs: gates OUTSIDE the native set (plonky2x's U32 gates, CosetInterpolationGate, any custom
   * gate) are supplied by the caller as straight-line programs over the row's wires and constants — the same
   * polynomials the gate's `eval_unfiltered` computes (for extension-field gates: expanded to base-field wires).
   * gate_types[g] = VX_GATE_PROGRAM, gate_params[g] = the gate's degree, program_offsets[g] = word offset of its
   * program inside `programs` (-1 for native gates).  Encoding: see VX_OP_* below.  All three may be NULL/0. */
  int32_t programs_len;            /* uint64 words in `programs` */
  const uint64_t* programs;
  const int32_t* program_offsets;  /* [num_gates] */
  /* ---- values the caller already HOLDS (plonky2 plonk/circuit_data.rs: CommonCircuitData, VerifierOnlyCircuitData,
   * FriParams — what `circuit.prove(&input)` at /root/reference/circuits/header_range.rs:167 carries).  Set the bit in
   * override_flags and the library uses the caller's value VERBATIM in the transcript, the FRI commit phase, the
   * proof layout and vx_verify; leave it clear (a zero-initialised tail) and the library derives the value itself
   * with the rule named below.  Passing them removes the library's own restatement of those rules from the path. */
  uint32_t override_flags;               /* VX_DESC_HAS_* */
  int32_t hiding;                        /* FriParams::hiding = config.zero_knowledge; must be 0 (not supported) */
  uint64_t circuit_digest[4];            /* VX_DESC_HAS_CIRCUIT_DIGEST: VerifierOnlyCircuitData::circuit_digest (else:
                                          * hash_no_pad(constants_sigmas_cap || [degree_bits]), empty domain separator) */
  int32_t num_fri_reduction_arity_bits;  /* VX_DESC_HAS_FRI_ARITIES: FriParams::reduction_arity_bits, each in [1, 4] */
  int32_t num_partial_products;          /* VX_DESC_HAS_NUM_PARTIAL_PRODUCTS: CommonCircuitData::num_partial_products;
                                          * must equal ceil(num_routed_wires / quotient_degree_factor) - 1 or the call fails */
  const int32_t* fri_reduction_arity_bits; /* (else: ConstantArityBits(4, 5) of standard_recursion_config) */
  /* ---- lookup argument (plonky2 v0.2.0 gates/lookup.rs, gates/lookup_table.rs, plonk/vanishing_poly.rs
   * check_lookup_constraints; used by plonky2x's byte / u32 range tables).  num_luts = 0 (a zero tail): no lookups.
   * With lookups the preprocessed columns are [selectors | lookup selectors | gate constants | sigmas]:
   * num_lookup_selectors = 4 + num_luts columns (TransSre, TransLdc, InitSre, LastLdc, then one "ends" selector per
   * table) sit between the gate selectors and the gate constants, and num_constants counts them. */
  int32_t num_luts;
  int32_t num_lookup_selectors;
  const int32_t* lut_lens;        /* [num_luts] entries of each table */
  const uint16_t* lut_inputs;     /* CommonCircuitData::luts, concatenated: (input, output) pairs */
  const uint16_t* lut_outputs;
  const int32_t* lookup_rows;     /* [num_luts][3] = (last_lu_gate, last_lut_gate, first_lut_gate): ProverOnlyCircuitData::lookup_rows */
} vx_circuit_desc;
#define VX_DESC_HAS_CIRCUIT_DIGEST 1u
#define VX_DESC_HAS_FRI_ARITIES 2u
#define VX_DESC_HAS_NUM_PARTIAL_PRODUCTS 4u

/* One instruction = one uint64:  op | dst << 8 | a << 16 | b << 32   (VX_OP_LDI is followed by one immediate word).
 * Registers r0..r63 hold field elements.  Every VX_OP_PUSH emits the next constraint of the gate, in order. */
#define VX_GATE_PROGRAM 5
#define VX_PROGRAM_REGS 64
#define VX_OP_END 0
#define VX_OP_LDW 1  /* r[dst] = local_wires[a] */
#define VX_OP_LDC 2  /* r[dst] = local_constants[num_selectors + a]   (the gate's own constants) */
#define VX_OP_LDI 3  /* r[dst] = immediate (next word, canonical) */
#define VX_OP_ADD 4  /* r[dst] = r[a] + r[b] */
#define VX_OP_SUB 5  /* r[dst] = r[a] - r[b] */
#define VX_OP_MUL 6  /* r[dst] = r[a] * r[b] */
#define VX_OP_PUSH 7 /* constraint <- r[a] */
#define VX_OP_LDP 8  /* r[dst] = public_inputs_hash[a] */
#define VX_OP_LDN 9  /* AIR programs only (vx_stark_*): r[dst] = next_row_values[a]; VX_OP_LDW reads the local row, VX_OP_LDP public input a */
/* AIR programs: the b field of VX_OP_PUSH is the constraint kind (starky ConstraintConsumer) */
#define VX_AIR_ALL_ROWS 0   /* constraint(c) */
#define VX_AIR_TRANSITION 1 /* constraint_transition(c): c * (x - g^-1) */
#define VX_AIR_FIRST_ROW 2  /* constraint_first_row(c):  c * L_first(x) */
#define VX_AIR_LAST_ROW 3   /* constraint_last_row(c):   c * L_last(x) */
#define VX_INS(op, dst, a, b) ((uint64_t)(op) | ((uint64_t)(dst) << 8) | ((uint64_t)(a) << 16) | ((uint64_t)(b) << 32))

/* CircuitBuilder::build's prover-side tail: commit the preprocessed polynomials (kept resident in HBM
 * for every later proof of this circuit) and derive circuit_digest. */
int vx_circuit_create(vx_ctx* ctx, const vx_circuit_desc* desc, vx_circuit** out);
void vx_circuit_free(vx_circuit* c);
/* One rehearsal proof of an all-zero witness, discarded: primes the context's buffer pool with every shape a proof of this circuit
 * needs and loads its program-gate kernels, so that the first real vx_prove costs what the later ones do.  vx_circuit_create and
 * vx_circuit_load call it themselves (best effort; VX_NO_WARM_ON_LOAD=1 turns that off); a host with several contexts per GPU calls
 * it once per context.  Skipped (VX_OK) for circuits with lookup tables. */
int vx_circuit_warm(vx_ctx* ctx, vx_circuit* circuit);

/* `.vxcircuit`: the library's self-describing container for a compiled circuit — the role plonky2x's ./build/<name>.circuit files
 * play for `build` / `prove` (/root/reference/succinct.json:7-8,17-18; save -> load round trip as in
 * `circuit.test_serializers`, /root/reference/circuits/header_range.rs:117-126).  Layout: vectorx_amd/csrc/circuit_io.h.
 * All four are host code (no vx_ctx, no GPU); every description is validated exactly like vx_circuit_create's.
 *   vx_circuit_serialize: description (+ the 2^cap_height x 4 constants_sigmas cap when non-NULL = verifier data,
 *     + the preprocessed values when with_preprocessed) -> bytes; *len is capacity in / size out.
 *   vx_circuit_parse: bytes -> a description owned by the library (vx_circuit_desc_free); its constants_sigmas points INTO
 *     `bytes` when that buffer is 8-byte aligned, so `bytes` must outlive the description; *cap_out = the stored cap or NULL.
 *   vx_circuit_load = parse + vx_circuit_create (the `prove` side of the CLI contract). */
size_t vx_circuit_serialized_size(const vx_circuit_desc* desc, int with_cap, int with_preprocessed);
int vx_circuit_serialize(const vx_circuit_desc* desc, const uint64_t* constants_sigmas_cap, int with_preprocessed, uint8_t* out, size_t* len);
int vx_circuit_parse(const uint8_t* bytes, size_t len, const vx_circuit_desc** desc_out, const uint64_t** cap_out);
void vx_circuit_desc_free(const vx_circuit_desc* desc);
int vx_circuit_load(vx_ctx* ctx, const uint8_t* bytes, size_t len, vx_circuit** out);
int vx_circuit_digest(vx_circuit* c, uint64_t digest_out[4]);
int vx_circuit_constants_sigmas_cap(vx_circuit* c, uint64_t* cap_out /* [2^cap_height][4] */);

/* Constraint programs are compiled to native gfx950 code when the circuit is created (hiprtc; a gate whose program
 * cannot be compiled — hiprtc missing, VX_NO_JIT=1 — is evaluated by the on-GPU interpreter instead, with identical
 * results).  Compiled code objects are cached per process, and across processes when the environment variable
 * VX_JIT_CACHE_DIR names a writable directory.  Reports how many VX_GATE_PROGRAM gates the circuit has, how many of
 * them were compiled, and why the others were not (note_out, optional). */
int vx_circuit_program_gates(vx_circuit* c, int* total_out, int* compiled_out, char* note_out, size_t note_cap);

/* plonky2::plonk::prover::prove_with_partition_witness.  `wires` is the finished witness matrix,
 * column-major [num_wires][2^degree_bits] (Witness.wire_values), host pointer or (wires_on_device)
 * a device pointer from vx_dev_alloc.  pow_witness_hint (optional): use this FRI proof-of-work witness
 * instead of grinding — upstream picks it with a parallel find_any, so it is not reproducible run to
 * run; without a hint the SMALLEST valid witness is chosen (deterministic).  The proof is written in
 * plonky2's `ProofWithPublicInputs::to_bytes` layout (util/serialization) into
 * out_buf; *out_len is capacity on entry and the byte length on return (VX_E_INVALID + required size
 * in *out_len if too small). */
int vx_prove(vx_ctx* ctx, vx_circuit* circuit, const uint64_t* wires, int wires_on_device,
             const uint64_t* pow_witness_hint, uint8_t* out_buf, size_t* out_len);
/* Upper bound of the proof size in bytes for this circuit. */
size_t vx_proof_size_bound(vx_circuit* circuit);

/* plonky2::plonk::circuit_data::CircuitData::verify (plonk/verifier.rs + fri/verifier.rs) on a serialised
 * ProofWithPublicInputs — the call the reference makes after every prove
 * (/root/reference/circuits/header_range.rs:167-170).  Host code (milliseconds; the reference verifies on the CPU too):
 * it uses only the verifier's view of the circuit — parameters, gate list, k_is, the constants_sigmas cap and the
 * digest.  Returns VX_OK, or VX_E_PROOF with the failing check in vx_last_error(). */
int vx_verify(vx_circuit* circuit, const uint8_t* proof, size_t proof_len);
/* The same check from verifier data alone — no vx_ctx, no GPU: `desc` as for vx_circuit_create (its constants_sigmas
 * may be NULL: a verifier never sees the preprocessed polynomials) plus the 2^cap_height x 4 constants_sigmas cap
 * (vx_circuit_constants_sigmas_cap of the prover's circuit = plonky2's VerifierOnlyCircuitData); the circuit digest is
 * recomputed from the cap. */
int vx_verify_standalone(const vx_circuit_desc* desc, const uint64_t* constants_sigmas_cap, const uint8_t* proof, size_t proof_len);

/* ---- STARKs on the same primitives (SURVEY.md §8 f-3 scoping spike; vectorx_amd/csrc/stark.hip.h) -----------------------------
 * The Curta / starkyx STARKs inside every real map and outer proof (BLAKE2b: /root/reference/circuits/builder/header.rs:18,
 * SHA-256 / Ed25519: circuits/builder/justification.rs:140-156, 237) are proven with the primitives above plus an AIR
 * constraint evaluator over two adjacent rows.  The AIR arrives as a constraint program: VX_OP_LDW = local row, VX_OP_LDN =
 * next row, VX_OP_LDP = public input, PUSH kind = VX_AIR_*.  Transcript and quotient follow plonky2's `starky` prover
 * (trace cap -> alphas -> quotient on the coset of size n * 2^ceil(log2(max(1, degree - 1))) -> zeta -> openings at zeta
 * and g zeta -> FRI); no permutation / cross-table arguments.  Proof bytes: trace_cap | quotient_cap | local | next | quotient
 * openings | FriProof | public inputs.  vx_stark_verify is host code. */
typedef struct vx_stark_desc {
  int32_t degree_bits, num_columns, num_public_inputs;
  int32_t rate_bits, cap_height, pow_bits, num_query_rounds, num_challenges; /* starky standard_fast_config: 1, 4, 16, 84, 2 */
  int32_t constraint_degree;              /* quotient_degree_factor = max(1, constraint_degree - 1) */
  int32_t program_len;
  const uint64_t* program;
  uint32_t override_flags;                /* VX_DESC_HAS_FRI_ARITIES (else ConstantArityBits(4, 5)) | VX_STARK_OPENINGS_DIGEST */
  int32_t num_fri_reduction_arity_bits;
  const int32_t* fri_reduction_arity_bits;
  /* A SECOND COMMITMENT ROUND (0 / 0 = none): after the trace cap the prover draws num_aux_challenges challenges and
   * commits num_aux_columns more columns that may depend on them — where starky commits its permutation Z polynomials and
   * Curta its lookup / bus accumulators.  The AIR program addresses them as columns num_columns .. num_columns +
   * num_aux_columns - 1 (VX_OP_LDW / VX_OP_LDN) and reads the challenges with VX_OP_LDCH.  The columns themselves are
   * witness data (running sums / products over rows): the CALLER computes them between vx_stark_begin and vx_stark_finish.
   * (As in starky, first- and last-row constraints are multiplied by a Lagrange selector of degree n - 1: their own degree
   * must stay <= constraint_degree - 1, or the quotient no longer fits its quotient_degree_factor chunks.) */
  int32_t num_aux_columns, num_aux_challenges;
  /* Values the prover announces AFTER the second commitment (0 = none): closing sums of lookup / bus accumulators — what
   * starky's cross-table lookups call ctl_zs_last.  They are observed into the transcript after the aux cap, travel at the end
   * of the proof, and the AIR program reads them with VX_OP_LDP at indices num_public_inputs .. num_public_inputs +
   * num_aux_public_inputs - 1 (typically in a last-row constraint  acc = closing sum). */
  int32_t num_aux_public_inputs;
} vx_stark_desc;
/* VX_STARK_OPENINGS_DIGEST (this library's own option, round 5): the transcript absorbs a TREE HASH of the opening set instead of the
 * opening set — the sequence [trace(zeta), aux(zeta), quotient(zeta), trace(g zeta), aux(g zeta)] (each F_p^2 value as two elements) is
 * zero-padded to 8 * 2^k elements (k >= 1), every 8 elements are a leaf hashed with hash_no_pad (one permutation), a two_to_one binary
 * tree over the leaf digests gives the root, and the four elements of the root are observed.  The prover computes it on the device next
 * to the evaluations; without the option a table of ~1000 columns costs ~1000 dependent Poseidon permutations on ONE host core at this
 * point of every proof (1.2 ms of a 10.4 ms SHA-256 table proof, the GPU idle).  The proof's bytes do not change shape; the challenges
 * after the openings do.  Prover and verifier must agree on the flag (it is part of the description). */
#define VX_STARK_OPENINGS_DIGEST 8u
#define VX_OP_LDCH 10 /* AIR programs only: r[dst] = aux challenge a */
int vx_stark_prove(vx_ctx* ctx, const vx_stark_desc* desc, const uint64_t* trace /* [num_columns][2^degree_bits] */, int trace_on_device,
                   const uint64_t* public_inputs, const uint64_t* pow_witness_hint, uint8_t* out_buf, size_t* out_len); /* num_aux_columns = 0 */
/* The two-round form: vx_stark_begin commits the trace and returns the aux challenges; vx_stark_finish takes the aux
 * columns [num_aux_columns][2^degree_bits] (host or device) and completes the proof:
 *   trace_cap | aux_cap | quotient_cap | trace(zeta) | trace(g zeta) | aux(zeta) | aux(g zeta) | quotient(zeta) | FriProof | public inputs.
 * With num_aux_columns = 0 the pair is equivalent to vx_stark_prove.  A session is used for one proof and then freed —
 * BEFORE its context is destroyed (it holds the trace commitment in that context's memory); a vx_stark_finish that fails
 * (e.g. output buffer too small: *out_len then holds the size needed) leaves the session usable. */
typedef struct vx_stark_session vx_stark_session;
int vx_stark_begin(vx_ctx* ctx, const vx_stark_desc* desc, const uint64_t* trace, int trace_on_device, const uint64_t* public_inputs,
                   uint64_t* aux_challenges_out /* [num_aux_challenges] */, vx_stark_session** out);
int vx_stark_finish(vx_stark_session* session, const uint64_t* aux_columns, int aux_on_device, const uint64_t* pow_witness_hint,
                    uint8_t* out_buf, size_t* out_len);
/* ONE STARK proof across `world` GPUs, split by LDE coset exactly like vx_prove_sharded (declared below, with vx_allgather_fn): every
 * rank holds the whole trace (and later the whole second-round columns), extends and hashes only ITS cosets, the owners of the
 * quotient domain's blocks evaluate the constraints there, and the ranks meet in small all-gathers (caps, quotient coset coefficients,
 * first FRI layer, query openings).  world a power of two <= 2^rate_bits (starky's rate_bits = 1: two ranks) and <= 2^cap_height.
 * vx_stark_finish / vx_stark_finish2 of such a session return the full proof on every rank, byte-identical to the unsharded one. */
typedef int (*vx_allgather_fn)(void* user, void* dev_buf, size_t bytes_per_rank);
int vx_stark_begin_sharded(vx_ctx* ctx, const vx_stark_desc* desc, const uint64_t* trace, int trace_on_device, const uint64_t* public_inputs,
                           int rank, int world, vx_allgather_fn allgather, void* user, uint64_t* aux_challenges_out, vx_stark_session** out);
void vx_stark_session_free(vx_stark_session* session);
/* The second-round columns ON THE DEVICE (round 4; vectorx_amd/csrc/aux.hip.h).  Lookup / bus columns have one shape everywhere:
 * FRACTIONS num(row) / den(row) of small expressions of the row's trace values and the challenges — a log-derivative helper
 * 1/(g - x) + 1/(g - y) = ((g - x) + (g - y)) / ((g - x)(g - y)), a table term mult / (g - t), a bus term flag / (g - tuple) — and RUNNING
 * SUMS over the rows of signed combinations of them.  The expressions arrive as one more program: VX_OP_LDW = the row's trace value,
 * VX_OP_LDCH = challenge, LDI / ADD / SUB / MUL; the 2 k-th and (2 k + 1)-th VX_OP_PUSH give numerator and denominator of fraction k
 * (a zero denominator yields 0).  Running sum j holds, on row i, the sum over rows 0 .. i-1 of  sum_k sum_coeffs[j][k] * fraction_k(row)
 * (coefficients in {-1, 0, +1}); its value on the LAST row comes back in closing_sums_out[j] (what a bus announces as its closing sum).
 * `trace_dev` = [num_columns][2^degree_bits] device memory, natural row order (the buffer vx_stark_begin was given); `out_dev` =
 * [num_fractions + num_sums][2^degree_bits] device memory: fraction k lands in column fraction_out[k], sum j in sum_out[j] (NULL: fractions
 * first, then sums) — hand it to vx_stark_finish2 with aux_on_device = 1.  An AIR whose second round is repeated per challenge set
 * calls this once per set with that set's challenges and an output pointer offset by the set's columns. */
typedef struct vx_aux_desc {
  int32_t num_columns, num_challenges, num_fractions, num_sums;
  int32_t program_len;
  const uint64_t* program;
  const int8_t* sum_coeffs;        /* [num_sums][num_fractions] */
  const int32_t* fraction_out;     /* [num_fractions] or NULL */
  const int32_t* sum_out;          /* [num_sums] or NULL */
} vx_aux_desc;
int vx_stark_aux_columns(vx_ctx* ctx, const vx_aux_desc* desc, const uint64_t* trace_dev, int degree_bits, const uint64_t* challenges,
                         uint64_t* out_dev, uint64_t* closing_sums_out);
/* The program is compiled to native gfx950 code with hiprtc on first use (same cache as gate / AIR programs; VX_NO_JIT=1 keeps the
 * on-GPU interpreter); this compiles it ahead of time, without a GPU: 1 = compiled now, 0 = already cached, negative VX_E_*. */
int vx_stark_aux_precompile(const vx_aux_desc* desc);
int vx_stark_verify(const vx_stark_desc* desc, const uint64_t* public_inputs, const uint8_t* proof, size_t proof_len);
/* ---- trace generation for the hash-chip tables, on the device (round 5).  The reference's witness generation for these tables is
 * Curta's (starkyx, un-vendored): `curta_blake2b_variable` over a map job's 8 headers (/root/reference/circuits/builder/header.rs:14-19),
 * `curta_sha256` over the data-root / state-root trees and the authority-set chain (builder/subchain_verification.rs:148-231,
 * builder/justification.rs:140-156), SHA-512 inside the EdDSA gadget (justification.rs:237-243).  These entry points fill the
 * corresponding table of THIS repository's AIRs (vectorx_amd/{sha256,sha512,blake2b_bytes}_air.py: same column maps) for `num_msgs`
 * messages hashed one after the other — `msgs` holds them back to back, message i = bytes [offsets[i], offsets[i + 1]) — directly
 * into `trace_dev` = [columns][2^degree_bits] device memory (1024 / 1995 / 775 columns), the buffer vx_stark_begin then takes with
 * trace_on_device = 1.  The host pads and walks the chain of chaining values; every cell is written by a device thread.
 * public_inputs_out (8 / 16 / 8 values, may be NULL) = what the table's AIR takes as public inputs; digests_out (32 / 64 / 32 bytes per
 * message, may be NULL) = the digests.  VX_E_INVALID when a message does not complete inside the rows (BLAKE2b: degree_bits >= 16). */
int vx_trace_sha256(vx_ctx* ctx, int degree_bits, const uint8_t* msgs, const uint64_t* offsets, int num_msgs, void* trace_dev,
                    uint64_t* public_inputs_out, uint8_t* digests_out);
int vx_trace_sha512(vx_ctx* ctx, int degree_bits, const uint8_t* msgs, const uint64_t* offsets, int num_msgs, void* trace_dev,
                    uint64_t* public_inputs_out, uint8_t* digests_out);
int vx_trace_blake2b(vx_ctx* ctx, int degree_bits, const uint8_t* msgs, const uint64_t* offsets, int num_msgs, void* trace_dev,
                     uint64_t* public_inputs_out, uint8_t* digests_out);
/* The SHA-512 table's BUS variant (sha512_air.py, bus = True: 2012 columns): the same table plus the first 64 bytes of every message —
 * for an Ed25519 verification R || A — latched as 16 little-endian 32-bit words and the first-block flag; that table sends
 * (R, A, digest) on the signature bus (vectorx_amd/sig_link_air.py). */
int vx_trace_sha512_bus(vx_ctx* ctx, int degree_bits, const uint8_t* msgs, const uint64_t* offsets, int num_msgs, void* trace_dev,
                        uint64_t* public_inputs_out, uint8_t* digests_out);
/* The batched EdDSA table (vectorx_amd/eddsa_air.py, Layout(limb_bits = 16, scalar_bits, full)): one instance of 16 + 42 * scalar_bits + 4
 * rows per signature equation [S]B - [h]A (the FULL program, full = 1, scalar_bits = 256: 32 + 42 * 256 + 11 rows — it also decompresses
 * A and R, reduces the SHA-512 digest mod L and checks S < L inside the instance).  `sigs` = [num_sigs][16] little-endian 64-bit words:
 * A.x, A.y (affine), S, h — full: [num_sigs][24], followed by the 512-bit digest (low half, high half; h must be digest mod L).  The rows
 * that remain hold filler instances (A = B, S = h = 0).  Replaces the witness generation of Curta's
 * `curta_eddsa_verify_sigs_conditional` (/root/reference/circuits/builder/justification.rs:237-243).  trace_dev = [475 | 530 full]
 * [2^degree_bits] at 256-bit scalars; results_out ([num_sigs][2][4] words, may be NULL) = the affine (x, y) every instance arrives at.
 * VX_E_INVALID when an A is not on the curve, a value the full program must find canonical is not (S >= L, ...), or num_sigs exceeds
 * (2^degree_bits - 1) / rows. */
int vx_trace_eddsa(vx_ctx* ctx, int degree_bits, int scalar_bits, int full, const uint64_t* sigs, int num_sigs, void* trace_dev, uint64_t* results_out);
/* The same for the constraint-program gates of a circuit (one kernel per program gate): returns the number compiled now. */
int vx_circuit_precompile(const vx_circuit_desc* desc, int* num_program_gates_out);
/* Compile an AIR program ahead of time: every chunk of the program (jit.hip.h cuts long programs into kernels of ~1200
 * instructions) is compiled with hiprtc into the process cache and, when VX_JIT_CACHE_DIR names a private directory, onto disk
 * — needs NO GPU, so a `build` step can run where the circuits are compiled and the proving host only loads code objects.
 * Returns the number of chunks compiled now (>= 0; chunks found in a cache are not counted), negative VX_E_* when hiprtc is
 * missing or a chunk fails to compile.  *num_chunks_out (may be NULL) = chunks the program has. */
int vx_stark_precompile(const vx_stark_desc* desc, int* num_chunks_out);
/* ---- several tables on ONE bus (the shape of Curta's chips: every chip is its own trace, lookups and the bus run across them).
 * One session per table; the challenges of the cross-table argument must be the same in every table and may only be drawn when
 * EVERY trace is committed:
 *   vx_stark_begin (each table)  ->  vx_stark_session_trace_cap (each)  ->  vx_stark_joint_challenges over all caps  ->
 *   vx_stark_set_aux_challenges (each: replaces the session's own challenges and binds the shared ones into its transcript)  ->
 *   the caller computes each table's second-round columns and closing sums  ->  vx_stark_finish2 (each).
 * Verification: vx_stark_proof_trace_cap (each proof) -> vx_stark_joint_challenges -> vx_stark_verify_shared (each; returns
 * the table's closing sums) -> the caller checks that the closing sums of the bus cancel (send minus receive = 0).
 * `aux_public_inputs` = [num_aux_public_inputs] values (see vx_stark_desc).  vx_stark_finish == vx_stark_finish2 with NULL. */
int vx_stark_session_trace_cap(vx_stark_session* session, uint64_t* cap_out /* [2^cap_height][4] */);
int vx_stark_set_aux_challenges(vx_stark_session* session, const uint64_t* shared_challenges /* [num_aux_challenges] */);
int vx_stark_finish2(vx_stark_session* session, const uint64_t* aux_columns, int aux_on_device, const uint64_t* aux_public_inputs,
                     const uint64_t* pow_witness_hint, uint8_t* out_buf, size_t* out_len);
int vx_stark_joint_challenges(const uint64_t* const* trace_caps, const int32_t* cap_heights, int num_tables, int num_challenges,
                              uint64_t* challenges_out);   /* host: Challenger over [num_tables, cap_0, cap_1, ..] */
int vx_stark_proof_trace_cap(const vx_stark_desc* desc, const uint8_t* proof, size_t proof_len, uint64_t* cap_out);
int vx_stark_verify_shared(const vx_stark_desc* desc, const uint64_t* public_inputs, const uint8_t* proof, size_t proof_len,
                           const uint64_t* shared_challenges /* NULL: the proof's own */, uint64_t* aux_public_inputs_out /* may be NULL */);
/* The whole verifier side of a bus in ONE call: the trace caps out of the proofs -> the joint challenges -> every proof verified with
 * them -> every closing sum (index i of each table's num_aux_public_inputs values; all tables declare the same number of shared
 * challenges and of closing sums) must cancel over the tables mod p.  VX_E_PROOF when a proof is invalid OR the bus does not balance.
 * `closing_sums_out` (may be NULL) = [num_tables][num_aux_public_inputs].
 * vx_stark_verify (no argument for closing sums) accepts a table with num_aux_public_inputs > 0 only when every sum is ZERO —
 * the meaning a table verified on its own can have; a caller that does its own accounting uses vx_stark_verify_shared. */
int vx_stark_verify_bus(const vx_stark_desc* const* descs, const uint64_t* const* public_inputs, const uint8_t* const* proofs,
                        const size_t* proof_lens, int num_tables, uint64_t* closing_sums_out);

/* ---- ONE proof sharded across the GPUs of a node (BASELINE.json configs[3]) -------------------
 * `world` in {1, 2, 4, 8} ranks (one vx_ctx + one copy of the circuit each; world <= 2^rate_bits and
 * <= 2^cap_height) produce the SAME proof bytes as vx_prove, every rank returning the full proof.  The split is by
 * LDE COSET: in this library's bit-reversed row order rank r owns LDE rows [r*N/world, (r+1)*N/world) = whole cosets
 * of the 8n-point domain, so the coset NTTs (PolynomialBatch::from_values' lde_values), leaf hashing and Merkle
 * subtrees (MerkleTree::new), the quotient evaluation (compute_quotient_polys incl. its "next row" neighbour), the
 * opening-proof LDE/combination and the first FRI layer are rank-local.  The ranks meet only in in-place
 * all-gathers of small buffers — the per-rank Merkle cap entries, the per-coset quotient coefficients
 * (2 * 8n * 8 bytes in total), the folded first FRI layer (N/16 F_p^2 values) and the query openings — which the
 * HOST supplies, because the communicator is the host's: RCCL via torch.distributed in the Python mirror, or the
 * in-process vx_group below (host threads + xGMI peer copies).  The callback is invoked on the calling thread with
 * the context's stream idle; `dev_buf` holds world * bytes_per_rank bytes of device memory with slot `rank`
 * filled; on return (0 = ok) every slot must hold the corresponding rank's data. */
/* vx_allgather_fn: declared above, with vx_stark_begin_sharded */
int vx_prove_sharded(vx_ctx* ctx, vx_circuit* circuit, const uint64_t* wires, int wires_on_device, int rank, int world,
                     vx_allgather_fn allgather, void* user, const uint64_t* pow_witness_hint, uint8_t* out_buf,
                     size_t* out_len);

/* In-process rank group: `world` host threads, one vx_ctx (normally one GPU) each.  vx_group_allgather is a
 * vx_allgather_fn whose `user` is the member handle: each rank pulls the other ranks' slots straight out of their
 * buffers with hipMemcpyPeerAsync (point-to-point over xGMI; every link carries one slot), between two barriers. */
typedef struct vx_group vx_group;
int vx_group_create(int world, vx_group** out);
void vx_group_destroy(vx_group* g);
/* Joining enables peer access (hipDeviceEnablePeerAccess, both directions) between the joining context's device and the
 * device of every member that joined before it, so the slots travel over xGMI directly. */
int vx_group_join(vx_group* g, int rank, vx_ctx* ctx, void** member_out); /* member handle is owned by the group */
int vx_group_peer_staged(vx_group* g); /* 1 if some pair of member devices has no peer access (copies are host-staged) */
int vx_group_allgather(void* member, void* dev_buf, size_t bytes_per_rank);
void vx_group_abort(vx_group* g); /* wake every rank waiting in vx_group_allgather with VX_E_COMM (a rank has failed) */
/* A rank that DIED cannot abort: a barrier inside vx_group_allgather gives up after this long (default 120 000 ms), aborts the
 * group and returns VX_E_COMM on every waiting rank.  A host whose rank fails with any other error between two exchanges
 * (vx_prove_sharded returned non-zero) calls vx_group_abort itself — the peers then return at once instead of timing out. */
int vx_group_set_timeout_ms(vx_group* g, long long milliseconds);

#ifdef __cplusplus
}
#endif
#endif /* VXPROVER_H */
