"""Loader for libvxsynth.so — the synthetic circuit + witness generator (caller-side stand-in for
plonky2x's CircuitBuilder::build and witness generation; see synth/synth_circuit.cpp for the declared
gate mix).  Plain host code, no GPU, no oracle."""
from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
_SO = _PKG / "libvxsynth.so"


class CircuitDesc(ctypes.Structure):
    """ctypes mirror of `vx_circuit_desc` (include/vxprover.h)."""
    _fields_ = [
        ("degree_bits", ctypes.c_int32), ("num_wires", ctypes.c_int32), ("num_routed_wires", ctypes.c_int32),
        ("num_challenges", ctypes.c_int32), ("rate_bits", ctypes.c_int32), ("cap_height", ctypes.c_int32),
        ("pow_bits", ctypes.c_int32), ("num_query_rounds", ctypes.c_int32),
        ("quotient_degree_factor", ctypes.c_int32), ("num_gates", ctypes.c_int32),
        ("gate_types", ctypes.c_void_p), ("gate_params", ctypes.c_void_p), ("selector_indices", ctypes.c_void_p),
        ("group_starts", ctypes.c_void_p), ("group_ends", ctypes.c_void_p),
        ("num_selectors", ctypes.c_int32), ("num_constants", ctypes.c_int32),
        ("constants_sigmas", ctypes.c_void_p), ("k_is", ctypes.c_void_p),
        ("num_public_inputs", ctypes.c_int32), ("pi_rows", ctypes.c_void_p), ("pi_cols", ctypes.c_void_p),
        ("programs_len", ctypes.c_int32), ("programs", ctypes.c_void_p), ("program_offsets", ctypes.c_void_p),
        # values the caller holds (CommonCircuitData / VerifierOnlyCircuitData); zero-initialised = derive
        ("override_flags", ctypes.c_uint32), ("hiding", ctypes.c_int32), ("circuit_digest", ctypes.c_uint64 * 4),
        ("num_fri_reduction_arity_bits", ctypes.c_int32), ("num_partial_products", ctypes.c_int32),
        ("fri_reduction_arity_bits", ctypes.c_void_p),
        # lookup argument (zero tail = none)
        ("num_luts", ctypes.c_int32), ("num_lookup_selectors", ctypes.c_int32), ("lut_lens", ctypes.c_void_p),
        ("lut_inputs", ctypes.c_void_p), ("lut_outputs", ctypes.c_void_p), ("lookup_rows", ctypes.c_void_p),
    ]


DESC_HAS_CIRCUIT_DIGEST = 1
DESC_HAS_FRI_ARITIES = 2
DESC_HAS_NUM_PARTIAL_PRODUCTS = 4


FLAG_PROGRAM_GATES = 1      # add ArithmeticExtensionGate + BaseSumGate rows, evaluated through constraint programs
FLAG_ARITH_AS_PROGRAM = 2   # hand the ArithmeticGate to the prover as a constraint program instead of the native gate
FLAG_MORE_PROGRAM_GATES = 4  # + ExponentiationGate (degree 4) and RandomAccessGate (degree 5) as programs: 3 selector groups
FLAG_LOOKUP = 16             # + one lookup table (LookupTableGate rows) and LookupGate rows looking values up in it
FLAG_U32_GATES = 32          # + plonky2-u32's U32Arithmetic / U32AddMany / U32Subtraction / U32RangeCheck / Comparison gates as programs (a voting-threshold block)
FLAG_RECURSION_GATES = 8     # + MulExtension, Reducing, ReducingExtension, PoseidonMds, CosetInterpolation{4 bits, degree 8} as programs

# ---- the declared row mix of a recursive verifier -------------------------------------------------------------------------------
# In the reference a reduce job verifies its two children's proofs in-circuit and a map / outer job verifies its Curta STARKs in-circuit
# (/root/reference/circuits/builder/subchain_verification.rs:78, 233-289; builder/header.rs:14-19; justification.rs:236-243), so most
# rows of those circuits are the gates of plonky2's `verify_proof`.  The mix below is a RECOLLECTION-BASED COUNT (no Rust here to print
# `builder.print_gate_counts`) of one `verify_proof` under standard_recursion_config for an inner proof of 2^12 rows — which lands, as
# upstream's does, just under 2^12 rows itself.  Per FRI query (28 of them): leaf hashing of the four initial oracles
# ceil(84/8) + ceil(135/8) + ceil(20/8) + ceil(16/8) = 33 PoseidonGate rows, four Merkle paths of 12 + 3 - 4 = 11 levels = 44, two FRI
# layers (32 base elements per leaf = 4 permutations, paths of 7 and 3 levels) = 18  ->  95 PoseidonGate rows; reducing 255 + 2
# openings with ReducingGate{43} = 7 rows; 2 CosetInterpolationGate rows; the cap / evaluation selections = 6 RandomAccessGate{4 bits}
# rows; 2 BaseSumGate rows (index bits), 1 ExponentiationGate row, ~3 ArithmeticExtensionGate and ~2 ArithmeticGate rows.  Once per
# proof: the challenger (~900 observed elements = ~115 PoseidonGate rows), the gate constraints at zeta (~200 ArithmeticExtensionGate,
# ~30 PoseidonMdsGate, ~20 MulExtensionGate, ~12 ReducingExtensionGate rows), the permutation checks (~24 ArithmeticGate rows).
# Total ~3 730 of 4 096 rows.  plonky2x circuits also carry a lookup table (byte range checks): 4 rows per thousand, declared.
MIX_KEYS = ("poseidon", "arithmetic", "arithmetic_extension", "base_sum", "exponentiation", "random_access", "mul_extension", "reducing",
            "reducing_extension", "poseidon_mds", "coset_interpolation", "lookup")          # the order of vxs_build5's mix_permille
RECURSIVE_VERIFIER_MIX = {"poseidon": 678, "arithmetic_extension": 69, "reducing": 48, "random_access": 41, "arithmetic": 20, "base_sum": 16,
                          "coset_interpolation": 14, "exponentiation": 7, "poseidon_mds": 7, "mul_extension": 5, "reducing_extension": 3,
                          "lookup": 4}                                                     # rows per 1000; the remaining 88 are NoopGate padding
RECURSION_FLAGS = FLAG_PROGRAM_GATES | FLAG_MORE_PROGRAM_GATES | FLAG_RECURSION_GATES | FLAG_LOOKUP   # = 29: the gate families the mix needs


def build() -> Path:
    r = subprocess.run(["make", "-C", str(_PKG / "synth")], capture_output=True, text=True)
    if r.returncode and not _SO.exists():
        raise RuntimeError("building libvxsynth.so failed:\n" + r.stdout + r.stderr)
    return _SO


_lib = None


def _load():
    global _lib
    if _lib is None:
        if not _SO.exists():
            build()
        L = ctypes.CDLL(str(_SO))
        L.vxs_build.restype = ctypes.c_void_p
        L.vxs_build.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_int]
        L.vxs_build2.restype = ctypes.c_void_p
        L.vxs_build2.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64]
        L.vxs_build3.restype = ctypes.c_void_p
        L.vxs_build3.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64, ctypes.c_int]
        L.vxs_build4.restype = ctypes.c_void_p
        L.vxs_build4.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_int]
        L.vxs_build5.restype = ctypes.c_void_p
        L.vxs_build5.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.vxs_gate_rows.restype = ctypes.c_int
        L.vxs_gate_rows.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.vxs_gate_keys.restype = ctypes.c_int
        L.vxs_gate_keys.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.vxs_gate_key_name.restype = ctypes.c_char_p
        L.vxs_gate_key_name.argtypes = [ctypes.c_int]
        L.vxs_row_counts_ext.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.vxs_recursion_rows.restype = ctypes.c_uint64
        L.vxs_recursion_rows.argtypes = [ctypes.c_void_p]
        L.vxs_u32_rows.restype = ctypes.c_uint64
        L.vxs_u32_rows.argtypes = [ctypes.c_void_p]
        L.vxs_patch_public_inputs.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.vxs_free.argtypes = [ctypes.c_void_p]
        L.vxs_desc.restype = ctypes.POINTER(CircuitDesc)
        L.vxs_desc.argtypes = [ctypes.c_void_p]
        L.vxs_witness.restype = ctypes.c_void_p
        L.vxs_witness.argtypes = [ctypes.c_void_p]
        L.vxs_public_inputs.restype = ctypes.c_void_p
        L.vxs_public_inputs.argtypes = [ctypes.c_void_p]
        L.vxs_row_counts.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.vxs_release_host_buffers.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        _lib = L
    return _lib


class SynthCircuit:
    """A synthetic standard_recursion_config circuit with a satisfying witness."""

    def __init__(self, degree_bits: int, seed: int = 0, poseidon_percent: int = 50, witness_seed: int | None = None,
                 flags: int = 0, quotient_degree_factor: int = 8, mix: dict | None = None):
        """`seed` fixes the circuit; `witness_seed` (default = seed) only the witness values; `flags`: FLAG_*;
        `quotient_degree_factor` (CircuitConfig::max_quotient_degree_factor, 3..8): below 7 the circuit has no PoseidonGate;
        `mix`: a declared row mix {MIX_KEYS name: rows per thousand} (e.g. RECURSIVE_VERIFIER_MIX with flags=RECURSION_FLAGS) instead of
        the fixed fractions; every family it names must be enabled in `flags`, the rest of the trace is NoopGate padding."""
        L = _load()
        mixbuf = None
        if mix is not None:
            unknown = set(mix) - set(MIX_KEYS)
            if unknown:
                raise ValueError(f"unknown gate families in the mix: {sorted(unknown)}")
            mixbuf = (ctypes.c_int32 * len(MIX_KEYS))(*[int(mix.get(k, 0)) for k in MIX_KEYS])
        self.mix = dict(mix) if mix is not None else None
        self.flags = flags
        self._h = L.vxs_build5(degree_bits, seed, poseidon_percent, seed if witness_seed is None else witness_seed, flags,
                               quotient_degree_factor, ctypes.cast(mixbuf, ctypes.c_void_p) if mixbuf is not None else None)
        if not self._h:
            raise ValueError("vxs_build rejected the parameters")
        self.degree_bits = degree_bits
        self.n = 1 << degree_bits
        self.desc_ptr = L.vxs_desc(self._h)
        self.desc = self.desc_ptr.contents
        self.num_wires = self.desc.num_wires
        self.seed, self.poseidon_percent = seed, poseidon_percent

    def witness(self) -> np.ndarray:
        """[num_wires][n] column-major view of the generator's buffer (copy before freeing)."""
        p = _load().vxs_witness(self._h)
        buf = (ctypes.c_uint64 * (self.num_wires * self.n)).from_address(p)
        return np.frombuffer(buf, dtype=np.uint64).reshape(self.num_wires, self.n)

    def public_inputs(self) -> np.ndarray:
        p = _load().vxs_public_inputs(self._h)
        return np.frombuffer((ctypes.c_uint64 * 4).from_address(p), dtype=np.uint64).copy()

    def patch_public_inputs(self, pi) -> tuple:
        """New public inputs on the same circuit: returns (row0, row2), the only two witness rows that change."""
        pi = np.ascontiguousarray(pi, dtype=np.uint64)
        r0 = np.empty(self.num_wires, np.uint64)
        r2 = np.empty(self.num_wires, np.uint64)
        _load().vxs_patch_public_inputs(self._h, pi.ctypes.data, r0.ctypes.data, r2.ctypes.data)
        return r0, r2

    def row_counts(self) -> dict:
        out = np.zeros(3, np.uint64)
        _load().vxs_row_counts(self._h, out.ctypes.data)
        ext = np.zeros(7, np.uint64)
        _load().vxs_row_counts_ext(self._h, ext.ctypes.data)
        d = {"poseidon": int(out[0]), "arithmetic": int(out[1]), "noop": int(out[2]), "other": 2}
        if int(ext[3]) or int(ext[4]):
            d["arithmetic_extension"], d["base_sum"] = int(ext[3]), int(ext[4])
        if int(ext[5]) or int(ext[6]):
            d["exponentiation"], d["random_access"] = int(ext[5]), int(ext[6])
        rec = int(_load().vxs_recursion_rows(self._h))
        u32 = int(_load().vxs_u32_rows(self._h))
        if u32:
            d["u32_each"] = u32         # rows of EACH of U32AddMany / U32Subtraction / U32RangeCheck / Comparison; U32Arithmetic has one more
        if rec:
            d["recursion_each"] = rec   # rows of EACH of MulExtension / Reducing / ReducingExtension / PoseidonMds / CosetInterpolation
        return d

    def gate_rows(self) -> dict:
        """{gate name: rows} for every gate that has rows (counted where the selector columns are written)"""
        L = _load()
        out = (ctypes.c_uint64 * 64)()
        k = L.vxs_gate_rows(self._h, out, 64)
        return {L.vxs_gate_key_name(i).decode(): int(out[i]) for i in range(k) if out[i]}

    def gate_names(self) -> list:
        """the gate's name for every gate index of the description (the order the prover's per-gate stages are numbered in)"""
        L = _load()
        out = (ctypes.c_int32 * 64)()
        k = L.vxs_gate_keys(self._h, out, 64)
        return [L.vxs_gate_key_name(out[i]).decode() for i in range(k)]

    # ---- overrides: what a plonky2 caller takes from CommonCircuitData / VerifierOnlyCircuitData (vxprover.h VX_DESC_HAS_*)
    def set_circuit_digest(self, digest) -> None:
        """Use this circuit_digest (4 field elements) instead of the library's own derivation."""
        for i in range(4):
            self.desc.circuit_digest[i] = int(digest[i])
        self.desc.override_flags |= DESC_HAS_CIRCUIT_DIGEST

    def set_fri_reduction_arity_bits(self, arities) -> None:
        """Use FriParams::reduction_arity_bits = `arities` instead of ConstantArityBits(4, 5)."""
        self._arities = (ctypes.c_int32 * max(1, len(arities)))(*arities)   # keep the buffer alive
        self.desc.fri_reduction_arity_bits = ctypes.cast(self._arities, ctypes.c_void_p).value
        self.desc.num_fri_reduction_arity_bits = len(arities)
        self.desc.override_flags |= DESC_HAS_FRI_ARITIES

    def set_num_partial_products(self, npp: int) -> None:
        self.desc.num_partial_products = int(npp)
        self.desc.override_flags |= DESC_HAS_NUM_PARTIAL_PRODUCTS

    def release_host_buffers(self, witness=True, preprocessed=True):
        _load().vxs_release_host_buffers(self._h, int(witness), int(preprocessed))

    def free(self):
        if self._h:
            _load().vxs_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
