"""The product library's verifier (`vx_verify_standalone`, csrc/verifier.h) is host code: it runs HERE, without a GPU,
from verifier data alone (circuit description + constants_sigmas cap).  Proofs come from the oracle's prover; the
product verifier and the oracle's independently restated verifier must agree on every accepted and every tampered
proof.  (The GPU-side twin is tests/test_gpu_verify.py.)"""
import numpy as np
import pytest

import oracle_lib
import vectorx_amd as vx
from vectorx_amd.synth import SynthCircuit

P = oracle_lib.P


def _verdict(sc, cap, proof):
    try:
        vx.verify_standalone(sc.desc_ptr, cap, proof)
        return ""
    except vx.VxError as e:
        assert e.code == vx.VX_E_PROOF
        return str(e)


@pytest.mark.parametrize("degree_bits,flags", [(3, 0), (4, 0), (5, 0), (6, 15), (7, 7), (9, 0)])
def test_product_verifier_accepts_oracle_proofs(oracle, degree_bits, flags):
    sc = SynthCircuit(degree_bits, seed=70 + degree_bits, poseidon_percent=40, flags=flags)
    sc.desc.pow_bits = 6
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    proof = oc.prove(sc.witness())
    assert oc.verify(proof) == ""
    assert _verdict(sc, oc.cap(), proof) == ""


def test_product_and_oracle_verifiers_agree_on_tampering(oracle):
    sc = SynthCircuit(6, seed=8, poseidon_percent=50, flags=5)
    sc.desc.pow_bits = 5
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    cap = oc.cap()
    proof = oc.prove(sc.witness())
    assert _verdict(sc, cap, proof) == ""
    rng = np.random.default_rng(11)
    offsets = [0, 100, 3 * 512 + 8, 3 * 512 + 16 * 200, len(proof) - 1, len(proof) - 33, len(proof) - 41]
    offsets += [int(x) for x in rng.integers(0, len(proof), size=50)]
    for off in offsets:
        bad = bytearray(proof)
        bad[off] ^= 1 << int(rng.integers(0, 8))
        bad = bytes(bad)
        got, want = _verdict(sc, cap, bad), oc.verify(bad)
        assert got != "" and want != "", (off, got, want)
    for bad in (proof[:-1], proof + b"\0", b"", proof[:777]):
        assert _verdict(sc, cap, bad) != ""
    # wrong verifier data: another circuit's cap / a modified cap
    other_sc = SynthCircuit(6, seed=9, poseidon_percent=20, flags=5)          # (kept alive: the description borrows its buffers)
    other = oracle_lib.OracleCircuit(oracle, other_sc.desc_ptr)
    assert _verdict(sc, other.cap(), proof) != ""
    cap2 = cap.copy()
    cap2[3, 1] = (int(cap2[3, 1]) + 1) % P
    assert _verdict(sc, cap2, proof) != ""


def test_unsatisfied_witness_is_rejected_by_the_product_verifier(oracle):
    sc = SynthCircuit(5, seed=2, poseidon_percent=50)
    sc.desc.pow_bits = 4
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    w = sc.witness().copy()
    w[3, 9] = (int(w[3, 9]) + 1) % P
    bad = oc.prove(w)
    assert oc.verify(bad) != "" and _verdict(sc, oc.cap(), bad) != ""


def test_standalone_verifier_argument_checks(oracle):
    sc = SynthCircuit(4, seed=1, poseidon_percent=50)
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    proof = oc.prove(sc.witness())
    L = vx.lib()
    assert L.vx_verify_standalone(None, None, None, 0) == vx.VX_E_INVALID
    keep = sc.desc.num_gates
    sc.desc.num_gates = 0
    with pytest.raises(vx.VxError) as e:
        vx.verify_standalone(sc.desc_ptr, oc.cap(), proof)
    assert e.value.code == vx.VX_E_INVALID
    sc.desc.num_gates = keep
    vx.verify_standalone(sc.desc_ptr, oc.cap(), proof)


def test_caller_supplied_digest_and_fri_arities_drive_the_transcript(oracle):
    """vx_circuit_desc may carry the values the Rust side holds in CommonCircuitData / VerifierOnlyCircuitData
    (circuit_digest, FriParams::reduction_arity_bits, num_partial_products).  When present, prover, oracle and the product
    verifier must all follow THEM, not their own derivation."""
    sc = SynthCircuit(7, seed=31, poseidon_percent=40)
    sc.desc.pow_bits = 5
    base = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    cap = base.cap()
    p_base = base.prove(sc.witness())
    assert _verdict(sc, cap, p_base) == ""
    # 1. a caller-supplied digest: different transcript, verifies only against the same digest
    digest = [(0x1234567 * (i + 1)) % P for i in range(4)]
    sc.set_circuit_digest(digest)
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    assert [int(x) for x in oc.digest()] == digest
    p_dig = oc.prove(sc.witness())
    assert p_dig != p_base and oc.verify(p_dig) == "" and _verdict(sc, cap, p_dig) == ""
    assert _verdict(sc, cap, p_base) != ""                      # the derived-digest proof no longer verifies
    # 2. caller-supplied FRI arities (degree 2^7: derived = [4]; supplied = [2, 1, 3], final poly of 2 coefficients)
    sc.set_fri_reduction_arity_bits([2, 1, 3])
    sc.set_num_partial_products(9)
    oc2 = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    p_ar = oc2.prove(sc.witness())
    assert len(p_ar) != len(p_dig) and oc2.verify(p_ar) == "" and _verdict(sc, cap, p_ar) == ""
    assert oc.verify(p_ar) != "" and _verdict(sc, cap, p_dig) != ""   # each proof only under its own parameters
    # 3. a disagreeing num_partial_products is refused, not silently mis-laid-out
    sc.set_num_partial_products(8)
    with pytest.raises(vx.VxError) as e:
        vx.verify_standalone(sc.desc_ptr, cap, p_ar)
    assert e.value.code == vx.VX_E_INVALID and "num_partial_products" in str(e.value)


def test_malformed_descriptions_are_refused_before_any_read(oracle):
    """every field later used as an index, shift or size is range-checked by the shared validator
    (csrc/desc_check.h) — by vx_verify_standalone here, by vx_circuit_create on the GPU (tests/test_gpu_prover.py)."""
    sc = SynthCircuit(4, seed=3, poseidon_percent=50, flags=1)
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    proof, cap = oc.prove(sc.witness()), oc.cap()
    vx.verify_standalone(sc.desc_ptr, cap, proof)
    d = sc.desc

    def refused(field, value):
        old = getattr(d, field)
        setattr(d, field, value)
        try:
            with pytest.raises(vx.VxError) as e:
                vx.verify_standalone(sc.desc_ptr, cap, proof)
            assert e.value.code == vx.VX_E_INVALID, (field, value, str(e.value))
        finally:
            setattr(d, field, old)

    for field, value in [("cap_height", -1), ("cap_height", 99), ("num_query_rounds", 0), ("num_query_rounds", -5),
                         ("num_query_rounds", 1 << 20), ("num_selectors", 0), ("num_selectors", d.num_constants + 1),
                         ("num_public_inputs", -1), ("num_wires", 0), ("num_routed_wires", d.num_wires + 1),
                         ("degree_bits", 0), ("degree_bits", 40), ("rate_bits", 0), ("num_challenges", 3), ("pow_bits", 64),
                         ("quotient_degree_factor", 4), ("hiding", 1), ("override_flags", 64), ("programs_len", -1)]:
        refused(field, value)
    # per-gate arrays: patch one entry of the int32 arrays in place
    import ctypes
    n = d.num_gates
    for name, bad in [("gate_types", 17), ("gate_types", -1), ("selector_indices", d.num_selectors), ("selector_indices", -1),
                      ("group_starts", n), ("group_ends", 0), ("group_ends", n + 1)]:
        arr = (ctypes.c_int32 * n).from_address(getattr(d, name))
        for g in (0, n - 1):
            old = arr[g]
            arr[g] = bad
            try:
                with pytest.raises(vx.VxError) as e:
                    vx.verify_standalone(sc.desc_ptr, cap, proof)
                assert e.value.code == vx.VX_E_INVALID, (name, g, bad)
            finally:
                arr[g] = old
    # gate parameters: an ArithmeticGate with more ops than wires, a ConstantGate with more constants than exist
    types = (ctypes.c_int32 * n).from_address(d.gate_types)
    params = (ctypes.c_int32 * n).from_address(d.gate_params)
    for g in range(n):
        if types[g] in (vx.VX_GATE_ARITHMETIC, vx.VX_GATE_CONSTANT):
            old = params[g]
            for bad in (-1, 10_000):
                params[g] = bad
                with pytest.raises(vx.VxError) as e:
                    vx.verify_standalone(sc.desc_ptr, cap, proof)
                assert e.value.code == vx.VX_E_INVALID
            params[g] = old
    vx.verify_standalone(sc.desc_ptr, cap, proof)   # restored description still verifies


@pytest.mark.parametrize("degree_bits,flags", [(5, 16), (6, 16 | 1), (8, 16), (9, 16 | 15)])
def test_lookup_argument_oracle_prover_and_product_verifier_agree(oracle, degree_bits, flags):
    """Lookup argument (plonky2 v0.2.0 gates/lookup.rs, gates/lookup_table.rs, vanishing_poly.rs::check_lookup_constraints):
    a circuit with one lookup table + LookupGate rows.  The oracle's prover and verifier and the product's independent
    verifier agree on the valid proof and on every way of breaking the lookup."""
    import ctypes
    from vectorx_amd.synth import FLAG_LOOKUP
    assert flags & FLAG_LOOKUP
    sc = SynthCircuit(degree_bits, seed=90 + degree_bits, poseidon_percent=40, flags=flags)
    sc.desc.pow_bits = 5
    assert sc.desc.num_luts == 1 and sc.desc.num_lookup_selectors == 5
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    cap = oc.cap()
    w = sc.witness()
    proof = oc.prove(w)
    assert oc.verify(proof) == "" and _verdict(sc, cap, proof) == ""
    last_lu, last_lut, first_lut = (ctypes.c_int32 * 3).from_address(sc.desc.lookup_rows)
    # (a) a looked-up pair that is not in the table, (b) a wrong multiplicity, (c) a changed table entry: all unprovable
    for col, row in [(1, last_lu), (3 * 2 + 2, last_lut), (3 * 1 + 1, first_lut)]:
        bad_w = w.copy()
        bad_w[col, row] = (int(bad_w[col, row]) + 1) % P
        bad = oc.prove(bad_w)
        assert oc.verify(bad) != "" and _verdict(sc, cap, bad) != "", (col, row)
    # tampering with the lookup openings (write_opening_set puts them right after plonk_zs_next) is caught by both verifiers
    d = sc.desc
    off_lookup = 3 * (32 << d.cap_height) + 16 * (d.num_constants + 80 + 135 + 2 + 2)
    for off in (off_lookup + 3, off_lookup + 16 * 14 + 9):
        bad = bytearray(proof)
        bad[off] ^= 1
        assert oc.verify(bytes(bad)) != "" and _verdict(sc, cap, bytes(bad)) != ""
    # the table itself is part of the verifier's data: another table refuses the proof
    outs = (ctypes.c_uint16 * 1).from_address(d.lut_outputs)
    outs[0] ^= 1
    assert _verdict(sc, cap, proof) != ""
    outs[0] ^= 1
    assert _verdict(sc, cap, proof) == ""


def test_more_lookup_tables_than_the_prover_holds_are_refused(oracle):
    """the prover's lookup path is sized for VX_MAX_LUTS = 8 tables; a description with more must be
    answered VX_E_INVALID by the one validator instead of overrunning LookupParams in vx_prove."""
    from vectorx_amd.synth import FLAG_LOOKUP
    sc = SynthCircuit(6, seed=77, poseidon_percent=40, flags=FLAG_LOOKUP)
    sc.desc.pow_bits = 5
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    cap, proof = oc.cap(), oc.prove(sc.witness())
    vx.verify_standalone(sc.desc_ptr, cap, proof)
    sc.desc.num_luts, sc.desc.num_lookup_selectors = 9, 13
    with pytest.raises(vx.VxError) as e:
        vx.verify_standalone(sc.desc_ptr, cap, proof)
    assert e.value.code == vx.VX_E_INVALID and "num_luts" in str(e.value)
    sc.desc.num_luts, sc.desc.num_lookup_selectors = 1, 5
    vx.verify_standalone(sc.desc_ptr, cap, proof)


@pytest.mark.parametrize("degree_bits,flags,qdf", [(5, 32, 8), (7, 32 | 15, 8), (8, 32 | 16, 8), (6, 32, 5)])
def test_u32_and_comparison_gates_oracle_prover_and_product_verifier_agree(oracle, degree_bits, flags, qdf):
    """plonky2-u32's U32Arithmetic / U32AddMany / U32Subtraction / U32RangeCheck / Comparison gates as constraint programs
    (what plonky2x's U32Variable ops instantiate: /root/reference/circuits/builder/justification.rs:164-186): both restated
    verifiers accept the honest proof and reject a wrong sum, a wrong borrow, an out-of-range limb and a flipped comparison."""
    sc = SynthCircuit(degree_bits, seed=3100 + degree_bits, poseidon_percent=40, flags=flags, quotient_degree_factor=qdf)
    sc.desc.pow_bits = 4
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    cap, w = oc.cap(), sc.witness()
    proof = oc.prove(w)
    assert oc.verify(proof) == "" and _verdict(sc, cap, proof) == ""
    import ctypes
    d = sc.desc
    n = sc.n
    sels = list((ctypes.c_int32 * d.num_gates).from_address(d.selector_indices))
    offs = list((ctypes.c_int32 * d.num_gates).from_address(d.program_offsets))
    cs = [np.frombuffer((ctypes.c_uint64 * n).from_address(d.constants_sigmas + 8 * n * k), dtype=np.uint64) for k in range(d.num_selectors)]
    u32_rows = [r for r in range(n) if any(offs[g] >= 0 and int(cs[sels[g]][r]) == g for g in range(d.num_gates))]
    assert u32_rows
    hits = 0
    for r in u32_rows[:: max(1, len(u32_rows) // 6)]:
        for col in (3, 4, 30, 2):                          # a result wire, a carry / high wire, a limb, an input / result_bool
            bad_w = w.copy()
            bad_w[col, r] = (int(bad_w[col, r]) + 1) % P
            try:
                bad = oc.prove(bad_w)
            except RuntimeError:
                hits += 1                                   # the quotient does not exist: unprovable
                continue
            assert oc.verify(bad) != "" and _verdict(sc, cap, bad) != "", (r, col)
            hits += 1
    assert hits >= 8
