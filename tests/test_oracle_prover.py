"""CPU tests of the oracle's whole-proof path, following the reference's own test pattern
(/root/reference/circuits/header_range.rs:167-170, circuits/builder/decoder.rs:263-264:
`circuit.prove(&input)` then `circuit.verify(&proof, ..)`): the restated prover's proofs must be accepted
by the restated verifier, and any tampering must be rejected.  Also pins the synthetic circuit
generator (the caller-side stand-in) and the serialisation layout.
"""
import ctypes

import numpy as np
import pytest

import oracle_lib
from vectorx_amd.synth import SynthCircuit

P = oracle_lib.P


@pytest.fixture(scope="module")
def small(oracle):
    sc = SynthCircuit(5, seed=7, poseidon_percent=50)
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    proof = oc.prove(sc.witness())
    return sc, oc, proof


def test_synthetic_circuit_shape():
    sc = SynthCircuit(6, seed=1, poseidon_percent=50)
    d = sc.desc
    assert (d.num_wires, d.num_routed_wires, d.num_challenges) == (135, 80, 2)        # standard_recursion_config
    assert (d.rate_bits, d.cap_height, d.pow_bits, d.num_query_rounds) == (3, 4, 16, 28)
    assert d.quotient_degree_factor == 8 and d.num_selectors == 2 and d.num_constants == 4
    rc = sc.row_counts()
    assert rc["poseidon"] + rc["arithmetic"] + rc["noop"] + rc["other"] == 64
    k = np.frombuffer((ctypes.c_uint64 * 80).from_address(d.k_is), dtype=np.uint64)
    assert int(k[0]) == 1 and int(k[1]) == 7 and int(k[5]) == 7 ** 5                 # k_is = 7^j
    w = sc.witness()
    assert w.shape == (135, 64) and int(w.max()) < P
    # deterministic in (degree_bits, seed, poseidon_percent)
    assert (SynthCircuit(6, seed=1, poseidon_percent=50).witness() == w).all()
    assert not (SynthCircuit(6, seed=2, poseidon_percent=50).witness() == w).all()


def test_witness_poseidon_rows_match_the_permutation(oracle):
    sc = SynthCircuit(5, seed=3, poseidon_percent=100)
    w = sc.witness()
    sel1 = np.frombuffer((ctypes.c_uint64 * (84 * 32)).from_address(sc.desc.constants_sigmas), dtype=np.uint64).reshape(84, 32)[1]
    rows = [i for i in range(32) if int(sel1[i]) == 4]
    assert len(rows) >= 20
    for r in rows[:6]:
        assert (oracle.poseidon_permute(w[0:12, r])[0] == w[12:24, r]).all()
    # row 2 hashes the public inputs; row 0 (PublicInputGate) carries the hash
    pi = sc.public_inputs()
    assert (w[0:4, 2] == pi).all()
    assert (w[0:4, 0] == oracle.hash_no_pad(pi)).all()


def test_prove_verify_round_trip(small):
    sc, oc, proof = small
    assert oc.verify(proof) == ""
    # public inputs are the tail of the serialised proof (write_proof_with_public_inputs)
    assert np.frombuffer(proof[-32:], dtype="<u8").tolist() == sc.public_inputs().tolist()


def test_verifier_only_circuit_has_the_same_digest_and_verdict(small, oracle):
    """VerifierOnlyCircuitData built from the cap alone: circuit_digest follows the one rule (Circuit::digest_of_cap —
    hash_no_pad(cap || hash_pad(empty separator) || [degree_bits])), recomputed here with the oracle's sponge as an independent spelling"""
    sc, oc, proof = small
    ov = oracle_lib.OracleCircuit(oracle, sc.desc_ptr, verifier_cap=oc.cap())
    assert (ov.digest() == oc.digest()).all()
    assert ov.verify(proof) == ""
    sep = oracle.hash_no_pad(np.array([1, 0, 0, 0, 0, 0, 0, 1], dtype=np.uint64))
    pre = np.concatenate([oc.cap().reshape(-1), sep, np.array([sc.degree_bits], dtype=np.uint64)])
    assert (oracle.hash_no_pad(pre) == oc.digest()).all()


def test_proof_is_deterministic_and_accepts_a_pow_hint(small):
    sc, oc, proof = small
    assert oc.prove(sc.witness()) == proof
    # pow_witness sits just before the 4 public inputs; re-proving with it as a hint reproduces the proof
    pw = int(np.frombuffer(proof[-40:-32], dtype="<u8")[0])
    assert oc.prove(sc.witness(), pow_hint=pw) == proof
    with pytest.raises(RuntimeError):
        oc.prove(sc.witness(), pow_hint=pw + 1 if (pw + 1) % 65536 else pw + 2)  # almost surely invalid


@pytest.mark.parametrize("offset", [0, 100, 3 * 512 + 8, 3 * 512 + 16 * 200, 20000, 60000, -48, -8])
def test_tampered_proofs_are_rejected(small, offset):
    sc, oc, proof = small
    bad = bytearray(proof)
    bad[offset] ^= 1
    assert oc.verify(bytes(bad)) != ""


def test_truncated_or_padded_proofs_are_rejected(small):
    sc, oc, proof = small
    assert oc.verify(proof[:-1]) != ""
    assert oc.verify(proof + b"\0") != ""
    assert oc.verify(b"") != ""


def test_unsatisfied_witness_is_rejected(small, oracle):
    sc, oc, _ = small
    w = sc.witness().copy()
    w[3, 9] = (int(w[3, 9]) + 1) % P
    bad = oc.prove(w)
    assert oc.verify(bad) != ""


@pytest.mark.parametrize("degree_bits,pct", [(3, 50), (4, 0), (6, 100), (7, 30)])
def test_round_trip_other_shapes(oracle, degree_bits, pct):
    sc = SynthCircuit(degree_bits, seed=degree_bits * 10 + pct, poseidon_percent=pct)
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    proof = oc.prove(sc.witness())
    assert oc.verify(proof) == ""
    # a proof for one circuit does not verify against another circuit
    sc2 = SynthCircuit(degree_bits, seed=999, poseidon_percent=(pct + 37) % 100)  # different gate layout
    oc2 = oracle_lib.OracleCircuit(oracle, sc2.desc_ptr)
    assert oc2.verify(proof) != ""


def test_constraint_program_gates(oracle):
    """Gates outside the native set travel as straight-line constraint programs (include/vxprover.h VX_OP_*).
    (1) the interpreter is validated against a native gate: the ArithmeticGate handed over as a program yields a
        byte-identical proof;  (2) ArithmeticExtensionGate + BaseSumGate<2> exist ONLY as programs: proofs verify, a
        non-binary limb or a wrong extension product is rejected."""
    from vectorx_amd.synth import FLAG_ARITH_AS_PROGRAM, FLAG_PROGRAM_GATES
    base = SynthCircuit(5, seed=3, poseidon_percent=50)
    as_prog = SynthCircuit(5, seed=3, poseidon_percent=50, flags=FLAG_ARITH_AS_PROGRAM)
    assert (base.witness() == as_prog.witness()).all()
    p_native = oracle_lib.OracleCircuit(oracle, base.desc_ptr).prove(base.witness())
    p_prog = oracle_lib.OracleCircuit(oracle, as_prog.desc_ptr).prove(as_prog.witness())
    assert p_native == p_prog
    pg = SynthCircuit(6, seed=4, poseidon_percent=40, flags=FLAG_PROGRAM_GATES)
    rc = pg.row_counts()
    assert pg.desc.num_gates == 7 and rc["arithmetic_extension"] >= 1 and rc["base_sum"] >= 1
    oc = oracle_lib.OracleCircuit(oracle, pg.desc_ptr)
    proof = oc.prove(pg.witness())
    assert oc.verify(proof) == ""
    n = 64
    bs_row = n - rc["noop"] - 1                       # last BaseSumGate row
    ext_row = bs_row - rc["base_sum"]                 # last ArithmeticExtensionGate row
    for (col, row) in ((5, bs_row), (0, bs_row), (6, ext_row), (3, ext_row)):
        w = pg.witness().copy()
        w[col, row] = (int(w[col, row]) + 1) % P
        assert oc.verify(oc.prove(w)) != "", (col, row)


def test_higher_degree_program_gates_and_three_selector_groups(oracle):
    """ExponentiationGate{66 bits} (degree 4) and RandomAccessGate{bits 4, 4 copies, 2 extra constants} (degree 5) as
    constraint programs: with the degree-2/3 program gates the 9 gates no longer fit two selector groups
    (gates/selectors.rs: max_degree 9), so this also covers a third selector polynomial."""
    from vectorx_amd.synth import FLAG_MORE_PROGRAM_GATES, FLAG_PROGRAM_GATES
    sc = SynthCircuit(6, seed=11, poseidon_percent=40, flags=FLAG_PROGRAM_GATES | FLAG_MORE_PROGRAM_GATES)
    rc = sc.row_counts()
    assert sc.desc.num_gates == 9 and sc.desc.num_selectors == 3 and sc.desc.num_constants == 5
    assert rc["exponentiation"] >= 1 and rc["random_access"] >= 1
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    w = sc.witness()
    proof = oc.prove(w)
    assert oc.verify(proof) == ""
    n = 64
    ra_row = n - rc["noop"] - 1                                  # last RandomAccessGate row
    exp_row = ra_row - rc["random_access"]                       # last ExponentiationGate row
    # the exponentiation row really computes base^exponent
    bits = [int(w[1 + i, exp_row]) for i in range(66)]
    assert int(w[67, exp_row]) == pow(int(w[0, exp_row]), sum(b << i for i, b in enumerate(bits)), P)
    # the random-access row really looks up claimed = list[index]
    idx = int(w[0, ra_row])
    assert int(w[1, ra_row]) == int(w[2 + idx, ra_row])
    for (col, row) in ((67, exp_row), (70, exp_row), (3, exp_row), (1, ra_row), (0, ra_row), (74, ra_row), (72, ra_row)):
        bad = w.copy()
        bad[col, row] = (int(bad[col, row]) + 1) % P
        assert oc.verify(oc.prove(bad)) != "", (col, row)


def test_recursion_gate_set_as_programs(oracle):
    """The rest of the recursive verifier's gate set — MulExtension, Reducing, ReducingExtension, PoseidonMds and
    CosetInterpolation{4 bits, degree 8} — as constraint programs: with every flag set the circuit has 14 gates in 4
    selector groups (a degree-8 gate must sit alone in its group).  Proofs verify, tampering with any of the new rows
    is rejected, and the CosetInterpolation row really interpolates: evaluation_value is the value at the point of
    the degree-<16 polynomial through the 16 (coset point, value) pairs, recomputed here with Lagrange's formula."""
    from vectorx_amd.synth import FLAG_RECURSION_GATES
    sc = SynthCircuit(7, seed=21, poseidon_percent=40, flags=15)
    d = sc.desc
    assert d.num_gates == 14 and d.num_selectors == 4 and d.num_constants == 6
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    w = sc.witness()
    assert oc.verify(oc.prove(w)) == ""
    rc = sc.row_counts()
    rec = rc["recursion_each"]
    last = 128 - rc["noop"] - 1
    rows = {"coset": last, "mds": last - rec, "redext": last - 2 * rec, "red": last - 3 * rec, "mulext": last - 4 * rec}
    for name, col in (("coset", 35), ("coset", 7), ("coset", 37), ("coset", 41), ("coset", 45), ("mds", 30), ("mds", 3), ("redext", 72),
                      ("redext", 0), ("red", 20), ("red", 60), ("mulext", 4), ("mulext", 13)):
        bad = w.copy()
        bad[col, rows[name]] = (int(bad[col, rows[name]]) + 1) % P
        assert oc.verify(oc.prove(bad)) != "", (name, col)

    # F_p^2 = F_p[X]/(X^2 - 7) in plain Python
    def emul(x, y):
        return ((x[0] * y[0] + 7 * x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)

    def esub(x, y):
        return ((x[0] - y[0]) % P, (x[1] - y[1]) % P)

    def einv(x):
        nrm = pow((x[0] * x[0] - 7 * x[1] * x[1]) % P, P - 2, P)
        return (x[0] * nrm % P, (-x[1]) * nrm % P)
    r = rows["coset"]
    shift = int(w[0, r])
    g16 = pow(7, (P - 1) // 16, P)                               # plonky2's primitive_root_of_unity(4)
    xs = [shift * pow(g16, j, P) % P for j in range(16)]
    vs = [(int(w[1 + 2 * j, r]), int(w[2 + 2 * j, r])) for j in range(16)]
    zeta = (int(w[33, r]), int(w[34, r]))
    acc = (0, 0)
    for j in range(16):
        num, den = (1, 0), 1
        for k in range(16):
            if k != j:
                num = emul(num, esub(zeta, (xs[k], 0)))
                den = den * (xs[j] - xs[k]) % P
        t = emul(vs[j], num)
        dinv = pow(den, P - 2, P)
        acc = ((acc[0] + t[0] * dinv) % P, (acc[1] + t[1] * dinv) % P)
    assert acc == (int(w[35, r]), int(w[36, r]))
    assert FLAG_RECURSION_GATES == 8 and einv((3, 5)) == einv((3, 5))


def test_recursive_verifier_mix_is_what_is_declared_and_proves(oracle):
    """The DAG's stand-in circuits (round 6): the recursive verifier's gate set in the DECLARED row mix.  At 2^12 rows the counts are the
    derivation written next to RECURSIVE_VERIFIER_MIX (28 queries x 95 PoseidonGate rows + the challenger, 196 ReducingGate rows, ...);
    the oracle proves and verifies the circuit, and a witness that breaks one recursion gate's relation is refused."""
    from vectorx_amd.mapreduce import circuit_shape
    from vectorx_amd.synth import MIX_KEYS, RECURSION_FLAGS, RECURSIVE_VERIFIER_MIX
    assert set(RECURSIVE_VERIFIER_MIX) <= set(MIX_KEYS) and sum(RECURSIVE_VERIFIER_MIX.values()) == 912
    assert circuit_shape(False) == {} and circuit_shape(True)["flags"] == RECURSION_FLAGS == 29
    sc = SynthCircuit(12, seed=5, witness_seed=6, **circuit_shape(True))
    rows = sc.gate_rows()
    assert sum(rows.values()) == 4096
    assert rows["PoseidonGate"] == 2776 and rows["ReducingGate"] == 196 and rows["ArithmeticExtensionGate"] == 282
    assert rows["RandomAccessGate"] == 167 and rows["CosetInterpolationGate"] == 57 and rows["ExponentiationGate"] == 28
    assert 300 <= rows["NoopGate"] <= 420                                             # the padding a 2^12-row verifier circuit carries
    assert len(sc.gate_names()) == sc.desc.num_gates == 16 and sc.desc.num_luts == 1
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    w = sc.witness()
    proof = oc.prove(w)
    assert oc.verify(proof) == ""
    # same circuit, other witness values: another proof, same verdict
    sc2 = SynthCircuit(12, seed=5, witness_seed=7, **circuit_shape(True))
    p2 = oc.prove(sc2.witness())
    assert p2 != proof and oc.verify(p2) == ""
    # a family the flags do not enable cannot be in the mix; shares above 1000 are refused
    with pytest.raises(ValueError):
        SynthCircuit(8, seed=1, flags=0, mix={"reducing": 10})
    with pytest.raises(ValueError):
        SynthCircuit(8, seed=1, flags=RECURSION_FLAGS, mix={"poseidon": 1001})
    with pytest.raises(ValueError):
        SynthCircuit(8, seed=1, flags=RECURSION_FLAGS, mix={"no_such_gate": 1})
    # tiny traces still hold every family (one row each)
    tiny = SynthCircuit(5, seed=1, **circuit_shape(True))
    assert min(tiny.gate_rows().get(g, 0) for g in ("CosetInterpolationGate", "PoseidonMdsGate", "LookupGate", "ReducingExtensionGate")) == 1
