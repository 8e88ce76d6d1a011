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
    other = oracle_lib.OracleCircuit(oracle, SynthCircuit(6, seed=9, poseidon_percent=20, flags=5).desc_ptr)
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
