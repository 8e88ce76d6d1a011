"""`vx_verify` — the product library's CircuitData::verify (host code, vectorx_amd/csrc/verifier.h) — written like the
reference's own tests (/root/reference/circuits/header_range.rs:167-170, circuits/builder/decoder.rs:263-264:
`circuit.prove(..)` then `circuit.verify(..)`), and cross-checked against the ORACLE's independently restated verifier:
both must accept the same proofs and reject the same tampered ones."""
import numpy as np
import pytest

import oracle_lib
import vectorx_amd as vx
from vectorx_amd.synth import SynthCircuit

pytestmark = pytest.mark.gpu
P = oracle_lib.P


@pytest.mark.parametrize("degree_bits,flags", [(3, 0), (5, 0), (6, 15), (8, 7), (12, 0), (13, 15)])
def test_prove_then_verify(ctx, oracle, degree_bits, flags):
    sc = SynthCircuit(degree_bits, seed=40 + degree_bits, poseidon_percent=45, flags=flags)
    sc.desc.pow_bits = 8
    c = vx.Circuit(ctx, sc.desc_ptr)
    proof = c.prove(sc.witness())
    c.verify(proof)                                                   # raises on failure
    assert oracle_lib.OracleCircuit(oracle, sc.desc_ptr).verify(proof) == ""
    c.free()


def test_verifier_and_oracle_verifier_agree_on_tampered_proofs(ctx, oracle):
    sc = SynthCircuit(6, seed=3, poseidon_percent=50, flags=5)
    sc.desc.pow_bits = 6
    c = vx.Circuit(ctx, sc.desc_ptr)
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    proof = c.prove(sc.witness())
    c.verify(proof)
    rng = np.random.default_rng(5)
    offsets = [0, 31, 100, 3 * 512 + 8, 3 * 512 + 16 * 100, len(proof) - 1, len(proof) - 33, len(proof) - 41]
    offsets += [int(x) for x in rng.integers(0, len(proof), size=60)]
    rejected = 0
    for off in offsets:
        bad = bytearray(proof)
        bad[off] ^= 1 << int(rng.integers(0, 8))
        bad = bytes(bad)
        want = oc.verify(bad)
        try:
            c.verify(bad)
            got = ""
        except vx.VxError as e:
            assert e.code == vx.VX_E_PROOF
            got = str(e)
        assert (got == "") == (want == ""), (off, got, want)
        rejected += got != ""
    assert rejected == len(offsets)                                    # every single-bit flip is caught
    for bad in (proof[:-1], proof + b"\0", b"", proof[:1000]):
        with pytest.raises(vx.VxError):
            c.verify(bad)
    c.free()


def test_proof_of_another_circuit_or_witness_is_rejected(ctx):
    a = SynthCircuit(5, seed=1, poseidon_percent=30)
    b = SynthCircuit(5, seed=2, poseidon_percent=70)
    for s in (a, b):
        s.desc.pow_bits = 4
    ca, cb = vx.Circuit(ctx, a.desc_ptr), vx.Circuit(ctx, b.desc_ptr)
    pa = ca.prove(a.witness())
    ca.verify(pa)
    with pytest.raises(vx.VxError):
        cb.verify(pa)
    w = a.witness().copy()
    w[3, 9] = (int(w[3, 9]) + 1) % P                                   # unsatisfied witness
    try:
        bad = ca.prove(w)
    except vx.VxError:
        bad = None
    if bad is not None:
        with pytest.raises(vx.VxError):
            ca.verify(bad)
    ca.free()
    cb.free()


def test_large_and_sharded_proofs_verify(ctx):
    from vectorx_amd import sharded
    sc = SynthCircuit(16, seed=9, poseidon_percent=50)
    c = vx.Circuit(ctx, sc.desc_ptr)
    w = sc.witness()
    proof = c.prove(w)
    c.verify(proof)
    ctxs = [vx.Context(0) for _ in range(4)]
    circuits = [vx.Circuit(x, sc.desc_ptr) for x in ctxs]
    for p in sharded.prove_sharded_threads(circuits, w):
        c.verify(p)
    for x in circuits:
        x.free()
    for x in ctxs:
        x.close()
    c.free()
