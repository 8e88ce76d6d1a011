"""Mutation fuzz of the host-side byte consumers of libvxprover.so (no GPU): vx_verify_standalone, vx_circuit_parse and
vx_stark_verify must refuse every mutated input with an error code — no crash, no hang, no acceptance."""
import numpy as np
import pytest

import oracle_lib
import vectorx_amd as vx
from stark_airs import logup
from vectorx_amd.synth import SynthCircuit


@pytest.fixture(scope="module")
def oracle():
    return oracle_lib.load()


def test_mutated_proofs_and_circuit_files_are_refused(oracle):
    rng = np.random.default_rng(5)
    sc = SynthCircuit(6, seed=4, poseidon_percent=40, flags=16 | 1)
    sc.desc.pow_bits = 4
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    proof, cap = oc.prove(sc.witness()), oc.cap()
    blob = vx.circuit_serialize(sc.desc_ptr, cap, True)
    stark, trace, pis = logup(6, pow_bits=4)
    sproof = oracle_lib.stark_prove(oracle, stark, trace, pis)
    vx.verify_standalone(sc.desc_ptr, cap, proof)
    vx.ParsedCircuit(blob)
    stark.verify(pis, sproof)
    for it in range(900):
        kind = it % 3
        b = bytearray([proof, blob, sproof][kind])
        m = int(rng.integers(0, 4))
        if m == 0:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif m == 1:
            b = b[:int(rng.integers(0, len(b)))]
        elif m == 2:
            i = int(rng.integers(0, len(b) - 8))
            b[i:i + 8] = rng.integers(0, 256, 8, dtype=np.uint8).tobytes()
        else:
            b += bytes(int(rng.integers(1, 64)))
        with pytest.raises(vx.VxError):
            if kind == 0:
                vx.verify_standalone(sc.desc_ptr, cap, bytes(b))
            elif kind == 1:
                vx.ParsedCircuit(bytes(b))
            else:
                stark.verify(pis, bytes(b))


def test_mutated_bus_proofs_and_stark_descriptions_are_refused(oracle):
    """The cross-table entry points (vx_stark_verify_shared, vx_stark_proof_trace_cap, vx_stark_joint_challenges) and
    vx_stark_precompile on mutated proofs, wrong shared challenges and damaged descriptions / programs: an error code, never a
    crash, never an acceptance."""
    import ctypes
    from vectorx_amd import sha256_air as sha
    from vectorx_amd import stark_bus
    rng = np.random.default_rng(9)
    digests = [bytes(rng.integers(0, 256, 32, dtype=np.uint8)) for _ in range(3)]
    s1, t1, p1 = sha.make_sink(4, digests, num_query_rounds=8, pow_bits=3)
    s2, t2, p2 = sha.make_sink(3, digests[:2], num_query_rounds=8, pow_bits=3)
    proofs, shared = oracle_lib.stark_prove_tables(oracle, [(s1, t1, p1), (s2, t2, p2)])
    sums = stark_bus.verify_tables([(s1, p1), (s2, p2)], proofs)
    assert len(sums) == 2 and not stark_bus.bus_balanced(sums)        # two receivers, nobody sends: valid proofs, unbalanced bus
    for it in range(300):
        b = bytearray(proofs[it % 2])
        m = it % 4
        if m == 0:
            b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif m == 1:
            b = b[:int(rng.integers(0, len(b)))]
        elif m == 2:
            b += bytes(int(rng.integers(1, 40)))
        st, pi = (s1, p1) if it % 2 == 0 else (s2, p2)
        sh = shared.copy()
        if m == 3:
            sh[int(rng.integers(0, sh.size))] ^= np.uint64(1) << np.uint64(int(rng.integers(0, 63)))   # intact proof, other challenges
        with pytest.raises(vx.VxError):
            st.verify(pi, bytes(b), sh)
    # descriptions: every integer field pushed out of range, program words damaged
    L = vx.lib()
    fields = ["degree_bits", "num_columns", "num_public_inputs", "rate_bits", "cap_height", "pow_bits", "num_query_rounds", "num_challenges",
              "constraint_degree", "program_len", "num_aux_columns", "num_aux_challenges", "num_aux_public_inputs"]
    for f in fields:
        old = getattr(s1.desc, f)
        for bad in (-1, 1 << 30):
            if f == "program_len" and bad > 0:
                continue          # an upper bound only: the program is read up to its END word
            setattr(s1.desc, f, bad)
            assert L.vx_stark_verify_shared(ctypes.cast(s1.desc_ptr, ctypes.c_void_p), p1.ctypes.data, proofs[0], len(proofs[0]), shared.ctypes.data, None) < 0, (f, bad)
            assert L.vx_stark_precompile(ctypes.cast(s1.desc_ptr, ctypes.c_void_p), None) < 0, (f, bad)
        setattr(s1.desc, f, old)
    s1.verify(p1, proofs[0], shared)
    for _ in range(60):
        i = int(rng.integers(0, len(s1._prog)))
        old = s1._prog[i]
        s1._prog[i] = int(rng.integers(0, 1 << 63))
        try:
            s1.verify(p1, proofs[0], shared)
            accepted = True
        except vx.VxError:
            accepted = False
        assert not accepted, i                                    # the program digest is part of the transcript
        s1._prog[i] = old
    s1.verify(p1, proofs[0], shared)
    caps = [s1.proof_trace_cap(proofs[0]), s2.proof_trace_cap(proofs[1])]
    arr = (ctypes.c_void_p * 2)(caps[0].ctypes.data, caps[1].ctypes.data)
    out = np.zeros(3, dtype=np.uint64)
    for hs, nt, nc in [((4, 4), 0, 3), ((4, 4), 65, 3), ((4, 4), 2, 0), ((4, 4), 2, 17), ((-1, 4), 2, 3), ((4, 25), 2, 3)]:
        h = (ctypes.c_int32 * 2)(*hs)
        assert L.vx_stark_joint_challenges(ctypes.cast(arr, ctypes.c_void_p), ctypes.cast(h, ctypes.c_void_p), nt, nc, out.ctypes.data) < 0
    assert L.vx_stark_proof_trace_cap(ctypes.cast(s1.desc_ptr, ctypes.c_void_p), proofs[0], 5, caps[0].ctypes.data) < 0
