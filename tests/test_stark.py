"""STARK spike, CPU side: the oracle's restated starky prover (oracle/stark.hpp) makes proofs of toy
two-row-window AIRs; the PRODUCT's host verifier (`vx_stark_verify`, csrc/stark.hip.h — written independently of the
oracle) must accept them and reject every tampered byte.  The GPU prover is held to the same bytes in
tests/test_gpu_stark.py."""
import numpy as np
import pytest

import oracle_lib
import vectorx_amd as vx
from stark_airs import cubic, fibonacci, logup, mulchain, mulmod

P = oracle_lib.P


@pytest.mark.parametrize("make,degree_bits,cfg", [(fibonacci, 3, {}), (fibonacci, 6, {}), (fibonacci, 9, dict(rate_bits=2, num_query_rounds=20)),
                                                  (cubic, 5, dict(rate_bits=1)), (cubic, 7, dict(rate_bits=3, num_query_rounds=10, fri_arities=[2, 3])),
                                                  (fibonacci, 8, dict(num_challenges=1, cap_height=2, pow_bits=5)),
                                                  (mulchain, 6, dict(groups=3)), (mulchain, 8, dict(groups=2, rate_bits=2, num_query_rounds=30)),
                                                  (logup, 6, {}), (logup, 8, dict(rate_bits=2, num_query_rounds=30, table_bits=5))])
def test_oracle_stark_proofs_are_accepted_by_the_product_verifier(oracle, make, degree_bits, cfg):
    cfg = dict(pow_bits=6, **cfg) if "pow_bits" not in cfg else cfg
    stark, trace, pis = make(degree_bits, **cfg)
    proof = oracle_lib.stark_prove(oracle, stark, trace, pis)
    stark.verify(pis, proof)
    assert oracle_lib.stark_prove(oracle, stark, trace, pis) == proof          # deterministic (smallest PoW witness)
    rng = np.random.default_rng(5)
    for off in [0, 40, 300, len(proof) // 2, len(proof) - 1, len(proof) - 30] + [int(x) for x in rng.integers(0, len(proof), 25)]:
        bad = bytearray(proof)
        bad[off] ^= 1 << int(rng.integers(0, 8))
        with pytest.raises(vx.VxError):
            stark.verify(pis, bytes(bad))
    with pytest.raises(vx.VxError):
        stark.verify((pis + np.uint64(1)) % np.uint64(P), proof)                 # other public inputs
    for bad in (proof[:-1], proof + b"\0", b""):
        with pytest.raises(vx.VxError):
            stark.verify(pis, bad)


def test_a_trace_that_violates_the_air_cannot_be_proven(oracle):
    stark, trace, pis = fibonacci(5, pow_bits=4)
    t = trace.copy()
    t[1, 17] = (int(t[1, 17]) + 1) % P
    # the quotient is not a polynomial of degree < n: the restated prover still outputs bytes, the verifier rejects them
    try:
        proof = oracle_lib.stark_prove(oracle, stark, t, pis)
    except RuntimeError:
        return
    with pytest.raises(vx.VxError):
        stark.verify(pis, proof)


def test_stark_description_checks(oracle):
    stark, trace, pis = fibonacci(4, pow_bits=3)
    proof = oracle_lib.stark_prove(oracle, stark, trace, pis)
    stark.verify(pis, proof)
    for field, value in [("degree_bits", 0), ("rate_bits", 0), ("num_columns", 0), ("num_query_rounds", 0), ("cap_height", 40),
                         ("constraint_degree", 9), ("num_challenges", 3), ("program_len", 3), ("override_flags", 1)]:
        old = getattr(stark.desc, field)
        setattr(stark.desc, field, value)
        with pytest.raises(vx.VxError) as e:
            stark.verify(pis, proof)
        assert e.value.code == vx.VX_E_INVALID, field
        setattr(stark.desc, field, old)
    stark.verify(pis, proof)


def test_vectorised_mulmod_of_the_trace_generators():
    import random
    random.seed(3)
    xs = [random.randrange(P) for _ in range(3000)] + [0, 1, P - 1, P - 2, 2**32, 2**32 - 1, 2**63, P - 2**32]
    ys = [random.randrange(P) for _ in range(3000)] + [P - 1, P - 1, P - 1, P - 2, 2**32, 2**32 - 1, 2**63, P - 2**32]
    r = mulmod(np.array(xs, dtype=np.uint64), np.array(ys, dtype=np.uint64))
    assert all(int(r[i]) == xs[i] * ys[i] % P for i in range(len(xs)))


def test_two_round_air_lookup_soundness(oracle):
    """The second commitment round (aux columns that depend on challenges drawn after the trace cap): a log-derivative lookup.
    A wrong multiplicity or a value outside the table either cannot be proven or yields a proof the verifier rejects; the
    description checks know the new fields."""
    stark, trace, pis = logup(7, pow_bits=4)
    proof = oracle_lib.stark_prove(oracle, stark, trace, pis)
    stark.verify(pis, proof)
    for col, row, val in [(2, 0, None), (0, 3, 12345), (1, 5, 777)]:
        t = trace.copy()
        t[col, row] = (int(t[col, row]) + 1) % P if val is None else val
        try:
            bad = oracle_lib.stark_prove(oracle, stark, t, pis)
        except RuntimeError:
            continue
        with pytest.raises(vx.VxError):
            stark.verify(pis, bad)
    # aux columns that do not follow the recurrence
    good_fn = stark.aux_fn
    stark.aux_fn = lambda tr, ch: (good_fn(tr, ch) + np.uint64(1)) % np.uint64(P)
    try:
        try:
            bad = oracle_lib.stark_prove(oracle, stark, trace, pis)
        except RuntimeError:
            bad = None
        if bad is not None:
            with pytest.raises(vx.VxError):
                stark.verify(pis, bad)
    finally:
        stark.aux_fn = good_fn
    # description checks
    stark.desc.num_aux_challenges = 0            # the program reads challenge 0
    with pytest.raises(vx.VxError) as e:
        stark.verify(pis, proof)
    assert e.value.code == vx.VX_E_INVALID
    stark.desc.num_aux_challenges = 2            # one per challenge set
    stark.desc.num_aux_columns = 0               # challenges without a second round, and the program reads column 3
    with pytest.raises(vx.VxError) as e:
        stark.verify(pis, proof)
    assert e.value.code == vx.VX_E_INVALID
    stark.desc.num_aux_columns = 2
    stark.verify(pis, proof)


def test_programs_can_be_compiled_ahead_of_time_without_a_gpu():
    """vx_stark_precompile / vx_circuit_precompile: hiprtc needs no device, so the `build` step of a host can leave code objects
    (VX_JIT_CACHE_DIR) for the proving machine.  Long AIR programs are cut into chunks (csrc/jit.hip.h)."""
    import os
    from vectorx_amd.synth import SynthCircuit
    stark, _, _ = cubic(6)
    first, chunks = stark.precompile()
    assert chunks == 1 and first in (0, 1)
    assert stark.precompile() == (0, 1)                       # now in the process cache
    os.environ["VX_JIT_AIR_CHUNK"] = "4"                      # a tiny chunk limit: the 3-constraint program splits, every chunk compiles
    try:
        stark2, _, _ = mulchain(5, groups=2)
        done, chunks = stark2.precompile()
        assert chunks >= 3 and 0 <= done <= chunks
    finally:
        del os.environ["VX_JIT_AIR_CHUNK"]
    sc = SynthCircuit(5, seed=2, poseidon_percent=40, flags=1)
    done, total = vx.circuit_precompile(sc.desc_ptr)
    assert total == 2 and 0 <= done <= 2
    assert vx.circuit_precompile(sc.desc_ptr) == (0, 2)
    plain = SynthCircuit(5, seed=2, poseidon_percent=40)
    assert vx.circuit_precompile(plain.desc_ptr) == (0, 0)
    sc.desc.num_gates = 0
    with pytest.raises(vx.VxError):
        vx.circuit_precompile(sc.desc_ptr)
    # the second-round (aux) programs compile ahead of time too (vx_stark_aux_precompile): one kernel per program
    from vectorx_amd import eddsa_air, sha256_air
    for ap in (sha256_air.make_stark(8).aux_program, eddsa_air.make_stark(eddsa_air.Layout(8, 32), 12).aux_program):
        assert ap.precompile() in (0, 1) and ap.precompile() == 0
    ap = sha256_air.make_stark(8).aux_program
    ap.desc.num_fractions = 0
    with pytest.raises(vx.VxError):
        ap.precompile()


def test_second_round_is_repeated_per_challenge_set(oracle):
    """one base-field challenge gives a lookup / bus ~2^-43 soundness; `Stark` repeats the second round once per challenge
    set (replicate_aux_program).  The repeated program reads set 1's challenge, column and closing sum; the first-round constraint
    is not repeated; corrupting ONLY set 1's accumulator is fatal."""
    stark, trace, pis = logup(6, pow_bits=4)
    assert stark.aux_reps == 2 and stark.desc.num_aux_columns == 2 and stark.desc.num_aux_challenges == 2
    words = [int(stark._prog[i]) for i in range(stark.desc.program_len)]
    ops = []
    i = 0
    while i < len(words):
        w = words[i]
        ops.append((w & 0xFF, (w >> 16) & 0xFFFF))
        i += 2 if (w & 0xFF) == vx.VX_OP_LDI else 1
    assert [a for op, a in ops if op == vx.VX_OP_LDCH] == [0, 1]
    assert sorted(a for op, a in ops if op in (vx.VX_OP_LDW, vx.VX_OP_LDN) and a >= 3) == [3, 3, 4, 4]
    assert sum(1 for op, _ in ops if op == vx.VX_OP_PUSH) == 4 + 3          # 3 second-round constraints repeated, v = pi[0] not
    assert sum(1 for op, _ in ops if op == vx.VX_OP_LDP) == 1
    single = vx.Stark(6, 3, 1, [int(w) for w in vx.replicate_aux_program(words, 3, 1, 1, 1, 0, 1)], constraint_degree=3,
                      num_aux_columns=2, num_aux_challenges=2, aux_reps=1)
    assert single.desc.program_len == len(words)                           # reps = 1 is the identity
    proof = oracle_lib.stark_prove(oracle, stark, trace, pis)
    stark.verify(pis, proof)
    good_fn, calls = stark.aux_fn, [0]

    def second_set_only(tr, ch):
        calls[0] += 1
        a = good_fn(tr, ch)
        return (a + np.uint64(1)) % np.uint64(P) if calls[0] % 2 == 0 else a
    stark.aux_fn = second_set_only
    try:
        try:
            bad = oracle_lib.stark_prove(oracle, stark, trace, pis)
        except RuntimeError:
            bad = None
        if bad is not None:
            with pytest.raises(vx.VxError):
                stark.verify(pis, bad)
    finally:
        stark.aux_fn = good_fn


def test_c_entry_points_look_at_the_closing_sums(oracle):
    """vx_stark_verify used to return VX_OK for a table with closing sums while dropping them.  Now a table verified on
    its own must close at zero, and vx_stark_verify_bus does the whole verifier side of a bus — joint challenges, every proof, the
    balance — behind the C ABI."""
    import ctypes
    import hashlib

    from vectorx_amd import sha256_air as sha
    from vectorx_amd import stark_bus
    cfg = dict(num_query_rounds=12, pow_bits=4)
    sha_stark = sha.make_stark(8, bus=True, **cfg)
    t, pis, digests = sha.generate_trace(8, [b"abc", b""])
    sink_stark, sink_t, sink_pis = sha.make_sink(4, digests, **cfg)
    proofs, _ = oracle_lib.stark_prove_tables(oracle, [(sha_stark, t, pis), (sink_stark, sink_t, sink_pis)])
    sums = vx.stark_verify_bus([(sha_stark, pis), (sink_stark, sink_pis)], proofs)
    assert sums.shape == (2, 2) and all((int(sums[0][i]) + int(sums[1][i])) % P == 0 and int(sums[0][i]) != 0 for i in range(2))
    # a sender alone: every proof of it is valid, its closing sums are not zero -> the plain entry point refuses it
    own = oracle_lib.stark_prove(oracle, sha_stark, t, pis)          # the table's OWN challenges (no bus session)
    assert any(int(s) for s in sha_stark.verify(pis, own))           # valid through vx_stark_verify_shared, which hands the sums back
    lone, _ = oracle_lib.stark_prove_tables(oracle, [(sha_stark, t, pis)])
    buf = np.frombuffer(own, dtype=np.uint8)
    L = vx.lib()
    rc = L.vx_stark_verify(ctypes.cast(sha_stark.desc_ptr, ctypes.c_void_p), np.ascontiguousarray(pis, dtype=np.uint64).ctypes.data, buf.ctypes.data, buf.size)
    assert rc == vx.VX_E_PROOF and b"closing sum" in L.vx_last_error()
    with pytest.raises(vx.VxError, match="cancel"):
        vx.stark_verify_bus([(sha_stark, pis)], lone)
    # valid proofs, unbalanced bus
    wrong = [digests[0], hashlib.sha256(b"not sent").digest()]
    sink2, sink2_t, sink2_pis = sha.make_sink(4, wrong, **cfg)
    proofs2, _ = oracle_lib.stark_prove_tables(oracle, [(sha_stark, t, pis), (sink2, sink2_t, sink2_pis)])
    with pytest.raises(vx.VxError, match="cancel"):
        vx.stark_verify_bus([(sha_stark, pis), (sink2, sink2_pis)], proofs2)
    assert not stark_bus.bus_balanced(stark_bus.verify_tables([(sha_stark, pis), (sink2, sink2_pis)], proofs2))
    # table order is part of the statement
    with pytest.raises(vx.VxError):
        vx.stark_verify_bus([(sink_stark, sink_pis), (sha_stark, pis)], proofs[::-1])
