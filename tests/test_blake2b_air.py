"""BLAKE2b-256 at chip density (vectorx_amd/blake2b_air.py — own AIR, not Curta's; the header-hash chip of
/root/reference/circuits/builder/header.rs:18): the trace is checked against hashlib, the constraint program against the trace
(every constraint of every row), every column class against a single-cell corruption, and the STARK pipeline on the CPU through
the oracle's prover + the product's independent host verifier; then three tables (SHA-256 + BLAKE2b senders, one sink) on ONE bus.
The GPU twin is tests/test_gpu_stark.py."""
import hashlib

import numpy as np
import pytest

import oracle_lib
import vectorx_amd as vx
from test_sha256_air import eval_program, violations
from vectorx_amd import blake2b_air as b2
from vectorx_amd import sha256_air as sha

P = b2.P
MESSAGES = [b"abc", b"", bytes(range(200)), b"x" * 128, b"y" * 129, b"The quick brown fox jumps over the lazy dog"]


@pytest.fixture(scope="module")
def blake10():
    prog, npush = b2.build_program()
    t, pis, digests = b2.generate_trace(10, MESSAGES)
    return prog, npush, t, pis, digests


def test_trace_digests_equal_hashlib(blake10):
    prog, npush, t, pis, digests = blake10
    # 1024 rows = 9 whole blocks of 106 rows: "abc" 1, "" 1, bytes(range(200)) 2, 128 x's 1 (a FULL final block), 129 y's 2, fox 1
    assert digests == [hashlib.blake2b(m, digest_size=32).digest() for m in MESSAGES]
    assert b"".join(int(x).to_bytes(4, "little") for x in pis) == hashlib.blake2b(MESSAGES[-1], digest_size=32).digest()
    assert t.shape == (b2.Cols.N, 1024) and npush > 1500 and len(prog) > 10000
    assert b2.Cols.N > 1000                                        # chip density: a thousand columns


def test_every_constraint_vanishes_on_the_trace_and_not_on_a_broken_one(blake10):
    prog, npush, t, pis, _ = blake10
    chal = np.array([0x1234567890ABCDEF % P], dtype=np.uint64)
    aux = b2.aux_columns(t, chal)
    cons = eval_program(prog, t, aux, chal, pis)
    assert len(cons) == npush
    n = t.shape[1]
    assert violations(cons, n) == []
    C = b2.Cols
    bit = lambda w, i: C.BITS + 64 * w + i
    cells = [(bit(b2.W_A, 5), 3), (bit(b2.W_B, 40), 50), (bit(b2.W_C, 63), 97), (bit(b2.W_D, 0), 20), (bit(b2.W_A1, 31), 7),
             (bit(b2.W_D1, 32), 8), (bit(b2.W_C1, 1), 9), (bit(b2.W_B1, 2), 10), (bit(b2.W_A2, 3), 11), (bit(b2.W_D2, 4), 12),
             (bit(b2.W_C2, 60), 13), (bit(b2.W_B2, 61), 14), (C.E1 + 9, 99), (C.V + 5, 30), (C.V + 31, 0 + 107), (C.H + 3, 40),
             (C.HN + 2, 100), (C.HN + 15, 105), (C.D + 1, 300), (C.M + 7, 60), (C.T, 10), (C.F, 10), (C.TB, 250), (C.K + 0, 33),
             (C.K + 3, 34), (C.K + 5, 35), (C.K + 6, 36), (C.BY + 2, 5), (C.BY + 7, 16), (C.SEL + 10, 200), (C.MULT, 3), (C.TBL, 77)]
    for col, row in cells:
        bad = t.copy()
        bad[col, row] = (int(bad[col, row]) + 1) % P
        a = b2.aux_columns(bad, chal) if col in (C.MULT, C.TBL) or C.BY <= col < C.BY + 8 else aux
        assert violations(eval_program(prog, bad, a, chal, pis), n), (col, row)
    wrong = pis.copy()
    wrong[0] ^= 1
    assert violations(eval_program(prog, t, aux, chal, wrong), n)
    # a byte outside the table: the decomposition m = by0 + 256 by1 + .. can still be met (by0 + 256, by1 - 1) and the helper columns
    # recomputed honestly — only the running sum notices: it no longer returns to zero
    bad = t.copy()
    row = 2 * b2.PERIOD + 2                                        # m[1] of bytes(range(200)): bytes 8..15
    assert int(bad[C.BY + 1, row]) >= 1
    bad[C.BY + 0, row] += 256
    bad[C.BY + 1, row] -= 1
    only = violations(eval_program(prog, bad, b2.aux_columns(bad, chal), chal, pis), n)
    assert [(idx, kind) for idx, kind, _ in only] == [(npush - 1, vx.VX_AIR_LAST_ROW)], only


def test_program_shape_and_message_schedule():
    gx, gy = b2._xy_rows()
    assert [len(a) + len(b) for a, b in zip(gx, gy)] == [12] * 16       # every m_j is used once per round
    assert sorted(t for r in gx for t in r) == sorted(t for r in gy for t in r) == list(range(96))
    for role in range(4):
        assert sorted(p[role] for p in b2.PATTERN) == sorted(list(range(4 * role, 4 * role + 4)) * 2)
    w, k = b2.g_words(2**64 - 1, 2**64 - 1, 2**64 - 1, 2**64 - 1, 2**64 - 1, 2**64 - 1)
    assert max(k) == 2 and all(x < 2**64 for x in w)


@pytest.mark.parametrize("degree_bits", [9])
def test_oracle_proves_and_the_product_verifier_accepts(oracle, degree_bits):
    stark = b2.make_stark(degree_bits, num_query_rounds=20, pow_bits=4)
    t, pis, digests = b2.generate_trace(degree_bits, MESSAGES)
    assert digests and digests[0] == hashlib.blake2b(b"abc", digest_size=32).digest()
    proof = oracle_lib.stark_prove(oracle, stark, t, pis)
    stark.verify(pis, proof)
    wrong = pis.copy()
    wrong[3] = (int(wrong[3]) + 1) % P
    with pytest.raises(vx.VxError):
        stark.verify(wrong, proof)
    bad = bytearray(proof)
    bad[len(bad) // 3] ^= 1
    with pytest.raises(vx.VxError):
        stark.verify(pis, bytes(bad))
    # a trace with one wrong carry cannot be proven (the quotient is not a polynomial of the right degree) or is rejected
    tb = t.copy()
    tb[b2.Cols.K + 1, 40] = (int(tb[b2.Cols.K + 1, 40]) + 1) % 3
    try:
        pb = oracle_lib.stark_prove(oracle, stark, tb, pis)
    except Exception:
        pb = None
    if pb is not None:
        with pytest.raises(vx.VxError):
            stark.verify(pis, pb)


def test_three_tables_on_one_bus(oracle):
    """The SHA-256 and the BLAKE2b table both SEND their completed digests, one sink RECEIVES all of them: the bus challenges are
    drawn over the three trace caps, each proof verifies with them, the three closing sums cancel; dropping one received digest
    leaves every proof valid and the bus unbalanced."""
    from vectorx_amd import stark_bus
    cfg = dict(num_query_rounds=12, pow_bits=4)
    sha_stark = sha.make_stark(8, bus=True, **cfg)
    st, spis, sdig = sha.generate_trace(8, [b"abc", b""])
    bl_stark = b2.make_stark(9, bus=True, **cfg)
    bt, bpis, bdig = b2.generate_trace(9, [b"abc", b"", bytes(range(200))])
    assert len(sdig) == 2 and len(bdig) == 3
    n = 8
    rows = [np.frombuffer(d, dtype=">u4").astype(np.uint64) for d in sdig] + [np.array(b2.digest_limbs(d), dtype=np.uint64) for d in bdig]

    def sink(rows_):
        stark, t, _ = sha.make_sink(3, [b"\0" * 32] * len(rows_), **cfg)
        for i, r in enumerate(rows_):
            t[:8, i] = r
        return stark, t, t[:8, 0].copy()

    sink_stark, sink_t, sink_pis = sink(rows)
    assert len(rows) < n
    tables = [(sha_stark, st, spis), (bl_stark, bt, bpis), (sink_stark, sink_t, sink_pis)]
    proofs, shared = oracle_lib.stark_prove_tables(oracle, tables)
    sums = stark_bus.verify_bus([(s, p) for s, _, p in tables], proofs)
    assert all(int(s[0]) != 0 for s in sums) and sum(int(s[0]) for s in sums) % P == 0
    short_stark, short_t, short_pis = sink(rows[:-1])
    tables2 = tables[:2] + [(short_stark, short_t, short_pis)]
    proofs2, _ = oracle_lib.stark_prove_tables(oracle, tables2)
    with pytest.raises(vx.VxError, match="balance"):
        stark_bus.verify_bus([(s, p) for s, _, p in tables2], proofs2)
