"""BLAKE2b-256 on bytes + a XOR lookup table (vectorx_amd/blake2b_bytes_air.py — own AIR, not Curta's; the header-hash chip of
/root/reference/circuits/builder/header.rs:18), four G functions per row, 28 rows per block, two half tables: the trace's digests are checked against hashlib, the
constraint program against the trace (every constraint of every row), every column class against a single-cell corruption, and the
STARK pipeline on the CPU through the oracle's prover + the product's host verifier.  The GPU twin is tests/test_gpu_stark.py."""
import hashlib

import numpy as np
import pytest

import oracle_lib
import vectorx_amd as vx
from test_sha256_air import eval_program, violations
from vectorx_amd import blake2b_bytes_air as b2

P = b2.P
C = b2.Cols
CHAL = np.array([0x1234567890ABCDEF % P, 0x0FEDCBA987654321 % P], dtype=np.uint64)
MESSAGES = [b"abc", b"", bytes(range(200)), b"x" * 128, b"y" * 129, bytes([7]) * 1000, b"The quick brown fox jumps over the lazy dog"]


@pytest.fixture(scope="module")
def b16():
    prog, npush = b2.build_program()
    t, pis, digests = b2.generate_trace(16, MESSAGES)
    aux = b2.aux_columns(t, CHAL)
    return prog, npush, t, pis, digests, aux


def test_trace_digests_equal_hashlib(b16):
    prog, npush, t, pis, digests, aux = b16
    assert digests == [hashlib.blake2b(m, digest_size=32).digest() for m in MESSAGES]
    assert digests[0].hex() == "bddd813c634239723171ef3fee98579b94964e3bb1cb3e427262c8c068d52319"      # BLAKE2b-256("abc")
    last = digests[-1]
    assert [int(x) for x in pis] == [int.from_bytes(last[4 * i:4 * i + 4], "little") for i in range(8)]
    assert (C.N, C.NAUX, b2.PERIOD) == (775, 119, 28) and t.shape == (775, 1 << 16)
    with pytest.raises(AssertionError):
        b2.generate_trace(15, [b"abc"])                  # the two 32 768-entry half tables do not fit before the last row


def _near(prog, trace, aux, pis, row, n):
    """the constraints on a window of rows around `row` (they are local: a row and its successor) -> violations inside the window;
    first- / last-row kinds and the window's wrap-around row are left out (the closing of the running sum is checked on its own)"""
    lo, hi = max(0, row - 2), min(n, row + 3)
    cons = eval_program(prog, np.ascontiguousarray(trace[:, lo:hi]), np.ascontiguousarray(aux[:, lo:hi]), CHAL, pis)
    w = hi - lo
    bad = []
    for idx, (kind, v) in enumerate(cons):
        if kind == vx.VX_AIR_ALL_ROWS and v.any():
            bad.append(idx)
        elif kind == vx.VX_AIR_TRANSITION and v[:w - 1].any():
            bad.append(idx)
    return bad


def test_every_constraint_vanishes_and_corrupted_cells_are_caught(b16):
    prog, npush, t, pis, digests, aux = b16
    n = t.shape[1]
    cons = eval_program(prog, t, aux, CHAL, pis)
    assert len(cons) == npush and violations(cons, n) == []
    del cons
    lookup_cols = {c for tp in b2.tuples() for c in tp if c is not None} | {C.tab(k, nm) for k in range(b2.NTAB) for nm in ("TA", "TB", "TC", "MULT")}
    g0 = 28 * 3 + 5                                       # a G row (column step) of the fourth block
    g1 = 28 * 3 + 6                                       # the diagonal step after it
    cells = [(C.SEL + 5, g0), (C.H + 3, g0), (C.HN + 2, 28 * 3 + 26), (C.D + 1, 28 * 5), (C.M + 7, g0), (C.T, g0), (C.F, g0), (C.TB, g0),
             (C.V + 9, 28 * 3 + 25), (C.al(1, 0), g0), (C.cl(2, 1), g1), (C.k(0, 0), g0), (C.k(3, 5), g1), (C.f(2, "TOP", 3), g0), (C.tab(0, "BITS", 3), 1003),
             # columns that take part in a lookup (second-round columns recomputed honestly for each)
             (C.f(0, "BIN", 2), g0), (C.f(1, "DIN", 5), g1), (C.f(2, "A1", 0), g0), (C.f(1, "F1", 4), g0), (C.f(3, "E2", 1), g1), (C.f(0, "C2", 2), g0),
             (C.f(3, "B2", 5), g1), (C.fin(1, "FA", 1), 28 * 3 + 25), (C.f(1, "E2", 3), 28 * 3), (C.BY + 2, 28 * 3 + 4), (C.tab(0, "TC"), 1002), (C.tab(1, "MULT"), 77)]
    for col, row in cells:
        bad = t.copy()
        bad[col, row] = (int(bad[col, row]) + 1) % P
        a = b2.aux_columns(bad, CHAL) if col in lookup_cols else aux     # honest second-round columns for the corrupted trace
        local = _near(prog, bad, a, pis, row, n)
        closing = int(a[C.NPAIR + b2.NTAB, n - 1]) != 0 or int(a[C.NPAIR + b2.NTAB, 0]) != 0
        assert local or closing, (col, row)
    wrong = pis.copy()
    wrong[0] = (int(wrong[0]) + 1) % P
    last = eval_program(prog, np.ascontiguousarray(t[:, n - 4:]), np.ascontiguousarray(aux[:, n - 4:]), CHAL, wrong)
    assert any(kind == vx.VX_AIR_LAST_ROW and int(v[-1]) for kind, v in last)
    # an XOR result that is NOT the XOR, with the arithmetic repaired around it, is caught by the lookup ALONE: flip one bit of FE on a
    # finalisation row and of FO with it (FO = FE ^ FH still holds) and let HN follow FO — every local relation holds again, only the
    # triple (FA, FB, FE) is outside the table and the running sum cannot close
    bad = t.copy()
    row = 28 * 3 + 25
    bad[C.fin(2, "FE"), row] ^= 1
    bad[C.fin(2, "FO"), row] ^= 1
    a = b2.aux_columns(bad, CHAL)
    local = _near(prog, bad, a, pis, row, n)
    hn_only = {idx for idx in local}
    assert int(a[C.NPAIR + b2.NTAB, n - 1]) != 0                # acc(last row) = 0 fails
    assert len(hn_only) <= 2                              # locally only the HN latch of that word (2 limbs at most) sees the changed FO


