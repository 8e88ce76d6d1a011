"""SHA-256 at chip density (vectorx_amd/sha256_air.py — own AIR, not Curta's): the trace is checked against
hashlib, the constraint program against the trace (every constraint of every row, vectorised), and the STARK pipeline on the
CPU through the oracle's prover + the product's independent host verifier.  The GPU twin is tests/test_gpu_stark.py."""
import hashlib

import numpy as np
import pytest

import oracle_lib
import stark_airs as airs
import vectorx_amd as vx
from vectorx_amd import sha256_air as sha

P = sha.P
MESSAGES = [b"abc", b"", b"The quick brown fox jumps over the lazy dog" * 3, bytes(range(200))]


def eval_program(words, trace, aux, chal, pis):
    """interpret the AIR program on ALL rows at once (registers = uint64 arrays): -> list of (kind, values[n])"""
    n = trace.shape[1]
    cols = np.concatenate([trace, aux]) if aux is not None else trace
    nxt = np.roll(cols, -1, axis=1)
    R, out, pc = {}, [], 0
    full = lambda v: np.full(n, v % P, dtype=np.uint64)
    while True:
        ins = words[pc]
        op, dst, a, b = ins & 0xFF, (ins >> 8) & 0xFF, (ins >> 16) & 0xFFFF, (ins >> 32) & 0xFFFF
        if op == vx.VX_OP_END:
            return out
        if op == vx.VX_OP_LDW:
            R[dst] = cols[a]
        elif op == vx.VX_OP_LDN:
            R[dst] = nxt[a]
        elif op == vx.VX_OP_LDI:
            pc += 1
            R[dst] = full(words[pc])
        elif op == vx.VX_OP_LDP:
            R[dst] = full(int(pis[a]))
        elif op == vx.VX_OP_LDCH:
            R[dst] = full(int(chal[a]))
        elif op == vx.VX_OP_ADD:
            R[dst] = airs.addmod(R[a], R[b])
        elif op == vx.VX_OP_SUB:
            R[dst] = airs.submod(R[a], R[b])
        elif op == vx.VX_OP_MUL:
            R[dst] = airs.mulmod(R[a], R[b])
        elif op == vx.VX_OP_PUSH:
            out.append((b, R[a]))
        else:
            raise AssertionError(op)
        pc += 1


def violations(cons, n):
    bad = []
    for idx, (kind, v) in enumerate(cons):
        if kind == vx.VX_AIR_ALL_ROWS:
            rows = np.nonzero(v)[0]
        elif kind == vx.VX_AIR_TRANSITION:
            rows = np.nonzero(v[:n - 1])[0]
        elif kind == vx.VX_AIR_FIRST_ROW:
            rows = np.nonzero(v[:1])[0]
        else:
            rows = np.nonzero(v[n - 1:])[0] + n - 1
        if rows.size:
            bad.append((idx, kind, rows[:4].tolist()))
    return bad


def caught_near(prog, trace, aux, chal, pis, row, n) -> bool:
    """Is a trace that differs from a valid one around `row` only refused?  The constraints are local (a row and its successor), so
    they are evaluated on a window around `row`, on the first rows (first-row kinds) and on the last rows (last-row kinds: where a
    running sum has to close) instead of on all n rows — the same verdict as `violations` over the whole trace at a fraction of the
    time.  The window's own wrap-around row is not a transition of the trace and is left out."""
    def part(lo, hi):
        return eval_program(prog, np.ascontiguousarray(trace[:, lo:hi]), np.ascontiguousarray(aux[:, lo:hi]), chal, pis), hi - lo
    cons, w = part(max(0, row - 2), min(n, row + 3))
    for kind, v in cons:
        if (kind == vx.VX_AIR_ALL_ROWS and v.any()) or (kind == vx.VX_AIR_TRANSITION and v[:w - 1].any()):
            return True
    head, _ = part(0, min(n, 4))
    tail, wt = part(max(0, n - 4), n)
    return any(kind == vx.VX_AIR_FIRST_ROW and int(v[0]) for kind, v in head) or any(kind == vx.VX_AIR_LAST_ROW and int(v[wt - 1]) for kind, v in tail)


@pytest.fixture(scope="module")
def sha9():
    prog, npush = sha.build_program()
    t, pis, digests = sha.generate_trace(9, MESSAGES)
    return prog, npush, t, pis, digests


def test_trace_digests_equal_hashlib(sha9):
    prog, npush, t, pis, digests = sha9
    # 512 rows = 7 whole blocks of 66 rows: "abc" (1), "" (1), the fox text (3), bytes(range(200)) (4 -> only 2 fit)
    assert digests == [hashlib.sha256(m).digest() for m in MESSAGES[:3]]
    assert b"".join(int(x).to_bytes(4, "big") for x in pis) == hashlib.sha256(MESSAGES[2]).digest()
    assert t.shape == (sha.Cols.N, 512) and npush > 2000 and len(prog) > 10000


def test_every_constraint_vanishes_on_the_trace_and_not_on_a_broken_one(sha9):
    prog, npush, t, pis, _ = sha9
    chal = np.array([0x1234567890ABCDEF % P], dtype=np.uint64)
    aux = sha.aux_columns(t, chal)
    cons = eval_program(prog, t, aux, chal, pis)
    assert len(cons) == npush
    assert violations(cons, t.shape[1]) == []
    C = sha.Cols
    for col, row in [(C.S + 5, 70), (C.WB + 32 * 3 + 1, 20), (C.CA, 33), (C.H + 2, 100), (C.NF, 65), (C.X1 + 7, 12), (C.MULT, 3),
                     (C.SEL + 10, 200), (C.D + 1, 300), (C.FFC + 3, 64)]:
        bad = t.copy()
        bad[col, row] = (int(bad[col, row]) + 1) % P
        assert violations(eval_program(prog, bad, sha.aux_columns(bad, chal) if col in (C.CA, C.MULT) else aux, chal, pis), t.shape[1]), (col, row)
    wrong = pis.copy()
    wrong[0] ^= 1
    assert violations(eval_program(prog, t, aux, chal, wrong), t.shape[1])


@pytest.mark.parametrize("degree_bits", [7, 8])
def test_oracle_proves_and_the_product_verifier_accepts(oracle, degree_bits):
    stark = sha.make_stark(degree_bits, num_query_rounds=20, pow_bits=4)
    t, pis, digests = sha.generate_trace(degree_bits, MESSAGES)
    assert digests and digests[0] == hashlib.sha256(b"abc").digest()
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
    tb[sha.Cols.CE, 40] = (int(tb[sha.Cols.CE, 40]) + 1) % 8
    try:
        pb = oracle_lib.stark_prove(oracle, stark, tb, pis)
    except Exception:
        pb = None
    if pb is not None:
        with pytest.raises(vx.VxError):
            stark.verify(pis, pb)


def test_two_tables_on_one_bus(oracle):
    """The SHA-256 table SENDS every completed digest, a second table RECEIVES them; the challenges of the bus are drawn over
    BOTH trace caps, each proof verifies with them, and the closing sums cancel.  Proofs by the oracle, verdicts by the product's
    host verifier (vx_stark_verify_shared); a sink that claims a different digest does not balance."""
    from vectorx_amd import stark_bus
    cfg = dict(num_query_rounds=16, pow_bits=4)
    sha_stark = sha.make_stark(8, bus=True, **cfg)
    t, pis, digests = sha.generate_trace(8, MESSAGES)
    assert len(digests) == 2                                   # "abc", "" complete inside 256 rows (3 blocks of 66)
    sink_stark, sink_t, sink_pis = sha.make_sink(4, digests, **cfg)
    proofs, shared = oracle_lib.stark_prove_tables(oracle, [(sha_stark, t, pis), (sink_stark, sink_t, sink_pis)])
    sums = stark_bus.verify_bus([(sha_stark, pis), (sink_stark, sink_pis)], proofs)
    assert int(sums[0][0]) != 0 and (int(sums[0][0]) + int(sums[1][0])) % P == 0
    # each proof is bound to the JOINT challenges: verified on its own (its own challenges) it fails
    with pytest.raises(vx.VxError):
        sha_stark.verify(pis, proofs[0])
    # the product's challenge derivation equals the oracle's
    caps = [sha_stark.proof_trace_cap(proofs[0]), sink_stark.proof_trace_cap(proofs[1])]
    assert sha_stark.desc.num_aux_challenges == 6                     # [gamma_range, beta, gamma_bus] x 2 challenge sets
    assert (vx.stark_joint_challenges(caps, [4, 4], 6) == shared).all()
    assert len(sums[0]) == 2 and (int(sums[0][1]) + int(sums[1][1])) % P == 0 and int(sums[0][1]) != int(sums[0][0])
    # a sink that receives a digest nobody sent: both proofs are valid on their own, the bus does not balance
    wrong = [digests[0], hashlib.sha256(b"not sent").digest()]
    sink2, sink2_t, sink2_pis = sha.make_sink(4, wrong, **cfg)
    proofs2, _ = oracle_lib.stark_prove_tables(oracle, [(sha_stark, t, pis), (sink2, sink2_t, sink2_pis)])
    with pytest.raises(vx.VxError, match="balance"):
        stark_bus.verify_bus([(sha_stark, pis), (sink2, sink2_pis)], proofs2)
    # swapping the two proofs' order changes the joint challenges: rejected
    with pytest.raises(vx.VxError):
        stark_bus.verify_tables([(sink_stark, sink_pis), (sha_stark, pis)], proofs[::-1])
