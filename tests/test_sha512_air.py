"""SHA-512 at chip density (vectorx_amd/sha512_air.py — own AIR, not Curta's; the hash inside EdDSA, RFC 8032 section 5.1.7,
/root/reference/circuits/builder/justification.rs:237): the trace is checked against hashlib, the constraint program against the
trace (every constraint of every row), column classes against single-cell corruptions, and the STARK pipeline on the CPU through
the oracle's prover + the product's independent host verifier.  CPU only: 1995 + 5 columns, 4037 constraints, a 31.7 k-word program."""
import hashlib

import numpy as np
import pytest

import oracle_lib
import vectorx_amd as vx
from test_sha256_air import eval_program, violations
from vectorx_amd import sha512_air as s5

P = s5.P
MESSAGES = [b"abc", b"", bytes(range(200)), b"x" * 111, b"y" * 112]      # 111 / 112 bytes: the padding's one-block / two-block edge
CHAL = np.array([0x1234567890ABCDEF % P], dtype=np.uint64)


@pytest.fixture(scope="module")
def sha9():
    prog, npush = s5.build_program()
    t, pis, digests = s5.generate_trace(9, MESSAGES)
    return prog, npush, t, pis, digests


def test_constants_are_derived_and_the_trace_equals_hashlib(sha9):
    prog, npush, t, pis, digests = sha9
    assert s5.K512[1] == 0x7137449123ef65cd and s5.IV[7] == 0x5be0cd19137e2179
    # 512 rows = 6 whole blocks of 82 rows: "abc" 1, "" 1, bytes(range(200)) 2, 111 x's 1, 112 y's 2 (only the first fits)
    assert digests == [hashlib.sha512(m).digest() for m in MESSAGES[:4]]
    got = b"".join(((int(pis[2 * k + 1]) << 32) | int(pis[2 * k])).to_bytes(8, "big") for k in range(8))
    assert got == hashlib.sha512(MESSAGES[3]).digest()
    assert t.shape == (s5.Cols.N, 512) and s5.Cols.N == 1995 and npush > 4000 and len(prog) > 30000


def test_every_constraint_vanishes_on_the_trace_and_not_on_a_broken_one(sha9):
    prog, npush, t, pis, _ = sha9
    aux = s5.aux_columns(t, CHAL)
    n = t.shape[1]
    cons = eval_program(prog, t, aux, CHAL, pis)
    assert len(cons) == npush and violations(cons, n) == []
    C = s5.Cols
    for col, row in [(C.S + 5, 90), (C.S + 64 * 4 + 63, 10), (C.WB + 64 * 3 + 40, 20), (C.WB + 64 * 15 + 1, 60), (C.CA, 33), (C.CA + 1, 34), (C.CE + 1, 35),
                     (C.CW, 36), (C.H + 3, 100), (C.NF, 81), (C.X0 + 50, 12), (C.X1 + 7, 12), (C.M + 9, 13), (C.Y0 + 62, 30), (C.Y1 + 60, 31), (C.MULT, 3),
                     (C.SEL + 10, 200), (C.D + 1, 300), (C.FFC + 3, 80), (C.FFC + 4, 80), (C.TBL, 5)]:
        bad = t.copy()
        bad[col, row] = (int(bad[col, row]) + 1) % P
        a = s5.aux_columns(bad, CHAL) if col in (C.MULT,) or C.CA <= col < C.TBL else aux
        if col == C.TBL:
            a = aux
        assert violations(eval_program(prog, bad, a, CHAL, pis), n), (col, row)
    wrong = pis.copy()
    wrong[5] ^= 1
    assert violations(eval_program(prog, t, aux, CHAL, wrong), n)


def test_oracle_proves_and_the_product_verifier_accepts(oracle):
    stark = s5.make_stark(8, num_query_rounds=16, pow_bits=4)
    t, pis, digests = s5.generate_trace(8, MESSAGES)
    assert digests[0] == hashlib.sha512(b"abc").digest()
    proof = oracle_lib.stark_prove(oracle, stark, t, pis)
    stark.verify(pis, proof)
    wrong = pis.copy()
    wrong[15] = (int(wrong[15]) + 1) % P
    with pytest.raises(vx.VxError):
        stark.verify(wrong, proof)
    bad = bytearray(proof)
    bad[len(bad) // 3] ^= 1
    with pytest.raises(vx.VxError):
        stark.verify(pis, bytes(bad))


def test_eddsa_hash_of_the_rfc8032_vector_through_the_table():
    """h = SHA-512(R || A || M) of RFC 8032 test 2 comes out of the table's trace (one block), and reduced mod L it is the scalar the
    Ed25519 table multiplies A by (tests/test_ed25519_air.py) — the three pieces of an EdDSA verification, each from a table."""
    pk = bytes.fromhex("3d4017c3e843895a92b70aa74d1b7ebc9c982ccf2ec4968cc0cd55f12af4660c")
    sig = bytes.fromhex("92a009a9f0d4cab8720e820b5f642540a2b27b5416503f8fb3762223ebdb69da085ac1e43e15996e458f3613d0f11d8c387b2eaeb4302aeeb00d291612bb0c00")
    msg = sig[:32] + pk + bytes.fromhex("72")
    t, pis, digests = s5.generate_trace(7, [msg])
    assert digests == [hashlib.sha512(msg).digest()]
    L = (1 << 252) + 27742317777372353535851937790883648493
    assert int.from_bytes(digests[0], "little") % L == int.from_bytes(hashlib.sha512(msg).digest(), "little") % L
