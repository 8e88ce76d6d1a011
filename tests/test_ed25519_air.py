"""Ed25519 scalar multiplication as non-native field arithmetic (vectorx_amd/ed25519_air.py — own AIR, not Curta's; the EdDSA chip of
/root/reference/circuits/builder/justification.rs:237): the trace's result is checked against an independent affine implementation
and the RFC 8032 test vector, the constraint program against the trace (every constraint of every row), every column class against a
single-cell corruption, and the STARK pipeline on the CPU through the oracle's prover + the product's host verifier.
The GPU twin is tests/test_gpu_stark.py."""
import hashlib

import numpy as np
import pytest

import oracle_lib
import vectorx_amd as vx
from test_sha256_air import eval_program, violations
from vectorx_amd import ed25519_air as ed

P = ed.P
CHAL = np.array([0x1234567890ABCDEF % P], dtype=np.uint64)


def test_curve_constants_and_reference_implementation():
    q, d = ed.Q25519, ed.D_ED
    on_curve = lambda x, y: (-x * x + y * y - 1 - d * x * x * y * y) % q == 0
    assert on_curve(ed.BX, ed.BY) and ed.BY == 4 * pow(5, q - 2, q) % q and ed.BX % 2 == 0
    L = (1 << 252) + 27742317777372353535851937790883648493
    assert ed.affine_scalar_mult(L) == (0, 1)                                  # the base point has order L
    # RFC 8032 section 7.1, test 1: secret key -> public key
    sk = bytes.fromhex("9d61b19deffd5a60ba844af492ec2cc44449c5697b326919703bac031cae7f60")
    h = hashlib.sha512(sk).digest()
    s = (int.from_bytes(h[:32], "little") & ((1 << 254) - 8)) | (1 << 254)
    assert ed.compress(ed.affine_scalar_mult(s)).hex() == "d75a980182b10ab7d54bfed3c964073a0ee172f3daa62325af021a68f707511a"


@pytest.fixture(scope="module")
def ed10():
    prog, npush = ed.build_program()
    k = 0xC0FFEE11
    t, pis, pt = ed.generate_trace(10, k)
    return prog, npush, t, pis, pt, k


def test_trace_result_equals_the_affine_reference(ed10):
    prog, npush, t, pis, pt, k = ed10
    assert pt == ed.affine_scalar_mult(k) and pt != (0, 1)
    assert int(pis[0]) == k and not pis[1:8].any()
    assert sum(int(pis[8 + j]) << (16 * j) for j in range(16)) == pt[0] and sum(int(pis[24 + j]) << (16 * j) for j in range(16)) == pt[1]
    assert t.shape == (ed.Cols.N, 1024) and ed.Cols.N > 600 and npush > 1000 and len(prog) > 15000
    for k2 in (0, 1, 2, 0xFFFFFFFF):                                           # 0 -> the identity, all-ones: every step adds
        _, _, p2 = ed.generate_trace(10, k2)
        assert p2 == ed.affine_scalar_mult(k2)


def test_every_constraint_vanishes_on_the_trace_and_not_on_a_broken_one(ed10):
    prog, npush, t, pis, _, _ = ed10
    aux = ed.aux_columns(t, CHAL)
    cons = eval_program(prog, t, aux, CHAL, pis)
    n = t.shape[1]
    assert len(cons) == npush and violations(cons, n) == []
    C = ed.Cols
    cells = [(C.REG + 3, 40), (C.REG + 32 * 4 + 7, 41), (C.REG + 32 * 8 + 31, 100), (C.X + 5, 10), (C.Y + 9, 11), (C.Y + 0, 16), (C.Z + 2, 12),
             (C.Z + 31, 28), (C.Z + 0, 29), (C.Q + 4, 13), (C.W + 17, 14), (C.W + 100, 15), (C.BIT, 70), (C.KACC, 95), (C.BND, 31), (C.BND, 1023),
             (C.POS + 3, 97), (C.J + 0, 500), (C.SEL + 5, 200), (C.TBL, 9), (C.MULT, 1)]
    for col, row in cells:
        bad = t.copy()
        bad[col, row] = (int(bad[col, row]) + 1) % P
        a = ed.aux_columns(bad, CHAL) if col in (C.TBL, C.MULT) or C.Z <= col < C.Z + C.NLOOK else aux
        assert violations(eval_program(prog, bad, a, CHAL, pis), n), (col, row)
    for idx in (0, 9, 30):                                                     # the scalar, x, y
        wrong = pis.copy()
        wrong[idx] = (int(wrong[idx]) + 1) % P
        assert violations(eval_program(prog, t, aux, CHAL, wrong), n), idx
    # a carry byte outside the table with the carry itself unchanged (low byte + 256, high byte - 1): every arithmetic relation
    # still holds, the helper columns are recomputed honestly — only the running sum notices
    bad = t.copy()
    row = 32 * 20 + 13
    assert int(bad[C.W + 2 * 7 + 1, row]) >= 1
    bad[C.W + 2 * 7, row] += 256
    bad[C.W + 2 * 7 + 1, row] -= 1
    got = violations(eval_program(prog, bad, ed.aux_columns(bad, CHAL), CHAL, pis), n)
    assert [(i, kd) for i, kd, _ in got] == [(npush - 1, vx.VX_AIR_LAST_ROW)], got


RFC8032_KEYS = [   # RFC 8032 section 7.1, tests 1 - 3: secret key -> public key
    ("9d61b19deffd5a60ba844af492ec2cc44449c5697b326919703bac031cae7f60", "d75a980182b10ab7d54bfed3c964073a0ee172f3daa62325af021a68f707511a"),
    ("4ccd089b28ff96da9db6c346ec114e0f5b8a319f35aba624da8cf6ed4fb8a6fb", "3d4017c3e843895a92b70aa74d1b7ebc9c982ccf2ec4968cc0cd55f12af4660c"),
    ("c5aa8df43f9f837bedb7442f31dcb7b166d38535076f094b85ce3a2e0b4458f7", "fc51cd8e6218a1a38da47ed00230f0580816ed13ba3303ac5deb911548908025"),
]


def test_rfc8032_keys_on_the_reference_implementation():
    for sk_hex, pk_hex in RFC8032_KEYS:
        h = hashlib.sha512(bytes.fromhex(sk_hex)).digest()
        s = (int.from_bytes(h[:32], "little") & ((1 << 254) - 8)) | (1 << 254)
        assert ed.compress(ed.affine_scalar_mult(s)).hex() == pk_hex


@pytest.mark.parametrize("sk_hex,pk_hex", RFC8032_KEYS[:2])
def test_rfc8032_public_key_with_all_256_scalar_bits(sk_hex, pk_hex):
    sk = bytes.fromhex(sk_hex)
    h = hashlib.sha512(sk).digest()
    s = (int.from_bytes(h[:32], "little") & ((1 << 254) - 8)) | (1 << 254)
    t, pis, pt = ed.generate_trace(13, s)
    assert ed.compress(pt).hex() == pk_hex
    assert [int(x) for x in pis[:8]] == [(s >> (32 * (7 - j))) & 0xFFFFFFFF for j in range(8)]       # consumed most significant first
    prog, npush = ed.build_program()
    cons = eval_program(prog, t, ed.aux_columns(t, CHAL), CHAL, pis)
    assert violations(cons, t.shape[1]) == []


def test_oracle_proves_and_the_product_verifier_accepts(oracle):
    stark = ed.make_stark(10, num_query_rounds=16, pow_bits=4)
    t, pis, pt = ed.generate_trace(10, 0x9E3779B9)
    proof = oracle_lib.stark_prove(oracle, stark, t, pis)
    stark.verify(pis, proof)
    for idx in (0, 8, 39):
        wrong = pis.copy()
        wrong[idx] = (int(wrong[idx]) + 1) % P
        with pytest.raises(vx.VxError):
            stark.verify(wrong, proof)
    bad = bytearray(proof)
    bad[len(bad) // 2] ^= 1
    with pytest.raises(vx.VxError):
        stark.verify(pis, bytes(bad))


def test_rfc8032_signature_verifies_through_two_scalar_multiplication_tables():
    """EdDSA verification [S]B = R + [h]A (RFC 8032 section 5.1.7) with BOTH scalar multiplications taken from the table's traces:
    [S]B from the base-point program, [h]A from a program with A's cached form baked in (`build_program(base=A)`); h = SHA-512(R || A || M)
    mod L and the final point addition / comparison are plain host arithmetic (own stand-in: Curta's gadget also proves the hash and
    the group equation).  Test vector 2 of section 7.1 (one-byte message); every constraint of both traces vanishes."""
    sk = bytes.fromhex("4ccd089b28ff96da9db6c346ec114e0f5b8a319f35aba624da8cf6ed4fb8a6fb")
    pk = bytes.fromhex("3d4017c3e843895a92b70aa74d1b7ebc9c982ccf2ec4968cc0cd55f12af4660c")
    msg = bytes.fromhex("72")
    sig = bytes.fromhex("92a009a9f0d4cab8720e820b5f642540a2b27b5416503f8fb3762223ebdb69da085ac1e43e15996e458f3613d0f11d8c387b2eaeb4302aeeb00d291612bb0c00")
    q, d = ed.Q25519, ed.D_ED
    L = (1 << 252) + 27742317777372353535851937790883648493

    def decompress(b):
        y = int.from_bytes(b, "little") & ((1 << 255) - 1)
        sign = b[31] >> 7
        x2 = (y * y - 1) * pow(d * y * y + 1, q - 2, q) % q
        x = pow(x2, (q + 3) // 8, q)
        if (x * x - x2) % q:
            x = x * pow(2, (q - 1) // 4, q) % q
        assert (x * x - x2) % q == 0
        return (q - x if (x & 1) != sign else x, y)

    A, R = decompress(pk), decompress(sig[:32])
    S = int.from_bytes(sig[32:], "little")
    h = int.from_bytes(hashlib.sha512(sig[:32] + pk + msg).digest(), "little") % L
    assert S < L
    t1, pis1, SB = ed.generate_trace(13, S)
    t2, pis2, hA = ed.generate_trace(13, h, base=A)
    assert SB == ed.affine_scalar_mult(S) and hA == ed.affine_scalar_mult(h, A)
    assert ed.affine_add(R, hA) == SB                                          # the signature is valid
    assert ed.affine_add(R, ed.affine_scalar_mult(h + 1, A)) != SB
    prog_b, _ = ed.build_program()
    prog_a, n_a = ed.build_program(base=A)
    assert prog_a != prog_b and len(prog_a) == len(prog_b)                     # same shape, other constants
    assert violations(eval_program(prog_b, t1, ed.aux_columns(t1, CHAL), CHAL, pis1), t1.shape[1]) == []
    assert violations(eval_program(prog_a, t2, ed.aux_columns(t2, CHAL), CHAL, pis2), t2.shape[1]) == []
    assert violations(eval_program(prog_b, t2, ed.aux_columns(t2, CHAL), CHAL, pis2), t2.shape[1])      # [h]A is not a trace of the B program
