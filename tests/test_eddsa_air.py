"""The batched EdDSA table (vectorx_amd/eddsa_air.py — own AIR, not Curta's; the chip behind
/root/reference/circuits/builder/justification.rs:237-243): many signature equations [S]B - [h]A = R per trace, 16-bit limbs, results on a
bus.  Checked here on the CPU: the RFC 8032 section 7.1 signatures verify through the table (trace result == R, against an independent
affine implementation too), every constraint of every row vanishes and does not on a corrupted cell of every column class, the
oracle proves table + sink and the product's host verifier (vx_stark_verify_bus) accepts exactly the honest bus.
The GPU twin is tests/test_gpu_stark.py."""
import random

import numpy as np
import pytest

import oracle_lib
import vectorx_amd as vx
from test_sha256_air import caught_near, eval_program, violations
from vectorx_amd import eddsa_air as ea

P = ea.P
CHAL = np.array([0x1234567890ABCDEF % P, 0x0FEDCBA987654321 % P, 0x1111111122222222], dtype=np.uint64)
RFC8032 = [   # RFC 8032 section 7.1, tests 1 - 3: secret key, public key, message, signature
    ("9d61b19deffd5a60ba844af492ec2cc44449c5697b326919703bac031cae7f60", "d75a980182b10ab7d54bfed3c964073a0ee172f3daa62325af021a68f707511a", "",
     "e5564300c360ac729086e2cc806e828a84877f1eb8e5d974d873e065224901555fb8821590a33bacc61e39701cf9b46bd25bf5f0595bbe24655141438e7a100b"),
    ("4ccd089b28ff96da9db6c346ec114e0f5b8a319f35aba624da8cf6ed4fb8a6fb", "3d4017c3e843895a92b70aa74d1b7ebc9c982ccf2ec4968cc0cd55f12af4660c", "72",
     "92a009a9f0d4cab8720e820b5f642540a2b27b5416503f8fb3762223ebdb69da085ac1e43e15996e458f3613d0f11d8c387b2eaeb4302aeeb00d291612bb0c00"),
    ("c5aa8df43f9f837bedb7442f31dcb7b166d38535076f094b85ce3a2e0b4458f7", "fc51cd8e6218a1a38da47ed00230f0580816ed13ba3303ac5deb911548908025", "af82",
     "6291d657deec24024827e69c3abe01a30ce548a284743a445e3680d7db5ac3ac18ff9b538d16f290ae67f760984dc6594a7c15e9716ed28dc027beceea1ec40a"),
]


def test_host_side_rfc8032():
    """the data generator: signing reproduces the RFC's signatures, decompression its public keys, the equation's inputs verify"""
    for sk, pk, msg, sig in RFC8032:
        p, s = ea.sign(bytes.fromhex(sk), bytes.fromhex(msg))
        assert p.hex() == pk and s.hex() == sig
        a, S, h, r = ea.equation_inputs(bytes.fromhex(pk), bytes.fromhex(msg), bytes.fromhex(sig))
        assert ea.reference_result(a, S, h) == r                       # [S]B - [h]A = R on the independent affine implementation
        assert ea.reference_result(a, S, (h + 1) % ea.ELL) != r
    with pytest.raises(ValueError):
        ea.decompress(bytes([2]) + bytes(31))                          # y = 2 is not on the curve


def _small_case(nsigs=2, seed=5):
    lay = ea.Layout(limb_bits=8, scalar_bits=32)
    rng = random.Random(seed)
    sigs = [(ea.affine_scalar_mult(rng.randrange(1, ea.ELL)), rng.randrange(1 << 32), rng.randrange(1 << 32)) for _ in range(nsigs)]
    return lay, sigs


@pytest.fixture(scope="module")
def small():
    lay, sigs = _small_case()
    prog, npush = ea.build_program(lay)
    t, res = ea.generate_trace(lay, 12, sigs)
    return lay, sigs, prog, npush, t, res


def test_short_scalars_against_the_affine_reference(small):
    lay, sigs, prog, npush, t, res = small
    assert lay.L == 16 + 42 * 32 + 4 and ea.capacity(lay, 12) == 3 and t.shape == (lay.N, 4096)
    assert res == [ea.reference_result(a, s, h) for a, s, h in sigs]
    rows, tuples = ea.send_tuples(lay, t)
    assert rows.tolist() == [lay.L - 1, 2 * lay.L - 1]                   # two active instances send; the filler instance does not
    assert [list(map(int, tp)) for tp in tuples] == [ea.tuple_of(lay, a, s, h, r) for (a, s, h), r in zip(sigs, res)]
    for edge in [((ea.BX, ea.BY), 0, 0), ((ea.BX, ea.BY), 1, 1), ((ea.BX, ea.BY), (1 << 32) - 1, (1 << 32) - 1), ((ea.BX, ea.BY), 5, 0)]:
        _, r = ea.generate_trace(lay, 12, [edge])                       # 0 / 0 -> the identity; S = h with A = B -> the identity again
        assert r == [ea.reference_result(*edge)]
    assert ea.generate_trace(lay, 12, [((ea.BX, ea.BY), 7, 7)])[1] == [(0, 1)]
    with pytest.raises(ValueError, match="curve"):
        ea.generate_trace(lay, 12, [((ea.BX, ea.BY + 1), 1, 1)])         # A off the curve: the on-curve row cannot produce 1


def test_every_constraint_vanishes_and_a_corrupted_cell_of_every_column_class_is_caught(small):
    lay, sigs, prog, npush, t, res = small
    n = t.shape[1]
    aux, closing = ea.aux_columns(lay, t, CHAL)
    cons = eval_program(prog, t, aux, CHAL, closing)
    assert len(cons) == npush and violations(cons, n) == []
    C, NL, L = lay, lay.NL, lay.L
    loop0 = 16 + 42 * 3                                                # a row inside the loop of instance 0
    cells = [(C.RT + 20, loop0 + 4), (C.RT + 0, L), (C.REG + 3, loop0), (C.REG + NL * 9 + 5, loop0 + 30), (C.REG + NL * 12, L + 40), (C.X + 5, loop0 + 1),
             (C.Y + 9, loop0 + 2), (C.Y, loop0 + 16), (C.Y + 1, loop0 + 30), (C.Z + 2, loop0 + 3), (C.Z, 7), (C.Z, L - 3), (C.Q + 4, loop0 + 5),
             (C.W + 17, loop0 + 6), (C.W + 2 * C.NC - 1, loop0 + 7), (C.BIT, loop0 + 8), (C.BIT + 1, loop0 + 9), (C.KACC, loop0 + 10), (C.KACC + 1, L + 100),
             (C.BND, loop0), (C.BND, 16 + 42 * 32 - 1), (C.FIN, 16 + 42 * 32 - 1), (C.FIN, loop0), (C.POS + 3, loop0), (C.J, L + 50), (C.SW, 100),
             (C.SW + 1, L + 200), (C.ACT, 50), (C.ACT, 2 * L + 10), (C.TBL, 9), (C.MULT, 1)]
    for col, row in cells:
        bad = t.copy()
        bad[col, row] = (int(bad[col, row]) + 1) % P
        a, cl = ea.aux_columns(lay, bad, CHAL)                           # honest second-round columns for the corrupted trace
        assert caught_near(prog, bad, a, CHAL, cl, row, n), (col, row)
    # a carry limb outside the table with the carry itself unchanged (low limb + 2^LB, high limb - 1): every arithmetic relation still
    # holds and the helper columns are recomputed honestly — only the closing of the lookup's running sum notices
    bad = t.copy()
    row = loop0 + 11
    assert int(bad[C.W + 2 * 7 + 1, row]) >= 1
    bad[C.W + 2 * 7, row] += 1 << C.LB
    bad[C.W + 2 * 7 + 1, row] -= 1
    a, cl = ea.aux_columns(lay, bad, CHAL)
    got = violations(eval_program(prog, bad, a, CHAL, cl), n)
    assert len(got) == 1 and got[0][1] == vx.VX_AIR_LAST_ROW, got
    # a closing sum that is not what the table sent
    assert violations(eval_program(prog, t, aux, CHAL, [(int(closing[0]) + 1) % P]), n)


def test_full_size_instances_with_16_bit_limbs_verify_the_rfc_signatures():
    """production shape: 256-bit scalars, 16-bit limbs, 2^17 rows = 12 instances; the three RFC signatures + a forged one"""
    lay = ea.Layout()
    assert (lay.L, lay.NL, lay.N, lay.NAUX, lay.NTUPLE) == (10772, 16, 475, 50, 80)
    sigs, rs = [], []
    for sk, pk, msg, sig in RFC8032:
        a, s, h, r = ea.equation_inputs(bytes.fromhex(pk), bytes.fromhex(msg), bytes.fromhex(sig))
        sigs.append((a, s, h))
        rs.append(r)
    a, s, h, r = ea.equation_inputs(bytes.fromhex(RFC8032[0][1]), b"another message", bytes.fromhex(RFC8032[0][3]))     # signature of a different message
    sigs.append((a, s, h))
    t, res = ea.generate_trace(lay, 17, sigs)
    assert res[:3] == rs and res[3] != r                                 # the table computes; whether the result IS R is the bus's verdict
    prog, npush = ea.build_program(lay)
    aux, closing = ea.aux_columns(lay, t, CHAL)
    assert violations(eval_program(prog, t, aux, CHAL, closing), t.shape[1]) == []
    assert int(t[lay.Z:lay.Z + lay.NLOOK].max()) < 1 << 16
    with pytest.raises(AssertionError):
        ea.generate_trace(lay, 16, sigs[:1])                             # 2^16 rows cannot host the 65 536-entry table before the last row


def test_oracle_proves_table_and_sink_and_the_c_verifier_judges_the_bus(oracle):
    lay, sigs = _small_case(nsigs=3, seed=9)
    cfg = dict(num_query_rounds=12, pow_bits=4)
    stark = ea.make_stark(lay, 12, **cfg)
    assert stark.desc.num_aux_columns == 2 * lay.NAUX and stark.desc.num_aux_challenges == 6 and stark.desc.num_aux_public_inputs == 2
    t, res = ea.generate_trace(lay, 12, sigs)
    nopi = np.zeros(0, dtype=np.uint64)
    honest = [ea.tuple_of(lay, a, s, h, r) for (a, s, h), r in zip(sigs, res)]
    sink, sink_t, _ = ea.make_sink(lay, honest, **cfg)
    proofs, _ = oracle_lib.stark_prove_tables(oracle, [(stark, t, nopi), (sink, sink_t, nopi)])
    sums = vx.stark_verify_bus([(stark, nopi), (sink, nopi)], proofs)
    assert sums.shape == (2, 2) and int(sums[0][0]) != 0
    # a sink that holds a different R for the second signature: both proofs valid, the bus does not balance
    forged = [list(tp) for tp in honest]
    forged[1][-1] ^= 1
    sink2, sink2_t, _ = ea.make_sink(lay, forged, **cfg)
    proofs2, _ = oracle_lib.stark_prove_tables(oracle, [(stark, t, nopi), (sink2, sink2_t, nopi)])
    with pytest.raises(vx.VxError, match="cancel"):
        vx.stark_verify_bus([(stark, nopi), (sink2, nopi)], proofs2)
    # a sink that expects a signature the table never checked
    sink3, sink3_t, _ = ea.make_sink(lay, honest + [ea.tuple_of(lay, (ea.BX, ea.BY), 1, 2, (0, 1))], **cfg)
    proofs3, _ = oracle_lib.stark_prove_tables(oracle, [(stark, t, nopi), (sink3, sink3_t, nopi)])
    with pytest.raises(vx.VxError, match="cancel"):
        vx.stark_verify_bus([(stark, nopi), (sink3, nopi)], proofs3)
    bad = bytearray(proofs[0])
    bad[len(bad) // 2] ^= 1
    with pytest.raises(vx.VxError):
        vx.stark_verify_bus([(stark, nopi), (sink, nopi)], [bytes(bad), proofs[1]])
