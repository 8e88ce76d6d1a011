"""Ed25519 signatures verified THROUGH TABLES ONLY (round 5): the SHA-512 table hashes R || A || M and sends (R, A, digest),
the EdDSA table (full program) decompresses, reduces the digest mod L, checks S < L and the group equation and sends (A, S, digest, R),
the link table joins the two on the digest and sends what a verifier holds — (A, S, R): the bytes of the public key and of the signature.
RFC 8032 section 7.1 vectors; the bus balances for them and for nothing else.  CPU: constraints in Python, the four tables proven by the
oracle, the product's host verifier (vx_stark_verify_bus) judging the bus.  GPU twin: tests/test_gpu_stark.py."""
import hashlib

import numpy as np
import pytest

import oracle_lib
import vectorx_amd as vx
from test_eddsa_air import CHAL, RFC8032
from test_sha256_air import eval_program, violations
from vectorx_amd import eddsa_air as ea
from vectorx_amd import sha512_air as s5
from vectorx_amd import sig_link_air as link

CFG = dict(num_query_rounds=12, pow_bits=4)


def _raw(k=3):
    out = []
    for _, pk, msg, sig in RFC8032[:k]:
        pk, msg, sig = bytes.fromhex(pk), bytes.fromhex(msg), bytes.fromhex(sig)
        out.append((pk, msg, sig, hashlib.sha512(sig[:32] + pk + msg).digest()))
    return out


def test_sha512_bus_variant_sends_the_first_64_bytes_and_the_digest():
    msgs = [sig[:32] + pk + msg for pk, msg, sig, _ in _raw()] + [b"short", bytes(range(200))]
    t, pis, digs = s5.generate_trace(10, msgs, bus=True)
    assert digs == [hashlib.sha512(m).digest() for m in msgs]
    base, _, _ = s5.generate_trace(10, msgs)
    assert (t[:s5.Cols.N] == base).all()                                   # the 1995 columns of the plain table are untouched
    prog, npush = s5.build_program(True)
    aux, closing = s5.aux_columns_bus(t, CHAL)
    pub = list(pis) + [int(closing[0])]
    assert violations(eval_program(prog, t, aux, CHAL, pub), t.shape[1]) == []
    rows, tuples = s5.bus_tuples(t)
    assert len(rows) == len(msgs)
    for m, tp in zip(msgs, tuples):
        padded = (m + b"\x80" + bytes(64))[:64]
        assert [int(x) for x in tp] == s5.le_words(padded) + s5.le_words(hashlib.sha512(m).digest()) + [s5.TAG_SHA512]
    C = s5.Cols
    for col, row in [(C.LW + 3, 100), (C.LW + 15, 7), (C.FIRST, 82), (C.FIRST, 0), (C.LW, 81)]:
        bad = t.copy()
        bad[col, row] = (int(bad[col, row]) + 1) % s5.P
        a, cl = s5.aux_columns_bus(bad, CHAL)
        assert violations(eval_program(prog, bad, a, CHAL, list(pis) + [int(cl[0])]), t.shape[1]), (col, row)


def test_link_table_constraints():
    raw = _raw()
    stark, t, _ = link.make_link([link.row_of(pk, sig, dig) for pk, _, sig, dig in raw], **CFG)
    aux, closing = link.aux_columns(t, CHAL)
    prog = link.build_program()
    assert violations(eval_program(prog, t, aux, CHAL, [int(closing[0])]), t.shape[1]) == []
    bad = t.copy()
    bad[link.DW + 5, 1] ^= 1
    assert violations(eval_program(prog, bad, aux, CHAL, [int(closing[0])]), t.shape[1])


def test_signatures_verify_through_tables_only_and_forgeries_unbalance_the_bus(oracle):
    raw = _raw(1)
    lay = ea.Layout(8, 256, full=True)
    nopi = np.zeros(0, dtype=np.uint64)
    sha = s5.make_stark(9, bus=True, **CFG)
    sha_t, sha_pis, digs = s5.generate_trace(9, [sig[:32] + pk + msg for pk, msg, sig, _ in raw], bus=True)
    assert digs == [d for _, _, _, d in raw]
    ed = ea.make_stark(lay, 14, **CFG)
    ed_t, res = ea.generate_trace(lay, 14, [ea.equation_inputs_full(pk, msg, sig) for pk, msg, sig, _ in raw])
    lk, lk_t, _ = link.make_link([link.row_of(pk, sig, dig) for pk, _, sig, dig in raw], **CFG)
    honest = [link.verifier_tuple(pk, sig) for pk, _, sig, _ in raw]
    sink, sink_t, _ = ea.make_sink(lay, honest, ntuple=25, **CFG)
    tables = [(sha, sha_t, sha_pis), (ed, ed_t, nopi), (lk, lk_t, nopi), (sink, sink_t, nopi)]
    proofs, _ = oracle_lib.stark_prove_tables(oracle, tables)
    descs = [(sha, sha_pis), (ed, nopi), (lk, nopi), (sink, nopi)]
    sums = vx.stark_verify_bus(descs, proofs)
    assert all(int(x) != 0 for x in sums[:, 0])                           # every table sent or received something; the four sums cancel
    # Forgeries, judged where the bus is judged — the closing sums of the four tables under shared challenges must cancel (computed here
    # with the tables' own second-round functions; the GPU twin proves a forged bus outright).  The honest four cancel:
    P = ea.P
    closing = lambda fn, tr: int(fn(tr, CHAL)[1][0])      # noqa: E731
    ed_cl = int(ea.aux_columns(lay, ed_t, CHAL)[1][0])
    sha_cl = closing(s5.aux_columns_bus, sha_t)
    assert (sha_cl + ed_cl + closing(link.aux_columns, lk_t) + closing(sink.aux_fn, sink_t)) % P == 0
    # a verifier holding S + L: the link row that would produce its tuple does not match what the EdDSA table sent; the honest link row
    # leaves the verifier's tuple unmatched — either way the sums do not cancel
    pk, _, sig, dig = raw[0]
    s = int.from_bytes(sig[32:], "little")
    forged_sig = sig[:32] + (s + ea.ELL).to_bytes(32, "little")
    sink2, sink2_t, _ = ea.make_sink(lay, [link.verifier_tuple(pk, forged_sig)], ntuple=25, **CFG)
    for link_rows in ([link.row_of(pk, forged_sig, dig)], [link.row_of(pk, sig, dig)]):
        _, lk2_t, _ = link.make_link(link_rows, **CFG)
        assert (sha_cl + ed_cl + closing(link.aux_columns, lk2_t) + closing(sink2.aux_fn, sink2_t)) % P != 0
    # a link row claiming a digest the SHA-512 table never produced for R || A
    _, lk3_t, _ = link.make_link([link.row_of(pk, sig, hashlib.sha512(b"other").digest())], **CFG)
    assert (sha_cl + ed_cl + closing(link.aux_columns, lk3_t) + closing(sink.aux_fn, sink_t)) % P != 0
    # R with its sign bit flipped in the verifier's bytes
    sig_flip = bytes(sig[:31]) + bytes([sig[31] ^ 0x80]) + sig[32:]
    sink4, sink4_t, _ = ea.make_sink(lay, [link.verifier_tuple(pk, sig_flip)], ntuple=25, **CFG)
    _, lk4_t, _ = link.make_link([link.row_of(pk, sig_flip, dig)], **CFG)
    assert (sha_cl + ed_cl + closing(link.aux_columns, lk4_t) + closing(sink4.aux_fn, sink4_t)) % P != 0


def test_a_bus_closes_only_when_every_challenge_set_closes_on_its_own():
    """`GeneratedSignatureBus.closed()` added up the closing sums of ALL challenge sets; X + Y = 0 with X != 0 passed although
    `vx_stark_verify_bus` checks every set by itself (single-set soundness).  Now index by index."""
    from vectorx_amd import stark_chips
    P = 0xFFFFFFFF00000001
    bus = object.__new__(stark_chips.GeneratedSignatureBus)
    bus.ctx = object()
    key = id(bus.ctx)
    bus.last = {key: (None, None, [[5, 9], [P - 5, P - 9]])}               # both sets close
    assert bus.closed()
    bus.last = {key: (None, None, [[5, 9], [P - 4, P - 10]])}              # set 0 is off by one, set 1 by minus one: the grand total is 0
    assert (5 + 9 + P - 4 + P - 10) % P == 0 and not bus.closed()
    bus.last = {key: (None, None, [[5, 9], [P - 5]])}                       # tables disagree on the number of sets
    assert not bus.closed()
