"""The FULL program of the batched EdDSA table (vectorx_amd/eddsa_air.py, Layout(full=True)): the instance takes the
BYTES a verifier holds — the public key's and R's encodings, S, the SHA-512 digest of R || A || M — and checks everything RFC 8032
5.1.7 asks between them and the group equation: decompression of A and R (canonical coordinates, sign bit = parity of x, the curve
equation), h = digest mod L (canonical), S < L.  CPU: the RFC 8032 section 7.1 signatures go through with the bus tuple equal to their
raw bytes; a flipped sign bit, S + L, a wrong h, a non-canonical coordinate each fail — the generator refuses the witness, the constraints
reject a forced one, the bus does not balance.  8-bit limbs here (2^14 rows hold the 256-entry table and one instance); the production
layout (16-bit limbs, 2^17 rows) is proven on the GPU in tests/test_gpu_stark.py."""
import hashlib

import numpy as np
import pytest

import oracle_lib
import vectorx_amd as vx
from test_eddsa_air import CHAL, RFC8032
from test_sha256_air import caught_near, eval_program, violations
from vectorx_amd import eddsa_air as ea

P = ea.P
LOG_N = 14


def _rfc(i):
    sk, pk, msg, sig = RFC8032[i]
    pk, msg, sig = bytes.fromhex(pk), bytes.fromhex(msg), bytes.fromhex(sig)
    return pk, msg, sig, hashlib.sha512(sig[:32] + pk + msg).digest()


@pytest.fixture(scope="module")
def full():
    lay = ea.Layout(8, 256, full=True)
    prog, npush = ea.build_program(lay)
    return lay, prog, npush


def _violations(lay, prog, t):
    aux, closing = ea.aux_columns(lay, t, CHAL)
    return violations(eval_program(prog, t, aux, CHAL, closing), t.shape[1])


def test_layout_and_program(full):
    lay, prog, npush = full
    base = ea.Layout(8, 256)
    assert (lay.NP, lay.NE, lay.NT) == (32, 11, 85) and lay.L == base.L + 23 and lay.N == base.N + 23 + 32 and lay.NTUPLE == 41 and lay.TAG == ea.TAG_EDDSA
    assert ea.capacity(ea.Layout(16, 256, full=True), 20) == 97           # the production table still holds 97 signatures per 2^20 rows
    # the base program is untouched by the extension (its goldens and native generator stay valid)
    assert len(ea.build_program(base)[0]) < len(prog)
    with pytest.raises(AssertionError):
        ea.Layout(8, 32, full=True)                                       # a 253-bit h needs the 256-step ladder


@pytest.mark.parametrize("which", [0, 1, 2])
def test_rfc8032_signatures_verify_from_their_bytes(full, which):
    lay, prog, npush = full
    pk, msg, sig, dig = _rfc(which)
    t, res = ea.generate_trace(lay, LOG_N, [ea.equation_inputs_full(pk, msg, sig)])
    assert res == [ea.decompress(sig[:32])]                               # the ladder arrives at R
    assert _violations(lay, prog, t) == []
    rows, tuples = ea.send_tuples(lay, t)
    assert rows.tolist() == [lay.L - 1]
    assert [int(x) for x in tuples[0]] == ea.tuple_of_full(lay, pk, sig, dig)     # the tuple IS the verifier's bytes
    assert int(t[lay.Z:lay.Z + lay.NLOOK].max()) < 1 << lay.LB


def test_a_wrong_h_a_wrong_digest_and_s_plus_l_are_refused(full):
    lay, prog, npush = full
    pk, msg, sig, dig = _rfc(1)
    a, s, h, d = ea.equation_inputs_full(pk, msg, sig)
    # h that is not digest mod L: the ladder would use it, the reduction rows produce the real one — the words cannot be both
    t, _ = ea.generate_trace(lay, LOG_N, [(a, s, (h + 1) % ea.ELL, d)], strict=False)
    assert _violations(lay, prog, t)
    # h + L: congruent, not canonical — the same binding catches it (the ladder's 256-bit scalar is not the canonical v)
    assert h + ea.ELL < 1 << 256
    t, _ = ea.generate_trace(lay, LOG_N, [(a, s, h + ea.ELL, d)], strict=False)
    assert _violations(lay, prog, t)
    # a digest that is not the one the scalar was reduced from
    t, _ = ea.generate_trace(lay, LOG_N, [(a, s, h, d ^ 1)], strict=False)
    assert _violations(lay, prog, t)
    # S + L: the same group element, not a canonical scalar (RFC 8032 5.1.7 step 1): no witness exists ...
    forged = sig[:32] + (s + ea.ELL).to_bytes(32, "little")
    with pytest.raises(ValueError, match="S >= L"):
        ea.equation_inputs_full(pk, msg, forged)
    a2, s2, h2, d2 = ea.equation_inputs_full(pk, msg, forged, check=False)
    assert s2 == s + ea.ELL and (h2, d2) == (h, d)                        # R and A are the same bytes: the same digest
    with pytest.raises(ValueError, match="below L"):
        ea.generate_trace(lay, LOG_N, [(a2, s2, h2, d2)])
    # ... and a forced one (the comparison witness wrapped mod 2^256) breaks the integer identity S + c = L - 1
    t, _ = ea.generate_trace(lay, LOG_N, [(a2, s2, h2, d2)], strict=False)
    assert _violations(lay, prog, t)


def test_sign_bits_and_canonical_coordinates(full):
    lay, prog, npush = full
    pk, msg, sig, dig = _rfc(0)
    a, s, h, d = ea.equation_inputs_full(pk, msg, sig)
    honest = ea.tuple_of_full(lay, pk, sig, dig)
    # the public key with its sign bit flipped decodes to (-x, y): on the curve, a different point — the table can prove THAT statement,
    # it arrives at another R, and the tuple it sends is not the verifier's
    flipped = pk[:31] + bytes([pk[31] ^ 0x80])
    a_neg = ea.decompress(flipped)
    assert a_neg == ((-a[0]) % ea.Q25519, a[1])
    t, res = ea.generate_trace(lay, LOG_N, [(a_neg, s, h, d)])
    assert _violations(lay, prog, t) == [] and res != [ea.decompress(sig[:32])]
    sent = [int(x) for x in ea.send_tuples(lay, t)[1][0]]
    assert sent[:8] == ea.tuple_of_full(lay, flipped, sig, dig)[:8] and sent != honest and sent[-8:] != honest[-8:]
    # R with its sign bit flipped: what the verifier holds is not what the table sends
    sig_flipped = bytes(sig[:31]) + bytes([sig[31] ^ 0x80]) + sig[32:]
    t, _ = ea.generate_trace(lay, LOG_N, [(a, s, h, d)])
    assert [int(x) for x in ea.send_tuples(lay, t)[1][0]] != ea.tuple_of_full(lay, pk, sig_flipped, dig)
    # a non-canonical x for A (x + p < 2^256 has the other parity): the comparison row has no witness
    with pytest.raises(ValueError, match="below p"):
        ea.generate_trace(lay, LOG_N, [((a[0] + ea.Q25519, a[1]), s, h, d)])
    # a corrupted cell of every new column class is caught
    t, _ = ea.generate_trace(lay, LOG_N, [(a, s, h, d)])
    C = lay
    for col, row in [(C.AENC + 7, 100), (C.AENC, C.L - 1), (C.DW + 3, 5000), (C.DW + 12, 20), (C.RENC + 7, C.L - 1), (C.RENC + 2, 9),
                     (C.Z, 21), (C.Z + 1, 21), (C.Q, 17), (C.Z + 5, 25), (C.REG + C.NL * 6 + 1, 28), (C.Z, C.L - 2), (C.SW + 8, 29), (C.SW, 30)]:
        bad = t.copy()
        bad[col, row] = (int(bad[col, row]) + 1) % P
        a_, cl_ = ea.aux_columns(lay, bad, CHAL)
        assert caught_near(prog, bad, a_, CHAL, cl_, row, bad.shape[1]), (col, row)


def test_oracle_proves_the_full_table_and_the_bus_judges_the_bytes(oracle):
    lay = ea.Layout(8, 256, full=True)
    cfg = dict(num_query_rounds=12, pow_bits=4)
    stark = ea.make_stark(lay, LOG_N, **cfg)
    pk, msg, sig, dig = _rfc(2)
    t, _ = ea.generate_trace(lay, LOG_N, [ea.equation_inputs_full(pk, msg, sig)])
    nopi = np.zeros(0, dtype=np.uint64)
    honest = ea.tuple_of_full(lay, pk, sig, dig)
    sink, sink_t, _ = ea.make_sink(lay, [honest], **cfg)
    proofs, _ = oracle_lib.stark_prove_tables(oracle, [(stark, t, nopi), (sink, sink_t, nopi)])
    sums = vx.stark_verify_bus([(stark, nopi), (sink, nopi)], proofs)
    assert int(sums[0][0]) != 0
    # (bus-level forgeries — S + L, another digest — are proven and judged in tests/test_sig_link.py, where the whole bus is present)
