"""VX_STARK_OPENINGS_DIGEST (include/vxprover.h): the transcript takes a tree hash of the opening set.  On the CPU: the oracle proves
with the option and the product verifier — which restates the tree on the host — accepts; a description without the option refuses the
same bytes and vice versa; an opening changed after the fact is refused.  The GPU twin (tests/test_gpu_stark.py) holds the device
computation of the digest against the oracle's, byte for byte."""
import hashlib

import numpy as np
import pytest

import oracle_lib
import vectorx_amd as vx
from vectorx_amd import sha256_air as sha

P = 0xFFFFFFFF00000001
MESSAGES = [b"abc", b"", b"x" * 100]


@pytest.fixture(scope="module")
def oracle():
    return oracle_lib.load()


def test_oracle_and_product_verifier_agree_on_the_digest_transcript(oracle):
    cfg = dict(num_query_rounds=12, pow_bits=4)
    plain = sha.make_stark(9, **cfg)
    digest = sha.make_stark(9, openings_digest=True, **cfg)
    assert digest.desc.override_flags == vx.VX_STARK_OPENINGS_DIGEST and plain.desc.override_flags == 0
    t, pis, digs = sha.generate_trace(9, MESSAGES)
    assert digs[0] == hashlib.sha256(b"abc").digest()
    p0 = oracle_lib.stark_prove(oracle, plain, t, pis)
    p1 = oracle_lib.stark_prove(oracle, digest, t, pis)
    plain.verify(pis, p0)
    digest.verify(pis, p1)
    assert len(p0) == len(p1) and p0 != p1                      # same shape; the challenges after the openings differ
    n_open = (2 * plain.desc.num_columns + 2 * plain.desc.num_aux_columns + 2) * 16
    caps = 3 * (32 << plain.desc.cap_height)
    assert p0[:caps + n_open] == p1[:caps + n_open]              # caps and openings are the same; FRI is not
    with pytest.raises(vx.VxError):
        plain.verify(pis, p1)
    with pytest.raises(vx.VxError):
        digest.verify(pis, p0)
    for k in (caps + 8, caps + n_open - 8):                     # the first and the last opening: every leaf of the tree is bound
        bad = bytearray(p1)
        bad[k] ^= 1
        with pytest.raises(vx.VxError):
            digest.verify(pis, bytes(bad))


def test_the_fri_arity_override_and_the_digest_combine(oracle):
    st = sha.make_stark(9, openings_digest=True, fri_arities=[2, 2], num_query_rounds=10, pow_bits=3)
    assert st.desc.override_flags == 2 | 8
    t, pis, _ = sha.generate_trace(9, MESSAGES)
    st.verify(pis, oracle_lib.stark_prove(oracle, st, t, pis))
    bad = sha.make_stark(9)
    bad.desc.override_flags = 16
    with pytest.raises(vx.VxError, match="override_flags"):
        bad.verify(pis, b"\x00" * 64)
