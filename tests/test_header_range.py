"""vectorx_amd/header_range.py on the CPU: the host mirror of the reference's circuit logic around the hashes
(/root/reference/circuits/builder/subchain_verification.rs:84-296, justification.rs:127-257, header_range.rs:31-58), with hashlib standing
where the DAG's GPU tables stand (tests/test_gpu_dag_pool.py runs the same statements over the tables' digests)."""
import dataclasses
import hashlib

import pytest

from vectorx_amd import avail_codec as ac
from vectorx_amd import header_range as hr
from vectorx_amd import mapreduce as mr

SHA = lambda m: hashlib.sha256(m).digest()                  # noqa: E731
B2B = lambda m: hashlib.blake2b(m, digest_size=32).digest()     # noqa: E731
SHAPE = dict(capacity=32, header_bytes=512, num_authorities=8, distinct_keys=2)


def run_statements(req, mutate=None):
    """the whole DAG's statements with hashlib for every hash -> (96 output bytes, root subchain)"""
    subs = []
    for j in range(req.capacity // hr.HEADERS_PER_MAP):
        hdrs = req.batch(j)
        state, data = hr.map_leaves(req.target_block, hdrs)
        msgs = hr.tree_messages(state) + hr.tree_messages(data)
        subs.append(hr.map_statement(req.trusted_block, req.target_block, j, hdrs, [B2B(h) for h in hdrs], msgs, [SHA(m) for m in msgs]))
    while len(subs) > 1:
        nxt = []
        for i in range(0, len(subs), 2):
            msgs = hr.reduce_messages(subs[i], subs[i + 1])
            nxt.append(hr.reduce_statement(subs[i], subs[i + 1], msgs, [SHA(m) for m in msgs]))
        subs = nxt
    just = req.justification()
    if mutate:
        just = mutate(just)
    chain = hr.authority_chain_messages(just.pubkeys)
    verified = [(pk, just.encoded_precommit, sg) for pk, sg, s in zip(just.pubkeys, just.signatures, just.validator_signed) if s]
    return hr.outer_statement(req.input_bytes, subs[0], just, chain, [SHA(m) for m in chain], verified), subs[0]


@pytest.mark.parametrize("num_headers", [32, 1, 8, 9, 19, 31])
def test_the_statements_arrive_at_the_host_computation(num_headers):
    req = hr.make_request(b"seed %d" % num_headers, num_headers=num_headers, **SHAPE)
    out, top = run_statements(req)
    assert out == hr.expected_output(req) and len(out) == 96
    inp = ac.unpack_header_range_input(req.input_bytes)
    assert (top.num_blocks, top.start_block, top.end_block) == (num_headers, inp["trusted_block"] + 1, inp["target_block"])
    assert top.start_parent == inp["trusted_header"] and top.end_header_hash == B2B(req.headers[num_headers - 1]) == out[:32]
    # the commitments are the contract's: a SHA-256 tree over the unhashed roots, zero leaves past the target block
    dec = [ac.decode_header(h) for h in req.headers[:num_headers]]
    assert out[32:64] == ac.simple_merkle_root([d["state_root"] for d in dec] + [bytes(32)] * (32 - num_headers))
    assert inp["authority_set_hash"] == ac.authority_set_commitment(req.justification().pubkeys)


def test_requests_are_self_consistent_and_differ():
    a, b = hr.make_request(b"a", **SHAPE), hr.make_request(b"b", **SHAPE)
    assert a.input_bytes != b.input_bytes and hr.expected_output(a) != hr.expected_output(b)
    assert hr.make_request(b"a", **SHAPE).headers == a.headers and hr.cached_request(b"a", **SHAPE) is hr.cached_request(b"a", **SHAPE)
    for k, h in enumerate(a.headers):
        d = ac.decode_header(h)
        assert len(h) == 512 and d["block_number"] == a.trusted_block + 1 + k
        assert d["parent_hash"] == (B2B(a.headers[k - 1]) if k else ac.unpack_header_range_input(a.input_bytes)["trusted_header"])
    just = a.justification()
    pc = ac.decode_precommit(just.encoded_precommit)
    assert pc["block_hash"] == B2B(a.headers[-1]) and pc["block_number"] == a.target_block and len(just.encoded_precommit) == 53
    from vectorx_amd import eddsa_air as ea
    for pk, sg in zip(just.pubkeys[:2], just.signatures[:2]):
        assert ea.verify(pk, just.encoded_precommit, sg)                 # RFC 8032 verification on the host
    assert not ea.verify(just.pubkeys[0], just.encoded_precommit, just.signatures[1])


def test_header_codec():
    h = ac.encode_header(b"\x01" * 32, 317857, b"\x02" * 32, b"middle bytes", b"\x03" * 32)
    assert ac.decode_header(h) == {"parent_hash": b"\x01" * 32, "block_number": 317857, "state_root": b"\x02" * 32, "data_root": b"\x03" * 32}
    for number, width in ((5, 1), (300, 2), (70000, 4), (1 << 30, 5)):          # the four compact modes move the state root (decoder.rs:121-128)
        h = ac.encode_header(bytes(32), number, b"\x07" * 32, b"", b"\x09" * 32)
        assert len(h) == 32 + width + 64 and ac.decode_header(h)["state_root"] == b"\x07" * 32 and ac.decode_header(h)["block_number"] == number
    assert ac.decode_header(b"") == {"parent_hash": bytes(32), "block_number": 0, "state_root": bytes(32), "data_root": bytes(32)}
    with pytest.raises(ValueError):
        ac.decode_header(b"\x00" * 40)
    s = hr.Subchain(3, 10, b"a" * 32, b"b" * 32, 12, b"c" * 32, b"d" * 32, b"e" * 32)
    assert len(s.pack()) == hr.Subchain.SIZE and hr.Subchain.unpack(s.pack()) == s
    rec = mr.record_of(b"proof bytes" + mr.with_statement(s.pack()), type("P", (), {"emits_statement": True}))
    assert rec[:32] == SHA(b"proof bytes" + mr.with_statement(s.pack())) and rec[32:] == s.pack()
    assert mr.record_of(b"proof bytes" + mr.with_statement(s.pack())) == rec[:32]                  # a plain prover: the digest alone


def test_what_the_circuit_asserts_is_refused_here():
    req = hr.make_request(b"refusals", **SHAPE)
    hdrs = req.batch(1)
    hashes = [B2B(h) for h in hdrs]
    state, data = hr.map_leaves(req.target_block, hdrs)
    msgs = hr.tree_messages(state) + hr.tree_messages(data)
    digs = [SHA(m) for m in msgs]
    args = (req.trusted_block, req.target_block, 1, hdrs, hashes, msgs, digs)
    good = hr.map_statement(*args)
    # a header that does not continue its predecessor (parent hash; block number)
    bad = list(hdrs)
    bad[3] = bytes([bad[3][0] ^ 1]) + bad[3][1:]
    with pytest.raises(hr.StatementError, match="header 3 of batch 1 is not linked"):
        hr.map_statement(req.trusted_block, req.target_block, 1, bad, hashes, msgs, digs)
    skip = list(hdrs)
    d = ac.decode_header(skip[5])
    skip[5] = ac.encode_header(d["parent_hash"], d["block_number"] + 1, d["state_root"], skip[5][68:-32], d["data_root"])
    with pytest.raises(hr.StatementError, match="header 5"):
        hr.map_statement(req.trusted_block, req.target_block, 1, skip, [B2B(h) for h in skip], msgs, digs)
    # a hash that is not the header's
    with pytest.raises(hr.StatementError, match="not linked"):
        hr.map_statement(req.trusted_block, req.target_block, 1, hdrs, [hashes[0], b"\x00" * 32] + hashes[2:], msgs, digs)
    # the batch of another position
    with pytest.raises(hr.StatementError, match="does not start at block"):
        hr.map_statement(req.trusted_block, req.target_block, 2, hdrs, hashes, msgs, digs)
    # a tree over other leaves, a node hashed from something else than its children's digests
    with pytest.raises(hr.StatementError, match="tree node"):
        hr.map_statement(req.trusted_block, req.target_block, 1, hdrs, hashes, [msgs[1]] + msgs[1:], digs)
    with pytest.raises(hr.StatementError, match="tree node"):
        hr.map_statement(req.trusted_block, req.target_block, 1, hdrs, hashes, msgs, [SHA(b"other")] + digs[1:])
    # reduce: the right subchain must continue the left one, and the table must have hashed the children's roots
    left = hr.map_statement(req.trusted_block, req.target_block, 0, req.batch(0), [B2B(h) for h in req.batch(0)],
                            *(lambda m: (m, [SHA(x) for x in m]))(sum((hr.tree_messages(l) for l in hr.map_leaves(req.target_block, req.batch(0))), [])))
    rm = hr.reduce_messages(left, good)
    merged = hr.reduce_statement(left, good, rm, [SHA(m) for m in rm])
    assert merged.num_blocks == 16 and merged.state_merkle_root == SHA(left.state_merkle_root + good.state_merkle_root)
    with pytest.raises(hr.StatementError, match="does not continue"):
        hr.reduce_statement(good, left, hr.reduce_messages(good, left), [SHA(m) for m in rm])
    with pytest.raises(hr.StatementError, match="children's roots"):
        hr.reduce_statement(left, good, [rm[1], rm[0]], [SHA(m) for m in rm])
    # outer: every assertion of header_range.rs / verify_simple_justification
    out, top = run_statements(req)
    just = req.justification()
    chain = hr.authority_chain_messages(just.pubkeys)
    cd = [SHA(m) for m in chain]
    ok = [(pk, just.encoded_precommit, sg) for pk, sg in zip(just.pubkeys, just.signatures)]
    assert hr.outer_statement(req.input_bytes, top, just, chain, cd, ok) == out
    inp = ac.unpack_header_range_input(req.input_bytes)

    def repack(**kw):
        v = dict(inp, **kw)
        return ac.pack_header_range_input(v["trusted_block"], v["trusted_header"], v["authority_set_id"], v["authority_set_hash"], v["target_block"])

    for raw, what in ((repack(trusted_header=b"\x05" * 32), "trusted header"), (repack(target_block=inp["target_block"] - 1), "target block"),
                      (repack(authority_set_hash=b"\x06" * 32), "not the committed one"), (repack(authority_set_id=8), "precommit is not for")):
        with pytest.raises(hr.StatementError, match=what):
            hr.outer_statement(raw, top, just, chain, cd, ok)
    with pytest.raises(hr.StatementError, match="commitment \\|\\| key"):
        hr.outer_statement(req.input_bytes, top, just, [chain[0], chain[2], chain[1]] + chain[3:], cd, ok)
    with pytest.raises(hr.StatementError, match="precommit is not for"):
        hr.outer_statement(req.input_bytes, dataclasses.replace(top, end_header_hash=b"\x01" * 32), just, chain, cd, ok)
    with pytest.raises(hr.StatementError, match="was not verified"):
        hr.outer_statement(req.input_bytes, top, just, chain, cd, [t for t in ok if t[0] != ok[-1][0]])          # the set repeats 2 keys
    few = dataclasses.replace(just, validator_signed=[True] * 5 + [False] * 3)            # 5 of 8 is not more than 2/3
    with pytest.raises(hr.StatementError, match="2/3"):
        hr.outer_statement(req.input_bytes, top, few, chain, cd, ok)
    enough = dataclasses.replace(just, validator_signed=[True] * 6 + [False] * 2)
    assert hr.outer_statement(req.input_bytes, top, enough, chain, cd, ok[:2]) == out


def test_rotate_statement_arrives_at_the_new_authority_set_hash_and_refuses_what_the_circuit_refuses():
    req = hr.make_rotate_request(b"rotate", 8, 2, 8)
    just = req.justification()
    cm, nm = hr.authority_chain_messages(just.pubkeys), hr.authority_chain_messages(req.new_pubkeys)
    cd, nd = [SHA(m) for m in cm], [SHA(m) for m in nm]
    ok = [(pk, just.encoded_precommit, sg) for pk, sg in zip(just.pubkeys, just.signatures)]
    hh = B2B(req.header)
    args = dict(input_bytes=req.input_bytes, header=req.header, header_hash=hh, just=just, chain_msgs=cm, chain_digests=cd, verified=ok,
                start_position=req.start_position, new_pubkeys=req.new_pubkeys, new_chain_msgs=nm, new_chain_digests=nd)
    out = hr.rotate_statement(**args)
    assert out == hr.expected_rotate_output(req) == ac.authority_set_commitment(req.new_pubkeys) and len(out) == 32
    inp = ac.unpack_rotate_input(req.input_bytes)
    assert inp["authority_set_hash"] == ac.authority_set_commitment(just.pubkeys)
    pc = ac.decode_precommit(just.encoded_precommit)
    assert pc["block_hash"] == hh and pc["block_number"] == ac.decode_header(req.header)["block_number"]

    def refused(what, **kw):
        with pytest.raises(hr.StatementError, match=what):
            hr.rotate_statement(**dict(args, **kw))

    refused("not the committed one", input_bytes=ac.pack_rotate_input(inp["authority_set_id"], b"\x01" * 32))
    refused("precommit is not for", input_bytes=ac.pack_rotate_input(inp["authority_set_id"] + 1, inp["authority_set_hash"]))
    refused("precommit is not for", header_hash=b"\x02" * 32)                       # the justification is about another header
    refused("was not verified", verified=[])
    other = list(req.new_pubkeys)
    other[5] = b"\x09" * 32
    refused("does not announce", new_pubkeys=other)                                # the header's log holds another key
    refused("does not announce", start_position=req.start_position + 1)
    refused("does not announce", new_pubkeys=req.new_pubkeys[:-1], new_chain_msgs=nm[:-1], new_chain_digests=nd[:-1])   # the encoded count differs
    refused("new authority chain hashed something else", new_chain_msgs=[nm[0], nm[2], nm[1]] + nm[3:])
    assert hr.rotate_statement(**dict(args, new_chain_digests=nd[:-1] + [b"\x07" * 32])) == b"\x07" * 32      # the output IS the table's last digest


def test_random_ranges_and_mutations_property():
    """hypothesis: for random capacities, range lengths and header sizes the statements arrive at the host computation; a flipped byte in
    any enabled header is either refused (a link / position assertion) or changes the output (a root or the target hash moved) — never
    silently accepted with the same output."""
    from hypothesis import given, settings
    from hypothesis import strategies as st

    @settings(max_examples=15, deadline=None, derandomize=True)
    @given(st.sampled_from([16, 32, 64]), st.data())
    def check(capacity, data):
        n = data.draw(st.integers(1, capacity))
        size = data.draw(st.sampled_from([101, 128, 333, 512]))
        shape = dict(capacity=capacity, header_bytes=size, num_authorities=4, distinct_keys=2, num_headers=n)
        req = hr.make_request(b"prop %d %d %d" % (capacity, n, size), **shape)
        out, top = run_statements(req)
        assert out == hr.expected_output(req) and top.num_blocks == n
        k = data.draw(st.integers(0, n - 1))
        pos = data.draw(st.integers(0, size - 1))
        h = bytearray(req.headers[k])
        h[pos] ^= 1 << data.draw(st.integers(0, 7))
        req.headers[k] = bytes(h)
        try:
            out2, _ = run_statements(req)
        except (hr.StatementError, ValueError):
            return
        assert out2 != out        # (a changed hash breaks the next link, or the precommit no longer names the target header)

    check()
