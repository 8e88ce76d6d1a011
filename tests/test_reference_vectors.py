"""The known-answer vectors the reference's own tests hold for the byte formats around the proving path (SURVEY.md §8c /
Appendix D), against the host-side restatement in vectorx_amd/avail_codec.py.  These are circuit OUTPUTS in the reference
(its gadgets are checked by proving a tiny circuit); here they pin the plain functions a caller needs to build a request
and to interpret an output."""
import hashlib
import json
import struct
from pathlib import Path

import pytest

from vectorx_amd import avail_codec as ac

V = json.loads((Path(__file__).parent / "golden" / "reference_vectors.json").read_text())


def test_compact_u32_table():
    for value, mode in V["compact_u32"]["cases"]:
        enc = ac.encode_compact_u32(value)
        padded = enc + bytes(ac.MAX_COMPACT_UINT_BYTES - len(enc))        # the reference pads to MAX_COMPACT_UINT_BYTES
        got, got_mode, used = ac.decode_compact_u32(padded)
        assert (got, got_mode) == (value, mode)
        assert used == len(enc) == {0: 1, 1: 2, 2: 4, 3: 5}[mode]
    # boundaries of each mode encode to the shortest form
    assert ac.encode_compact_u32(63) == bytes([0xFC]) and ac.encode_compact_u32(64) == bytes([0x01, 0x01])
    assert ac.encode_compact_u32(1 << 30) == bytes([3, 0, 0, 0, 0x40])
    for bad in (b"", bytes([1]), bytes([2, 0]), bytes([7, 0, 0, 0, 0, 0]), bytes([3, 1, 2])):
        with pytest.raises(ValueError):
            ac.decode_compact_u32(bad)


def test_precommit_vector():
    raw = bytes(V["precommit"]["encoded"])
    assert len(raw) == ac.ENCODED_PRECOMMIT_LENGTH
    d = ac.decode_precommit(raw)
    assert d["block_number"] == V["precommit"]["block_number"] and d["authority_set_id"] == V["precommit"]["authority_set_id"]
    assert ac.encode_precommit(d["block_hash"], d["block_number"], d["round"], d["authority_set_id"]) == raw
    for bad in (raw[:-1], bytes([0]) + raw[1:]):
        with pytest.raises(ValueError):
            ac.decode_precommit(bad)


def test_io_packing_lengths_and_roundtrip():
    io = V["io_packing"]
    raw = ac.pack_header_range_input(317857, bytes(range(32)), 298, bytes(range(32, 64)), 318113)
    assert len(raw) == io["header_range_input_len"]
    d = ac.unpack_header_range_input(raw)
    assert (d["trusted_block"], d["authority_set_id"], d["target_block"]) == (317857, 298, 318113)
    assert raw[:4] == struct.pack(">I", 317857)                               # abi.encodePacked integers are big-endian
    assert len(ac.pack_rotate_input(298, bytes(32))) == io["rotate_input_len"]
    assert len(ac.pack_header_range_output(bytes(32), bytes(32), bytes(32))) == io["header_range_output_len"]


def test_commitments_follow_their_definitions():
    pks = [bytes([i]) * 32 for i in range(1, 6)]
    h = b""
    for pk in pks:
        h = hashlib.sha256(h + pk).digest()
    assert ac.authority_set_commitment(pks) == h and ac.authority_set_commitment([]) == b""
    # unhashed leaves, zero padding to a power of two
    a, b, c = (bytes([x]) * 32 for x in (7, 8, 9))
    z = bytes(32)
    H = lambda x, y: hashlib.sha256(x + y).digest()
    assert ac.simple_merkle_root([a]) == a
    assert ac.simple_merkle_root([a, b, c]) == H(H(a, b), H(c, z))
    s, d = ac.header_range_commitments([a, b, c], [c, b, a], 8)
    assert s == H(H(H(a, b), H(c, z)), H(H(z, z), H(z, z)))
    assert d == H(H(H(c, b), H(a, z)), H(H(z, z), H(z, z)))
    with pytest.raises(ValueError):
        ac.header_range_commitments([a] * 9, [a] * 9, 8)


def test_synthetic_epoch_end_header_passes_the_circuits_checks_and_tampering_does_not():
    """BASELINE.json configs[0] plumbing: a synthetic epoch-end header with an FRNK scheduled-change log and 300
    dummy keys, checked the way /root/reference/circuits/builder/rotate.rs:81-93, 95-166, 223-283 checks it."""
    import hashlib
    from vectorx_amd import avail_codec as ac
    header, start, pubkeys = ac.synthetic_epoch_end_header(b"seed")
    assert len(pubkeys) == 300 and len(header) <= ac.MAX_HEADER_SIZE == 35840
    assert header[start + 1] == 4 and header[start + 2:start + 6] == b"FRNK"
    out = ac.rotate_output(header, start, pubkeys)
    h = b""
    for pk in pubkeys:
        h = hashlib.sha256(h + pk).digest()
    assert out == h
    body = start + 6
    _, mode, ln = ac.decode_compact_u32(header[body:body + 5])          # message length: 1 + 2 + 300 * 40 + 4 = 12007 < 2^14 -> two-byte mode
    assert mode == 1 and ln == 2 and header[body + ln] == 1
    count, _, cl = ac.decode_compact_u32(header[body + ln + 1:body + ln + 6])
    assert count == 300 and cl == 2
    first = body + ln + 1 + cl
    for off, what in [(start + 1, "consensus"), (start + 3, "consensus"), (body + ln, "scheduled"), (first + 5, "public key"),
                      (first + 32, "weight"), (first + 40 * 299 + 39, "weight"), (first + 40 * 300 + 2, "delay")]:
        bad = bytearray(header)
        bad[off] ^= 1
        with pytest.raises(ValueError, match=what):
            ac.rotate_output(bytes(bad), start, pubkeys)
    with pytest.raises(ValueError):
        ac.verify_epoch_end_header(header, start, 299, pubkeys[:299])      # the witness' count must equal the encoded one
    small, s2, pk2 = ac.synthetic_epoch_end_header(b"x", num_authorities=5)
    assert ac.rotate_output(small, s2, pk2) == ac.authority_set_commitment(pk2)
