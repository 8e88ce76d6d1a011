"""Byte formats on either side of the proving path (host logic, no GPU): what the reference packs into a proof request and
what the circuits commit to in their public outputs.  Each function restates one helper of the reference and is pinned to
the vectors the reference's own tests hold (tests/golden/reference_vectors.json, tests/test_reference_vectors.py).

* SCALE compact integers (the block number inside an encoded Avail header): `decode_compact_int` gadget,
  /root/reference/circuits/builder/decoder.rs:33-103, test table :238-249.
* GRANDPA precommit message `1 | block_hash[32] | block_number u32 LE | round u64 LE | authority_set_id u64 LE` (53 bytes):
  /root/reference/circuits/input/mod.rs:262-289, in-circuit decoder /root/reference/circuits/builder/decoder.rs:106-140,
  test vector :388-395.
* authority set commitment = SHA-256 chained over the compressed Ed25519 keys: /root/reference/circuits/input/mod.rs:250-260
  (in-circuit: circuits/builder/justification.rs:140-161).
* header-range commitments = SHA-256 binary tree over UNHASHED 32-byte leaves (state roots / data roots), zero-padded to the
  tree size: /root/reference/circuits/input/mod.rs:464-489, 493-528.
* function I/O packing: header_range input abi.encodePacked(uint32, bytes32, uint64, bytes32, uint32) = 80 bytes
  (/root/reference/bin/vectorx.rs:106-112), output abi.encode(bytes32, bytes32, bytes32) = 96 bytes
  (/root/reference/circuits/header_range.rs:56-58); rotate input abi.encodePacked(uint64, bytes32) = 40 bytes, output
  bytes32 (/root/reference/circuits/rotate.rs:87-88, 108).
"""
from __future__ import annotations

import hashlib
import struct

MAX_COMPACT_UINT_BYTES = 5
ENCODED_PRECOMMIT_LENGTH = 53


def encode_compact_u32(value: int) -> bytes:
    """parity-scale-codec `Compact<u32>::encode`: the two low bits of the first byte select the width."""
    if not 0 <= value < 1 << 32:
        raise ValueError("compact u32 out of range")
    if value < 1 << 6:
        return bytes([value << 2])
    if value < 1 << 14:
        return struct.pack("<H", (value << 2) | 1)
    if value < 1 << 30:
        return struct.pack("<I", (value << 2) | 2)
    return bytes([3]) + struct.pack("<I", value)     # big-integer mode: (4 - 4) << 2 | 3, then 4 bytes LE


def decode_compact_u32(data: bytes) -> tuple[int, int, int]:
    """(value, compress_mode, encoded length) of a compact integer at the start of `data` (<= 5 bytes are looked at); modes
    0 / 1 / 2 = 1 / 2 / 4 bytes with the value shifted left by two, mode 3 = one length byte followed by 4 value bytes —
    the only big-integer length a u32 block number can have, which is what the gadget supports."""
    if not data:
        raise ValueError("empty compact integer")
    mode = data[0] & 3
    if mode == 0:
        return data[0] >> 2, 0, 1
    if mode == 1:
        if len(data) < 2:
            raise ValueError("truncated compact integer")
        return struct.unpack("<H", data[:2])[0] >> 2, 1, 2
    if mode == 2:
        if len(data) < 4:
            raise ValueError("truncated compact integer")
        return struct.unpack("<I", data[:4])[0] >> 2, 2, 4
    if data[0] >> 2 != 0 or len(data) < 5:
        raise ValueError("compact integer wider than a u32")
    return struct.unpack("<I", data[1:5])[0], 3, 5


def decode_precommit(precommit: bytes) -> dict:
    if len(precommit) != ENCODED_PRECOMMIT_LENGTH or precommit[0] != 1:
        raise ValueError("not an encoded precommit message")
    block_number, = struct.unpack("<I", precommit[33:37])
    rnd, set_id = struct.unpack("<QQ", precommit[37:53])
    return {"block_hash": precommit[1:33], "block_number": block_number, "round": rnd, "authority_set_id": set_id}


def encode_precommit(block_hash: bytes, block_number: int, rnd: int, authority_set_id: int) -> bytes:
    if len(block_hash) != 32:
        raise ValueError("block hash must be 32 bytes")
    return bytes([1]) + block_hash + struct.pack("<IQQ", block_number, rnd, authority_set_id)


HASH_SIZE = 32
DATA_ROOT_OFFSET_FROM_END = 32        # /root/reference/circuits/consts.rs:3


def encode_header(parent_hash: bytes, block_number: int, state_root: bytes, middle: bytes, data_root: bytes) -> bytes:
    """An Avail header as far as the circuits look at it (decode_header below): parent hash | compact block number | state root |
    ... | data root in the last 32 bytes.  `middle` stands for everything in between (extrinsics root, digest logs, the rest of the
    extension)."""
    if len(parent_hash) != 32 or len(state_root) != 32 or len(data_root) != 32:
        raise ValueError("hashes are 32 bytes")
    return parent_hash + encode_compact_u32(block_number) + state_root + bytes(middle) + data_root


def decode_header(header: bytes) -> dict:
    """`decode_header` (/root/reference/circuits/builder/decoder.rs:104-158): {parent_hash, block_number, state_root, data_root}.  The
    circuit reads a zero-padded array of MAX_HEADER_SIZE bytes and a size; an EMPTY header (size 0: the padding of a range shorter than
    the circuit's capacity) decodes to zeros, as the all-zero array does there."""
    if len(header) == 0:
        return {"parent_hash": bytes(32), "block_number": 0, "state_root": bytes(32), "data_root": bytes(32)}
    if len(header) < HASH_SIZE + 1 + HASH_SIZE + DATA_ROOT_OFFSET_FROM_END:
        raise ValueError("header too short")
    number, _, width = decode_compact_u32(header[HASH_SIZE:HASH_SIZE + MAX_COMPACT_UINT_BYTES])
    start = HASH_SIZE + width
    return {"parent_hash": header[:HASH_SIZE], "block_number": number, "state_root": header[start:start + HASH_SIZE],
            "data_root": header[len(header) - DATA_ROOT_OFFSET_FROM_END:]}


def authority_set_commitment(pubkeys) -> bytes:
    """H(H(...H(pk_0) || pk_1 ...) || pk_{k-1}) with H = SHA-256; the empty set hashes to the empty string."""
    h = b""
    for pk in pubkeys:
        if len(pk) != 32:
            raise ValueError("compressed Ed25519 keys are 32 bytes")
        h = hashlib.sha256(h + pk).digest()
    return h


def simple_merkle_root(leaves) -> bytes:
    """Binary SHA-256 tree whose leaves are NOT hashed; the leaf list is zero-padded to a power of two."""
    nodes = [bytes(l) for l in leaves]
    if not nodes:
        return b""
    if any(len(l) != 32 for l in nodes):
        raise ValueError("leaves are 32-byte roots")
    while len(nodes) & (len(nodes) - 1):
        nodes.append(bytes(32))
    while len(nodes) > 1:
        nodes = [hashlib.sha256(nodes[2 * i] + nodes[2 * i + 1]).digest() for i in range(len(nodes) // 2)]
    return nodes[0]


def header_range_commitments(state_roots, data_roots, tree_size: int) -> tuple[bytes, bytes]:
    """(state_root_commitment, data_root_commitment) of the headers (trusted, target]: both lists padded with zero leaves to
    `tree_size` (256 or 512: the N of header_range_N)."""
    if tree_size & (tree_size - 1) or not 0 < len(state_roots) <= tree_size or len(state_roots) != len(data_roots):
        raise ValueError("range does not fit the commitment tree")
    pad = [bytes(32)] * (tree_size - len(state_roots))
    return simple_merkle_root(list(state_roots) + pad), simple_merkle_root(list(data_roots) + pad)


def pack_header_range_input(trusted_block: int, trusted_header: bytes, authority_set_id: int, authority_set_hash: bytes,
                            target_block: int) -> bytes:
    if len(trusted_header) != 32 or len(authority_set_hash) != 32:
        raise ValueError("hashes are 32 bytes")
    return struct.pack(">I", trusted_block) + trusted_header + struct.pack(">Q", authority_set_id) + authority_set_hash + \
        struct.pack(">I", target_block)


def unpack_header_range_input(raw: bytes) -> dict:
    if len(raw) != 80:
        raise ValueError("header_range input is 80 bytes")
    return {"trusted_block": struct.unpack(">I", raw[0:4])[0], "trusted_header": raw[4:36],
            "authority_set_id": struct.unpack(">Q", raw[36:44])[0], "authority_set_hash": raw[44:76],
            "target_block": struct.unpack(">I", raw[76:80])[0]}


def pack_header_range_output(target_header: bytes, state_root_commitment: bytes, data_root_commitment: bytes) -> bytes:
    out = target_header + state_root_commitment + data_root_commitment
    if len(out) != 96:
        raise ValueError("header_range output is three bytes32 values")
    return out


def pack_rotate_input(authority_set_id: int, authority_set_hash: bytes) -> bytes:
    if len(authority_set_hash) != 32:
        raise ValueError("hashes are 32 bytes")
    return struct.pack(">Q", authority_set_id) + authority_set_hash


def unpack_rotate_input(raw: bytes) -> dict:
    if len(raw) != 40:
        raise ValueError("rotate input is 40 bytes")
    return {"authority_set_id": struct.unpack(">Q", raw[0:8])[0], "authority_set_hash": raw[8:40]}


# ---- the epoch-end header a `rotate` proof is about (BASELINE.json configs[0]: plumbing, host side) -----------------------------
# /root/reference/circuits/builder/rotate.rs:81-93 (consensus log: flag 0x04 + engine id "FRNK"), :95-135 (compact message length,
# scheduled-change flag 0x01), :137-166 (compact authority count), :223-283 (pubkey[32] | weight u64 LE = 1 per validator, then a
# u32 delay = 0); sizes /root/reference/circuits/consts.rs:9-52.
MAX_HEADER_SIZE = 280 * 128           # consts.rs:9-16: 280 BLAKE2b chunks
MAX_AUTHORITY_SET_SIZE = 300          # consts.rs:52
CONSENSUS_ENGINE_ID = bytes([70, 82, 78, 75])     # "FRNK"
PUBKEY_LENGTH, WEIGHT_LENGTH, DELAY_LENGTH = 32, 8, 4


def encode_scheduled_change_log(pubkeys, delay: int = 0) -> bytes:
    """DigestItem::Consensus(FRNK, ConsensusLog::ScheduledChange{next_authorities, delay}) as the circuit reads it, starting
    at the byte BEFORE the consensus flag (the circuit skips it: `start_position` points there)."""
    body = bytes([1]) + encode_compact_u32(len(pubkeys))          # scheduled-change flag, authority count
    for pk in pubkeys:
        if len(pk) != PUBKEY_LENGTH:
            raise ValueError("compressed Ed25519 keys are 32 bytes")
        body += pk + struct.pack("<Q", 1)                         # every Avail validator has weight 1
    body += struct.pack("<I", delay)
    return bytes([0]) + bytes([4]) + CONSENSUS_ENGINE_ID + encode_compact_u32(len(body)) + body


def synthetic_epoch_end_header(seed: bytes, num_authorities: int = MAX_AUTHORITY_SET_SIZE, block_number: int = 4321) -> tuple:
    """-> (header bytes, start_position, pubkeys): a SCALE-shaped header (parent hash | compact number | state root |
    extrinsics root | digest) whose digest holds one authority-set change log with `num_authorities` dummy keys derived from
    `seed` — the input a `rotate` witness generator fetches from the chain (circuits/rotate.rs:95-99), synthesised."""
    if not 0 < num_authorities <= MAX_AUTHORITY_SET_SIZE:
        raise ValueError("authority set size outside (0, 300]")
    h = lambda tag: hashlib.sha256(seed + tag).digest()           # noqa: E731
    pubkeys = [h(b"pk" + struct.pack("<I", i)) for i in range(num_authorities)]
    head = h(b"parent") + encode_compact_u32(block_number) + h(b"state") + h(b"extrinsics") + encode_compact_u32(2)
    pre_runtime = bytes([6]) + b"BABE" + encode_compact_u32(8) + h(b"slot")[:8]      # an unrelated log in front
    log = encode_scheduled_change_log(pubkeys)
    header = head + pre_runtime + log[1:]                          # log[0] is the skipped byte = the last byte in front of the flag
    start = len(head) + len(pre_runtime) - 1
    if len(header) > MAX_HEADER_SIZE:
        raise ValueError("header longer than MAX_HEADER_SIZE")
    return header, start, pubkeys


def verify_epoch_end_header(header: bytes, start_position: int, num_authorities: int, pubkeys) -> None:
    """The checks `verify_epoch_end_header` makes in-circuit (builder/rotate.rs:168-283), on the host; raises ValueError."""
    if num_authorities == 0 or num_authorities > MAX_AUTHORITY_SET_SIZE or len(header) > MAX_HEADER_SIZE:
        raise ValueError("bad authority count / header size")
    p = header[start_position:]
    if p[1] != 4 or p[2:6] != CONSENSUS_ENGINE_ID:
        raise ValueError("start_position is not a GRANDPA consensus log")
    _, _, ln = decode_compact_u32(p[6:11])                          # message length: skipped, not checked (rotate.rs:114)
    cur = 6 + ln
    if p[cur] != 1:
        raise ValueError("not a scheduled-change message")
    cur += 1
    count, _, ln = decode_compact_u32(p[cur:cur + 5])
    if count != num_authorities:
        raise ValueError("encoded authority count differs from the witness")
    cur += ln
    for i in range(num_authorities):
        v = p[cur + i * 40:cur + (i + 1) * 40]
        if v[:32] != pubkeys[i]:
            raise ValueError(f"validator {i}: public key differs")
        if v[32:40] != struct.pack("<Q", 1):
            raise ValueError(f"validator {i}: weight is not 1")
    end = cur + num_authorities * 40
    if p[end:end + 4] != bytes(4):
        raise ValueError("delay is not 0")


def rotate_output(header: bytes, start_position: int, pubkeys) -> bytes:
    """new_authority_set_hash: what the rotate circuit writes (circuits/rotate.rs:101-108) once the header checks pass"""
    verify_epoch_end_header(header, start_position, len(pubkeys), pubkeys)
    return authority_set_commitment(pubkeys)
