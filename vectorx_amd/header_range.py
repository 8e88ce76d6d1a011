"""What a header_range request IS and what its proof STATES — the host-side mirror of the circuit logic around the hashes and signatures:

  * `verify_subchain`'s map and reduce bodies (/root/reference/circuits/builder/subchain_verification.rs:84-233, 234-289) and the two
    assertions after them (:292-296): `map_statement`, `reduce_statement`, `Subchain` (= `MapReduceSubchainVariable`, :27-45);
  * `verify_simple_justification` (/root/reference/circuits/builder/justification.rs:195-257) with the chained authority set
    commitment (:127-162) and the voting threshold (:164-186): `outer_statement`;
  * `HeaderRangeCircuit::define` (/root/reference/circuits/header_range.rs:31-58): 80 input bytes -> 96 output bytes;
  * `RotateCircuit::define` / `rotate` (/root/reference/circuits/rotate.rs:80-109, builder/rotate.rs:278-323): 40 input bytes -> the new
    authority set's hash: `rotate_statement`.

Every HASH comes from outside: the DAG hands in the digests its GPU tables computed (BLAKE2b over the headers, SHA-256 over the tree
nodes and the authority chain; the signatures through the signature bus) together with the messages those tables hashed, and the
functions here check that the messages are the ones the statement needs — what the circuit's wiring does in the reference.  A
violated assertion raises StatementError, where the reference's witness generation would fail.

`make_request` derives a self-consistent synthetic request from a seed: a chain of headers linked by their BLAKE2b-256 hashes, an
authority set of real Ed25519 keys, a justification (the precommit over the target header, signed).  `expected_output` computes the
96 output bytes with hashlib and vectorx_amd/avail_codec.py alone — the check the DAG's output is held against.
"""
from __future__ import annotations

import hashlib
import struct
import threading
from dataclasses import dataclass

from . import avail_codec as ac

HEADERS_PER_MAP = 8           # /root/reference/circuits/consts.rs:6
MAX_HEADER_SIZE = 35840       # /root/reference/circuits/consts.rs:16
ZERO32 = bytes(32)


class StatementError(ValueError):
    """an assertion of the circuit does not hold for this request"""


@dataclass(frozen=True)
class Subchain:
    """`MapReduceSubchainVariable` (/root/reference/circuits/builder/subchain_verification.rs:27-45)"""
    num_blocks: int
    start_block: int
    start_header_hash: bytes
    start_parent: bytes
    end_block: int
    end_header_hash: bytes
    state_merkle_root: bytes
    data_merkle_root: bytes

    SIZE = 172

    def pack(self) -> bytes:
        return struct.pack("<II", self.num_blocks, self.start_block) + self.start_header_hash + self.start_parent + \
            struct.pack("<I", self.end_block) + self.end_header_hash + self.state_merkle_root + self.data_merkle_root

    @staticmethod
    def unpack(raw: bytes) -> "Subchain":
        if len(raw) != Subchain.SIZE:
            raise ValueError("a packed subchain statement is 172 bytes")
        nb, sb = struct.unpack("<II", raw[:8])
        eb, = struct.unpack("<I", raw[72:76])
        return Subchain(nb, sb, raw[8:40], raw[40:72], eb, raw[76:108], raw[108:140], raw[140:172])


def _check(ok: bool, what: str):
    if not ok:
        raise StatementError(what)


# ---- the SHA-256 trees --------------------------------------------------------------------------------------------------------------
def tree_messages(leaves, sha256=lambda m: hashlib.sha256(m).digest()) -> list:
    """the 64-byte messages of the binary tree over 2^k UNHASHED 32-byte leaves, level by level (k = 3: 4 + 2 + 1 = 7 messages)"""
    level, out = [bytes(l) for l in leaves], []
    assert len(level) >= 2 and len(level) & (len(level) - 1) == 0 and all(len(l) == 32 for l in level)
    while len(level) > 1:
        msgs = [level[2 * i] + level[2 * i + 1] for i in range(len(level) // 2)]
        out += msgs
        level = [sha256(m) for m in msgs]
    return out


def tree_root_checked(leaves, messages, digests) -> bytes:
    """The root of the tree over `leaves`, every node's hash taken from `digests` (one per message, in tree_messages' order) — after
    checking that `messages` are the leaves' pairs and then the DIGESTS' pairs, i.e. that the table hashed this tree and no other."""
    level, k = [bytes(l) for l in leaves], 0
    while len(level) > 1:
        half = len(level) // 2
        for i in range(half):
            _check(bytes(messages[k + i]) == level[2 * i] + level[2 * i + 1], "a tree node was hashed from something else than its two children")
        level = [bytes(d) for d in digests[k:k + half]]
        k += half
    _check(k == len(messages) == len(digests), "the tree has another number of nodes")
    return level[0]


# ---- map / reduce / outer -----------------------------------------------------------------------------------------------------------
def map_leaves(global_end_block: int, headers) -> tuple:
    """(state root leaves, data root leaves) of a batch: the decoded roots of the headers up to the target block, zero leaves after it
    (`get_root_from_hashed_leaves(.., nb_enabled_leaves)`, subchain_verification.rs:213-220)"""
    state, data, noop = [], [], False
    for h in headers:
        d = ac.decode_header(h)
        enabled = not noop and len(h) > 0
        state.append(d["state_root"] if enabled else ZERO32)
        data.append(d["data_root"] if enabled else ZERO32)
        noop = noop or d["block_number"] == global_end_block or len(h) == 0
    return state, data


def map_statement(global_start_block: int, global_end_block: int, job_index: int, headers, header_hashes, tree_msgs, tree_digests) -> Subchain:
    """The map body (subchain_verification.rs:84-233) for batch `job_index`: `headers` = its HEADERS_PER_MAP encoded headers (b"" past
    the target block), `header_hashes` = their BLAKE2b-256 digests, `tree_msgs` / `tree_digests` = what the SHA-256 table hashed for
    the two trees (state first) and its digests."""
    n = HEADERS_PER_MAP
    _check(len(headers) == n and len(header_hashes) == n, "a map job hashes HEADERS_PER_MAP headers")
    batch_start = global_start_block + job_index * n + 1
    batch_end = global_start_block + job_index * n + n
    disabled = global_end_block < batch_start
    noop = disabled
    nums, parents = [], []
    end_block, end_hash, num_headers = 0, ZERO32, 0
    for i in range(n):
        d = ac.decode_header(headers[i])
        nums.append(d["block_number"])
        parents.append(d["parent_hash"])
        if i > 0:
            linked = parents[i] == bytes(header_hashes[i - 1]) and nums[i] == nums[i - 1] + 1
            _check(noop or linked, f"header {i} of batch {job_index} is not linked to its predecessor")
        if not noop:
            end_block, end_hash = nums[i], bytes(header_hashes[i])
            num_headers += 1
        noop = noop or nums[i] == global_end_block
    _check(disabled or nums[0] == batch_start, f"batch {job_index} does not start at block {batch_start}")
    _check(noop or end_block == batch_end, f"batch {job_index} does not end at block {batch_end}")
    state, data = [], []
    for i in range(n):
        d = ac.decode_header(headers[i])
        state.append(d["state_root"] if i < num_headers else ZERO32)
        data.append(d["data_root"] if i < num_headers else ZERO32)
    half = n - 1
    _check(len(tree_msgs) == 2 * half and len(tree_digests) == 2 * half, "a map job hashes two 8-leaf trees")
    state_root = tree_root_checked(state, tree_msgs[:half], tree_digests[:half])
    data_root = tree_root_checked(data, tree_msgs[half:], tree_digests[half:])
    return Subchain(num_headers, nums[0], bytes(header_hashes[0]), parents[0], end_block, end_hash, state_root, data_root)


def reduce_messages(left: Subchain, right: Subchain) -> list:
    return [left.state_merkle_root + right.state_merkle_root, left.data_merkle_root + right.data_merkle_root]


def reduce_statement(left: Subchain, right: Subchain, msgs, digests) -> Subchain:
    """The reduce body (subchain_verification.rs:234-289); `msgs` / `digests` = the two messages the SHA-256 table hashed and its digests"""
    inactive = right.num_blocks == 0
    linked = left.end_header_hash == right.start_parent and left.end_block == right.start_block - 1
    _check(inactive or linked, "the right subchain does not continue the left one")
    _check([bytes(m) for m in msgs] == reduce_messages(left, right), "a reduce job hashed something else than its children's roots")
    _check(len(digests) == 2, "a reduce job hashes two nodes")
    return Subchain(left.num_blocks + right.num_blocks, left.start_block, left.start_header_hash, left.start_parent,
                    left.end_block if inactive else right.end_block, left.end_header_hash if inactive else right.end_header_hash,
                    bytes(digests[0]), bytes(digests[1]))


def authority_chain_messages(pubkeys, sha256=lambda m: hashlib.sha256(m).digest()) -> list:
    """what `compute_authority_set_commitment` hashes (justification.rs:140-161): pk_0, then commitment_so_far || pk_i"""
    msgs, h = [], b""
    for pk in pubkeys:
        msgs.append(h + bytes(pk))
        h = sha256(msgs[-1])
    return msgs


@dataclass
class Justification:
    """`JustificationVariable`: the precommit message, the authority set, who signed and their signatures"""
    encoded_precommit: bytes
    pubkeys: list
    signatures: list
    validator_signed: list

    @property
    def num_authorities(self) -> int:
        return len(self.pubkeys)


def justification_checks(block_number: int, block_hash: bytes, authority_set_id: int, authority_set_hash: bytes, just: Justification,
                         chain_msgs, chain_digests, verified):
    """`verify_simple_justification` (/root/reference/circuits/builder/justification.rs:195-257): the authority set is the committed one
    (chain_msgs / chain_digests: what the SHA-256 table hashed for the commitment chain, and its digests), the precommit is for this
    block under this set, everyone marked as signed has a signature among `verified` (the (public key, message, signature) triples a
    balanced signature bus has verified), and more than 2/3 signed."""
    # 1) the authority set commitment
    _check(just.num_authorities >= 1, "an authority set has at least one member")
    _check(len(chain_msgs) == len(chain_digests) == just.num_authorities, "the authority chain has one hash per authority")
    prev = b""
    for pk, m, d in zip(just.pubkeys, chain_msgs, chain_digests):
        _check(bytes(m) == prev + bytes(pk), "the authority chain hashed something else than commitment || key")
        prev = bytes(d)
    _check(prev == authority_set_hash, "the authority set is not the committed one")
    # 2) the precommit message
    pc = ac.decode_precommit(just.encoded_precommit)
    _check(pc["block_number"] == block_number and pc["authority_set_id"] == authority_set_id and pc["block_hash"] == block_hash,
           "the precommit is not for the target block under this authority set")
    # 3) the signatures of everyone marked as signed
    have = {(bytes(pk), bytes(m), bytes(sg)) for pk, m, sg in verified}
    for pk, sg, signed in zip(just.pubkeys, just.signatures, just.validator_signed):
        _check(not signed or (bytes(pk), just.encoded_precommit, bytes(sg)) in have, "a signature marked as present was not verified")
    # 4) more than 2/3 signed
    _check(sum(1 for s in just.validator_signed if s) * 3 > just.num_authorities * 2, "not more than 2/3 of the authorities signed")


def outer_statement(input_bytes: bytes, subchain: Subchain, just: Justification, chain_msgs, chain_digests, verified) -> bytes:
    """`HeaderRangeCircuit::define` above the MapReduce (header_range.rs:31-58): the two assertions that tie the subchain to the input
    (subchain_verification.rs:292-296), verify_simple_justification on the target header -> the 96 output bytes."""
    inp = ac.unpack_header_range_input(input_bytes)
    _check(inp["trusted_header"] == subchain.start_parent, "the header chain does not start at the trusted header")
    _check(inp["target_block"] == subchain.end_block, "the header chain does not end at the target block")
    justification_checks(inp["target_block"], subchain.end_header_hash, inp["authority_set_id"], inp["authority_set_hash"], just,
                         chain_msgs, chain_digests, verified)
    return ac.pack_header_range_output(subchain.end_header_hash, subchain.state_merkle_root, subchain.data_merkle_root)


def rotate_statement(input_bytes: bytes, header: bytes, header_hash: bytes, just: Justification, chain_msgs, chain_digests, verified,
                     start_position: int, new_pubkeys, new_chain_msgs, new_chain_digests) -> bytes:
    """`RotateCircuit::define` / `rotate` (/root/reference/circuits/rotate.rs:80-109, builder/rotate.rs:278-323): the epoch end header
    hashes to `header_hash` (the BLAKE2b table's digest of exactly these bytes — the caller checks that), the CURRENT authority set
    justifies it, the header's consensus log holds the NEW set (verify_epoch_end_header: avail_codec), and the output is the new
    set's commitment — the last digest of the chain the SHA-256 table hashed over the new keys."""
    inp = ac.unpack_rotate_input(input_bytes)
    number = ac.decode_header(header)["block_number"]
    justification_checks(number, bytes(header_hash), inp["authority_set_id"], inp["authority_set_hash"], just, chain_msgs, chain_digests, verified)
    try:
        ac.verify_epoch_end_header(header, start_position, len(new_pubkeys), new_pubkeys)
    except (ValueError, IndexError) as e:
        raise StatementError(f"the epoch end header does not announce this authority set: {e}") from None
    _check(len(new_chain_msgs) == len(new_chain_digests) == len(new_pubkeys), "the new authority chain has one hash per authority")
    prev = b""
    for pk, m, d in zip(new_pubkeys, new_chain_msgs, new_chain_digests):
        _check(bytes(m) == prev + bytes(pk), "the new authority chain hashed something else than commitment || key")
        prev = bytes(d)
    return prev


# ---- synthetic requests ---------------------------------------------------------------------------------------------------------------
@dataclass
class Request:
    input_bytes: bytes            # the 80 bytes of the function call
    capacity: int                 # headers the circuit holds (HEADERS_PER_MAP x map jobs: 512 for header_range_512)
    headers: list                 # capacity encoded headers; b"" past the target block
    keys: list                    # [(secret, public)] of the distinct authorities (the set repeats them)
    seed: bytes
    _just: Justification = None
    _lock: threading.Lock = None

    @property
    def trusted_block(self) -> int:
        return ac.unpack_header_range_input(self.input_bytes)["trusted_block"]

    @property
    def target_block(self) -> int:
        return ac.unpack_header_range_input(self.input_bytes)["target_block"]

    def batch(self, job_index: int) -> list:
        return self.headers[job_index * HEADERS_PER_MAP:(job_index + 1) * HEADERS_PER_MAP]

    def justification(self) -> Justification:
        """signed on first use (RFC 8032 signing in Python: 0.2 s per distinct key) — the request's INPUT, made by validators, fetched
        from the chain by the reference (/root/reference/circuits/input/mod.rs: get_justification_from_block)"""
        with self._lock:
            if self._just is None:
                from . import eddsa_air as ea
                inp = ac.unpack_header_range_input(self.input_bytes)
                n_real = self.target_block - self.trusted_block
                target_hash = hashlib.blake2b(self.headers[n_real - 1], digest_size=32).digest()
                pc = ac.encode_precommit(target_hash, inp["target_block"], 1, inp["authority_set_id"])
                sigs = [ea.sign(sk, pc)[1] for sk, _ in self.keys]
                k = len(self.keys)
                na = self._num_authorities
                self._just = Justification(pc, [self.keys[i % k][1] for i in range(na)], [sigs[i % k] for i in range(na)], [True] * na)
            return self._just


def make_request(seed: bytes, capacity: int = 512, header_bytes: int = MAX_HEADER_SIZE, num_authorities: int = 300, distinct_keys: int = 8,
                 num_headers: int = None, authority_set_id: int = 7) -> Request:
    """A synthetic header_range request: `num_headers` (default: capacity) chained headers of `header_bytes` bytes after a trusted block,
    state / data roots and filler derived from the seed, `num_authorities` authorities cycling `distinct_keys` real key pairs."""
    num_headers = capacity if num_headers is None else num_headers
    assert 1 <= num_headers <= capacity and capacity % HEADERS_PER_MAP == 0 and 32 + 5 + 32 + 32 <= header_bytes <= MAX_HEADER_SIZE
    xof = hashlib.shake_256(b"vectorx header_range request|" + bytes(seed))
    trusted_block = 100_000 + int.from_bytes(xof.digest(4), "little") % 1_000_000        # compact mode 2 (4 bytes) throughout
    blob = hashlib.shake_256(b"vectorx headers|" + bytes(seed)).digest(32 + num_headers * header_bytes)
    parent = blob[:32]                                                                   # the trusted header's hash
    trusted_header, headers = parent, []
    for k in range(num_headers):
        raw = blob[32 + k * header_bytes:32 + (k + 1) * header_bytes]
        number = ac.encode_compact_u32(trusted_block + 1 + k)
        h = parent + number + raw[32 + len(number):]              # state root, middle and data root: the seed's bytes
        headers.append(h)
        parent = hashlib.blake2b(h, digest_size=32).digest()
    headers += [b""] * (capacity - num_headers)
    keys = [_authority_key(i) for i in range(distinct_keys)]       # the SET is fixed; requests differ in what it signs
    pubkeys = [keys[i % distinct_keys][1] for i in range(num_authorities)]
    inp = ac.pack_header_range_input(trusted_block, trusted_header, authority_set_id, ac.authority_set_commitment(pubkeys), trusted_block + num_headers)
    req = Request(inp, capacity, headers, keys, bytes(seed))
    req._lock = threading.Lock()
    req._num_authorities = num_authorities
    return req


@dataclass
class RotateRequest:
    input_bytes: bytes            # the 40 bytes of the function call: current authority set id, current authority set hash
    header: bytes                 # the epoch end header
    start_position: int           # where its consensus log starts (the byte in front of the flag)
    new_pubkeys: list
    keys: list
    seed: bytes
    _just: Justification = None
    _lock: threading.Lock = None

    def justification(self) -> Justification:
        with self._lock:
            if self._just is None:
                from . import eddsa_air as ea
                inp = ac.unpack_rotate_input(self.input_bytes)
                number = ac.decode_header(self.header)["block_number"]
                pc = ac.encode_precommit(hashlib.blake2b(self.header, digest_size=32).digest(), number, 1, inp["authority_set_id"])
                sigs = [ea.sign(sk, pc)[1] for sk, _ in self.keys]
                k, na = len(self.keys), self._num_authorities
                self._just = Justification(pc, [self.keys[i % k][1] for i in range(na)], [sigs[i % k] for i in range(na)], [True] * na)
            return self._just


def make_rotate_request(seed: bytes, num_authorities: int = 300, distinct_keys: int = 8, new_authorities: int = 300, authority_set_id: int = 7) -> RotateRequest:
    """A synthetic rotate request: an epoch end header announcing `new_authorities` keys (avail_codec.synthetic_epoch_end_header),
    justified by the current set of `num_authorities` authorities cycling `distinct_keys` real key pairs."""
    xof = hashlib.shake_256(b"vectorx rotate request|" + bytes(seed))
    number = 100_000 + int.from_bytes(xof.digest(4), "little") % 1_000_000
    header, start, new_pubkeys = ac.synthetic_epoch_end_header(bytes(seed), new_authorities, number)
    keys = [_authority_key(i) for i in range(distinct_keys)]
    pubkeys = [keys[i % distinct_keys][1] for i in range(num_authorities)]
    req = RotateRequest(ac.pack_rotate_input(authority_set_id, ac.authority_set_commitment(pubkeys)), header, start, new_pubkeys, keys, bytes(seed))
    req._lock = threading.Lock()
    req._num_authorities = num_authorities
    return req


def expected_rotate_output(req: RotateRequest) -> bytes:
    """the 32 output bytes computed WITHOUT the tables (avail_codec.rotate_output: the header checks + hashlib)"""
    return ac.rotate_output(req.header, req.start_position, req.new_pubkeys)


_keys, _keys_lock = {}, threading.Lock()


def _authority_key(i: int) -> tuple:
    """(secret, public) of synthetic authority i — derived once per process (0.2 s of Python curve arithmetic each)"""
    with _keys_lock:
        if i not in _keys:
            from . import eddsa_air as ea
            sk = hashlib.sha256(b"vectorx authority|" + bytes([i])).digest()
            _keys[i] = (sk, ea.sign(sk, b"")[0])
        return _keys[i]


def expected_output(req: Request) -> bytes:
    """the 96 output bytes computed WITHOUT the DAG: hashlib over the headers, avail_codec's commitments over the decoded roots"""
    n = req.target_block - req.trusted_block
    real = req.headers[:n]
    decoded = [ac.decode_header(h) for h in real]
    state, data = ac.header_range_commitments([d["state_root"] for d in decoded], [d["data_root"] for d in decoded], req.capacity)
    return ac.pack_header_range_output(hashlib.blake2b(real[-1], digest_size=32).digest(), state, data)


_cache, _cache_lock = {}, threading.Lock()


def cached_request(seed: bytes, rotate: bool = False, **shape):
    """one Request (RotateRequest) per (seed, shape) and process: every lane of a worker reads the same header chain"""
    key = (bytes(seed), rotate, tuple(sorted(shape.items())))
    with _cache_lock:
        req = _cache.get(key)
        if req is None:
            if len(_cache) >= 6:
                _cache.pop(next(iter(_cache)))
            req = _cache[key] = (make_rotate_request if rotate else make_request)(seed, **shape)
        return req
