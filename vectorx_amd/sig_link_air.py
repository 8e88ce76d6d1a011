"""The link table of the signature bus (round 5): what ties the three statements about one Ed25519 signature together.

Three tables speak about a signature with public key A, signature (R, S) and message M, each in its own tuple FORMAT (the last
element of a tuple is a constant tag naming the format, so that no format can alias another):

  * the SHA-512 table (`sha512_air`, bus variant) SENDS   (R encoding, A encoding, digest, TAG_SHA512)    — it hashed a message that
    starts with R || A and this is its digest;
  * the EdDSA table (`eddsa_air`, full program) SENDS     (A encoding, S, digest, R encoding, TAG_EDDSA)  — A and R decompress, S < L,
    and [S]B - [digest mod L]A = R;
  * a verifier (the plonky2 circuit that embeds the STARKs; `eddsa_air.make_sink` in the tests) RECEIVES  (A encoding, S, R encoding,
    TAG_VERIFIER) — the bytes of the public key and of the signature, nothing else.

This table has one row per signature holding (A encoding, S, digest, R encoding) and a flag; a flagged row RECEIVES the first two
tuples and SENDS the third.  The bus balances iff, for every signature the verifier holds, the SHA-512 table produced the digest of a
message starting with exactly that R and A, and the EdDSA table accepted exactly that (A, S, digest, R): the digest — the one value
the verifier does not hold — is existentially bound by the two tables.  (What stays with the caller: that the REST of the hashed
message is the M it means — in VectorX the precommit message, /root/reference/circuits/builder/justification.rs:140-156.)
Own framing, like the tables it links (/root/reference/circuits/builder/justification.rs:237-243 verifies whole signatures inside
Curta's gadget).  Plain host code: a constraint program, a trace, second-round columns; no GPU, no oracle."""
from __future__ import annotations

import numpy as np

from . import VX_AIR_ALL_ROWS, VX_AIR_FIRST_ROW, VX_AIR_LAST_ROW, VX_AIR_TRANSITION, VX_OP_ADD, VX_OP_END, VX_OP_LDCH, VX_OP_LDP, VX_OP_MUL, VX_OP_SUB, Stark
from . import hostfield as hf
from .eddsa_air import TAG_EDDSA, TAG_SHA512, TAG_VERIFIER, _horner
from .sha256_air import P, _Emit

AENC, SW, DW, RENC, FLAG, N = 0, 8, 16, 32, 40, 41       # 8 + 8 + 16 + 8 words, the flag
AUX_U1, AUX_U2, AUX_U3, AUX_ACC, NAUX = N, N + 1, N + 2, N + 3, 4
FORMATS = {      # format -> (columns in tuple order, tag)
    "eddsa": (list(range(AENC, AENC + 8)) + list(range(SW, SW + 8)) + list(range(DW, DW + 16)) + list(range(RENC, RENC + 8)), TAG_EDDSA),
    "sha512": (list(range(RENC, RENC + 8)) + list(range(AENC, AENC + 8)) + list(range(DW, DW + 16)), TAG_SHA512),
    "verifier": (list(range(AENC, AENC + 8)) + list(range(SW, SW + 8)) + list(range(RENC, RENC + 8)), TAG_VERIFIER),
}


def build_program():
    """aux challenges [unused, beta, gamma_bus] (shared with the other tables of the bus); aux public input 0 = the closing sum"""
    e = _Emit(scratch=40)
    ONE, BETA, G = 63, 62, 61
    e.ldi(ONE, 1)
    e.ins(VX_OP_LDCH, BETA, 1)
    e.ins(VX_OP_LDCH, G, 2)
    flag = e.ldw(FLAG)
    t = e.op(VX_OP_SUB, flag, ONE)
    e.push(e.op(VX_OP_MUL, t, flag), VX_AIR_ALL_ROWS)
    acc, accn = e.ldw(AUX_ACC), e.ldw(AUX_ACC, nxt=True)
    step = e.op(VX_OP_SUB, accn, acc)
    for name, ucol, sign in (("eddsa", AUX_U1, -1), ("sha512", AUX_U2, -1), ("verifier", AUX_U3, +1)):
        m0 = e.top
        cols, tag = FORMATS[name]
        tup = e.tmp()
        e.ldi(tup, tag)
        for col in reversed(cols):
            m1 = e.top
            e.op(VX_OP_MUL, tup, BETA, tup)
            e.op(VX_OP_ADD, tup, e.ldw(col), tup)
            e.release(m1)
        d = e.op(VX_OP_SUB, G, tup)
        u = e.ldw(ucol)
        t = e.op(VX_OP_MUL, u, d)
        e.push(e.op(VX_OP_SUB, t, flag), VX_AIR_ALL_ROWS)            # u (gamma - tuple) = flag
        e.op(VX_OP_SUB if sign > 0 else VX_OP_ADD, step, u, step)    # acc' - acc = u3 - u1 - u2
        e.release(m0)
    e.push(step, VX_AIR_TRANSITION)
    e.push(acc, VX_AIR_FIRST_ROW)
    closing = e.tmp()
    e.ins(VX_OP_LDP, closing, 0)
    e.push(e.op(VX_OP_SUB, acc, closing), VX_AIR_LAST_ROW)
    e.ins(VX_OP_END)
    return e.w


def row_of(public_key: bytes, signature: bytes, digest: bytes) -> list:
    """the 40 words of a signature's row: A's encoding, S (most significant word first, as the EdDSA table carries it), the digest, R's encoding"""
    le = lambda b: [int.from_bytes(b[4 * j:4 * j + 4], "little") for j in range(len(b) // 4)]   # noqa: E731
    return le(public_key) + le(signature[32:])[::-1] + le(digest) + le(signature[:32])


NVERIFIER = len(FORMATS["verifier"][0]) + 1      # 24 words + the format tag


def verifier_tuple(public_key: bytes, signature: bytes) -> list:
    """what the verifier's side of the bus receives: the bytes of the public key and of the signature"""
    r = row_of(public_key, signature, bytes(64))
    return [r[c] for c in FORMATS["verifier"][0]] + [TAG_VERIFIER]


def aux_columns(trace, chal):
    n = trace.shape[1]
    beta, g = int(chal[1]), int(chal[2])
    rows = np.nonzero(trace[FLAG])[0]
    us = []
    for name in ("eddsa", "sha512", "verifier"):
        cols, tag = FORMATS[name]
        u = np.zeros(n, dtype=np.uint64)
        if rows.size:
            elems = np.concatenate([trace[np.array(cols)][:, rows].T, np.full((rows.size, 1), tag, dtype=np.uint64)], axis=1)
            u[rows] = hf.invmod(hf.submod(np.full(rows.size, g, dtype=np.uint64), _horner(elems, beta)))
        us.append(u)
    step = hf.submod(hf.submod(us[2], us[0]), us[1])
    acc, _ = hf.exclusive_prefix_sum(step)
    return np.stack(us + [acc]), np.array([int(acc[n - 1])], dtype=np.uint64)


def aux_program():
    """the GPU form of `aux_columns`: three fractions flag / (gamma_bus - tuple_k), one running sum u3 - u1 - u2"""
    from . import AuxProgram
    e = _Emit(scratch=40)
    BETA, G = 62, 61
    e.ins(VX_OP_LDCH, BETA, 1)
    e.ins(VX_OP_LDCH, G, 2)
    for name in ("eddsa", "sha512", "verifier"):
        m0 = e.top
        cols, tag = FORMATS[name]
        tup = e.tmp()
        e.ldi(tup, tag)
        for col in reversed(cols):
            m1 = e.top
            e.op(VX_OP_MUL, tup, BETA, tup)
            e.op(VX_OP_ADD, tup, e.ldw(col), tup)
            e.release(m1)
        e.push(e.ldw(FLAG), 0)
        e.push(e.op(VX_OP_SUB, G, tup), 0)
        e.release(m0)
    e.ins(VX_OP_END)
    return AuxProgram(N, 3, e.w, 3, [[-1, -1, 1]], fraction_out=[0, 1, 2], sum_out=[3], api_sums=(0,))


def make_stark(degree_bits: int, **cfg) -> Stark:
    cfg.setdefault("rate_bits", 1)
    stark = Stark(degree_bits, N, 0, build_program(), constraint_degree=3, num_aux_columns=NAUX, num_aux_challenges=3, aux_fn=aux_columns,
                  num_aux_public_inputs=1, **cfg)
    stark.aux_program = aux_program()
    return stark


def trace_of(rows, degree_bits: int) -> np.ndarray:
    """[41][2^degree_bits]: one flagged row per signature (`rows` = [row_of(...)]), the rest empty"""
    n = 1 << degree_bits
    assert len(rows) <= n - 1
    t = np.zeros((N, n), dtype=np.uint64)
    if rows:
        t[:FLAG, :len(rows)] = np.array(rows, dtype=np.uint64).T
        t[FLAG, :len(rows)] = 1
    return t


def make_link(rows, degree_bits=None, **cfg):
    """-> (Stark, trace, public inputs (none)) of the link table for `rows` = [row_of(...)] (the last trace row stays empty: it is inert
    in a running sum)"""
    k = len(rows)
    db = max(4, (k + 1).bit_length()) if degree_bits is None else degree_bits
    n = 1 << db
    assert k <= n - 1
    t = np.zeros((N, n), dtype=np.uint64)
    for i, r in enumerate(rows):
        t[:FLAG, i] = np.array(r, dtype=np.uint64)
        t[FLAG, i] = 1
    cfg.setdefault("rate_bits", 1)
    stark = Stark(db, N, 0, build_program(), constraint_degree=3, num_aux_columns=NAUX, num_aux_challenges=3, aux_fn=aux_columns,
                  num_aux_public_inputs=1, **cfg)
    stark.aux_program = aux_program()
    return stark, t, np.zeros(0, dtype=np.uint64)
