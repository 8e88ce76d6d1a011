"""An Ed25519 scalar-multiplication AIR — caller-side stand-in for the elliptic-curve chip that Curta (starkyx v1.0.0,
/root/reference/Cargo.lock:7232-7249) puts under VectorX's outer / rotate proofs: the EdDSA verification of the GRANDPA
justification (/root/reference/circuits/builder/justification.rs:237, `curta_eddsa_verify_sigs_conditional`).  OWN AIR, NOT
CURTA'S (the starkyx sources are not in the reference tree).  The third shape next to the SHA-256 table (bits) and the BLAKE2b
table (bits + values): NON-NATIVE FIELD ARITHMETIC.  Every row is one instruction Z = X * Y + E (mod 2^255 - 19) of a
32-instruction straight-line program — one double-and-add step of [k]B in extended twisted-Edwards coordinates — over a file of nine
256-bit registers held as 32 byte limbs each:

    X(x) Y(x) + E(x) - Z(x) - Q(x) P(x) = (x - 256) W(x)          coefficient by coefficient (63 constraints of degree 2)

with the quotient Q and the carry polynomial W (62 coefficients in (-2^15, 2^15), stored offset as two bytes) as witnesses and
EVERY byte of Z, Q, W (188 per row) looked up in a 256-entry table through a log-derivative argument in the second commitment
round (94 pair helpers + the table helper + the running sum).  The operands are bound to the register file by the row type
(cyclic one-hot of 32), constants (1, -1, -2, 2 and the cached base point, or the identity's cached form when the scalar bit is 0)
enter through the Y slot, the result is written back.  Rows 28..31 of a step take the inverse of Z as a free (looked-up) witness,
check Z * ZI = 1 and leave the affine coordinates; the scalar's bits (MSB first, one per step) are packed into 32-bit words compared
with the public inputs at word boundaries.  Public inputs: the scalar's 32-bit words in the order they are consumed (most
significant first; a trace of n rows consumes n / 1024 of the 8 slots), then x and y of [k]B as 16 limbs of 16 bits each.
A trace of 2^degree_bits rows runs 2^degree_bits / 32 steps = that many scalar bits (2^13 rows: all 256).
`build_program(base=A)` bakes another point's cached form into the program instead of B's (the program digest is part of the
transcript, so that program IS the statement "[k] * A"): with one table of each kind an RFC 8032 signature verifies —
[S]B = R + [h]A — the hash and the final addition staying on the host (tests/test_ed25519_air.py).
Plain host code: it emits a constraint program (include/vxprover.h VX_OP_*), generates the trace and the second-round columns,
and is checked against an independent affine implementation of the curve and the RFC 8032 test vector; no GPU, no oracle.
"""
from __future__ import annotations

import numpy as np

from . import (VX_AIR_ALL_ROWS, VX_AIR_FIRST_ROW, VX_AIR_LAST_ROW, VX_AIR_TRANSITION, VX_OP_ADD, VX_OP_END, VX_OP_LDCH, VX_OP_LDP, VX_OP_MUL,
               VX_OP_SUB, Stark)
from .blake2b_air import _inv_of
from .sha256_air import P, _Emit

PERIOD = 32
Q25519 = (1 << 255) - 19
D_ED = (-121665 * pow(121666, Q25519 - 2, Q25519)) % Q25519
BX = 15112221349535400772501151409588531511454012693041857206046113283949847762202
BY = 46316835694926478169428394003475163141307993866256225615783033603165251855960
NREG = 9
W_OFFSET = 1 << 15

# Y-slot constants; C0..C2 = the cached base point (y - x, y + x, 2 d x y) when the step's bit is 1, the identity's (1, 1, 0) otherwise
CONSTS = {"ONE": 1, "MINUS1": Q25519 - 1, "MINUS2": Q25519 - 2, "TWO": 2}
def cached_point(pt):
    """the cached form (y - x, y + x, 2 d x y) of an affine point, each paired with the identity's value for a zero scalar bit"""
    x, y = pt
    return {"C0": ((y - x) % Q25519, 1), "C1": ((y + x) % Q25519, 1), "C2": ((2 * D_ED * x * y) % Q25519, 0)}


CACHED = cached_point((BX, BY))

R = lambda i: ("r", i)       # noqa: E731
K = lambda name: ("c", name)  # noqa: E731
FREE = "free"
# (X, Y, E, destination register): registers 0..3 = the point (X1, Y1, Z1, T1), 4..8 temporaries
OPS = [
    # ---- doubling (dbl-2008-hwcd, a = -1) ----
    (R(0), R(0), None, 4),           # A = X1^2
    (R(1), R(1), None, 5),           # B = Y1^2
    (R(2), R(2), None, 6),           # ZZ = Z1^2
    (R(0), K("ONE"), R(1), 7),       # S = X1 + Y1
    (R(7), R(7), None, 7),           # S2 = S^2
    (R(4), K("ONE"), R(5), 8),       # AB = A + B
    (R(8), K("MINUS1"), R(7), 7),    # E = S2 - AB
    (R(4), K("MINUS1"), R(5), 4),    # G = B - A
    (R(6), K("MINUS2"), R(4), 6),    # F = G - 2 ZZ
    (R(8), K("MINUS1"), None, 8),    # H = -(A + B)
    (R(7), R(6), None, 0),           # X3 = E F
    (R(4), R(8), None, 1),           # Y3 = G H
    (R(7), R(8), None, 3),           # T3 = E H
    (R(6), R(4), None, 2),           # Z3 = F G
    # ---- mixed addition of the cached base point, or of the identity when bit = 0 ----
    (R(0), K("MINUS1"), R(1), 4),    # Y1 - X1
    (R(0), K("ONE"), R(1), 5),       # Y1 + X1
    (R(4), K("C0"), None, 4),        # A = (Y1 - X1) c0
    (R(5), K("C1"), None, 5),        # B = (Y1 + X1) c1
    (R(3), K("C2"), None, 6),        # C = T1 c2
    (R(2), K("TWO"), None, 7),       # D = 2 Z1
    (R(4), K("MINUS1"), R(5), 8),    # E = B - A
    (R(4), K("ONE"), R(5), 4),       # H = B + A
    (R(6), K("MINUS1"), R(7), 5),    # F = D - C
    (R(6), K("ONE"), R(7), 6),       # G = D + C
    (R(8), R(5), None, 0),           # X3 = E F
    (R(6), R(4), None, 1),           # Y3 = G H
    (R(8), R(4), None, 3),           # T3 = E H
    (R(5), R(6), None, 2),           # Z3 = F G
    # ---- affine coordinates ----
    FREE,                            # ZI (a looked-up witness) -> r4
    (R(2), R(4), None, 5),           # Z1 ZI, must be 1
    (R(0), R(4), None, 6),           # x = X1 ZI
    (R(1), R(4), None, 7),           # y = Y1 ZI
]
ROW_FREE, ROW_ONE = 28, 29
FREE_DST = 4
assert len(OPS) == PERIOD


class Cols:
    SEL = 0                    # one-hot row type, 32
    REG = 32                   # REG + 32 r + i: byte i of register r
    X = REG + 32 * NREG        # operand slots and the result, 32 bytes each
    Y = X + 32
    Z = Y + 32
    Q = Z + 32                 # quotient
    W = Q + 32                 # carry coefficient k (0..61): W + 2 k (low byte), W + 2 k + 1 (high byte) of w_k + 2^15
    BIT = W + 124
    KACC = BIT + 1             # the bits of the current scalar word so far
    BND = KACC + 1             # 1 on the last row of the last step of a scalar word
    POS = BND + 1              # one-hot position of the step inside its scalar word, 32
    J = POS + 32               # one-hot scalar word being filled, 8
    TBL = J + 8
    MULT = TBL + 1
    N = MULT + 1
    NLOOK = 188                # looked-up byte columns: Z, Q, W (contiguous from Z)
    AUX_H = N                  # 94 pair helpers
    AUX_HT = N + 94
    AUX_ACC = N + 95
    NAUX = 96


def _bytes(v):
    return list(int(v).to_bytes(32, "little"))


def _sel_rows(pred):
    return [t for t, op in enumerate(OPS) if op != FREE and pred(op)]


def build_program(base=None):
    """-> (program words, number of constraints); one program for every trace length (1024 rows per scalar word).
    `base`: an affine point other than the curve's base point — its cached form is baked into the program as constants, i.e. the
    program (whose digest the transcript observes) is the statement "[k] * THIS point"."""
    C = Cols
    CACHED = cached_point(base) if base is not None else globals()["CACHED"]
    e = _Emit(scratch=24)
    ONE, ZERO, GAMMA, S31, BND, BITr, NOTFREE, C256 = 63, 62, 61, 60, 59, 58, 57, 56
    PH = list(range(24, 56))       # phase registers: selector sums, recomputed at the start of each phase
    e.ldi(ONE, 1)
    e.ldi(ZERO, 0)
    e.ldi(C256, 256)
    e.ins(VX_OP_LDCH, GAMMA, 0)
    e.ldw(C.SEL + 31, dst=S31)
    e.ldw(C.BND, dst=BND)
    e.ldw(C.BIT, dst=BITr)
    e.op(VX_OP_SUB, ONE, e.ldw(C.SEL + ROW_FREE), NOTFREE)
    e.release(0)
    npush = 0
    tmp = e.tmp

    def push(r, kind):
        nonlocal npush
        e.push(r, kind)
        npush += 1

    def sum_sel(rows, dst):
        m0 = e.top
        first = True
        for r in rows:
            x = e.ldw(C.SEL + r)
            e.op(VX_OP_ADD, x, ZERO if first else dst, dst)
            first = False
            e.release(m0)
        if first:
            e.op(VX_OP_ADD, ZERO, ZERO, dst)

    def lin(dst, terms):
        """dst = sum of (selector register) * (column) over terms; zero terms -> 0"""
        m0 = e.top
        first = True
        for sel, col in terms:
            v = e.ldw(col)
            e.op(VX_OP_MUL, v, sel, v)
            e.op(VX_OP_ADD, v, ZERO if first else dst, dst)
            first = False
            e.release(m0)
        if first:
            e.op(VX_OP_ADD, ZERO, ZERO, dst)

    # ---- phase 1: X slot = the register the row type names ----
    xregs = sorted({op[0][1] for op in OPS if op != FREE})
    for n_, r in enumerate(xregs):
        sum_sel(_sel_rows(lambda op, r=r: op[0] == R(r)), PH[n_])
    for i in range(32):
        m0 = e.top
        t = tmp()
        lin(t, [(PH[n_], C.REG + 32 * r + i) for n_, r in enumerate(xregs)])
        e.op(VX_OP_SUB, e.ldw(C.X + i), t, t)
        push(t, VX_AIR_ALL_ROWS)
        e.release(m0)
    # ---- phase 2: Y slot = a register, a constant, or the cached point picked by the bit ----
    yregs = sorted({op[1][1] for op in OPS if op != FREE and op[1][0] == "r"})
    for n_, r in enumerate(yregs):
        sum_sel(_sel_rows(lambda op, r=r: op[1] == R(r)), PH[n_])
    cbase = len(yregs)
    cnames = list(CONSTS) + list(CACHED)
    for n_, name in enumerate(cnames):
        sum_sel(_sel_rows(lambda op, name=name: op[1] == K(name)), PH[cbase + n_])
    for i in range(32):
        m0 = e.top
        t = tmp()
        lin(t, [(PH[n_], C.REG + 32 * r + i) for n_, r in enumerate(yregs)])
        for n_, name in enumerate(cnames):
            if name in CONSTS:
                b = _bytes(CONSTS[name])[i]
                if b == 0:
                    continue
                m1 = e.top
                c = tmp()
                e.ldi(c, b)
            else:
                on, off = (_bytes(v)[i] for v in CACHED[name])
                if on == 0 and off == 0:
                    continue
                m1 = e.top
                c = tmp()
                e.ldi(c, (on - off) % P)
                e.op(VX_OP_MUL, c, BITr, c)
                c2 = tmp()
                e.ldi(c2, off)
                e.op(VX_OP_ADD, c, c2, c)
            e.op(VX_OP_MUL, c, PH[cbase + n_], c)
            e.op(VX_OP_ADD, t, c, t)
            e.release(m1)
        e.op(VX_OP_SUB, e.ldw(C.Y + i), t, t)
        push(t, VX_AIR_ALL_ROWS)
        e.release(m0)
    # ---- phase 3: the multiply-add relation, coefficient by coefficient (disabled on the free row) ----
    eregs = sorted({op[2][1] for op in OPS if op != FREE and op[2] is not None})
    for n_, r in enumerate(eregs):
        sum_sel(_sel_rows(lambda op, r=r: op[2] == R(r)), PH[n_])
    pb = _bytes(Q25519)
    assert pb[0] == 237 and pb[31] == 127 and all(b == 255 for b in pb[1:31])
    c237, c127, c255, coff = PH[8], PH[9], PH[10], PH[11]
    e.ldi(c237, 237)
    e.ldi(c127, 127)
    e.ldi(c255, 255)
    e.ldi(coff, W_OFFSET)

    def carry(k, dst):
        """dst = w_k = lo + 256 hi - 2^15"""
        m0 = e.top
        hi = e.ldw(C.W + 2 * k + 1)
        e.op(VX_OP_MUL, hi, C256, hi)
        e.op(VX_OP_ADD, hi, e.ldw(C.W + 2 * k), dst)
        e.op(VX_OP_SUB, dst, coff, dst)
        e.release(m0)

    for k in range(63):
        m0 = e.top
        d = tmp()
        first = True
        for i in range(max(0, k - 31), min(31, k) + 1):
            m1 = e.top
            t = e.op(VX_OP_MUL, e.ldw(C.X + i), e.ldw(C.Y + k - i))
            e.op(VX_OP_ADD, t, ZERO if first else d, d)
            first = False
            e.release(m1)
        if k < 32:
            m1 = e.top
            t = tmp()
            lin(t, [(PH[n_], C.REG + 32 * r + k) for n_, r in enumerate(eregs)])
            e.op(VX_OP_ADD, d, t, d)
            e.op(VX_OP_SUB, d, e.ldw(C.Z + k), d)
            e.release(m1)
        # (Q P)_k = 237 Q_k + 255 sum_{1 <= k - j <= 30} Q_j + 127 Q_{k - 31}
        m1 = e.top
        if k < 32:
            t = e.op(VX_OP_MUL, e.ldw(C.Q + k), c237)
            e.op(VX_OP_SUB, d, t, d)
            e.release(m1)
        lo, hi = max(0, k - 30), min(31, k - 1)
        if lo <= hi:
            s = tmp()
            for n_, j in enumerate(range(lo, hi + 1)):
                q = e.ldw(C.Q + j)
                e.op(VX_OP_ADD, q, ZERO if n_ == 0 else s, s)
                e.release(s + 1)
            e.op(VX_OP_MUL, s, c255, s)
            e.op(VX_OP_SUB, d, s, d)
            e.release(m1)
        if 0 <= k - 31 < 32:
            t = e.op(VX_OP_MUL, e.ldw(C.Q + k - 31), c127)
            e.op(VX_OP_SUB, d, t, d)
            e.release(m1)
        # d_k = w_{k-1} - 256 w_k   (w_{-1} = w_62 = 0)
        if k >= 1:
            w = tmp()
            carry(k - 1, w)
            e.op(VX_OP_SUB, d, w, d)
            e.release(m1)
        if k <= 61:
            w = tmp()
            carry(k, w)
            e.op(VX_OP_MUL, w, C256, w)
            e.op(VX_OP_ADD, d, w, d)
            e.release(m1)
        e.op(VX_OP_MUL, d, NOTFREE, d)
        push(d, VX_AIR_ALL_ROWS)
        e.release(m0)
    # ---- phase 4: write-back, Z1 ZI = 1, first row = the identity point ----
    for r in range(NREG):
        rows = [t for t, op in enumerate(OPS) if (op == FREE and r == FREE_DST) or (op != FREE and op[3] == r)]
        sum_sel(rows, PH[r])
    for r in range(NREG):
        for i in range(32):
            m0 = e.top
            v, vn = e.ldw(C.REG + 32 * r + i), e.ldw(C.REG + 32 * r + i, nxt=True)
            t = e.op(VX_OP_SUB, e.ldw(C.Z + i), v)
            e.op(VX_OP_MUL, t, PH[r], t)
            e.op(VX_OP_ADD, t, v, t)
            e.op(VX_OP_SUB, vn, t, t)
            push(t, VX_AIR_TRANSITION)
            e.release(m0)
            m0 = e.top
            if r in (1, 2) and i == 0:
                push(e.op(VX_OP_SUB, e.ldw(C.REG + 32 * r + i), ONE), VX_AIR_FIRST_ROW)
            else:
                push(e.ldw(C.REG + 32 * r + i), VX_AIR_FIRST_ROW)
            e.release(m0)
    s_one = PH[10]
    e.ldw(C.SEL + ROW_ONE, dst=s_one)
    for i in range(32):
        m0 = e.top
        z = e.ldw(C.Z + i)
        if i == 0:
            z = e.op(VX_OP_SUB, z, ONE)
        push(e.op(VX_OP_MUL, z, s_one), VX_AIR_ALL_ROWS)
        e.release(m0)
    # ---- row type: cyclic one-hot ----
    for i in range(PERIOD):
        m0 = e.top
        push(e.op(VX_OP_SUB, e.ldw(C.SEL + i, nxt=True), e.ldw(C.SEL + (i - 1) % PERIOD)), VX_AIR_TRANSITION)
        r = e.ldw(C.SEL + i)
        if i == 0:
            r = e.op(VX_OP_SUB, r, ONE)
        push(r, VX_AIR_FIRST_ROW)
        e.release(m0)
    # ---- the scalar: one bit per step, MSB first, packed into 32-bit words that must equal the public inputs ----
    m0 = e.top
    t = e.op(VX_OP_SUB, BITr, ONE)
    push(e.op(VX_OP_MUL, t, BITr), VX_AIR_ALL_ROWS)
    not31 = e.op(VX_OP_SUB, ONE, S31)
    t = e.op(VX_OP_SUB, e.ldw(C.BIT, nxt=True), BITr)
    push(e.op(VX_OP_MUL, t, not31), VX_AIR_TRANSITION)
    e.release(m0)
    for i in range(32):
        m0 = e.top
        cur, prev, nx = e.ldw(C.POS + i), e.ldw(C.POS + (i - 1) % 32), e.ldw(C.POS + i, nxt=True)
        t = e.op(VX_OP_SUB, prev, cur)
        e.op(VX_OP_MUL, t, S31, t)
        e.op(VX_OP_ADD, t, cur, t)
        push(e.op(VX_OP_SUB, nx, t), VX_AIR_TRANSITION)
        push(e.op(VX_OP_SUB, cur, ONE) if i == 0 else cur, VX_AIR_FIRST_ROW)
        e.release(m0)
    m0 = e.top
    t = e.op(VX_OP_MUL, S31, e.ldw(C.POS + 31))
    push(e.op(VX_OP_SUB, BND, t), VX_AIR_ALL_ROWS)
    e.release(m0)
    for j in range(8):                               # word j = the j-th word consumed (most significant first)
        m0 = e.top
        cur, nx = e.ldw(C.J + j), e.ldw(C.J + j, nxt=True)
        below = e.ldw(C.J + j - 1) if j > 0 else ZERO
        t = e.op(VX_OP_SUB, below, cur)
        e.op(VX_OP_MUL, t, BND, t)
        e.op(VX_OP_ADD, t, cur, t)
        push(e.op(VX_OP_SUB, nx, t), VX_AIR_TRANSITION)
        push(e.op(VX_OP_SUB, cur, ONE) if j == 0 else cur, VX_AIR_FIRST_ROW)
        e.release(m0)
    m0 = e.top
    k, kn, bn = e.ldw(C.KACC), e.ldw(C.KACC, nxt=True), e.ldw(C.BIT, nxt=True)
    push(e.op(VX_OP_SUB, k, BITr), VX_AIR_FIRST_ROW)
    t = e.op(VX_OP_ADD, k, bn)
    e.op(VX_OP_MUL, t, S31, t)                       # s31 (K + bit')
    u = e.op(VX_OP_ADD, k, k)
    e.op(VX_OP_MUL, u, BND, u)                       # 2 K on a word boundary
    e.op(VX_OP_SUB, t, u, t)
    e.op(VX_OP_ADD, t, k, t)
    push(e.op(VX_OP_SUB, kn, t), VX_AIR_TRANSITION)  # K' = K + s31 (K + bit') - bnd 2 K
    w = tmp()
    first = True
    for j in range(8):
        m1 = e.top
        pi = tmp()
        e.ins(VX_OP_LDP, pi, j)
        e.op(VX_OP_MUL, pi, e.ldw(C.J + j), pi)
        e.op(VX_OP_ADD, pi, ZERO if first else w, w)
        first = False
        e.release(m1)
    e.op(VX_OP_SUB, k, w, w)
    push(e.op(VX_OP_MUL, w, BND), VX_AIR_ALL_ROWS)   # bnd (K - PI[word]) = 0
    e.release(m0)
    # ---- the result: x = register 6, y = the Z column of the last row, as 16 limbs of 16 bits ----
    for j in range(16):
        for base, pi0 in ((C.REG + 32 * 6, 8), (C.Z, 24)):
            m0 = e.top
            t = e.op(VX_OP_MUL, e.ldw(base + 2 * j + 1), C256)
            e.op(VX_OP_ADD, t, e.ldw(base + 2 * j), t)
            pi = tmp()
            e.ins(VX_OP_LDP, pi, pi0 + j)
            push(e.op(VX_OP_SUB, t, pi), VX_AIR_LAST_ROW)
            e.release(m0)
    # ---- the byte table and the lookups of every byte of Z, Q, W ----
    m0 = e.top
    tb, tbn = e.ldw(C.TBL), e.ldw(C.TBL, nxt=True)
    inc = e.op(VX_OP_SUB, tbn, tb)
    e.op(VX_OP_SUB, inc, ONE, inc)
    push(e.op(VX_OP_MUL, inc, tbn), VX_AIR_TRANSITION)
    c255 = tmp()
    e.ldi(c255, 255)
    t = e.op(VX_OP_SUB, tb, c255)
    push(e.op(VX_OP_MUL, t, inc), VX_AIR_TRANSITION)
    push(tb, VX_AIR_FIRST_ROW)
    e.release(m0)
    m0 = e.top
    acc, accn = e.ldw(C.AUX_ACC), e.ldw(C.AUX_ACC, nxt=True)
    step = e.op(VX_OP_SUB, accn, acc)
    for q in range(C.NLOOK // 2):
        m1 = e.top
        g0 = e.op(VX_OP_SUB, GAMMA, e.ldw(C.Z + 2 * q))
        g1 = e.op(VX_OP_SUB, GAMMA, e.ldw(C.Z + 2 * q + 1))
        h = e.ldw(C.AUX_H + q)
        e.op(VX_OP_SUB, step, h, step)
        t = e.op(VX_OP_MUL, g0, g1)
        e.op(VX_OP_MUL, t, h, t)
        e.op(VX_OP_SUB, t, g0, t)
        e.op(VX_OP_SUB, t, g1, t)
        push(t, VX_AIR_ALL_ROWS)
        e.release(m1)
    gt = e.op(VX_OP_SUB, GAMMA, e.ldw(C.TBL))
    ht = e.ldw(C.AUX_HT)
    e.op(VX_OP_ADD, step, ht, step)
    t = e.op(VX_OP_MUL, ht, gt)
    push(e.op(VX_OP_SUB, t, e.ldw(C.MULT)), VX_AIR_ALL_ROWS)
    push(step, VX_AIR_TRANSITION)
    push(acc, VX_AIR_FIRST_ROW)
    push(acc, VX_AIR_LAST_ROW)
    e.release(m0)
    e.ins(VX_OP_END)
    return e.w, npush


# ---------------------------------------------------------------------------------------------------------------------
def _const_value(name, bit, cached=None):
    if name in CONSTS:
        return CONSTS[name]
    on, off = (cached or CACHED)[name]
    return on if bit else off


def mul_add_witness(x, y, ev, z):
    """-> (quotient bytes, 124 carry bytes) of x y + e = z + q p"""
    q, rem = divmod(x * y + ev - z, Q25519)
    assert rem == 0 and 0 <= q < (1 << 256)
    xb, yb, eb, zb, qb, pb = (_bytes(v) for v in (x, y, ev, z, q, Q25519))
    wb = []
    prev = 0
    for k in range(63):
        d = sum(xb[i] * yb[k - i] for i in range(max(0, k - 31), min(31, k) + 1)) - sum(qb[j] * pb[k - j] for j in range(max(0, k - 31), min(31, k) + 1))
        if k < 32:
            d += eb[k] - zb[k]
        t = prev - d
        assert t % 256 == 0
        prev = t // 256
        if k <= 61:
            assert -W_OFFSET < prev < W_OFFSET
            wb += [(prev + W_OFFSET) & 255, (prev + W_OFFSET) >> 8]
    assert prev == 0
    return qb, wb


def generate_trace(degree_bits: int, scalar: int, base=None) -> tuple:
    """-> (trace [N][n] uint64, public inputs [40], (x, y) of [scalar]B — or of [scalar]base).  The scalar has n / 32 bits."""
    C = Cols
    cached = cached_point(base) if base is not None else None
    n = 1 << degree_bits
    nbits = n // PERIOD
    assert 0 <= scalar < (1 << nbits) and 10 <= degree_bits <= 13
    t = np.zeros((C.N, n), dtype=np.uint64)
    reg = [0, 1, 1, 0] + [0] * (NREG - 4)
    look = np.zeros(256, dtype=np.int64)
    kacc = 0
    xa = ya = 0
    for step in range(nbits):
        bit = (scalar >> (nbits - 1 - step)) & 1
        pos, word = step % 32, step // 32
        kacc = bit if pos == 0 else 2 * kacc + bit
        for r, op in enumerate(OPS):
            row = step * PERIOD + r
            t[C.SEL + r, row] = 1
            for k in range(NREG):
                t[C.REG + 32 * k:C.REG + 32 * k + 32, row] = _bytes(reg[k])
            t[C.BIT, row], t[C.KACC, row] = bit, kacc
            t[C.POS + pos, row] = 1
            t[C.J + word, row] = 1
            t[C.BND, row] = 1 if (r == 31 and pos == 31) else 0
            if op == FREE:
                x = y = 0
                z = pow(reg[2], Q25519 - 2, Q25519)
                dst = FREE_DST
                qb, wb = [0] * 32, [W_OFFSET & 255, W_OFFSET >> 8] * 62
            else:
                xs, ys, es, dst = op
                x = reg[xs[1]]
                y = reg[ys[1]] if ys[0] == "r" else _const_value(ys[1], bit, cached)
                ev = reg[es[1]] if es is not None else 0
                z = (x * y + ev) % Q25519
                qb, wb = mul_add_witness(x, y, ev, z)
            if r == ROW_ONE:
                assert z == 1
            t[C.X:C.X + 32, row] = _bytes(x)
            t[C.Y:C.Y + 32, row] = _bytes(y)
            t[C.Z:C.Z + 32, row] = _bytes(z)
            t[C.Q:C.Q + 32, row] = qb
            t[C.W:C.W + 124, row] = wb
            reg[dst] = z
            if r == 30:
                xa = z
            if r == 31:
                ya = z
    t[C.TBL] = np.arange(n, dtype=np.uint64) % 256
    look = np.bincount(t[C.Z:C.Z + C.NLOOK, :n - 1].astype(np.int64).reshape(-1), minlength=256)
    t[C.MULT, :256] = look.astype(np.uint64)
    limbs16 = lambda v: [(v >> (16 * j)) & 0xFFFF for j in range(16)]       # noqa: E731
    nw = nbits // 32
    words = [(scalar >> (32 * (nw - 1 - j))) & 0xFFFFFFFF for j in range(nw)] + [0] * (8 - nw)
    pis = np.array(words + limbs16(xa) + limbs16(ya), dtype=np.uint64)
    return t, pis, (xa, ya)


def aux_columns(trace, chal):
    """second-round columns [94 pair helpers, ht, acc] for the challenge gamma"""
    C = Cols
    n = trace.shape[1]
    g = int(chal[0])
    inv = _inv_of(g, trace[C.Z:C.Z + C.NLOOK].reshape(-1)).reshape(C.NLOOK, n)
    h = [(inv[2 * q] + inv[2 * q + 1]) % P for q in range(C.NLOOK // 2)]
    ht = (trace[C.MULT].astype(object) * _inv_of(g, trace[C.TBL])) % P
    step = (-ht) % P
    for c in h:
        step = (step + c) % P
    acc = np.zeros(n, dtype=object)
    run = 0
    for i in range(n):
        acc[i] = run
        run = (run + int(step[i])) % P
    return np.stack([np.array(c, dtype=np.uint64) for c in h + [ht, acc]])


def make_stark(degree_bits: int, base=None, **cfg) -> Stark:
    assert 10 <= degree_bits <= 13, "one scalar word = 32 steps = 1024 rows; 2^13 rows = 256 bits"
    prog, _ = build_program(base)
    cfg.setdefault("rate_bits", 1)
    return Stark(degree_bits, Cols.N, 40, prog, constraint_degree=3, num_aux_columns=Cols.NAUX, num_aux_challenges=1, aux_fn=aux_columns, **cfg)


# ---- an independent affine implementation of the curve, for the tests ------------------------------------------------
def affine_add(p1, p2):
    (x1, y1), (x2, y2) = p1, p2
    k = D_ED * x1 * x2 * y1 * y2 % Q25519
    x3 = (x1 * y2 + x2 * y1) * pow(1 + k, Q25519 - 2, Q25519) % Q25519
    y3 = (y1 * y2 + x1 * x2) * pow(1 - k, Q25519 - 2, Q25519) % Q25519
    return x3, y3


def affine_scalar_mult(k, pt=(BX, BY)):
    acc = (0, 1)
    while k:
        if k & 1:
            acc = affine_add(acc, pt)
        pt = affine_add(pt, pt)
        k >>= 1
    return acc


def compress(pt):
    x, y = pt
    return (y | ((x & 1) << 255)).to_bytes(32, "little")
