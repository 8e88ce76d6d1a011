"""A SHA-512 AIR at chip density — the hash inside EdDSA (h = SHA-512(R || A || M): RFC 8032 section 5.1.7; Curta's EdDSA gadget proves
it next to the curve arithmetic, /root/reference/circuits/builder/justification.rs:237).  OWN AIR, NOT CURTA'S: the 64-bit sibling of
vectorx_amd/sha256_air.py with the one change 64-bit words force on a 64-bit field — every addition is done on 32-bit LIMBS
(low limb with carry into the high limb), as in the BLAKE2b table.  82 rows per 128-byte block under a cyclic one-hot row type
(80 rounds, the feed-forward row, the hand-over row); the working state a..h and a 16-word schedule window as bits (1536 columns),
one auxiliary bit per 3-input XOR and per Maj (320 columns), chaining value / digest latch as limb values, the six addition
carries (a', e', W' x low / high limb; values in [0, 8)) range-checked by a log-derivative lookup into an 8-entry table in the
second commitment round.  1995 + 5 columns.  Public inputs: the 16 limbs of the digest of the last message completed inside the trace.
Constants (K, IV) are derived here from the primes, not typed in.  Plain host code; checked against `hashlib.sha512`; no GPU, no oracle.
"""
from __future__ import annotations

import hashlib
import struct
from math import isqrt

import numpy as np

from . import (VX_AIR_ALL_ROWS, VX_AIR_FIRST_ROW, VX_AIR_LAST_ROW, VX_AIR_TRANSITION, VX_OP_ADD, VX_OP_END, VX_OP_LDCH, VX_OP_LDP, VX_OP_MUL,
               VX_OP_SUB, Stark)
from .sha256_air import P, _Emit

PERIOD = 82
M64 = (1 << 64) - 1


def _primes(n):
    ps, k = [], 2
    while len(ps) < n:
        if all(k % p for p in ps):
            ps.append(k)
        k += 1
    return ps


def _icbrt(n):
    x = int(round(n ** (1 / 3)))
    while x ** 3 > n:
        x -= 1
    while (x + 1) ** 3 <= n:
        x += 1
    return x


K512 = [_icbrt(p << 192) & M64 for p in _primes(80)]          # fractional parts of the cube roots of the first 80 primes
IV = [isqrt(p << 128) & M64 for p in _primes(8)]              # ... of the square roots of the first 8
assert K512[0] == 0x428a2f98d728ae22 and K512[79] == 0x6c44198c4a475817 and IV[0] == 0x6a09e667f3bcc908


class Cols:
    S = 0                      # S + 64 k + i: bit i of state word k (a b c d e f g h)
    WB = 512                   # WB + 64 k + i: bit i of W_{t-k}
    X0 = 1536                  # a_{i+28} ^ a_{i+34}
    X1 = 1600                  # e_{i+14} ^ e_{i+18}
    M = 1664                   # a_i b_i
    Y0 = 1728                  # w14_{i+1} ^ w14_{i+8}
    Y1 = 1792                  # w1_{i+19} ^ w1_{i+61}
    SEL = 1856                 # one-hot row type, 82
    H = 1938                   # chaining value: 8 words x 2 limbs
    D = 1954                   # last completed digest: 8 words x 2 limbs
    FFC = 1970                 # feed-forward carries: 8 words x 2 limbs (bits)
    NF = 1986
    CA, CE, CW = 1987, 1989, 1991    # + limb: carries of a', e', W' (values < 8)
    TBL, MULT = 1993, 1994
    N = 1995
    AUX_H, AUX_HT, AUX_ACC = 1995, 1998, 1999        # three pair helpers, the table helper, the running sum
    NAUX = 5
    # bus variant (round 5: the digest of R || A || M travels to the EdDSA table — vectorx_amd/sig_link_air.py): 17 more trace columns
    LW = 1995                  # the first 64 bytes of the CURRENT message as 16 little-endian 32-bit words (an Ed25519 R, then A), latched
    FIRST = 2011               # 1 on the rows of a message's first block
    N_BUS = 2012
    NAUX_BUS = 7               # + the bus helper and the bus sum


def _rotr(x, r):
    return ((x >> r) | (x << (64 - r))) & M64


TAG_SHA512 = 2                 # eddsa_air.TAG_SHA512: the last element of this table's tuple on the signature bus


def le_word_of_bits(base, j):
    """[(bit column, weight)]: little-endian 32-bit word j of the BYTE STRING whose big-endian 64-bit words sit, as bits, at base + 64 m + i
    (bit i of word m): byte q of word m is its bits 56 - 8 q .. 63 - 8 q"""
    m, q0 = j // 2, 4 * (j % 2)
    return [(base + 64 * m + 56 - 8 * (q0 + q) + t, 1 << (8 * q + t)) for q in range(4) for t in range(8)]


def build_program(bus=False):
    """-> (program words, number of constraints).  bus=True: the table also SENDS, when a message ends, the tuple (first 64 bytes of
    the message, digest, TAG_SHA512) — 16 + 16 little-endian 32-bit words — on the signature bus: for an Ed25519 verification the
    message is R || A || M, so the tuple is (R's encoding, A's encoding, the digest the EdDSA table reduces mod L).  Aux challenges
    [gamma_range, beta, gamma_bus]; two more second-round columns; the closing sum of the sends is aux public input 0."""
    C = Cols
    e = _Emit()
    N = C.N_BUS if bus else C.N
    AUX_H, AUX_HT, AUX_ACC, AUX_U, AUX_BUS = N, N + 3, N + 4, N + 5, N + 6
    ONE, TWO32, GAMMA, IS_ROUND, S80, S81, SCHED, NFr, ZERO = 63, 62, 61, 60, 59, 58, 57, 56, 55
    KLO, KHI = 54, 53
    e.ldi(ONE, 1)
    e.ldi(TWO32, 1 << 32)
    e.ldi(ZERO, 0)
    e.ins(VX_OP_LDCH, GAMMA, 0)
    npush = 0

    def push(r, kind):
        nonlocal npush
        e.push(r, kind)
        npush += 1

    e.ldw(C.SEL + 80, dst=S80)
    e.ldw(C.SEL + 81, dst=S81)
    e.op(VX_OP_SUB, ONE, S80, IS_ROUND)
    e.op(VX_OP_SUB, IS_ROUND, S81, IS_ROUND)
    e.ldw(C.NF, dst=NFr)
    mark = e.top
    first = True
    for i in range(15, 79):                            # the next row is a round >= 16
        r = e.ldw(C.SEL + i)
        e.op(VX_OP_ADD, r, ZERO if first else SCHED, SCHED)
        first = False
        e.release(mark)
    for limb, dst in ((0, KLO), (1, KHI)):
        first = True
        for i in range(80):
            r = e.ldw(C.SEL + i)
            k = e.tmp()
            e.ldi(k, (K512[i] >> (32 * limb)) & 0xFFFFFFFF)
            e.op(VX_OP_MUL, r, k, r)
            e.op(VX_OP_ADD, r, ZERO if first else dst, dst)
            first = False
            e.release(mark)

    def boolean(col):
        m0 = e.top
        r = e.ldw(col)
        t = e.op(VX_OP_SUB, r, ONE)
        e.op(VX_OP_MUL, t, r, t)
        push(t, VX_AIR_ALL_ROWS)
        e.release(m0)

    for c in range(C.S, C.S + 512):
        boolean(c)
    for c in range(C.WB, C.WB + 1024):
        boolean(c)
    for c in range(C.FFC, C.FFC + 16):
        boolean(c)
    boolean(C.NF)

    def xor_into(dst, x, y):
        m0 = e.top
        t = e.op(VX_OP_MUL, x, y)
        t2 = e.op(VX_OP_ADD, t, t)
        s = e.op(VX_OP_ADD, x, y)
        e.op(VX_OP_SUB, s, t2, dst)
        e.release(m0)

    def define_xor(col_out, col_x, col_y):
        m0 = e.top
        x, y, o = e.ldw(col_x), e.ldw(col_y), e.ldw(col_out)
        t = e.tmp()
        xor_into(t, x, y)
        e.op(VX_OP_SUB, o, t, t)
        push(t, VX_AIR_ALL_ROWS)
        e.release(m0)

    A, B, Cc, Dd, E, F, G, Hh = (C.S + 64 * k for k in range(8))
    W1, W14 = C.WB + 64 * 1, C.WB + 64 * 14
    for i in range(64):
        define_xor(C.X0 + i, A + (i + 28) % 64, A + (i + 34) % 64)
        define_xor(C.X1 + i, E + (i + 14) % 64, E + (i + 18) % 64)
        define_xor(C.Y0 + i, W14 + (i + 1) % 64, W14 + (i + 8) % 64)
        define_xor(C.Y1 + i, W1 + (i + 19) % 64, W1 + (i + 61) % 64)
        m0 = e.top
        a, b, m = e.ldw(A + i), e.ldw(B + i), e.ldw(C.M + i)
        t = e.op(VX_OP_MUL, a, b)
        e.op(VX_OP_SUB, m, t, t)
        push(t, VX_AIR_ALL_ROWS)
        e.release(m0)

    def limb(base, l, nxt=False, dst=None):
        """sum_i 2^i col[base + 32 l + i], i < 32"""
        acc = e.tmp() if dst is None else dst
        m0 = e.top
        e.ldw(base + 32 * l + 31, nxt, dst=acc)
        for i in range(30, -1, -1):
            e.op(VX_OP_ADD, acc, acc, acc)
            b = e.ldw(base + 32 * l + i, nxt)
            e.op(VX_OP_ADD, acc, b, acc)
            e.release(m0)
        return acc

    def limb_of(bit_expr, l, dst=None):
        acc = e.tmp() if dst is None else dst
        m0 = e.top
        bit_expr(32 * l + 31, acc)
        for i in range(30, -1, -1):
            e.op(VX_OP_ADD, acc, acc, acc)
            t = e.tmp()
            bit_expr(32 * l + i, t)
            e.op(VX_OP_ADD, acc, t, acc)
            e.release(m0)
        return acc

    def big_sigma0(i, dst):
        xor_into(dst, e.ldw(C.X0 + i), e.ldw(A + (i + 39) % 64))

    def big_sigma1(i, dst):
        xor_into(dst, e.ldw(C.X1 + i), e.ldw(E + (i + 41) % 64))

    def maj(i, dst):
        a, b, c, m = e.ldw(A + i), e.ldw(B + i), e.ldw(Cc + i), e.ldw(C.M + i)
        t = e.op(VX_OP_ADD, a, b)
        e.op(VX_OP_SUB, t, m, t)
        e.op(VX_OP_SUB, t, m, t)
        e.op(VX_OP_MUL, t, c, t)
        e.op(VX_OP_ADD, t, m, dst)

    def ch(i, dst):
        ee, f, g = e.ldw(E + i), e.ldw(F + i), e.ldw(G + i)
        t = e.op(VX_OP_SUB, f, g)
        e.op(VX_OP_MUL, t, ee, t)
        e.op(VX_OP_ADD, t, g, dst)

    def small_sigma0(i, dst):
        y = e.ldw(C.Y0 + i)
        if i + 7 < 64:
            xor_into(dst, y, e.ldw(W14 + i + 7))
        else:
            e.op(VX_OP_ADD, y, ZERO, dst)

    def small_sigma1(i, dst):
        y = e.ldw(C.Y1 + i)
        if i + 6 < 64:
            xor_into(dst, y, e.ldw(W1 + i + 6))
        else:
            e.op(VX_OP_ADD, y, ZERO, dst)

    # ---- the round, limb by limb: T1 = h + Sigma1(e) + Ch + K + W,  a' = T1 + Sigma0(a) + Maj,  e' = d + T1 ----
    for l in (0, 1):
        m0 = e.top
        t1 = limb(Hh, l)
        for r in (limb_of(big_sigma1, l), limb_of(ch, l)):
            e.op(VX_OP_ADD, t1, r, t1)
            e.release(r)
        e.op(VX_OP_ADD, t1, KLO if l == 0 else KHI, t1)
        r = limb(C.WB, l)
        e.op(VX_OP_ADD, t1, r, t1)
        e.release(r)
        t2 = limb_of(big_sigma0, l)
        r = limb_of(maj, l)
        e.op(VX_OP_ADD, t2, r, t2)
        e.release(r)
        for nxt_base, carry, extra in ((A, C.CA, t2), (E, C.CE, None)):
            m1 = e.top
            lhs = limb(nxt_base, l, nxt=True)                 # a'_l + 2^32 c_l - (terms_l + c_{l-1})
            c = e.ldw(carry + l)
            e.op(VX_OP_MUL, c, TWO32, c)
            e.op(VX_OP_ADD, lhs, c, lhs)
            e.op(VX_OP_SUB, lhs, t1, lhs)
            if extra is not None:
                e.op(VX_OP_SUB, lhs, extra, lhs)
            else:
                e.op(VX_OP_SUB, lhs, limb(Dd, l), lhs)
            if l == 1:
                e.op(VX_OP_SUB, lhs, e.ldw(carry), lhs)
            e.op(VX_OP_MUL, lhs, IS_ROUND, lhs)
            push(lhs, VX_AIR_TRANSITION)
            e.release(m1)
        e.release(m0)
    for k in (1, 2, 3, 5, 6, 7):
        for i in range(64):
            m0 = e.top
            cur, nx = e.ldw(C.S + 64 * (k - 1) + i), e.ldw(C.S + 64 * k + i, nxt=True)
            t = e.op(VX_OP_SUB, nx, cur)
            e.op(VX_OP_MUL, t, IS_ROUND, t)
            push(t, VX_AIR_TRANSITION)
            e.release(m0)
    # ---- message schedule ----
    for k in range(1, 16):
        for i in range(64):
            m0 = e.top
            cur, nx = e.ldw(C.WB + 64 * (k - 1) + i), e.ldw(C.WB + 64 * k + i, nxt=True)
            push(e.op(VX_OP_SUB, nx, cur), VX_AIR_TRANSITION)
            e.release(m0)
    for l in (0, 1):
        m0 = e.top
        lhs = limb(C.WB, l, nxt=True)
        cw = e.ldw(C.CW + l)
        e.op(VX_OP_MUL, cw, TWO32, cw)
        e.op(VX_OP_ADD, lhs, cw, lhs)
        for r in (limb_of(small_sigma1, l), limb(C.WB + 64 * 6, l), limb_of(small_sigma0, l), limb(C.WB + 64 * 15, l)):
            e.op(VX_OP_SUB, lhs, r, lhs)
        if l == 1:
            e.op(VX_OP_SUB, lhs, e.ldw(C.CW), lhs)
        e.op(VX_OP_MUL, lhs, SCHED, lhs)
        push(lhs, VX_AIR_TRANSITION)
        e.release(m0)
    # ---- row type ----
    for i in range(PERIOD):
        m0 = e.top
        push(e.op(VX_OP_SUB, e.ldw(C.SEL + i, nxt=True), e.ldw(C.SEL + (i - 1) % PERIOD)), VX_AIR_TRANSITION)
        r = e.ldw(C.SEL + i)
        if i == 0:
            r = e.op(VX_OP_SUB, r, ONE)
        push(r, VX_AIR_FIRST_ROW)
        e.release(m0)
    # ---- chaining value, feed-forward, hand-over, digest latch ----
    not81 = e.op(VX_OP_SUB, ONE, S81)
    for k in range(8):
        for l in (0, 1):
            hcol, dcol = C.H + 2 * k + l, C.D + 2 * k + l
            ivl = (IV[k] >> (32 * l)) & 0xFFFFFFFF
            m0 = e.top
            h, hn = e.ldw(hcol), e.ldw(hcol, nxt=True)
            t = e.op(VX_OP_SUB, hn, h)
            push(e.op(VX_OP_MUL, t, not81), VX_AIR_TRANSITION)                 # H' = H unless the row is 81
            e.release(m0)
            m0 = e.top
            sn = limb(C.S + 64 * k, l, nxt=True)                                  # row 80: S'_l + 2^32 c_l = H_l + S_l + c_{l-1}
            c = e.ldw(C.FFC + 2 * k + l)
            e.op(VX_OP_MUL, c, TWO32, c)
            e.op(VX_OP_ADD, sn, c, sn)
            e.op(VX_OP_SUB, sn, e.ldw(hcol), sn)
            e.op(VX_OP_SUB, sn, limb(C.S + 64 * k, l), sn)
            if l == 1:
                e.op(VX_OP_SUB, sn, e.ldw(C.FFC + 2 * k), sn)
            push(e.op(VX_OP_MUL, sn, S80), VX_AIR_TRANSITION)
            e.release(m0)
            m0 = e.top
            sk = limb(C.S + 64 * k, l)                                            # row 81: H' = nf IV + (1 - nf) S;  D' = D + s81 nf (S - D)
            iv = e.tmp()
            e.ldi(iv, ivl)
            t = e.op(VX_OP_SUB, iv, sk)
            e.op(VX_OP_MUL, t, NFr, t)
            e.op(VX_OP_ADD, t, sk, t)
            e.op(VX_OP_SUB, e.ldw(hcol, nxt=True), t, t)
            push(e.op(VX_OP_MUL, t, S81), VX_AIR_TRANSITION)
            dk, dn = e.ldw(dcol), e.ldw(dcol, nxt=True)
            u = e.op(VX_OP_SUB, sk, dk)
            e.op(VX_OP_MUL, u, NFr, u)
            e.op(VX_OP_MUL, u, S81, u)
            e.op(VX_OP_ADD, u, dk, u)
            push(e.op(VX_OP_SUB, dn, u), VX_AIR_TRANSITION)
            e.release(m0)
            m0 = e.top
            iv = e.tmp()
            e.ldi(iv, ivl)
            push(e.op(VX_OP_SUB, e.ldw(hcol), iv), VX_AIR_FIRST_ROW)
            push(e.op(VX_OP_SUB, limb(C.S + 64 * k, l), iv), VX_AIR_FIRST_ROW)
            push(e.ldw(dcol), VX_AIR_FIRST_ROW)
            pi = e.tmp()
            e.ins(VX_OP_LDP, pi, 2 * k + l)
            push(e.op(VX_OP_SUB, e.ldw(dcol), pi), VX_AIR_LAST_ROW)
            e.release(m0)
    for k in range(8):                                                            # row 81 -> next block: S' = S + nf (IV - S) bit by bit
        for i in range(64):
            m0 = e.top
            s, sn = e.ldw(C.S + 64 * k + i), e.ldw(C.S + 64 * k + i, nxt=True)
            t = e.op(VX_OP_SUB, ONE if (IV[k] >> i) & 1 else ZERO, s)
            e.op(VX_OP_MUL, t, NFr, t)
            e.op(VX_OP_ADD, t, s, t)
            e.op(VX_OP_SUB, sn, t, t)
            push(e.op(VX_OP_MUL, t, S81), VX_AIR_TRANSITION)
            e.release(m0)
    # ---- range check of the six carries: lookup into tbl = 0..7 repeating ----
    m0 = e.top
    tb, tbn = e.ldw(C.TBL), e.ldw(C.TBL, nxt=True)
    inc = e.op(VX_OP_SUB, tbn, tb)
    e.op(VX_OP_SUB, inc, ONE, inc)
    push(e.op(VX_OP_MUL, inc, tbn), VX_AIR_TRANSITION)
    seven = e.tmp()
    e.ldi(seven, 7)
    t = e.op(VX_OP_SUB, tb, seven)
    push(e.op(VX_OP_MUL, t, inc), VX_AIR_TRANSITION)
    push(tb, VX_AIR_FIRST_ROW)
    e.release(m0)
    m0 = e.top
    acc, accn = e.ldw(AUX_ACC), e.ldw(AUX_ACC, nxt=True)
    step = e.op(VX_OP_SUB, accn, acc)
    for q, base in enumerate((C.CA, C.CE, C.CW)):
        m1 = e.top
        g0 = e.op(VX_OP_SUB, GAMMA, e.ldw(base))
        g1 = e.op(VX_OP_SUB, GAMMA, e.ldw(base + 1))
        h = e.ldw(AUX_H + q)
        e.op(VX_OP_SUB, step, h, step)
        t = e.op(VX_OP_MUL, g0, g1)
        e.op(VX_OP_MUL, t, h, t)
        e.op(VX_OP_SUB, t, g0, t)
        e.op(VX_OP_SUB, t, g1, t)
        push(t, VX_AIR_ALL_ROWS)
        e.release(m1)
    gt = e.op(VX_OP_SUB, GAMMA, e.ldw(C.TBL))
    ht = e.ldw(AUX_HT)
    e.op(VX_OP_ADD, step, ht, step)
    t = e.op(VX_OP_MUL, ht, gt)
    push(e.op(VX_OP_SUB, t, e.ldw(C.MULT)), VX_AIR_ALL_ROWS)
    push(step, VX_AIR_TRANSITION)
    push(acc, VX_AIR_FIRST_ROW)
    push(acc, VX_AIR_LAST_ROW)
    e.release(m0)
    if bus:
        from . import VX_OP_LDP as _LDP
        # ---- the latch: FIRST marks a message's first block; LW holds its first 64 bytes as little-endian words over the whole message ----
        keep = e.top
        fr, frn = e.ldw(C.FIRST), e.ldw(C.FIRST, nxt=True)
        t = e.op(VX_OP_SUB, fr, ONE)
        push(e.op(VX_OP_MUL, t, fr), VX_AIR_ALL_ROWS)
        push(e.op(VX_OP_SUB, fr, ONE), VX_AIR_FIRST_ROW)
        t = e.op(VX_OP_SUB, NFr, fr)                         # FIRST' = FIRST + S81 (NF - FIRST)
        e.op(VX_OP_MUL, t, S81, t)
        e.op(VX_OP_ADD, t, fr, t)
        push(e.op(VX_OP_SUB, frn, t), VX_AIR_TRANSITION)
        ends = e.op(VX_OP_MUL, S81, NFr)                     # this row ends a message
        holds = e.op(VX_OP_SUB, ONE, ends)
        bind = e.op(VX_OP_MUL, e.ldw(C.SEL + 7), fr)         # row 7 of a first block: the window holds W_7 .. W_0
        keep2 = e.top
        for j in range(16):
            m1 = e.top
            lw, lwn = e.ldw(C.LW + j), e.ldw(C.LW + j, nxt=True)
            t = e.op(VX_OP_SUB, lwn, lw)
            push(e.op(VX_OP_MUL, t, holds), VX_AIR_TRANSITION)
            v = e.tmp()
            e.op(VX_OP_ADD, ZERO, ZERO, v)
            for col, wgt in le_word_of_bits(0, j):           # word m of the message = window entry 7 - m on row 7
                m2 = e.top
                mm, bit = col // 64, col % 64
                b = e.ldw(C.WB + 64 * (7 - mm) + bit)
                c = e.tmp()
                e.ldi(c, wgt)
                e.op(VX_OP_MUL, b, c, b)
                e.op(VX_OP_ADD, v, b, v)
                e.release(m2)
            t = e.op(VX_OP_SUB, lw, v)
            push(e.op(VX_OP_MUL, t, bind), VX_AIR_ALL_ROWS)
            e.release(m1)
        e.release(keep2)
        # ---- the send: (LW[0..16), digest as 16 little-endian words, TAG) on the row that ends a message ----
        beta, gbus = e.tmp(), e.tmp()
        e.ins(VX_OP_LDCH, beta, 1)
        e.ins(VX_OP_LDCH, gbus, 2)
        tup = e.tmp()
        e.ldi(tup, TAG_SHA512)
        for j in range(15, -1, -1):                          # the digest = the new chaining value, as bits in S on row 81
            m1 = e.top
            v = e.tmp()
            e.op(VX_OP_ADD, ZERO, ZERO, v)
            for col, wgt in le_word_of_bits(C.S, j):
                m2 = e.top
                b = e.ldw(col)
                c = e.tmp()
                e.ldi(c, wgt)
                e.op(VX_OP_MUL, b, c, b)
                e.op(VX_OP_ADD, v, b, v)
                e.release(m2)
            e.op(VX_OP_MUL, tup, beta, tup)
            e.op(VX_OP_ADD, tup, v, tup)
            e.release(m1)
        for j in range(15, -1, -1):
            m1 = e.top
            e.op(VX_OP_MUL, tup, beta, tup)
            e.op(VX_OP_ADD, tup, e.ldw(C.LW + j), tup)
            e.release(m1)
        dlt = e.op(VX_OP_SUB, gbus, tup)
        u, bacc, baccn = e.ldw(AUX_U), e.ldw(AUX_BUS), e.ldw(AUX_BUS, nxt=True)
        t = e.op(VX_OP_MUL, u, dlt)
        push(e.op(VX_OP_SUB, t, ends), VX_AIR_ALL_ROWS)      # u (gamma - tuple) = [this row ends a message]
        t = e.op(VX_OP_SUB, baccn, bacc)
        push(e.op(VX_OP_SUB, t, u), VX_AIR_TRANSITION)
        push(bacc, VX_AIR_FIRST_ROW)
        closing = e.tmp()
        e.ins(_LDP, closing, 16)                             # aux public input 0 (after the 16 public inputs): everything this table sent
        push(e.op(VX_OP_SUB, bacc, closing), VX_AIR_LAST_ROW)
        e.release(keep)
    e.ins(VX_OP_END)
    return e.w, npush


# ---------------------------------------------------------------------------------------------------------------------
def pad_message(msg: bytes) -> list:
    ml = len(msg)
    data = msg + b"\x80" + b"\x00" * ((111 - ml) % 128) + struct.pack(">QQ", 0, 8 * ml)
    return [list(struct.unpack(">16Q", data[i:i + 128])) for i in range(0, len(data), 128)]


def le_words(b: bytes) -> list:
    return [int.from_bytes(b[4 * j:4 * j + 4], "little") for j in range(len(b) // 4)]


def generate_trace(degree_bits: int, messages, bus=False) -> tuple:
    """-> (trace [1995 | 2012 bus][n] uint64, public inputs [16], digests of the messages completed inside the trace)"""
    C = Cols
    n = 1 << degree_bits
    blocks = []
    for m in messages:
        bl = pad_message(m)
        for j, b in enumerate(bl):
            blocks.append((b, j == len(bl) - 1))
    lim = lambda w: (w & 0xFFFFFFFF, w >> 32)   # noqa: E731
    words = np.zeros((24, n), dtype=np.uint64)      # 8 state + 16 window words per row
    aux = np.zeros((5, n), dtype=np.uint64)         # x0 x1 m y0 y1
    t = np.zeros((C.N_BUS if bus else C.N, n), dtype=np.uint64)
    state, H, D = list(IV), list(IV), [0] * 8
    first_of_message, latched = True, [0] * 16
    wwin = [0] * 16
    digests = []
    carries = np.zeros((6, n), dtype=np.int64)
    row = bi = 0
    while row < n:
        wordsb, last = blocks[bi] if bi < len(blocks) else ([0] * 16, False)
        real = bi < len(blocks)
        bi += 1
        W = list(wordsb) + [0] * 64
        if bus and first_of_message:                     # a new message: latch its first 64 bytes as little-endian 32-bit words
            latched = le_words(struct.pack(">8Q", *W[:8]))
        for r in range(PERIOD):
            if row >= n:
                break
            t[C.SEL + r, row] = 1
            if bus:
                t[C.LW:C.LW + 16, row] = latched
                t[C.FIRST, row] = 1 if first_of_message else 0
            for k in range(8):
                t[C.H + 2 * k, row], t[C.H + 2 * k + 1, row] = lim(H[k])
                t[C.D + 2 * k, row], t[C.D + 2 * k + 1, row] = lim(D[k])
            wwin = [W[r] if r < 80 else 0] + wwin[:15]
            for k in range(8):
                words[k, row] = state[k]
            for k in range(16):
                words[8 + k, row] = wwin[k]
            a, b_, c_, d_, e_, f_, g_, h_ = state
            w1, w14 = wwin[1], wwin[14]
            aux[0, row] = _rotr(a, 28) ^ _rotr(a, 34)
            aux[1, row] = _rotr(e_, 14) ^ _rotr(e_, 18)
            aux[2, row] = a & b_
            aux[3, row] = _rotr(w14, 1) ^ _rotr(w14, 8)
            aux[4, row] = _rotr(w1, 19) ^ _rotr(w1, 61)
            if r < 80:
                S1 = _rotr(e_, 14) ^ _rotr(e_, 18) ^ _rotr(e_, 41)
                chv = (e_ & f_) ^ (~e_ & g_ & M64)
                S0 = _rotr(a, 28) ^ _rotr(a, 34) ^ _rotr(a, 39)
                mj = (a & b_) ^ (a & c_) ^ (b_ & c_)
                t1 = [h_, S1, chv, K512[r], W[r]]
                for idx, terms in ((0, t1 + [S0, mj]), (2, t1 + [d_])):
                    lo = sum(x & 0xFFFFFFFF for x in terms)
                    carries[idx, row] = lo >> 32
                    hi = sum(x >> 32 for x in terms) + (lo >> 32)
                    carries[idx + 1, row] = hi >> 32
                new_a, new_e = sum(t1 + [S0, mj]) & M64, sum(t1 + [d_]) & M64
                state = [new_a, a, b_, c_, new_e, e_, f_, g_]
                if 15 <= r <= 78:
                    s1 = _rotr(w1, 19) ^ _rotr(w1, 61) ^ (w1 >> 6)
                    s0 = _rotr(w14, 1) ^ _rotr(w14, 8) ^ (w14 >> 7)
                    terms = [s1, wwin[6], s0, wwin[15]]
                    lo = sum(x & 0xFFFFFFFF for x in terms)
                    carries[4, row] = lo >> 32
                    carries[5, row] = (sum(x >> 32 for x in terms) + (lo >> 32)) >> 32
                    W[r + 1] = sum(terms) & M64
            elif r == 80:
                new = []
                for k in range(8):
                    lo = (H[k] & 0xFFFFFFFF) + (state[k] & 0xFFFFFFFF)
                    hi = (H[k] >> 32) + (state[k] >> 32) + (lo >> 32)
                    t[C.FFC + 2 * k, row], t[C.FFC + 2 * k + 1, row] = lo >> 32, hi >> 32
                    new.append((H[k] + state[k]) & M64)
                state = new
            else:
                nf = 1 if (last and real) else 0
                t[C.NF, row] = nf
                if nf:
                    D = list(state)
                    digests.append(b"".join(struct.pack(">Q", x) for x in state))
                    state = list(IV)
                H = list(state)
            row += 1
        first_of_message = bool(last and real)           # the next block starts a message iff this one ended one
    for k in range(24):
        base = C.S + 64 * k if k < 8 else C.WB + 64 * (k - 8)
        for i in range(64):
            t[base + i] = (words[k] >> np.uint64(i)) & np.uint64(1)
    for q, base in enumerate((C.X0, C.X1, C.M, C.Y0, C.Y1)):
        for i in range(64):
            t[base + i] = (aux[q] >> np.uint64(i)) & np.uint64(1)
    for q, col in enumerate((C.CA, C.CA + 1, C.CE, C.CE + 1, C.CW, C.CW + 1)):
        t[col] = carries[q].astype(np.uint64)
    t[C.TBL] = np.arange(n, dtype=np.uint64) % 8
    counts = np.bincount(carries[:, :n - 1].reshape(-1), minlength=8)
    assert counts.size == 8, "a carry left [0, 8)"
    t[C.MULT, :8] = counts.astype(np.uint64)
    pis = t[C.D:C.D + 16, n - 1].copy()
    return t, pis, digests


def aux_columns(trace, chal):
    """second-round columns [h_a, h_e, h_w, ht, acc]"""
    C = Cols
    n = trace.shape[1]
    g = int(chal[0])
    inv = np.array([pow((g - v) % P, P - 2, P) for v in range(8)], dtype=object)
    look = lambda col: inv[trace[col].astype(np.int64)]   # noqa: E731
    h = [(look(b) + look(b + 1)) % P for b in (C.CA, C.CE, C.CW)]
    ht = (trace[C.MULT].astype(object) * look(C.TBL)) % P
    step = (h[0] + h[1] + h[2] - ht) % P
    acc = np.zeros(n, dtype=object)
    run = 0
    for i in range(n):
        acc[i] = run
        run = (run + int(step[i])) % P
    return np.stack([np.array(c, dtype=np.uint64) for c in h + [ht, acc]])


def bus_tuples(trace):
    """-> (rows that send, [rows][33] elements: the message's first 64 bytes and its digest as little-endian 32-bit words, TAG_SHA512)"""
    C = Cols
    rows = np.nonzero((trace[C.SEL + 81] == 1) & (trace[C.NF] == 1))[0]
    out = np.zeros((rows.size, 33), dtype=np.uint64)
    for j in range(16):
        out[:, j] = trace[C.LW + j, rows]
        acc = np.zeros(rows.size, dtype=np.uint64)
        for col, wgt in le_word_of_bits(C.S, j):
            acc += trace[col, rows] * np.uint64(wgt)
        out[:, 16 + j] = acc
    out[:, 32] = TAG_SHA512
    return rows, out


def aux_columns_bus(trace, chal):
    """second-round columns of the bus variant: [h_a, h_e, h_w, ht, acc, bus_u, bus_acc] and the closing sum of the sends"""
    from . import hostfield as hf
    from .eddsa_air import _horner
    n = trace.shape[1]
    base = aux_columns(trace, chal[:1])
    beta, g = int(chal[1]), int(chal[2])
    rows, tuples = bus_tuples(trace)
    u = np.zeros(n, dtype=np.uint64)
    if rows.size:
        u[rows] = hf.invmod(hf.submod(np.full(rows.size, g, dtype=np.uint64), _horner(tuples, beta)))
    acc, _ = hf.exclusive_prefix_sum(u)
    return np.concatenate([base, np.stack([u, acc])]), np.array([int(acc[n - 1])], dtype=np.uint64)


def aux_program():
    """the GPU form of `aux_columns` (vx_stark_aux_columns): three pair helpers and the table term as fractions, one running sum"""
    from . import AuxProgram
    C = Cols
    e = _Emit(scratch=40)
    GAMMA = 63
    e.ins(VX_OP_LDCH, GAMMA, 0)
    for base in (C.CA, C.CE, C.CW):
        m0 = e.top
        g0, g1 = e.op(VX_OP_SUB, GAMMA, e.ldw(base)), e.op(VX_OP_SUB, GAMMA, e.ldw(base + 1))
        e.push(e.op(VX_OP_ADD, g0, g1), 0)
        e.push(e.op(VX_OP_MUL, g0, g1), 0)
        e.release(m0)
    e.push(e.ldw(C.MULT), 0)
    e.push(e.op(VX_OP_SUB, GAMMA, e.ldw(C.TBL)), 0)
    e.ins(VX_OP_END)
    return AuxProgram(C.N, 1, e.w, 4, [[1, 1, 1, -1]])


def aux_program_bus():
    """the GPU form of `aux_columns_bus`: the four range-check fractions, the bus fraction [this row ends a message] / (gamma_bus - tuple),
    the two running sums; the bus sum's closing value is the set's aux public input"""
    from . import VX_OP_MUL as _MUL
    from . import AuxProgram
    C = Cols
    e = _Emit(scratch=40)
    GAMMA, BETA, GBUS, ZERO = 63, 62, 61, 60
    e.ldi(ZERO, 0)
    e.ins(VX_OP_LDCH, GAMMA, 0)
    e.ins(VX_OP_LDCH, BETA, 1)
    e.ins(VX_OP_LDCH, GBUS, 2)
    for base in (C.CA, C.CE, C.CW):
        m0 = e.top
        g0, g1 = e.op(VX_OP_SUB, GAMMA, e.ldw(base)), e.op(VX_OP_SUB, GAMMA, e.ldw(base + 1))
        e.push(e.op(VX_OP_ADD, g0, g1), 0)
        e.push(e.op(_MUL, g0, g1), 0)
        e.release(m0)
    e.push(e.ldw(C.MULT), 0)
    e.push(e.op(VX_OP_SUB, GAMMA, e.ldw(C.TBL)), 0)
    tup = e.tmp()
    e.ldi(tup, TAG_SHA512)
    for j in range(15, -1, -1):
        m1 = e.top
        v = e.tmp()
        e.op(VX_OP_ADD, ZERO, ZERO, v)
        for col, wgt in le_word_of_bits(C.S, j):
            m2 = e.top
            b = e.ldw(col)
            c = e.tmp()
            e.ldi(c, wgt)
            e.op(_MUL, b, c, b)
            e.op(VX_OP_ADD, v, b, v)
            e.release(m2)
        e.op(_MUL, tup, BETA, tup)
        e.op(VX_OP_ADD, tup, v, tup)
        e.release(m1)
    for j in range(15, -1, -1):
        m1 = e.top
        e.op(_MUL, tup, BETA, tup)
        e.op(VX_OP_ADD, tup, e.ldw(C.LW + j), tup)
        e.release(m1)
    e.push(e.op(_MUL, e.ldw(C.SEL + 81), e.ldw(C.NF)), 0)
    e.push(e.op(VX_OP_SUB, GBUS, tup), 0)
    e.ins(VX_OP_END)
    return AuxProgram(C.N_BUS, 3, e.w, 5, [[1, 1, 1, -1, 0], [0, 0, 0, 0, 1]], fraction_out=[0, 1, 2, 3, 5], sum_out=[4, 6], api_sums=(1,))


def make_stark(degree_bits: int, bus=False, **cfg) -> Stark:
    cfg.setdefault("rate_bits", 1)
    if bus:
        prog, _ = build_program(True)
        st = Stark(degree_bits, Cols.N_BUS, 16, prog, constraint_degree=3, num_aux_columns=Cols.NAUX_BUS, num_aux_challenges=3, aux_fn=aux_columns_bus,
                   num_aux_public_inputs=1, **cfg)
        st.aux_program = aux_program_bus()
        return st
    prog, _ = build_program()
    st = Stark(degree_bits, Cols.N, 16, prog, constraint_degree=3, num_aux_columns=Cols.NAUX, num_aux_challenges=1, aux_fn=aux_columns, **cfg)
    st.aux_program = aux_program()
    return st


def reference_digests(messages):
    return [hashlib.sha512(m).digest() for m in messages]
