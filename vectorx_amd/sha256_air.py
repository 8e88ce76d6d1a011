"""A SHA-256 AIR at chip density — caller-side stand-in for the hash chips that Curta (starkyx v1.0.0,
/root/reference/Cargo.lock:7232-7249) puts under every VectorX map / outer proof (`curta_sha256`:
/root/reference/circuits/builder/justification.rs:140-156; the BLAKE2b chip at circuits/builder/header.rs:18 has the same
shape).  OWN AIR, NOT CURTA'S: the starkyx sources are not in the reference tree, so this is a bit-decomposed SHA-256 of
this repository's own design, written to load `vx_stark_begin / vx_stark_finish` the way a real chip does — a thousand
columns, two thousand constraints of degree <= 3, a log-derivative range check in the second commitment round — and to be
checkable against `hashlib`.  Plain host code: it emits a constraint program (include/vxprover.h VX_OP_*), generates the
trace and the second-round columns; no GPU, no oracle.

Layout.  A block of the (already padded) message occupies 66 consecutive rows, marked by a cyclic one-hot `s`:
  rows 0..63   the 64 rounds: the row holds the working state BEFORE round t as bits — S[0..3] = a, b, c, d and
               S[4..7] = e, f, g, h — the message-schedule window Wb[k] = W_{t-k} (k < 16) as bits, and H = the chaining
               value of the block; the round writes a', e' into the NEXT row and shifts the windows;
  row 64       the state after round 63; the transition adds the chaining value word by word (carries `ffc`);
  row 65       the new chaining value as bits; the transition starts the next block from it, or from the IV when the
               flag `nf` ("next block starts a new message") is set, in which case the value is latched into D.
Every 3-input XOR is split through one auxiliary bit (x0, x1 for Sigma0 / Sigma1, y0, y1 for sigma0 / sigma1) and Maj through
m = a b, so that every gadget has degree 2 and the row-type selectors can multiply it: constraint degree 3.
The three additions' carries ca, ce, cw are VALUES in [0, 8), range-checked by a log-derivative lookup into the column
`tbl` (0..7 repeating, multiplicities `mult`) with three second-round columns (two partial-sum helpers and the running sum).
Public inputs: the 8 words of D in the last row = the digest of the last message completed inside the trace.
"""
from __future__ import annotations

import hashlib
import struct

import numpy as np

from . import (VX_AIR_ALL_ROWS, VX_AIR_FIRST_ROW, VX_AIR_LAST_ROW, VX_AIR_TRANSITION, VX_OP_ADD, VX_OP_END, VX_OP_LDCH, VX_OP_LDI, VX_OP_LDN,
               VX_OP_LDP, VX_OP_LDW, VX_OP_MUL, VX_OP_PUSH, VX_OP_SUB, Stark, vx_ins)

P = 0xFFFFFFFF00000001
PERIOD = 66

K256 = [0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
        0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
        0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
        0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
        0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
        0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
IV = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]


class Cols:
    """column map (1024 trace columns + 3 second-round columns)"""
    S = 0                     # S + 32 k + i: bit i of state word k (a b c d e f g h)
    WB = 256                  # WB + 32 k + i: bit i of W_{t-k}
    X0 = 768                  # a_{i+2} ^ a_{i+13}
    X1 = 800                  # e_{i+6} ^ e_{i+11}
    M = 832                   # a_i b_i
    Y0 = 864                  # w14_{i+7} ^ w14_{i+18}      (w14 = Wb[14] = W_{t-14}: sigma0's argument for the NEXT row's W)
    Y1 = 896                  # w1_{i+17} ^ w1_{i+19}
    SEL = 928                 # one-hot row type, 66 columns
    H = 994                   # chaining value, 8 words
    D = 1002                  # last completed message digest, 8 words
    FFC = 1010                # feed-forward carries, 8 bits (row 64)
    NF = 1018                 # next block starts a new message (row 65)
    CA, CE, CW = 1019, 1020, 1021
    TBL, MULT = 1022, 1023
    N = 1024
    AUX_H1, AUX_H2, AUX_ACC = 1024, 1025, 1026   # second round (program columns >= N)
    AUX_BUS_U, AUX_BUS_ACC = 1027, 1028          # bus variant only: send / (gamma - tuple) and its running sum


def _rotr(x, r):
    return ((x >> r) | (x << (32 - r))) & 0xFFFFFFFF


class _Emit:
    """straight-line program emitter with a trivial register discipline: r0..r47 scratch (bump-allocated per constraint),
    r48.. persistent"""

    def __init__(self, scratch=48):
        self.w = []
        self.top = 0
        self.scratch = scratch

    def ins(self, op, dst=0, a=0, b=0):
        self.w.append(vx_ins(op, dst, a, b))

    def tmp(self):
        r = self.top
        self.top += 1
        assert r < self.scratch, "scratch registers exhausted"
        return r

    def release(self, mark):
        self.top = mark

    def ldw(self, col, nxt=False, dst=None):
        r = self.tmp() if dst is None else dst
        self.ins(VX_OP_LDN if nxt else VX_OP_LDW, r, col)
        return r

    def ldi(self, dst, v):
        self.ins(VX_OP_LDI, dst)
        self.w.append(v % P)

    def op(self, op, a, b, dst=None):
        r = self.tmp() if dst is None else dst
        self.ins(op, r, a, b)
        return r

    def push(self, r, kind):
        self.ins(VX_OP_PUSH, 0, r, kind)


def build_program(bus=False):
    """-> (program words, number of constraints).  bus=True: the table also SENDS every completed digest on a bus shared with
    other tables (vectorx_amd/stark_bus.py): aux challenges [gamma_range, beta, gamma_bus], two more second-round columns, and
    the closing sum of its sends as aux public input 0 (LDP index 8)."""
    C = Cols
    e = _Emit()
    ONE, TWO32, GAMMA, IS_ROUND, S64, S65, SCHED, NFr, KREG, ZERO = 63, 62, 61, 60, 59, 58, 57, 56, 55, 54
    e.ldi(ONE, 1)
    e.ldi(TWO32, 1 << 32)
    e.ldi(ZERO, 0)
    e.ins(VX_OP_LDCH, GAMMA, 0)
    npush = 0

    def push(r, kind):
        nonlocal npush
        e.push(r, kind)
        npush += 1

    # ---- row-type selectors: is_round = 1 - s64 - s65; sched = sum_{i=15..62} s_i; K = sum_i s_i K_i ----
    e.ldw(C.SEL + 64, dst=S64)
    e.ldw(C.SEL + 65, dst=S65)
    e.op(VX_OP_SUB, ONE, S64, IS_ROUND)
    e.op(VX_OP_SUB, IS_ROUND, S65, IS_ROUND)
    e.ldw(C.NF, dst=NFr)
    mark = e.top
    first = True
    for i in range(15, 63):
        r = e.ldw(C.SEL + i)
        if first:
            e.op(VX_OP_ADD, r, ZERO, SCHED)
            first = False
        else:
            e.op(VX_OP_ADD, SCHED, r, SCHED)
        e.release(mark)
    first = True
    for i in range(64):
        r = e.ldw(C.SEL + i)
        k = e.tmp()
        e.ldi(k, K256[i])
        e.op(VX_OP_MUL, r, k, r)
        if first:
            e.op(VX_OP_ADD, r, ZERO, KREG)
            first = False
        else:
            e.op(VX_OP_ADD, KREG, r, KREG)
        e.release(mark)

    def boolean(col):
        m0 = e.top
        r = e.ldw(col)
        t = e.op(VX_OP_SUB, r, ONE)
        e.op(VX_OP_MUL, t, r, t)
        push(t, VX_AIR_ALL_ROWS)
        e.release(m0)

    # ---- booleanity: state bits, schedule window bits, feed-forward carries, nf ----
    for c in range(C.S, C.S + 256):
        boolean(c)
    for c in range(C.WB, C.WB + 512):
        boolean(c)
    for c in range(C.FFC, C.FFC + 8):
        boolean(c)
    boolean(C.NF)

    def xor_into(dst, x, y):
        """dst = x + y - 2 x y (registers)"""
        m0 = e.top
        t = e.op(VX_OP_MUL, x, y)
        t2 = e.op(VX_OP_ADD, t, t)
        s = e.op(VX_OP_ADD, x, y)
        e.op(VX_OP_SUB, s, t2, dst)
        e.release(m0)

    def define_xor(col_out, col_x, col_y):
        """constraint: out = x ^ y on boolean columns (all rows, degree 2)"""
        m0 = e.top
        x, y, o = e.ldw(col_x), e.ldw(col_y), e.ldw(col_out)
        t = e.tmp()
        xor_into(t, x, y)
        e.op(VX_OP_SUB, o, t, t)
        push(t, VX_AIR_ALL_ROWS)
        e.release(m0)

    A, B, Cc, Dd, E, F, G, Hh = (C.S + 32 * k for k in range(8))
    for i in range(32):
        define_xor(C.X0 + i, A + (i + 2) % 32, A + (i + 13) % 32)
        define_xor(C.X1 + i, E + (i + 6) % 32, E + (i + 11) % 32)
        define_xor(C.Y0 + i, C.WB + 32 * 14 + (i + 7) % 32, C.WB + 32 * 14 + (i + 18) % 32)
        define_xor(C.Y1 + i, C.WB + 32 * 1 + (i + 17) % 32, C.WB + 32 * 1 + (i + 19) % 32)
        m0 = e.top
        a, b, m = e.ldw(A + i), e.ldw(B + i), e.ldw(C.M + i)
        t = e.op(VX_OP_MUL, a, b)
        e.op(VX_OP_SUB, m, t, t)
        push(t, VX_AIR_ALL_ROWS)
        e.release(m0)

    def word(base, nxt=False, dst=None):
        """sum_i 2^i col[base + i] by Horner from the top bit"""
        acc = e.tmp() if dst is None else dst
        m0 = e.top
        e.ldw(base + 31, nxt, dst=acc)
        for i in range(30, -1, -1):
            e.op(VX_OP_ADD, acc, acc, acc)
            b = e.ldw(base + i, nxt)
            e.op(VX_OP_ADD, acc, b, acc)
            e.release(m0)
        return acc

    def word_of(bit_expr, dst=None):
        """sum_i 2^i bit_expr(i) where bit_expr(i, dst_reg) leaves a degree-<=2 bit value in dst_reg"""
        acc = e.tmp() if dst is None else dst
        m0 = e.top
        bit_expr(31, acc)
        for i in range(30, -1, -1):
            e.op(VX_OP_ADD, acc, acc, acc)
            t = e.tmp()
            bit_expr(i, t)
            e.op(VX_OP_ADD, acc, t, acc)
            e.release(m0)
        return acc

    def big_sigma0(i, dst):
        x, z = e.ldw(C.X0 + i), e.ldw(A + (i + 22) % 32)
        xor_into(dst, x, z)

    def big_sigma1(i, dst):
        x, z = e.ldw(C.X1 + i), e.ldw(E + (i + 25) % 32)
        xor_into(dst, x, z)

    def maj(i, dst):       # m + c (a + b - 2 m)
        a, b, c, m = e.ldw(A + i), e.ldw(B + i), e.ldw(Cc + i), e.ldw(C.M + i)
        t = e.op(VX_OP_ADD, a, b)
        t = e.op(VX_OP_SUB, t, m, t)
        t = e.op(VX_OP_SUB, t, m, t)
        t = e.op(VX_OP_MUL, t, c, t)
        e.op(VX_OP_ADD, t, m, dst)

    def ch(i, dst):        # g + e (f - g)
        ee, f, g = e.ldw(E + i), e.ldw(F + i), e.ldw(G + i)
        t = e.op(VX_OP_SUB, f, g)
        t = e.op(VX_OP_MUL, t, ee, t)
        e.op(VX_OP_ADD, t, g, dst)

    def small_sigma0(i, dst):
        y = e.ldw(C.Y0 + i)
        if i + 3 < 32:
            xor_into(dst, y, e.ldw(C.WB + 32 * 14 + i + 3))
        else:
            e.op(VX_OP_ADD, y, ZERO, dst)

    def small_sigma1(i, dst):
        y = e.ldw(C.Y1 + i)
        if i + 10 < 32:
            xor_into(dst, y, e.ldw(C.WB + 32 * 1 + i + 10))
        else:
            e.op(VX_OP_ADD, y, ZERO, dst)

    # ---- the round (rows 0..63 -> next row): T1 = h + Sigma1(e) + Ch + K + W,  a' = T1 + Sigma0(a) + Maj,  e' = d + T1 ----
    m0 = e.top
    t1 = word(Hh)
    r = word_of(big_sigma1)
    e.op(VX_OP_ADD, t1, r, t1)
    e.release(r)
    r = word_of(ch)
    e.op(VX_OP_ADD, t1, r, t1)
    e.release(r)
    e.op(VX_OP_ADD, t1, KREG, t1)
    r = word(C.WB)
    e.op(VX_OP_ADD, t1, r, t1)
    e.release(r)
    t2 = word_of(big_sigma0)
    r = word_of(maj)
    e.op(VX_OP_ADD, t2, r, t2)
    e.release(r)
    lhs = word(A, nxt=True)                       # a' + 2^32 ca - (T1 + T2)
    ca = e.ldw(C.CA)
    e.op(VX_OP_MUL, ca, TWO32, ca)
    e.op(VX_OP_ADD, lhs, ca, lhs)
    e.op(VX_OP_SUB, lhs, t1, lhs)
    e.op(VX_OP_SUB, lhs, t2, lhs)
    e.op(VX_OP_MUL, lhs, IS_ROUND, lhs)
    push(lhs, VX_AIR_TRANSITION)
    e.release(lhs)
    lhs = word(E, nxt=True)                       # e' + 2^32 ce - (d + T1)
    ce = e.ldw(C.CE)
    e.op(VX_OP_MUL, ce, TWO32, ce)
    e.op(VX_OP_ADD, lhs, ce, lhs)
    e.op(VX_OP_SUB, lhs, t1, lhs)
    d = word(Dd)
    e.op(VX_OP_SUB, lhs, d, lhs)
    e.op(VX_OP_MUL, lhs, IS_ROUND, lhs)
    push(lhs, VX_AIR_TRANSITION)
    e.release(m0)
    # state window shifts under is_round: b' = a, c' = b, d' = c, f' = e, g' = f, h' = g
    for k in (1, 2, 3, 5, 6, 7):
        for i in range(32):
            m0 = e.top
            cur, nx = e.ldw(C.S + 32 * (k - 1) + i), e.ldw(C.S + 32 * k + i, nxt=True)
            t = e.op(VX_OP_SUB, nx, cur)
            e.op(VX_OP_MUL, t, IS_ROUND, t)
            push(t, VX_AIR_TRANSITION)
            e.release(m0)
    # ---- message schedule: the window always shifts; W' = sigma1(W_{t-1}) + W_{t-6} + sigma0(W_{t-14}) + W_{t-15} when the
    #      next row is a round >= 16 ----
    for k in range(1, 16):
        for i in range(32):
            m0 = e.top
            cur, nx = e.ldw(C.WB + 32 * (k - 1) + i), e.ldw(C.WB + 32 * k + i, nxt=True)
            t = e.op(VX_OP_SUB, nx, cur)
            push(t, VX_AIR_TRANSITION)
            e.release(m0)
    m0 = e.top
    lhs = word(C.WB, nxt=True)
    cw = e.ldw(C.CW)
    e.op(VX_OP_MUL, cw, TWO32, cw)
    e.op(VX_OP_ADD, lhs, cw, lhs)
    for r in (word_of(small_sigma1), word(C.WB + 32 * 6), word_of(small_sigma0), word(C.WB + 32 * 15)):
        e.op(VX_OP_SUB, lhs, r, lhs)
    e.op(VX_OP_MUL, lhs, SCHED, lhs)
    push(lhs, VX_AIR_TRANSITION)
    e.release(m0)
    # ---- row type: cyclic shift of the one-hot; first row = (1, 0, .., 0) ----
    for i in range(PERIOD):
        m0 = e.top
        cur, nx = e.ldw(C.SEL + (i - 1) % PERIOD), e.ldw(C.SEL + i, nxt=True)
        t = e.op(VX_OP_SUB, nx, cur)
        push(t, VX_AIR_TRANSITION)
        e.release(m0)
        m0 = e.top
        r = e.ldw(C.SEL + i)
        if i == 0:
            r = e.op(VX_OP_SUB, r, ONE)
        push(r, VX_AIR_FIRST_ROW)
        e.release(m0)
    # ---- chaining value: constant inside a block (rows 0..64), the feed-forward at row 64, the hand-over at row 65 ----
    not65 = e.op(VX_OP_SUB, ONE, S65)
    for k in range(8):
        m0 = e.top
        h, hn = e.ldw(C.H + k), e.ldw(C.H + k, nxt=True)
        t = e.op(VX_OP_SUB, hn, h)
        e.op(VX_OP_MUL, t, not65, t)
        push(t, VX_AIR_TRANSITION)                                   # H' = H unless the row is 65
        e.release(m0)
        m0 = e.top
        sn = word(C.S + 32 * k, nxt=True)                            # row 64: S'_k + 2^32 c_k = H_k + S_k
        c = e.ldw(C.FFC + k)
        e.op(VX_OP_MUL, c, TWO32, c)
        e.op(VX_OP_ADD, sn, c, sn)
        e.op(VX_OP_SUB, sn, e.ldw(C.H + k), sn)
        e.op(VX_OP_SUB, sn, word(C.S + 32 * k), sn)
        e.op(VX_OP_MUL, sn, S64, sn)
        push(sn, VX_AIR_TRANSITION)
        e.release(m0)
        m0 = e.top
        # row 65: H'_k = nf IV_k + (1 - nf) S_k ;  D'_k = D_k + s65 nf (S_k - D_k)
        sk = word(C.S + 32 * k)
        iv = e.tmp()
        e.ldi(iv, IV[k])
        t = e.op(VX_OP_SUB, iv, sk)
        e.op(VX_OP_MUL, t, NFr, t)
        e.op(VX_OP_ADD, t, sk, t)                                    # nf IV + (1 - nf) S
        hn = e.ldw(C.H + k, nxt=True)
        e.op(VX_OP_SUB, hn, t, t)
        e.op(VX_OP_MUL, t, S65, t)
        push(t, VX_AIR_TRANSITION)
        dk, dn = e.ldw(C.D + k), e.ldw(C.D + k, nxt=True)
        u = e.op(VX_OP_SUB, sk, dk)
        e.op(VX_OP_MUL, u, NFr, u)
        e.op(VX_OP_MUL, u, S65, u)
        e.op(VX_OP_ADD, u, dk, u)
        e.op(VX_OP_SUB, dn, u, u)
        push(u, VX_AIR_TRANSITION)
        e.release(m0)
        # first row: the chaining value and the state are the IV, D = 0;  last row: D = the public digest
        m0 = e.top
        iv = e.tmp()
        e.ldi(iv, IV[k])
        t = e.op(VX_OP_SUB, e.ldw(C.H + k), iv)
        push(t, VX_AIR_FIRST_ROW)
        t = e.op(VX_OP_SUB, word(C.S + 32 * k), iv)
        push(t, VX_AIR_FIRST_ROW)
        push(e.ldw(C.D + k), VX_AIR_FIRST_ROW)
        pi = e.tmp()
        e.ins(VX_OP_LDP, pi, k)
        t = e.op(VX_OP_SUB, e.ldw(C.D + k), pi)
        push(t, VX_AIR_LAST_ROW)
        e.release(m0)
    # row 65 -> next block: S' = S + nf (IV - S) bit by bit
    for k in range(8):
        for i in range(32):
            m0 = e.top
            s, sn = e.ldw(C.S + 32 * k + i), e.ldw(C.S + 32 * k + i, nxt=True)
            if (IV[k] >> i) & 1:
                t = e.op(VX_OP_SUB, ONE, s)
            else:
                t = e.op(VX_OP_SUB, ZERO, s)
            e.op(VX_OP_MUL, t, NFr, t)
            e.op(VX_OP_ADD, t, s, t)
            e.op(VX_OP_SUB, sn, t, t)
            e.op(VX_OP_MUL, t, S65, t)
            push(t, VX_AIR_TRANSITION)
            e.release(m0)
    # ---- range check of the carries: log-derivative lookup into tbl = 0, 1, .., 7, 0, 1, .. ----
    m0 = e.top
    tb, tbn = e.ldw(C.TBL), e.ldw(C.TBL, nxt=True)
    inc = e.op(VX_OP_SUB, tbn, tb)
    e.op(VX_OP_SUB, inc, ONE, inc)                                   # tbl' - tbl - 1
    t = e.op(VX_OP_MUL, inc, tbn)
    push(t, VX_AIR_TRANSITION)                                       # no increment => tbl' = 0 ...
    seven = e.tmp()
    e.ldi(seven, 7)
    t = e.op(VX_OP_SUB, tb, seven)
    e.op(VX_OP_MUL, t, inc, t)
    push(t, VX_AIR_TRANSITION)                                       # ... and tbl = 7
    push(tb, VX_AIR_FIRST_ROW)
    ga, ge, gw, gt = (e.op(VX_OP_SUB, GAMMA, e.ldw(c)) for c in (C.CA, C.CE, C.CW, C.TBL))
    h1, h2, acc, accn, mult = e.ldw(C.AUX_H1), e.ldw(C.AUX_H2), e.ldw(C.AUX_ACC), e.ldw(C.AUX_ACC, nxt=True), e.ldw(C.MULT)
    t = e.op(VX_OP_MUL, ga, ge)
    e.op(VX_OP_MUL, t, h1, t)
    e.op(VX_OP_SUB, t, ga, t)
    e.op(VX_OP_SUB, t, ge, t)
    push(t, VX_AIR_ALL_ROWS)                                         # h1 (g - ca)(g - ce) = (g - ca) + (g - ce)
    t = e.op(VX_OP_MUL, gw, gt)
    e.op(VX_OP_MUL, t, h2, t)
    e.op(VX_OP_SUB, t, gt, t)
    u = e.op(VX_OP_MUL, mult, gw)
    e.op(VX_OP_ADD, t, u, t)
    push(t, VX_AIR_ALL_ROWS)                                         # h2 (g - cw)(g - tbl) = (g - tbl) - mult (g - cw)
    t = e.op(VX_OP_SUB, accn, acc)
    e.op(VX_OP_SUB, t, h1, t)
    e.op(VX_OP_SUB, t, h2, t)
    push(t, VX_AIR_TRANSITION)
    push(acc, VX_AIR_FIRST_ROW)
    push(acc, VX_AIR_LAST_ROW)
    e.release(m0)
    if bus:
        # ---- bus: at a hand-over row that ends a message (s65 nf = 1) the digest words go out as one tuple
        #      t = sum_k beta^k S_k;  u (gamma - t) = s65 nf;  acc' = acc + u;  acc_0 = 0, acc_last = the closing sum ----
        m0 = e.top
        beta, gbus = e.tmp(), e.tmp()
        e.ins(VX_OP_LDCH, beta, 1)
        e.ins(VX_OP_LDCH, gbus, 2)
        t = word(C.S + 32 * 7)
        for k in range(6, -1, -1):
            e.op(VX_OP_MUL, t, beta, t)
            wk = word(C.S + 32 * k)
            e.op(VX_OP_ADD, t, wk, t)
            e.release(wk)
        d = e.op(VX_OP_SUB, gbus, t)
        u, acc, accn = e.ldw(C.AUX_BUS_U), e.ldw(C.AUX_BUS_ACC), e.ldw(C.AUX_BUS_ACC, nxt=True)
        send = e.op(VX_OP_MUL, S65, NFr)
        r = e.op(VX_OP_MUL, u, d)
        e.op(VX_OP_SUB, r, send, r)
        push(r, VX_AIR_ALL_ROWS)
        r = e.op(VX_OP_SUB, accn, acc)
        e.op(VX_OP_SUB, r, u, r)
        push(r, VX_AIR_TRANSITION)
        push(acc, VX_AIR_FIRST_ROW)
        closing = e.tmp()
        e.ins(VX_OP_LDP, closing, 8)
        r = e.op(VX_OP_SUB, acc, closing)
        push(r, VX_AIR_LAST_ROW)
        e.release(m0)
    e.ins(VX_OP_END)
    return e.w, npush


# ---------------------------------------------------------------------------------------------------------------------
def pad_message(msg: bytes) -> list:
    """SHA-256 padding -> list of 16-word blocks"""
    ml = len(msg)
    data = msg + b"\x80" + b"\x00" * ((55 - ml) % 64) + struct.pack(">Q", 8 * ml)
    return [list(struct.unpack(">16I", data[i:i + 64])) for i in range(0, len(data), 64)]


def _bits(v):
    return [(v >> i) & 1 for i in range(32)]


def generate_trace(degree_bits: int, messages) -> tuple:
    """-> (trace [1024][n] uint64, public inputs [8], digests of the messages completed inside the trace).
    `messages`: byte strings hashed one after the other; rows that remain after the last message keep hashing blocks of an
    endless zero-message so that every row is a valid round.  The LAST COMPLETED message's digest is the public input."""
    C = Cols
    n = 1 << degree_bits
    t = np.zeros((C.N, n), dtype=np.uint64)
    blocks = []                                   # (16 words, starts a new message AFTER this block)
    for m in messages:
        bl = pad_message(m)
        for j, b in enumerate(bl):
            blocks.append((b, j == len(bl) - 1))
    row = 0
    state = list(IV)
    H = list(IV)
    D = [0] * 8
    wwin = [0] * 16                               # W_{t-k}
    digests = []
    bi = 0
    carries = np.zeros((3, n), dtype=np.int64)
    while row < n:
        words, last = blocks[bi] if bi < len(blocks) else ([0] * 16, False)
        real = bi < len(blocks)
        bi += 1
        W = list(words) + [0] * 48
        for r in range(PERIOD):
            if row >= n:
                break
            t[C.SEL + r, row] = 1
            for k in range(8):
                t[C.H + k, row] = H[k]
                t[C.D + k, row] = D[k]
            if r < 64:
                if r >= 16:
                    pass                           # W[r] was set by the previous row's schedule step
                wwin = [W[r]] + wwin[:15]
            else:
                wwin = [0] + wwin[:15]
            for k in range(16):
                for i, b in enumerate(_bits(wwin[k])):
                    t[C.WB + 32 * k + i, row] = b
            for k in range(8):
                for i, b in enumerate(_bits(state[k])):
                    t[C.S + 32 * k + i, row] = b
            a, b_, c_, d_, e_, f_, g_, h_ = state
            w1, w14 = wwin[1], wwin[14]
            for i in range(32):
                t[C.X0 + i, row] = ((a >> ((i + 2) % 32)) ^ (a >> ((i + 13) % 32))) & 1
                t[C.X1 + i, row] = ((e_ >> ((i + 6) % 32)) ^ (e_ >> ((i + 11) % 32))) & 1
                t[C.M + i, row] = (a >> i) & (b_ >> i) & 1
                t[C.Y0 + i, row] = ((w14 >> ((i + 7) % 32)) ^ (w14 >> ((i + 18) % 32))) & 1
                t[C.Y1 + i, row] = ((w1 >> ((i + 17) % 32)) ^ (w1 >> ((i + 19) % 32))) & 1
            if r < 64:
                S1 = _rotr(e_, 6) ^ _rotr(e_, 11) ^ _rotr(e_, 25)
                chv = (e_ & f_) ^ (~e_ & g_ & 0xFFFFFFFF)
                T1 = h_ + S1 + chv + K256[r] + W[r]
                S0 = _rotr(a, 2) ^ _rotr(a, 13) ^ _rotr(a, 22)
                mj = (a & b_) ^ (a & c_) ^ (b_ & c_)
                full_a, full_e = T1 + S0 + mj, d_ + T1
                carries[0, row], carries[1, row] = full_a >> 32, full_e >> 32
                state = [full_a & 0xFFFFFFFF, a, b_, c_, full_e & 0xFFFFFFFF, e_, f_, g_]
                if 15 <= r <= 62:                 # the next row is a round >= 16: its W
                    s1 = _rotr(w1, 17) ^ _rotr(w1, 19) ^ (w1 >> 10)
                    s0 = _rotr(w14, 7) ^ _rotr(w14, 18) ^ (w14 >> 3)
                    full = s1 + wwin[6] + s0 + wwin[15]
                    carries[2, row] = full >> 32
                    W[r + 1] = full & 0xFFFFFFFF
            elif r == 64:
                new = []
                for k in range(8):
                    full = H[k] + state[k]
                    t[C.FFC + k, row] = full >> 32
                    new.append(full & 0xFFFFFFFF)
                state = new
            else:                                   # r == 65: hand-over
                nf = 1 if (last or not real) else 0
                if not real:
                    nf = 0                          # the endless zero-message after the last real one never ends
                t[C.NF, row] = nf
                if nf:
                    D = list(state)
                    digests.append(b"".join(struct.pack(">I", x) for x in state))
                    state = list(IV)
                H = list(state)
            row += 1
    t[C.CA], t[C.CE], t[C.CW] = carries[0].astype(np.uint64), carries[1].astype(np.uint64), carries[2].astype(np.uint64)
    t[C.TBL] = np.arange(n, dtype=np.uint64) % 8
    # multiplicities: how often each value is looked up by rows 0..n-2 (the last row is inert), placed on the first 8 rows
    counts = np.bincount(carries[:, :n - 1].reshape(-1), minlength=8)
    assert counts.size == 8, "a carry left [0, 8)"
    t[C.MULT, :8] = counts.astype(np.uint64)
    pis = np.array([int(t[C.D + k, n - 1]) for k in range(8)], dtype=np.uint64)
    return t, pis, digests


def aux_columns(trace, chal):
    """second-round columns [h1, h2, acc] for the challenge gamma (host arithmetic on Python integers)"""
    C = Cols
    n = trace.shape[1]
    g = int(chal[0])
    ca, ce, cw, tb, mu = (trace[c].astype(object) for c in (C.CA, C.CE, C.CW, C.TBL, C.MULT))
    inv = {v: pow((g - v) % P, P - 2, P) for v in range(8)}
    h1 = np.zeros(n, dtype=np.uint64)
    h2 = np.zeros(n, dtype=np.uint64)
    acc = np.zeros(n, dtype=np.uint64)
    run = 0
    for i in range(n):
        a = (inv[int(ca[i])] + inv[int(ce[i])]) % P
        b = (inv[int(cw[i])] - int(mu[i]) * inv[int(tb[i])]) % P
        h1[i], h2[i], acc[i] = a, b, run
        run = (run + a + b) % P
    return np.stack([h1, h2, acc])


def bus_tuple(words, beta):
    """sum_k beta^k w_k mod p — how a digest travels on the bus"""
    t = 0
    for w in reversed([int(x) for x in words]):
        t = (t * beta + w) % P
    return t


def aux_columns_bus(trace, chal):
    """second-round columns of the bus variant: [h1, h2, acc, bus_u, bus_acc] and the closing sum of the sends"""
    C = Cols
    n = trace.shape[1]
    base = aux_columns(trace, chal[:1])
    beta, g = int(chal[1]), int(chal[2])
    u = np.zeros(n, dtype=np.uint64)
    acc = np.zeros(n, dtype=np.uint64)
    run = 0
    send_rows = np.nonzero((trace[C.SEL + 65] == 1) & (trace[C.NF] == 1))[0]
    is_send = set(int(r) for r in send_rows)
    pw = 1 << np.arange(32, dtype=np.uint64)
    for i in range(n):
        acc[i] = run
        if i in is_send:
            words = [int((trace[C.S + 32 * k:C.S + 32 * k + 32, i] * pw).sum()) for k in range(8)]
            u[i] = pow((g - bus_tuple(words, beta)) % P, P - 2, P)
            run = (run + int(u[i])) % P
    return np.concatenate([base, np.stack([u, acc])]), np.array([int(acc[n - 1])], dtype=np.uint64)


def aux_program():
    """the GPU form of `aux_columns` (vx_stark_aux_columns; the table without the bus): h1 = 1/(g - ca) + 1/(g - ce),
    h2 = 1/(g - cw) - mult/(g - tbl) as fractions, acc = the running sum of h1 + h2"""
    from . import AuxProgram
    C = Cols
    e = _Emit(scratch=40)
    GAMMA = 63
    e.ins(VX_OP_LDCH, GAMMA, 0)
    ga, ge, gw, gt = (e.op(VX_OP_SUB, GAMMA, e.ldw(c)) for c in (C.CA, C.CE, C.CW, C.TBL))
    e.push(e.op(VX_OP_ADD, ga, ge), 0)
    e.push(e.op(VX_OP_MUL, ga, ge), 0)
    t = e.op(VX_OP_MUL, e.ldw(C.MULT), gw)
    e.push(e.op(VX_OP_SUB, gt, t), 0)
    e.push(e.op(VX_OP_MUL, gw, gt), 0)
    e.ins(VX_OP_END)
    return AuxProgram(C.N, 1, e.w, 2, [[1, 1]])


def make_stark(degree_bits: int, bus=False, **cfg) -> Stark:
    prog, _ = build_program(bus)
    cfg.setdefault("rate_bits", 1)
    if bus:
        return Stark(degree_bits, Cols.N, 8, prog, constraint_degree=3, num_aux_columns=5, num_aux_challenges=3, aux_fn=aux_columns_bus,
                     num_aux_public_inputs=1, **cfg)
    st = Stark(degree_bits, Cols.N, 8, prog, constraint_degree=3, num_aux_columns=3, num_aux_challenges=1, aux_fn=aux_columns, **cfg)
    st.aux_program = aux_program()
    return st


# ---- the other end of the bus: a table that RECEIVES digests (one per flagged row) -------------------------------------------
def sink_program():
    """columns d0..d7, flag; second round [u, acc]; aux challenges [unused, beta, gamma] (the bus challenges shared with the sender);
    public inputs = the first row's digest words; aux public input 0 = the closing sum (minus the received terms)."""
    e = _Emit()
    ONE = 63
    e.ldi(ONE, 1)
    npush = 0
    flag = e.ldw(8)
    t = e.op(VX_OP_SUB, flag, ONE)
    e.op(VX_OP_MUL, t, flag, t)
    e.push(t, VX_AIR_ALL_ROWS)
    beta, g = e.tmp(), e.tmp()
    e.ins(VX_OP_LDCH, beta, 1)
    e.ins(VX_OP_LDCH, g, 2)
    tup = e.ldw(7)
    for k in range(6, -1, -1):
        e.op(VX_OP_MUL, tup, beta, tup)
        e.op(VX_OP_ADD, tup, e.ldw(k), tup)
    d = e.op(VX_OP_SUB, g, tup)
    u, acc, accn = e.ldw(9), e.ldw(10), e.ldw(10, nxt=True)
    r = e.op(VX_OP_MUL, u, d)
    e.op(VX_OP_SUB, r, flag, r)
    e.push(r, VX_AIR_ALL_ROWS)                       # u (gamma - t) = flag
    r = e.op(VX_OP_SUB, accn, acc)
    e.op(VX_OP_ADD, r, u, r)
    e.push(r, VX_AIR_TRANSITION)                     # acc' = acc - u
    e.push(acc, VX_AIR_FIRST_ROW)
    cl = e.tmp()
    e.ins(VX_OP_LDP, cl, 8)
    r = e.op(VX_OP_SUB, acc, cl)
    e.push(r, VX_AIR_LAST_ROW)
    for k in range(8):
        pi = e.tmp()
        e.ins(VX_OP_LDP, pi, k)
        r = e.op(VX_OP_SUB, e.ldw(k), pi)
        e.push(r, VX_AIR_FIRST_ROW)
        e.release(pi)
    e.ins(VX_OP_END)
    return e.w


def sink_aux(trace, chal):
    n = trace.shape[1]
    beta, g = int(chal[1]), int(chal[2])
    u = np.zeros(n, dtype=np.uint64)
    acc = np.zeros(n, dtype=np.uint64)
    run = 0
    for i in range(n):
        acc[i] = run
        if int(trace[8, i]):
            u[i] = pow((g - bus_tuple(trace[:8, i], beta)) % P, P - 2, P)
            run = (run - int(u[i])) % P
    return np.stack([u, acc]), np.array([int(acc[n - 1])], dtype=np.uint64)


def make_sink(degree_bits: int, digests, fmt=">8I", **cfg):
    """-> (stark, trace [9][n], public inputs): row i receives digests[i] (32-byte strings, as the eight 32-bit values `fmt` unpacks:
    big-endian words for the SHA-256 table, little-endian limbs for the BLAKE2b table); the last row stays empty (it is inert in a
    running sum)."""
    n = 1 << degree_bits
    assert len(digests) < n
    t = np.zeros((9, n), dtype=np.uint64)
    for i, dg in enumerate(digests):
        t[:8, i] = struct.unpack(fmt, dg)
        t[8, i] = 1
    cfg.setdefault("rate_bits", 1)
    stark = Stark(degree_bits, 9, 8, sink_program(), constraint_degree=3, num_aux_columns=2, num_aux_challenges=3, aux_fn=sink_aux,
                  num_aux_public_inputs=1, **cfg)
    return stark, t, t[:8, 0].copy()


def reference_digests(messages):
    return [hashlib.sha256(m).digest() for m in messages]
