"""A BLAKE2b-256 AIR at chip density — caller-side stand-in for the BLAKE2b chip that Curta (starkyx v1.0.0,
/root/reference/Cargo.lock:7232-7249) puts under every VectorX map proof: the header hashes of
/root/reference/circuits/builder/header.rs:18 (`curta_blake2b_variable`) and the data-root / state-root chain of
circuits/builder/subchain_verification.rs.  OWN AIR, NOT CURTA'S (the starkyx sources are not in the reference tree): a second
table next to vectorx_amd/sha256_air.py with a different shape — 64-bit words kept as two 32-bit limbs, one G mixing function
per row with the four active words bit-decomposed and the other twelve copied as values, the message bytes range-checked by a
log-derivative lookup into a 256-entry table in the second commitment round — and checkable against `hashlib.blake2b`.
Plain host code: it emits a constraint program (include/vxprover.h VX_OP_*), generates the trace and the second-round
columns; no GPU, no oracle.

Layout.  A 128-byte block occupies 106 consecutive rows, marked by a cyclic one-hot `s`:
  row 0         init: the bits A hold the byte counter T; the transition writes the work vector
                v = (h[0..8], IV[0..4], IV4 ^ T, IV5, IV6 ^ (F ? ~0 : 0), IV7) into V;
  rows 1..96    the 96 G functions (12 rounds x 4 columns + 4 diagonals): the row holds V (16 words x 2 limbs, VALUES), the four
                active words a, b, c, d as bits (bound to V through the position selectors), every intermediate of G as bits
                (a1 = a + b + x, d1 = (d ^ a1) >>> 32, c1 = c + d1, b1 = (b ^ c1) >>> 24, a2 = a1 + b1 + y, d2 = (d1 ^ a2) >>> 16,
                c2 = c1 + d2, b2 = (b1 ^ c2) >>> 63) and the carries; the transition writes a2, b2, c2, d2 back into V;
                x, y are the message words m[sigma[r][2i]], m[sigma[r][2i + 1]] picked by the row type; rows 1..16 also hold the
                eight bytes of m[row - 1], looked up in the byte table;
  rows 97..104  finalisation, word k: a = v[k], b = v[k + 8], c = h[k] as bits; HN[k] = a ^ b ^ c (through E1 = a ^ b);
  row 105       hand-over: the next block's chaining value is HN, or the parameterised IV when this block was final (F = 1), in
                which case the first four words of HN are latched into D (the 32-byte digest).
The G constraints carry no selector: on rows that are not G rows they hold for x = y = 0 on whatever the bit columns contain,
so every XOR is degree 2 and every selector-gated relation degree <= 3.  Additions are limb-wise with carries in {0, 1, 2}
(three operands) or {0, 1}.  The counter obeys T = TB + 128 on non-final blocks, TB' = (1 - F) T at the hand-over; the final
block's T (the message length) and its zero padding are the caller's statement, like the padding of the SHA-256 table.
Public inputs: the 8 limbs of D in the last row = the digest of the last message completed inside the trace.
"""
from __future__ import annotations

import hashlib
import struct

import numpy as np

from . import (VX_AIR_ALL_ROWS, VX_AIR_FIRST_ROW, VX_AIR_LAST_ROW, VX_AIR_TRANSITION, VX_OP_ADD, VX_OP_END, VX_OP_LDCH, VX_OP_LDI, VX_OP_LDN,
               VX_OP_LDP, VX_OP_LDW, VX_OP_MUL, VX_OP_PUSH, VX_OP_SUB, Stark)
from .sha256_air import P, _Emit, bus_tuple

PERIOD = 106
ROW_INIT, ROW_G0, ROW_FIN0, ROW_HAND = 0, 1, 97, 105
MASK64 = (1 << 64) - 1

IV = [0x6a09e667f3bcc908, 0xbb67ae8584caa73b, 0x3c6ef372fe94f82b, 0xa54ff53a5f1d36f1, 0x510e527fade682d1, 0x9b05688c2b3e6c1f,
      0x1f83d9abfb41bd6b, 0x5be0cd19137e2179]
IVP = [IV[0] ^ 0x01010020] + IV[1:]            # parameter block: digest_length 32, no key, fanout = depth = 1
SIGMA = [[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15], [14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3],
         [11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4], [7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8],
         [9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13], [2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9],
         [12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11], [13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10],
         [6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5], [10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0]]
# position i of a round -> the indices of (a, b, c, d) in v: four columns, then four diagonals
PATTERN = [(0, 4, 8, 12), (1, 5, 9, 13), (2, 6, 10, 14), (3, 7, 11, 15), (0, 5, 10, 15), (1, 6, 11, 12), (2, 7, 8, 13), (3, 4, 9, 14)]


class Cols:
    """column map (1063 trace columns + 6 second-round columns, + 2 in the bus variant)"""
    SEL = 0                   # one-hot row type, 106 columns
    V = 106                   # V + 2 j + l: limb l (0 = low 32 bits) of work-vector word j
    H = 138                   # chaining value, 8 words x 2 limbs
    HN = 154                  # the next chaining value, filled by rows 97..104
    D = 170                   # the last completed digest: 4 words x 2 limbs
    M = 178                   # message block, 16 words x 2 limbs
    T, F, TB = 210, 211, 212  # byte counter of this block, final flag, counter of the previous block of the same message
    K = 213                   # carries: a1 (lo, hi), c1 (lo, hi), a2 (lo, hi), c2 (lo, hi)
    BY = 221                  # the eight bytes of m[row - 1] on rows 1..16, zero elsewhere
    TBL, MULT = 229, 230      # byte table 0..255 repeating, multiplicities
    BITS = 231                # BITS + 64 w + i: bit i of word w in (A, B, C, D, A1, D1, C1, B1, A2, D2, C2, B2)
    E1 = BITS + 12 * 64       # a ^ b
    N = E1 + 64
    AUX_H = N                 # four helpers: 1/(g - by[2q]) + 1/(g - by[2q+1])
    AUX_HT = N + 4            # mult / (g - tbl)
    AUX_ACC = N + 5
    AUX_BUS_U, AUX_BUS_ACC = N + 6, N + 7


W_A, W_B, W_C, W_D, W_A1, W_D1, W_C1, W_B1, W_A2, W_D2, W_C2, W_B2 = range(12)
K_A1, K_C1, K_A2, K_C2 = 0, 2, 4, 6


def _bit(w, i):
    return Cols.BITS + 64 * w + i


def _xy_rows():
    """gx[j] = the G rows (0..95) whose x operand is m[j]; gy likewise"""
    gx = [[] for _ in range(16)]
    gy = [[] for _ in range(16)]
    for t in range(96):
        r, i = divmod(t, 8)
        gx[SIGMA[r % 10][2 * i]].append(t)
        gy[SIGMA[r % 10][2 * i + 1]].append(t)
    return gx, gy


def build_program(bus=False):
    """-> (program words, number of constraints).  bus=True: the table also SENDS every completed digest (8 limbs) on a bus shared
    with other tables (vectorx_amd/stark_bus.py): aux challenges [gamma_range, beta, gamma_bus], two more second-round columns, and
    the closing sum of its sends as aux public input 0 (LDP index 8)."""
    C = Cols
    e = _Emit(scratch=40)
    # persistent registers (>= 40); scratch 0..39 is bump-allocated
    ONE, ZERO, TWO32, GAMMA, IS_G, IS_FIN, S0, S105, NOT105, XLO, XHI, YLO, YHI, Fr = 63, 62, 61, 60, 59, 58, 57, 56, 55, 54, 53, 52, 51, 50
    PAT = list(range(40, 48))
    e.ldi(ONE, 1)
    e.ldi(ZERO, 0)
    e.ldi(TWO32, 1 << 32)
    e.ins(VX_OP_LDCH, GAMMA, 0)
    npush = 0

    def push(r, kind):
        nonlocal npush
        e.push(r, kind)
        npush += 1

    tmp = e.tmp

    def sum_sel(rows, dst):
        """dst = sum of the row-type selectors of `rows`"""
        m0 = e.top
        first = True
        for r in rows:
            x = e.ldw(C.SEL + r)
            if first:
                e.op(VX_OP_ADD, x, ZERO, dst)
                first = False
            else:
                e.op(VX_OP_ADD, dst, x, dst)
            e.release(m0)
        if first:
            e.op(VX_OP_ADD, ZERO, ZERO, dst)

    # ---- row-type selectors ----
    e.ldw(C.SEL + ROW_INIT, dst=S0)
    e.ldw(C.SEL + ROW_HAND, dst=S105)
    e.op(VX_OP_SUB, ONE, S105, NOT105)
    e.ldw(C.F, dst=Fr)
    for p in range(8):
        sum_sel([ROW_G0 + 8 * r + p for r in range(12)], PAT[p])
    e.op(VX_OP_ADD, PAT[0], PAT[1], IS_G)
    for p in range(2, 8):
        e.op(VX_OP_ADD, IS_G, PAT[p], IS_G)
    sum_sel(range(ROW_FIN0, ROW_FIN0 + 8), IS_FIN)
    # ---- the message operands of the row: x = sum_j m_j [row uses m_j as x], y likewise (degree 2) ----
    gx, gy = _xy_rows()
    for rows_of, lo, hi in ((gx, XLO, XHI), (gy, YLO, YHI)):
        first = True
        for j in range(16):
            m0 = e.top
            g = tmp()
            sum_sel([ROW_G0 + t for t in rows_of[j]], g)
            for l, dst in ((0, lo), (1, hi)):
                m = e.ldw(C.M + 2 * j + l)
                e.op(VX_OP_MUL, m, g, m)
                if first:
                    e.op(VX_OP_ADD, m, ZERO, dst)
                else:
                    e.op(VX_OP_ADD, dst, m, dst)
            first = False
            e.release(m0)

    def boolean(col):
        m0 = e.top
        r = e.ldw(col)
        t = e.op(VX_OP_SUB, r, ONE)
        e.op(VX_OP_MUL, t, r, t)
        push(t, VX_AIR_ALL_ROWS)
        e.release(m0)

    def ternary(col):
        m0 = e.top
        r = e.ldw(col)
        t = e.op(VX_OP_SUB, r, ONE)
        e.op(VX_OP_MUL, t, r, t)
        u = e.op(VX_OP_SUB, r, ONE)
        e.op(VX_OP_SUB, u, ONE, u)
        e.op(VX_OP_MUL, t, u, t)
        push(t, VX_AIR_ALL_ROWS)
        e.release(m0)

    # ---- booleanity / carry ranges ----
    for c in range(C.BITS, C.N):
        boolean(c)
    boolean(C.F)
    for k in (K_A1, K_A1 + 1, K_A2, K_A2 + 1):
        ternary(C.K + k)
    for k in (K_C1, K_C1 + 1, K_C2, K_C2 + 1):
        boolean(C.K + k)

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
        t = tmp()
        xor_into(t, x, y)
        e.op(VX_OP_SUB, o, t, t)
        push(t, VX_AIR_ALL_ROWS)
        e.release(m0)

    # ---- the XORs of G (rotations are index shifts: bit i of x >>> r is bit (i + r) mod 64 of x) and E1 = a ^ b ----
    for i in range(64):
        define_xor(_bit(W_D1, i), _bit(W_D, (i + 32) % 64), _bit(W_A1, (i + 32) % 64))
        define_xor(_bit(W_B1, i), _bit(W_B, (i + 24) % 64), _bit(W_C1, (i + 24) % 64))
        define_xor(_bit(W_D2, i), _bit(W_D1, (i + 16) % 64), _bit(W_A2, (i + 16) % 64))
        define_xor(_bit(W_B2, i), _bit(W_B1, (i + 63) % 64), _bit(W_C2, (i + 63) % 64))
        define_xor(C.E1 + i, _bit(W_A, i), _bit(W_B, i))

    def word(w, l, dst):
        """dst = sum_i 2^i bit(w, 32 l + i), Horner from the top bit"""
        m0 = e.top
        e.ldw(_bit(w, 32 * l + 31), dst=dst)
        for i in range(30, -1, -1):
            e.op(VX_OP_ADD, dst, dst, dst)
            b = e.ldw(_bit(w, 32 * l + i))
            e.op(VX_OP_ADD, dst, b, dst)
            e.release(m0)
        return dst

    def bind(wreg, gate_terms, value_terms):
        """(sum of gate selectors) * word - sum_k selector_k * value column_k = 0  (degree 2, all rows)"""
        m0 = e.top
        g = tmp()
        e.op(VX_OP_ADD, gate_terms[0], ZERO, g)
        for r in gate_terms[1:]:
            e.op(VX_OP_ADD, g, r, g)
        e.op(VX_OP_MUL, g, wreg, g)
        for sel, col in value_terms:
            v = e.ldw(col)
            e.op(VX_OP_MUL, v, sel, v)
            e.op(VX_OP_SUB, g, v, g)
            e.release(v)
        push(g, VX_AIR_ALL_ROWS)
        e.release(m0)

    def fin_terms(base, l):
        """[(selector register of row 97 + k, column base + 2 k + l)] — loads the eight selectors into scratch"""
        out = []
        for k in range(8):
            out.append((e.ldw(C.SEL + ROW_FIN0 + k), base + 2 * k + l))
        return out

    # ---- per limb: the additions, the binding of a, b, c, d to V / H / T, and the write-back ----
    for l in (0, 1):
        mark = e.top
        wa, wb, wc, wd, wa1, wc1, wa2, wd2, wc2, wb2, wx = (tmp() for _ in range(11))
        # a1 = a + b + x
        word(W_A, l, wa)
        word(W_B, l, wb)
        word(W_A1, l, wa1)
        m0 = e.top
        t = e.op(VX_OP_ADD, wa, wb)
        e.op(VX_OP_ADD, t, XLO if l == 0 else XHI, t)
        if l:
            e.op(VX_OP_ADD, t, e.ldw(C.K + K_A1), t)
        e.op(VX_OP_SUB, t, wa1, t)
        k = e.ldw(C.K + K_A1 + l)
        e.op(VX_OP_MUL, k, TWO32, k)
        e.op(VX_OP_SUB, t, k, t)
        push(t, VX_AIR_ALL_ROWS)
        e.release(m0)
        # bind a: G rows <- V[ia], final rows <- V[k], init row <- T (low limb) / 0 (high limb)
        m0 = e.top
        terms = [(PAT[p], C.V + 2 * PATTERN[p][0] + l) for p in range(8)] + fin_terms(C.V, l)
        if l == 0:
            terms.append((S0, C.T))
        bind(wa, [IS_G, IS_FIN, S0], terms)
        e.release(m0)
        m0 = e.top
        bind(wb, [IS_G, IS_FIN], [(PAT[p], C.V + 2 * PATTERN[p][1] + l) for p in range(8)] + fin_terms(C.V + 16, l))
        e.release(m0)
        # c1 = c + d1
        word(W_C, l, wc)
        word(W_D1, l, wx)
        word(W_C1, l, wc1)
        m0 = e.top
        t = e.op(VX_OP_ADD, wc, wx)
        if l:
            e.op(VX_OP_ADD, t, e.ldw(C.K + K_C1), t)
        e.op(VX_OP_SUB, t, wc1, t)
        k = e.ldw(C.K + K_C1 + l)
        e.op(VX_OP_MUL, k, TWO32, k)
        e.op(VX_OP_SUB, t, k, t)
        push(t, VX_AIR_ALL_ROWS)
        e.release(m0)
        m0 = e.top
        bind(wc, [IS_G, IS_FIN], [(PAT[p], C.V + 2 * PATTERN[p][2] + l) for p in range(8)] + fin_terms(C.H, l))
        e.release(m0)
        word(W_D, l, wd)
        bind(wd, [IS_G], [(PAT[p], C.V + 2 * PATTERN[p][3] + l) for p in range(8)])
        # a2 = a1 + b1 + y
        word(W_B1, l, wx)
        word(W_A2, l, wa2)
        m0 = e.top
        t = e.op(VX_OP_ADD, wa1, wx)
        e.op(VX_OP_ADD, t, YLO if l == 0 else YHI, t)
        if l:
            e.op(VX_OP_ADD, t, e.ldw(C.K + K_A2), t)
        e.op(VX_OP_SUB, t, wa2, t)
        k = e.ldw(C.K + K_A2 + l)
        e.op(VX_OP_MUL, k, TWO32, k)
        e.op(VX_OP_SUB, t, k, t)
        push(t, VX_AIR_ALL_ROWS)
        e.release(m0)
        # c2 = c1 + d2
        word(W_D2, l, wd2)
        word(W_C2, l, wc2)
        m0 = e.top
        t = e.op(VX_OP_ADD, wc1, wd2)
        if l:
            e.op(VX_OP_ADD, t, e.ldw(C.K + K_C2), t)
        e.op(VX_OP_SUB, t, wc2, t)
        k = e.ldw(C.K + K_C2 + l)
        e.op(VX_OP_MUL, k, TWO32, k)
        e.op(VX_OP_SUB, t, k, t)
        push(t, VX_AIR_ALL_ROWS)
        e.release(m0)
        word(W_B2, l, wb2)
        # init values of the work vector (degree 1): v[0..8] = H, v[8..12] = IV[0..4], v12 = IV4 ^ T, v13 = IV5, v14 = IV6 ^ F.., v15 = IV7
        def init_value(j, dst):
            if j < 8:
                e.ldw(C.H + 2 * j + l, dst=dst)
            elif j == 12 and l == 0:
                # sum_i 2^i (IV4_i ? 1 - a_i : a_i) = IV4_lo + sum_i (+-2^i) a_i, on the T bits the init row holds in A
                const = IV[4] & 0xFFFFFFFF
                e.ldi(dst, const)
                m1 = e.top
                for i in range(32):
                    b, c = e.ldw(_bit(W_A, i)), tmp()
                    e.ldi(c, (1 << i) if not (const >> i) & 1 else P - (1 << i))
                    e.op(VX_OP_MUL, b, c, b)
                    e.op(VX_OP_ADD, dst, b, dst)
                    e.release(m1)
            elif j == 14:
                iv = (IV[6] >> (32 * l)) & 0xFFFFFFFF
                m1 = e.top
                c = tmp()
                e.ldi(c, ((iv ^ 0xFFFFFFFF) - iv) % P)
                e.op(VX_OP_MUL, c, Fr, c)
                c2 = tmp()
                e.ldi(c2, iv)
                e.op(VX_OP_ADD, c, c2, dst)
                e.release(m1)
            else:
                e.ldi(dst, (IV[j - 8] >> (32 * l)) & 0xFFFFFFFF)

        # write-back: (1 - s105)(V'_j - V_j) - (P_col(j) + P_diag(j)) (out_role(j) - V_j) - s0 (init_j - V_j) = 0
        outs = (wa2, wb2, wc2, wd2)
        for j in range(16):
            m0 = e.top
            role = j // 4
            pats = [p for p in range(8) if PATTERN[p][role] == j]
            assert len(pats) == 2
            v, vn = e.ldw(C.V + 2 * j + l), e.ldw(C.V + 2 * j + l, nxt=True)
            t = e.op(VX_OP_SUB, vn, v)
            e.op(VX_OP_MUL, t, NOT105, t)
            pp = e.op(VX_OP_ADD, PAT[pats[0]], PAT[pats[1]])
            u = e.op(VX_OP_SUB, outs[role], v)
            e.op(VX_OP_MUL, u, pp, u)
            e.op(VX_OP_SUB, t, u, t)
            iv = tmp()
            init_value(j, iv)
            e.op(VX_OP_SUB, iv, v, iv)
            e.op(VX_OP_MUL, iv, S0, iv)
            e.op(VX_OP_SUB, t, iv, t)
            push(t, VX_AIR_TRANSITION)
            e.release(m0)
        if l == 1:                                # init row: the counter has no high limb (messages < 2^32 bytes)
            m0 = e.top
            t = e.op(VX_OP_MUL, wa, S0)
            push(t, VX_AIR_ALL_ROWS)
            e.release(m0)
        # finalisation: HN'[k] = HN[k] + s_{97+k} (word(E1 ^ C) - HN[k]); wx = word of (e1 ^ c), degree 2
        m0 = e.top
        for i in range(31, -1, -1):
            if i != 31:
                e.op(VX_OP_ADD, wx, wx, wx)
            t = tmp()
            xor_into(t, e.ldw(C.E1 + 32 * l + i), e.ldw(_bit(W_C, 32 * l + i)))
            if i == 31:
                e.op(VX_OP_ADD, t, ZERO, wx)
            else:
                e.op(VX_OP_ADD, wx, t, wx)
            e.release(m0)
        for k in range(8):
            m0 = e.top
            hn, hnn = e.ldw(C.HN + 2 * k + l), e.ldw(C.HN + 2 * k + l, nxt=True)
            t = e.op(VX_OP_SUB, wx, hn)
            e.op(VX_OP_MUL, t, e.ldw(C.SEL + ROW_FIN0 + k), t)
            u = e.op(VX_OP_SUB, hnn, hn)
            e.op(VX_OP_SUB, u, t, u)
            push(u, VX_AIR_TRANSITION)
            e.release(m0)
        e.release(mark)

    # ---- chaining value, digest latch, message / counter / flag constancy ----
    for k in range(8):
        for l in (0, 1):
            m0 = e.top
            h, hn, hnew = e.ldw(C.H + 2 * k + l), e.ldw(C.H + 2 * k + l, nxt=True), e.ldw(C.HN + 2 * k + l)
            t = e.op(VX_OP_SUB, hn, h)
            e.op(VX_OP_MUL, t, NOT105, t)
            push(t, VX_AIR_TRANSITION)                                   # H' = H unless the row is the hand-over
            iv = tmp()
            e.ldi(iv, (IVP[k] >> (32 * l)) & 0xFFFFFFFF)
            u = e.op(VX_OP_SUB, iv, hnew)
            e.op(VX_OP_MUL, u, Fr, u)
            e.op(VX_OP_ADD, u, hnew, u)                                  # F IVP + (1 - F) HN
            e.op(VX_OP_SUB, hn, u, u)
            e.op(VX_OP_MUL, u, S105, u)
            push(u, VX_AIR_TRANSITION)
            t = e.op(VX_OP_SUB, h, iv)
            push(t, VX_AIR_FIRST_ROW)
            e.release(m0)
    for m in range(8):
        m0 = e.top
        d, dn, hnew = e.ldw(C.D + m), e.ldw(C.D + m, nxt=True), e.ldw(C.HN + m)
        u = e.op(VX_OP_SUB, hnew, d)
        e.op(VX_OP_MUL, u, Fr, u)
        e.op(VX_OP_MUL, u, S105, u)
        e.op(VX_OP_ADD, u, d, u)
        e.op(VX_OP_SUB, dn, u, u)
        push(u, VX_AIR_TRANSITION)                                       # D' = D + s105 F (HN - D)
        push(d, VX_AIR_FIRST_ROW)
        pi = tmp()
        e.ins(VX_OP_LDP, pi, m)
        t = e.op(VX_OP_SUB, d, pi)
        push(t, VX_AIR_LAST_ROW)
        e.release(m0)
    for col in list(range(C.M, C.M + 32)) + [C.T, C.F, C.TB]:
        m0 = e.top
        t = e.op(VX_OP_SUB, e.ldw(col, nxt=True), e.ldw(col))
        e.op(VX_OP_MUL, t, NOT105, t)
        push(t, VX_AIR_TRANSITION)
        e.release(m0)
    m0 = e.top
    c128 = tmp()
    e.ldi(c128, 128)
    t = e.op(VX_OP_SUB, e.ldw(C.T), e.ldw(C.TB))
    e.op(VX_OP_SUB, t, c128, t)
    nf = e.op(VX_OP_SUB, ONE, Fr)
    e.op(VX_OP_MUL, t, nf, t)
    push(t, VX_AIR_ALL_ROWS)                                             # (1 - F)(T - TB - 128) = 0
    u = e.op(VX_OP_MUL, nf, e.ldw(C.T))
    e.op(VX_OP_SUB, e.ldw(C.TB, nxt=True), u, u)
    e.op(VX_OP_MUL, u, S105, u)
    push(u, VX_AIR_TRANSITION)                                           # hand-over: TB' = (1 - F) T
    push(e.ldw(C.TB), VX_AIR_FIRST_ROW)
    e.release(m0)
    # ---- row type: cyclic shift of the one-hot; first row = (1, 0, .., 0) ----
    for i in range(PERIOD):
        m0 = e.top
        cur, nx = e.ldw(C.SEL + (i - 1) % PERIOD), e.ldw(C.SEL + i, nxt=True)
        push(e.op(VX_OP_SUB, nx, cur), VX_AIR_TRANSITION)
        e.release(m0)
        m0 = e.top
        r = e.ldw(C.SEL + i)
        if i == 0:
            r = e.op(VX_OP_SUB, r, ONE)
        push(r, VX_AIR_FIRST_ROW)
        e.release(m0)
    # ---- rows 1..16: the bytes of m[row - 1] ----
    m0 = e.top
    sel16 = tmp()
    sum_sel(range(ROW_G0, ROW_G0 + 16), sel16)
    for l in (0, 1):
        m1 = e.top
        w = e.ldw(C.BY + 4 * l + 3)
        c256 = tmp()
        e.ldi(c256, 256)
        for q in (2, 1, 0):
            e.op(VX_OP_MUL, w, c256, w)
            e.op(VX_OP_ADD, w, e.ldw(C.BY + 4 * l + q), w)
        e.op(VX_OP_MUL, w, sel16, w)
        for t16 in range(16):
            v = e.ldw(C.M + 2 * t16 + l)
            e.op(VX_OP_MUL, v, e.ldw(C.SEL + ROW_G0 + t16), v)
            e.op(VX_OP_SUB, w, v, w)
            e.release(c256 + 1)
        push(w, VX_AIR_ALL_ROWS)
        e.release(m1)
    e.release(m0)
    # ---- the byte table 0, 1, .., 255, 0, 1, .. and the log-derivative lookup of every BY column into it ----
    m0 = e.top
    tb, tbn = e.ldw(C.TBL), e.ldw(C.TBL, nxt=True)
    inc = e.op(VX_OP_SUB, tbn, tb)
    e.op(VX_OP_SUB, inc, ONE, inc)
    push(e.op(VX_OP_MUL, inc, tbn), VX_AIR_TRANSITION)                   # no increment => tbl' = 0 ...
    c255 = tmp()
    e.ldi(c255, 255)
    t = e.op(VX_OP_SUB, tb, c255)
    e.op(VX_OP_MUL, t, inc, t)
    push(t, VX_AIR_TRANSITION)                                           # ... and tbl = 255
    push(tb, VX_AIR_FIRST_ROW)
    e.release(m0)
    m0 = e.top
    acc, accn = e.ldw(C.AUX_ACC), e.ldw(C.AUX_ACC, nxt=True)
    step = e.op(VX_OP_SUB, accn, acc)
    for q in range(4):
        m1 = e.top
        g0 = e.op(VX_OP_SUB, GAMMA, e.ldw(C.BY + 2 * q))
        g1 = e.op(VX_OP_SUB, GAMMA, e.ldw(C.BY + 2 * q + 1))
        h = e.ldw(C.AUX_H + q)
        e.op(VX_OP_SUB, step, h, step)
        t = e.op(VX_OP_MUL, g0, g1)
        e.op(VX_OP_MUL, t, h, t)
        e.op(VX_OP_SUB, t, g0, t)
        e.op(VX_OP_SUB, t, g1, t)
        push(t, VX_AIR_ALL_ROWS)                                         # h (g - x)(g - y) = (g - x) + (g - y)
        e.release(m1)
    gt = e.op(VX_OP_SUB, GAMMA, e.ldw(C.TBL))
    ht = e.ldw(C.AUX_HT)
    e.op(VX_OP_ADD, step, ht, step)
    t = e.op(VX_OP_MUL, ht, gt)
    e.op(VX_OP_SUB, t, e.ldw(C.MULT), t)
    push(t, VX_AIR_ALL_ROWS)                                             # ht (g - tbl) = mult
    push(step, VX_AIR_TRANSITION)                                        # acc' = acc + sum h - ht
    push(acc, VX_AIR_FIRST_ROW)
    push(acc, VX_AIR_LAST_ROW)
    e.release(m0)
    if bus:
        # ---- bus: the hand-over row of a final block sends the digest limbs HN[0..8] as one tuple ----
        m0 = e.top
        beta, gbus = tmp(), tmp()
        e.ins(VX_OP_LDCH, beta, 1)
        e.ins(VX_OP_LDCH, gbus, 2)
        t = e.ldw(C.HN + 7)
        for m in range(6, -1, -1):
            e.op(VX_OP_MUL, t, beta, t)
            e.op(VX_OP_ADD, t, e.ldw(C.HN + m), t)
            e.release(t + 1)
        d = e.op(VX_OP_SUB, gbus, t)
        u, acc, accn = e.ldw(C.AUX_BUS_U), e.ldw(C.AUX_BUS_ACC), e.ldw(C.AUX_BUS_ACC, nxt=True)
        send = e.op(VX_OP_MUL, S105, Fr)
        r = e.op(VX_OP_MUL, u, d)
        e.op(VX_OP_SUB, r, send, r)
        push(r, VX_AIR_ALL_ROWS)
        r = e.op(VX_OP_SUB, accn, acc)
        e.op(VX_OP_SUB, r, u, r)
        push(r, VX_AIR_TRANSITION)
        push(acc, VX_AIR_FIRST_ROW)
        closing = tmp()
        e.ins(VX_OP_LDP, closing, 8)
        push(e.op(VX_OP_SUB, acc, closing), VX_AIR_LAST_ROW)
        e.release(m0)
    e.ins(VX_OP_END)
    return e.w, npush


# ---------------------------------------------------------------------------------------------------------------------
def _rotr64(x, r):
    return ((x >> r) | (x << (64 - r))) & MASK64


def g_words(a, b, c, d, x, y):
    """one G: -> (the 12 words in column order, the 8 carries)"""
    k = [0] * 8

    def add(terms, ki):
        lo = sum(t & 0xFFFFFFFF for t in terms)
        k[ki] = lo >> 32
        hi = sum(t >> 32 for t in terms) + k[ki]
        k[ki + 1] = hi >> 32
        return (lo & 0xFFFFFFFF) | ((hi & 0xFFFFFFFF) << 32)

    a1 = add((a, b, x), K_A1)
    d1 = _rotr64(d ^ a1, 32)
    c1 = add((c, d1), K_C1)
    b1 = _rotr64(b ^ c1, 24)
    a2 = add((a1, b1, y), K_A2)
    d2 = _rotr64(d1 ^ a2, 16)
    c2 = add((c1, d2), K_C2)
    b2 = _rotr64(b1 ^ c2, 63)
    return [a, b, c, d, a1, d1, c1, b1, a2, d2, c2, b2], k


def split_message(msg: bytes) -> list:
    """-> [(16 words, T, F)] per 128-byte block (the empty message is one zero block with T = 0)"""
    nb = max(1, (len(msg) + 127) // 128)
    data = msg + b"\x00" * (128 * nb - len(msg))
    out = []
    for i in range(nb):
        last = i == nb - 1
        out.append((list(struct.unpack("<16Q", data[128 * i:128 * i + 128])), len(msg) if last else 128 * (i + 1), 1 if last else 0))
    return out


def generate_trace(degree_bits: int, messages) -> tuple:
    """-> (trace [1063][n] uint64, public inputs [8], digests of the messages completed inside the trace).
    `messages`: byte strings hashed one after the other; rows that remain after the last message keep hashing blocks of an
    endless zero-message (never final), so every row is a valid row.  The LAST COMPLETED digest is the public input."""
    C = Cols
    n = 1 << degree_bits
    assert degree_bits >= 9, "the byte table needs 256 rows before the (inert) last row"
    blocks = []
    for m in messages:
        blocks.extend(split_message(m))
    lim = lambda w: (w & 0xFFFFFFFF, w >> 32)
    # per-row python ints, expanded to columns at the end
    words = np.zeros((12, n), dtype=np.uint64)
    e1 = np.zeros(n, dtype=np.uint64)
    t = np.zeros((C.N, n), dtype=np.uint64)
    H = list(IVP)
    HN = [0] * 8
    D = [0] * 4
    V = [0] * 16
    TB = 0
    digests = []
    row, bi = 0, 0
    lookups = np.zeros(256, dtype=np.int64)
    while row < n:
        if bi < len(blocks):
            M, T, F = blocks[bi]
        else:
            M, T, F = [0] * 16, TB + 128, 0
        bi += 1
        for r in range(PERIOD):
            if row >= n:
                break
            t[C.SEL + r, row] = 1
            for j in range(16):
                t[C.V + 2 * j, row], t[C.V + 2 * j + 1, row] = lim(V[j])
                t[C.M + 2 * j, row], t[C.M + 2 * j + 1, row] = lim(M[j])
            for k in range(8):
                t[C.H + 2 * k, row], t[C.H + 2 * k + 1, row] = lim(H[k])
                t[C.HN + 2 * k, row], t[C.HN + 2 * k + 1, row] = lim(HN[k])
            for k in range(4):
                t[C.D + 2 * k, row], t[C.D + 2 * k + 1, row] = lim(D[k])
            t[C.T, row], t[C.F, row], t[C.TB, row] = T, F, TB
            a = b = c = d = x = y = 0
            if r == ROW_INIT:
                a = T
            elif r < ROW_FIN0:
                rnd, i = divmod(r - ROW_G0, 8)
                ia, ib, ic, id_ = PATTERN[i]
                a, b, c, d = V[ia], V[ib], V[ic], V[id_]
                x, y = M[SIGMA[rnd % 10][2 * i]], M[SIGMA[rnd % 10][2 * i + 1]]
                if r - ROW_G0 < 16:
                    by = struct.pack("<Q", M[r - ROW_G0])
                    for q in range(8):
                        t[C.BY + q, row] = by[q]
            elif r < ROW_HAND:
                k = r - ROW_FIN0
                a, b, c = V[k], V[k + 8], H[k]
            ws, ks = g_words(a, b, c, d, x, y)
            for w in range(12):
                words[w, row] = ws[w]
            e1[row] = a ^ b
            for q in range(8):
                t[C.K + q, row] = ks[q]
            if row < n - 1:
                for q in range(8):
                    lookups[int(t[C.BY + q, row])] += 1
            # the transition out of this row
            if r == ROW_INIT:
                V = H[:8] + IV[:4] + [IV[4] ^ T, IV[5], IV[6] ^ (MASK64 if F else 0), IV[7]]
            elif r < ROW_FIN0:
                V = list(V)
                V[ia], V[ib], V[ic], V[id_] = ws[W_A2], ws[W_B2], ws[W_C2], ws[W_D2]
            elif r < ROW_HAND:
                HN = list(HN)
                HN[r - ROW_FIN0] = a ^ b ^ c
            else:
                if F:
                    D = HN[:4]
                    digests.append(struct.pack("<4Q", *D))
                    H = list(IVP)
                    TB = 0
                else:
                    H = list(HN)
                    TB = T
            row += 1
    for w in range(12):
        for i in range(64):
            t[_bit(w, i)] = (words[w] >> np.uint64(i)) & np.uint64(1)
    for i in range(64):
        t[C.E1 + i] = (e1 >> np.uint64(i)) & np.uint64(1)
    t[C.TBL] = np.arange(n, dtype=np.uint64) % 256
    t[C.MULT, :256] = lookups.astype(np.uint64)
    pis = t[C.D:C.D + 8, n - 1].copy()
    return t, pis, digests


def _inv_of(g, values):
    """1 / (g - v) mod p for every v of a uint64 array (one modular inversion per DISTINCT value)"""
    uniq, idx = np.unique(values, return_inverse=True)
    inv = np.array([pow((g - int(v)) % P, P - 2, P) for v in uniq], dtype=object)
    return inv[idx]


def aux_columns(trace, chal):
    """second-round columns [h0..h3, ht, acc] for the challenge gamma"""
    C = Cols
    n = trace.shape[1]
    g = int(chal[0])
    by = [_inv_of(g, trace[C.BY + q]) for q in range(8)]
    h = [(by[2 * q] + by[2 * q + 1]) % P for q in range(4)]
    ht = (trace[C.MULT].astype(object) * _inv_of(g, trace[C.TBL])) % P
    step = (h[0] + h[1] + h[2] + h[3] - ht) % P
    acc = np.zeros(n, dtype=object)
    run = 0
    for i in range(n):
        acc[i] = run
        run = (run + int(step[i])) % P
    return np.stack([np.array(c, dtype=np.uint64) for c in (h[0], h[1], h[2], h[3], ht, acc)])


def digest_limbs(dg: bytes) -> list:
    return list(struct.unpack("<8I", dg))


def aux_columns_bus(trace, chal):
    """second-round columns of the bus variant: [h0..h3, ht, acc, bus_u, bus_acc] and the closing sum of the sends"""
    C = Cols
    n = trace.shape[1]
    base = aux_columns(trace, chal[:1])
    beta, g = int(chal[1]), int(chal[2])
    u = np.zeros(n, dtype=np.uint64)
    acc = np.zeros(n, dtype=np.uint64)
    run = 0
    is_send = set(int(r) for r in np.nonzero((trace[C.SEL + ROW_HAND] == 1) & (trace[C.F] == 1))[0])
    for i in range(n):
        acc[i] = run
        if i in is_send:
            u[i] = pow((g - bus_tuple(trace[C.HN:C.HN + 8, i], beta)) % P, P - 2, P)
            run = (run + int(u[i])) % P
    return np.concatenate([base, np.stack([u, acc])]), np.array([int(acc[n - 1])], dtype=np.uint64)


def make_stark(degree_bits: int, bus=False, **cfg) -> Stark:
    prog, _ = build_program(bus)
    cfg.setdefault("rate_bits", 1)
    if bus:
        return Stark(degree_bits, Cols.N, 8, prog, constraint_degree=3, num_aux_columns=8, num_aux_challenges=3, aux_fn=aux_columns_bus,
                     num_aux_public_inputs=1, **cfg)
    return Stark(degree_bits, Cols.N, 8, prog, constraint_degree=3, num_aux_columns=6, num_aux_challenges=1, aux_fn=aux_columns, **cfg)


def reference_digests(messages):
    return [hashlib.blake2b(m, digest_size=32).digest() for m in messages]
