"""BLAKE2b-256 on BYTES and a XOR lookup table — the second generation of the header-hash table (round 3's blake2b_air.py carries every
G intermediate as 64 bit columns: 1063 + 12 columns x 106 rows = 114 k cells per compression, and 64 map jobs x 2240 compressions made
it the largest single item of a header_range_512 with its STARKs, profiles/r04_bench_n1_a.json).  Caller-side stand-in for Curta's
`curta_blake2b_variable` (/root/reference/circuits/builder/header.rs:18; starkyx v1.0.0, /root/reference/Cargo.lock:7232-7249) — OWN AIR,
NOT CURTA'S (the starkyx sources are not in the tree), but closer to how a byte-oriented chip works: 64-bit words are 8 bytes, a XOR is
eight lookups of (a, b, a ^ b) in a 65 536-entry table, the rotations by 32, 24 and 16 are byte permutations (free: wiring), the
rotation by 63 is one top bit per byte, additions run on 32-bit limbs with carries.

Layout: a 128-byte block occupies 28 rows (cyclic one-hot SEL):
  row 0        init: the slot columns present the initial work vector as if a diagonal step had produced it; T ^ IV4 through the
               finalisation XOR columns;
  rows 1..24   the 12 rounds, FOUR G functions per row (odd rows: the column step, even rows: the diagonal step).  Slot i of a row
               holds its inputs b, d as bytes (BIN, DIN) and a, c as limbs (AL, CL) — copied from the previous row's outputs by
               transition constraints whose wiring depends on the NEXT row's parity — and every intermediate as bytes:
                 a1 = a + b + x;  e1 = d ^ a1 (d1 = e1 >>> 32);  c1 = c + d1;  f1 = b ^ c1 (b1 = f1 >>> 24);
                 a2 = a1 + b1 + y; e2 = d1 ^ a2 (d2 = e2 >>> 16);  c2 = c1 + d2;  f2 = b1 ^ c2;  b2 = f2 <<< 1 (TOP bits);
               rows 1..16 also hold the bytes of m[row - 1] (range check of the message);
  rows 25, 26  finalisation, four words per row (groups g = 0..3 of FA, FB, FE, FH, FO): HN[k] = v[k] ^ v[k + 8] ^ h[k] for k = 4 r + g
               (the work vector is latched in V);
  row 27       hand-over: the next block's chaining value is HN, or the parameterised IV after a final block, whose digest is latched in D.
40 lookups per slot (32 XOR triples + the 8 bytes of b2 as (b2, 0, b2)), 64 for the finalisation, 8 for the message bytes: 232 per row,
into a XOR table held as TWO half tables side by side (rows i < 32 768 of half k list a = 128 k + (i >> 8), b = i & 255): each half's
columns (TA, TB, TC = TA ^ TB) are constrained through 16 bit columns, so EVERY table row is a valid triple whatever the prover puts
there — the argument needs no counter constraints, only that the prover lists the entries it uses.  A triple enters as
a + beta b + beta^2 c with a second challenge beta, so that no combination of non-bytes aliases a table entry.
775 + 2 x 119 columns x 28 rows = 28 k cells per compression — a quarter of round 3's — and, what counts, 2240 compressions (one map
job) fit 2^16 rows instead of 2^18: the first version of this table (one 65 536-row table, 34 rows per compression, 8 finalisation
rows) needed 2^17 rows for them and proved in 26 ms; this one needs half the rows.  Traces of >= 2^16 rows.

Public inputs: the 8 limbs of D in the last row = the digest of the last message completed inside the trace.
Plain host code: it emits a constraint program (include/vxprover.h VX_OP_*), generates the trace and the second-round columns; checked
against `hashlib.blake2b` (tests/test_blake2b_bytes_air.py); no GPU, no oracle."""
from __future__ import annotations

import hashlib

import numpy as np

from . import (VX_AIR_ALL_ROWS, VX_AIR_FIRST_ROW, VX_AIR_LAST_ROW, VX_AIR_TRANSITION, VX_OP_ADD, VX_OP_END, VX_OP_LDCH, VX_OP_LDP, VX_OP_MUL,
               VX_OP_SUB, Stark)
from . import hostfield as hf
from .blake2b_air import IV, IVP, MASK64, SIGMA, split_message
from .sha256_air import P, _Emit

PERIOD = 28
ROW_INIT, ROW_G0, ROW_GLAST, ROW_FIN0, ROW_HAND = 0, 1, 24, 25, 27
NFINROW, NFING = 2, 4         # finalisation: 2 rows x 4 word groups
NTAB, TAB_ROWS = 2, 32768     # two half tables of 32 768 entries
NSLOT = 4
FIELDS = ["BIN", "DIN", "A1", "E1", "C1", "F1", "A2", "E2", "C2", "F2", "TOP", "B2"]      # 8 byte columns each
K_A1, K_C1, K_A2, K_C2 = 0, 2, 4, 6


class Cols:
    SEL = 0
    H = SEL + PERIOD          # chaining value, 8 words x 2 limbs
    HN = H + 16
    D = HN + 16               # the last completed digest, 4 words x 2 limbs
    M = D + 8                 # message block, 16 words x 2 limbs
    T, F, TB = M + 32, M + 33, M + 34
    V = M + 35                # the work vector after the 12 rounds, latched for the finalisation rows
    SLOT = V + 32             # per slot: 12 byte fields, then AL(2), CL(2) limbs, then 8 carries
    SLOT_W = 8 * len(FIELDS) + 4 + 8
    FIN = SLOT + NSLOT * SLOT_W    # per group g: FA, FB, FE, FH, FO (8 bytes each)
    BY = FIN + 40 * NFING
    TAB = BY + 8                   # per half table: TA, TB, TC = TA ^ TB, the bits of TA and TB (16), multiplicities
    N = TAB + 20 * NTAB
    NTUP = NSLOT * 40 + 16 * NFING + 8
    NPAIR = NTUP // 2
    AUX_H = N
    AUX_HT = N + NPAIR             # one table helper per half table
    AUX_ACC = AUX_HT + NTAB
    NAUX = NPAIR + NTAB + 1

    @staticmethod
    def fin(g, name, j=0):
        return Cols.FIN + 40 * g + 8 * ("FA", "FB", "FE", "FH", "FO").index(name) + j

    @staticmethod
    def tab(k, name, j=0):
        return Cols.TAB + 20 * k + {"TA": 0, "TB": 1, "TC": 2, "BITS": 3, "MULT": 19}[name] + j

    @staticmethod
    def f(g, name, j=0):
        return Cols.SLOT + g * Cols.SLOT_W + 8 * FIELDS.index(name) + j

    @staticmethod
    def al(g, l):
        return Cols.SLOT + g * Cols.SLOT_W + 8 * len(FIELDS) + l

    @staticmethod
    def cl(g, l):
        return Cols.SLOT + g * Cols.SLOT_W + 8 * len(FIELDS) + 2 + l

    @staticmethod
    def k(g, i):
        return Cols.SLOT + g * Cols.SLOT_W + 8 * len(FIELDS) + 4 + i


def tuples():
    """the looked-up triples of a row as (column a, column b or None, column c), in the order the pair helpers take them"""
    C, out = Cols, []
    for g in range(NSLOT):
        for j in range(8):
            out.append((C.f(g, "DIN", j), C.f(g, "A1", j), C.f(g, "E1", j)))
            out.append((C.f(g, "BIN", j), C.f(g, "C1", j), C.f(g, "F1", j)))
            out.append((C.f(g, "E1", (j + 4) % 8), C.f(g, "A2", j), C.f(g, "E2", j)))        # d1 = e1 >>> 32
            out.append((C.f(g, "F1", (j + 3) % 8), C.f(g, "C2", j), C.f(g, "F2", j)))        # b1 = f1 >>> 24
            out.append((C.f(g, "B2", j), None, C.f(g, "B2", j)))                             # range: (b2, 0, b2)
    for g in range(NFING):
        for j in range(8):
            out.append((C.fin(g, "FA", j), C.fin(g, "FB", j), C.fin(g, "FE", j)))
            out.append((C.fin(g, "FE", j), C.fin(g, "FH", j), C.fin(g, "FO", j)))
    for j in range(8):
        out.append((C.BY + j, None, C.BY + j))
    assert len(out) == C.NTUP
    return out


def _msg_index(s, i, which):
    """message word used by slot i of G row s (1..24) as x (which = 0) or y (1)"""
    r, half = divmod(s - 1, 2)
    return SIGMA[r % 10][2 * (4 * half + i) + which]


def build_program():
    """-> (program words for ONE challenge set [gamma, beta], number of constraints)"""
    C = Cols
    e = _Emit(scratch=40)
    ONE, ZERO, C256, TWO32, GAMMA, BETA, BETA2, ISG, ISFIN, S0, S33, NOT33, Fr, COLN, DIAGN, S24, FINHOLD = 63, 62, 61, 60, 59, 58, 57, 56, 55, 54, 53, 52, 51, 50, 49, 48, 47
    e.ldi(ONE, 1)
    e.ldi(ZERO, 0)
    e.ldi(C256, 256)
    e.ldi(TWO32, 1 << 32)
    e.ins(VX_OP_LDCH, GAMMA, 0)
    e.ins(VX_OP_LDCH, BETA, 1)
    e.op(VX_OP_MUL, BETA, BETA, BETA2)
    npush = 0
    tmp = e.tmp

    def push(r, kind):
        nonlocal npush
        e.push(r, kind)
        npush += 1

    def sum_sel(rows, dst, nxt=False):
        m0 = e.top
        first = True
        for r in rows:
            x = e.ldw(C.SEL + r, nxt=nxt)
            e.op(VX_OP_ADD, x, ZERO if first else dst, dst)
            first = False
            e.release(m0)
        if first:
            e.op(VX_OP_ADD, ZERO, ZERO, dst)

    def limb(cols4, dst, nxt=False):
        """dst = c0 + 256 c1 + 65536 c2 + 2^24 c3"""
        m0 = e.top
        e.ldw(cols4[3], nxt=nxt, dst=dst)
        for c in (cols4[2], cols4[1], cols4[0]):
            e.op(VX_OP_MUL, dst, C256, dst)
            e.op(VX_OP_ADD, dst, e.ldw(c, nxt=nxt), dst)
            e.release(m0)
        return dst

    def boolean(col):
        m0 = e.top
        r = e.ldw(col)
        t = e.op(VX_OP_SUB, r, ONE)
        push(e.op(VX_OP_MUL, t, r), VX_AIR_ALL_ROWS)
        e.release(m0)

    def ternary(col):
        m0 = e.top
        r = e.ldw(col)
        t = e.op(VX_OP_SUB, r, ONE)
        e.op(VX_OP_MUL, t, r, t)
        u = e.op(VX_OP_SUB, r, ONE)
        e.op(VX_OP_SUB, u, ONE, u)
        push(e.op(VX_OP_MUL, t, u), VX_AIR_ALL_ROWS)
        e.release(m0)

    g_rows = list(range(ROW_G0, ROW_GLAST + 1))
    sum_sel(g_rows, ISG)
    sum_sel(range(ROW_FIN0, ROW_FIN0 + NFINROW), ISFIN)
    sum_sel(range(ROW_FIN0, ROW_FIN0 + NFINROW - 1), FINHOLD)
    sum_sel([s for s in g_rows if s % 2 == 1], COLN, nxt=True)
    sum_sel([s for s in g_rows if s % 2 == 0], DIAGN, nxt=True)
    e.ldw(C.SEL + ROW_INIT, dst=S0)
    e.ldw(C.SEL + ROW_HAND, dst=S33)
    e.ldw(C.SEL + ROW_GLAST, dst=S24)
    e.op(VX_OP_SUB, ONE, S33, NOT33)
    e.ldw(C.F, dst=Fr)
    # ---- ranges: top bits, the final flag, the table's bits, the carries ----
    for g in range(NSLOT):
        for j in range(8):
            boolean(C.f(g, "TOP", j))
        for i in (K_A1, K_A1 + 1, K_A2, K_A2 + 1):
            ternary(C.k(g, i))
        for i in (K_C1, K_C1 + 1, K_C2, K_C2 + 1):
            boolean(C.k(g, i))
    boolean(C.F)
    for k in range(NTAB):
        for i in range(16):
            boolean(C.tab(k, "BITS", i))
    # ---- wiring: the inputs of a G row are the previous row's outputs (column step <-> diagonal step) ----
    isgn = tmp()
    e.op(VX_OP_ADD, COLN, DIAGN, isgn)
    for i in range(NSLOT):
        for l in (0, 1):
            m0 = e.top
            src = tmp()
            limb([C.f(i, "A2", 4 * l + q) for q in range(4)], src)
            t = e.op(VX_OP_SUB, e.ldw(C.al(i, l), nxt=True), src)
            push(e.op(VX_OP_MUL, t, isgn), VX_AIR_TRANSITION)
            e.release(m0)
            m0 = e.top
            src = tmp()
            limb([C.f((i + 2) % 4, "C2", 4 * l + q) for q in range(4)], src)
            t = e.op(VX_OP_SUB, e.ldw(C.cl(i, l), nxt=True), src)
            push(e.op(VX_OP_MUL, t, isgn), VX_AIR_TRANSITION)
            e.release(m0)
        for j in range(8):
            m0 = e.top
            nb = e.ldw(C.f(i, "BIN", j), nxt=True)
            t = e.op(VX_OP_SUB, nb, e.ldw(C.f((i + 3) % 4, "B2", j)))
            e.op(VX_OP_MUL, t, COLN, t)
            u = e.op(VX_OP_SUB, nb, e.ldw(C.f((i + 1) % 4, "B2", j)))
            e.op(VX_OP_MUL, u, DIAGN, u)
            push(e.op(VX_OP_ADD, t, u), VX_AIR_TRANSITION)
            e.release(m0)
            m0 = e.top
            nd = e.ldw(C.f(i, "DIN", j), nxt=True)
            t = e.op(VX_OP_SUB, nd, e.ldw(C.f((i + 1) % 4, "E2", (j + 2) % 8)))     # d2 = e2 >>> 16
            e.op(VX_OP_MUL, t, COLN, t)
            u = e.op(VX_OP_SUB, nd, e.ldw(C.f((i + 3) % 4, "E2", (j + 2) % 8)))
            e.op(VX_OP_MUL, u, DIAGN, u)
            push(e.op(VX_OP_ADD, t, u), VX_AIR_TRANSITION)
            e.release(m0)
    e.release(isgn)
    # ---- the G relations of every slot (gated by the row type; the message operands carry their own selectors) ----
    for i in range(NSLOT):
        for l in (0, 1):
            mark = e.top
            x, y = tmp(), tmp()
            for which, dst in ((0, x), (1, y)):
                e.op(VX_OP_ADD, ZERO, ZERO, dst)
                for j in range(16):
                    rows = [s for s in g_rows if _msg_index(s, i, which) == j]
                    if not rows:
                        continue
                    m0 = e.top
                    gsel = tmp()
                    sum_sel(rows, gsel)
                    e.op(VX_OP_MUL, gsel, e.ldw(C.M + 2 * j + l), gsel)
                    e.op(VX_OP_ADD, dst, gsel, dst)
                    e.release(m0)

            def add_relation(out_cols, in_terms, kbase, msg=None):
                """ISG (out + 2^32 K_l - sum(in) - [l] K_0) - msg = 0"""
                m0 = e.top
                t = tmp()
                limb(out_cols, t)
                kk = e.op(VX_OP_MUL, e.ldw(C.k(i, kbase + l)), TWO32)
                e.op(VX_OP_ADD, t, kk, t)
                for term in in_terms:
                    m1 = e.top
                    if isinstance(term, int):
                        v = e.ldw(term)
                    else:
                        v = tmp()
                        limb(term, v)
                    e.op(VX_OP_SUB, t, v, t)
                    e.release(m1)
                if l:
                    e.op(VX_OP_SUB, t, e.ldw(C.k(i, kbase)), t)
                e.op(VX_OP_MUL, t, ISG, t)
                if msg is not None:
                    e.op(VX_OP_SUB, t, msg, t)
                push(t, VX_AIR_ALL_ROWS)
                e.release(m0)

            by = lambda name, idx: [C.f(i, name, q % 8) for q in idx]     # noqa: E731
            lo4 = range(4 * l, 4 * l + 4)
            add_relation(by("A1", lo4), [C.al(i, l), by("BIN", lo4)], K_A1, msg=x)                                   # a1 = a + b + x
            add_relation(by("C1", lo4), [C.cl(i, l), by("E1", [q + 4 for q in lo4])], K_C1)                         # c1 = c + (e1 >>> 32)
            add_relation(by("A2", lo4), [by("A1", lo4), by("F1", [q + 3 for q in lo4])], K_A2, msg=y)              # a2 = a1 + (f1 >>> 24) + y
            add_relation(by("C2", lo4), [by("C1", lo4), by("E2", [q + 2 for q in lo4])], K_C2)                      # c2 = c1 + (e2 >>> 16)
            e.release(mark)
        for j in range(8):                                  # b2 = f2 <<< 1: byte j = 2 (f2_j - 128 top_j) + top_{j-1}
            m0 = e.top
            t = e.op(VX_OP_ADD, e.ldw(C.f(i, "F2", j)), e.ldw(C.f(i, "F2", j)))
            u = e.op(VX_OP_MUL, e.ldw(C.f(i, "TOP", j)), C256)
            e.op(VX_OP_SUB, t, u, t)
            e.op(VX_OP_ADD, t, e.ldw(C.f(i, "TOP", (j + 7) % 8)), t)
            e.op(VX_OP_SUB, e.ldw(C.f(i, "B2", j)), t, t)
            push(e.op(VX_OP_MUL, t, ISG), VX_AIR_ALL_ROWS)
            e.release(m0)
    # ---- the work vector after the last round, latched for the finalisation rows ----
    out_of = {}
    for j in range(NSLOT):
        out_of[j] = (j, "A2", 0)
        out_of[4 + (j + 1) % 4] = (j, "B2", 0)
        out_of[8 + (j + 2) % 4] = (j, "C2", 0)
        out_of[12 + (j + 3) % 4] = (j, "E2", 2)              # d2 = e2 >>> 16
    for w in range(16):
        g, name, rot = out_of[w]
        for l in (0, 1):
            m0 = e.top
            src = tmp()
            limb([C.f(g, name, (4 * l + q + rot) % 8) for q in range(4)], src)
            vn, v = e.ldw(C.V + 2 * w + l, nxt=True), e.ldw(C.V + 2 * w + l)
            t = e.op(VX_OP_SUB, vn, src)
            e.op(VX_OP_MUL, t, S24, t)
            u = e.op(VX_OP_SUB, vn, v)
            e.op(VX_OP_MUL, u, FINHOLD, u)
            push(e.op(VX_OP_ADD, t, u), VX_AIR_TRANSITION)
            e.release(m0)
    # ---- finalisation rows r = 0, 1, group g: FA = v[4 r + g], FB = v[4 r + g + 8], FH = h[4 r + g]; HN[4 r + g] <- FO = FA ^ FB ^ FH.
    #      Init row: group 0 holds FA = T, FB = IV4 ----
    for g in range(NFING):
        for l in (0, 1):
            for name, src, extra in (("FA", C.V, "T"), ("FB", C.V + 16, "IV4"), ("FH", C.H, None)):
                m0 = e.top
                t = tmp()
                limb([C.fin(g, name, 4 * l + q) for q in range(4)], t)
                gate = e.op(VX_OP_ADD, ISFIN, S0 if (extra and g == 0) else ZERO)
                e.op(VX_OP_MUL, t, gate, t)
                for r in range(NFINROW):
                    m1 = e.top
                    v = e.op(VX_OP_MUL, e.ldw(src + 2 * (4 * r + g) + l), e.ldw(C.SEL + ROW_FIN0 + r))
                    e.op(VX_OP_SUB, t, v, t)
                    e.release(m1)
                if g == 0 and extra == "T" and l == 0:
                    e.op(VX_OP_SUB, t, e.op(VX_OP_MUL, e.ldw(C.T), S0), t)
                if g == 0 and extra == "IV4":
                    c = tmp()
                    e.ldi(c, (IV[4] >> (32 * l)) & 0xFFFFFFFF)
                    e.op(VX_OP_MUL, c, S0, c)
                    e.op(VX_OP_SUB, t, c, t)
                push(t, VX_AIR_ALL_ROWS)
                e.release(m0)
            m0 = e.top
            wx = tmp()
            limb([C.fin(g, "FO", 4 * l + q) for q in range(4)], wx)
            for r in range(NFINROW):
                m1 = e.top
                k = 4 * r + g
                hn, hnn = e.ldw(C.HN + 2 * k + l), e.ldw(C.HN + 2 * k + l, nxt=True)
                t = e.op(VX_OP_SUB, wx, hn)
                e.op(VX_OP_MUL, t, e.ldw(C.SEL + ROW_FIN0 + r), t)
                u = e.op(VX_OP_SUB, hnn, hn)
                push(e.op(VX_OP_SUB, u, t), VX_AIR_TRANSITION)          # HN'[k] = HN[k] + s_{25+r} (FO_g - HN[k])
                e.release(m1)
            e.release(m0)
    # ---- init row: the slots present v = (h, IV[0..4], IV4 ^ T, IV5, IV6 ^ (F ? ~0 : 0), IV7) in the diagonal-output arrangement ----
    for w in range(16):
        g, name, rot = out_of[w]
        for l in (0, 1):
            m0 = e.top
            t = tmp()
            limb([C.f(g, name, (4 * l + q + rot) % 8) for q in range(4)], t)
            if w < 8:
                e.op(VX_OP_SUB, t, e.ldw(C.H + 2 * w + l), t)
            elif w == 12:
                v = tmp()
                limb([C.fin(0, "FE", 4 * l + q) for q in range(4)], v)
                e.op(VX_OP_SUB, t, v, t)
            elif w == 14:
                iv = (IV[6] >> (32 * l)) & 0xFFFFFFFF
                c = tmp()
                e.ldi(c, ((iv ^ 0xFFFFFFFF) - iv) % P)
                e.op(VX_OP_MUL, c, Fr, c)
                c2 = tmp()
                e.ldi(c2, iv)
                e.op(VX_OP_ADD, c, c2, c)
                e.op(VX_OP_SUB, t, c, t)
            else:
                c = tmp()
                e.ldi(c, (IV[w - 8] >> (32 * l)) & 0xFFFFFFFF)
                e.op(VX_OP_SUB, t, c, t)
            push(e.op(VX_OP_MUL, t, S0), VX_AIR_ALL_ROWS)
            e.release(m0)
    m0 = e.top
    t = tmp()
    limb([C.fin(0, "FA", 4 + q) for q in range(4)], t)
    push(e.op(VX_OP_MUL, t, S0), VX_AIR_ALL_ROWS)                        # the counter has no high limb (messages < 2^32 bytes)
    e.release(m0)
    # ---- chaining value, digest latch, message / counter / flag constancy (as in blake2b_air.py) ----
    for k in range(8):
        for l in (0, 1):
            m0 = e.top
            h, hn, hnew = e.ldw(C.H + 2 * k + l), e.ldw(C.H + 2 * k + l, nxt=True), e.ldw(C.HN + 2 * k + l)
            t = e.op(VX_OP_SUB, hn, h)
            push(e.op(VX_OP_MUL, t, NOT33), VX_AIR_TRANSITION)           # H' = H unless the row is the hand-over
            iv = tmp()
            e.ldi(iv, (IVP[k] >> (32 * l)) & 0xFFFFFFFF)
            u = e.op(VX_OP_SUB, iv, hnew)
            e.op(VX_OP_MUL, u, Fr, u)
            e.op(VX_OP_ADD, u, hnew, u)                                  # F IVP + (1 - F) HN
            e.op(VX_OP_SUB, hn, u, u)
            push(e.op(VX_OP_MUL, u, S33), VX_AIR_TRANSITION)
            push(e.op(VX_OP_SUB, h, iv), VX_AIR_FIRST_ROW)
            e.release(m0)
    for m in range(8):
        m0 = e.top
        d, dn, hnew = e.ldw(C.D + m), e.ldw(C.D + m, nxt=True), e.ldw(C.HN + m)
        u = e.op(VX_OP_SUB, hnew, d)
        e.op(VX_OP_MUL, u, Fr, u)
        e.op(VX_OP_MUL, u, S33, u)
        e.op(VX_OP_ADD, u, d, u)
        push(e.op(VX_OP_SUB, dn, u), VX_AIR_TRANSITION)                  # D' = D + s33 F (HN - D)
        push(d, VX_AIR_FIRST_ROW)
        pi = tmp()
        e.ins(VX_OP_LDP, pi, m)
        push(e.op(VX_OP_SUB, d, pi), VX_AIR_LAST_ROW)
        e.release(m0)
    for col in list(range(C.M, C.M + 32)) + [C.T, C.F, C.TB]:
        m0 = e.top
        t = e.op(VX_OP_SUB, e.ldw(col, nxt=True), e.ldw(col))
        push(e.op(VX_OP_MUL, t, NOT33), VX_AIR_TRANSITION)
        e.release(m0)
    m0 = e.top
    c128 = tmp()
    e.ldi(c128, 128)
    t = e.op(VX_OP_SUB, e.ldw(C.T), e.ldw(C.TB))
    e.op(VX_OP_SUB, t, c128, t)
    nf = e.op(VX_OP_SUB, ONE, Fr)
    push(e.op(VX_OP_MUL, t, nf), VX_AIR_ALL_ROWS)                        # (1 - F)(T - TB - 128) = 0
    u = e.op(VX_OP_MUL, nf, e.ldw(C.T))
    e.op(VX_OP_SUB, e.ldw(C.TB, nxt=True), u, u)
    push(e.op(VX_OP_MUL, u, S33), VX_AIR_TRANSITION)                     # hand-over: TB' = (1 - F) T
    push(e.ldw(C.TB), VX_AIR_FIRST_ROW)
    e.release(m0)
    # ---- row type: cyclic shift of the one-hot ----
    for i in range(PERIOD):
        m0 = e.top
        push(e.op(VX_OP_SUB, e.ldw(C.SEL + i, nxt=True), e.ldw(C.SEL + (i - 1) % PERIOD)), VX_AIR_TRANSITION)
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
        w = tmp()
        limb([C.BY + 4 * l + q for q in range(4)], w)
        e.op(VX_OP_MUL, w, sel16, w)
        for t16 in range(16):
            m2 = e.top
            v = e.op(VX_OP_MUL, e.ldw(C.M + 2 * t16 + l), e.ldw(C.SEL + ROW_G0 + t16))
            e.op(VX_OP_SUB, w, v, w)
            e.release(m2)
        push(w, VX_AIR_ALL_ROWS)
        e.release(m1)
    e.release(m0)
    # ---- the XOR table, two halves: TA, TB and TC = TA ^ TB are tied to 16 bit columns — every row is a valid triple by construction ----
    for k in range(NTAB):
        m0 = e.top
        sa, sb, sc = tmp(), tmp(), tmp()
        for i in range(7, -1, -1):
            m1 = e.top
            a, b = e.ldw(C.tab(k, "BITS", i)), e.ldw(C.tab(k, "BITS", 8 + i))
            x = e.op(VX_OP_MUL, a, b)
            e.op(VX_OP_ADD, x, x, x)
            sx = e.op(VX_OP_ADD, a, b)
            e.op(VX_OP_SUB, sx, x, sx)                                   # a ^ b
            for acc, bit in ((sa, a), (sb, b), (sc, sx)):
                if i == 7:
                    e.op(VX_OP_ADD, bit, ZERO, acc)
                else:
                    e.op(VX_OP_ADD, acc, acc, acc)
                    e.op(VX_OP_ADD, acc, bit, acc)
            e.release(m1)
        push(e.op(VX_OP_SUB, e.ldw(C.tab(k, "TA")), sa), VX_AIR_ALL_ROWS)
        push(e.op(VX_OP_SUB, e.ldw(C.tab(k, "TB")), sb), VX_AIR_ALL_ROWS)
        push(e.op(VX_OP_SUB, e.ldw(C.tab(k, "TC")), sc), VX_AIR_ALL_ROWS)
        e.release(m0)
    # ---- the lookups: every triple enters as a + beta b + beta^2 c ----
    m0 = e.top
    acc, accn = e.ldw(C.AUX_ACC), e.ldw(C.AUX_ACC, nxt=True)
    step = e.op(VX_OP_SUB, accn, acc)

    def gamma_minus(tp):
        a, b, c = tp
        t = e.op(VX_OP_MUL, e.ldw(c), BETA2)
        if b is not None:
            u = e.op(VX_OP_MUL, e.ldw(b), BETA)
            e.op(VX_OP_ADD, t, u, t)
        e.op(VX_OP_ADD, t, e.ldw(a), t)
        return e.op(VX_OP_SUB, GAMMA, t, t)

    tps = tuples()
    for q in range(C.NPAIR):
        m1 = e.top
        g0 = tmp()
        e.op(VX_OP_ADD, gamma_minus(tps[2 * q]), ZERO, g0)
        e.release(g0 + 1)
        g1 = tmp()
        e.op(VX_OP_ADD, gamma_minus(tps[2 * q + 1]), ZERO, g1)
        e.release(g1 + 1)
        h = e.ldw(C.AUX_H + q)
        e.op(VX_OP_SUB, step, h, step)
        t = e.op(VX_OP_MUL, g0, g1)
        e.op(VX_OP_MUL, t, h, t)
        e.op(VX_OP_SUB, t, g0, t)
        e.op(VX_OP_SUB, t, g1, t)
        push(t, VX_AIR_ALL_ROWS)                                         # h (g - t0)(g - t1) = (g - t0) + (g - t1)
        e.release(m1)
    for k in range(NTAB):
        m1 = e.top
        gt = gamma_minus((C.tab(k, "TA"), C.tab(k, "TB"), C.tab(k, "TC")))
        ht = e.ldw(C.AUX_HT + k)
        e.op(VX_OP_ADD, step, ht, step)
        t = e.op(VX_OP_MUL, ht, gt)
        push(e.op(VX_OP_SUB, t, e.ldw(C.tab(k, "MULT"))), VX_AIR_ALL_ROWS)   # ht_k (g - table triple) = mult_k
        e.release(m1)
    push(step, VX_AIR_TRANSITION)                                        # acc' = acc + sum h - ht_0 - ht_1
    push(acc, VX_AIR_FIRST_ROW)
    push(acc, VX_AIR_LAST_ROW)
    e.release(m0)
    e.ins(VX_OP_END)
    return e.w, npush


# ---------------------------------------------------------------------------------------------------------------------
def _rotr(x, r):
    return ((x >> r) | (x << (64 - r))) & MASK64


def _bytes8(x):
    return [(x >> (8 * j)) & 255 for j in range(8)]


def _g(a, b, c, d, x, y):
    """one G: -> dict of the slot's field values (64-bit words), carries, outputs"""
    k = [0] * 8

    def add(terms, ki):
        lo = sum(t & 0xFFFFFFFF for t in terms)
        k[ki] = lo >> 32
        hi = sum(t >> 32 for t in terms) + k[ki]
        k[ki + 1] = hi >> 32
        return (lo & 0xFFFFFFFF) | ((hi & 0xFFFFFFFF) << 32)

    a1 = add((a, b, x), K_A1)
    e1 = d ^ a1
    d1 = _rotr(e1, 32)
    c1 = add((c, d1), K_C1)
    f1 = b ^ c1
    b1 = _rotr(f1, 24)
    a2 = add((a1, b1, y), K_A2)
    e2 = d1 ^ a2
    d2 = _rotr(e2, 16)
    c2 = add((c1, d2), K_C2)
    f2 = b1 ^ c2
    b2 = _rotr(f2, 63)
    top = sum(((f2 >> (8 * j + 7)) & 1) << (8 * j) for j in range(8))     # TOP byte j = bit 7 of f2's byte j
    return {"BIN": b, "DIN": d, "A1": a1, "E1": e1, "C1": c1, "F1": f1, "A2": a2, "E2": e2, "C2": c2, "F2": f2, "TOP": top, "B2": b2,
            "AL": a, "CL": c, "K": k, "out": (a2, b2, c2, d2)}


def generate_trace(degree_bits: int, messages) -> tuple:
    """-> (trace [N][n] uint64, public inputs [8], digests of the messages completed inside the trace).  `messages`: byte strings hashed
    one after the other; the rows that remain keep hashing blocks of an endless zero-message (never final)."""
    C = Cols
    n = 1 << degree_bits
    assert degree_bits >= 16, "each half of the XOR table needs 32 768 rows before the last row"
    t = np.zeros((C.N, n), dtype=np.uint64)
    blocks = [blk for m in messages for blk in split_message(m)]
    nblocks = -(-n // PERIOD)
    digests, h, tb_prev, hn, dlatch = [], list(IVP), 0, [0] * 8, [0] * 4
    filler_t = 0
    lim2 = lambda w: (w & 0xFFFFFFFF, w >> 32)          # noqa: E731

    def put_limbs(col, words, row):
        for k, w in enumerate(words):
            t[col + 2 * k, row], t[col + 2 * k + 1, row] = lim2(w)

    def put_bytes(col, word, row):
        for j in range(8):
            t[col + j, row] = (word >> (8 * j)) & 255

    for b in range(nblocks):
        if b < len(blocks):
            m, tcount, fin = blocks[b]
        else:
            filler_t += 128
            m, tcount, fin = [0] * 16, (tb_prev + 128), 0
        base = b * PERIOD
        rows = range(base, min(n, base + PERIOD))
        r0, r1 = base, min(n, base + PERIOD)
        t[C.SEL + np.arange(r1 - r0), np.arange(r0, r1)] = 1
        hm = [x for w in h for x in lim2(w)]
        mm = [x for w in m for x in lim2(w)]
        dd = [x for w in dlatch for x in lim2(w)]
        t[C.H:C.H + 16, r0:r1] = np.array(hm, dtype=np.uint64)[:, None]
        t[C.M:C.M + 32, r0:r1] = np.array(mm, dtype=np.uint64)[:, None]
        t[C.D:C.D + 8, r0:r1] = np.array(dd, dtype=np.uint64)[:, None]
        t[C.T, r0:r1], t[C.F, r0:r1], t[C.TB, r0:r1] = tcount, fin, tb_prev
        v = list(h) + list(IV[:4]) + [IV[4] ^ tcount, IV[5], IV[6] ^ (MASK64 if fin else 0), IV[7]]
        # row 0: the initial work vector in the diagonal-output arrangement; the tuples stay valid XOR triples
        row = base
        if row < n:
            for j in range(NSLOT):
                a2, b2, c2, d2 = v[j], v[4 + (j + 1) % 4], v[8 + (j + 2) % 4], v[12 + (j + 3) % 4]
                e2 = _rotr(d2, 48)                       # d2 = e2 >>> 16
                e1 = _rotr(e2 ^ a2, 32)                  # tuple 3: e2_j = e1_{(j+4)%8} ^ a2_j
                for name, val in (("A2", a2), ("B2", b2), ("C2", c2), ("E2", e2), ("E1", e1), ("A1", e1), ("F2", c2)):
                    put_bytes(C.f(j, name), val, row)
            put_bytes(C.fin(0, "FA"), tcount, row)
            put_bytes(C.fin(0, "FB"), IV[4], row)
            put_bytes(C.fin(0, "FE"), tcount ^ IV[4], row)
            put_bytes(C.fin(0, "FO"), tcount ^ IV[4], row)
            put_limbs(C.HN, hn, row)
        # rows 1..24
        for s in range(ROW_G0, ROW_GLAST + 1):
            row = base + s
            if row >= n:
                break
            r, half = divmod(s - 1, 2)
            outs = []
            for i in range(NSLOT):
                if half == 0:
                    idx = (i, 4 + i, 8 + i, 12 + i)
                else:
                    idx = (i, 4 + (i + 1) % 4, 8 + (i + 2) % 4, 12 + (i + 3) % 4)
                x, y = m[_msg_index(s, i, 0)], m[_msg_index(s, i, 1)]
                gr = _g(v[idx[0]], v[idx[1]], v[idx[2]], v[idx[3]], x, y)
                # the slot's 12 byte fields are 96 contiguous columns, then AL(2), CL(2), K(8): two assignments per slot
                base_col = C.f(i, FIELDS[0])
                t[base_col:base_col + 96, row] = np.frombuffer(b"".join(gr[name].to_bytes(8, "little") for name in FIELDS), dtype=np.uint8)
                t[base_col + 96:base_col + 108, row] = [*lim2(gr["AL"]), *lim2(gr["CL"]), *gr["K"]]
                outs.append((idx, gr["out"]))
            for idx, o in outs:
                for q in range(4):
                    v[idx[q]] = o[q]
            if s <= 16:
                put_bytes(C.BY, m[s - 1], row)
        if base + 1 < n:                 # HN holds through the G rows
            t[C.HN:C.HN + 16, base + 1:min(n, base + ROW_GLAST + 1)] = np.array([x for w in hn for x in lim2(w)], dtype=np.uint64)[:, None]
        # rows 25, 26: finalisation, four words per row
        for r in range(NFINROW):
            row = base + ROW_FIN0 + r
            if row >= n:
                break
            put_limbs(C.V, v, row)
            put_limbs(C.HN, hn, row)
            hn = list(hn)
            for g in range(NFING):
                k = 4 * r + g
                put_bytes(C.fin(g, "FA"), v[k], row)
                put_bytes(C.fin(g, "FB"), v[k + 8], row)
                put_bytes(C.fin(g, "FE"), v[k] ^ v[k + 8], row)
                put_bytes(C.fin(g, "FH"), h[k], row)
                put_bytes(C.fin(g, "FO"), v[k] ^ v[k + 8] ^ h[k], row)
                hn[k] = v[k] ^ v[k + 8] ^ h[k]
        row = base + ROW_HAND
        if row < n:
            put_limbs(C.HN, hn, row)
        if base + PERIOD <= n:                           # the block completed inside the trace
            if fin:
                dlatch = hn[:4]
                digests.append(b"".join(w.to_bytes(8, "little") for w in hn[:4]))
                h, tb_prev = list(IVP), 0
            else:
                h, tb_prev = list(hn), tcount
    rows = np.arange(n)
    mult = np.zeros(65536, dtype=np.int64)
    for a, b, _ in tuples():
        idx = t[a, :n - 1].astype(np.int64) * 256 + (t[b, :n - 1].astype(np.int64) if b is not None else 0)
        mult += np.bincount(idx, minlength=65536)
    for k in range(NTAB):                                     # half k, row i (repeating every 32 768 rows): a = 128 k + (i >> 8) % 128, b = i & 255
        ta = np.uint64(128 * k) + ((rows >> 8) & 127).astype(np.uint64)
        tb = (rows & 255).astype(np.uint64)
        t[C.tab(k, "TA")], t[C.tab(k, "TB")], t[C.tab(k, "TC")] = ta, tb, ta ^ tb
        for i in range(8):
            t[C.tab(k, "BITS", i)] = (ta >> np.uint64(i)) & np.uint64(1)
            t[C.tab(k, "BITS", 8 + i)] = (tb >> np.uint64(i)) & np.uint64(1)
        t[C.tab(k, "MULT"), :TAB_ROWS] = mult[TAB_ROWS * k:TAB_ROWS * (k + 1)].astype(np.uint64)
    pis = np.array([x for w in dlatch for x in lim2(w)], dtype=np.uint64)
    _ = filler_t
    return t, pis, digests


def aux_columns(trace, chal):
    """second-round columns [116 pair helpers, ht of each half table, acc] for the challenges [gamma, beta]"""
    C = Cols
    n = trace.shape[1]
    g, beta = int(chal[0]) % P, int(chal[1]) % P
    beta2 = beta * beta % P
    ab = np.arange(65536, dtype=np.uint64)
    ta, tb = ab >> np.uint64(8), ab & np.uint64(255)
    gam = np.full(65536, g, dtype=np.uint64)

    def triple(a, b, c):
        v = hf.mulmod(c % np.uint64(P), np.full(c.shape, beta2, dtype=np.uint64))
        if b is not None:
            v = hf.addmod(v, hf.mulmod(b % np.uint64(P), np.full(b.shape, beta, dtype=np.uint64)))
        return hf.addmod(v, a % np.uint64(P))

    tab = hf.invmod(hf.submod(gam, triple(ta, tb, ta ^ tb)))            # 1 / (gamma - triple) of every table entry, by a * 256 + b
    out = np.zeros((C.NAUX, n), dtype=np.uint64)
    step = np.zeros(n, dtype=np.uint64)
    inv = []
    for a, b, c in tuples():
        av, cv = trace[a], trace[c]
        bv = trace[b] if b is not None else np.zeros(n, dtype=np.uint64)
        ok = (av < 256) & (bv < 256) & (cv == (av ^ bv))
        if ok.all():
            inv.append(tab[(av * np.uint64(256) + bv).astype(np.int64)])
        else:                                                          # corrupted traces of the tests: the honest helper of a triple outside the table
            iv = tab[((av & np.uint64(255)) * np.uint64(256) + (bv & np.uint64(255))).astype(np.int64)].copy()
            bad = np.nonzero(~ok)[0]
            tv = triple(av[bad], bv[bad] if b is not None else None, cv[bad])
            iv[bad] = hf.invmod(hf.submod(np.full(bad.size, g, dtype=np.uint64), tv))
            inv.append(iv)
    for q in range(C.NPAIR):
        h = hf.addmod(inv[2 * q], inv[2 * q + 1])
        out[q] = h
        step = hf.addmod(step, h)
    for k in range(NTAB):
        tv, tbv, tcv = trace[C.tab(k, "TA")], trace[C.tab(k, "TB")], trace[C.tab(k, "TC")]
        okt = (tv < 256) & (tbv < 256) & (tcv == (tv ^ tbv))
        it = tab[((tv & np.uint64(255)) * np.uint64(256) + (tbv & np.uint64(255))).astype(np.int64)].copy()
        if not okt.all():
            bad = np.nonzero(~okt)[0]
            it[bad] = hf.invmod(hf.submod(np.full(bad.size, g, dtype=np.uint64), triple(tv[bad], tbv[bad], tcv[bad])))
        ht = hf.mulmod(trace[C.tab(k, "MULT")] % np.uint64(P), it)
        out[C.NPAIR + k] = ht
        step = hf.submod(step, ht)
    out[C.NPAIR + NTAB], _ = hf.exclusive_prefix_sum(step)
    return out


def aux_program():
    """the GPU form of `aux_columns` (vx_stark_aux_columns): 116 pair helpers + the two table terms as fractions, one running sum that closes at 0"""
    from . import AuxProgram
    C = Cols
    e = _Emit(scratch=40)
    GAMMA, BETA, BETA2 = 63, 62, 61
    e.ins(VX_OP_LDCH, GAMMA, 0)
    e.ins(VX_OP_LDCH, BETA, 1)
    e.op(VX_OP_MUL, BETA, BETA, BETA2)

    def gamma_minus(tp, dst):
        a, b, c = tp
        m = e.top
        t = e.op(VX_OP_MUL, e.ldw(c), BETA2)
        if b is not None:
            e.op(VX_OP_ADD, t, e.op(VX_OP_MUL, e.ldw(b), BETA), t)
        e.op(VX_OP_ADD, t, e.ldw(a), t)
        e.op(VX_OP_SUB, GAMMA, t, dst)
        e.release(m)

    tps = tuples()
    for q in range(C.NPAIR):
        m0 = e.top
        g0, g1 = e.tmp(), e.tmp()
        gamma_minus(tps[2 * q], g0)
        gamma_minus(tps[2 * q + 1], g1)
        e.push(e.op(VX_OP_ADD, g0, g1), 0)
        e.push(e.op(VX_OP_MUL, g0, g1), 0)
        e.release(m0)
    for k in range(NTAB):
        m0 = e.top
        gt = e.tmp()
        gamma_minus((C.tab(k, "TA"), C.tab(k, "TB"), C.tab(k, "TC")), gt)
        e.push(e.ldw(C.tab(k, "MULT")), 0)
        e.push(gt, 0)
        e.release(m0)
    e.ins(VX_OP_END)
    return AuxProgram(C.N, 2, e.w, C.NPAIR + NTAB, [[1] * C.NPAIR + [-1] * NTAB], api_sums=())


def make_stark(degree_bits: int, **cfg) -> Stark:
    assert degree_bits >= 16
    prog, _ = build_program()
    cfg.setdefault("rate_bits", 1)
    st = Stark(degree_bits, Cols.N, 8, prog, constraint_degree=3, num_aux_columns=Cols.NAUX, num_aux_challenges=2, aux_fn=aux_columns, **cfg)
    st.aux_program = aux_program()
    return st


def reference_digests(messages):
    return [hashlib.blake2b(m, digest_size=32).digest() for m in messages]
