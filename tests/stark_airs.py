"""Toy AIRs as constraint programs (vxprover.h VX_OP_*, VX_AIR_*) + their traces, shared by the CPU and GPU STARK tests."""
import numpy as np

import vectorx_amd as vx

P = 0xFFFFFFFF00000001
I = vx.vx_ins


def fibonacci(degree_bits, x0=0, x1=1, **cfg):
    """plonky2's starky/src/fibonacci_stark.rs: columns [x0, x1], public inputs [x0_0, x1_0, x1_last]:
         first row:   x0 = pi[0], x1 = pi[1]        last row: x1 = pi[2]
         transition:  x0' = x1,  x1' = x0 + x1       (constraint degree 2 -> quotient_degree_factor 1)"""
    prog = [I(vx.VX_OP_LDW, 0, 0), I(vx.VX_OP_LDW, 1, 1), I(vx.VX_OP_LDN, 2, 0), I(vx.VX_OP_LDN, 3, 1),
            I(vx.VX_OP_LDP, 4, 0), I(vx.VX_OP_LDP, 5, 1), I(vx.VX_OP_LDP, 6, 2),
            I(vx.VX_OP_SUB, 7, 0, 4), I(vx.VX_OP_PUSH, 0, 7, vx.VX_AIR_FIRST_ROW),
            I(vx.VX_OP_SUB, 7, 1, 5), I(vx.VX_OP_PUSH, 0, 7, vx.VX_AIR_FIRST_ROW),
            I(vx.VX_OP_SUB, 7, 1, 6), I(vx.VX_OP_PUSH, 0, 7, vx.VX_AIR_LAST_ROW),
            I(vx.VX_OP_SUB, 7, 2, 1), I(vx.VX_OP_PUSH, 0, 7, vx.VX_AIR_TRANSITION),
            I(vx.VX_OP_ADD, 8, 0, 1), I(vx.VX_OP_SUB, 7, 3, 8), I(vx.VX_OP_PUSH, 0, 7, vx.VX_AIR_TRANSITION),
            I(vx.VX_OP_END)]
    n = 1 << degree_bits
    t = np.zeros((2, n), dtype=np.uint64)
    a, b = x0 % P, x1 % P
    for i in range(n):
        t[0, i], t[1, i] = a, b
        a, b = b, (a + b) % P
    stark = vx.Stark(degree_bits, 2, 3, prog, constraint_degree=2, **cfg)
    return stark, t, np.array([x0 % P, x1 % P, int(t[1, n - 1])], dtype=np.uint64)


def cubic(degree_bits, seed=3, **cfg):
    """A degree-3 AIR (quotient_degree_factor 2, evaluated on two cosets): one column, y' = y^3 + c with c a program
    immediate; first row y = pi[0], last row y = pi[1]."""
    c = 0x1234567
    prog = [I(vx.VX_OP_LDW, 0, 0), I(vx.VX_OP_LDN, 1, 0), I(vx.VX_OP_LDI, 2), c, I(vx.VX_OP_LDP, 3, 0), I(vx.VX_OP_LDP, 4, 1),
            I(vx.VX_OP_SUB, 5, 0, 3), I(vx.VX_OP_PUSH, 0, 5, vx.VX_AIR_FIRST_ROW),
            I(vx.VX_OP_MUL, 6, 0, 0), I(vx.VX_OP_MUL, 6, 6, 0), I(vx.VX_OP_ADD, 6, 6, 2), I(vx.VX_OP_SUB, 6, 1, 6),
            I(vx.VX_OP_PUSH, 0, 6, vx.VX_AIR_TRANSITION),
            I(vx.VX_OP_SUB, 5, 0, 4), I(vx.VX_OP_PUSH, 0, 5, vx.VX_AIR_LAST_ROW),
            I(vx.VX_OP_END)]
    n = 1 << degree_bits
    t = np.zeros((1, n), dtype=np.uint64)
    y = seed % P
    for i in range(n):
        t[0, i] = y
        y = (pow(y, 3, P) + c) % P
    stark = vx.Stark(degree_bits, 1, 2, prog, constraint_degree=3, **cfg)
    return stark, t, np.array([seed % P, int(t[0, n - 1])], dtype=np.uint64)


def mulmod(a, b):
    """element-wise a * b mod p on uint64 arrays (32-bit limbs; 2^64 = 2^32 - 1, 2^96 = -1 mod p)"""
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    M = np.uint64(0xFFFFFFFF)
    S = np.uint64(32)
    a0, a1, b0, b1 = a & M, a >> S, b & M, b >> S
    p00, p01, p10, p11 = a0 * b0, a0 * b1, a1 * b0, a1 * b1
    mid = p01 + (p00 >> S)
    mid2 = p10 + (mid & M)
    lo = ((mid2 & M) << S) | (p00 & M)
    hi = p11 + (mid >> S) + (mid2 >> S)
    hh, hl = hi >> S, hi & M
    with np.errstate(over="ignore"):
        t0 = lo - hh
        t0 = np.where(lo < hh, t0 - M, t0)          # borrow: + p = - EPS (mod 2^64)
        t1 = hl * M
        r = t0 + t1
        r = np.where(r < t1, r + M, r)              # carry: 2^64 = EPS
        r = np.where(r >= np.uint64(P), r - np.uint64(P), r)
    return r


def mulchain(degree_bits, groups=2, seed=5, **cfg):
    """A wider degree-3 AIR (the constraint degree of Curta's chips): `groups` blocks of four columns [a, b, c, e] with
         transition: a' = a + 1,  b' = 3 b,  c' = c + a        all rows: e = a b c
         first row (block 0): a = pi[0]                        last row (block 0): c = pi[1]
    The trace is closed-form per column, so it is generated vectorised at any size (tools/stark_bench.py uses 2^18+ rows)."""
    n = 1 << degree_bits
    ncols = 4 * groups
    prog = [I(vx.VX_OP_LDI, 60), 1, I(vx.VX_OP_LDI, 61), 3]
    for g in range(groups):
        a, b, c, e = 4 * g, 4 * g + 1, 4 * g + 2, 4 * g + 3
        prog += [I(vx.VX_OP_LDW, 0, a), I(vx.VX_OP_LDW, 1, b), I(vx.VX_OP_LDW, 2, c), I(vx.VX_OP_LDW, 3, e),
                 I(vx.VX_OP_LDN, 4, a), I(vx.VX_OP_LDN, 5, b), I(vx.VX_OP_LDN, 6, c),
                 I(vx.VX_OP_ADD, 7, 0, 60), I(vx.VX_OP_SUB, 7, 4, 7), I(vx.VX_OP_PUSH, 0, 7, vx.VX_AIR_TRANSITION),
                 I(vx.VX_OP_MUL, 7, 1, 61), I(vx.VX_OP_SUB, 7, 5, 7), I(vx.VX_OP_PUSH, 0, 7, vx.VX_AIR_TRANSITION),
                 I(vx.VX_OP_ADD, 7, 2, 0), I(vx.VX_OP_SUB, 7, 6, 7), I(vx.VX_OP_PUSH, 0, 7, vx.VX_AIR_TRANSITION),
                 I(vx.VX_OP_MUL, 7, 0, 1), I(vx.VX_OP_MUL, 7, 7, 2), I(vx.VX_OP_SUB, 7, 3, 7), I(vx.VX_OP_PUSH, 0, 7, vx.VX_AIR_ALL_ROWS)]
        if g == 0:
            prog += [I(vx.VX_OP_LDP, 8, 0), I(vx.VX_OP_SUB, 8, 0, 8), I(vx.VX_OP_PUSH, 0, 8, vx.VX_AIR_FIRST_ROW),
                     I(vx.VX_OP_LDP, 8, 1), I(vx.VX_OP_SUB, 8, 2, 8), I(vx.VX_OP_PUSH, 0, 8, vx.VX_AIR_LAST_ROW)]
    prog.append(I(vx.VX_OP_END))
    # 3^i for i < n by doubling
    pw = np.ones(1, dtype=np.uint64)
    step = np.uint64(3)
    while pw.size < n:
        pw = np.concatenate([pw, mulmod(pw, np.full(pw.size, step, dtype=np.uint64))])
        step = mulmod(np.array([step]), np.array([step]))[0]
    t = np.zeros((ncols, n), dtype=np.uint64)
    idx = np.arange(n, dtype=np.uint64)
    for g in range(groups):
        a0, b0, c0 = (seed * 7 + 11 * g) % (1 << 20), (seed * 1000003 + 17 * g + 1) % P, (seed + 3 * g) % (1 << 20)
        a = idx + np.uint64(a0)                                        # < 2^21 + n: no reduction
        b = mulmod(pw, np.full(n, b0, dtype=np.uint64))
        c = np.uint64(c0) + np.concatenate([[np.uint64(0)], np.cumsum(a[:-1], dtype=np.uint64)])   # < 2^45
        t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3] = a, b, c, mulmod(mulmod(a, b), c)
    stark = vx.Stark(degree_bits, ncols, 2, prog, constraint_degree=3, **cfg)
    return stark, t, np.array([int(t[0, 0]), int(t[2, n - 1])], dtype=np.uint64)


def invmod(a):
    """element-wise inverse mod p (Fermat) on a uint64 array"""
    a = np.asarray(a, dtype=np.uint64)
    r = np.ones_like(a)
    base = a.copy()
    e = P - 2
    while e:
        if e & 1:
            r = mulmod(r, base)
        base = mulmod(base, base)
        e >>= 1
    return r


def addmod(a, b):
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    with np.errstate(over="ignore"):
        s = a + b
        s = np.where((s < a) | (s >= np.uint64(P)), s - np.uint64(P), s)
    return s


def submod(a, b):
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return np.where(a >= b, a - b, a - b + np.uint64(P))


def logup(degree_bits, table_bits=4, seed=11, **cfg):
    """A TWO-ROUND AIR: a log-derivative lookup (the argument behind Curta's lookup tables / bus, and the role starky's
    permutation Z columns play).  Trace columns [v, t, m]: every v is an entry of the table column t (a table of
    2^table_bits values repeated down the column), m = how often the row's t is looked up; after the trace commitment ONE
    challenge gamma is drawn and the prover commits the running sum
         acc_0 = 0,   acc_{i+1} = acc_i + 1/(gamma - v_i) - m_i/(gamma - t_i)   over the rows 0 .. n-2 (the last row is inert:
       first- and last-row constraints are multiplied by a Lagrange selector of degree n - 1, so their own degree must stay
       below the constraint degree — closing the sum in the last row with the degree-3 term would not fit):
       transition  (acc' - acc)(gamma - v)(gamma - t) - (gamma - t) + m (gamma - v) = 0           (degree 3)
       first row   acc = 0             last row   acc = 0
    Public inputs: [v_0] (first row: v = pi[0]) so that the proof is tied to a statement."""
    n = 1 << degree_bits
    T = 1 << table_bits
    rng = np.random.default_rng(seed)
    table = rng.integers(1, 1 << 40, size=T, dtype=np.uint64)
    t = np.tile(table, n // T) if n >= T else table[:n].copy()
    picks = rng.integers(0, min(T, n), size=n)
    v = table[picks]
    # multiplicities: the count of each table entry is placed on the FIRST occurrence of that entry in column t
    m = np.zeros(n, dtype=np.uint64)
    m[:min(T, n)] = np.bincount(picks[:n - 1], minlength=min(T, n)).astype(np.uint64)   # the last row's v is not looked up
    trace = np.stack([v, t, m]).astype(np.uint64)
    V, Tt, M, ACC = 0, 1, 2, 3
    prog = [I(vx.VX_OP_LDCH, 10, 0),
            I(vx.VX_OP_LDW, 0, V), I(vx.VX_OP_LDW, 1, Tt), I(vx.VX_OP_LDW, 2, M), I(vx.VX_OP_LDW, 3, ACC), I(vx.VX_OP_LDN, 4, ACC),
            I(vx.VX_OP_SUB, 5, 10, 0),                    # gamma - v
            I(vx.VX_OP_SUB, 6, 10, 1),                    # gamma - t
            I(vx.VX_OP_MUL, 7, 5, 6),                     # (gamma - v)(gamma - t)
            I(vx.VX_OP_MUL, 8, 2, 5),                     # m (gamma - v)
            I(vx.VX_OP_SUB, 9, 4, 3), I(vx.VX_OP_MUL, 9, 9, 7), I(vx.VX_OP_SUB, 9, 9, 6), I(vx.VX_OP_ADD, 9, 9, 8),
            I(vx.VX_OP_PUSH, 0, 9, vx.VX_AIR_TRANSITION),
            I(vx.VX_OP_PUSH, 0, 3, vx.VX_AIR_FIRST_ROW),
            I(vx.VX_OP_PUSH, 0, 3, vx.VX_AIR_LAST_ROW),
            I(vx.VX_OP_LDP, 12, 0), I(vx.VX_OP_SUB, 12, 0, 12), I(vx.VX_OP_PUSH, 0, 12, vx.VX_AIR_FIRST_ROW),
            I(vx.VX_OP_END)]

    def aux_fn(tr, chal):
        g = np.full(n, chal[0], dtype=np.uint64)
        term = submod(invmod(submod(g, tr[0] % np.uint64(P))), mulmod(tr[2] % np.uint64(P), invmod(submod(g, tr[1] % np.uint64(P)))))
        acc = np.zeros(n, dtype=np.uint64)
        run = 0
        tl = [int(x) for x in term]
        for i in range(1, n):                     # exclusive prefix sum mod p
            run = (run + tl[i - 1]) % P
            acc[i] = run
        return acc.reshape(1, n)

    stark = vx.Stark(degree_bits, 3, 1, prog, constraint_degree=3, num_aux_columns=1, num_aux_challenges=1, aux_fn=aux_fn, **cfg)
    return stark, trace, np.array([int(v[0])], dtype=np.uint64)
