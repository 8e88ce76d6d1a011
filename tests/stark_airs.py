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
