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
