"""Goldilocks arithmetic on numpy uint64 arrays — host-side helpers of the trace / second-round column generators (caller-side witness
generation; nothing on the proving path uses them).  p = 2^64 - 2^32 + 1: 2^64 = 2^32 - 1 and 2^96 = -1 (mod p)."""
from __future__ import annotations

import numpy as np

P = 0xFFFFFFFF00000001
_P = np.uint64(P)
_M = np.uint64(0xFFFFFFFF)
_S = np.uint64(32)


def addmod(a, b):
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    with np.errstate(over="ignore"):
        s = a + b
        s = np.where(s < a, s + _M, s)          # wrapped past 2^64: + (2^64 mod p)
        return np.where(s >= _P, s - _P, s)


def submod(a, b):
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    with np.errstate(over="ignore"):
        d = a - b
        return np.where(a < b, d + _P, d)


def mulmod(a, b):
    """element-wise a * b mod p (canonical operands) through 32-bit halves"""
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    a0, a1, b0, b1 = a & _M, a >> _S, b & _M, b >> _S
    with np.errstate(over="ignore"):
        p00, p01, p10, p11 = a0 * b0, a0 * b1, a1 * b0, a1 * b1
        mid = p01 + (p00 >> _S)
        mid2 = p10 + (mid & _M)
        lo = ((mid2 & _M) << _S) | (p00 & _M)
        hi = p11 + (mid >> _S) + (mid2 >> _S)
        hh, hl = hi >> _S, hi & _M
        t0 = lo - hh
        t0 = np.where(lo < hh, t0 - _M, t0)     # borrow: + p = - (2^32 - 1) (mod 2^64)
        t1 = hl * _M
        r = t0 + t1
        r = np.where(r < t1, r + _M, r)
        return np.where(r >= _P, r - _P, r)


def powmod(a, e: int):
    a = np.asarray(a, dtype=np.uint64)
    r = np.ones_like(a)
    base = a
    while e:
        if e & 1:
            r = mulmod(r, base)
        base = mulmod(base, base)
        e >>= 1
    return r


def invmod(a):
    """element-wise inverse (0 -> 0)"""
    return powmod(a, P - 2)


def exclusive_prefix_sum(v):
    """acc[i] = v[0] + .. + v[i-1] mod p (acc[0] = 0), and the total — for arrays of up to 2^31 elements"""
    v = np.asarray(v, dtype=np.uint64)
    lo = np.cumsum(v & _M, dtype=np.uint64)          # < 2^32 * 2^31: no wrap
    hi = np.cumsum(v >> _S, dtype=np.uint64)
    two32 = np.full(1, 1 << 32, dtype=np.uint64)
    incl = addmod(mulmod(hi % _P, two32), lo % _P)
    acc = np.empty_like(incl)
    acc[0] = 0
    acc[1:] = incl[:-1]
    return acc, int(incl[-1])
