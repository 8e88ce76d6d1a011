#!/usr/bin/env python3
"""Derive the "fast partial round" form of Poseidon-Goldilocks (width 12, 22 partial rounds).

plonky2 v0.2.0 (plonky2/src/hash/poseidon_goldilocks.rs) ships FAST_PARTIAL_FIRST_ROUND_CONSTANT,
FAST_PARTIAL_ROUND_CONSTANTS, FAST_PARTIAL_ROUND_INITIAL_MATRIX, FAST_PARTIAL_ROUND_VS and
FAST_PARTIAL_ROUND_W_HATS as literals; the upstream source is not available here, so this script
re-derives an equivalent set from the round constants and the MDS matrix alone (Poseidon paper,
appendix B) and PROVES the equivalence numerically against the naive permutation (known-answer vectors
of SURVEY.md B.2 plus random states).  Only the permutation's input/output behaviour matters — it is
bit-identical to the naive form — so whether the literals equal upstream's is irrelevant.

Derivation.  Partial round r (r = 0..21):  s <- M * S(s + c_r),  S = x^7 on lane 0 only.
 (1) constants: for r = 21..1:  e = M^-1 c'_r;  k_{r-1} = e[0];  c'_{r-1} = c_{r-1} + (0, e[1:])   (c'_21 = c_21)
     -> one full vector c'_0 before the first S, then a scalar k_r added to lane 0 AFTER the S-box of
        round r (k_21 = 0).
 (2) matrices: M_21 = M;  M_r = N''_r N'_r with N'_r = diag(1, Mhat_r) (Mhat_r = M_r[1:,1:]) and the sparse
     N''_r = [[m00, w_hat^T], [v, I]],  w_hat^T = M_r[0,1:] Mhat_r^-1,  v = M_r[1:,0];  N'_r commutes with S, so
     M_{r-1} = N'_r M.   The dense part that is left over is N'_0 (the "initial matrix"), applied once.
 Result:  x = N'_0 (s + c'_0);  for r in 0..21: x0 = x0^7 + k_r;  x = N''_r x.

Writes oracle/poseidon_fast_constants.h (the oracle's permutation; the product's kernels use integer-power blocks built at
compile time in poseidon.hip.h and no longer read these tables).   --check verifies the committed header.
"""
import random
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from gen_poseidon_constants import P, round_constants  # noqa: E402

CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
DIAG = [8] + [0] * 11
W = 12
N_PARTIAL = 22


def mds_matrix():
    # out[r] = sum_i CIRC[i] * v[(i + r) % 12] + DIAG[r] v[r]   =>  M[r][(i + r) % 12] += CIRC[i]
    M = [[0] * W for _ in range(W)]
    for r in range(W):
        for i in range(W):
            M[r][(i + r) % W] = (M[r][(i + r) % W] + CIRC[i]) % P
        M[r][r] = (M[r][r] + DIAG[r]) % P
    return M


def matmul(A, B):
    n, m, k = len(A), len(B[0]), len(B)
    return [[sum(A[i][t] * B[t][j] for t in range(k)) % P for j in range(m)] for i in range(n)]


def matvec(A, v):
    return [sum(a * b for a, b in zip(row, v)) % P for row in A]


def inverse(A):
    n = len(A)
    a = [row[:] + [int(i == j) for j in range(n)] for i, row in enumerate(A)]
    for c in range(n):
        piv = next(r for r in range(c, n) if a[r][c] % P)
        a[c], a[piv] = a[piv], a[c]
        inv = pow(a[c][c], P - 2, P)
        a[c] = [x * inv % P for x in a[c]]
        for r in range(n):
            if r != c and a[r][c]:
                f = a[r][c]
                a[r] = [(x - f * y) % P for x, y in zip(a[r], a[c])]
    return [row[n:] for row in a]


def derive():
    rc = round_constants()
    M = mds_matrix()
    Minv = inverse(M)
    c = [rc[12 * (4 + r): 12 * (4 + r) + 12] for r in range(N_PARTIAL)]
    # (1) constants
    k = [0] * N_PARTIAL
    cp = c[N_PARTIAL - 1][:]
    for r in range(N_PARTIAL - 1, 0, -1):
        e = matvec(Minv, cp)
        k[r - 1] = e[0]
        cp = [c[r - 1][0]] + [(c[r - 1][i] + e[i]) % P for i in range(1, W)]
    first = cp
    # (2) matrices
    w_hats, vs = [None] * N_PARTIAL, [None] * N_PARTIAL
    Mr = M
    init = None
    for r in range(N_PARTIAL - 1, -1, -1):
        Mhat = [row[1:] for row in Mr[1:]]
        Mhat_inv = inverse(Mhat)
        row0 = [Mr[0][1:]]
        w_hats[r] = matmul(row0, Mhat_inv)[0]
        vs[r] = [Mr[i][0] for i in range(1, W)]
        assert Mr[0][0] == M[0][0]
        Np = [[int(i == j) if (i == 0 or j == 0) else Mhat[i - 1][j - 1] for j in range(W)] for i in range(W)]
        init = Mhat
        Mr = matmul(Np, M)
    return rc, M, first, k, init, w_hats, vs


def sbox(x):
    return pow(x, 7, P)


def permute_naive(s, rc, M):
    s = s[:]
    for r in range(30):
        s = [(x + rc[12 * r + i]) % P for i, x in enumerate(s)]
        if r < 4 or r >= 26:
            s = [sbox(x) for x in s]
        else:
            s[0] = sbox(s[0])
        s = matvec(M, s)
    return s


def permute_fast(s, rc, M, first, k, init, w_hats, vs):
    s = s[:]
    for r in range(4):
        s = [sbox((x + rc[12 * r + i]) % P) for i, x in enumerate(s)]
        s = matvec(M, s)
    s = [(x + f) % P for x, f in zip(s, first)]
    s = [s[0]] + matvec(init, s[1:])
    m00 = M[0][0]
    for r in range(N_PARTIAL):
        s0 = (sbox(s[0]) + k[r]) % P
        d = (m00 * s0 + sum(a * b for a, b in zip(w_hats[r], s[1:]))) % P
        s = [d] + [(s[i] + s0 * vs[r][i - 1]) % P for i in range(1, W)]
    for r in range(26, 30):
        s = [sbox((x + rc[12 * r + i]) % P) for i, x in enumerate(s)]
        s = matvec(M, s)
    return s


def header_text(first, k, init, w_hats, vs):
    def arr(vals):
        return ", ".join(f"0x{v:016x}ULL" for v in vals)
    L = [
        "/* GENERATED by tools/gen_poseidon_fast_constants.py — do not edit.",
        " * Fast-partial-round form of Poseidon-Goldilocks, re-derived from the round constants and the MDS",
        " * matrix and checked bit-for-bit against the naive permutation (see the script's docstring).",
        " * Plays the role of plonky2 v0.2.0's FAST_PARTIAL_* tables (plonky2/src/hash/poseidon_goldilocks.rs). */",
        "#ifndef VX_POSEIDON_FAST_CONSTANTS_H", "#define VX_POSEIDON_FAST_CONSTANTS_H",
        "#define VX_FAST_PARTIAL_FIRST_ROUND_CONSTANT_INIT { " + arr(first) + " }",
        "#define VX_FAST_PARTIAL_ROUND_CONSTANTS_INIT { " + arr(k) + " }",
        "#define VX_FAST_PARTIAL_INITIAL_MATRIX_INIT { \\",
    ]
    for row in init:
        L.append("  { " + arr(row) + " }, \\")
    L.append("}")
    L.append("#define VX_FAST_PARTIAL_W_HATS_INIT { \\")
    for row in w_hats:
        L.append("  { " + arr(row) + " }, \\")
    L.append("}")
    L.append("#define VX_FAST_PARTIAL_VS_INIT { \\")
    for row in vs:
        L.append("  { " + arr(row) + " }, \\")
    L.append("}")
    L.append("#endif")
    return "\n".join(L) + "\n"


def main():
    rc, M, first, k, init, w_hats, vs = derive()
    assert k[N_PARTIAL - 1] == 0
    kat_in = [[0] * 12, list(range(12)), [P - 1] * 12]
    kat_out0 = [0x3c18a9786cb0b359, 0xd64e1e3efc5b8e9e, 0xbe0085cfc57a8357]
    rnd = random.Random(1)
    tests = kat_in + [[rnd.randrange(P) for _ in range(12)] for _ in range(20)]
    for t, s in enumerate(tests):
        a = permute_naive(s, rc, M)
        b = permute_fast(s, rc, M, first, k, init, w_hats, vs)
        assert a == b, f"fast form differs from naive on test {t}"
        if t < 3:
            assert a[0] == kat_out0[t]
    root = Path(__file__).resolve().parent.parent
    paths = [root / "oracle" / "poseidon_fast_constants.h"]
    text = header_text(first, k, init, w_hats, vs)
    if "--check" in sys.argv:
        ok = all(p.exists() and p.read_text() == text for p in paths)
        print("ok" if ok else "MISMATCH")
        sys.exit(0 if ok else 1)
    for p in paths:
        p.write_text(text)
    print("fast partial-round constants verified on", len(tests), "states; wrote", *paths)


if __name__ == "__main__":
    main()
