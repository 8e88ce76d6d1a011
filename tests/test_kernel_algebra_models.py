"""Integer models of the two algebraic rewrites the round-4 kernels rest on — exact arithmetic in Python, no GPU, no oracle: they
document WHY the device code (vectorx_amd/csrc/poseidon.hip.h, ntt2.hip.h) may do what it does; that it DOES it is the job of the
`-m gpu` parity tests (permutation KATs, iterated permutations, every Merkle shape, every NTT length).

1. Poseidon hashing schedule (`POSEIDON_SCHED_H`): the dense layer that ends full round 3 opens the first integer-power block of the
   partial rounds (a block of 4 whose first S-box is the identity), then 4 + 4 + 4 + 4 + 3.  Block recurrences and constants as
   `make_int_block` / `make_block_consts_g(sched, r0 = 3)` build them.
2. The overwrite-mode sponge reads only the capacity rows of a permutation's last layer when a full chunk follows, only the digest rows
   after the last chunk (`poseidon_sponge_step_nc`).
3. `gl_mul_2exp<64 + k>`:  x * 2^(64+k) = ((x mod 2^32) << k mod 2^32) * (2^32 - 1) - (x >> (32 - k))   (mod p)."""
import random
import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
from gen_poseidon_constants import P, round_constants  # noqa: E402

RC = round_constants()
CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]


def mds():
    m = [[0] * 12 for _ in range(12)]
    for r in range(12):
        for i in range(12):
            m[r][(i + r) % 12] += CIRC[i]
    m[0][0] += 8
    return m


M = mds()


def matmul(x, y):
    return [[sum(x[i][t] * y[t][j] for t in range(12)) for j in range(12)] for i in range(12)]


def q_pow(k):
    q = [row[:] for row in M]
    q[0] = [0] * 12
    r = [[int(i == j) for j in range(12)] for i in range(12)]
    for _ in range(k):
        r = matmul(q, r)
    return r


def int_block(b):
    a, bb, c = {}, {}, {}
    for j in range(1, b):
        a[j] = matmul(M, q_pow(j - 1))[0][:]
        for i in range(1, j):
            bb[(j, i)] = matmul(M, q_pow(j - 1 - i))[0][0]
    cc = matmul(M, q_pow(b - 1))
    for i in range(1, b):
        x = matmul(M, q_pow(b - 1 - i))
        c[i] = [x[r][0] for r in range(12)]
    return a, bb, cc, c


def block_consts(sched, r0):
    kappa, big_k = [], []
    for b in sched:
        cv, kp = [0] * 12, [0] * 4
        for j in range(1, b + 1):
            t = [(sum(M[i][q] * cv[q] for q in range(12)) + RC[(r0 + j) * 12 + i]) % P for i in range(12)]
            if j < b:
                kp[j], cv = t[0], [0] + t[1:]
            else:
                kk = t
        kappa.append(kp)
        big_k.append(kk)
        r0 += b
    return kappa, big_k


def sbox(x):
    return pow(x, 7, P)


def run_block(s, b, kappa, big_k, first_identity):
    a, bb, cc, c = int_block(b)
    y = [0] * b
    y[0] = s[0] if first_identity else sbox(s[0])
    s = [y[0]] + s[1:]
    for j in range(1, b):
        y[j] = sbox((kappa[j] + sum(a[j][i] * s[i] for i in range(12)) + sum(bb[(j, i)] * y[i] for i in range(1, j))) % P)
    return [(big_k[r] + sum(cc[r][i] * s[i] for i in range(12)) + sum(c[i][r] * y[i] for i in range(1, b))) % P for r in range(12)]


def matvec(v, rows=range(12)):
    return [sum(M[r][i] * v[i] for i in range(12)) % P if r in rows else None for r in range(12)]


def permute_naive(s):
    s = s[:]
    for r in range(30):
        s = [(x + RC[12 * r + i]) % P for i, x in enumerate(s)]
        s = [sbox(x) for x in s] if (r < 4 or r >= 26) else [sbox(s[0])] + s[1:]
        s = matvec(s)
    return s


def permute_hashing_schedule(s, last_rows=range(12)):
    s = [(x + RC[i]) % P for i, x in enumerate(s)]
    for r in range(3):
        s = [(a + RC[12 * (r + 1) + i]) % P for i, a in enumerate(matvec([sbox(x) for x in s]))]
    s = [sbox(x) for x in s]                                   # round 3; its dense layer opens block 0
    sched = [4, 4, 4, 4, 4, 3]
    kappa, big_k = block_consts(sched, 3)
    for bi, b in enumerate(sched):
        s = run_block(s, b, kappa[bi], big_k[bi], bi == 0)
    for r in range(26, 29):
        s = [(a + RC[12 * (r + 1) + i]) % P for i, a in enumerate(matvec([sbox(x) for x in s]))]
    return matvec([sbox(x) for x in s], last_rows)             # round 29: only the rows asked for


def test_hashing_schedule_equals_the_naive_rounds():
    rnd = random.Random(5)
    states = [[0] * 12, list(range(12)), [P - 1] * 12] + [[rnd.randrange(P) for _ in range(12)] for _ in range(3)]
    for s in states:
        assert permute_hashing_schedule(s) == permute_naive(s)
    assert permute_naive([0] * 12)[0] == 0x3C18A9786CB0B359       # SURVEY B.2 known answer


def test_block_coefficients_fit_the_multipliers_and_accumulators():
    for b, limit in ((4, (1 << 32) - 1), (3, 1 << 25)):
        a, bb, cc, c = int_block(b)
        assert max(max(row) for row in cc) < 1 << 32
        assert max(sum(cc[r]) + sum(c[i][r] for i in c) for r in range(12)) < limit


def sponge(inputs, step):
    """hash_n_to_m_no_pad, overwrite mode, m = 4"""
    s = [0] * 12
    for c in range(0, len(inputs), 8):
        chunk = inputs[c:c + 8]
        s[:len(chunk)] = chunk
        s = step(s, c + 16 <= len(inputs), c + 8 >= len(inputs))
    return s[:4]


def live_rows_step(s, next_full, last):
    rows = range(8, 12) if next_full else (range(4) if last else range(12))
    out = permute_hashing_schedule(s, rows)
    return [0xDEAD if v is None else v for v in out]           # dead rows: garbage the next chunk must overwrite (or nobody reads)


@pytest.mark.parametrize("width", [5, 7, 8, 9, 15, 16, 17, 20, 24, 25, 33])
def test_sponge_reads_only_the_live_rows(width):
    rnd = random.Random(width)
    inputs = [rnd.randrange(P) for _ in range(width)]
    assert sponge(inputs, live_rows_step) == sponge(inputs, lambda s, nf, la: permute_naive(s))


def test_shift_by_64_plus_k_without_a_multiply():
    rnd = random.Random(9)
    eps = (1 << 32) - 1
    xs = [0, 1, P - 1, (1 << 64) - 1, 1 << 32, (1 << 32) - 1, 1 << 63] + [rnd.randrange(1 << 64) for _ in range(200)]
    for k in range(32):
        for x in xs:                                           # any u64 representative
            y0 = ((x & eps) << k) & eps
            yh = x >> (32 - k)
            assert y0 * eps < P and yh < P                     # one canonical subtraction finishes it
            assert (y0 * eps - yh) % P == (x << (64 + k)) % P
