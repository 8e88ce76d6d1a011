"""GPU test of the hand-written Goldilocks primitives (carry-chained inline asm in goldilocks.hip.h) against
Python big-integer arithmetic, on adversarial operands: the rare carry / borrow corner cases of the
2^64 = 2^32 - 1 fold (values around 0, 2^32, 2^64 - 2^32, p, 2^64) must be exact, not just random ones."""
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001
M64 = (1 << 64) - 1


def _edge_values():
    base = [0, 1, 2, 3, 0xFFFFFFFF, 0x100000000, 0x100000001, 0xFFFFFFFE, P - 2, P - 1, P, P + 1, M64, M64 - 1,
            0xFFFFFFFF00000000, 0xFFFFFFFEFFFFFFFF, 0xFFFFFFFE00000001, 0x00000001FFFFFFFF, 0x8000000000000000,
            0x7FFFFFFFFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF << 32, (1 << 63) + (1 << 31), 0x0000000100000000 - 2,
            0xFFFFFFFF80000000, 0x80000000FFFFFFFF, 0xFFFF0000FFFF0001, 0x00000000FFFFFFFF + (0xFFFFFFFE << 32)]
    return sorted(set(v & M64 for v in base))


def test_field_ops_on_edge_and_random_operands(ctx):
    ev = _edge_values()
    pairs = list(itertools.product(ev, ev))
    rng = np.random.default_rng(0)
    rnd = rng.integers(0, M64, size=(20000, 2), dtype=np.uint64, endpoint=True)
    # products whose high half is tiny / huge exercise the borrow path of the reduction
    small = rng.integers(0, 1 << 33, size=(5000, 2), dtype=np.uint64)
    a = np.array([p[0] for p in pairs] + rnd[:, 0].tolist() + small[:, 0].tolist(), dtype=np.uint64)
    b = np.array([p[1] for p in pairs] + rnd[:, 1].tolist() + small[:, 1].tolist(), dtype=np.uint64)
    ai = [int(x) for x in a]
    bi = [int(x) for x in b]
    exp = {
        0: [x * y % P for x, y in zip(ai, bi)],
        1: [(x + y) % P for x, y in zip(ai, bi)],
        2: [(x - y) % P for x, y in zip(ai, bi)],
        3: [(x * y + x) % P for x, y in zip(ai, bi)],
        5: [x * y % P for x, y in zip(ai, bi)],
    }
    for op, e in exp.items():
        got = ctx.field_op(op, a, b)
        bad = [i for i in range(len(e)) if int(got[i]) != e[i]]
        assert not bad, f"op {op}: first mismatch a={ai[bad[0]]:#x} b={bi[bad[0]]:#x} got={int(got[bad[0]]):#x} exp={e[bad[0]]:#x}"


def test_field_inverse(ctx):
    rng = np.random.default_rng(1)
    a = np.concatenate([np.array([1, 2, P - 1, 0xFFFFFFFF, 1 << 32, 7], dtype=np.uint64),
                        rng.integers(1, P, size=2000, dtype=np.uint64)])
    inv = ctx.field_op(4, a, a)
    assert all(int(x) * int(y) % P == 1 for x, y in zip(a, inv))
