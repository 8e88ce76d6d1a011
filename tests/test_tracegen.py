"""The row writers of the native trace generators (vectorx_amd/csrc/tracegen_core.h + tracegen_prep.h — the functions the device
kernels call), compiled for the HOST by tests/tracegen_host.cpp and compared CELL BY CELL with the numpy generators of the AIR modules
and with hashlib.  CPU only; the same comparison through the device kernels is tests/test_gpu_tracegen.py."""
import ctypes
import hashlib
import subprocess
from pathlib import Path

import numpy as np
import pytest

from vectorx_amd import blake2b_bytes_air, sha256_air, sha512_air

HERE = Path(__file__).resolve().parent


@pytest.fixture(scope="module")
def tgh():
    import os
    so = HERE / "libtracegen_host.so"
    src = HERE / "tracegen_host.cpp"
    deps = [src] + sorted((HERE.parent / "vectorx_amd" / "csrc").glob("tracegen_*.h"))
    if os.environ.get("VX_TRACEGEN_HOST_SO"):            # tools/sanitize_host.sh: an ASan + UBSan build of the same file
        so = Path(os.environ["VX_TRACEGEN_HOST_SO"])
    elif not so.exists() or so.stat().st_mtime < max(d.stat().st_mtime for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", str(so), str(src)], check=True)
    L = ctypes.CDLL(str(so))
    for f in (L.tgh_sha256, L.tgh_sha512, L.tgh_blake2b, L.tgh_sha512_bus):
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 2 + [ctypes.c_int] + [ctypes.c_void_p] * 3
    return L


def run(fn, ncols, npis, dwords, log_n, msgs):
    blob = b"".join(msgs)
    buf = np.frombuffer(blob, dtype=np.uint8) if blob else np.zeros(1, np.uint8)
    off = np.zeros(len(msgs) + 1, dtype=np.uint64)
    np.cumsum([len(m) for m in msgs], out=off[1:])
    trace = np.full((ncols, 1 << log_n), 0xDEAD, dtype=np.uint64)      # every cell must be written
    pis = np.zeros(npis, dtype=np.uint64)
    dig = np.zeros(max(1, dwords * len(msgs)), dtype=np.uint64)
    rc = fn(log_n, buf.ctypes.data, off.ctypes.data, len(msgs), trace.ctypes.data, pis.ctypes.data, dig.ctypes.data)
    return rc, trace, pis, dig


def seeded(n, lens, seed):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, 256, size=lens[i % len(lens)], dtype=np.uint8).tobytes() for i in range(n)]


@pytest.mark.parametrize("log_n,msgs", [
    (9, [b"abc", b"", b"x" * 55, b"y" * 56, b"z" * 64]),
    (10, seeded(6, [64, 100, 1, 119], 1)),
    (8, [b"q" * 130]),
    (7, []),
])
def test_sha256_rows_equal_the_numpy_generator(tgh, log_n, msgs):
    rc, trace, pis, dig = run(tgh.tgh_sha256, 1024, 8, 8, log_n, msgs)
    assert rc == 0
    ref, rpis, rdig = sha256_air.generate_trace(log_n, msgs)
    assert len(rdig) == len(msgs)
    bad = np.argwhere(trace != ref)
    assert bad.size == 0, f"first differing cells (column, row): {bad[:8].tolist()}"
    assert (pis == rpis).all()
    for i, m in enumerate(msgs):
        assert b"".join(int(w).to_bytes(4, "big") for w in dig[8 * i:8 * i + 8]) == hashlib.sha256(m).digest() == rdig[i]


@pytest.mark.parametrize("log_n,msgs", [
    (9, [b"abc", b"", b"x" * 111, b"y" * 112]),
    (10, seeded(5, [117, 128, 3, 240], 2)),
    (8, []),
])
def test_sha512_rows_equal_the_numpy_generator(tgh, log_n, msgs):
    rc, trace, pis, dig = run(tgh.tgh_sha512, 1995, 16, 8, log_n, msgs)
    assert rc == 0
    ref, rpis, rdig = sha512_air.generate_trace(log_n, msgs)
    assert len(rdig) == len(msgs)
    bad = np.argwhere(trace != ref)
    assert bad.size == 0, f"first differing cells (column, row): {bad[:8].tolist()}"
    assert (pis == rpis).all()
    for i, m in enumerate(msgs):
        assert b"".join(int(w).to_bytes(8, "big") for w in dig[8 * i:8 * i + 8]) == hashlib.sha512(m).digest() == rdig[i]


def test_sha512_bus_variant_rows_equal_the_numpy_generator(tgh):
    """the table that sends (R, A, digest) on the signature bus: 17 more columns (the message's first 64 bytes latched, the first-block flag)"""
    msgs = [b"R" * 32 + b"A" * 32 + b"message", seeded(1, [300], 8)[0], b"short", b"", seeded(1, [64], 9)[0]]
    rc, trace, pis, dig = run(tgh.tgh_sha512_bus, 2012, 16, 8, 10, msgs)
    assert rc == 0
    ref, rpis, rdig = sha512_air.generate_trace(10, msgs, bus=True)
    bad = np.argwhere(trace != ref)
    assert bad.size == 0, f"first differing cells (column, row): {bad[:8].tolist()}"
    assert (pis == rpis).all() and len(rdig) == len(msgs)


@pytest.mark.parametrize("msgs", [
    [b"abc", b"", b"x" * 128, b"y" * 129, seeded(1, [1000], 3)[0]],
    seeded(8, [128 * 20, 128 * 20 - 5], 4),
])
def test_blake2b_rows_equal_the_numpy_generator(tgh, msgs):
    log_n = 16
    rc, trace, pis, dig = run(tgh.tgh_blake2b, 775, 8, 4, log_n, msgs)
    assert rc == 0
    ref, rpis, rdig = blake2b_bytes_air.generate_trace(log_n, msgs)
    assert len(rdig) == len(msgs)
    bad = np.argwhere(trace != ref)
    assert bad.size == 0, f"first differing cells (column, row): {bad[:8].tolist()}"
    assert (pis == rpis).all()
    for i, m in enumerate(msgs):
        assert b"".join(int(w).to_bytes(8, "little") for w in dig[4 * i:4 * i + 4]) == hashlib.blake2b(m, digest_size=32).digest() == rdig[i]


def test_a_message_that_does_not_fit_is_refused(tgh):
    rc, *_ = run(tgh.tgh_sha256, 1024, 8, 8, 7, [b"a" * 64] * 2)        # 4 blocks of 66 rows > 128 rows
    assert rc == 1
    rc, *_ = run(tgh.tgh_blake2b, 775, 8, 4, 15, [b"a"])                 # the XOR table needs 2^16 rows
    assert rc == 2


def _eddsa_host(tgh, lay, log_n, sigs):
    full = lay.full
    stride = 6 if full else 4
    arr = np.zeros((max(1, len(sigs)), stride, 4), dtype=np.uint64)
    words = lambda v: [(int(v) >> (64 * w)) & 0xFFFFFFFFFFFFFFFF for w in range(4)]      # noqa: E731
    for i, sg in enumerate(sigs):
        arr[i, 0], arr[i, 1], arr[i, 2], arr[i, 3] = words(sg[0][0]), words(sg[0][1]), words(sg[1]), words(sg[2])
        if full:
            arr[i, 4], arr[i, 5] = words(sg[3]), words(int(sg[3]) >> 256)
    trace = np.full((lay.N, 1 << log_n), 0xDEAD, dtype=np.uint64)
    res = np.zeros((max(1, len(sigs)), 2, 4), dtype=np.uint64)
    tgh.tgh_eddsa.restype = ctypes.c_int
    tgh.tgh_eddsa.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    rc = tgh.tgh_eddsa(log_n, lay.NB, int(full), arr.ctypes.data, len(sigs), trace.ctypes.data, res.ctypes.data)
    val = lambda w: sum(int(w[k]) << (64 * k) for k in range(4))      # noqa: E731
    return rc, trace, [(val(res[i, 0]), val(res[i, 1])) for i in range(len(sigs))]


@pytest.mark.parametrize("scalar_bits,log_n,nsig,distinct", [(32, 17, 5, 3), (64, 17, 46, 2), (256, 17, 3, 3)])
def test_eddsa_rows_equal_the_numpy_generator(tgh, scalar_bits, log_n, nsig, distinct):
    """the batched EdDSA table: real signature equations (cut to `scalar_bits` bits of S and h), then filler instances, an unfinished tail"""
    from vectorx_amd import eddsa_air as ea
    from vectorx_amd import stark_chips
    lay = ea.Layout(16, scalar_bits)
    full, _ = stark_chips.eddsa_signatures(nsig, distinct)
    mask = (1 << scalar_bits) - 1
    sigs = [(a, s & mask, h & mask) for (a, s, h) in full]
    ref, rres = ea.generate_trace(lay, log_n, sigs)
    rc, trace, res = _eddsa_host(tgh, lay, log_n, sigs)
    assert rc == 0
    bad = np.argwhere(trace != ref)
    assert bad.size == 0, f"first differing cells (column, row): {bad[:8].tolist()}"
    assert res == rres
    if scalar_bits == 256:          # whole scalars: the instance arrives at R of the RFC 8032 signature
        assert rres == stark_chips.eddsa_signatures(nsig, distinct)[1]


def test_full_eddsa_rows_equal_the_numpy_generator_and_refuse_what_it_refuses(tgh):
    """the FULL program (decompression, digest mod L, S < L inside the instance): RFC 8032 signatures from their bytes"""
    from test_eddsa_air import RFC8032
    from vectorx_amd import eddsa_air as ea
    lay, log_n = ea.Layout(16, 256, full=True), 17
    raw = [(bytes.fromhex(pk), bytes.fromhex(msg), bytes.fromhex(sig)) for _, pk, msg, sig in RFC8032]
    sigs = [ea.equation_inputs_full(pk, msg, sig) for pk, msg, sig in raw]
    ref, rres = ea.generate_trace(lay, log_n, sigs)
    rc, trace, res = _eddsa_host(tgh, lay, log_n, sigs)
    assert rc == 0
    bad = np.argwhere(trace != ref)
    assert bad.size == 0, f"first differing cells (column, row): {bad[:8].tolist()}"
    assert res == rres == [ea.decompress(sig[:32]) for _, _, sig in raw]
    a, s, h, d = sigs[0]
    assert _eddsa_host(tgh, lay, log_n, [(a, s + ea.ELL, h, d)])[0] == 12                       # S >= L: no comparison witness
    assert _eddsa_host(tgh, lay, log_n, [((a[0] + ea.Q25519, a[1]), s, h, d)])[0] in (11, 12)   # a non-canonical x
