"""vx_trace_sha256 / vx_trace_sha512 / vx_trace_blake2b on the GPU (csrc/tracegen.hip.h) through the C ABI: the trace the device writes
equals the numpy generator's cell for cell, the digests equal hashlib, and the table proves from the device-generated trace to the same
bytes as from the host-generated one.  (The row writers themselves are compared on the CPU in tests/test_tracegen.py.)"""
import hashlib

import numpy as np
import pytest

from vectorx_amd import blake2b_bytes_air, sha256_air, sha512_air, stark_chips

pytestmark = pytest.mark.gpu

AIRS = {"sha256": sha256_air, "sha512": sha512_air, "blake2b": blake2b_bytes_air}


def test_device_sha512_bus_variant_equals_the_numpy_generator(ctx):
    """the SHA-512 table that sends (R, A, digest) on the signature bus (2012 columns), generated on the device"""
    msgs = [b"R" * 32 + b"A" * 32 + b"message", seeded(1, [300], 8)[0], b"short", b""]
    n = 1 << 10
    d = ctx.alloc(2012 * n * 8)
    try:
        ctx.upload(d, np.full((2012, n), 0xDEAD, dtype=np.uint64))
        pis, digests = ctx.trace_hash_table("sha512_bus", 10, msgs, d)
        got = ctx.download(d, 2012 * n * 8).reshape(2012, n)
    finally:
        ctx.free(d)
    ref, rpis, rdig = sha512_air.generate_trace(10, msgs, bus=True)
    bad = np.argwhere(got != ref)
    assert bad.size == 0, f"first differing cells (column, row): {bad[:8].tolist()}"
    assert (pis == rpis).all() and digests == rdig == [hashlib.sha512(m).digest() for m in msgs]
HASH = {"sha256": lambda m: hashlib.sha256(m).digest(), "sha512": lambda m: hashlib.sha512(m).digest(),
        "blake2b": lambda m: hashlib.blake2b(m, digest_size=32).digest()}


def seeded(n, lens, seed):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, 256, size=lens[i % len(lens)], dtype=np.uint8).tobytes() for i in range(n)]


CASES = [("sha256", 9, [b"abc", b"", b"x" * 55, b"y" * 56, b"z" * 64]), ("sha256", 11, seeded(14, [64], 5)), ("sha256", 8, []),
         ("sha512", 9, [b"abc", b"", b"x" * 111, b"y" * 112]), ("sha512", 11, seeded(8, [117], 6)),
         ("blake2b", 16, [b"abc", b"", b"x" * 128, b"y" * 129] + seeded(4, [128 * 30, 128 * 30 - 7], 7)),
         ("blake2b", 16, [b"\x00" * 4096, b"\xff" * 4096, b"\x11" * 4096])]          # constant bytes: the hot-key path of the lookup histogram


@pytest.mark.parametrize("which,log_n,msgs", CASES, ids=[f"{c[0]}-2^{c[1]}-{len(c[2])}msgs" for c in CASES])
def test_device_trace_equals_the_numpy_generator(ctx, which, log_n, msgs):
    ncols = ctx.TRACE_TABLES[which][0]
    n = 1 << log_n
    d = ctx.alloc(ncols * n * 8)
    try:
        ctx.upload(d, np.full((ncols, n), 0xDEAD, dtype=np.uint64))          # every cell must be written by the kernels
        pis, digests = ctx.trace_hash_table(which, log_n, msgs, d)
        got = ctx.download(d, ncols * n * 8).reshape(ncols, n)
    finally:
        ctx.free(d)
    ref, rpis, rdig = AIRS[which].generate_trace(log_n, msgs)
    bad = np.argwhere(got != ref)
    assert bad.size == 0, f"first differing cells (column, row): {bad[:8].tolist()}"
    assert (pis == rpis).all()
    assert digests == [HASH[which](m) for m in msgs] == rdig


def test_a_message_that_does_not_fit_is_an_error_not_a_truncated_table(ctx):
    import vectorx_amd as vx
    d = ctx.alloc(1024 * 128 * 8)
    try:
        with pytest.raises(vx.VxError, match="do not complete"):
            ctx.trace_hash_table("sha256", 7, [b"a" * 64] * 2, d)
        with pytest.raises(vx.VxError, match="degree_bits >= 16"):
            ctx.trace_hash_table("blake2b", 15, [b"a"], d)
    finally:
        ctx.free(d)


@pytest.mark.parametrize("which,log_n,msgs", [("sha256", 9, seeded(2, [64], 11)), ("blake2b", 16, seeded(8, [128 * 16], 12))])
def test_a_table_proves_from_the_device_generated_trace(ctx, which, log_n, msgs):
    air = AIRS[which]
    stark = air.make_stark(log_n)
    tab = stark_chips.GeneratedHashTable(ctx, which, stark, log_n, lambda job: msgs, [ctx], which)
    ref_trace, ref_pis, _ = air.generate_trace(log_n, msgs)
    res = stark_chips.ResidentTable(ctx, stark, ref_trace, ref_pis, which)
    try:
        proof = tab.prove(ctx, ("map", 0, 0, b""))
        spent = tab.take_spent(ctx)
        assert spent and spent[0][0] == "trace_generation" and spent[0][1] > 0
        stark.verify(ref_pis, proof)
        assert proof == res.prove(), "same trace, same transcript: the proofs must be the same bytes"
    finally:
        tab.free()
        res.free()


@pytest.mark.parametrize("scalar_bits,nsig,distinct", [(32, 5, 3), (256, 7, 4)])
def test_device_eddsa_trace_equals_the_numpy_generator(ctx, scalar_bits, nsig, distinct):
    from vectorx_amd import eddsa_air as ea
    log_n = 17
    lay = ea.Layout(16, scalar_bits)
    full, full_r = stark_chips.eddsa_signatures(nsig, distinct)
    mask = (1 << scalar_bits) - 1
    sigs = [(a, s & mask, h & mask) for (a, s, h) in full]
    n = 1 << log_n
    d = ctx.alloc(lay.N * n * 8)
    try:
        ctx.upload(d, np.full((lay.N, n), 0xDEAD, dtype=np.uint64))
        res = ctx.trace_eddsa_table(log_n, scalar_bits, sigs, d)
        got = ctx.download(d, lay.N * n * 8).reshape(lay.N, n)
    finally:
        ctx.free(d)
    ref, rres = ea.generate_trace(lay, log_n, sigs)
    bad = np.argwhere(got != ref)
    assert bad.size == 0, f"first differing cells (column, row): {bad[:8].tolist()}"
    assert res == rres
    if scalar_bits == 256:
        assert res == full_r          # the instance arrives at R of the RFC 8032 signature


def test_device_full_eddsa_trace_equals_the_numpy_generator_and_refuses_what_it_refuses(ctx):
    """the FULL program on the device (decompression, digest mod L, S < L inside the instance): RFC 8032 signatures from their bytes"""
    import vectorx_amd as vx
    from test_eddsa_air import RFC8032
    from vectorx_amd import eddsa_air as ea
    lay, log_n = ea.Layout(16, 256, full=True), 17
    raw = [(bytes.fromhex(pk), bytes.fromhex(msg), bytes.fromhex(sig)) for _, pk, msg, sig in RFC8032]
    sigs = [ea.equation_inputs_full(pk, msg, sig) for pk, msg, sig in raw]
    n = 1 << log_n
    d = ctx.alloc(lay.N * n * 8)
    try:
        ctx.upload(d, np.full((lay.N, n), 0xDEAD, dtype=np.uint64))
        res = ctx.trace_eddsa_table(log_n, 256, sigs, d, full=True)
        got = ctx.download(d, lay.N * n * 8).reshape(lay.N, n)
        ref, rres = ea.generate_trace(lay, log_n, sigs)
        bad = np.argwhere(got != ref)
        assert bad.size == 0, f"first differing cells (column, row): {bad[:8].tolist()}"
        assert res == rres == [ea.decompress(sig[:32]) for _, _, sig in raw]
        a, s, h, dg = sigs[0]
        with pytest.raises(vx.VxError, match="canonical"):
            ctx.trace_eddsa_table(log_n, 256, [(a, s + ea.ELL, h, dg)], d, full=True)            # S >= L
    finally:
        ctx.free(d)


def test_eddsa_table_proves_from_the_device_generated_trace_and_refuses_a_point_off_the_curve(ctx):
    import vectorx_amd as vx
    from vectorx_amd import eddsa_air as ea
    log_n, lay = 17, ea.Layout()
    sigs, rs = stark_chips.eddsa_signatures(ea.capacity(lay, log_n), 2)
    stark = ea.make_stark(lay, log_n)
    tabs = stark_chips.GeneratedEddsaTables(ctx, stark, lay, log_n, lambda job: sigs, [ctx])
    ref_trace, ref_res = ea.generate_trace(lay, log_n, sigs)
    res = stark_chips.ResidentTable(ctx, stark, ref_trace, np.zeros(0, dtype=np.uint64), "eddsa")
    try:
        proof = tabs.prove(ctx, None)
        assert tabs.last[id(ctx)][1] == rs == ref_res
        sums = stark.verify(np.zeros(0, dtype=np.uint64), proof)
        assert any(int(x) for x in sums)                 # the bus carries the results out
        assert proof == res.prove()
        (ax, ay), s_, h_ = sigs[0]
        with pytest.raises(vx.VxError, match="not on the curve"):
            ctx.trace_eddsa_table(log_n, 256, [((ax, (ay + 1) % ea.Q25519), s_, h_)], tabs.bufs[id(ctx)][0])
    finally:
        tabs.free()
        res.free()


def test_the_one_lane_and_the_four_lane_simulation_write_the_same_trace(ctx):
    """Round 6: `tg_ed_simulate4_kernel` (four lanes per signature, the ladder step's 13 levels of independent rows) replaced the
    one-lane-per-signature walk, which stays behind VX_TRACE_EDDSA_ONE_LANE=1: same cells (the full program, 20 instances so that the
    last group of a block is only partly filled), same results."""
    import hashlib
    import os
    import subprocess
    import sys
    from pathlib import Path
    from vectorx_amd import eddsa_air as ea
    root = Path(__file__).resolve().parent.parent
    log_n = 18
    lay = ea.Layout(16, 256, full=True)
    sigs, rs = stark_chips.eddsa_signatures_full(20, 4)
    n = 1 << log_n
    d = ctx.alloc(lay.N * n * 8)
    try:
        assert ctx.trace_eddsa_table(log_n, lay.NB, sigs, d, full=True) == rs
        mine = hashlib.sha256(ctx.download(d, lay.N * n * 8).tobytes()).hexdigest()
    finally:
        ctx.free(d)
    code = (
        "import sys, hashlib; sys.path.insert(0, %r)\n"
        "import vectorx_amd as vx\n"
        "from vectorx_amd import eddsa_air as ea, stark_chips\n"
        "ctx = vx.Context(0); lay = ea.Layout(16, 256, full=True)\n"
        "sigs, rs = stark_chips.eddsa_signatures_full(20, 4)\n"
        "d = ctx.alloc(lay.N * (1 << 18) * 8)\n"
        "assert ctx.trace_eddsa_table(18, lay.NB, sigs, d, full=True) == rs\n"
        "print(hashlib.sha256(ctx.download(d, lay.N * (1 << 18) * 8).tobytes()).hexdigest())\n"
    ) % str(root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env={**os.environ, "VX_TRACE_EDDSA_ONE_LANE": "1"})
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    assert r.stdout.strip().splitlines()[-1] == mine
