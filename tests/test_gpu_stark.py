"""STARK spike on the GPU: `vx_stark_prove` — trace commitment, AIR quotient (constraint program
interpreted per row), openings and FRI on the same kernels as vx_prove — must give BYTE-IDENTICAL proofs to the oracle's
restated starky prover (oracle/stark.hpp), and the product's host verifier must accept them."""
import numpy as np
import pytest

import oracle_lib
import vectorx_amd as vx
from stark_airs import cubic, fibonacci, logup, mulchain

pytestmark = pytest.mark.gpu
P = oracle_lib.P


@pytest.mark.parametrize("make,degree_bits,cfg", [(fibonacci, 3, {}), (fibonacci, 5, {}), (fibonacci, 8, {}), (fibonacci, 12, {}),
                                                  (fibonacci, 10, dict(rate_bits=2, num_query_rounds=20)),
                                                  (fibonacci, 9, dict(rate_bits=3, num_challenges=1, cap_height=2)),
                                                  (cubic, 5, dict(rate_bits=1)), (cubic, 9, dict(rate_bits=1)), (cubic, 11, dict(rate_bits=2)),
                                                  (cubic, 7, dict(rate_bits=3, num_query_rounds=10, fri_arities=[2, 3])),
                                                  (fibonacci, 16, {}), (mulchain, 6, dict(groups=1)), (mulchain, 10, dict(groups=5)),
                                                  (mulchain, 13, dict(groups=8, rate_bits=2, num_query_rounds=40)),
                                                  (logup, 5, {}), (logup, 9, dict(table_bits=6)), (logup, 12, dict(rate_bits=2, num_query_rounds=40))])
def test_stark_proof_bytes_identical_to_oracle(ctx, oracle, make, degree_bits, cfg):
    cfg = dict(pow_bits=8, **cfg)
    stark, trace, pis = make(degree_bits, **cfg)
    gp = stark.prove(ctx, trace, pis)
    op = oracle_lib.stark_prove(oracle, stark, trace, pis)
    assert len(gp) == len(op)
    assert gp == op
    stark.verify(pis, gp)
    assert stark.prove(ctx, trace, pis) == gp                    # deterministic
    bad = bytearray(gp)
    bad[len(bad) // 3] ^= 2
    with pytest.raises(vx.VxError):
        stark.verify(pis, bytes(bad))


def test_stark_pow_hint_and_noncanonical_trace(ctx, oracle):
    stark, trace, pis = fibonacci(7, pow_bits=6)
    ref = stark.prove(ctx, trace, pis)
    pw = int(np.frombuffer(ref[-32:-24], dtype="<u8")[0])           # pow_witness sits before the 3 public inputs
    assert stark.prove(ctx, trace, pis, pow_witness=pw) == ref
    for cand in range(max(0, pw - 2), pw):
        with pytest.raises(vx.VxError):
            stark.prove(ctx, trace, pis, pow_witness=cand)
    t = trace.copy()
    small = t < np.uint64(2**32 - 1)
    t[small] = t[small] + np.uint64(P)                               # x + p: same field element
    assert stark.prove(ctx, t, pis) == ref


def test_stark_violated_air_is_refused_or_rejected(ctx):
    stark, trace, pis = fibonacci(6, pow_bits=4)
    t = trace.copy()
    t[0, 9] = (int(t[0, 9]) + 5) % P
    try:
        proof = stark.prove(ctx, t, pis)
    except vx.VxError:
        return
    with pytest.raises(vx.VxError):
        stark.verify(pis, proof)


def test_air_programs_are_compiled_to_native_code_and_match_the_interpreter(ctx, oracle):
    """vx_stark_prove compiles the AIR program with hiprtc (jit.hip.h, the lowering gate programs use); VX_NO_JIT=1 keeps the
    on-GPU interpreter.  Both give the oracle's proof, byte for byte; the profile names which one ran."""
    import os
    for make, lg in [(cubic, 10), (fibonacci, 11), (logup, 10)]:
        stark, trace, pis = make(lg, pow_bits=6)
        expect = oracle_lib.stark_prove(oracle, stark, trace, pis)
        ctx.prof_enable(True)
        ctx.prof_reset()
        assert stark.prove(ctx, trace, pis) == expect
        stages = ctx.prof()
        assert "air_quotient_eval_jit" in stages and "air_quotient_eval" not in stages, sorted(stages)
        os.environ["VX_NO_JIT"] = "1"
        try:
            ctx.prof_reset()
            assert stark.prove(ctx, trace, pis) == expect
            stages = ctx.prof()
            assert "air_quotient_eval" in stages and "air_quotient_eval_jit" not in stages, sorted(stages)
        finally:
            del os.environ["VX_NO_JIT"]
            ctx.prof_enable(False)


def test_two_round_session_api(ctx, oracle):
    """vx_stark_begin / vx_stark_finish directly: device-resident aux columns, a too-small output buffer leaves the session
    usable, a second finish is refused, vx_stark_prove refuses an AIR with a second round."""
    import ctypes
    stark, trace, pis = logup(8, pow_bits=4)
    expect = oracle_lib.stark_prove(oracle, stark, trace, pis)
    L = vx.lib()
    vp = ctypes.c_void_p
    chal = np.zeros(stark.desc.num_aux_challenges, dtype=np.uint64)      # one per challenge set (Stark repeats the second round)
    sess = vp()
    t = np.ascontiguousarray(trace, dtype=np.uint64)
    assert L.vx_stark_begin(ctx._h, ctypes.cast(stark.desc_ptr, vp), t.ctypes.data, 0, pis.ctypes.data, chal.ctypes.data, ctypes.byref(sess)) == 0
    aux = stark.run_aux(t, chal)[0]
    d_aux = ctx.alloc(aux.nbytes)
    ctx.upload(d_aux, aux)
    small = np.empty(16, dtype=np.uint8)
    n = ctypes.c_size_t(small.size)
    assert L.vx_stark_finish(sess, vp(d_aux), 1, None, small.ctypes.data, ctypes.byref(n)) == vx.VX_E_INVALID and n.value == len(expect)
    out = np.empty(n.value, dtype=np.uint8)
    assert L.vx_stark_finish(sess, vp(d_aux), 1, None, out.ctypes.data, ctypes.byref(n)) == 0
    assert out[:n.value].tobytes() == expect
    assert L.vx_stark_finish(sess, vp(d_aux), 1, None, out.ctypes.data, ctypes.byref(n)) == vx.VX_E_INVALID
    L.vx_stark_session_free(sess)
    ctx.free(d_aux)
    big = np.empty(1 << 20, dtype=np.uint8)
    n = ctypes.c_size_t(big.size)
    assert L.vx_stark_prove(ctx._h, ctypes.cast(stark.desc_ptr, vp), t.ctypes.data, 0, pis.ctypes.data, None, big.ctypes.data, ctypes.byref(n)) == vx.VX_E_INVALID


# ---- chip density: the SHA-256 AIR of vectorx_amd/sha256_air.py — 1024 + 3 columns, 2072 constraints of degree <= 3,
#      a 16 k-word program, a log-derivative range check in the second commitment round — own AIR, not Curta's ----------------------
SHA_MESSAGES = [b"abc", b"", b"The quick brown fox jumps over the lazy dog" * 3, bytes(range(200)), b"vectorx" * 100]


@pytest.mark.parametrize("degree_bits,cfg", [(7, {}), (9, dict(rate_bits=2, num_query_rounds=30)), (11, dict(num_query_rounds=40))])
def test_sha256_air_proof_bytes_identical_to_oracle(ctx, oracle, degree_bits, cfg):
    import hashlib
    from vectorx_amd import sha256_air as sha
    cfg = dict(dict(pow_bits=8, num_query_rounds=20), **cfg)
    stark = sha.make_stark(degree_bits, **cfg)
    trace, pis, digests = sha.generate_trace(degree_bits, SHA_MESSAGES)
    assert digests == [hashlib.sha256(m).digest() for m in SHA_MESSAGES[:len(digests)]] and digests
    ctx.prof_enable(True)
    ctx.prof_reset()
    gp = stark.prove(ctx, trace, pis)
    stages = ctx.prof()
    ctx.prof_enable(False)
    assert "air_quotient_eval_jit" in stages, sorted(stages)           # the 16 k-word program compiles (VX_PROGRAM_REGS / length limits do not bind)
    op = oracle_lib.stark_prove(oracle, stark, trace, pis)
    assert gp == op
    stark.verify(pis, gp)
    wrong = pis.copy()
    wrong[7] = (int(wrong[7]) + 1) % P
    with pytest.raises(vx.VxError):
        stark.verify(wrong, gp)
    bad = trace.copy()                                                   # a single flipped state bit: refused or rejected
    bad[sha.Cols.S + 40, 77] ^= np.uint64(1)
    try:
        bp = stark.prove(ctx, bad, pis)
    except vx.VxError:
        bp = None
    if bp is not None:
        with pytest.raises(vx.VxError):
            stark.verify(pis, bp)


@pytest.mark.parametrize("which,degree_bits,cfg", [("sha256", 7, {}), ("sha256", 11, dict(num_query_rounds=40)), ("sha512", 9, dict(rate_bits=2))])
def test_openings_digest_on_the_device_equals_the_oracles(ctx, oracle, which, degree_bits, cfg):
    """VX_STARK_OPENINGS_DIGEST: the tree hash of the opening set computed ON THE DEVICE next to the evaluations (one leaf-hash launch +
    the fused tree top inside the openings' one submission) gives the oracle's transcript: byte-identical proofs for narrow and wide
    tables (258 ... 1004 leaves of 8 elements), stage `openings_digest` on the clock; the product verifier accepts, and refuses the same table's
    proof made without the option."""
    import importlib
    air = importlib.import_module("vectorx_amd." + which + "_air")
    cfg = dict(dict(pow_bits=6, num_query_rounds=16), **cfg)
    msgs = [b"abc", b"", b"y" * 200]
    stark, plain = air.make_stark(degree_bits, openings_digest=True, **cfg), air.make_stark(degree_bits, **cfg)
    trace, pis, _ = air.generate_trace(degree_bits, msgs)
    ctx.prof_enable(True)
    ctx.prof_reset()
    gp = stark.prove(ctx, trace, pis)
    stages = ctx.prof()
    ctx.prof_enable(False)
    assert "openings_digest" in stages and stages["eval_ext"]["calls"] == 1, sorted(stages)
    assert gp == oracle_lib.stark_prove(oracle, stark, trace, pis)
    stark.verify(pis, gp)
    p0 = plain.prove(ctx, trace, pis)
    assert p0 != gp and len(p0) == len(gp)
    with pytest.raises(vx.VxError):
        stark.verify(pis, p0)
    with pytest.raises(vx.VxError):
        plain.verify(pis, gp)


def test_sha256_air_interpreted_equals_compiled(ctx, oracle):
    import os
    from vectorx_amd import sha256_air as sha
    stark = sha.make_stark(7, pow_bits=6, num_query_rounds=12)
    trace, pis, _ = sha.generate_trace(7, SHA_MESSAGES)
    expect = stark.prove(ctx, trace, pis)
    os.environ["VX_NO_JIT"] = "1"
    try:
        ctx.prof_enable(True)
        ctx.prof_reset()
        assert stark.prove(ctx, trace, pis) == expect
        assert "air_quotient_eval" in ctx.prof()
    finally:
        del os.environ["VX_NO_JIT"]
        ctx.prof_enable(False)


# ---- a second chip: the BLAKE2b AIR of vectorx_amd/blake2b_air.py — 1063 + 6 columns, 1561 constraints, message bytes looked up in a
#      256-entry table (8 lookups per row, five helper columns) — the header-hash chip of circuits/builder/header.rs:18; own AIR ------
BLAKE_MESSAGES = [b"abc", b"", bytes(range(200)), b"x" * 128, b"y" * 129, b"vectorx" * 100]


@pytest.mark.parametrize("degree_bits,cfg", [(9, {}), (10, dict(rate_bits=2, num_query_rounds=30)), (12, dict(num_query_rounds=40))])
def test_blake2b_air_proof_bytes_identical_to_oracle(ctx, oracle, degree_bits, cfg):
    import hashlib
    from vectorx_amd import blake2b_air as b2
    cfg = dict(dict(pow_bits=8, num_query_rounds=20), **cfg)
    stark = b2.make_stark(degree_bits, **cfg)
    trace, pis, digests = b2.generate_trace(degree_bits, BLAKE_MESSAGES)
    assert digests == [hashlib.blake2b(m, digest_size=32).digest() for m in BLAKE_MESSAGES[:len(digests)]] and len(digests) >= 3
    ctx.prof_enable(True)
    ctx.prof_reset()
    gp = stark.prove(ctx, trace, pis)
    stages = ctx.prof()
    ctx.prof_enable(False)
    assert "air_quotient_eval_jit" in stages, sorted(stages)
    assert gp == oracle_lib.stark_prove(oracle, stark, trace, pis)
    stark.verify(pis, gp)
    wrong = pis.copy()
    wrong[0] = (int(wrong[0]) + 1) % P
    with pytest.raises(vx.VxError):
        stark.verify(wrong, gp)
    bad = trace.copy()                                                   # a single flipped bit of an intermediate of G: refused or rejected
    bad[b2.Cols.BITS + 64 * b2.W_C1 + 17, 77] ^= np.uint64(1)
    try:
        bp = stark.prove(ctx, bad, pis)
    except vx.VxError:
        bp = None
    if bp is not None:
        with pytest.raises(vx.VxError):
            stark.verify(pis, bp)


def test_host_trace_commits_behind_its_upload_with_carried_state_hashing(ctx, oracle, monkeypatch):
    """vx_stark_begin with a HOST trace: column blocks cross PCIe behind the transforms, and for big traces the leaf sponge runs in
    carried-state launches (threshold lowered here): same bytes as the oracle and as the device-resident trace."""
    from vectorx_amd import blake2b_air as b2
    stark = b2.make_stark(9, pow_bits=6, num_query_rounds=12)
    trace, pis, _ = b2.generate_trace(9, BLAKE_MESSAGES)
    expect = oracle_lib.stark_prove(oracle, stark, trace, pis)
    monkeypatch.setenv("VX_HASH_PIPELINE_MIN_BYTES", "0")
    ctx.prof_enable(True)
    ctx.prof_reset()
    assert stark.prove(ctx, trace, pis) == expect
    calls = ctx.prof()["hash_leaves"]["calls"]
    ctx.prof_enable(False)
    assert calls == 12 + 1 + 1, calls           # the 1063-column trace in twelve pieces, then the aux and the quotient batches in one each
    monkeypatch.setenv("VX_NO_UPLOAD_OVERLAP", "1")
    assert stark.prove(ctx, trace, pis) == expect


def test_blake2b_air_interpreted_equals_compiled(ctx):
    import os
    from vectorx_amd import blake2b_air as b2
    stark = b2.make_stark(9, pow_bits=6, num_query_rounds=12)
    trace, pis, _ = b2.generate_trace(9, BLAKE_MESSAGES)
    expect = stark.prove(ctx, trace, pis)
    os.environ["VX_NO_JIT"] = "1"
    try:
        ctx.prof_enable(True)
        ctx.prof_reset()
        assert stark.prove(ctx, trace, pis) == expect
        assert "air_quotient_eval" in ctx.prof()
    finally:
        del os.environ["VX_NO_JIT"]
        ctx.prof_enable(False)


def test_three_tables_on_one_bus_bytes_identical_to_oracle(ctx, oracle):
    """SHA-256 and BLAKE2b tables SEND their digests, one sink RECEIVES them all: three sessions, joint challenges over the three
    trace caps, three proofs byte-identical to the oracle's, closing sums cancel."""
    from vectorx_amd import blake2b_air as b2
    from vectorx_amd import sha256_air as sha
    from vectorx_amd import stark_bus
    cfg = dict(num_query_rounds=16, pow_bits=6)
    sha_stark = sha.make_stark(9, bus=True, **cfg)
    st, spis, sdig = sha.generate_trace(9, SHA_MESSAGES)
    bl_stark = b2.make_stark(10, bus=True, **cfg)
    bt, bpis, bdig = b2.generate_trace(10, BLAKE_MESSAGES)
    rows = [np.frombuffer(d, dtype=">u4").astype(np.uint64) for d in sdig] + [np.array(b2.digest_limbs(d), dtype=np.uint64) for d in bdig]
    sink_stark, sink_t, _ = sha.make_sink(5, [b"\0" * 32] * len(rows), **cfg)
    for i, r in enumerate(rows):
        sink_t[:8, i] = r
    sink_pis = sink_t[:8, 0].copy()
    tables = [(sha_stark, st, spis), (bl_stark, bt, bpis), (sink_stark, sink_t, sink_pis)]
    proofs, shared = stark_bus.prove_tables(ctx, tables)
    expect, shared_o = oracle_lib.stark_prove_tables(oracle, tables)
    assert (shared == shared_o).all() and list(proofs) == list(expect)
    sums = stark_bus.verify_bus([(s, p) for s, _, p in tables], proofs)
    assert sum(int(s[0]) for s in sums) % P == 0 and all(int(s[0]) for s in sums)


# ---- a third chip: Ed25519 scalar multiplication as non-native field arithmetic (vectorx_amd/ed25519_air.py — 617 + 96 columns, one
#      multiply-add mod 2^255 - 19 per row over byte limbs, 188 byte lookups per row through 94 helper columns); own AIR ----------------
@pytest.mark.parametrize("degree_bits,scalar,cfg", [
    (10, 0xC0FFEE11, {}), (11, 0x0123456789ABCDEF, dict(rate_bits=2, num_query_rounds=30)),
    (13, ((int.from_bytes(__import__("hashlib").sha512(bytes.fromhex("9d61b19deffd5a60ba844af492ec2cc44449c5697b326919703bac031cae7f60")).digest()[:32],
                          "little") & ((1 << 254) - 8)) | (1 << 254)), dict(num_query_rounds=40))])
def test_ed25519_air_proof_bytes_identical_to_oracle(ctx, oracle, degree_bits, scalar, cfg):
    from vectorx_amd import ed25519_air as ed
    cfg = dict(dict(pow_bits=8, num_query_rounds=20), **cfg)
    stark = ed.make_stark(degree_bits, **cfg)
    trace, pis, pt = ed.generate_trace(degree_bits, scalar)
    assert pt == ed.affine_scalar_mult(scalar)
    if degree_bits == 13:                                               # RFC 8032 section 7.1 test 1: the public key of that secret key
        assert ed.compress(pt).hex() == "d75a980182b10ab7d54bfed3c964073a0ee172f3daa62325af021a68f707511a"
    ctx.prof_enable(True)
    ctx.prof_reset()
    gp = stark.prove(ctx, trace, pis)
    stages = ctx.prof()
    ctx.prof_enable(False)
    assert "air_quotient_eval_jit" in stages, sorted(stages)
    assert gp == oracle_lib.stark_prove(oracle, stark, trace, pis)
    stark.verify(pis, gp)
    wrong = pis.copy()
    wrong[24] = (int(wrong[24]) + 1) % P                                # another y
    with pytest.raises(vx.VxError):
        stark.verify(wrong, gp)
    bad = trace.copy()                                                   # one quotient byte off: refused or rejected
    bad[ed.Cols.Q + 3, 32 * 5 + 10] = (int(bad[ed.Cols.Q + 3, 32 * 5 + 10]) + 1) % 256
    try:
        bp = stark.prove(ctx, bad, pis)
    except vx.VxError:
        bp = None
    if bp is not None:
        with pytest.raises(vx.VxError):
            stark.verify(pis, bp)


def test_ed25519_air_interpreted_equals_compiled(ctx):
    import os
    from vectorx_amd import ed25519_air as ed
    stark = ed.make_stark(10, pow_bits=6, num_query_rounds=12)
    trace, pis, _ = ed.generate_trace(10, 0x9E3779B9)
    expect = stark.prove(ctx, trace, pis)
    os.environ["VX_NO_JIT"] = "1"
    try:
        ctx.prof_enable(True)
        ctx.prof_reset()
        assert stark.prove(ctx, trace, pis) == expect
        assert "air_quotient_eval" in ctx.prof()
    finally:
        del os.environ["VX_NO_JIT"]
        ctx.prof_enable(False)


def test_sha512_air_proof_bytes_identical_to_oracle_and_to_the_golden(ctx, oracle):
    """the fourth chip (vectorx_amd/sha512_air.py: 1995 + 5 columns, 4037 constraints, a 31.7 k-word program in 25 chunks): GPU proof ==
    oracle proof == the digest frozen in tests/golden/chip_goldens.json (same inputs and configuration as the generator)"""
    import hashlib
    import json
    from pathlib import Path

    from vectorx_amd import sha512_air as s5
    stark = s5.make_stark(8, num_query_rounds=12, pow_bits=4)
    trace, pis, digests = s5.generate_trace(8, [b"abc", b"", bytes(range(200))])
    assert digests[0] == hashlib.sha512(b"abc").digest()
    gp = stark.prove(ctx, trace, pis)
    assert gp == oracle_lib.stark_prove(oracle, stark, trace, pis)
    gold = json.loads((Path(__file__).resolve().parent / "golden" / "chip_goldens.json").read_text())["sha512"]
    assert hashlib.sha256(gp).hexdigest() == gold["oracle_proof_sha256"] and len(gp) == gold["oracle_proof_bytes"]
    stark.verify(pis, gp)
    wrong = pis.copy()
    wrong[0] = (int(wrong[0]) + 1) % P
    with pytest.raises(vx.VxError):
        stark.verify(wrong, gp)


def test_two_tables_on_one_bus_bytes_identical_to_oracle(ctx, oracle):
    """vx_stark_begin x 2 -> joint challenges over both trace caps -> vx_stark_set_aux_challenges -> vx_stark_finish2 with the
    closing sums (vectorx_amd/stark_bus.py): both proofs byte-identical to the oracle's, the bus balances, a proof moved into
    another bus (other joint challenges) is refused."""
    import hashlib
    from vectorx_amd import sha256_air as sha
    from vectorx_amd import stark_bus
    cfg = dict(num_query_rounds=16, pow_bits=6)
    sha_stark = sha.make_stark(9, bus=True, **cfg)
    t, pis, digests = sha.generate_trace(9, SHA_MESSAGES)
    assert len(digests) >= 3
    sink_stark, sink_t, sink_pis = sha.make_sink(5, digests, **cfg)
    tables = [(sha_stark, t, pis), (sink_stark, sink_t, sink_pis)]
    proofs, shared = stark_bus.prove_tables(ctx, tables)
    expect, shared_o = oracle_lib.stark_prove_tables(oracle, tables)
    assert (shared == shared_o).all()
    assert proofs[0] == expect[0] and proofs[1] == expect[1]
    sums = stark_bus.verify_bus([(sha_stark, pis), (sink_stark, sink_pis)], proofs)
    assert (int(sums[0][0]) + int(sums[1][0])) % P == 0 and int(sums[0][0]) != 0
    other_sink, other_t, other_pis = sha.make_sink(5, digests[:-1] + [hashlib.sha256(b"x").digest()], **cfg)
    proofs2, _ = stark_bus.prove_tables(ctx, [(sha_stark, t, pis), (other_sink, other_t, other_pis)])
    with pytest.raises(vx.VxError):
        stark_bus.verify_bus([(sha_stark, pis), (other_sink, other_pis)], proofs2)          # valid proofs, unbalanced bus
    with pytest.raises(vx.VxError):
        stark_bus.verify_tables([(sha_stark, pis), (sink_stark, sink_pis)], [proofs2[0], proofs[1]])   # a proof from another bus


@pytest.mark.parametrize("limb_bits,scalar_bits,degree_bits,nsigs", [(8, 32, 12, 2), (8, 64, 13, 2), (16, 32, 17, 40)])
def test_batched_eddsa_table_and_its_sink_bytes_identical_to_oracle(ctx, oracle, limb_bits, scalar_bits, degree_bits, nsigs):
    """vectorx_amd/eddsa_air.py: many signature equations per trace, results on a bus.  Table + sink through
    vx_stark_begin / set_aux_challenges / finish2: both proofs byte-identical to the oracle's, vx_stark_verify_bus accepts, a sink
    with another R leaves the bus unbalanced.  (16, 32, 17): the 65 536-entry limb table of the production shape, 40 instances."""
    import random

    from vectorx_amd import eddsa_air as ea
    from vectorx_amd import stark_bus
    lay = ea.Layout(limb_bits, scalar_bits)
    rng = random.Random(degree_bits)
    base = [(ea.affine_scalar_mult(rng.randrange(1, ea.ELL)), rng.randrange(1 << scalar_bits), rng.randrange(1 << scalar_bits)) for _ in range(min(nsigs, 4))]
    sigs = [base[i % len(base)] for i in range(nsigs)]
    cfg = dict(num_query_rounds=12, pow_bits=5)
    stark = ea.make_stark(lay, degree_bits, **cfg)
    t, res = ea.generate_trace(lay, degree_bits, sigs)
    assert res[:len(base)] == [ea.reference_result(a, s, h) for a, s, h in base]
    nopi = np.zeros(0, dtype=np.uint64)
    honest = [ea.tuple_of(lay, a, s, h, r) for (a, s, h), r in zip(sigs, res)]
    sink, sink_t, _ = ea.make_sink(lay, honest, **cfg)
    tables = [(stark, t, nopi), (sink, sink_t, nopi)]
    proofs, shared = stark_bus.prove_tables(ctx, tables)
    expect, shared_o = oracle_lib.stark_prove_tables(oracle, tables)
    assert (shared == shared_o).all() and proofs[0] == expect[0] and proofs[1] == expect[1]
    sums = vx.stark_verify_bus([(stark, nopi), (sink, nopi)], proofs)
    assert int(sums[0][0]) != 0 and int(sums[0][1]) != 0
    forged = [list(tp) for tp in honest]
    forged[-1][-1] ^= 1
    sink2, sink2_t, _ = ea.make_sink(lay, forged, **cfg)
    proofs2, _ = stark_bus.prove_tables(ctx, [(stark, t, nopi), (sink2, sink2_t, nopi)])
    with pytest.raises(vx.VxError, match="cancel"):
        vx.stark_verify_bus([(stark, nopi), (sink2, nopi)], proofs2)


@pytest.mark.parametrize("limb_bits,degree_bits", [(8, 14), (16, 17)])
def test_full_eddsa_table_verifies_signatures_from_their_bytes_and_is_byte_identical_to_the_oracle(ctx, oracle, limb_bits, degree_bits):
    """The FULL program (round 5; Layout(full=True)): point decompression of A and R, h = SHA-512 digest mod L and S < L
    INSIDE the table — the bus tuple is the verifier's bytes (public key, S, digest, R's encoding).  RFC 8032 section 7.1 signatures,
    table + sink on the GPU: both proofs byte-identical to the oracle's, the bus balances; with S + L, a flipped sign bit of R or another
    digest in the verifier's bytes it does not.  (16, 17) = the production layout: 16-bit limbs, 12 instances per 2^17 rows."""
    import hashlib

    from test_eddsa_air import RFC8032
    from vectorx_amd import eddsa_air as ea
    from vectorx_amd import stark_bus
    lay = ea.Layout(limb_bits, 256, full=True)
    raw = []
    for sk, pk, msg, sig in RFC8032[:ea.capacity(lay, degree_bits)]:
        pk, msg, sig = bytes.fromhex(pk), bytes.fromhex(msg), bytes.fromhex(sig)
        raw.append((pk, msg, sig, hashlib.sha512(sig[:32] + pk + msg).digest()))
    sigs = [ea.equation_inputs_full(pk, msg, sig) for pk, msg, sig, _ in raw]
    cfg = dict(num_query_rounds=12, pow_bits=5)
    stark = ea.make_stark(lay, degree_bits, **cfg)
    t, res = ea.generate_trace(lay, degree_bits, sigs)
    assert res == [ea.decompress(sig[:32]) for _, _, sig, _ in raw]
    nopi = np.zeros(0, dtype=np.uint64)
    honest = [ea.tuple_of_full(lay, pk, sig, dig) for pk, _, sig, dig in raw]
    sink, sink_t, _ = ea.make_sink(lay, honest, **cfg)
    tables = [(stark, t, nopi), (sink, sink_t, nopi)]
    proofs, shared = stark_bus.prove_tables(ctx, tables)
    expect, shared_o = oracle_lib.stark_prove_tables(oracle, tables)
    assert (shared == shared_o).all() and proofs[0] == expect[0] and proofs[1] == expect[1]
    sums = vx.stark_verify_bus([(stark, nopi), (sink, nopi)], proofs)
    assert int(sums[0][0]) != 0 and int(sums[0][1]) != 0
    pk, _, sig, dig = raw[-1]
    s = int.from_bytes(sig[32:], "little")
    for forged_last in (ea.tuple_of_full(lay, pk, sig[:32] + (s + ea.ELL).to_bytes(32, "little"), dig),                       # S + L
                        ea.tuple_of_full(lay, pk, bytes(sig[:31]) + bytes([sig[31] ^ 0x80]) + sig[32:], dig),                # R's sign bit
                        ea.tuple_of_full(lay, pk, sig, hashlib.sha512(b"another message").digest())):                         # a wrong h
        sink2, sink2_t, _ = ea.make_sink(lay, honest[:-1] + [forged_last], **cfg)
        proofs2, _ = stark_bus.prove_tables(ctx, [(stark, t, nopi), (sink2, sink2_t, nopi)])
        with pytest.raises(vx.VxError, match="cancel"):
            vx.stark_verify_bus([(stark, nopi), (sink2, nopi)], proofs2)


def test_signatures_verify_through_tables_only_on_the_gpu(ctx, oracle):
    """The whole signature bus on the GPU (round 5): SHA-512 table (bus variant: sends R, A, digest) + EdDSA table (full
    program, production layout: sends A, S, digest, R) + link table (joins them on the digest, sends A, S, R) + a verifier's sink holding
    the bytes of the RFC 8032 section 7.1 public keys and signatures.  All four proofs byte-identical to the oracle's, the bus balances; with
    S + L in the verifier's bytes it does not."""
    import hashlib

    from test_eddsa_air import RFC8032
    from vectorx_amd import eddsa_air as ea
    from vectorx_amd import sha512_air as s5
    from vectorx_amd import sig_link_air as link
    from vectorx_amd import stark_bus
    cfg = dict(num_query_rounds=12, pow_bits=5)
    raw = []
    for _, pk, msg, sig in RFC8032:
        pk, msg, sig = bytes.fromhex(pk), bytes.fromhex(msg), bytes.fromhex(sig)
        raw.append((pk, msg, sig, hashlib.sha512(sig[:32] + pk + msg).digest()))
    lay = ea.Layout(16, 256, full=True)
    nopi = np.zeros(0, dtype=np.uint64)
    sha = s5.make_stark(9, bus=True, **cfg)
    sha_t, sha_pis, digs = s5.generate_trace(9, [sig[:32] + pk + msg for pk, msg, sig, _ in raw], bus=True)
    assert digs == [d for _, _, _, d in raw]
    ed = ea.make_stark(lay, 17, **cfg)
    ed_t, res = ea.generate_trace(lay, 17, [ea.equation_inputs_full(pk, msg, sig) for pk, msg, sig, _ in raw])
    lk, lk_t, _ = link.make_link([link.row_of(pk, sig, dig) for pk, _, sig, dig in raw], **cfg)
    honest = [link.verifier_tuple(pk, sig) for pk, _, sig, _ in raw]
    sink, sink_t, _ = ea.make_sink(lay, honest, ntuple=25, **cfg)
    tables = [(sha, sha_t, sha_pis), (ed, ed_t, nopi), (lk, lk_t, nopi), (sink, sink_t, nopi)]
    descs = [(sha, sha_pis), (ed, nopi), (lk, nopi), (sink, nopi)]
    proofs, shared = stark_bus.prove_tables(ctx, tables)
    expect, shared_o = oracle_lib.stark_prove_tables(oracle, tables)
    assert (shared == shared_o).all() and proofs == expect
    sums = vx.stark_verify_bus(descs, proofs)
    assert all(int(x) != 0 for x in sums[:, 0])
    # the same bus with every trace resident in HBM and every second round computed on the GPU (stark_chips.prove_bus_device — what the
    # DAG's outer job runs; the SHA-512 and EdDSA traces generated on the device): the same four proofs
    from vectorx_amd import stark_chips
    bufs, items = [], []
    try:
        for st, tr, pi in tables:
            n_ = tr.shape[1]
            d_t, d_a = ctx.alloc(tr.shape[0] * n_ * 8), ctx.alloc(st.desc.num_aux_columns * n_ * 8)
            bufs += [d_t, d_a]
            ctx.upload(d_t, tr)
            items.append((st, d_t, pi, d_a))
        ctx.trace_hash_table("sha512_bus", 9, [sig[:32] + pk + msg for pk, msg, sig, _ in raw], items[0][1])
        assert ctx.trace_eddsa_table(17, 256, [ea.equation_inputs_full(pk, msg, sig) for pk, msg, sig, _ in raw], items[1][1], full=True) == res
        dev_proofs, dev_shared, dev_sums = stark_chips.prove_bus_device(ctx, items)
        assert (dev_shared == shared).all() and dev_proofs == proofs
        assert [[int(x) for x in a] for a in dev_sums] == [[int(x) for x in row] for row in sums]
    finally:
        for d_ in bufs:
            ctx.free(d_)
    pk, _, sig, dig = raw[-1]
    s = int.from_bytes(sig[32:], "little")
    forged_sig = sig[:32] + (s + ea.ELL).to_bytes(32, "little")
    sink2, sink2_t, _ = ea.make_sink(lay, honest[:-1] + [link.verifier_tuple(pk, forged_sig)], ntuple=25, **cfg)
    lk2, lk2_t, _ = link.make_link([link.row_of(p_, s_, d_) for p_, _, s_, d_ in raw[:-1]] + [link.row_of(pk, forged_sig, dig)], **cfg)
    proofs2, _ = stark_bus.prove_tables(ctx, tables[:2] + [(lk2, lk2_t, nopi), (sink2, sink2_t, nopi)])
    with pytest.raises(vx.VxError, match="cancel"):
        vx.stark_verify_bus(descs[:2] + [(lk2, nopi), (sink2, nopi)], proofs2)


def test_batched_eddsa_interpreted_equals_compiled(ctx):
    import os

    from vectorx_amd import eddsa_air as ea
    lay = ea.Layout(8, 32)
    stark = ea.make_stark(lay, 12, pow_bits=6, num_query_rounds=12)
    t, _ = ea.generate_trace(lay, 12, [((ea.BX, ea.BY), 0xC0FFEE, 0xBADF00D)])
    nopi = np.zeros(0, dtype=np.uint64)
    expect = stark.prove(ctx, t, nopi)
    os.environ["VX_NO_JIT"] = "1"
    try:
        assert stark.prove(ctx, t, nopi) == expect
    finally:
        del os.environ["VX_NO_JIT"]


def test_blake2b_bytes_table_bytes_identical_to_oracle(ctx, oracle):
    """vectorx_amd/blake2b_bytes_air.py (round 4: bytes + a XOR lookup in two 32 768-entry half tables, four G functions per row, 775 + 238
    columns): the table cannot be smaller than 2^16 rows, so this is the one proof of it against the oracle — here, where the oracle has the GPU
    box's 16 cores.  GPU proof == oracle proof; the product's host verifier accepts it and rejects another digest / a flipped byte."""
    import hashlib

    from vectorx_amd import blake2b_bytes_air as b2
    msgs = [b"abc", b"", bytes(range(200)), b"y" * 129, bytes([7]) * 5000]
    stark = b2.make_stark(16, num_query_rounds=10, pow_bits=4)
    t, pis, digests = b2.generate_trace(16, msgs)
    assert digests == [hashlib.blake2b(m, digest_size=32).digest() for m in msgs]
    gp = stark.prove(ctx, t, pis)
    assert gp == oracle_lib.stark_prove(oracle, stark, t, pis)
    stark.verify(pis, gp)
    wrong = pis.copy()
    wrong[3] = (int(wrong[3]) + 1) % P
    with pytest.raises(vx.VxError):
        stark.verify(wrong, gp)
    bad = bytearray(gp)
    bad[len(bad) // 3] ^= 1
    with pytest.raises(vx.VxError):
        stark.verify(pis, bytes(bad))


def test_second_round_columns_on_the_gpu_equal_the_host_ones(ctx):
    """vx_stark_aux_columns (round 4): the fractions + running sums of a table's second round computed on the device from the resident
    trace — bit-identical to the numpy generators (every column of both challenge sets, and the closing sums) for the batched EdDSA
    table (pair helpers, table term, bus term; two running sums) and the byte BLAKE2b table (184 triples with beta, one sum)."""
    from vectorx_amd import blake2b_bytes_air as b2
    from vectorx_amd import eddsa_air as ea
    lay = ea.Layout(8, 32)
    st = ea.make_stark(lay, 12)
    t, _ = ea.generate_trace(lay, 12, [((ea.BX, ea.BY), 0xC0FFEE, 0xBADF00D), ((ea.BX, ea.BY), 5, 7)])
    cases = [(st, t)]
    st2 = b2.make_stark(16)
    t2, _, _ = b2.generate_trace(16, [b"abc", bytes(range(200))])
    cases.append((st2, t2))
    from vectorx_amd import sha256_air as sha
    from vectorx_amd import sha512_air as s5
    cases.append((sha.make_stark(9), sha.generate_trace(9, SHA_MESSAGES)[0]))
    cases.append((s5.make_stark(8), s5.generate_trace(8, [b"abc", b""])[0]))
    rng = np.random.default_rng(7)
    for stark, trace in cases:
        chal = rng.integers(1, P, size=stark.desc.num_aux_challenges, dtype=np.uint64)
        want, want_api = stark.run_aux(trace, chal)
        d_t = ctx.alloc(trace.nbytes)
        ctx.upload(d_t, np.ascontiguousarray(trace))
        d_a = ctx.alloc(want.nbytes)
        api = stark.run_aux_gpu(ctx, d_t, chal, d_a)
        got = ctx.download(d_a, want.nbytes).view(np.uint64).reshape(want.shape)
        assert (got == want).all(), np.nonzero((got != want).any(axis=1))[0][:8]
        assert [int(x) for x in api] == [int(x) for x in want_api]
        ctx.free(d_t)
        ctx.free(d_a)


def test_aux_program_descriptions_are_checked(ctx):
    import ctypes
    from vectorx_amd import eddsa_air as ea
    ap = ea.aux_program(ea.Layout(8, 32))
    L = vx.lib()
    d_t = ctx.alloc(8 * 16 * 900)
    d_o = ctx.alloc(8 * 16 * 100)
    closing = np.zeros(4, dtype=np.uint64)
    chal = np.ones(3, dtype=np.uint64)
    for field, value in (("num_fractions", ap.desc.num_fractions + 1), ("num_columns", 5), ("num_challenges", 1), ("num_sums", 65), ("program_len", 3)):
        old = getattr(ap.desc, field)
        setattr(ap.desc, field, value)
        rc = L.vx_stark_aux_columns(ctx._h, ctypes.byref(ap.desc), ctypes.c_void_p(d_t), 4, chal.ctypes.data, ctypes.c_void_p(d_o), closing.ctypes.data)
        assert rc == vx.VX_E_INVALID, field
        setattr(ap.desc, field, old)
    ctx.free(d_t)
    ctx.free(d_o)


def test_lanes_proving_the_same_resident_table_concurrently(ctx):
    """the DAG scheduler's lanes prove ONE resident table from several host threads (vectorx_amd/stark_chips.py::ResidentTable): the
    second round is computed on the GPU per proof with the running sums scanned in place, so every lane needs its own buffer — a shared
    one made the proofs of a pass differ (found by the DAG leg's root check in round 4)"""
    import threading

    from vectorx_amd import eddsa_air as ea
    from vectorx_amd import stark_chips
    lay = ea.Layout(8, 32)
    stark = ea.make_stark(lay, 12, num_query_rounds=12, pow_bits=4)
    t, _ = ea.generate_trace(lay, 12, [((ea.BX, ea.BY), 0xC0FFEE, 0xBADF00D), ((ea.BX, ea.BY), 5, 7), ((ea.BX, ea.BY), 11, 13)])
    tab = stark_chips.ResidentTable(ctx, stark, t, np.zeros(0, dtype=np.uint64), "eddsa-small")
    lanes = [vx.Context(0) for _ in range(3)]
    try:
        want = tab.prove()
        assert want == stark.prove(ctx, t, np.zeros(0, dtype=np.uint64))      # GPU second round == host second round, through whole proofs
        got, errs = [[] for _ in lanes], []

        def run(k):
            try:
                for _ in range(6):
                    got[k].append(tab.prove(lanes[k]))
            except BaseException as e:   # noqa: BLE001
                errs.append(e)
        threads = [threading.Thread(target=run, args=(k,)) for k in range(len(lanes))]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        assert not errs, errs
        assert all(p == want for g in got for p in g)
    finally:
        tab.free()
        for l in lanes:
            l.close()
