#!/usr/bin/env python3
"""One-off soak of the STARK path: random AIRs / sizes / configurations proven on the GPU (compiled and interpreted AIR
programs alternating) and by the oracle; proofs must be byte-identical, vx_stark_verify must accept them and reject a bit flip; a third
of the cases are proven again sharded by coset over 2 .. 2^rate_bits ranks (threads, one device) and every rank's proof must be the same bytes.

    python tools/soak_stark.py [seconds] [seed] [max_degree_bits] > gpurun_out/soak_stark.jsonl
"""
import faulthandler
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib  # noqa: E402  (checker only)
import vectorx_amd as vx  # noqa: E402
from stark_airs import cubic, fibonacci, logup, mulchain  # noqa: E402
from vectorx_amd import blake2b_air, ed25519_air, sha256_air, sharded  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
db_max = int(sys.argv[3]) if len(sys.argv) > 3 else 13
oracle = oracle_lib.load()
ctx = vx.Context(0)
rank_ctxs = []                                   # round 6: contexts of the ranks of a proof sharded by coset (made on first use, kept)
t_end = time.time() + budget
n_ok, n_bad, by_kind = 0, 0, {}
while time.time() < t_end:
    kind = str(rng.choice(["fibonacci", "cubic", "mulchain", "logup", "logup", "sha256", "blake2b", "ed25519", "eddsa"]))
    lg = int(rng.integers(5 if kind == "logup" else 3, db_max + 1))
    if kind == "sha256":
        lg = int(rng.integers(7, min(db_max, 10) + 1))
    if kind == "blake2b":
        lg = int(rng.integers(9, max(min(db_max, 10), 9) + 1))
    rate_bits = int(rng.choice([1, 1, 2, 3]))
    cfg = dict(rate_bits=rate_bits, pow_bits=int(rng.choice([0, 4, 8])), num_query_rounds=int(rng.choice([5, 20, 84])),
               num_challenges=int(rng.choice([1, 2])), cap_height=int(rng.integers(0, min(4, lg + rate_bits) + 1)))
    if kind == "ed25519":
        lg = int(rng.integers(10, 12))
    if kind == "eddsa":
        lg = 12
    if kind in ("sha256", "blake2b", "ed25519", "eddsa"):      # one compiled program shape (2 challenges); the rest of the configuration varies
        cfg["num_challenges"] = 2
        cfg["rate_bits"] = rate_bits = int(rng.choice([1, 2]))
    if rng.random() < 0.4:                       # round 5: the transcript over a tree hash of the openings (VX_STARK_OPENINGS_DIGEST)
        cfg["openings_digest"] = True
    if rng.random() < 0.3:
        ar, left = [], lg
        while left > 0 and len(ar) < 5 and rng.random() < 0.8:
            a = int(rng.integers(1, min(4, left) + 1))
            if lg + rate_bits - sum(ar) - a < cfg["cap_height"]:
                break
            ar.append(a)
            left -= a
        cfg["fri_arities"] = ar
    if kind == "fibonacci":
        stark, trace, pis = fibonacci(lg, x0=int(rng.integers(0, 1 << 40)), x1=int(rng.integers(1, 1 << 40)), **cfg)
    elif kind == "cubic":
        stark, trace, pis = cubic(lg, seed=int(rng.integers(1, 1 << 40)), **cfg)
    elif kind == "mulchain":
        stark, trace, pis = mulchain(lg, groups=int(rng.integers(1, 7)), seed=int(rng.integers(1, 1 << 20)), **cfg)
    elif kind == "sha256":
        msgs = [bytes(rng.integers(0, 256, size=int(rng.integers(0, 200)), dtype=np.uint8)) for _ in range(6)]
        stark = sha256_air.make_stark(lg, **cfg)
        trace, pis, _ = sha256_air.generate_trace(lg, msgs)
    elif kind == "ed25519":
        stark = ed25519_air.make_stark(lg, **cfg)
        trace, pis, _ = ed25519_air.generate_trace(lg, int(rng.integers(0, 1 << 32)) | (int(rng.integers(0, 1 << 32)) << 32 if lg == 11 else 0))
    elif kind == "eddsa":     # round 4: the batched signature table (byte limbs, 32-bit scalars: 3 instances fit 2^12 rows)
        from vectorx_amd import eddsa_air
        lay = eddsa_air.Layout(8, 32)
        sigs = [(eddsa_air.affine_scalar_mult(int(rng.integers(1, 1 << 62))), int(rng.integers(0, 1 << 32)), int(rng.integers(0, 1 << 32)))
                for _ in range(int(rng.integers(1, 4)))]
        stark = eddsa_air.make_stark(lay, lg, **cfg)
        trace, _res = eddsa_air.generate_trace(lay, lg, sigs)
        pis = np.zeros(0, dtype=np.uint64)
    elif kind == "blake2b":
        msgs = [bytes(rng.integers(0, 256, size=int(rng.integers(0, 300)), dtype=np.uint8)) for _ in range(6)]
        stark = blake2b_air.make_stark(lg, **cfg)
        trace, pis, _ = blake2b_air.generate_trace(lg, msgs)
    else:
        stark, trace, pis = logup(lg, table_bits=int(rng.integers(2, min(lg - 1, 6) + 1)), seed=int(rng.integers(1, 1 << 20)), **cfg)
    jit = bool(rng.random() < 0.7)
    print(json.dumps({"case": n_ok + n_bad, "kind": kind, "degree_bits": lg, "jit": jit, **{k: v for k, v in cfg.items()}}), file=sys.stderr, flush=True)
    faulthandler.dump_traceback_later(180, exit=True)
    if not jit:
        os.environ["VX_NO_JIT"] = "1"
    try:
        got = stark.prove(ctx, trace, pis)
    finally:
        os.environ.pop("VX_NO_JIT", None)
    want = oracle_lib.stark_prove(oracle, stark, trace, pis)
    ok = got == want
    world = 1
    max_world_bits = min(rate_bits, cfg["cap_height"])          # vx_stark_begin_sharded: a rank owns whole cosets and whole cap subtrees
    if max_world_bits >= 1 and rng.random() < 0.5:  # round 6: the same proof over 2 .. 2^max_world_bits ranks (threads, one device)
        world = 1 << int(rng.integers(1, max_world_bits + 1))
        while len(rank_ctxs) < world:
            rank_ctxs.append(vx.Context(0))
        faulthandler.cancel_dump_traceback_later()
        faulthandler.dump_traceback_later(180, exit=True)
        if not jit:
            os.environ["VX_NO_JIT"] = "1"
        try:
            ok = ok and all(p == got for p in sharded.prove_stark_sharded_threads(rank_ctxs[:world], stark, trace, pis, timeout_ms=120_000))
        finally:
            os.environ.pop("VX_NO_JIT", None)
    try:
        stark.verify(pis, got)
    except vx.VxError:
        ok = False
    bad = bytearray(got)
    bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
    try:
        stark.verify(pis, bytes(bad))
        ok = False
    except vx.VxError:
        pass
    faulthandler.cancel_dump_traceback_later()
    key = f"{kind}/rate{rate_bits}" + ("" if jit else "/interpreted") + ("/A" if "fri_arities" in cfg else "") + ("/D" if cfg.get("openings_digest") else "") + (f"/G{world}" if world > 1 else "")
    by_kind[key] = by_kind.get(key, 0) + 1
    if ok:
        n_ok += 1
    else:
        n_bad += 1
        print(json.dumps({"FAIL": {"kind": kind, "degree_bits": lg, "jit": jit, "world": world, **cfg}}), flush=True)
print(json.dumps({"stark_cases": n_ok + n_bad, "identical_and_verified": n_ok, "failures": n_bad, "seconds": budget, "max_degree_bits": db_max,
                  "sharded_cases": sum(v for k, v in by_kind.items() if "/G" in k), "by_kind": by_kind}), flush=True)
