#!/usr/bin/env python3
"""One-off soak: random circuits (size, gate mix, flags, config variants, rank counts) proven on the GPU and by the
oracle; proofs must be byte-identical, `vx_verify` must accept them and reject a random bit flip.

    python tools/soak_differential.py [seconds] [seed] [min_degree_bits] [max_degree_bits] > gpurun_out/soak.jsonl
"""
import faulthandler
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib  # noqa: E402  (checker only)
import vectorx_amd as vx  # noqa: E402
from vectorx_amd import sharded  # noqa: E402
from vectorx_amd.synth import SynthCircuit  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
db_min = int(sys.argv[3]) if len(sys.argv) > 3 else 3
db_max = int(sys.argv[4]) if len(sys.argv) > 4 else 13
oracle = oracle_lib.load()
oracle.L.vxo_set_num_threads(16 if db_max > 13 else 8)
ctx = vx.Context(0)
lanes = [vx.Context(0) for _ in range(7)]
t_end = time.time() + budget
n_ok, n_bad, by_kind = 0, 0, {}
while time.time() < t_end:
    flags = int(rng.choice([0, 0, 1, 2, 3, 4, 5, 7, 8, 15, 16, 16, 17, 19, 21, 31, 32, 32, 33, 39, 48, 63]))   # 16 = lookup table, 32 = U32 / comparison gates
    lo = 5 if flags & 60 else (4 if flags & 1 else 3)
    db = int(rng.integers(max(lo, db_min), db_max + 1))
    pct = int(rng.integers(0, 101))
    # quotient_degree_factor below the blow-up (CircuitConfig::max_quotient_degree_factor): the gate families that fit
    qdf = 8
    if rng.random() < 0.25:
        qdf = int(rng.integers(3, 8))
        if qdf < 8:
            flags &= ~8                  # CosetInterpolationGate has degree 8
        if qdf < 5:
            flags &= ~(4 | 32)           # RandomAccess (degree 5) / Exponentiation (4) family; the U32 gates (degree 4) need qdf >= 5
    mix = None
    if qdf == 8 and rng.random() < 0.15:     # round 6: the DAG's circuit family — the recursive verifier's gate set in its declared row mix
        from vectorx_amd.synth import RECURSION_FLAGS, RECURSIVE_VERIFIER_MIX
        flags, mix, db = RECURSION_FLAGS, RECURSIVE_VERIFIER_MIX, max(db, 5)
    sc = SynthCircuit(db, seed=int(rng.integers(1, 1 << 30)), poseidon_percent=pct, witness_seed=int(rng.integers(1, 1 << 30)), flags=flags,
                      quotient_degree_factor=qdf, mix=mix)
    sc.desc.pow_bits = int(rng.choice([0, 3, 8, 12]))
    if rng.random() < 0.3:
        sc.desc.num_challenges = 1
    if rng.random() < 0.3:
        sc.desc.cap_height = int(rng.choice([0, 1, 2, 3]))
    if rng.random() < 0.3:
        sc.desc.num_query_rounds = int(rng.choice([1, 5, 40]))
    # values the caller holds (vx_circuit_desc tail, round 2): a caller-supplied digest / FRI arity list / num_partial_products
    overrides = ""
    if rng.random() < 0.25:
        sc.set_circuit_digest([int(x) for x in rng.integers(0, 1 << 62, 4)])
        overrides += "D"
    if rng.random() < 0.25:
        ar, left = [], db
        while left > 0 and len(ar) < 6 and rng.random() < 0.8:
            a = int(rng.integers(1, min(4, left) + 1))
            if db + 3 - sum(ar) - a < sc.desc.cap_height:
                break
            ar.append(a)
            left -= a
        sc.set_fri_reduction_arity_bits(ar)
        overrides += "A"
    if rng.random() < 0.25:
        sc.set_num_partial_products((80 + qdf - 1) // qdf - 1)
        overrides += "P"
    world = int(rng.choice([1, 1, 2, 4, 8]))
    if world > (1 << sc.desc.cap_height):
        world = 1 << sc.desc.cap_height
    # progress on stderr + a watchdog: a case that stalls dumps every thread's Python stack and ends the run (rc != 0)
    print(json.dumps({"case": n_ok + n_bad, "degree_bits": db, "flags": flags, "recursion_mix": mix is not None, "pct": pct, "qdf": qdf, "world": world, "overrides": overrides,
                      "pow_bits": int(sc.desc.pow_bits), "nch": int(sc.desc.num_challenges), "cap_height": int(sc.desc.cap_height),
                      "queries": int(sc.desc.num_query_rounds)}), file=sys.stderr, flush=True)
    faulthandler.dump_traceback_later(300 if db_max > 13 else 120, exit=True)
    w = sc.witness()
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    want = oc.prove(w)
    cs = [vx.Circuit(c, sc.desc_ptr) for c in [ctx] + lanes[:world - 1]]
    got = sharded.prove_sharded_threads(cs, w) if world > 1 else [cs[0].prove(w)]
    ok = all(p == want for p in got)
    try:
        cs[0].verify(got[0])
    except vx.VxError:
        ok = False
    bad = bytearray(got[0])
    bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
    try:
        cs[0].verify(bytes(bad))
        ok = False
    except vx.VxError:
        pass
    key = f"flags{flags}" + ("mix" if mix is not None else "") + f"/world{world}" + (f"/qdf{qdf}" if qdf != 8 else "") + (f"/{overrides}" if overrides else "")
    by_kind[key] = by_kind.get(key, 0) + 1
    if ok:
        n_ok += 1
    else:
        n_bad += 1
        print(json.dumps({"FAIL": {"degree_bits": db, "flags": flags, "pct": pct, "world": world, "pow_bits": sc.desc.pow_bits,
                                   "nch": sc.desc.num_challenges, "cap_height": sc.desc.cap_height, "queries": sc.desc.num_query_rounds, "overrides": overrides, "qdf": qdf}}), flush=True)
    faulthandler.cancel_dump_traceback_later()
    for c in cs:
        c.free()
    oc.free()
    sc.free()
print(json.dumps({"degree_bits": [db_min, db_max], "cases": n_ok + n_bad, "identical_and_verified": n_ok, "failures": n_bad, "seconds": budget, "by_kind": by_kind}), flush=True)
