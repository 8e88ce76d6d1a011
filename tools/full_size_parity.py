#!/usr/bin/env python3
"""One-off parity check at the benchmark sizes: the GPU proof of the header_range_256 / header_range_512 stand-ins
(n = 2^20, 2^21) against the ORACLE's proof of the same circuit and witness, byte for byte.  The oracle needs minutes
per proof on the box's 16 cores, so this is not part of the test suite (which checks byte identity up to 2^17 and the
verifiers at 2^20 / 2^21); the outcome is recorded in profiles/.

    python tools/full_size_parity.py 20 21 > gpurun_out/full_size_parity.jsonl
"""
import hashlib
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib  # noqa: E402  (checker only)
import bench_prove  # noqa: E402
import vectorx_amd as vx  # noqa: E402
from vectorx_amd.synth import SynthCircuit  # noqa: E402

oracle = oracle_lib.load()
oracle.L.vxo_set_num_threads(bench_prove.usable_cores())
ctx = vx.Context(0)
for log_n in [int(a) for a in sys.argv[1:]] or [20]:
    sc = SynthCircuit(log_n, seed=0x5EED0000, poseidon_percent=50)
    w = sc.witness()
    c = vx.Circuit(ctx, sc.desc_ptr)
    t0 = time.time()
    gp = c.prove(w)
    t_gpu = time.time() - t0
    c.verify(gp)
    t0 = time.time()
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    t_build = time.time() - t0
    same_digest = bool((oc.digest() == c.digest()).all())
    t0 = time.time()
    op = oc.prove(w)
    t_cpu = time.time() - t0
    print(json.dumps({"log_n": log_n, "proof_bytes": len(gp), "gpu_sha256": hashlib.sha256(gp).hexdigest(),
                      "oracle_sha256": hashlib.sha256(op).hexdigest(), "byte_identical": gp == op, "circuit_digest_identical": same_digest,
                      "oracle_verifies_gpu_proof": oc.verify(gp) == "", "gpu_prove_s_incl_h2d": round(t_gpu, 3),
                      "oracle_circuit_build_s": round(t_build, 1), "oracle_prove_s": round(t_cpu, 1),
                      "oracle_threads": bench_prove.usable_cores()}), flush=True)
    c.free()
    oc.free()
    sc.free()
ctx.close()
