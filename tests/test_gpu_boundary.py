"""GPU tests of the boundary itself: a plain-C program driving libvxprover.so through include/vxprover.h (no
Python in the loop), and two host threads driving two contexts on the same GPU concurrently (the header promises
contexts are independent — plonky2 calls the prover from rayon/tokio worker threads)."""
import subprocess
import threading
from pathlib import Path

import numpy as np
import pytest

import vectorx_amd as vx
from vectorx_amd.synth import SynthCircuit

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _fnv1a(b: bytes) -> int:
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_plain_c_consumer_of_the_abi(ctx, tmp_path):
    exe = tmp_path / "c_abi_smoke"
    libdir = ROOT / "vectorx_amd"
    r = subprocess.run(["gcc", "-O2", "-I", str(ROOT / "include"), str(ROOT / "tests" / "c_abi_smoke.c"), "-o", str(exe),
                        "-L", str(libdir), "-lvxprover", "-lvxsynth", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe), "10", "7"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    fields = dict(kv.split("=") for kv in r.stdout.split()[1:])
    # the same circuit + witness through the ctypes mirror gives the same proof bytes
    sc = SynthCircuit(10, seed=7, poseidon_percent=50)
    gc = vx.Circuit(ctx, sc.desc_ptr)
    proof = gc.prove(sc.witness())
    assert int(fields["len"]) == len(proof)
    assert int(fields["fnv"], 16) == _fnv1a(proof)
    assert int(fields["digest0"], 16) == int(gc.digest()[0])
    gc.free()


def test_two_contexts_from_two_threads(ctx):
    """Two host threads, two vx_ctx on the same device, proving different circuits at the same time: each result
    must equal the proof obtained sequentially."""
    specs = [(11, 21), (12, 22)]
    ref = {}
    for db, seed in specs:
        sc = SynthCircuit(db, seed=seed, poseidon_percent=50)
        gc = vx.Circuit(ctx, sc.desc_ptr)
        ref[(db, seed)] = gc.prove(sc.witness())
        gc.free()
    out, errs = {}, []

    def worker(db, seed):
        try:
            c2 = vx.Context(0)
            sc = SynthCircuit(db, seed=seed, poseidon_percent=50)
            gc = vx.Circuit(c2, sc.desc_ptr)
            for _ in range(3):
                out[(db, seed)] = gc.prove(sc.witness())
                assert out[(db, seed)] == ref[(db, seed)]
            gc.free()
            c2.close()
        except Exception as e:  # pragma: no cover
            errs.append(repr(e))

    ths = [threading.Thread(target=worker, args=s) for s in specs]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    assert out == ref


def test_mapreduce_dag_on_one_gpu_matches_the_oracle_dag(ctx, oracle):
    """The proof DAG (4 map + 3 reduce + 1 outer, tiny circuits — since round 6 with the recursive verifier's gate set, GpuProver's
    default) proven on the GPU gives the same root digest as the same DAG proven by the oracle: every one of the 8 proofs is
    byte-identical, dependencies included."""
    import hashlib
    import oracle_lib
    from vectorx_amd import mapreduce as mr
    from _mp_dag_worker import OracleProver
    spec = mr.DagSpec(num_map=4, map_log_n=6, reduce_log_n=5, outer_log_n=6)
    gp = []

    def make_gpu(kind, log_n, jobs):
        p = mr.GpuProver(ctx, kind, log_n, jobs)
        gp.append(p)
        return p
    g = mr.run_dag(spec, make_gpu, None, ctx.sync)
    assert spec.recursion
    o = mr.run_dag(spec, lambda kind, log_n, jobs: OracleProver(oracle, kind, log_n, jobs, recursion=True), None)
    assert g["proofs"] == o["proofs"] == 8
    assert g["root"] == o["root"]
    assert g["my_proofs"].keys() == o["my_proofs"].keys()
    for k in g["my_proofs"]:
        assert g["my_proofs"][k] == o["my_proofs"][k], k
    for p in gp:
        p.free()
    # the same DAG with 3 jobs of a layer in flight on the GPU (3 contexts / host threads): same proofs
    lanes = [vx.Context(0), vx.Context(0)]
    gp2 = []

    def make_lanes(kind, log_n, jobs):
        p = mr.GpuProver(ctx, kind, log_n, jobs, extra_lanes=lanes)
        gp2.append(p)
        return p
    g2 = mr.run_dag(spec, make_lanes, None, ctx.sync, in_flight=3)
    assert g2["root"] == o["root"] and g2["my_proofs"] == o["my_proofs"]
    for p in gp2:
        p.free()
    # the bench's DAG leg: a few base witnesses per circuit kind, each job's public inputs patched into the lane's own copy:
    # every proof verifies, the root does not depend on how many jobs are in flight
    roots = []
    for in_flight, use_lanes in ((3, lanes), (1, [])):
        gp3 = []

        def make_few(kind, log_n, jobs, use_lanes=use_lanes):
            p = mr.GpuProver(ctx, kind, log_n, jobs, extra_lanes=use_lanes, distinct_witnesses=2)
            gp3.append(p)
            return p
        g3 = mr.run_dag(spec, make_few, None, ctx.sync, in_flight=in_flight)
        roots.append(g3["root"])
        for (li, j), proof in g3["my_proofs"].items():
            kind = spec.layers()[li][0]
            next(p for p in gp3 if p.kind == kind).circuit.verify(proof)
        for p in gp3:
            p.free()
    assert roots[0] == roots[1] and roots[0] != o["root"]
    for l in lanes:
        l.close()


@pytest.mark.parametrize("degree_bits,flags", [(6, 0), (9, 15), (12, 1)])
def test_vxcircuit_save_load_gives_identical_proofs(ctx, oracle, degree_bits, flags):
    """`.vxcircuit` round trip on the device (vx_circuit_serialize -> vx_circuit_load): same digest, same cap, byte-identical
    proofs — the analogue of `circuit.test_serializers` (/root/reference/circuits/header_range.rs:117-126) followed by the
    prove/verify pair of :167-170."""
    import vectorx_amd as vx
    from vectorx_amd.synth import SynthCircuit
    import oracle_lib
    sc = SynthCircuit(degree_bits, seed=4400 + degree_bits, poseidon_percent=50, flags=flags)
    sc.desc.pow_bits = 7
    g0 = vx.Circuit(ctx, sc.desc_ptr)
    cap = g0.constants_sigmas_cap()
    blob = vx.circuit_serialize(sc.desc_ptr, cap, with_preprocessed=True)
    g1 = vx.Circuit.load(ctx, blob)
    assert (g1.digest() == g0.digest()).all() and (g1.constants_sigmas_cap() == cap).all()
    w = sc.witness()
    p0, p1 = g0.prove(w), g1.prove(w)
    assert p0 == p1 == oracle_lib.OracleCircuit(oracle, sc.desc_ptr).prove(w)
    g1.verify(p0)
    assert g1.program_gates()[0] == g0.program_gates()[0]
    # a verifier-only file cannot be loaded as a prover key
    with pytest.raises(vx.VxError):
        vx.Circuit.load(ctx, vx.circuit_serialize(sc.desc_ptr, cap, with_preprocessed=False))
    bad = bytearray(blob)
    bad[len(bad) // 2] ^= 4
    with pytest.raises(vx.VxError):
        vx.Circuit.load(ctx, bytes(bad))
    g0.free()
    g1.free()


def test_function_cli_build_and_prove_on_the_gpu(tmp_path):
    """`build` then `prove input.json` -> output.json with the real GpuBackend; the result equals the one the CPU test
    backend produces from the same request (tests/test_function_cli.py)."""
    import json
    from vectorx_amd import function as fn
    from test_function_cli import OracleBackend, _header_range_input, _request
    for function, extra, raw in [("rotate", ["--rotate-log-n", "8"], b"\x00" * 7 + b"\x2a" + bytes(range(32))),
                                 ("header_range_256", ["--map-log-n", "6", "--reduce-log-n", "5", "--outer-log-n", "7"], _header_range_input())]:
        b = tmp_path / ("build_" + function)
        req, outp = tmp_path / f"{function}.input.json", tmp_path / f"{function}.output.json"
        req.write_text(_request(raw))
        assert fn.main(["build", "--function", function, "--build-dir", str(b), *extra]) == 0
        assert fn.main(["prove", str(req), "--function", function, "--build-dir", str(b), "--output", str(outp)]) == 0
        res = json.loads(outp.read_text())
        proof_cpu, out_cpu, stats = fn.prove(function, raw, b, OracleBackend())    # same build artefacts, CPU checker
        assert res["data"]["proof"] == "0x" + proof_cpu.hex() and res["data"]["output"] == "0x" + out_cpu.hex()
        assert stats["proofs"] == (1 if function == "rotate" else 64)
