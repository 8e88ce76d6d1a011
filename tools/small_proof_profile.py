#!/usr/bin/env python3
"""Latency of LONE small proofs (the reduce / map / outer sizes of the header_range DAG): wall ms per proof and the HIP-event stage
times, one JSON line per size.  `python tools/small_proof_profile.py 14 16 18 19`
`--recursion`: the circuits carry the recursive verifier's gate set in its declared mix (vectorx_amd/synth.py RECURSIVE_VERIFIER_MIX), and
the line gains `quotient_by_gate_ms`: the quotient's per-gate kernels by gate name (the library numbers the stages `qgate_<gate index>`)."""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import vectorx_amd as vx
    from vectorx_amd.synth import SynthCircuit
    from vectorx_amd.mapreduce import circuit_shape
    recursion = "--recursion" in sys.argv[1:]
    sizes = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [14, 16, 18, 19]
    ctx = vx.Context(0)
    for lg in sizes:
        sc = SynthCircuit(lg, seed=202, poseidon_percent=50, witness_seed=1, **circuit_shape(recursion))
        circuit = vx.Circuit(ctx, sc.desc_ptr)
        w = sc.witness()
        d_w = ctx.alloc(w.nbytes)
        ctx.upload(d_w, w)
        for _ in range(3):
            circuit.prove(dev_ptr=d_w)
        steps = 20
        ctx.prof_enable(True)
        ctx.prof_reset()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            circuit.prove(dev_ptr=d_w)
        ctx.sync()
        ms = (time.perf_counter() - t0) / steps * 1e3
        prof = ctx.prof()
        ctx.prof_enable(False)
        stages = {k: round(v["ms"] / steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
        rec = {"log_n": lg, "recursion_mix": recursion, "ms_per_proof": round(ms, 3)}
        names = sc.gate_names()
        by_gate = {"+".join(names[int(g)] for g in k[6:].split("+")): v for k, v in stages.items() if k.startswith("qgate_")}
        stages = {k: v for k, v in stages.items() if not k.startswith("qgate_")}
        # nested brackets (quotient_lookup_terms and the per-gate launches sit inside quotient_eval / quotient_program_gates_jit) are not summed twice
        nested = ("quotient_lookup_terms", "quotient_program_gates_jit", "quotient_program_gates")
        rec["kernel_ms"] = round(sum(v for k, v in stages.items() if k not in nested), 3)
        rec["stage_ms"] = stages
        if by_gate:
            rec["quotient_by_gate_ms"] = by_gate
            rec["gate_rows"] = sc.gate_rows()
        print(json.dumps(rec), flush=True)
        circuit.free()
        ctx.free(d_w)
        sc.free()
    ctx.close()


if __name__ == "__main__":
    main()
