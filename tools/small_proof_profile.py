#!/usr/bin/env python3
"""Latency of LONE small proofs (the reduce / map / outer sizes of the header_range DAG): wall ms per proof and the HIP-event stage
times, one JSON line per size.  `python tools/small_proof_profile.py 14 16 18 19`"""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import vectorx_amd as vx
    from vectorx_amd.synth import SynthCircuit
    sizes = [int(a) for a in sys.argv[1:]] or [14, 16, 18, 19]
    ctx = vx.Context(0)
    for lg in sizes:
        sc = SynthCircuit(lg, seed=202, poseidon_percent=50, witness_seed=1)
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
        print(json.dumps({"log_n": lg, "ms_per_proof": round(ms, 3), "kernel_ms": round(sum(stages.values()), 3), "stage_ms": stages}), flush=True)
        circuit.free()
        ctx.free(d_w)
        sc.free()
    ctx.close()


if __name__ == "__main__":
    main()
