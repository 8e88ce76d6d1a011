#!/usr/bin/env python3
"""Latency of LONE small proofs (the reduce / map / outer sizes of the header_range DAG): wall ms per proof and the HIP-event stage
times, one JSON line per size.  `python tools/small_proof_profile.py 14 16 18 19`
`--recursion`: the circuits carry the recursive verifier's gate set in its declared mix (vectorx_amd/synth.py RECURSIVE_VERIFIER_MIX), and
the line gains `gate_rows`; `--per-gate`: the program gates as one kernel each instead of the fused kernel, `quotient_by_gate_ms` names each
(the library numbers those stages `qgate_<gate index>`).  The quotient's kernels are always listed (`quotient_by_kernel_ms`)."""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import bench_prove
    import vectorx_amd as vx
    recursion = "--recursion" in sys.argv[1:]
    sizes = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [14, 16, 18, 19]
    ctx = vx.Context(0)
    for lg in sizes:
        print(json.dumps(bench_prove.lone_proof_profile(ctx, lg, recursion, steps=20, per_gate="--per-gate" in sys.argv[1:])), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
