#!/usr/bin/env python3
"""STARK path measurement: one `vx_stark_prove` of a wide degree-3 AIR on one MI355X.

Workload: tests/stark_airs.py::mulchain — `groups` blocks of four columns, three transition constraints and one degree-3
all-rows constraint per block (Curta's chips have constraint degree 3; their real AIRs need the starkyx sources), starky's
standard_fast_config (rate_bits 1, 84 queries, 16 PoW bits).  The trace is uploaded once; the timed region is
vx_stark_prove from a DEVICE-resident trace (trace commitment, AIR quotient on 2 cosets, quotient commitment, openings,
FRI), HIP events on the library's stream per stage.  Prints one JSON line.

    python tools/stark_bench.py --log-n 18 --groups 16 --steps 5 --warmup 2
"""
import argparse
import ctypes
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=18)
    ap.add_argument("--groups", type=int, default=16, help="blocks of 4 trace columns")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--air", default="mulchain", choices=["mulchain", "sha256", "blake2b", "ed25519", "eddsa", "blake2b_bytes"],
                    help="sha256: vectorx_amd/sha256_air.py — 1024 + 3 columns, 2072 constraints, two commitment rounds (own AIR, not Curta's)")
    ap.add_argument("--check", action="store_true", help="verify the last proof with vx_stark_verify")
    ap.add_argument("--host-trace", action="store_true", help="chip AIRs: the trace starts in page-locked HOST memory on every proof (PCIe-inclusive rate; "
                    "the upload hides behind the transforms and the carried-state leaf hashing)")
    args = ap.parse_args()
    import vectorx_amd as vx
    if args.air == "sha256":
        return sha256_bench(args, vx)
    if args.air == "blake2b":
        return sha256_bench(args, vx, "blake2b")
    if args.air == "ed25519":
        return sha256_bench(args, vx, "ed25519")
    if args.air == "blake2b_bytes":   # bytes + XOR lookup, four G per row (vectorx_amd/blake2b_bytes_air.py): --log-n 16 = one map job's 2240 compressions
        from vectorx_amd import stark_chips
        ctx = vx.Context(0)
        print(json.dumps(stark_chips.bench_blake2b_bytes(ctx, max(16, args.log_n) if args.log_n != 18 else 16, args.steps, args.warmup)), flush=True)
        ctx.close()
        return
    if args.air == "eddsa":          # the batched signature table (vectorx_amd/eddsa_air.py): --log-n 20 = 97 signatures per proof
        from vectorx_amd import stark_chips
        ctx = vx.Context(0)
        print(json.dumps(stark_chips.bench_eddsa(ctx, args.log_n, args.steps, args.warmup, check=args.check)), flush=True)
        ctx.close()
        return
    from stark_airs import mulchain
    stark, trace, pis = mulchain(args.log_n, groups=args.groups)
    ctx = vx.Context(0)
    L = vx.lib()
    nbytes = trace.nbytes
    dptr = ctx.alloc(nbytes)
    ctx.upload(dptr, trace)
    cap = 1 << 24
    out = np.empty(cap, dtype=np.uint8)
    vp = ctypes.c_void_p

    def prove():
        n = ctypes.c_size_t(cap)
        rc = L.vx_stark_prove(ctx._h, ctypes.cast(stark.desc_ptr, vp), vp(dptr), 1, pis.ctypes.data, None, out.ctypes.data, ctypes.byref(n))
        if rc != 0:
            raise RuntimeError(L.vx_last_error().decode())
        return n.value

    for _ in range(args.warmup):
        prove()
    ctx.prof_enable(True)
    ctx.prof_reset()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        nb = prove()
    ctx.sync()
    dt = (time.perf_counter() - t0) / args.steps
    stages = {k: round(v["ms"] / args.steps, 3) for k, v in ctx.prof().items()}
    if args.check:
        stark.verify(pis, out[:nb].tobytes())
    n = 1 << args.log_n
    print(json.dumps({
        "metric": "vx_stark_prove proofs/sec (scoping spike, synthetic AIR)", "value": 1.0 / dt, "unit": "proofs/sec", "ms_per_proof": dt * 1e3,
        "config": {"workload": f"mulchain AIR: {4 * args.groups} columns x 2^{args.log_n} rows, {4 * args.groups + 2} constraints of degree <= 3, "
                               "rate_bits 1, cap_height 4, 84 queries, 16 PoW bits, trace resident in HBM",
                   "trace_bytes": int(nbytes), "proof_bytes": int(nb)},
        "stage_ms_per_proof": stages, "steps": args.steps, "warmup": args.warmup, "n_gpus": 1, "data": "synthetic",
        "dtype": "u64 (Goldilocks field, integer modular arithmetic)",
        "trace_cells_per_s": 4 * args.groups * n / dt}))
    ctx.free(dptr)
    ctx.close()


def sha256_bench(args, vx, which="sha256"):
    """the chip-sized AIRs (vectorx_amd/stark_chips.py): trace AND second-round columns resident in HBM unless --host-trace"""
    from vectorx_amd import stark_chips
    ctx = vx.Context(0)
    try:
        print(json.dumps(stark_chips.bench_chip(ctx, which, args.log_n, args.steps, args.warmup, host_trace=args.host_trace, check=args.check)))
    finally:
        ctx.close()


if __name__ == "__main__":
    main()
