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
    ap.add_argument("--air", default="mulchain", choices=["mulchain", "sha256", "blake2b", "ed25519"],
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
    """the SHA-256 AIR (or, which="blake2b", the BLAKE2b AIR of vectorx_amd/blake2b_air.py): trace AND second-round columns resident in HBM (vx_stark_begin / vx_stark_finish with device pointers);
    the caller's aux-column computation (host arithmetic) happens once, outside the timed loop — witness generation stays on
    the caller's side of the boundary"""
    if which == "ed25519":
        from vectorx_amd import ed25519_air as sha
        name, block_bytes, naux = "Ed25519 scalar-multiplication", 0, sha.Cols.NAUX
        what = "188 bytes of result, quotient and carries per row looked up in a 256-entry table (log-derivative)"
    elif which == "blake2b":
        from vectorx_amd import blake2b_air as sha
        name, block_bytes, naux = "BLAKE2b-256", 128, 6
        what = "message bytes range-checked by a log-derivative lookup into a 256-entry table"
    else:
        from vectorx_amd import sha256_air as sha
        name, block_bytes, naux = "SHA-256", 64, 3
        what = "log-derivative range check"
    t_gen = time.perf_counter()
    n = 1 << args.log_n
    nblocks = n // sha.PERIOD
    msgs = [bytes([i & 255]) * (block_bytes * 7 + 20) for i in range(max(1, nblocks // 8))]     # 8-block messages
    stark = sha.make_stark(args.log_n)
    if which == "ed25519":
        trace, pis, _ = sha.generate_trace(args.log_n, int.from_bytes(bytes(range(7, 7 + n // 256)), "little"))
    else:
        trace, pis, digests = sha.generate_trace(args.log_n, msgs)
    t_gen = time.perf_counter() - t_gen
    ctx = vx.Context(0)
    L = vx.lib()
    vp = ctypes.c_void_p
    d_trace = ctx.alloc(trace.nbytes)
    ctx.upload(d_trace, trace)
    h_trace = None
    if args.host_trace:
        h_trace = ctx.host_alloc(trace.shape)
        h_trace[:] = trace
    cap = 1 << 25
    out = np.empty(cap, dtype=np.uint8)
    chal = np.zeros(1, dtype=np.uint64)
    d_aux = ctx.alloc(naux * n * 8)
    state = {"chal": None}

    def prove():
        sess = vp()
        if h_trace is not None:
            rc = L.vx_stark_begin(ctx._h, ctypes.cast(stark.desc_ptr, vp), vp(h_trace.ctypes.data), 0, pis.ctypes.data, chal.ctypes.data, ctypes.byref(sess))
        else:
            rc = L.vx_stark_begin(ctx._h, ctypes.cast(stark.desc_ptr, vp), vp(d_trace), 1, pis.ctypes.data, chal.ctypes.data, ctypes.byref(sess))
        if rc != 0:
            raise RuntimeError(L.vx_last_error().decode())
        try:
            if state["chal"] != int(chal[0]):          # same trace => same challenge: the aux columns are computed once
                aux = np.ascontiguousarray(sha.aux_columns(trace, chal), dtype=np.uint64)
                ctx.upload(d_aux, aux)
                state["chal"] = int(chal[0])
            nb = ctypes.c_size_t(cap)
            rc = L.vx_stark_finish(sess, vp(d_aux), 1, None, out.ctypes.data, ctypes.byref(nb))
            if rc != 0:
                raise RuntimeError(L.vx_last_error().decode())
            return nb.value
        finally:
            L.vx_stark_session_free(sess)

    t_first = time.perf_counter()
    prove()                                              # includes the hiprtc compilation of the 16 k-word program
    t_first = time.perf_counter() - t_first
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
    ev = stages.get("air_quotient_eval_jit", stages.get("air_quotient_eval", 0.0))
    hashing = stages.get("hash_leaves", 0.0) + stages.get("merkle_levels", 0.0)
    prog, npush = sha.build_program()
    print(json.dumps({
        "metric": f"vx_stark_begin + vx_stark_finish proofs/sec ({name} AIR at chip density; own AIR, not Curta's)", "value": 1.0 / dt, "unit": "proofs/sec",
        "ms_per_proof": dt * 1e3, f"{which}_blocks_per_s": nblocks / dt,
        "config": {"workload": f"{name} AIR: {sha.Cols.N} + {naux} columns x 2^{args.log_n} rows ({nblocks} {'double-and-add steps' if which == 'ed25519' else 'compression blocks'} of {sha.PERIOD} rows), {npush} constraints "
                               f"of degree <= 3, program {len(prog)} words, {what} in a second commitment round, rate_bits 1, "
                               "cap_height 4, 84 queries, 16 PoW bits, " + ("trace in page-locked HOST memory at the start of every proof, aux columns resident in HBM"
                                                                         if args.host_trace else "trace + aux columns resident in HBM"),
                   "trace_bytes": int(trace.nbytes), "proof_bytes": int(nb), "evaluator": "compiled" if "air_quotient_eval_jit" in stages else "interpreted"},
        "stage_ms_per_proof": stages, "evaluator_ms": ev, "hashing_ms": hashing, "evaluator_share": round(ev / (dt * 1e3), 4),
        "hashing_share": round(hashing / (dt * 1e3), 4), "first_proof_seconds_incl_jit": round(t_first, 2), "trace_generation_seconds_host": round(t_gen, 2),
        "steps": args.steps, "warmup": args.warmup, "n_gpus": 1, "data": "synthetic", "dtype": "u64 (Goldilocks field, integer modular arithmetic)",
        "trace_cells_per_s": sha.Cols.N * n / dt}))
    ctx.free(d_trace)
    ctx.free(d_aux)
    ctx.close()


if __name__ == "__main__":
    main()
