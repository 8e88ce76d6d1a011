#!/usr/bin/env python3
"""Native trace generation (vx_trace_*) alone: wall ms and HIP-event ms per table at the shapes the header_range jobs use, with the
achieved write bandwidth (8 B per cell, every cell written once).  `python tools/tracegen_bench.py`; under rocprofv3 --kernel-trace
--stats the three kernels of each generator show up separately."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import vectorx_amd as vx
    ctx = vx.Context(0)
    rng = np.random.default_rng(1)
    cases = [("blake2b", 16, 8, 128 * 280, "map job: 8 headers x 280 blocks"), ("sha256", 11, 14, 64, "map job: 14 tree nodes"),
             ("sha256", 9, 2, 64, "reduce job: 2 merge nodes"), ("sha256", 16, 300, 64, "outer job: 300 keys"),
             ("sha512", 16, 300, 117, "outer job: 300 signed messages")]
    for which, log_n, nmsg, mlen, what in cases:
        msgs = [rng.integers(0, 256, size=mlen, dtype=np.uint8).tobytes() for _ in range(nmsg)]
        ncols = ctx.TRACE_TABLES[which][0]
        nbytes = ncols * (1 << log_n) * 8
        d = ctx.alloc(nbytes)
        for _ in range(2):
            ctx.trace_hash_table(which, log_n, msgs, d)
        steps = 10
        ctx.prof_enable(True)
        ctx.prof_reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.trace_hash_table(which, log_n, msgs, d)
        ctx.sync()
        wall = (time.perf_counter() - t0) / steps * 1e3
        ev = ctx.prof()["trace_generation"]["ms"] / steps
        ctx.prof_enable(False)
        ctx.free(d)
        print(json.dumps({"table": which, "rows_log2": log_n, "what": what, "trace_MB": round(nbytes / 1e6, 1), "wall_ms": round(wall, 3),
                          "device_ms": round(ev, 3), "write_GBps_device": round(nbytes / (ev * 1e-3) / 1e9, 1)}), flush=True)
    # the batched EdDSA table of the outer job: 97 signature equations per 2^20 rows
    from vectorx_amd import eddsa_air as ea
    from vectorx_amd import stark_chips
    for full, log_n in ((False, 17), (False, 20), (True, 20)):
        lay = ea.Layout(full=full)
        cap = ea.capacity(lay, log_n)
        sigs, rs = (stark_chips.eddsa_signatures_full if full else stark_chips.eddsa_signatures)(cap, 8)
        nbytes = lay.N * (1 << log_n) * 8
        d = ctx.alloc(nbytes)
        assert ctx.trace_eddsa_table(log_n, lay.NB, sigs, d, full=full) == rs
        steps = 5
        ctx.prof_enable(True)
        ctx.prof_reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.trace_eddsa_table(log_n, lay.NB, sigs, d, full=full)
        ctx.sync()
        wall = (time.perf_counter() - t0) / steps * 1e3
        ev = ctx.prof()["trace_generation"]["ms"] / steps
        ctx.prof_enable(False)
        ctx.free(d)
        print(json.dumps({"table": "eddsa full" if full else "eddsa", "rows_log2": log_n, "what": f"{cap} signature equations", "trace_MB": round(nbytes / 1e6, 1),
                          "wall_ms": round(wall, 3), "device_ms": round(ev, 3), "write_GBps_device": round(nbytes / (ev * 1e-3) / 1e9, 1)}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
