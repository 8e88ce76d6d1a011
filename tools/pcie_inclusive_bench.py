#!/usr/bin/env python3
"""PCIe-inclusive proving rate: vx_prove handed a HOST witness (pageable numpy memory, and
page-locked memory from vx_host_alloc) versus the HBM-resident witness bench.py times.  Never bench.py's `value`."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 21
    import vectorx_amd as vx
    from vectorx_amd.synth import SynthCircuit
    ctx = vx.Context(0)
    sc = SynthCircuit(log_n, seed=0x5EED0000, poseidon_percent=50)
    circuit = vx.Circuit(ctx, sc.desc_ptr)
    w = sc.witness().copy()                 # pageable
    pinned = ctx.host_alloc(w.shape)
    pinned[:] = w
    d = ctx.alloc(w.nbytes)
    ctx.upload(d, w)
    sc.release_host_buffers(True, True)

    def timeit(fn, reps=3):
        fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.sync()
        return (time.perf_counter() - t0) / reps

    t_dev = timeit(lambda: circuit.prove(dev_ptr=d))
    t_pin = timeit(lambda: circuit.prove(pinned))
    t_page = timeit(lambda: circuit.prove(w))
    print(json.dumps({"log_n": log_n, "witness_GB": w.nbytes / 1e9,
                      "ms_hbm_resident": t_dev * 1e3, "ms_pinned_host": t_pin * 1e3, "ms_pageable_host": t_page * 1e3,
                      "proofs_per_s": {"hbm_resident": 1 / t_dev, "pinned_host": 1 / t_pin, "pageable_host": 1 / t_page},
                      "h2d_GBps_pinned": w.nbytes / 1e9 / max(1e-9, t_pin - t_dev),
                      "h2d_GBps_pageable": w.nbytes / 1e9 / max(1e-9, t_page - t_dev)}))
    ctx.host_free(pinned)
    circuit.free()
    ctx.free(d)
    ctx.close()


if __name__ == "__main__":
    main()
