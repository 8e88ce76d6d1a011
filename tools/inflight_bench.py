#!/usr/bin/env python3
"""Experiment: proofs in flight on one GPU (one context + host thread each) vs throughput, by proof size.
usage: inflight_bench.py [log_n] [max_in_flight]"""
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import vectorx_amd as vx  # noqa: E402
from vectorx_amd.synth import SynthCircuit  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 21
max_f = int(sys.argv[2]) if len(sys.argv) > 2 else 2
K = 24 if log_n <= 18 else 6


def setup(seed):
    ctx = vx.Context(0)
    sc = SynthCircuit(log_n, seed=seed, poseidon_percent=50)
    c = vx.Circuit(ctx, sc.desc_ptr)
    w = sc.witness()
    d = ctx.alloc(w.nbytes)
    ctx.upload(d, w)
    sc.release_host_buffers(True, True)
    c.prove(dev_ptr=d)  # warm
    return ctx, c, d


def run(x, k):
    for _ in range(k):
        x[1].prove(dev_ptr=x[2])


slots = []
f = 1
while f <= max_f:
    while len(slots) < f:
        slots.append(setup(len(slots) + 1))
    t0 = time.perf_counter()
    ths = [threading.Thread(target=run, args=(x, K // f)) for x in slots[:f]]
    [t.start() for t in ths]
    [t.join() for t in ths]
    dt = time.perf_counter() - t0
    n = (K // f) * f
    print(f"2^{log_n}: {f} in flight: {n / dt:.2f} proofs/s ({dt / n * 1e3:.2f} ms per proof)", flush=True)
    f *= 2
