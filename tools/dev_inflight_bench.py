#!/usr/bin/env python3
"""Experiment: does keeping 2 proofs in flight on one GPU (2 contexts / 2 host threads) raise throughput?"""
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import vectorx_amd as vx  # noqa: E402
from vectorx_amd.synth import SynthCircuit  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 21
K = 6


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


a = setup(1)
t0 = time.perf_counter()
for _ in range(K):
    a[1].prove(dev_ptr=a[2])
t1 = time.perf_counter() - t0
print(f"1 in flight: {K / t1:.3f} proofs/s ({t1 / K * 1e3:.1f} ms)")
b = setup(2)


def run(x, k):
    for _ in range(k):
        x[1].prove(dev_ptr=x[2])


t0 = time.perf_counter()
ths = [threading.Thread(target=run, args=(x, K // 2)) for x in (a, b)]
[t.start() for t in ths]
[t.join() for t in ths]
t2 = time.perf_counter() - t0
print(f"2 in flight: {K / t2:.3f} proofs/s ({t2 / K * 1e3:.1f} ms per proof)")
