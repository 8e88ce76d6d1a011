"""Dev: where does the time of a SMALL proof go (launch/sync overhead vs kernels)?"""
import sys, time
sys.path.insert(0, '/root/repo')
import vectorx_amd as vx
from vectorx_amd.synth import SynthCircuit
for log_n in (14, 16, 18):
    sc = SynthCircuit(log_n, seed=1, poseidon_percent=50)
    ctx = vx.Context(0)
    c = vx.Circuit(ctx, sc.desc_ptr)
    w = sc.witness()
    d = ctx.alloc(w.nbytes); ctx.upload(d, w)
    for _ in range(3): c.prove(dev_ptr=d)
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(10): c.prove(dev_ptr=d)
    ctx.sync(); wall = (time.perf_counter() - t0) / 10
    ctx.prof_enable(True); ctx.prof_reset()
    for _ in range(10): c.prove(dev_ptr=d)
    pr = ctx.prof(); ctx.prof_enable(False)
    tot = sum(v['ms'] for v in pr.values()) / 10
    print(f"2^{log_n}: wall {wall*1e3:.2f} ms (no profiler), sum of stage GPU times {tot:.2f} ms")
    print("   ", {k: round(v['ms'] / 10, 3) for k, v in sorted(pr.items(), key=lambda kv: -kv[1]['ms'])})
    ctx.free(d); c.free(); ctx.close()
