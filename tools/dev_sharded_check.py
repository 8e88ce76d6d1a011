"""Dev check: one proof split across G ranks (threads + vx_group on ONE GPU) must equal the single-GPU proof."""
import sys, time
sys.path.insert(0, '/root/repo')
import vectorx_amd as vx
from vectorx_amd.synth import SynthCircuit
from vectorx_amd.sharded import prove_sharded_threads

for db, flags in ((6, 0), (4, 0), (5, 0), (9, 7), (12, 0), (14, 0)):
    sc = SynthCircuit(db, seed=77 + db, poseidon_percent=40, flags=flags if db >= 5 else 0)
    sc.desc.pow_bits = 8
    w = sc.witness()
    ctx0 = vx.Context(0)
    c0 = vx.Circuit(ctx0, sc.desc_ptr)
    ref = c0.prove(w)
    for world in (1, 2, 4, 8):
        ctxs = [vx.Context(0) for _ in range(world)]
        circs = [vx.Circuit(c, sc.desc_ptr) for c in ctxs]
        t = time.time()
        proofs = prove_sharded_threads(circs, w)
        dt = time.time() - t
        ok = all(p == ref for p in proofs)
        print(f"degree_bits={db} flags={flags} world={world}: identical={ok} ({dt*1e3:.0f} ms)", flush=True)
        if not ok:
            p = proofs[0]
            first = next((i for i in range(min(len(p), len(ref))) if p[i] != ref[i]), None)
            print("   len", len(p), len(ref), "first diff at byte", first)
        for c in circs: c.free()
        for c in ctxs: c.close()
    c0.free(); ctx0.close()
