"""Dev: cost of the constraint-program interpreter at n = 2^log_n (stage quotient_program_gates) for the synth flags."""
import sys, json
sys.path.insert(0, '/root/repo')
import vectorx_amd as vx
from vectorx_amd.synth import SynthCircuit

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for flags in (0, 2, 7):
    sc = SynthCircuit(log_n, seed=1, poseidon_percent=50, flags=flags)
    ctx = vx.Context(0)
    import time; t0 = time.time()
    c = vx.Circuit(ctx, sc.desc_ptr)
    print('  circuit_create %.2f s' % (time.time() - t0), c.program_gates())
    w = sc.witness()
    d = ctx.alloc(w.nbytes); ctx.upload(d, w)
    c.prove(dev_ptr=d)
    ctx.prof_enable(True); ctx.prof_reset()
    c.prove(dev_ptr=d)
    pr = ctx.prof()
    print(flags, 'program words', sc.desc.programs_len, {k: round(v['ms'], 2) for k, v in pr.items() if k.startswith('quotient')}, flush=True)
    ctx.free(d); c.free(); ctx.close()
