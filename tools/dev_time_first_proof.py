import sys, time
sys.path.insert(0,'/root/repo')
import vectorx_amd as vx
from vectorx_amd.synth import SynthCircuit
ctx=vx.Context(0)
for db in (19,18,16):
    sc=SynthCircuit(db, seed=303, poseidon_percent=50)
    c=vx.Circuit(ctx, sc.desc_ptr)
    w=sc.witness(); d=ctx.alloc(w.nbytes); ctx.upload(d,w)
    ts=[]
    for i in range(4):
        ctx.sync(); t=time.perf_counter(); p=c.prove(dev_ptr=d); ctx.sync(); ts.append((time.perf_counter()-t)*1e3)
    print(db, ['%.1f'%x for x in ts], flush=True)
    c.free(); ctx.free(d); sc.free()
