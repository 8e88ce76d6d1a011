import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, oracle_lib
import vectorx_amd as vx
from vectorx_amd.synth import SynthCircuit
o = oracle_lib.load()
ctx = vx.Context(0)
for db in (3, 5, 6, 8, 10, 12):
    sc = SynthCircuit(db, seed=db, poseidon_percent=50)
    oc = oracle_lib.OracleCircuit(o, sc.desc_ptr)
    gc = vx.Circuit(ctx, sc.desc_ptr)
    print(db, 'digest equal:', (gc.digest()==oc.digest()).all(), 'cap equal:', (gc.constants_sigmas_cap()==oc.cap()).all())
    t=time.time(); gp = gc.prove(sc.witness()); tg=time.time()-t
    t=time.time(); op = oc.prove(sc.witness()); to=time.time()-t
    print('  gpu %.3fs oracle %.3fs  len %d %d  equal: %s' % (tg, to, len(gp), len(op), gp==op))
    if gp != op:
        n = min(len(gp), len(op))
        diff = [i for i in range(n) if gp[i]!=op[i]]
        print('  first diff byte', diff[0] if diff else None, 'ndiff', len(diff))
    print('  oracle verifies gpu proof:', repr(oc.verify(gp)))
