import sys, faulthandler, json
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import vectorx_amd as vx
from vectorx_amd import sharded
from vectorx_amd.synth import SynthCircuit
db, flags, qdf, world, nq, nch = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
ar = [int(x) for x in sys.argv[7].split(",")] if len(sys.argv) > 7 and sys.argv[7] != "-" else None
sc = SynthCircuit(db, seed=1234, poseidon_percent=35, witness_seed=99, flags=flags, quotient_degree_factor=qdf)
sc.desc.pow_bits = 0
sc.desc.num_challenges = nch
sc.desc.num_query_rounds = nq
if ar is not None:
    sc.set_fri_reduction_arity_bits(ar)
faulthandler.dump_traceback_later(90, exit=True)
ctxs = [vx.Context(0) for _ in range(world)]
cs = [vx.Circuit(c, sc.desc_ptr) for c in ctxs]
w = sc.witness()
got = sharded.prove_sharded_threads(cs, w) if world > 1 else [cs[0].prove(w)]
one = vx.Circuit(ctxs[0], sc.desc_ptr).prove(w) if world > 1 else got[0]
print("OK", len(got[0]), all(p == one for p in got), flush=True)
