import sys, faulthandler; faulthandler.enable()
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import vectorx_amd as vx
from vectorx_amd.synth import SynthCircuit
qdf, flags, lg = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = vx.Context(0)
sc = SynthCircuit(lg, seed=8000+lg, poseidon_percent=50, flags=flags, quotient_degree_factor=qdf)
sc.desc.pow_bits = 6
gc = vx.Circuit(ctx, sc.desc_ptr)
print("circuit ok", flush=True)
p = gc.prove(sc.witness())
print("proved", len(p), flush=True)
gc.verify(p)
print("verified", flush=True)
