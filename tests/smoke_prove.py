"""One small whole proof on cuda:0 checked against the oracle — called by __graft_entry__.smoke().  Also one proof of a
circuit with constraint-program gates + a lookup table, and two STARKs (Fibonacci AIR; a two-round lookup AIR), each byte-compared
with the oracle."""


def run(ctx, oracle):
    import oracle_lib
    import vectorx_amd as vx
    from vectorx_amd.synth import SynthCircuit

    sc = SynthCircuit(8, seed=2026, poseidon_percent=50)
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    gc = vx.Circuit(ctx, sc.desc_ptr)
    assert (gc.digest() == oc.digest()).all(), "circuit_digest differs from the oracle"
    gp = gc.prove(sc.witness())
    op = oc.prove(sc.witness())
    assert gp == op, "GPU proof bytes differ from the oracle's proof"
    assert oc.verify(gp) == "", "restated verifier rejects the GPU proof"
    gc.verify(gp)   # the product library's own CircuitData::verify
    gc.free()
    print(f"smoke ok: vx_prove (n=2^8, {len(gp)} bytes) is byte-identical to the oracle proof and verifies")

    sc = SynthCircuit(7, seed=2027, poseidon_percent=40, flags=16 | 1)     # lookup table + program gates
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    gc = vx.Circuit(ctx, sc.desc_ptr)
    gp = gc.prove(sc.witness())
    assert gp == oc.prove(sc.witness()), "GPU proof of the lookup circuit differs from the oracle's"
    gc.verify(gp)
    gc.free()
    print(f"smoke ok: lookup argument + constraint-program gates (n=2^7, {len(gp)} bytes) byte-identical to the oracle")

    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from stark_airs import fibonacci, logup
    stark, trace, pis = fibonacci(8, pow_bits=8)
    sp = stark.prove(ctx, trace, pis)
    assert sp == oracle_lib.stark_prove(oracle, stark, trace, pis), "GPU STARK proof differs from the oracle's"
    stark.verify(pis, sp)
    print(f"smoke ok: vx_stark_prove (Fibonacci AIR, n=2^8, {len(sp)} bytes) byte-identical to the oracle and verifies")
    stark, trace, pis = logup(7, pow_bits=6)               # second commitment round: vx_stark_begin / vx_stark_finish
    sp = stark.prove(ctx, trace, pis)
    assert sp == oracle_lib.stark_prove(oracle, stark, trace, pis), "GPU two-round STARK proof differs from the oracle's"
    stark.verify(pis, sp)
    print(f"smoke ok: vx_stark_begin / vx_stark_finish (log-derivative lookup AIR, n=2^7, {len(sp)} bytes) byte-identical to the oracle")
