"""One small whole proof on cuda:0 checked against the oracle — called by __graft_entry__.smoke()."""


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
