"""BASELINE.json configs[2] byte for byte, in the driver-run suite: the GPU proof of the header_range_512 stand-in
(n = 2^21 rows x 135 wires — the circuit and witness bench.py times) equals the ORACLE's proof of the same circuit and witness.  The
oracle needs about four minutes for it on the GPU box's 16 host cores (61 s to commit the preprocessed polynomials, ~180 s to prove),
so the file is named to run LAST: under `pytest -x` every faster test has had its say before this one starts — and since round 6 the
oracle's proof is computed in the BACKGROUND of the whole GPU run (tests/conftest.py, tests/_bg_oracle.py: same generator parameters,
same oracle), so that the suite no longer waits four minutes with the GPU idle; run on its own, the test asks the oracle itself.
(tools/full_size_parity.py is the same check as a command-line tool; profiles/r0N_full_size_parity.jsonl hold its earlier records.)"""
import hashlib

import pytest

import bench_prove
import oracle_lib
import vectorx_amd as vx
from vectorx_amd.synth import SynthCircuit

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(1500)
def test_header_range_512_sized_proof_bytes_identical_to_oracle(ctx, oracle, background_oracle_proof):
    log_n = 21
    oracle.L.vxo_set_num_threads(bench_prove.usable_cores())
    sc = SynthCircuit(log_n, seed=0x5EED0000, poseidon_percent=50)          # bench.py's circuit and (rank 0) witness
    w = sc.witness()
    gc = vx.Circuit(ctx, sc.desc_ptr)
    gp = gc.prove(w)
    gc.verify(gp)
    gpu_digest, cap = gc.digest().copy(), gc.constants_sigmas_cap().copy()
    gc.free()                                                                 # 11 GB of HBM back before the CPU takes over
    bg = background_oracle_proof(log_n, 0x5EED0000)     # made by tests/_bg_oracle.py while the earlier tests ran (same circuit, same witness)
    if bg is not None:
        op, rec = bg
        assert rec["digest"] == [int(x) for x in gpu_digest] and rec["cap_sha256"] == hashlib.sha256(cap.tobytes()).hexdigest()
        assert len(gp) == len(op) and hashlib.sha256(gp).hexdigest() == hashlib.sha256(op).hexdigest() == rec["proof_sha256"]
        assert gp == op
    else:
        oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
        assert (oc.digest() == gpu_digest).all() and (oc.cap() == cap).all()
        op = oc.prove(w)
        assert len(gp) == len(op) and hashlib.sha256(gp).hexdigest() == hashlib.sha256(op).hexdigest()
        assert gp == op
        assert oc.verify(gp) == ""
        oc.free()
    sc.free()
