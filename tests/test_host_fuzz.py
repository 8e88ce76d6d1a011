"""Mutation fuzz of the host-side byte consumers of libvxprover.so (no GPU): vx_verify_standalone, vx_circuit_parse and
vx_stark_verify must refuse every mutated input with an error code — no crash, no hang, no acceptance."""
import numpy as np
import pytest

import oracle_lib
import vectorx_amd as vx
from stark_airs import logup
from vectorx_amd.synth import SynthCircuit


@pytest.fixture(scope="module")
def oracle():
    return oracle_lib.load()


def test_mutated_proofs_and_circuit_files_are_refused(oracle):
    rng = np.random.default_rng(5)
    sc = SynthCircuit(6, seed=4, poseidon_percent=40, flags=16 | 1)
    sc.desc.pow_bits = 4
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    proof, cap = oc.prove(sc.witness()), oc.cap()
    blob = vx.circuit_serialize(sc.desc_ptr, cap, True)
    stark, trace, pis = logup(6, pow_bits=4)
    sproof = oracle_lib.stark_prove(oracle, stark, trace, pis)
    vx.verify_standalone(sc.desc_ptr, cap, proof)
    vx.ParsedCircuit(blob)
    stark.verify(pis, sproof)
    for it in range(900):
        kind = it % 3
        b = bytearray([proof, blob, sproof][kind])
        m = int(rng.integers(0, 4))
        if m == 0:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif m == 1:
            b = b[:int(rng.integers(0, len(b)))]
        elif m == 2:
            i = int(rng.integers(0, len(b) - 8))
            b[i:i + 8] = rng.integers(0, 256, 8, dtype=np.uint8).tobytes()
        else:
            b += bytes(int(rng.integers(1, 64)))
        with pytest.raises(vx.VxError):
            if kind == 0:
                vx.verify_standalone(sc.desc_ptr, cap, bytes(b))
            elif kind == 1:
                vx.ParsedCircuit(bytes(b))
            else:
                stark.verify(pis, bytes(b))
