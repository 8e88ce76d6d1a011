"""quotient_degree_factor below the blow-up (CircuitConfig::max_quotient_degree_factor < 8) on the CPU side: the oracle prover
and verifier, the product library's stand-alone verifier (vx_verify_standalone, host code of libvxprover.so — no GPU), the
description checks and the .vxcircuit container.  plonky2x builds with standard_recursion_config (8); smaller factors
change the partial-product chunking (chunks of qdf wires), the selector grouping (max_degree = qdf + 1), the number of
quotient chunks (qdf per challenge, after trim_to_len) and the lookup polynomials' degree (qdf - 1).
The GPU prover's side of this is tests/test_gpu_prover.py::test_quotient_degree_factor_below_the_blowup."""
import ctypes

import numpy as np
import pytest

import oracle_lib
import vectorx_amd as vx
from vectorx_amd.synth import SynthCircuit

P = oracle_lib.P


@pytest.fixture(scope="module")
def oracle():
    return oracle_lib.load()


@pytest.mark.parametrize("qdf,flags,degree_bits", [(3, 0, 5), (4, 1, 6), (5, 16, 6), (6, 2 | 1, 6), (7, 16 | 1, 7)])
def test_oracle_and_product_verifier_accept(oracle, qdf, flags, degree_bits):
    sc = SynthCircuit(degree_bits, seed=500 + qdf, poseidon_percent=50, flags=flags, quotient_degree_factor=qdf)
    assert sc.desc.quotient_degree_factor == qdf
    sc.desc.pow_bits = 4
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    w = sc.witness()
    proof = oc.prove(w)
    assert oc.verify(proof) == ""
    vx.verify_standalone(sc.desc_ptr, oc.cap(), proof)
    # qdf quotient chunks per challenge and ceil(80 / qdf) - 1 partial products: the proof sizes differ from the factor-8 circuit's
    npp = (80 + qdf - 1) // qdf - 1
    assert sc.desc.num_partial_products in (0, npp)
    # tampering with an opening is caught by both verifiers
    bad = bytearray(proof)
    bad[len(bad) // 3] ^= 1
    assert oc.verify(bytes(bad)) != ""
    with pytest.raises(vx.VxError):
        vx.verify_standalone(sc.desc_ptr, oc.cap(), bytes(bad))
    # an unsatisfied witness: the quotient no longer fits qdf * n coefficients, which plonky2 reports at prove time
    # (trim_to_len(..).expect("Quotient has failed, the vanishing polynomial is not divisible by Z_H"))
    wb = w.copy()
    wb[3, 9] = (int(wb[3, 9]) + 1) % P
    with pytest.raises(RuntimeError):
        oc.prove(wb)


def test_description_checks_follow_the_filtered_degree(oracle):
    """A gate whose filtered constraint degree exceeds quotient_degree_factor + 1 is refused: lowering the factor of a
    circuit that was grouped for 8 must fail, and so must a factor above the blow-up."""
    sc = SynthCircuit(6, seed=3, poseidon_percent=50)
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    proof = oc.prove(sc.witness())
    cap = oc.cap()
    vx.verify_standalone(sc.desc_ptr, cap, proof)
    for bad_qdf in (6, 4, 0, 9, 16):
        sc.desc.quotient_degree_factor = bad_qdf
        with pytest.raises(vx.VxError) as e:
            vx.verify_standalone(sc.desc_ptr, cap, proof)
        assert e.value.code == vx.VX_E_INVALID
    sc.desc.quotient_degree_factor = 8
    vx.verify_standalone(sc.desc_ptr, cap, proof)


def test_vxcircuit_container_keeps_the_factor(oracle):
    sc = SynthCircuit(5, seed=9, poseidon_percent=0, flags=1, quotient_degree_factor=4)
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    blob = vx.circuit_serialize(sc.desc_ptr, oc.cap(), False)
    pc = vx.ParsedCircuit(blob)
    assert pc.desc.quotient_degree_factor == 4
    proof = oc.prove(sc.witness())
    vx.verify_standalone(pc.desc_ptr, pc.cap, proof)
