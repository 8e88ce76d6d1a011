"""`.vxcircuit` — save -> load round trips of the compiled-circuit container, host side (no GPU): the analogue of the
reference's `circuit.test_serializers(&gate_registry, &hint_registry)` (/root/reference/circuits/header_range.rs:117-126,
circuits/rotate.rs:152-161, circuits/builder/subchain_verification.rs:479-482).  The GPU side (vx_circuit_load -> identical
proofs) is tests/test_gpu_boundary.py."""
import ctypes

import numpy as np
import pytest

import oracle_lib
import vectorx_amd as vx
from vectorx_amd.synth import CircuitDesc, SynthCircuit

INT_FIELDS = ["degree_bits", "num_wires", "num_routed_wires", "num_challenges", "rate_bits", "cap_height", "pow_bits", "num_query_rounds",
              "quotient_degree_factor", "num_gates", "num_selectors", "num_constants", "num_public_inputs", "programs_len",
              "override_flags", "hiding", "num_partial_products"]


def _arr(ptr, n, ty):
    return list((ty * n).from_address(ptr)) if n and ptr else []


def _same_description(a: CircuitDesc, b: CircuitDesc):
    for f in INT_FIELDS:
        assert getattr(a, f) == getattr(b, f), f
    n = a.num_gates
    for f in ["gate_types", "gate_params", "selector_indices", "group_starts", "group_ends"]:
        assert _arr(getattr(a, f), n, ctypes.c_int32) == _arr(getattr(b, f), n, ctypes.c_int32), f
    assert _arr(a.k_is, a.num_routed_wires, ctypes.c_uint64) == _arr(b.k_is, a.num_routed_wires, ctypes.c_uint64)
    assert _arr(a.pi_rows, a.num_public_inputs, ctypes.c_uint32) == _arr(b.pi_rows, a.num_public_inputs, ctypes.c_uint32)
    assert _arr(a.pi_cols, a.num_public_inputs, ctypes.c_uint32) == _arr(b.pi_cols, a.num_public_inputs, ctypes.c_uint32)
    assert _arr(a.programs, a.programs_len, ctypes.c_uint64) == _arr(b.programs, a.programs_len, ctypes.c_uint64)
    if a.programs_len:
        assert _arr(a.program_offsets, n, ctypes.c_int32) == _arr(b.program_offsets, n, ctypes.c_int32)


@pytest.mark.parametrize("degree_bits,flags", [(3, 0), (5, 1), (6, 15), (8, 0)])
def test_save_load_round_trip(oracle, degree_bits, flags):
    sc = SynthCircuit(degree_bits, seed=600 + degree_bits, poseidon_percent=40, flags=flags)
    sc.desc.pow_bits = 5
    if degree_bits == 6:                       # the optional fields travel too
        sc.set_fri_reduction_arity_bits([3, 1])
        sc.set_circuit_digest([11, 22, 33, 44])
        sc.set_num_partial_products(9)
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    cap = oc.cap()
    blob = vx.circuit_serialize(sc.desc_ptr, cap, with_preprocessed=True)
    again = vx.circuit_serialize(sc.desc_ptr, cap, with_preprocessed=True)
    assert blob == again and blob[:8] == b"VXCIRCT1"                      # deterministic
    pc = vx.ParsedCircuit(blob)
    _same_description(sc.desc, pc.desc)
    assert (pc.cap == cap).all()
    n_vals = (sc.desc.num_constants + sc.desc.num_routed_wires) << degree_bits
    assert _arr(pc.desc.constants_sigmas, n_vals, ctypes.c_uint64) == _arr(sc.desc.constants_sigmas, n_vals, ctypes.c_uint64)
    if degree_bits == 6:
        assert _arr(pc.desc.fri_reduction_arity_bits, 2, ctypes.c_int32) == [3, 1]
        assert list(pc.desc.circuit_digest) == [11, 22, 33, 44]
    assert vx.circuit_serialize(pc.desc_ptr, pc.cap, with_preprocessed=True) == blob   # save(load(save(x))) == save(x)
    # the loaded description drives the oracle to the same digest, cap and proof as the original one
    oc2 = oracle_lib.OracleCircuit(oracle, pc.desc_ptr)
    assert (oc2.digest() == oc.digest()).all() and (oc2.cap() == cap).all()
    w = sc.witness()
    proof = oc.prove(w)
    assert oc2.prove(w) == proof
    # verifier-only file: no preprocessed values, the cap instead — enough for vx_verify_standalone
    vblob = vx.circuit_serialize(sc.desc_ptr, cap, with_preprocessed=False)
    assert len(vblob) < len(blob)
    vc = vx.ParsedCircuit(vblob)
    assert not vc.desc.constants_sigmas
    vx.verify_standalone(vc.desc_ptr, vc.cap, proof)


def test_corrupt_and_truncated_files_are_rejected(oracle):
    sc = SynthCircuit(4, seed=9, poseidon_percent=50, flags=1)
    blob = vx.circuit_serialize(sc.desc_ptr, None, with_preprocessed=True)
    vx.ParsedCircuit(blob)
    rng = np.random.default_rng(3)
    for off in [0, 7, 9, 12, 16, 20, 60, 100, len(blob) // 2, len(blob) - 9, len(blob) - 1] + [int(x) for x in rng.integers(0, len(blob), 40)]:
        bad = bytearray(blob)
        bad[off] ^= 1 << int(rng.integers(0, 8))
        with pytest.raises(vx.VxError) as e:
            vx.ParsedCircuit(bytes(bad))
        assert e.value.code == vx.VX_E_INVALID, off
    for bad in (b"", blob[:50], blob[:-8], blob[:-1], blob + b"\0" * 8, b"VXCIRCT1" + b"\0" * 200):
        with pytest.raises(vx.VxError):
            vx.ParsedCircuit(bad)
    # a structurally valid file whose description is out of range is refused by the shared validator
    sc.desc.cap_height = 40
    with pytest.raises(vx.VxError):
        vx.circuit_serialize(sc.desc_ptr, None, with_preprocessed=True)
