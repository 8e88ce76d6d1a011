#!/usr/bin/env python3
"""Generates tests/golden/chip_goldens.json — RESTATEMENT-GENERATED, NOT UPSTREAM-GENERATED (and the chips are this repository's own
AIRs, not Curta's).  Freezes, for the four chip-sized tables on fixed inputs: the SHA-256 of the constraint program, the public
inputs, the SHA-256 of the trace, and the SHA-256 of the ORACLE's STARK proof — so that a later edit of an AIR emitter, of a trace
generator, of the STARK transcript or of the oracle cannot change them silently together with the GPU path they check
(tests/test_chip_goldens.py on the CPU; the GPU tests compare GPU and oracle proofs live).

    python tests/golden/make_chip_goldens.py        (needs oracle/liboracle.so)
"""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib  # noqa: E402

CFG = dict(num_query_rounds=12, pow_bits=4)
MESSAGES = [b"abc", b"", bytes(range(200))]


def cases():
    from vectorx_amd import blake2b_air, ed25519_air, sha256_air, sha512_air
    yield "sha256", sha256_air.make_stark(8, **CFG), sha256_air.generate_trace(8, MESSAGES)[:2], sha256_air.build_program()[0]
    yield "blake2b", blake2b_air.make_stark(9, **CFG), blake2b_air.generate_trace(9, MESSAGES)[:2], blake2b_air.build_program()[0]
    yield "ed25519", ed25519_air.make_stark(10, **CFG), ed25519_air.generate_trace(10, 0xC0FFEE11)[:2], ed25519_air.build_program()[0]
    yield "sha512", sha512_air.make_stark(8, **CFG), sha512_air.generate_trace(8, MESSAGES)[:2], sha512_air.build_program()[0]


def sha_u64(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u8").tobytes()).hexdigest()


def compute():
    o = oracle_lib.load()
    out = {"_note": "restatement-generated, not upstream-generated; own AIRs, not Curta's; config " + json.dumps(CFG)}
    for name, stark, (trace, pis), prog in cases():
        proof = oracle_lib.stark_prove(o, stark, trace, pis)
        out[name] = {"program_sha256": sha_u64(np.array(prog, dtype=np.uint64)), "program_words": len(prog), "trace_shape": list(trace.shape),
                     "trace_sha256": sha_u64(trace), "public_inputs": [int(x) for x in pis], "oracle_proof_bytes": len(proof),
                     "oracle_proof_sha256": hashlib.sha256(proof).hexdigest()}
    return out


if __name__ == "__main__":
    (Path(__file__).resolve().parent / "chip_goldens.json").write_text(json.dumps(compute(), indent=1) + "\n")
