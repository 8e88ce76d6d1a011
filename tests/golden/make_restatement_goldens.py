#!/usr/bin/env python3
"""Generates tests/golden/restatement_goldens.json — RESTATEMENT-GENERATED, NOT UPSTREAM-GENERATED.

The reference's own tests hold no golden NTT output, Merkle cap or proof, so these vectors do not
pin the oracle to plonky2; they freeze the oracle's CURRENT answers on deterministic inputs so that (i) a later edit
of the oracle cannot silently change them together with the GPU path it checks, and (ii) the GPU path is compared
with data that travels with the repository.  Inputs are closed-form (impulse, ramp, i*j+1), outputs are stored whole
when small and as SHA-256 of the little-endian u64 stream otherwise.

    python tests/golden/make_restatement_goldens.py        (needs oracle/liboracle.so and vectorx_amd/libvxsynth.so)
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
from vectorx_amd.synth import SynthCircuit  # noqa: E402

P = oracle_lib.P


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u8").tobytes()).hexdigest()


def ramp(ncols, n):
    """column c, row i -> (i * (c + 1) + c) mod p : closed form, no PRNG"""
    i = np.arange(n, dtype=np.uint64)[None, :]
    c = np.arange(ncols, dtype=np.uint64)[:, None]
    return (i * (c + 1) + c) % np.uint64(P)


def main():
    o = oracle_lib.load()
    g = {"_note": "RESTATEMENT-GENERATED (oracle/ = CPU restatement of plonky2 v0.2.0), NOT upstream-generated; see make_restatement_goldens.py",
         "ntt": [], "merkle": [], "commit": [], "proof": []}
    # NTT of an impulse at index 1 and of a ramp, every kind (0 fft, 1 ifft, 2 coset_fft, 3 coset_ifft; shift 7)
    for log_n in (3, 8, 13):
        n = 1 << log_n
        imp = np.zeros((1, n), np.uint64)
        imp[0, 1] = 1
        for kind in (0, 1, 2, 3):
            for name, cols in (("impulse1", imp), ("ramp2", ramp(2, n))):
                out = o.ntt_batch(cols, kind, 7)
                e = {"log_n": log_n, "kind": kind, "input": name, "sha256": sha(out)}
                if log_n == 3:
                    e["values"] = [[int(v) for v in row] for row in out]
                g["ntt"].append(e)
    # Merkle caps of ramp matrices (rows = leaves)
    for n_leaves, width, cap_h in ((16, 3, 0), (64, 9, 2), (256, 135, 4)):
        leaves = np.ascontiguousarray(ramp(width, n_leaves).T)
        dig, cap = o.merkle(leaves, cap_h)
        g["merkle"].append({"n_leaves": n_leaves, "width": width, "cap_height": cap_h, "digests_sha256": sha(dig),
                            "cap": [[int(v) for v in h] for h in cap]})
    # PolynomialBatch::from_values of a ramp trace
    for log_n, ncols in ((4, 5), (8, 20)):
        r = o.commit(ramp(ncols, 1 << log_n), 3, 4)
        g["commit"].append({"log_n": log_n, "ncols": ncols, "rate_bits": 3, "cap_height": 4,
                            "coeffs_sha256": sha(r["coeffs"]), "leaves_sha256": sha(r["leaves"]),
                            "cap": [[int(v) for v in h] for h in r["cap"]]})
    # whole proofs of the synthetic circuit (deterministic: smallest proof-of-work witness)
    for degree_bits, seed, pct, flags in ((3, 1, 50, 0), (5, 2, 40, 0), (6, 3, 40, 15)):
        sc = SynthCircuit(degree_bits, seed=seed, poseidon_percent=pct, flags=flags)
        oc = oracle_lib.OracleCircuit(o, sc.desc_ptr)
        proof = oc.prove(sc.witness())
        assert oc.verify(proof) == ""
        pw = int(np.frombuffer(proof[-40:-32], dtype="<u8")[0])
        g["proof"].append({"degree_bits": degree_bits, "seed": seed, "poseidon_percent": pct, "flags": flags,
                           "circuit_digest": [int(v) for v in oc.digest()], "witness_sha256": sha(sc.witness()),
                           "proof_len": len(proof), "proof_sha256": hashlib.sha256(proof).hexdigest(), "pow_witness": pw,
                           "wires_cap_first_hash": [int(v) for v in np.frombuffer(proof[:32], dtype="<u8")]})
    out = Path(__file__).resolve().parent / "restatement_goldens.json"
    out.write_text(json.dumps(g, indent=1) + "\n")
    print("wrote", out, out.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
