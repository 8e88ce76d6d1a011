"""tests/golden/restatement_goldens.json — RESTATEMENT-GENERATED vectors (not upstream-generated; the reference holds
none for this path): the oracle must keep reproducing them (CPU), and the HIP path must reproduce them
through the C ABI (GPU) from data that travels with the repository."""
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

import oracle_lib
from vectorx_amd.synth import SynthCircuit

G = json.loads((Path(__file__).resolve().parent / "golden" / "restatement_goldens.json").read_text())
P = oracle_lib.P


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u8").tobytes()).hexdigest()


def ramp(ncols, n):
    i = np.arange(n, dtype=np.uint64)[None, :]
    c = np.arange(ncols, dtype=np.uint64)[:, None]
    return (i * (c + 1) + c) % np.uint64(P)


def ntt_input(e):
    n = 1 << e["log_n"]
    if e["input"] == "impulse1":
        a = np.zeros((1, n), np.uint64)
        a[0, 1] = 1
        return a
    return ramp(2, n)


def check_ntt(fn):
    for e in G["ntt"]:
        out = fn(ntt_input(e), e["kind"], 7)
        assert sha(out) == e["sha256"], e
        if "values" in e:
            assert [[int(v) for v in row] for row in out] == e["values"]


def check_merkle(fn):
    for e in G["merkle"]:
        leaves = np.ascontiguousarray(ramp(e["width"], e["n_leaves"]).T)
        dig, cap = fn(leaves, e["cap_height"])
        assert sha(dig) == e["digests_sha256"], e
        assert [[int(v) for v in h] for h in cap] == e["cap"]


def test_note_says_what_these_are():
    assert "NOT upstream-generated" in G["_note"]


def test_impulse_transform_is_the_root_powers(oracle):
    """independent anchor for the NTT vectors: fft(impulse at 1)[k] = w^k"""
    e = next(x for x in G["ntt"] if x["log_n"] == 3 and x["kind"] == 0 and x["input"] == "impulse1")
    w = pow(7, (P - 1) // 8, P)
    assert e["values"][0] == [pow(w, k, P) for k in range(8)]


def test_oracle_reproduces_ntt_goldens(oracle):
    check_ntt(oracle.ntt_batch)


def test_oracle_reproduces_merkle_and_commit_goldens(oracle):
    check_merkle(oracle.merkle)
    for e in G["commit"]:
        r = oracle.commit(ramp(e["ncols"], 1 << e["log_n"]), e["rate_bits"], e["cap_height"])
        assert sha(r["coeffs"]) == e["coeffs_sha256"] and sha(r["leaves"]) == e["leaves_sha256"]
        assert [[int(v) for v in h] for h in r["cap"]] == e["cap"]


def test_oracle_reproduces_proof_goldens(oracle):
    for e in G["proof"]:
        sc = SynthCircuit(e["degree_bits"], seed=e["seed"], poseidon_percent=e["poseidon_percent"], flags=e["flags"])
        assert sha(sc.witness()) == e["witness_sha256"]          # the generator is part of what is frozen
        oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
        assert [int(v) for v in oc.digest()] == e["circuit_digest"]
        proof = oc.prove(sc.witness())
        assert len(proof) == e["proof_len"] and hashlib.sha256(proof).hexdigest() == e["proof_sha256"]
        assert int(np.frombuffer(proof[-40:-32], dtype="<u8")[0]) == e["pow_witness"]


@pytest.mark.gpu
def test_gpu_reproduces_ntt_merkle_commit_goldens(ctx):
    import vectorx_amd as vx
    check_ntt(ctx.ntt_batch)
    check_merkle(ctx.merkle_cap)
    for e in G["commit"]:
        b = vx.PolynomialBatch.from_values(ctx, ramp(e["ncols"], 1 << e["log_n"]), rate_bits=e["rate_bits"], cap_height=e["cap_height"])
        assert [[int(v) for v in h] for h in b.cap()] == e["cap"]
        b.free()


@pytest.mark.gpu
def test_gpu_reproduces_proof_goldens(ctx):
    import vectorx_amd as vx
    for e in G["proof"]:
        sc = SynthCircuit(e["degree_bits"], seed=e["seed"], poseidon_percent=e["poseidon_percent"], flags=e["flags"])
        c = vx.Circuit(ctx, sc.desc_ptr)
        assert [int(v) for v in c.digest()] == e["circuit_digest"]
        proof = c.prove(sc.witness())
        assert len(proof) == e["proof_len"] and hashlib.sha256(proof).hexdigest() == e["proof_sha256"]
        c.free()
