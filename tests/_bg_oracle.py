"""Background worker of the GPU suite (tests/conftest.py starts it when the two full-size byte-identity tests are selected): the ORACLE's
proofs of the 2^21- and 2^20-row bench circuits, computed on the host cores WHILE the other GPU tests run — the suite spent 365 of its
840 s waiting for these two proofs with the GPU idle.  Same circuit generator parameters as the tests, same oracle; a test
that finds no result here (run on its own, worker failed) computes the proof itself.

    python tests/_bg_oracle.py OUTDIR LOGN:SEED:PCT [LOGN:SEED:PCT ...]
writes OUTDIR/oracle_proof_<LOGN>_<SEED>.bin (+ .json: seconds, circuit digest, cap sha256), atomically, one job after the other."""
import hashlib
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")      # do not spin at barriers: the cores are shared with the tests in the foreground
    import bench_prove
    import oracle_lib
    from vectorx_amd.synth import SynthCircuit
    out = Path(sys.argv[1])
    oracle = oracle_lib.load()
    oracle.L.vxo_set_num_threads(bench_prove.usable_cores())
    for spec in sys.argv[2:]:
        log_n, seed, pct = (int(x, 0) for x in spec.split(":"))
        t0 = time.perf_counter()
        sc = SynthCircuit(log_n, seed=seed, poseidon_percent=pct)
        oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
        proof = oc.prove(sc.witness())
        rec = {"log_n": log_n, "seed": seed, "poseidon_percent": pct, "seconds": round(time.perf_counter() - t0, 1),
               "digest": [int(x) for x in oc.digest()], "cap_sha256": hashlib.sha256(oc.cap().tobytes()).hexdigest(),
               "proof_sha256": hashlib.sha256(proof).hexdigest()}
        tmp = out / f".tmp_{log_n}_{seed}"
        tmp.write_bytes(proof)
        (out / f"oracle_proof_{log_n}_{seed}.json").write_text(json.dumps(rec))
        tmp.rename(out / f"oracle_proof_{log_n}_{seed}.bin")          # the .bin appears last and whole
        oc.free()
        sc.free()


if __name__ == "__main__":
    main()
