"""Worker for tests/test_multiproc.py: the MapReduce proof DAG (vectorx_amd/mapreduce.py) over gloo ranks with the
ORACLE as the prover (tiny circuits) — checks layer barriers, digest exchange and that every proof verifies."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import oracle_lib  # noqa: E402
from vectorx_amd import dist_harness as H  # noqa: E402
from vectorx_amd import mapreduce as mr  # noqa: E402
from vectorx_amd.synth import SynthCircuit  # noqa: E402


class OracleProver:
    def __init__(self, oracle, kind, log_n, jobs, recursion=False):
        """recursion: the circuits of mapreduce.GpuProver's default (the recursive verifier's gate set, >= 2^5 rows); the gloo tests'
        2^3 / 2^4-row circuits keep the two-gate stand-in"""
        seed = {"map": 101, "reduce": 202, "outer": 303}[kind]
        shape = mr.circuit_shape(recursion)
        self.sc = SynthCircuit(log_n, seed=seed, poseidon_percent=50, witness_seed=0, **shape)
        self.oc = oracle_lib.OracleCircuit(oracle, self.sc.desc_ptr)
        self.wit = {}
        for (li, j) in jobs:
            sj = SynthCircuit(log_n, seed=seed, poseidon_percent=50, witness_seed=1000 * li + j + 1, **shape)
            self.wit[(li, j)] = sj.witness().copy()
            sj.free()

    def prove(self, key, pi, lane=0):
        w = self.wit[key]
        r0, r2 = self.sc.patch_public_inputs(pi)
        w[:, 0] = r0
        w[:, 2] = r2
        proof = self.oc.prove(w)
        assert self.oc.verify(proof) == "", "DAG proof does not verify"
        return proof


def main():
    rank, world, local_rank = H.env_rank()
    dist = H.init("gloo", local_rank)
    oracle = oracle_lib.load()
    oracle.L.vxo_set_num_threads(1)
    spec = mr.DagSpec(num_map=4, map_log_n=4, reduce_log_n=3, outer_log_n=4)
    res = mr.run_dag(spec, lambda kind, log_n, jobs: OracleProver(oracle, kind, log_n, jobs), dist)
    counts = [None] * world
    if dist is not None:
        dist.all_gather_object(counts, len(res["my_proofs"]))
    else:
        counts = [len(res["my_proofs"])]
    if rank == 0:
        print(json.dumps({"world": world, "root": res["root"].hex(), "proofs": res["proofs"], "per_rank": counts,
                          "layers": [(l["kind"], l["jobs"]) for l in res["per_layer"]]}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
