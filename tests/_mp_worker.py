"""Worker for tests/test_multiproc.py: the N>1 harness path of bench.py on CPU (gloo), no GPU needed.
Each rank 'proves' its own unit with the ORACLE (tests may use it) so the sharding/timing/aggregation logic —
per-rank seeds, barrier-bracketed timed region, MAX over ranks, whole-job value — is exercised end to end."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import oracle_lib  # noqa: E402
from vectorx_amd import dist_harness as H  # noqa: E402
from vectorx_amd.synth import SynthCircuit  # noqa: E402


def main():
    rank, world, local_rank = H.env_rank()
    dist = H.init("gloo", local_rank)
    oracle = oracle_lib.load()
    oracle.L.vxo_set_num_threads(1)
    sc = SynthCircuit(4, seed=0x5EED0000 + rank, poseidon_percent=50)   # per-rank witness, as bench.py does
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    proofs = []

    def step():
        proofs.append(oc.prove(sc.witness()))

    dt = H.run_timed(step, steps=2, warmup=1, sync=lambda: None, dist=dist)
    ok = all(oc.verify(p) == "" for p in proofs)
    # every rank must see the same MAX time
    import torch
    t = torch.tensor([dt], dtype=torch.float64)
    lo = t.clone()
    if dist is not None:
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    pis = sc.public_inputs().tolist()
    gathered = [None] * world
    if dist is not None:
        dist.all_gather_object(gathered, (rank, ok, pis))
    else:
        gathered = [(rank, ok, pis)]
    if rank == 0:
        agg = H.aggregate(world, 2, dt)
        print(json.dumps({"world": world, "dt_max": dt, "dt_min_of_max": float(lo.item()), "value": agg["value"],
                          "ranks": gathered}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
