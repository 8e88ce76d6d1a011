#!/usr/bin/env python3
"""header_range_512 as the reference really proves it: a MapReduce DAG of 64 map + 63 reduce + 1 outer plonky2 proofs
(SURVEY.md §8 f-1), scheduled over the G GPUs of a node by vectorx_amd/mapreduce.py.

    python tools/dag_bench.py --num-map 64 --map-log-n 18 --reduce-log-n 16 --outer-log-n 19
    python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 --master-port P tools/dag_bench.py ...

The circuit SIZES are stand-ins (the real map/reduce/outer degrees need the Rust builder); the DAG, its layer barriers
and the prover are real.  Witnesses are resident in HBM before the timed region (witness generation is CPU work outside
the hot path).  Not part of bench.py's contract; only the world-1 path could be run on this pool (one GPU per call).
"""
import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-map", type=int, default=64)
    ap.add_argument("--map-log-n", type=int, default=18)
    ap.add_argument("--reduce-log-n", type=int, default=16)
    ap.add_argument("--outer-log-n", type=int, default=19)
    ap.add_argument("--in-flight", type=int, default=1, help="jobs of a layer kept in flight per GPU (contexts / host threads)")
    ap.add_argument("--no-barriers", action="store_true", help="single process only: dependency-driven schedule (a job starts when its own children are done)")
    args = ap.parse_args()
    import torch
    import vectorx_amd as vx
    from vectorx_amd import dist_harness as H
    from vectorx_amd import mapreduce as mr

    rank, world, local_rank = H.env_rank()
    dist = H.init("nccl", local_rank)
    ctx = vx.Context(local_rank)
    lanes = [vx.Context(local_rank) for _ in range(args.in_flight - 1)]
    spec = mr.DagSpec(args.num_map, args.map_log_n, args.reduce_log_n, args.outer_log_n)
    provers = []

    def make(kind, log_n, jobs):
        p = mr.GpuProver(ctx, kind, log_n, jobs, spec.poseidon_percent, extra_lanes=lanes, recursion=spec.recursion)
        provers.append(p)
        return p

    def sync():
        ctx.sync()
        for l in lanes:
            l.sync()
        torch.cuda.synchronize()

    res = mr.run_dag(spec, make, dist, sync, in_flight=args.in_flight, barriers=not args.no_barriers)
    secs = res["seconds"]
    if dist is not None:
        t = torch.tensor([secs], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        secs = float(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": "header_range DAG proofs/sec (one header_range proof = map + reduce tree + outer plonky2 proofs)",
            "value": 1.0 / secs, "unit": "header_range proofs/sec", "n_gpus": world, "dag_seconds": secs,
            "plonky2_proofs_per_dag": res["proofs"], "plonky2_proofs_per_sec": res["proofs"] / secs, "scaling": "strong",
            "root": res["root"].hex(),
            "per_layer_ms": [(l["kind"], l["jobs"], round(l["ms"], 1)) for l in res["per_layer"]],
            "config": {"num_map": args.num_map, "map_log_n": args.map_log_n, "reduce_log_n": args.reduce_log_n,
                       "outer_log_n": args.outer_log_n, "in_flight_per_gpu": args.in_flight,
                       "note": "circuit sizes are synthetic stand-ins"}}), flush=True)
    for p in provers:
        p.free()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    for l in lanes:
        l.close()
    ctx.close()


if __name__ == "__main__":
    main()
