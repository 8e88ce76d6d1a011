#!/usr/bin/env python3
"""Latency-mode (strong-scaling) measurement of the SHARDED wire commitment across G GPUs
(BASELINE.json configs[3]: column-shard iNTT+LDE -> one RCCL all-to-all over xGMI -> row-shard hashing).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 --master-port P \
        tools/sharded_commit_bench.py --log-n 21 --steps 5 --warmup 2

NOT part of bench.py's contract (whose N-GPU mode is proof-level sharding): this pool only hands out one GPU per
call, so the G > 1 path has been exercised on CPU/gloo (tests/test_multiproc.py) and on one GPU (world 1) only.
Prints one JSON line on rank 0: commits/sec of ONE trace split over G GPUs, with the all-to-all time broken out.
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=21)
    ap.add_argument("--ncols", type=int, default=135)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    import torch
    import vectorx_amd as vx
    from vectorx_amd import dist_harness as H
    from vectorx_amd import sharded

    rank, world, local_rank = H.env_rank()
    dist = H.init("nccl", local_rank)
    ctx = vx.Context(local_rank)
    dev = torch.device("cuda", local_rank)
    be = sharded.GpuBackend(ctx, dev)
    per, blocks = sharded.column_blocks(args.ncols, world)
    lo, hi = blocks[rank]
    rng = np.random.default_rng(0x5EED0000)
    local = np.zeros((per, 1 << args.log_n), np.uint64)
    for c in range(args.ncols):                       # same global matrix regardless of G
        col = rng.integers(0, 0xFFFFFFFF00000001, size=1 << args.log_n, dtype=np.uint64)
        if lo <= c < hi:
            local[c - lo] = col
    d_local = be.from_host(local)
    caps = []

    def step():
        cap, _ = sharded.commit_sharded(be, dist, d_local, args.ncols, args.log_n, 3, 4)
        caps.append(cap)

    def sync():
        ctx.sync()
        torch.cuda.synchronize()

    dt = H.run_timed(step, args.steps, args.warmup, sync, dist, device=dev)
    if rank == 0:
        print(json.dumps({"metric": "sharded wire-trace commits/sec (one trace over G GPUs)", "value": args.steps / dt,
                          "n_gpus": world, "ms_per_commit": dt / args.steps * 1e3, "scaling": "strong",
                          "cap0": [int(x) for x in caps[-1][0]],
                          "config": {"workload": f"PolynomialBatch::from_values n=2^{args.log_n} x {args.ncols} cols, blowup 8",
                                     "parallelism": f"column-shard -> all-to-all -> row-shard, G={world}"}}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
