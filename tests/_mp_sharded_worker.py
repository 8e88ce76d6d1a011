"""Worker for tests/test_multiproc.py::test_sharded_commit_gloo — the sharded PolynomialBatch commitment
(vectorx_amd/sharded.py) on CPU tensors over gloo, with the ORACLE standing in for the two compute steps, so the
sharding / all-to-all / cap-gather index logic is checked end to end against the unsharded oracle commitment."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import oracle_lib  # noqa: E402
from vectorx_amd import dist_harness as H  # noqa: E402
from vectorx_amd import sharded  # noqa: E402


class OracleBackend:
    """CPU stand-in for sharded.GpuBackend (tests only)."""

    def __init__(self, oracle):
        self.o = oracle

    def empty(self, *shape):
        return torch.empty(shape, dtype=torch.int64)

    def lde_columns(self, values, log_n, rate_bits):
        v = values.numpy().view(np.uint64)
        r = self.o.commit(v, rate_bits, 0)                       # leaves [N][mc], rows in bit-reversed order
        return torch.from_numpy(np.ascontiguousarray(r["leaves"].T).view(np.int64))

    def hash_rows(self, rows, ncols, cap_height):
        a = rows.numpy().view(np.uint64)[:ncols]                 # [ncols][Nb] column-major
        _, cap = self.o.merkle(np.ascontiguousarray(a.T), cap_height)
        return cap


def main():
    rank, world, local_rank = H.env_rank()
    dist = H.init("gloo", local_rank)
    oracle = oracle_lib.load()
    oracle.L.vxo_set_num_threads(1)
    log_n, ncols, rate_bits, cap_h = 6, 21, 3, 4
    rng = np.random.default_rng(42)                                # same matrix on every rank
    vals = oracle_lib.rand_field(rng, (ncols, 1 << log_n))
    per, blocks = sharded.column_blocks(ncols, world)
    lo, hi = blocks[rank]
    local = np.zeros((per, 1 << log_n), np.uint64)
    local[: hi - lo] = vals[lo:hi]
    cap, rows = sharded.commit_sharded(OracleBackend(oracle), dist, torch.from_numpy(local.view(np.int64)), ncols, log_n,
                                       rate_bits, cap_h)
    ref = oracle.commit(vals, rate_bits, cap_h)
    ok_cap = bool((cap == ref["cap"]).all())
    # my row shard must be exactly rows [rank*Nb, (rank+1)*Nb) of the unsharded LDE, all real columns
    N = (1 << log_n) << rate_bits
    Nb = N // world
    mine = rows.numpy().view(np.uint64)[:ncols].T
    ok_rows = bool((mine == ref["leaves"][rank * Nb:(rank + 1) * Nb]).all())
    res = [None] * world
    if dist is not None:
        dist.all_gather_object(res, (rank, ok_cap, ok_rows))
    else:
        res = [(rank, ok_cap, ok_rows)]
    if rank == 0:
        print(json.dumps({"world": world, "results": res}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
