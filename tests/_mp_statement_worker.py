"""Worker for tests/test_header_range_dag.py: the DAG with per-job "tables" (hashlib stand-ins) and the production statements over gloo
ranks — the records a rank's reduce jobs consume were stated by jobs of the other rank."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import _cpu_tables  # noqa: E402
from vectorx_amd import dag_tables  # noqa: E402
from vectorx_amd import dist_harness as H  # noqa: E402
from vectorx_amd import header_range as hr  # noqa: E402
from vectorx_amd import mapreduce as mr  # noqa: E402


def main():
    rank, world, local_rank = H.env_rank()
    dist = H.init("gloo", local_rank)
    spec = mr.DagSpec(4, 10, 9, 11)
    kinds = ("map", "reduce", "outer") if rank == 0 else ("map", "reduce")
    per_kind, _, _ = dag_tables.build_per_job(None, [None], kinds=kinds, small=True, num_map=spec.num_map, factory=_cpu_tables.CpuTables())
    res = mr.run_dag(spec, lambda kind, log_n, jobs: _cpu_tables.TablesProver(kind, per_kind[kind]), dist, input_seed=b"two ranks")
    counts = [None] * world
    if dist is not None:
        dist.all_gather_object(counts, len(res["my_proofs"]))
    else:
        counts = [len(res["my_proofs"])]
    if rank == 0:
        req = hr.cached_request(b"two ranks", **dag_tables.request_shape(True, spec.num_map))
        print(json.dumps({"world": world, "root": res["root"].hex(), "per_rank": counts, "output_ok": res["root"][32:] == hr.expected_output(req)}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
