"""vectorx_amd/dag_pool.py with the real prover at tiny sizes: two worker PROCESSES of two lanes each on the one GPU, per-job STARK
tables generated on the device — the root digest must equal the one-process scheduler's (mapreduce.run_dag) for the same request."""
import pytest

import vectorx_amd as vx
from vectorx_amd import dag_tables
from vectorx_amd import mapreduce as mr
from vectorx_amd.dag_pool import DagPool

pytestmark = pytest.mark.gpu


def test_pool_root_equals_the_one_process_root_with_per_job_tables(ctx):
    spec = mr.DagSpec(4, 10, 9, 11)
    # started first: the workers only touch the GPU when wait_ready() configures them
    pool = DagPool(spec, devices=(0,), workers_per_device=2, lanes=2, with_starks=True, small_tables=True, table_mode="per_job").start()
    try:
        ready = pool.wait_ready(timeout=1500)
        assert len(ready) == 2 and {r["device"] for r in ready} == {0}
        with_tables = pool.run(b"request 1")
        again = pool.run(b"request 1", schedule="layers")
        assert with_tables["outer_tables_hoisted"] and not again["outer_tables_hoisted"]
        other = pool.run(b"request 2")
        plain = pool.run(b"request 1", with_tables=False)
    finally:
        pool.close()
    assert with_tables["root"] == again["root"] != other["root"]
    assert plain["root"] != with_tables["root"]
    assert with_tables["split"].get("trace_generation", 0) > 0 and "trace_generation" not in plain["split"]
    assert min(with_tables["jobs_by_worker"]) > 0

    # the same request through ONE process: run_dag with a GpuProver per kind and the same per-job tables
    per_kind, tables, _ = dag_tables.build(ctx, small=True, mode="per_job", lanes=[ctx])
    provers = {}

    def make(kind, log_n, jobs):
        provers[kind] = mr.GpuProver(ctx, kind, log_n, jobs, spec.poseidon_percent, distinct_witnesses=4, starks=per_kind[kind])
        return provers[kind]

    try:
        one = mr.run_dag(spec, make, None, ctx.sync, in_flight=1, input_seed=b"request 1")
    finally:
        for p in provers.values():
            p.free()
        for t in tables:
            t.free()
    assert one["root"] == with_tables["root"]
