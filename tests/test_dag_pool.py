"""vectorx_amd/dag_pool.py on the CPU: worker processes, job placement, the layer barriers, error propagation — with the stand-in
prover of tests/_pool_factory.py.  The GPU counterpart (real proofs, root == the one-process root) is tests/test_gpu_dag_pool.py."""
import pytest

from vectorx_amd import mapreduce as mr
from vectorx_amd.dag_pool import DagPool

import os
from pathlib import Path

import _pool_factory


@pytest.fixture(autouse=True)
def workers_find_the_factory(monkeypatch):
    monkeypatch.setenv("PYTHONPATH", str(Path(__file__).resolve().parent) + os.pathsep + os.environ.get("PYTHONPATH", ""))


def one_process_root(spec, seed, with_tables=True):
    class P(_pool_factory.FakeProver):
        def prove(self, key, pis, lane=0, input_seed=b"", **kw):
            return super().prove(key, pis, lane, input_seed=input_seed, with_tables=with_tables)
    res = mr.run_dag(spec, lambda kind, log_n, jobs: P(kind, delay=0.0), None, in_flight=2, input_seed=seed)
    return res["root"]


@pytest.mark.parametrize("workers,lanes", [(1, 1), (2, 2), (3, 1)])
@pytest.mark.parametrize("with_starks", [False, True])
def test_pool_root_equals_the_one_process_root(workers, lanes, with_starks):
    spec = mr.DagSpec(8, 10, 9, 11)
    pool = DagPool(spec, devices=(0,), workers_per_device=workers, lanes=lanes, factory="_pool_factory:make", with_starks=with_starks).start()
    try:
        ready = pool.wait_ready(timeout=120)
        assert sorted(r["worker"] for r in ready) == list(range(workers))
        cfg_tables = pool.cfg["with_starks"]
        for seed in (b"", b"request 7"):
            res = pool.run(seed)
            assert res["root"] == one_process_root(spec, seed) == pool.run(seed, schedule="layers")["root"]
            assert res["proofs"] == 8 + 7 + 1 and sum(res["jobs_by_worker"]) == 16
            assert [l["jobs"] for l in res["per_layer"]] == [8, 4, 2, 1, 1]
            assert res["outer_tables_hoisted"] == with_starks
            assert set(res["split"]) == {"plonky2", "trace_generation"} | ({"eddsa"} if cfg_tables else set())
        if workers > 1:
            assert min(res["jobs_by_worker"]) > 0, "every worker takes jobs"
        res = pool.run(b"x", with_tables=False)
        assert res["root"] == one_process_root(spec, b"x", with_tables=False) != pool.run(b"x")["root"]
    finally:
        pool.close()


def test_statements_travel_from_children_to_parents_in_every_scheduler():
    """A prover that emits statements: a job's record is its digest followed by its statement, a parent's prover receives its
    children's records, the root record ends with the outer job's statement — the same in the pool (tables hoisted or not: the
    statement of the outer job is made when its child is known), and in the one-process schedulers."""
    import hashlib
    import struct
    spec = mr.DagSpec(4, 10, 9, 11)
    leaves = [struct.pack("<I", j) * 2 for j in range(4)]
    l1 = [hashlib.sha256(leaves[0] + leaves[1]).digest(), hashlib.sha256(leaves[2] + leaves[3]).digest()]
    want = hashlib.sha256(l1[0] + l1[1]).digest()[::-1]
    pool = DagPool(spec, workers_per_device=2, lanes=2, factory="_pool_factory:make_stating", with_starks=True).start()
    try:
        pool.wait_ready(timeout=120)
        pool.load_request(b"r")                                # reaches the factory's hook on every worker; optional
        a = pool.run(b"r")
        b = pool.run(b"r", schedule="layers")
        plain = pool.run(b"r", with_tables=False)
    finally:
        pool.close()
    assert a["outer_tables_hoisted"] and not b["outer_tables_hoisted"]
    assert a["root"] == b["root"] and a["root"][32:] == want and len(a["root"]) == 64
    assert [len(r) for r in a["records"][0].values()] == [40] * 4 and len(plain["root"]) == 32
    make = lambda kind, log_n, jobs: _pool_factory.StatingProver(kind, delay=0.0)      # noqa: E731
    assert mr.run_dag(spec, make, None, in_flight=2, input_seed=b"r")["root"] == a["root"]
    assert mr.run_dag(spec, make, None, in_flight=2, input_seed=b"r", barriers=False)["root"] == a["root"]
    assert mr.run_dag(spec, make, None, in_flight=1, input_seed=b"r")["root"] == a["root"]


def test_the_outer_job_goes_to_worker_0_and_a_worker_error_surfaces():
    spec = mr.DagSpec(4, 10, 9, 11)
    pool = DagPool(spec, workers_per_device=2, lanes=1, factory="_pool_factory:make_failing").start()
    try:
        pool.wait_ready(timeout=120)
        with pytest.raises(RuntimeError, match="boom in reduce job 1"):
            pool.run(b"")
        # replies of the failed run may still be queued on the other connections: the pool says so instead of reading them as its next results
        with pytest.raises(RuntimeError, match="unusable after a failed run"):
            pool.run(b"")
    finally:
        pool.close()


def test_a_peer_that_connects_and_says_nothing_does_not_hang_the_coordinator():
    """Listener.accept() + a blocking recv() of the hello message let any local process that found the socket hang wait_ready"""
    import threading
    import time
    from multiprocessing.connection import Client
    spec = mr.DagSpec(4, 10, 9, 11)
    pool = DagPool(spec, workers_per_device=1, lanes=1, factory="_pool_factory:make").start()
    try:
        addr, key = pool._listener.address, pool._authkey
        def mute():
            try:
                c = Client(addr, family="AF_UNIX", authkey=key)      # knows the key, never says hello
                time.sleep(60)
                c.close()
            except Exception:
                pass
        threading.Thread(target=mute, daemon=True).start()
        t0 = time.perf_counter()
        ready = pool.wait_ready(timeout=120)
        assert len(ready) == 1 and time.perf_counter() - t0 < 100
        assert pool.run(b"q")["root"] == one_process_root(spec, b"q")
    finally:
        pool.close()


def test_two_devices_worth_of_workers_share_one_coordinator():
    spec = mr.DagSpec(4, 10, 9, 11)
    pool = DagPool(spec, devices=(0, 1), workers_per_device=1, lanes=2, factory="_pool_factory:make").start()
    try:
        ready = pool.wait_ready(timeout=120)
        assert [r["device"] for r in sorted(ready, key=lambda r: r["worker"])] == [0, 1]
        assert pool.run(b"s")["root"] == one_process_root(spec, b"s")
    finally:
        pool.close()


def test_a_worker_that_dies_during_setup_is_an_error_within_seconds():
    import time
    spec = mr.DagSpec(4, 10, 9, 11)
    pool = DagPool(spec, workers_per_device=2, lanes=1, factory="_pool_factory:make_dying").start()
    try:
        t0 = time.perf_counter()
        with pytest.raises(RuntimeError, match="exited with code 7"):
            pool.wait_ready(timeout=600)
        assert time.perf_counter() - t0 < 30
    finally:
        pool.close()
