"""world_size-2 gloo test of the one-process-per-GPU harness (CPU only): the path bench.py takes for N>1."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_proof_level_sharding_gloo():
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "tests" / "_mp_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["world"] == 2
    assert abs(out["dt_max"] - out["dt_min_of_max"]) < 1e-12          # all ranks agreed on the MAX
    assert abs(out["value"] - 2 * 2 / out["dt_max"]) < 1e-9            # whole-job rate = world * steps / max time
    ranks = sorted(out["ranks"])
    assert [r[0] for r in ranks] == [0, 1] and all(r[1] for r in ranks)  # both ranks produced verifying proofs
    assert ranks[0][2] != ranks[1][2]                                   # ...of DIFFERENT witnesses (sharded units)


def test_single_process_harness():
    from vectorx_amd import dist_harness as H
    calls = []
    dt = H.run_timed(lambda: calls.append(1), steps=3, warmup=2, sync=lambda: None, dist=None)
    assert len(calls) == 5 and dt > 0
    assert abs(H.aggregate(4, 3, 0.5)["value"] - 24.0) < 1e-12


def _run_workers(script, nproc):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(ROOT / "tests" / script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_sharded_commit_gloo_world2():
    """column-shard LDE -> all-to-all -> row-shard hashing -> cap all-gather == the unsharded commitment"""
    out = _run_workers("_mp_sharded_worker.py", 2)
    assert out["world"] == 2 and all(ok_cap and ok_rows for _, ok_cap, ok_rows in out["results"])


def test_sharded_commit_gloo_world4():
    out = _run_workers("_mp_sharded_worker.py", 4)
    assert out["world"] == 4 and all(ok_cap and ok_rows for _, ok_cap, ok_rows in out["results"])


def test_sharded_column_blocks():
    from vectorx_amd import sharded
    per, blocks = sharded.column_blocks(135, 4)
    assert per == 34 and blocks == [(0, 34), (34, 68), (68, 102), (102, 135)]
    per, blocks = sharded.column_blocks(20, 8)
    assert per == 3 and blocks[6] == (18, 20) and blocks[7] == (20, 20)     # trailing ranks own only padding


def test_mapreduce_dag_is_partition_independent():
    """4 map + 3 reduce + 1 outer proofs: the root digest must not depend on how the jobs are spread over ranks
    (world 1 vs world 2), every proof must verify, and the per-layer barriers must see every job exactly once."""
    one = _run_workers("_mp_dag_worker.py", 1)
    two = _run_workers("_mp_dag_worker.py", 2)
    assert one["proofs"] == two["proofs"] == 8
    assert one["layers"] == [["map", 4], ["reduce", 2], ["reduce", 1], ["outer", 1]]
    assert one["root"] == two["root"]
    assert one["per_rank"] == [8]
    assert sum(two["per_rank"]) == 8 and min(two["per_rank"]) >= 3, two["per_rank"]   # round-robin per layer: 5 + 3


def test_dependency_driven_schedule_gives_the_same_dag():
    """barriers=False (one process): a job starts when ITS children are done, lanes pull from one ready queue.  Same proofs and root
    as the layered schedule, every job proven exactly once, no parent before its children — checked with a recording fake prover
    whose "proof" is a hash of the job and its public inputs, so any wrong or early input would change the root."""
    import hashlib
    import threading
    import time as _time

    from vectorx_amd import mapreduce as mr

    class FakeProver:
        log = []
        lock = threading.Lock()

        def __init__(self, kind, log_n, jobs):
            self.kind, self.jobs = kind, set(jobs)

        def prove(self, key, pi, lane=0):
            assert key in self.jobs
            _time.sleep(0.001 * ((key[0] * 7 + key[1] * 3) % 5))          # jobs finish out of order
            with FakeProver.lock:
                FakeProver.log.append(key)
            return hashlib.sha256(self.kind.encode() + repr(key).encode() + bytes(memoryview(pi))).digest() * 3

    spec = mr.DagSpec(num_map=16, map_log_n=4, reduce_log_n=3, outer_log_n=4)
    make = lambda kind, log_n, jobs: FakeProver(kind, log_n, jobs)                 # noqa: E731
    layered = mr.run_dag(spec, make, None, in_flight=3, input_seed=b"x")
    FakeProver.log.clear()
    free = mr.run_dag(spec, make, None, in_flight=3, input_seed=b"x", barriers=False)
    assert free["root"] == layered["root"] and free["my_proofs"] == layered["my_proofs"] and free["proofs"] == 32
    order = {k: i for i, k in enumerate(FakeProver.log)}
    assert len(order) == 32 == len(FakeProver.log)
    layers = spec.layers()
    for (li, j), at in order.items():
        if li == 0:
            continue
        kids = [0] if len(layers[li - 1][1]) == 1 else [2 * j, 2 * j + 1]
        assert all(order[(li - 1, c)] < at for c in kids), (li, j)
    assert any(order[(1, j)] < max(order[(0, m)] for m in range(16)) for j in range(8))   # a reduce proof ran before the last map proof
    assert mr.run_dag(spec, make, None, in_flight=1, input_seed=b"x", barriers=False)["root"] == layered["root"]
    assert mr.run_dag(spec, make, None, in_flight=3, input_seed=b"y", barriers=False)["root"] != layered["root"]


def test_dependency_driven_schedule_with_real_proofs(oracle):
    """the same with the oracle prover (real plonky2-style proofs, each verified): layered and dependency-driven roots are equal"""
    import threading

    import _mp_dag_worker as w
    from vectorx_amd import mapreduce as mr
    oracle.L.vxo_set_num_threads(1)
    try:
        spec = mr.DagSpec(num_map=4, map_log_n=4, reduce_log_n=3, outer_log_n=4)
        lock = threading.Lock()

        class Locked(w.OracleProver):                      # the generator keeps the current public inputs: one caller at a time
            def prove(self, key, pi, lane=0):
                with lock:
                    return super().prove(key, pi)

        make = lambda kind, log_n, jobs: Locked(oracle, kind, log_n, jobs)               # noqa: E731
        a = mr.run_dag(spec, make, None)
        b = mr.run_dag(spec, make, None, in_flight=3, barriers=False)
        assert a["root"] == b["root"] and a["my_proofs"] == b["my_proofs"] and b["proofs"] == 8
    finally:
        import os
        oracle.L.vxo_set_num_threads(os.cpu_count() or 1)      # the library is shared by every later test of this process


def test_dag_spec_header_range_512():
    from vectorx_amd.mapreduce import DagSpec
    s = DagSpec()
    assert s.num_proofs() == 128 and [len(j) for _, j in s.layers()] == [64, 32, 16, 8, 4, 2, 1, 1]


def test_torch_allgather_gloo():
    """the in-place all-gather vx_prove_sharded calls back into, host-staged over gloo (world 2 and 4)"""
    for world in (2, 4):
        out = _run_workers("_mp_allgather_worker.py", world)
        assert out["world"] == world
        assert all(ok for _, ok, _, _ in out["results"])
        assert all(calls == 3 and nbytes == (4 + 64 + 1000) * 8 * (world - 1) for _, _, calls, nbytes in out["results"])


def test_a_job_is_its_plonky2_proof_followed_by_its_stark_proofs():
    """vectorx_amd/mapreduce.py::prove_with_tables: the composition GpuProver.prove uses — order, digest coverage, per-kind seconds"""
    import hashlib
    import threading

    from vectorx_amd import mapreduce as mr

    class Table:
        def __init__(self, blob):
            self.blob, self.seen = blob, []

        def prove(self, ctx, job=None):
            self.seen.append((ctx, job))
            return self.blob

        def take_spent(self, ctx):          # a table may report finer-grained work: trace generation next to proving
            return [("trace_generation", 0.0)]

    a, b = Table(b"AA"), Table(b"B")
    split, lock, mine = {}, threading.Lock(), []
    key = ("map", 0, 5, b"seed")
    job = mr.prove_with_tables(lambda: b"main", [("blake2b", a), ("sha256", b)], "lane-1", split, lock, job=key, spent_out=mine)
    assert job == b"mainAAB" and a.seen == [("lane-1", key)] and set(split) == {"plonky2", "blake2b", "sha256", "trace_generation"}
    assert [k for k, _ in mine] == ["plonky2", "trace_generation", "blake2b", "trace_generation", "sha256"]
    assert mr.prove_with_tables(lambda: b"main", [], None) == b"main"
    # a parent's public inputs change when a child's STARK proof changes
    d0 = hashlib.sha256(job).digest()
    d1 = hashlib.sha256(mr.prove_with_tables(lambda: b"main", [("blake2b", Table(b"AX")), ("sha256", b)], None)).digest()
    assert (mr.child_inputs(1, 0, {0: d0, 1: d0}) != mr.child_inputs(1, 0, {0: d1, 1: d0})).any()
