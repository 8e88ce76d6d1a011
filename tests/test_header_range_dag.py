"""The DAG's statements end to end on the CPU: dag_tables.build_per_job with hashlib standing in for the GPU tables
(tests/_cpu_tables.py), the production statement closures, records handed from children to parents by every scheduler — one process
(layered, dependency-driven, several lanes) and two gloo ranks — arrive at the function's 96 output bytes.  The GPU twin with the real
tables: tests/test_gpu_dag_pool.py."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

import _cpu_tables
from vectorx_amd import dag_tables
from vectorx_amd import header_range as hr
from vectorx_amd import mapreduce as mr

ROOT = Path(__file__).resolve().parent.parent


def run(spec, seed, num_headers=None, **kw):
    per_kind, _, _ = dag_tables.build_per_job(None, [None], small=True, num_map=spec.num_map, num_headers=num_headers, factory=_cpu_tables.CpuTables())
    make = lambda kind, log_n, jobs: _cpu_tables.TablesProver(kind, per_kind[kind])      # noqa: E731
    return mr.run_dag(spec, make, None, input_seed=seed, **kw)


@pytest.mark.parametrize("num_headers", [None, 19, 3])
def test_every_scheduler_arrives_at_the_output(num_headers):
    spec = mr.DagSpec(4, 10, 9, 11)
    req = hr.cached_request(b"cpu dag", **dag_tables.request_shape(True, spec.num_map, num_headers))
    want = hr.expected_output(req)
    a = run(spec, b"cpu dag", num_headers)
    assert len(a["root"]) == 32 + 96 and a["root"][32:] == want
    assert run(spec, b"cpu dag", num_headers, in_flight=3)["root"] == a["root"]
    assert run(spec, b"cpu dag", num_headers, in_flight=3, barriers=False)["root"] == a["root"]
    assert run(spec, b"another", num_headers)["root"][32:] != want


def test_a_forged_signature_and_a_broken_chain_stop_the_dag():
    spec = mr.DagSpec(4, 10, 9, 11)
    shape = dag_tables.request_shape(True, spec.num_map)
    bad = hr.cached_request(b"forged", **shape)
    just = bad.justification()
    just.signatures[1] = just.signatures[1][:32] + bytes(32)
    with pytest.raises(hr.StatementError, match="signature bus does not balance"):
        run(spec, b"forged")
    broken = hr.cached_request(b"broken chain", **shape)
    broken.headers[9] = bytes([broken.headers[9][0] ^ 1]) + broken.headers[9][1:]
    with pytest.raises(hr.StatementError, match="not linked"):
        run(spec, b"broken chain")


def test_two_gloo_ranks_hand_the_statements_across():
    """world 2: map / reduce jobs dealt round-robin, records (digest + statement) all-gathered at the layer barriers — the reduce jobs
    of one rank hash roots the other rank stated; same root as one process"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, OMP_NUM_THREADS="1")
    outs = []
    for nproc in (1, 2):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(port + nproc), str(ROOT / "tests" / "_mp_statement_worker.py")]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        outs.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]))
    assert outs[0]["root"] == outs[1]["root"] and outs[1]["world"] == 2 and outs[0]["output_ok"] and outs[1]["output_ok"]
    assert sum(outs[1]["per_rank"]) == 8 and min(outs[1]["per_rank"]) >= 3
