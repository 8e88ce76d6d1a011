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
