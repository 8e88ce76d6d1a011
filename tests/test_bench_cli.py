"""bench.py's launch contract, the parts that need no GPU: `--gpus N` is never silently ignored."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _run(args, **env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, timeout=300, env=env, cwd=str(ROOT))


def test_more_gpus_than_visible_is_refused_without_a_result_line():
    import torch
    n = torch.cuda.device_count()
    r = _run(["--gpus", str(n + 2), "--log-n", "10"])
    assert r.returncode != 0
    assert "refusing" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_must_match_gpus_flag():
    r = _run(["--gpus", "4", "--log-n", "10"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0
    assert "WORLD_SIZE=2" in r.stderr and not r.stdout.strip()


def test_extra_legs_import_without_a_gpu_and_the_chip_harness_knows_its_tables():
    """The bench's extra legs (host witness, DAG, chip STARKs) live in bench_prove.py / vectorx_amd/stark_chips.py: importable on a
    CPU-only machine (a syntax error there would cost the driver's whole bench line), and the harness names the three chip tables."""
    sys.path.insert(0, str(ROOT))
    import bench_prove
    from vectorx_amd import stark_chips
    assert callable(bench_prove.dag_leg) and callable(bench_prove.chip_leg) and callable(bench_prove.host_witness_leg)
    assert stark_chips.CHIPS == ("sha256", "blake2b", "ed25519")
    r = _run(["--help"])
    assert r.returncode == 0 and "--no-chip-leg" in r.stdout and "--no-dag-leg" in r.stdout
