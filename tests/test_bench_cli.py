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


def test_launcher_keeps_rank0s_line_when_a_rank_dies_after_it(monkeypatch, capsys):
    """`python bench.py --gpus N` run plainly relays rank 0's line; a rank lost AFTER the line (in the multi-rank legs or the teardown)
    must not cost the measurement: the line goes out with the launcher's exit code in it and the launcher returns 0; no line -> the code."""
    import json
    import types
    sys.path.insert(0, str(ROOT))
    import bench
    import torch
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    args = types.SimpleNamespace(gpus=2, ranks_on_one_device=False)
    line = json.dumps({"metric": "header_range_512 proofs/sec", "value": 9.9, "n_gpus": 2})
    monkeypatch.setattr(bench.subprocess, "run", lambda *a, **k: types.SimpleNamespace(returncode=1, stdout="noise\n" + line + "\n"))
    assert bench.launch_ranks(args) == 0
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert out["value"] == 9.9 and out["ranks_exit_code_after_the_line"] == 1
    monkeypatch.setattr(bench.subprocess, "run", lambda *a, **k: types.SimpleNamespace(returncode=7, stdout="noise only\n"))
    assert bench.launch_ranks(args) == 7
    assert not [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")]
    monkeypatch.setattr(bench.subprocess, "run", lambda *a, **k: types.SimpleNamespace(returncode=0, stdout=line + "\n"))
    assert bench.launch_ranks(args) == 0
    assert "ranks_exit_code_after_the_line" not in json.loads(capsys.readouterr().out.strip().splitlines()[-1])


def test_the_line_the_driver_keeps_is_under_6_kb_and_holds_every_headline_number():
    """VERDICT r5 #2b: the driver's record keeps the last 8 KB of stdout, a 16 KB line of prose lost `dag_seconds`, `value_end_to_end`
    and the stage split.  bench.py prints bench_prove.compact_line(line): numbers stay, descriptions live in
    profiles/bench_line_glossary.md.  Checked on a complete line of an earlier run (profiles/r05_bench_n1_default.json, 16 KB)."""
    import json
    sys.path.insert(0, str(ROOT))
    import bench_prove
    full = json.loads((ROOT / "profiles" / "r05_bench_n1_default.json").read_text())
    assert len(json.dumps(full)) > 12000
    c = bench_prove.compact_line(full)
    line = json.dumps(c)
    assert len(line) < 6144
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert c[k] == full[k] or abs(c[k] - full[k]) <= 1e-5 * abs(full[k])
    assert c["config"]["workload"].startswith("header_range_512")
    assert c["roofline"]["frac"] == full["roofline"]["frac"] and c["roofline"]["bound"] == "hbm" and c["roofline"]["traffic"] > 0
    assert c["cpu_baseline"]["cores"] == full["cpu_baseline"]["cores"] and c["cpu_baseline"]["kind"] == "port"
    assert abs(c["value_end_to_end"] - full["value_end_to_end"]) < 1e-4
    assert c["stage_ms_per_step"]["hash_leaves"] == full["stage_ms_per_step"]["hash_leaves"]
    for leg in ("dag_header_range_512", "dag_header_range_512_with_starks"):
        assert abs(c[leg]["dag_seconds"] - full[leg]["dag_seconds"]) < 1e-4 and len(c[leg]["dag_seconds_all_passes"]) == 3
        assert c[leg]["lane_seconds_by_kind"] == full[leg]["lane_seconds_by_kind"]
    assert c["dag_header_range_512_with_starks"]["output_equals_host_computation"] is True
    assert abs(c["rotate"]["seconds"] - full["rotate"]["seconds"]) < 1e-4
    # no prose left: every string is short or one of the contract's
    def strings(o, key=None):
        if isinstance(o, dict):
            for k, v in o.items():
                yield from strings(v, k)
        elif isinstance(o, list):
            for v in o:
                yield from strings(v, key)
        elif isinstance(o, str):
            yield key, o
    assert all(len(v) <= 48 or k in bench_prove._KEEP_TEXT for k, v in strings(c))
    assert (ROOT / c["glossary"]).exists()
    # a line that is already small is left alone (apart from the rounding)
    small = {"metric": "m", "value": 1.23456789, "cpu_baseline": {"value": 0.5, "sample": "x" * 100}}
    cs = bench_prove.compact_line(small)
    assert cs["value"] == 1.23457 and "sample" not in cs["cpu_baseline"] and "dropped_for_size" not in cs
