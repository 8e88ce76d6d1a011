"""The four chip-sized AIRs against tests/golden/chip_goldens.json (restatement-generated, see tests/golden/make_chip_goldens.py):
program words, trace, public inputs and the oracle's proof bytes on fixed inputs must not drift."""
import importlib.util
import json
from pathlib import Path

GOLD = Path(__file__).resolve().parent / "golden"


def test_chip_programs_traces_and_oracle_proofs_match_the_committed_goldens(oracle):
    spec = importlib.util.spec_from_file_location("make_chip_goldens", GOLD / "make_chip_goldens.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.loads((GOLD / "chip_goldens.json").read_text())
    got = mod.compute()
    assert set(got) == set(want)
    for name in got:
        assert got[name] == want[name], name
