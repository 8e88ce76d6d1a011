import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


import os

_JIT_CACHE = ROOT / ".jit_cache"   # filled by __graft_entry__.build() (vx_stark_precompile): compiled AIR chunks, loaded instead of recompiled
if "VX_JIT_CACHE_DIR" not in os.environ and _JIT_CACHE.is_dir():
    os.environ["VX_JIT_CACHE_DIR"] = str(_JIT_CACHE)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def ctx():
    import vectorx_amd as vx
    c = vx.Context(0)  # raises VxError (no fallback) when the HIP library or the GPU is missing
    yield c
    c.close()


# ---- the two full-size oracle proofs, computed in the background of a GPU run ------------------------------------------------------
# (tests/_bg_oracle.py): started once the collection shows that their tests will run, joined by `background_oracle_proof`.
BIG_ORACLE_JOBS = {"test_header_range_512_sized_proof_bytes_identical_to_oracle": (21, 0x5EED0000, 50),
                   "test_header_range_256_sized_proof_bytes_identical_to_oracle": (20, 20, 50)}
_BG = {"proc": None, "dir": None}


def pytest_collection_finish(session):
    if os.environ.get("VX_NO_BACKGROUND_ORACLE") or _BG["proc"] is not None:
        return
    jobs = [spec for item in session.items for name, spec in BIG_ORACLE_JOBS.items() if item.name == name]
    if not jobs:
        return
    import subprocess
    import tempfile
    _BG["dir"] = tempfile.mkdtemp(prefix="vx_bg_oracle_")
    # (in the order the tests will ask for them: the 2^20 proof is needed in the middle of the run, the 2^21 one by the last test)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    _BG["proc"] = subprocess.Popen([sys.executable, str(ROOT / "tests" / "_bg_oracle.py"), _BG["dir"]] + [f"{a}:{b}:{c}" for a, b, c in jobs],
                                   env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def pytest_sessionfinish(session, exitstatus):
    p = _BG["proc"]
    if p is not None and p.poll() is None:
        p.kill()
    if _BG["dir"]:
        import shutil
        shutil.rmtree(_BG["dir"], ignore_errors=True)


@pytest.fixture
def background_oracle_proof():
    """-> f(log_n, seed) = (proof bytes, record) once the background worker has them, or None when there is no worker (the test then asks
    the oracle itself).  Waits as long as the worker is alive."""
    import json
    import time

    def get(log_n, seed):
        p, d = _BG["proc"], _BG["dir"]
        if p is None or d is None:
            return None
        binf, jsf = Path(d) / f"oracle_proof_{log_n}_{seed}.bin", Path(d) / f"oracle_proof_{log_n}_{seed}.json"
        while not binf.exists():
            if p.poll() is not None and not binf.exists():
                return None                                              # the worker is gone without this result
            time.sleep(0.5)
        return binf.read_bytes(), json.loads(jsf.read_text())
    return get
