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
