#!/usr/bin/env python3
"""bench.py's rotate leg alone: one JSON line.   python tools/rotate_bench.py [--small]"""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    cache = ROOT / ".jit_cache"
    if "VX_JIT_CACHE_DIR" not in os.environ and cache.is_dir():
        os.environ["VX_JIT_CACHE_DIR"] = str(cache)
    import bench_prove
    import vectorx_amd as vx
    small = "--small" in sys.argv
    ctx = vx.Context(0)
    print(json.dumps(bench_prove.rotate_leg(ctx, 0, log_n=11 if small else 19, small=small)), flush=True)


if __name__ == "__main__":
    main()
