#!/usr/bin/env python3
"""The header_range_512 DAG on the worker pool alone (bench.py's DAG legs without the rest of the bench): one JSON line per
configuration.   python tools/dag_pool_bench.py [--no-recursion] [workers lanes]...   e.g.  2 3  3 2  2 4
--no-recursion: the two-gate stand-in circuits of rounds 1-5 instead of the recursive verifier's gate set (mapreduce.DagSpec.recursion)."""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import bench_prove
    from vectorx_amd import mapreduce as mr
    from vectorx_amd.dag_pool import DagPool
    cache = ROOT / ".jit_cache"
    if "VX_JIT_CACHE_DIR" not in os.environ and cache.is_dir():
        os.environ["VX_JIT_CACHE_DIR"] = str(cache)
    recursion = "--no-recursion" not in sys.argv[1:]
    nums = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [2, 3]
    for w, k in zip(nums[0::2], nums[1::2]):
        pool = DagPool(mr.DagSpec(64, 18, 16, 19, recursion=recursion), devices=(0,), workers_per_device=w, lanes=k, with_starks=True, table_mode="per_job").start()
        try:
            out = bench_prove.dag_pool_legs(pool, with_starks=True)
        finally:
            pool.close()
        for name, rec in out.items():
            print(json.dumps({"leg": name, "workers": w, "lanes": k, "recursion_mix": recursion, **{x: rec[x] for x in ("dag_seconds", "dag_seconds_all_passes", "lane_seconds_by_kind",
                              "per_layer_ms", "per_layer_ms_layer_barriers", "jobs_by_worker", "setup_seconds_untimed", "root", "request_load_seconds_untimed", "input", "output",
                              "output_equals_host_computation") if x in rec}}), flush=True)


if __name__ == "__main__":
    main()
