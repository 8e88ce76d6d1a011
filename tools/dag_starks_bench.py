#!/usr/bin/env python3
"""The header_range_512 DAG WITH its STARK tables on one GPU (bench.py's `dag_header_range_512_with_starks` leg on its own):
    python tools/dag_starks_bench.py [--in-flight K ...]      one JSON line per K"""
import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--in-flight", type=int, nargs="+", default=[3])
    args = ap.parse_args()
    import bench_prove
    import vectorx_amd as vx
    for k in args.in_flight:
        ctx = vx.Context(0)
        r = bench_prove.dag_with_starks_leg(ctx, 0, in_flight=k)
        r.pop("what", None)
        r.pop("tables", None)
        print(json.dumps(r), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
