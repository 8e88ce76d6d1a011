#!/bin/bash
# Round 4: leaf hashing of SMALL traces — thread-per-row (VX_COOP_LEAF_MAX_ROWS=0) against 16 lanes per row (default), chip tables of 2^9 .. 2^14 rows.
# Output: gpurun_out/r04_small_trace_latency.jsonl  (one line per table x mode)
export VX_JIT_CACHE_DIR=$PWD/.jit_cache
out=gpurun_out/r04_small_trace_latency.jsonl
: > $out
for n in 10 11 12 13 14; do
  for mode in 0 16384 32768; do
    VX_COOP_LEAF_MAX_ROWS=$mode python tools/stark_bench.py --air sha256 --log-n $n --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'air': 'sha256', 'log_n': $n, 'coop_max_rows': $mode, 'ms_per_proof': round(d['ms_per_proof'], 3), 'hash_leaves_ms': d['stage_ms_per_proof'].get('hash_leaves'), 'merkle_levels_ms': d['stage_ms_per_proof'].get('merkle_levels'), 'columns': d['config']['columns']}))" >> $out
  done
done
cat $out
