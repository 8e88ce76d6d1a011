#!/bin/bash
set -u
OUT=gpurun_out/r03f
mkdir -p "$OUT"
export TMPDIR=/tmp
export VX_JIT_CACHE_DIR=$PWD/.jit_cache
python3 -m pytest tests/test_gpu_prover.py tests/test_gpu_sharded.py tests/test_gpu_verify.py -m gpu -x -q -k "lookup or randomized or quotient_degree" > "$OUT/pytest_lookup.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest_lookup.log"; tail -4 "$OUT/pytest_lookup.log"
for fl in 16 29; do python3 bench.py --log-n 20 --circuit-flags $fl --steps 4 --warmup 2 --no-cpu-baseline --no-host-witness-leg --no-dag-leg > "$OUT/bench_flags$fl.json" 2> "$OUT/bench_flags$fl.err"; python3 -c "
import json,sys; d=json.loads(open('$OUT/bench_flags$fl.json').read().strip().splitlines()[-1]); print($fl, d['ms_per_step'], {k:v for k,v in d['stage_ms_per_step'].items() if 'quotient' in k})"; done
timeout 500 python3 tools/soak_differential.py 300 4242 5 11 > "$OUT/soak.jsonl" 2> "$OUT/soak.err"; tail -1 "$OUT/soak.jsonl" | cut -c1-300
