#!/bin/bash
set -u
OUT=gpurun_out/r03h
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 -m pytest tests/test_gpu_prover.py -m gpu -x -q -k "jit_code_objects or malformed" > "$OUT/pytest.log" 2>&1; tail -3 "$OUT/pytest.log"
python3 tools/dev_variant_bench.py vectorx_amd/libvxprover.so variants/q_desync.so variants/q_desync_dup.so variants/q_loop_nodesync.so vectorx_amd/libvxprover.so variants/q_desync_dup.so > "$OUT/variants.log" 2>&1; cat "$OUT/variants.log"
python3 bench.py --steps 5 --warmup 2 --no-dag-leg --no-host-witness-leg --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; python3 -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['alu_bound_dominant_kernel'])"
