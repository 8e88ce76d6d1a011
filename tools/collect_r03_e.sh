#!/bin/bash
set -u
OUT=gpurun_out/r03e
mkdir -p "$OUT"
export TMPDIR=/tmp
export VX_JIT_CACHE_DIR=$PWD/.jit_cache
python3 -m pytest tests/test_gpu_prover.py tests/test_gpu_sharded.py tests/test_gpu_boundary.py tests/test_gpu_stark.py -m gpu -x -q -k "program or u32 or lookup or quotient_degree or randomized or vxcircuit or function or sha256 or bus or interpreted or compiled" > "$OUT/pytest_prog.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest_prog.log"; tail -4 "$OUT/pytest_prog.log"
for fl in 29 61; do python3 bench.py --log-n 20 --circuit-flags $fl --steps 4 --warmup 2 --no-cpu-baseline --no-host-witness-leg --no-dag-leg > "$OUT/bench_flags$fl.json" 2> "$OUT/bench_flags$fl.err"; python3 -c "
import json,sys; d=json.loads(open('$OUT/bench_flags$fl.json').read().strip().splitlines()[-1]); print($fl, d['ms_per_step'], {k:v for k,v in d['stage_ms_per_step'].items() if 'quotient' in k})"; done
PMC="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$OUT/pmc_flags" -- python3 bench.py --log-n 20 --circuit-flags 29 --steps 1 --warmup 0 --no-cpu-baseline --no-host-witness-leg --no-dag-leg > "$OUT/pmc_flags.json" 2> "$OUT/pmc_flags.err"
timeout 900 python3 tools/soak_differential.py 600 777 3 12 > "$OUT/soak.jsonl" 2> "$OUT/soak.err"; tail -1 "$OUT/soak.jsonl" | cut -c1-600
find "$OUT" -name "*.csv" -size +20M -delete
