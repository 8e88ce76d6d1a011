#!/bin/bash
set -u
OUT=gpurun_out/r03g
mkdir -p "$OUT"
export TMPDIR=/tmp
for rep in 1 2; do
for cfg in "1500" "1" "3000" "100000"; do
  VX_JIT_GATE_GROUP_INS=$cfg python3 bench.py --log-n 20 --circuit-flags 29 --steps 4 --warmup 2 --no-cpu-baseline --no-host-witness-leg --no-dag-leg > "$OUT/b.json" 2> "$OUT/b.err"
  python3 -c "
import json,sys; d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]); print('group_ins', $cfg, round(d['ms_per_step'],1), {k:v for k,v in d['stage_ms_per_step'].items() if 'quotient' in k})" | tee -a "$OUT/groups.log"
done; done
VX_JIT_GATE_GROUP_INS=1500 python3 bench.py --log-n 20 --circuit-flags 61 --steps 4 --warmup 2 --no-cpu-baseline --no-host-witness-leg --no-dag-leg > "$OUT/b61.json" 2> "$OUT/b61.err"; python3 -c "
import json,sys; d=json.loads(open('$OUT/b61.json').read().strip().splitlines()[-1]); print('flags61 grouped', round(d['ms_per_step'],1), {k:v for k,v in d['stage_ms_per_step'].items() if 'quotient' in k})" | tee -a "$OUT/groups.log"
python3 -m pytest tests/test_gpu_prover.py -m gpu -x -q -k "program or u32" > "$OUT/pytest.log" 2>&1; tail -3 "$OUT/pytest.log"
