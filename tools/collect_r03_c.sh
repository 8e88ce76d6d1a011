#!/bin/bash
set -u
OUT=gpurun_out/r03c
mkdir -p "$OUT"
export TMPDIR=/tmp
ls -la .jit_cache | head -3 > "$OUT/jit_cache_perms.txt" 2>&1; id >> "$OUT/jit_cache_perms.txt"
python3 -m pytest tests -m gpu -x -q --durations=15 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
tail -25 "$OUT/pytest_gpu.log"
( time python3 bench.py --steps 10 --warmup 3 ) > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"; tail -4 "$OUT/bench_n1.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-witness-leg --no-dag-leg > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof_stats.err"
PMC="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$OUT/pmc_sq_prove" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-witness-leg --no-dag-leg > "$OUT/pmc_sq_prove.json" 2> "$OUT/pmc_sq_prove.err"
PMC2="SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"
rocprofv3 --pmc $PMC2 --kernel-trace --output-format csv -d "$OUT/pmc_sq_prove2" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-witness-leg --no-dag-leg > "$OUT/pmc_sq_prove2.json" 2> "$OUT/pmc_sq_prove2.err"; tail -3 "$OUT/pmc_sq_prove2.err"
rocprofv3 --list-avail > "$OUT/rocprof_list_avail.txt" 2>&1
./tools/ubench_int.bin > "$OUT/ubench_int.md" 2> "$OUT/ubench_int.err"; tail -12 "$OUT/ubench_int.md"
for lg in 13 15; do python3 tools/stark_bench.py --air sha256 --log-n $lg --steps 3 --warmup 1 --check >> "$OUT/stark_bench.jsonl" 2>> "$OUT/stark_bench.err"; done; cat "$OUT/stark_bench.jsonl" | cut -c1-1500
find "$OUT" -name "*.csv" -size +20M -delete
