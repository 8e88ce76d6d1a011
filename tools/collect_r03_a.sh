#!/bin/bash
# round 3, first GPU call: full -m gpu suite, baseline bench, SQ counters on the PROVE workload
set -u
OUT=gpurun_out/r03a
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
tail -5 "$OUT/pytest_gpu.log"
python3 bench.py --steps 10 --warmup 3 > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
PMC="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$OUT/pmc_sq_prove" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-witness-leg > "$OUT/pmc_sq_prove.json" 2> "$OUT/pmc_sq_prove.err"
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$OUT/pmc_sq_prove_flags" -- python3 bench.py --log-n 20 --circuit-flags 29 --steps 1 --warmup 0 --no-cpu-baseline --no-host-witness-leg > "$OUT/pmc_sq_prove_flags.json" 2> "$OUT/pmc_sq_prove_flags.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-witness-leg > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof_stats.err"
find "$OUT" -name "*.csv" -size +20M -delete
ls -R "$OUT" | head -60
