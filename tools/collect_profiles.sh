#!/bin/bash
# Collects, on the GPU box, every measurement the files under profiles/ are made from (one gpurun call):
#   tools/collect_profiles.sh <outdir under gpurun_out/>
# rocprofv3 runs get the program directly after `--`; counters are collected in their own runs (kernel trace only).
set -u
OUT=${1:-gpurun_out/profiles}
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
python3 bench.py --log-n 20 --steps 20 --warmup 5 --no-cpu-baseline --no-host-witness-leg --no-dag-leg > "$OUT/bench_n1_2p20.json" 2> "$OUT/bench_n1_2p20.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-witness-leg --no-dag-leg > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof_stats.err"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_sq" -- python3 bench.py --workload commit --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_sq.json" 2> "$OUT/pmc_sq.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --workload commit --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --workload commit --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err"
python3 tools/dag_bench.py --in-flight 3 > "$OUT/dag_512.json" 2> "$OUT/dag_512.err"
./tools/ubench_int.bin > "$OUT/ubench_int.md" 2> "$OUT/ubench_int.err"
find "$OUT" -name "*.csv" -size +20M -delete   # per-dispatch traces of the long runs are not kept
ls -R "$OUT" | head -80
