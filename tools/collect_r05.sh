#!/bin/bash
# Collects, on the GPU box, the measurements the round-5 files under profiles/ are made from (one gpurun call):
#   tools/collect_r05.sh <outdir under gpurun_out/>
# rocprofv3 runs get the program directly after `--`; counters are collected in their own runs (kernel trace only).
set -u
OUT=${1:-gpurun_out/r05_profiles}
mkdir -p "$OUT"
export TMPDIR=/tmp
export VX_JIT_CACHE_DIR=$PWD/.jit_cache
python3 bench.py --steps 20 --warmup 5 --cpu-baseline-full > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
cp gpurun_out/cpu_full_size.json "$OUT/cpu_full_size.json" 2>/dev/null
python3 bench.py --log-n 20 --steps 20 --warmup 5 --no-cpu-baseline --no-host-witness-leg --no-dag-leg --no-chip-leg > "$OUT/bench_n1_2p20.json" 2> "$OUT/bench_n1_2p20.err"
NOLEGS="--no-cpu-baseline --no-host-witness-leg --no-dag-leg --no-chip-leg"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 5 --warmup 2 $NOLEGS > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof_stats.err"
PMC="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$OUT/pmc_sq_prove" -- python3 bench.py --steps 1 --warmup 0 $NOLEGS > "$OUT/pmc_sq_prove.json" 2> "$OUT/pmc_sq_prove.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --workload commit --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --workload commit --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err"
python3 tools/small_proof_profile.py 14 16 18 19 > "$OUT/small_proof_profile.jsonl" 2> "$OUT/small_proof_profile.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_small" -- python3 tools/small_proof_profile.py 16 > /dev/null 2> "$OUT/rocprof_small.err"
python3 tools/tracegen_bench.py > "$OUT/tracegen.jsonl" 2> "$OUT/tracegen.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_tracegen" -- python3 tools/tracegen_bench.py > /dev/null 2> "$OUT/rocprof_tracegen.err"
python3 tools/dag_pool_bench.py 3 2 2 3 4 2 3 3 > "$OUT/dag_pool.jsonl" 2> "$OUT/dag_pool.err"
python3 tools/sharded_prove_bench.py 21 1,2,4,8 dev > "$OUT/sharded_prove_bench_21.jsonl" 2> "$OUT/sharded_prove_bench_21.err"
timeout 500 python3 tools/soak_differential.py 300 2025 > "$OUT/soak_differential.jsonl" 2> "$OUT/soak_differential.err"
timeout 300 python3 tools/soak_stark.py 200 31337 12 > "$OUT/soak_stark.jsonl" 2> "$OUT/soak_stark.err"
find "$OUT" -name "*.csv" -size +20M -delete   # per-dispatch traces of the long runs are not kept
ls -R "$OUT" | head -80
