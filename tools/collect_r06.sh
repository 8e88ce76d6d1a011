#!/bin/bash
# Collects, on the GPU box, the measurements the round-6 files under profiles/ are made from:
#   tools/collect_r06.sh a|b <outdir under gpurun_out/>      (two gpurun calls: a = the bench line, kernel statistics, counters; b = the rest)
# rocprofv3 runs get the program directly after `--`; counters are collected in their own runs (kernel trace only).
set -u
PART=${1:-a}
OUT=${2:-gpurun_out/r06_profiles}
mkdir -p "$OUT"
export TMPDIR=/tmp
export VX_JIT_CACHE_DIR=$PWD/.jit_cache
NOLEGS="--no-cpu-baseline --no-host-witness-leg --no-dag-leg --no-chip-leg --no-rotate-leg"
PMC="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
if [ "$PART" = a ]; then
  python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
  cp gpurun_out/bench_line_full.json "$OUT/bench_n1_full.json" 2>/dev/null
  python3 bench.py --log-n 20 --steps 20 --warmup 5 $NOLEGS > "$OUT/bench_n1_2p20.json" 2> "$OUT/bench_n1_2p20.err"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 5 --warmup 2 $NOLEGS > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof_stats.err"
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$OUT/pmc_sq_prove" -- python3 bench.py --steps 1 --warmup 0 $NOLEGS > "$OUT/pmc_sq_prove.json" 2> "$OUT/pmc_sq_prove.err"
  # the DAG's circuit family as the workload: the recursive verifier's gate set in its declared mix, at the outer size (2^19 rows)
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$OUT/pmc_sq_prove_recursion" -- python3 bench.py --log-n 19 --recursion-mix --steps 1 --warmup 0 $NOLEGS > "$OUT/pmc_sq_prove_recursion.json" 2> "$OUT/pmc_sq_prove_recursion.err"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_recursion" -- python3 bench.py --log-n 19 --recursion-mix --steps 5 --warmup 2 $NOLEGS > "$OUT/bench_recursion_under_rocprof.json" 2> "$OUT/rocprof_stats_recursion.err"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --workload commit --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --workload commit --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err"
else
  python3 tools/small_proof_profile.py 14 16 18 19 > "$OUT/small_proof_profile.jsonl" 2> "$OUT/small_proof_profile.err"
  python3 tools/small_proof_profile.py --recursion 16 18 19 > "$OUT/small_proof_profile_recursion.jsonl" 2>> "$OUT/small_proof_profile.err"
  python3 tools/small_proof_profile.py --recursion --per-gate 16 18 19 > "$OUT/small_proof_profile_recursion_per_gate.jsonl" 2>> "$OUT/small_proof_profile.err"
  python3 tools/tracegen_bench.py > "$OUT/tracegen.jsonl" 2> "$OUT/tracegen.err"
  python3 tools/dag_pool_bench.py 3 3 > "$OUT/dag_pool.jsonl" 2> "$OUT/dag_pool.err"
  python3 tools/dag_pool_bench.py --no-recursion 3 3 > "$OUT/dag_pool_two_gate_stand_in.jsonl" 2>> "$OUT/dag_pool.err"
  python3 tools/rotate_bench.py > "$OUT/rotate_leg.json" 2> "$OUT/rotate_leg.err"
  python3 tools/sharded_prove_bench.py 21 1,2,4,8 dev > "$OUT/sharded_prove_bench_21.jsonl" 2> "$OUT/sharded_prove_bench_21.err"
  VX_SHARD_REPLICATE_OPENINGS=1 python3 tools/sharded_prove_bench.py 21 8 dev > "$OUT/sharded_prove_bench_21_openings_replicated.jsonl" 2>> "$OUT/sharded_prove_bench_21.err"
  timeout 400 python3 tools/soak_differential.py 300 2026 > "$OUT/soak_differential.jsonl" 2> "$OUT/soak_differential.err"
  timeout 300 python3 tools/soak_stark.py 200 31337 12 > "$OUT/soak_stark.jsonl" 2> "$OUT/soak_stark.err"
fi
find "$OUT" -name "*.csv" -size +20M -delete   # per-dispatch traces of the long runs are not kept
ls -R "$OUT" | head -80
