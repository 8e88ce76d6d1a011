#!/bin/bash
# Round 4: every measurement the r04_* files under profiles/ are made from, in ONE gpurun call:
#   tools/collect_r04.sh <outdir under gpurun_out/>
# rocprofv3 runs get the program directly after `--`; counters are collected in their own runs (kernel trace only).
set -u
OUT=${1:-gpurun_out/r04}
mkdir -p "$OUT"
export TMPDIR=/tmp
export VX_JIT_CACHE_DIR=$PWD/.jit_cache
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
python3 bench.py --log-n 20 --steps 20 --warmup 5 --no-cpu-baseline --no-host-witness-leg --no-dag-leg --no-chip-leg > "$OUT/bench_n1_2p20.json" 2> "$OUT/bench_n1_2p20.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-witness-leg --no-dag-leg --no-chip-leg > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof_stats.err"
PMC="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$OUT/pmc_sq_prove" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-witness-leg --no-dag-leg --no-chip-leg > "$OUT/pmc_sq_prove.json" 2> "$OUT/pmc_sq_prove.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --workload commit --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --workload commit --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err"
# the STARK path: the batched EdDSA table, the byte BLAKE2b table, the round-3 chip tables, kernel stats of one EdDSA proof
python3 tools/stark_bench.py --air eddsa --log-n 20 --steps 3 --warmup 1 --check >> "$OUT/stark_bench.jsonl" 2>> "$OUT/stark_bench.err"
python3 tools/stark_bench.py --air blake2b_bytes --log-n 16 --steps 3 --warmup 1 >> "$OUT/stark_bench.jsonl" 2>> "$OUT/stark_bench.err"
for air in sha256 blake2b ed25519; do python3 tools/stark_bench.py --air $air --log-n 13 --steps 3 --warmup 1 --check >> "$OUT/stark_bench.jsonl" 2>> "$OUT/stark_bench.err"; done
python3 tools/stark_bench.py --air blake2b --log-n 18 --steps 3 --warmup 1 >> "$OUT/stark_bench.jsonl" 2>> "$OUT/stark_bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_eddsa" -- python3 tools/stark_bench.py --air eddsa --log-n 20 --steps 3 --warmup 1 > "$OUT/eddsa_under_rocprof.json" 2> "$OUT/rocprof_eddsa.err"
python3 tools/dag_starks_bench.py --in-flight 3 > "$OUT/dag_starks.jsonl" 2> "$OUT/dag_starks.err"
python3 tools/dag_bench.py --in-flight 3 > "$OUT/dag_512.json" 2> "$OUT/dag_512.err"
bash tools/small_trace_latency.sh > /dev/null 2>&1; cp gpurun_out/r04_small_trace_latency.jsonl "$OUT/" 2>/dev/null
python3 tools/sharded_prove_bench.py 21 1,2,4,8 dev > "$OUT/sharded_prove_bench_21.jsonl" 2> "$OUT/sharded_prove_bench_21.err"
timeout 500 python3 tools/soak_stark.py 240 41337 12 > "$OUT/soak_stark.jsonl" 2> "$OUT/soak_stark.err"
timeout 900 python3 tools/soak_differential.py 600 4242 > "$OUT/soak_differential.jsonl" 2> "$OUT/soak_differential.err"
find "$OUT" -name "*.csv" -size +20M -delete   # per-dispatch traces of the long runs are not kept
ls -R "$OUT" | head -80
