set -u
OUT=gpurun_out/r04e
mkdir -p "$OUT"
export TMPDIR=/tmp
export VX_JIT_CACHE_DIR=$PWD/.jit_cache
python3 bench.py --log-n 20 --steps 20 --warmup 5 --no-cpu-baseline --no-host-witness-leg --no-dag-leg --no-chip-leg --no-dag-stark-leg > "$OUT/bench_n1_2p20.json" 2> "$OUT/bench_n1_2p20.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-witness-leg --no-dag-leg --no-chip-leg --no-dag-stark-leg > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof_stats.err"
PMC="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$OUT/pmc_sq_prove" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-witness-leg --no-dag-leg --no-chip-leg --no-dag-stark-leg > "$OUT/pmc_sq_prove.json" 2> "$OUT/pmc_sq_prove.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --workload commit --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --workload commit --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err"
python3 tools/dag_bench.py --in-flight 3 > "$OUT/dag_512.json" 2> "$OUT/dag_512.err"
python3 tools/dag_starks_bench.py --in-flight 3 > "$OUT/dag_starks.jsonl" 2> "$OUT/dag_starks.err"
python3 tools/sharded_prove_bench.py 21 1,2,4,8 dev > "$OUT/sharded_prove_bench_21.jsonl" 2> "$OUT/sharded_prove_bench_21.err"
find "$OUT" -name "*.csv" -size +20M -delete
ls -R "$OUT" | head -40
