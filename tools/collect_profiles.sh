#!/bin/bash
# Collects, on the GPU box, every measurement the round-3 files under profiles/ are made from (one gpurun call):
#   tools/collect_profiles.sh <outdir under gpurun_out/>
# rocprofv3 runs get the program directly after `--`; counters are collected in their own runs (kernel trace only).
set -u
OUT=${1:-gpurun_out/profiles}
mkdir -p "$OUT"
export TMPDIR=/tmp
export VX_JIT_CACHE_DIR=$PWD/.jit_cache
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
python3 bench.py --log-n 20 --steps 20 --warmup 5 --no-cpu-baseline --no-host-witness-leg --no-dag-leg --no-chip-leg > "$OUT/bench_n1_2p20.json" 2> "$OUT/bench_n1_2p20.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-witness-leg --no-dag-leg --no-chip-leg > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof_stats.err"
PMC="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$OUT/pmc_sq_prove" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-witness-leg --no-dag-leg --no-chip-leg > "$OUT/pmc_sq_prove.json" 2> "$OUT/pmc_sq_prove.err"
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$OUT/pmc_sq_prove_flags" -- python3 bench.py --log-n 20 --circuit-flags 29 --steps 1 --warmup 0 --no-cpu-baseline --no-host-witness-leg --no-dag-leg --no-chip-leg > "$OUT/pmc_sq_prove_flags.json" 2> "$OUT/pmc_sq_prove_flags.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --workload commit --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --workload commit --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err"
python3 tools/dag_bench.py --in-flight 3 > "$OUT/dag_512.json" 2> "$OUT/dag_512.err"
python3 tools/stark_bench.py --air ed25519 --log-n 13 --steps 3 --warmup 1 --check >> "$OUT/stark_bench.jsonl" 2>> "$OUT/stark_bench.err"
for air in sha256 blake2b; do for lg in 13 15; do python3 tools/stark_bench.py --air $air --log-n $lg --steps 3 --warmup 1 --check >> "$OUT/stark_bench.jsonl" 2>> "$OUT/stark_bench.err"; done; done
python3 tools/stark_bench.py --log-n 18 --groups 16 --steps 5 --warmup 2 --check >> "$OUT/stark_bench.jsonl" 2>> "$OUT/stark_bench.err"
python3 tools/sharded_prove_bench.py 21 1,2,4,8 dev > "$OUT/sharded_prove_bench_21.jsonl" 2> "$OUT/sharded_prove_bench_21.err"
timeout 400 python3 tools/soak_stark.py 240 31337 12 > "$OUT/soak_stark.jsonl" 2> "$OUT/soak_stark.err"
python3 tools/full_size_parity.py 21 > "$OUT/full_size_parity.jsonl" 2> "$OUT/full_size_parity.err"
find "$OUT" -name "*.csv" -size +20M -delete   # per-dispatch traces of the long runs are not kept
ls -R "$OUT" | head -80
