#!/bin/bash
set -u
OUT=gpurun_out/r03b
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 -m pytest tests/test_gpu_prover.py tests/test_gpu_stark.py tests/test_gpu_sharded.py -m gpu -x -q -k "u32 or sha256 or bench or quotient or proof_bytes_identical_to_oracle" > "$OUT/pytest_new.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest_new.log"
tail -8 "$OUT/pytest_new.log"
python3 tools/dev_variant_bench.py vectorx_amd/libvxprover.so variants/q_p6g5.so variants/q_p10g5.so variants/q_p8g4.so variants/q_p8g6.so > "$OUT/variants.log" 2>&1
cat "$OUT/variants.log"
python3 tools/stark_bench.py --air sha256 --log-n 13 --steps 3 --warmup 1 --check > "$OUT/stark_sha256_13.json" 2> "$OUT/stark_sha256_13.err"; tail -3 "$OUT/stark_sha256_13.err"; cat "$OUT/stark_sha256_13.json"
( time python3 bench.py --steps 5 --warmup 2 ) > "$OUT/bench_dag.json" 2> "$OUT/bench_dag.err"; tail -4 "$OUT/bench_dag.err"
