#!/bin/bash
set -u
OUT=gpurun_out/r03d
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_prover.py -m gpu -x -q -k "ntt or polynomial or proof_bytes_identical_to_oracle or large" > "$OUT/pytest_ntt.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest_ntt.log"; tail -4 "$OUT/pytest_ntt.log"
python3 tools/dev_variant_bench.py vectorx_amd/libvxprover.so variants/ntt_nonpersistent.so vectorx_amd/libvxprover.so variants/ntt_nonpersistent.so > "$OUT/variants.log" 2>&1
VX_NTT_NO_PERSIST=1 python3 tools/dev_variant_bench.py vectorx_amd/libvxprover.so >> "$OUT/variants.log" 2>&1
VX_NTT_BLOCKS_PER_CU=1 python3 tools/dev_variant_bench.py vectorx_amd/libvxprover.so >> "$OUT/variants.log" 2>&1
VX_NTT_BLOCKS_PER_CU=4 python3 tools/dev_variant_bench.py vectorx_amd/libvxprover.so >> "$OUT/variants.log" 2>&1
cat "$OUT/variants.log"
python3 bench.py --steps 5 --warmup 2 --no-dag-leg --no-host-witness-leg --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; tail -2 "$OUT/bench.err"
