#!/bin/bash
set -u
OUT=gpurun_out/r03i
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 -m pytest tests/test_gpu_prover.py tests/test_gpu_sharded.py tests/test_gpu_boundary.py -m gpu -x -q -k "proof_bytes_identical_to_oracle or sharded_proof_is_byte or vxcircuit or config_variants or quotient_degree" > "$OUT/pytest.log" 2>&1; tail -3 "$OUT/pytest.log"
python3 tools/dev_variant_bench.py vectorx_amd/libvxprover.so variants/q_no_l0table.so variants/q_desync.so variants/q_desync_dup.so vectorx_amd/libvxprover.so variants/q_no_l0table.so variants/q_desync_dup.so > "$OUT/variants.log" 2>&1; cat "$OUT/variants.log"
