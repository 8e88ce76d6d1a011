export VX_JIT_CACHE_DIR=$PWD/.jit_cache
for lib in vectorx_amd/libvxprover.so vectorx_amd/libvxprover_directstore.so vectorx_amd/libvxprover.so vectorx_amd/libvxprover_directstore.so; do
  VXPROVER_LIB=$PWD/$lib python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-witness-leg --no-dag-leg --no-chip-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', round(d['ms_per_step'], 2), 'lde', d['stage_ms_per_step']['lde'], 'intt', d['stage_ms_per_step']['intt'], 'roofline', d['roofline']['frac'], 'quot_intt', d['stage_ms_per_step']['quotient_intt'], 'fri_lde', d['stage_ms_per_step']['fri_lde'])"
done
VXPROVER_LIB=$PWD/vectorx_amd/libvxprover_directstore.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
